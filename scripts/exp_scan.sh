for sc in 0 1; do
PGO_CHAIN_SCAN=$sc python scripts/small_child.py INTEL 50 1 256
PGO_CHAIN_SCAN=$sc python scripts/small_child.py MIT 0 1 256
PGO_CHAIN_SCAN=$sc python scripts/small_child.py INTEL 50 1 128
done
