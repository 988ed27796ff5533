python scripts/exp_phase.py
for c in INTEL:50:1 INTEL:50:0 MIT:0:1 CSAIL:0:1 FR079:0:1; do IFS=: read n o m <<< "$c"; python scripts/small_child.py $n $o $m -1; done
python scripts/small_child.py INTEL 50 1 64
python scripts/small_child.py M3500 0 1 -1
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "chain or precond or lm_solve or dcs or batch" 2>&1 | tail -2
