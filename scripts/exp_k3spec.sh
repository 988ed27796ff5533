#!/bin/bash
# timing-only: K3 (one tile per workgroup) with the block stream and column indices requested at 256 t, before the tile's descriptor
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/k3spec; mkdir -p $O
for rep in 1 2 3; do
  for sp in 0 1; do
    if [ $sp = 0 ]; then unset PGO_K3_SPEC; else export PGO_K3_SPEC=1; fi
    PGO_LIB=$PWD/toy-robust-backend-slam_amd/libpgo_exp.so timeout -k 10 200 python3 scripts/k3_probe.py -1 > $O/sp_${sp}_$rep.log 2>&1 || { tail -5 $O/sp_${sp}_$rep.log; exit 1; }
    echo "SPEC $sp rep $rep: $(grep -h checksum $O/sp_${sp}_$rep.log | sed 's/.*k_spmv/k_spmv/' | cut -c1-40)"
  done
done
