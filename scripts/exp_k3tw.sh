#!/bin/bash
# k_spmv_1 with finer tiles of its own (experiment build, PGO_K3_TW): 256 (default) vs 128 vs 64 threads / incidences per workgroup
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/k3tw
mkdir -p $O
for rep in 1 2; do
  for tw in 0 128 64; do
    if [ $tw = 0 ]; then unset PGO_K3_TW; else export PGO_K3_TW=$tw; fi
    PGO_LIB=$PWD/toy-robust-backend-slam_amd/libpgo_exp.so timeout -k 10 200 python3 scripts/k3_probe.py -1 > $O/tw_${tw}_$rep.log 2>&1 || { tail -5 $O/tw_${tw}_$rep.log; exit 1; }
    echo "TW $tw rep $rep: $(grep -h "k_spmv_1<\|checksum" $O/tw_${tw}_$rep.log | tr '\n' ' ')"
  done
done
