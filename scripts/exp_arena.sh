#!/bin/bash
# one device allocation for the whole handle, buffers carved at different alignments / skews: kernel times (deterministic per policy?)
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/arena; mkdir -p $O
for rep in 1 2 3; do
for cfg in "none 256 0 0" "none 256 0 1" "none 256 0 2" "2048 2097152 0 0"; do
set -- $cfg
if [ $1 = none ]; then unset PGO_ARENA_MB; else export PGO_ARENA_MB=$1; fi
export PGO_ARENA_ALIGN=$2 PGO_ARENA_SKEW=$3 PGO_ALLOC_POW2=$4
PGO_LIB=$PWD/toy-robust-backend-slam_amd/libpgo_exp.so python3 - > $O/a_$1_$2_$3_$4_$rep.log 2>&1 <<PY || { tail -5 $O/a_$1_$2_$3_$4_$rep.log; exit 1; }
import sys
sys.path.insert(0, ".")
import toy_robust_backend_slam_amd as P
g = P.synth_manhattan(1000000, 4.0, 0.10, 20260410)
for pad in (-1, 0):
    P.set_knob("pad_tiles", pad)
    s = P.Solver(g, P.Options(method=1, max_iters=2, pcg_rtol=0.1, pcg_max_iters=50))
    s.lm_begin(); s.lm_step(1)
    k3 = s.bench_spmv(8); k2 = s.bench_assemble(5); pc = s.bench_precond(5); k1 = s.bench_eval(5, True)
    print("arena $1 pow2 $4 pad %2d rep $rep: k_spmv %.1f  k_assemble %.1f  precond %.1f  k1 %.1f" % (pad, k3.ms_avg * 1e3, k2.ms_avg * 1e3, pc.ms_avg * 1e3, k1.ms_avg * 1e3))
    s.close()
PY
tail -2 $O/a_$1_$2_$3_$4_$rep.log
done; done
