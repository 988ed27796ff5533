#!/bin/bash
# K2 / K3 on the padded and on the dense layout, isolated kernel times (k3_probe), interleaved
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/k2pad; mkdir -p $O
for rep in 1 2 3; do
for pad in -1 0; do
python3 - > $O/p_${pad}_$rep.log 2>&1 <<PY || { tail -5 $O/p_${pad}_$rep.log; exit 1; }
import sys
sys.path.insert(0, ".")
import toy_robust_backend_slam_amd as P
P.set_knob("pad_tiles", $pad)
g = P.synth_manhattan(1000000, 4.0, 0.10, 20260410)
s = P.Solver(g, P.Options(method=1, max_iters=2, pcg_rtol=0.1, pcg_max_iters=50))
s.lm_begin(); s.lm_step(1)
k3 = s.bench_spmv(8); k2 = s.bench_assemble(5)
print("pad_tiles $pad rep $rep: k_spmv %.1f us  k_assemble %.1f us" % (k3.ms_avg * 1e3, k2.ms_avg * 1e3))
PY
tail -1 $O/p_${pad}_$rep.log
done; done
