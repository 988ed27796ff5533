#!/bin/bash
# K3 experiment: tile order x nt loads x pose ordering; bench_spmv time (HIP events) per variant
set -o pipefail
cat > /tmp/k3.py <<'PY'
import os, sys, time
sys.path.insert(0, os.getcwd())
import toy_robust_backend_slam_amd as P
g = P.synth_manhattan(1000000)
po = int(os.environ.get("PO", "-1"))
t0 = time.time()
s = P.Solver(g, P.Options(method=1, max_iters=2, pcg_rtol=0.1, pcg_max_iters=50, pose_ordering=po))
tc = time.time() - t0
s.lm_begin(); s.lm_step(1)
k = s.bench_spmv(30)
print("TILE_ORDER=%s NT=%s PO=%d create %.2fs  k_spmv %.1f us" % (os.environ.get("PGO_TILE_ORDER", "1"), os.environ.get("PGO_SPMV_NT", "0"), po, tc, k.ms_avg * 1e3), flush=True)
PY
for to in 0 1; do for nt in 0 1; do for po in -1 1; do  # PGO_TILE_ORDER: 1 = BFS tile order (off by default)
  PGO_TILE_ORDER=$to PGO_SPMV_NT=$nt PO=$po timeout -k 10 120 python /tmp/k3.py || exit 1
done; done; done
