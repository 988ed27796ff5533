#!/bin/bash
# FETCH_SIZE of k_spmv per variant (PGO_TILE_ORDER x PGO_SPMV_NT): does the processing order change the fabric traffic?
set -o pipefail
export TMPDIR=/tmp
cat > /tmp/k3.py <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import toy_robust_backend_slam_amd as P
g = P.synth_manhattan(1000000)
s = P.Solver(g, P.Options(method=1, max_iters=2, pcg_rtol=0.1, pcg_max_iters=50))
s.lm_begin(); s.lm_step(1)
k = s.bench_spmv(6)
print("k_spmv %.1f us" % (k.ms_avg * 1e3), flush=True)
PY
for to in 0 1; do for nt in 0 1; do
  OUT=gpurun_out/k3pmc_${to}_${nt}; rm -rf $OUT; mkdir -p $OUT
  PGO_TILE_ORDER=$to PGO_SPMV_NT=$nt timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT -- python3 /tmp/k3.py > $OUT/log.txt 2>&1 || { tail -5 $OUT/log.txt; exit 1; }
  python3 - <<PY
import csv, glob
v = []
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "FETCH_SIZE" and "k_spmv" in r["Kernel_Name"]:
            v.append(float(r["Counter_Value"]))
v = v[-6:]
print("TILE_ORDER=$to NT=$nt: k_spmv FETCH_SIZE avg %.0f KiB x 1.999 = %.1f MB (%d launches)" % (sum(v) / len(v), sum(v) / len(v) * 1024 * 1.999 / 1e6, len(v)))
PY
  rm -rf $OUT
done; done
