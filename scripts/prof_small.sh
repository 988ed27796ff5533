#!/bin/bash
# kernel trace of a small exact solve (INTEL + 50 outliers, METHOD 1): per-kernel durations and the gaps between
# consecutive launches inside the PCG loop
set -o pipefail
OUT=gpurun_out/prof_small; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cat > /tmp/small.py <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import toy_robust_backend_slam_amd as P
g = P.ReadG2O("tests/golden/data/INTEL.g2o"); g.add_random_C(50, 1)
s = P.Solver(g, P.Options(method=1, use_graphs=int(os.environ.get("UG", "1"))))
s.solve(); s.set_poses(np.array(g.poses))
sm = s.solve()
print("INTEL+50 M1: %d its, %d pcg its, %.3f s -> %.1f GN it/s, %.2f us per PCG iteration" % (sm.iterations, sm.total_pcg_iters, sm.seconds_total, sm.iterations / sm.seconds_total, 1e6 * sm.seconds_linear / sm.total_pcg_iters), s.info().as_dict())
PY
python3 /tmp/small.py || exit 1
UG=0 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 /tmp/small.py > $OUT/log.txt 2>&1 || { tail -5 $OUT/log.txt; exit 1; }
python3 - <<PY
import csv, glob
from collections import defaultdict
rows = []
for f in glob.glob("$OUT/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].split("(")[0].replace("pgo::dev::", "").replace("void ", "")
        rows.append((float(r["Start_Timestamp"]), float(r["End_Timestamp"]), n))
rows.sort()
d = defaultdict(list); gaps = defaultdict(list)
for i, (a, b, n) in enumerate(rows):
    d[n].append((b - a) / 1e3)
    if i + 1 < len(rows):
        gaps[n + " -> " + rows[i + 1][2]].append((rows[i + 1][0] - b) / 1e3)
for n, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:8]:
    v.sort(); print("%-34s calls %6d total %8.2f ms  median %6.2f us  p90 %6.2f us" % (n[:34], len(v), sum(v) / 1e3, v[len(v) // 2], v[int(0.9 * len(v))]))
for n, v in sorted(gaps.items(), key=lambda kv: -len(kv[1]))[:5]:
    v.sort(); print("gap %-60s n %6d median %6.2f us" % (n[:60], len(v), v[len(v) // 2]))
PY
find $OUT -name "*kernel_trace.csv" -delete
