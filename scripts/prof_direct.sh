#!/bin/bash
# kernel trace of the direct (chain + low-rank) solve on INTEL + 50 outliers, METHOD 1: per-kernel durations
set -o pipefail
OUT=gpurun_out/prof_direct; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cat > /tmp/direct.py <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import toy_robust_backend_slam_amd as P
name = os.environ.get("DS", "INTEL")
g = P.ReadG2O("tests/golden/data/%s.g2o" % name)
if name == "INTEL": g.add_random_C(50, 1)
s = P.Solver(g, P.Options(method=1, linear_solver=2))
s.solve(); s.set_poses(np.array(g.poses))
sm = s.solve()
print("%s M1 direct: %d its, %.3f s -> %.1f GN it/s, linear %.3f ms per LM iteration" % (name, sm.iterations, sm.seconds_total, sm.iterations / sm.seconds_total, 1e3 * sm.seconds_linear / sm.iterations), s.info().as_dict())
PY
python3 /tmp/direct.py || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 /tmp/direct.py > $OUT/log.txt 2>&1 || { tail -5 $OUT/log.txt; exit 1; }
python3 - <<PY
import csv, glob
from collections import defaultdict
rows = []
for f in glob.glob("$OUT/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].split("(")[0].replace("pgo::dev::", "").replace("void ", "")
        rows.append((float(r["Start_Timestamp"]), float(r["End_Timestamp"]), n))
rows.sort()
d = defaultdict(list)
for i, (a, b, n) in enumerate(rows):
    d[n].append((b - a) / 1e3)
tot = sum(sum(v) for v in d.values())
for n, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:14]:
    v.sort(); print("%-34s calls %6d total %8.2f ms (%4.1f %%) median %7.2f us  p90 %7.2f us" % (n[:34], len(v), sum(v) / 1e3, 100 * sum(v) / tot, v[len(v) // 2], v[int(0.9 * len(v))]))
PY
find $OUT -name "*kernel_trace.csv" -delete
