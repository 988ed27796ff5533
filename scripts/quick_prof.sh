#!/bin/bash
# quick per-kernel timing of the bench workload: rocprofv3 kernel trace -> full-launch averages of the PCG kernels
set -o pipefail
OUT=gpurun_out/quick_prof
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 bench.py --steps 3 --warmup 1 --cpu-iters 0 --workloads 0 --passes 1 --kernel-reps 3 "$@" > $OUT/log.txt 2>&1 || { tail -5 $OUT/log.txt; exit 1; }
python3 - <<PY
import csv, glob
from collections import defaultdict
d = defaultdict(list)
for f in glob.glob("$OUT/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        n = n.split("(")[0].replace("pgo::dev::", "").replace("void ", "")
        d[n].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3)
for n, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:9]:
    v.sort(); p90 = v[min(len(v) - 1, int(0.9 * len(v)))]; f = [x for x in v if x > 0.5 * p90]
    print("%-40s calls %5d total %8.2f ms  full %4d avg %8.2f us" % (n[:40], len(v), sum(v) / 1e3, len(f), sum(f) / len(f)))
PY
tail -1 $OUT/log.txt | cut -c1-120
find $OUT -name "*kernel_trace.csv" -delete
