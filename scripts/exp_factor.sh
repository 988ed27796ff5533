#!/bin/bash
# k_chain_factor / K2 / k_prepare durations from a rocprofv3 kernel trace of a short run (LM iterations at 1M poses)
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/factor; mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 scripts/k3_probe.py -1 1000000 > $O/trace.log 2>&1 || { tail -5 $O/trace.log; exit 1; }
python3 - <<PY
import csv, glob
for f in glob.glob("$O/trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if any(k in r["Name"] for k in ("k_chain_factor", "k_assemble", "k_prepare")):
            print(r["Name"][:60], r["Calls"], "avg us %.1f" % (float(r["AverageNs"]) / 1e3))
PY
find $O -name "*kernel_trace.csv" -size +8M -delete
