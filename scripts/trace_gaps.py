"""Timeline of one PCG iteration from a rocprofv3 --kernel-trace CSV: per kernel, its duration and the idle gap before it
(start - end of the previous dispatch), averaged over the steady PCG loop.  usage: trace_gaps.py <dir with *kernel_trace.csv>"""
import csv, glob, os, sys
from collections import defaultdict
rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    with open(f, newline="") as fh:
        rows += list(csv.DictReader(fh))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    n = n.split("(")[0]
    for k in ("k_spmv_1", "k_spmv_p", "k_spmv_t", "k_fold_partials", "k_cg_update1_cl", "k_cg_update2", "k_cg_sr_cl", "k_cg_sr_scal", "k_finalize"):
        if k in n:
            return k
    return None
dur, gap, cnt = defaultdict(float), defaultdict(float), defaultdict(int)
prev_end = None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    k = short(r["Kernel_Name"])
    if k and prev_end is not None and e - s > 2500:      # (skip the no-op launches after convergence)
        dur[k] += (e - s) / 1e3
        gap[k] += (s - prev_end) / 1e3
        cnt[k] += 1
    prev_end = e
tot = 0.0
for k in dur:
    print("%-18s n %6d  avg %7.2f us  gap before %6.2f us" % (k, cnt[k], dur[k] / cnt[k], gap[k] / cnt[k]))
    if cnt[k] > 0.5 * max(cnt.values()):
        tot += (dur[k] + gap[k]) / cnt[k]
print("one PCG iteration (kernels launched every iteration, with their gaps): %.1f us" % tot)
