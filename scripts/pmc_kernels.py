"""FETCH_SIZE / WRITE_SIZE per launch of the hot kernels from a rocprofv3 --pmc output directory.
usage: pmc_kernels.py DIR COUNTER [label]      (FETCH_SIZE is reported x 2: MI355X_MICROARCH.md, HBM section)"""
import csv, glob, sys
from collections import defaultdict
d, counter = sys.argv[1], sys.argv[2]
label = sys.argv[3] if len(sys.argv) > 3 else d
v = defaultdict(list)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        n = r["Kernel_Name"]
        for k in ("k_spmv", "k_assemble", "k_edge_eval", "k_cg_init_cl", "k_cg_update1_cl", "k_cg_update2", "k_chain_factor", "k_cg_fused"):
            if k in n:
                if k == "k_edge_eval":
                    k += "<jac>" if ("ILb1" in n or "<true" in n) else "<cost>"
                v[k].append(float(r["Counter_Value"]))
mult = 2.0 if counter == "FETCH_SIZE" else 1.0
for k in sorted(v):
    x = sorted(v[k])
    x = [t for t in x if t > 0.5 * x[-1]]     # full launches only (PCG kernels early-out after convergence)
    print("%s %s %s: %.1f MB per launch (x%.0f applied; %d launches)" % (label, counter, k, sum(x) / len(x) * 1024 * mult / 1e6, mult, len(x)))
