#!/bin/bash
# A/B on one box: padded tile slots (default) against the dense incidence layout, bench workload
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/pad; mkdir -p $O
for rep in 1 2 3 4 5 6; do
for lay in auto dense; do
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --passes 1 --workloads 0 --cpu-iters 0 --cpu-iters-1t 0 --layout $lay > $O/$lay.$rep.json 2> $O/$lay.$rep.err || { tail -5 $O/$lay.$rep.err; exit 1; }
python - <<PY
import json
d = json.load(open("$O/$lay.$rep.json"))
print("layout $lay rep $rep: GN it/s %.2f  ms/step %.3f  pcg/step %.1f cost %r" % (d["value"], d["ms_per_step"], d["pcg_iters_per_step"], d["cost_first_last"][1]), {k.split(" ")[0]: round(v["ms"] * 1e3, 1) for k, v in d["kernels"].items()})
PY
done; done
