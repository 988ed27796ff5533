#!/bin/bash
# bench (20 steps after 5) + in-loop kernel times of the current build
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --cpu-iters 0 --workloads 0 "$@" > gpurun_out/exp_bench.json 2> gpurun_out/exp_bench.err || { tail -5 gpurun_out/exp_bench.err; exit 1; }
python - <<PY
import json
d = json.load(open("gpurun_out/exp_bench.json"))
print("it/s %.2f ms/step %.2f pcg/step %.1f" % (d["value"], d["ms_per_step"], d["pcg_iters_per_step"]), d["seconds"])
print({k: round(v["ms"] * 1e3, 1) for k, v in d["kernels"].items()}, "roofline frac %.3f" % d["roofline"]["frac"])
PY
bash scripts/quick_prof.sh "$@" 2>&1 | tail -11
