#!/bin/bash
# end-of-round evidence on ONE box: rocprofv3 passes (profile.sh) first, then the driver's bench command; the bench line is
# kept under profiles/ next to the profile of the same box
set -o pipefail
TAG=${1:-r03}
mkdir -p gpurun_out/profiles
bash scripts/profile.sh $TAG || exit 1
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_${TAG}.json 2> gpurun_out/bench_${TAG}.err || { tail -5 gpurun_out/bench_${TAG}.err; exit 1; }
cp gpurun_out/bench_${TAG}.json gpurun_out/profiles/${TAG}_bench.json
python3 - <<PY
import json
d = json.load(open("gpurun_out/bench_${TAG}.json"))
print("GN it/s %.2f  ms/step %.3f  passes %s  roofline %.3f traffic %s" % (d["value"], d["ms_per_step"], d.get("passes_ms_per_step"), d["roofline"]["frac"], d["roofline"].get("traffic")))
print({k: round(v["ms"] * 1e3, 1) for k, v in d["kernels"].items()})
print({k: round(v.get("gn_it_per_s", 0), 1) for k, v in d.get("workloads", {}).items()})
print(d.get("parity"))
print({k: v for k, v in d["cpu_baseline"].items() if k in ("value", "intel_plus_50_direct_solve")})
PY
