#!/bin/bash
# quick shake-out of bench.py's legs on a small graph, then the driver's command; output under gpurun_out/r03/
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r03
mkdir -p $O
echo "nproc $(nproc)  cpu.max $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)  cpuset $(cat /sys/fs/cgroup/cpuset.cpus.effective 2>/dev/null | cut -c1-60)"
lscpu | grep -E "Model name|Socket|Core|Thread|NUMA node\(s\)" | head -6
timeout -k 10 600 python bench.py --poses 100000 --steps 5 --warmup 2 --passes 1 > $O/bench_small.json 2> $O/bench_small.err || { tail -20 $O/bench_small.err; exit 1; }
python - <<PY
import json
d = json.load(open("$O/bench_small.json"))
print("100k: GN it/s %.1f" % d["value"], json.dumps(d["cpu_baseline"], indent=None)[:1500])
print(json.dumps(d["workloads"].get("time_to_cost"))[:1500])
PY
T0=$(date +%s)
timeout -k 10 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_full.json 2> $O/bench_full.err || { tail -20 $O/bench_full.err; exit 1; }
echo "full bench wall $(( $(date +%s) - T0 )) s"
python - <<PY
import json
d = json.load(open("$O/bench_full.json"))
print("GN it/s %.2f  ms/step %.3f  passes %s  roofline %.3f" % (d["value"], d["ms_per_step"], [round(x, 2) for x in d.get("passes_ms_per_step")], d["roofline"]["frac"]))
print({k.split(" ")[0]: round(v["ms"] * 1e3, 1) for k, v in d["kernels"].items()})
for k, v in d.get("workloads", {}).items():
    if k != "time_to_cost": print(k, {a: (round(b, 3) if isinstance(b, float) else b) for a, b in v.items()})
print(json.dumps(d["workloads"].get("time_to_cost"), indent=1))
print(d.get("parity"))
print(json.dumps({k: v for k, v in d["cpu_baseline"].items() if k not in ("sample", "direct_solve")}, indent=1))
PY
