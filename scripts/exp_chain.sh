#!/bin/bash
# experiment: chain apply kernel variants (PGO_CHAIN_KERNEL = scan | lean2 | lean4) -- correctness tests, then bench + in-loop kernel times
set -o pipefail
mkdir -p gpurun_out
for v in lean2 lean4; do
  PGO_CHAIN_KERNEL=$v timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "chain or precond" > gpurun_out/exp_chain_test_$v.log 2>&1 || { tail -30 gpurun_out/exp_chain_test_$v.log; exit 1; }
  tail -2 gpurun_out/exp_chain_test_$v.log
done
for v in scan lean2 lean4; do
  PGO_CHAIN_KERNEL=$v timeout -k 10 200 python bench.py --steps 20 --warmup 5 --cpu-iters 0 > gpurun_out/exp_chain_bench_$v.json 2> gpurun_out/exp_chain_bench_$v.err || { tail -5 gpurun_out/exp_chain_bench_$v.err; exit 1; }
  python - <<PY
import json
d = json.load(open("gpurun_out/exp_chain_bench_$v.json"))
print("$v", "it/s %.2f ms/step %.2f pcg/step %.1f" % (d["value"], d["ms_per_step"], d["pcg_iters_per_step"]), d["seconds"])
PY
done
for v in scan lean2 lean4; do
  echo "== quick_prof $v"
  PGO_CHAIN_KERNEL=$v bash scripts/quick_prof.sh 2>&1 | tail -12
done
