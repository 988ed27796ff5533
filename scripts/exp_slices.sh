#!/bin/bash
set -o pipefail
for ce in 100 50 34 20 10 6; do
  timeout -k 10 200 python bench.py --steps 20 --warmup 5 --cpu-iters 0 --workloads 0 --passes 3 --pcg-check-every $ce > gpurun_out/exp_sl.json 2> gpurun_out/exp_sl.err || { tail -5 gpurun_out/exp_sl.err; exit 1; }
  python -c "
import json; d=json.load(open('gpurun_out/exp_sl.json'))
print('check_every $ce: ms/step', [round(x,2) for x in d['passes_ms_per_step']], 'k_spmv', round(d['kernels']['k_spmv']['ms']*1e3,1))"
done
