#!/bin/bash
# A/B on ONE box: env settings x repetitions -> ms per LM iteration (20 steps after 5) and k_spmv HIP-event time
for rep in 1 2; do for to in 0 1; do
  PGO_TILE_ORDER=$to python bench.py --passes 1 --cpu-iters 0 --workloads 0 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('PGO_TILE_ORDER=$to rep $rep: %.2f ms/LM it  k_spmv %.1f us  K2 %.1f us' % (d['ms_per_step'], d['kernels']['k_spmv']['ms'] * 1e3, d['kernels']['k_assemble']['ms'] * 1e3))"
done; done
