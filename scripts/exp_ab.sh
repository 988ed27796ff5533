#!/bin/bash
# A/B on ONE box, interleaved repetitions: grids of the chain apply and of the flat vector kernels
for rep in 1 2 3; do for v in 2048:2048 768:1024 512:1024 512:2048; do IFS=: read cg fg <<< "$v"
  PGO_CHAIN_GRID=$cg PGO_FLAT_GRID=$fg python bench.py --passes 1 --cpu-iters 0 --workloads 0 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('rep $rep CHAIN_GRID=$cg FLAT_GRID=$fg: %.2f ms/LM it  k_spmv %.1f us' % (d['ms_per_step'], d['kernels']['k_spmv']['ms'] * 1e3))"
done; done
