#!/bin/bash
# host enqueue time per PCG iteration: single rank (graph), forced RCCL collectives captured / eager (world = 1, all-gather path)
for mode in "plain::" "rccl-graph:1:1" "rccl-eager:1:0"; do IFS=: read name fd gc <<< "$mode"
  PGO_BENCH_FORCE_DIST=$fd PGO_FORCE_COLLECTIVES=$fd PGO_GRAPH_COLLECTIVES=${gc:-1} python bench.py --steps 6 --warmup 2 --passes 1 --cpu-iters 0 --workloads 0 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); h = d['handle']
print('$name: %.2f ms/LM it, %.1f PCG it/LM it -> %.1f us device per PCG it; host enqueue %.2f us per PCG it; graph replay %d' % (d['ms_per_step'], d['pcg_iters_per_step'], 1e3 * d['seconds']['linear'] / max(1, d['pcg_iters_per_step'] * 8) , h['host_enqueue_us_per_pcg_iter'], h['pcg_graph_replay']))"
done
