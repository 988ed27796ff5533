#!/bin/bash
# bench.py as the driver launches it for N > 1, rehearsed on ONE GPU: gloo + host-staged shm communicator, all ranks on cuda:0
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/rehearsal; mkdir -p $O
for n in 2 4; do
  T0=$(date +%s)
  timeout -k 10 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus $n --steps 4 --warmup 1 --comm shm > $O/n$n.json 2> $O/n$n.err || { tail -20 $O/n$n.err; exit 1; }
  echo "N=$n wall $(( $(date +%s) - T0 )) s"
  python - <<PY
import json
d = json.loads([l for l in open("$O/n$n.json") if l.startswith("{")][-1])
print({k: d[k] for k in ("value", "n_gpus", "ms_per_step", "scaling")}, d["config"], d.get("handle"), d.get("halo_exchange_check"), d.get("parity"))
PY
done
