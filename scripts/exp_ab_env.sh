#!/bin/bash
# A/B of one environment switch of the experiment build on a bench workload: exp_ab_env.sh POSES NAME VALUE_A VALUE_B ... ("-" = unset)
set -o pipefail
export TMPDIR=/tmp
POSES=$1; NAME=$2; shift; shift
O=gpurun_out/ab_$NAME; mkdir -p $O
for rep in 1 2; do
for v in "$@"; do
if [ "$v" = "-" ]; then unset $NAME; else export $NAME=$v; fi
PGO_LIB=$PWD/toy-robust-backend-slam_amd/libpgo_exp.so timeout -k 10 300 python bench.py --poses $POSES --steps 20 --warmup 5 --passes 1 --workloads 0 --cpu-iters 0 --cpu-iters-1t 0 > $O/$POSES.$v.$rep.json 2> $O/$POSES.$v.$rep.err || { tail -5 $O/$POSES.$v.$rep.err; exit 1; }
python - <<PY
import json
d = json.load(open("$O/$POSES.$v.$rep.json"))
print("$POSES poses $NAME=$v rep $rep: GN it/s %.2f  ms/step %.3f" % (d["value"], d["ms_per_step"]), {k.split(" ")[0]: round(v["ms"] * 1e3, 1) for k, v in d["kernels"].items()})
PY
done; done
