set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/k3nt; mkdir -p $O
for rep in 1 2; do
for cfg in "0 -" "2 -" "4 -" "4 100000" "2 100000"; do
set -- $cfg
if [ $1 = 0 ]; then unset PGO_K3_NT; else export PGO_K3_NT=$1; fi
if [ $2 = - ]; then unset PGO_FOLD_MIN; else export PGO_FOLD_MIN=$2; fi
PGO_LIB=$PWD/toy-robust-backend-slam_amd/libpgo_exp.so timeout -k 10 300 python bench.py --steps 20 --warmup 5 --passes 1 --workloads 0 --cpu-iters 0 --cpu-iters-1t 0 > $O/$1.$2.$rep.json 2> $O/$1.$2.$rep.err || { tail -5 $O/$1.$2.$rep.err; exit 1; }
python - <<PY
import json
d = json.load(open("$O/$1.$2.$rep.json"))
print("NT=$1 FOLD_MIN=$2 rep $rep: GN it/s %.2f  ms/step %.3f  pcg/step %.1f" % (d["value"], d["ms_per_step"], d["pcg_iters_per_step"]), {k.split(" ")[0]: round(v["ms"] * 1e3, 1) for k, v in d["kernels"].items() if "spmv" in k})
PY
done; done
