set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/chain_nw; mkdir -p $O
run() { # label
PGO_LIB=$PWD/toy-robust-backend-slam_amd/libpgo_exp.so timeout -k 10 300 python bench.py --steps 20 --warmup 5 --passes 1 --workloads 0 --cpu-iters 0 --cpu-iters-1t 0 > $O/$1.json 2> $O/$1.err || { tail -5 $O/$1.err; exit 1; }
python - <<PY
import json
d = json.load(open("$O/$1.json"))
print("$1: GN it/s %.2f  ms/step %.3f  pcg/step %.1f" % (d["value"], d["ms_per_step"], d["pcg_iters_per_step"]), {k.split(" ")[0]: round(v["ms"] * 1e3, 1) for k, v in d["kernels"].items() if "spmv" in k or "precond" in k})
PY
}
for rep in 1 2; do
unset PGO_CHAIN_SMALL_TILES PGO_CHAIN_SCAN; run default.$rep
export PGO_CHAIN_SMALL_TILES=100000 PGO_CHAIN_SCAN=0; run nw1_serial.$rep
export PGO_CHAIN_SMALL_TILES=100000; unset PGO_CHAIN_SCAN; run nw1_scan.$rep
done
