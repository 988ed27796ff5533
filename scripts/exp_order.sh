#!/bin/bash
# Round 3, experiment 1 (GPU box): single-rank locality ordering A/B (GN it/s, kernel times, FETCH_SIZE) and the
# Infinity-Cache question (the product kernel on a p table spread past 256 MiB).  Output: gpurun_out/r03/
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r03
mkdir -p $O
if [ "${SKIP_TESTS:-0}" != "1" ]; then
  timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
  tail -3 $O/tests.log
fi
for rep in 1 2; do for po in 0 1; do
  timeout -k 10 300 python bench.py --steps 20 --warmup 5 --passes 3 --workloads 0 --cpu-iters 0 --pose-ordering $po > $O/order_${po}_${rep}.json 2> $O/order_${po}_${rep}.err || { tail -5 $O/order_${po}_${rep}.err; exit 1; }
  python - <<PY
import json
d = json.load(open("$O/order_${po}_${rep}.json"))
print("pose_ordering $po rep $rep: GN it/s %.2f  ms/step %.3f  pcg/step %.1f  create %.2f s  " % (d["value"], d["ms_per_step"], d["pcg_iters_per_step"], d["seconds"]["create"]),
      {k.split(" ")[0]: round(v["ms"] * 1e3, 1) for k, v in d["kernels"].items()})
PY
done; done
for po in 0 1; do for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/pmc_${po}_$c
  timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $O/pmc_${po}_$c -- python3 scripts/k3_probe.py $po > $O/pmc_${po}_$c.log 2>&1 || { tail -5 $O/pmc_${po}_$c.log; exit 1; }
  python3 scripts/pmc_kernels.py $O/pmc_${po}_$c $c "pose_ordering=$po"
  rm -rf $O/pmc_${po}_$c
done; done
# Infinity Cache or HBM?  the same product with the gathered vector spread over 24 (product) / 96 / 288 bytes per pose
for po in 0 1; do for st in 0 12 36; do
  PGO_LIB=$PWD/toy-robust-backend-slam_amd/libpgo_exp.so PGO_SPMV_PSTRIDE=$st timeout -k 10 200 python3 scripts/k3_probe.py $po > $O/mall_${po}_$st.log 2>&1 || { tail -5 $O/mall_${po}_$st.log; exit 1; }
  echo "p stride $st doubles: $(tail -1 $O/mall_${po}_$st.log)"
done; done
for st in 0 36; do
  rm -rf $O/pmc_mall_$st
  PGO_LIB=$PWD/toy-robust-backend-slam_amd/libpgo_exp.so PGO_SPMV_PSTRIDE=$st timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_mall_$st -- python3 scripts/k3_probe.py 0 > $O/pmc_mall_$st.log 2>&1 || { tail -5 $O/pmc_mall_$st.log; exit 1; }
  python3 scripts/pmc_kernels.py $O/pmc_mall_$st FETCH_SIZE "natural order, p stride $st"
  rm -rf $O/pmc_mall_$st
done
timeout -k 10 120 rocprofv3 --list-avail > $O/counters.txt 2>&1; grep -i -E "mall|hbm|dram|TCC_EA0_RD|TCC_HIT|TCC_MISS|TCC_REQ|BUBBLE" $O/counters.txt | cut -c1-160 | sort -u | head -40
