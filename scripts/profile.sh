#!/bin/bash
# rocprofv3 passes for the bench workload (run on the GPU box through gpurun).
#   pass 1: --kernel-trace --stats   (per-kernel time)
#   pass 2: --pmc FETCH_SIZE         (HBM read-side counter; own pass, TCC slots)
#   pass 3: --pmc WRITE_SIZE
# Raw output goes to gpurun_out/prof_<tag>/ ; scripts/summarize_profile.py condenses it into profiles/.
set -o pipefail
TAG=${1:-r03}
OUT=gpurun_out/prof_${TAG}
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="bench.py --steps 3 --warmup 1 --passes 1 --workloads 0 --cpu-iters 0 --kernel-reps 5"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ARGS > $OUT/trace.log 2>&1 || { tail -5 $OUT/trace.log; exit 1; }
PMCARGS="bench.py --steps 1 --warmup 0 --passes 1 --workloads 0 --cpu-iters 0 --kernel-reps 3 --pcg-max-iters 10"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $PMCARGS > $OUT/pmc_fetch.log 2>&1 || { tail -5 $OUT/pmc_fetch.log; exit 1; }
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $PMCARGS > $OUT/pmc_write.log 2>&1 || { tail -5 $OUT/pmc_write.log; exit 1; }
find $OUT -name "*.csv" | xargs ls -la
# keep the merge-back small: drop the per-dispatch trace (tens of MB), keep stats + counters
python3 scripts/summarize_profile.py $OUT $TAG && find $OUT -name "*kernel_trace.csv" -size +8M -delete
tail -3 $OUT/trace.log
# the summaries travel back through gpurun_out/ (the only directory merged into the caller's tree)
mkdir -p gpurun_out/profiles && cp profiles/${TAG}_* profiles/pmc_traffic.json gpurun_out/profiles/ 2>/dev/null
