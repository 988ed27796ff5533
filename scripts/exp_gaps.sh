set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/gaps; mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 bench.py --steps 3 --warmup 1 --passes 1 --workloads 0 --cpu-iters 0 --cpu-iters-1t 0 --kernel-reps 3 > $O/trace.log 2>&1 || { tail -5 $O/trace.log; exit 1; }
python3 scripts/trace_gaps.py $O/trace | tee $O/gaps.txt
find $O -name "*kernel_trace.csv" -size +8M -delete
