#!/bin/bash
# run on the GPU box via gpurun: GPU test suite, output kept under gpurun_out/
mkdir -p gpurun_out
timeout -k 10 ${1:-900} python -m pytest tests -m gpu -x -q -s --durations=15 > gpurun_out/pytest_gpu.log 2>&1
rc=$?
tail -40 gpurun_out/pytest_gpu.log
exit $rc
