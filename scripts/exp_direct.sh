#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 300 python scripts/exp_direct.py ${1:-} > gpurun_out/direct.log 2>&1
rc=$?
tail -30 gpurun_out/direct.log
exit $rc
