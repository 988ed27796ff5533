#!/bin/bash
# a longer randomised hypothesis campaign of tests/test_gpu_fuzz.py on the GPU box (the suite itself runs 30 / 20 fixed examples)
# usage: exp_fuzz.sh [examples] [pytest -k expression]
set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out/fuzz
PGO_FUZZ_EXAMPLES=${1:-400} PYTHONUNBUFFERED=1 timeout -k 10 1000 python -u -m pytest tests/test_gpu_fuzz.py -q -m gpu --timeout 900 -k "${2:-test}" 2>&1 | tee gpurun_out/fuzz/fuzz.log | grep -v "^E    *\[\|^E    *array\|^E      *[0-9 \[-]" | tail -60
