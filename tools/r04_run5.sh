#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04e
mkdir -p $OUT
cd $ROOT
timeout 1500 python -m pytest tests/test_parity_gpu.py -m gpu -q -k "any_512n or not_powers or boxcar or segments_inside or short_callback or golden or raw_dc" > $OUT/gpu_tests_w.txt 2>&1
grep -E "passed|failed|FAILED" $OUT/gpu_tests_w.txt | tail -40
