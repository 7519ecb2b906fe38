#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04c
mkdir -p $OUT
cd $ROOT
timeout 600 python tools/two_queue_probe.py > $OUT/two_queue.txt 2>&1
cat $OUT/two_queue.txt | grep -v amdgpu.ids
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/gpu_tests.txt 2>&1
tail -4 $OUT/gpu_tests.txt
