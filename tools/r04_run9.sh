#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
mkdir -p gpurun_out/r04
ROUND=r04 bash tools/collect_profiles.sh > gpurun_out/r04/collect.log 2>&1
tail -3 gpurun_out/r04/collect.log
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/r04/gpu_tests.txt 2>&1
grep -E "passed|failed|FAILED" gpurun_out/r04/gpu_tests.txt | tail -8
