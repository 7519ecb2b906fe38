#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04i
mkdir -p $OUT
cd $ROOT
timeout 600 python tools/placement_classes.py > $OUT/classes.txt 2>&1
grep -v amdgpu.ids $OUT/classes.txt
timeout 600 python -m pytest tests/test_cli_gpu.py -m gpu -q -k "rtl_power_cli" 2>&1 | tail -3
RTLFM_SWEEP=400 RTLFM_SWEEP_CB=120 RTLFM_SWEEP_POWER=200 RTLFM_SWEEP_POWER_BIG=40 RTLFM_TAIL_FUZZ=400 timeout 2400 python -m pytest tests/test_parity_gpu.py tests/test_power_gpu.py tests/test_tail_long_runs_gpu.py -m gpu -q -k "random or fuzz or long_runs or fine_bins" > $OUT/fuzz.txt 2>&1
tail -3 $OUT/fuzz.txt
