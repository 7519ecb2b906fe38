#!/bin/bash
# tools/prof_power.sh <tag> — PMC passes for the rtl_power kernel (config 4) on the GPU box
set -u
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { local name=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 $ROOT/tools/bench_power.py --steps 2 --cpu-seconds 0 > $OUT/$name.log 2>&1; }
run sqA SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES
run sqB SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD
cd $ROOT
python3 tools/pmc_summary.py $OUT k_power_scan | tee $OUT/summary.txt
