#!/bin/bash
# tools/prof_pmc.sh <tag> [bench args...] — PMC passes for the fused kernel on the GPU box.
# Counters are collected in their own runs (no tracing domains mixed in), one
# hardware pass per rocprofv3 invocation, as MI355X_MICROARCH.md prescribes.
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { # name counters...
  local name=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --check 0 --pmc 0 --sustain 0 "${BENCH_ARGS[@]}" > $OUT/$name.log 2>&1
}
BENCH_ARGS=("$@")
run sqA SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES
run sqB SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD
run fetch FETCH_SIZE GRBM_GUI_ACTIVE
run write WRITE_SIZE GRBM_GUI_ACTIVE
cd $ROOT
python3 tools/pmc_summary.py $OUT | tee $OUT/summary.txt
