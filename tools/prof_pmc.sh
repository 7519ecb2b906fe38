#!/bin/bash
# tools/prof_pmc.sh <tag> [bench args...] — PMC passes on the GPU box for the kernels named in
# PMC_KERNELS (space-separated substrings of kernel names, default "k_fused"): one summary per kernel
# from the same runs.  Counters are collected in their own runs (no tracing domains mixed in), one
# hardware pass per rocprofv3 invocation, as MI355X_MICROARCH.md prescribes.  bench.py notices the
# profiler and neither builds nor starts the PCIe leg's child process (nothing execs under the preload).
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { # name counters...
  local name=$1; shift
  # --pmc-child: the input is one random fill instead of the signal generator's thousands of small dispatches
  # (rocprofv3's counter mode stalls on those at 4096 streams: 25 GPU-minutes were lost finding out), no build,
  # no child processes; the outer timeout bounds a pass that stalls anyway
  timeout 300 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 $ROOT/bench.py --pmc-child --steps 3 --warmup 1 --no-cpu-baseline --check 0 --pmc 0 --sustain 0 --e2e 0 --ceiling 0 --also 0 --colocate 1 "${BENCH_ARGS[@]}" > $OUT/$name.log 2>&1
}
BENCH_ARGS=("$@")
run sqA SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES
run sqB SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD
run fetch FETCH_SIZE GRBM_GUI_ACTIVE
run write WRITE_SIZE GRBM_GUI_ACTIVE
cd $ROOT
for k in ${PMC_KERNELS:-${PMC_KERNEL:-k_fused}}; do
  { echo "# rocprofv3 --pmc passes of: python3 bench.py --steps 3 --warmup 1 ${BENCH_ARGS[*]} ; kernel *$k*"; python3 tools/pmc_summary.py $OUT "$k"; } | tee $OUT/summary_$k.txt
done
