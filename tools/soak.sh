#!/bin/bash
# tools/soak.sh [reps] — on the GPU box: the parity file (its allocation history first, as in the run that once failed),
# then tests/test_soak_gpu.py with RTLFM_SOAK=reps (default 10000) in the SAME pytest process; the log lands in
# gpurun_out/soak/ (copy soak_log.txt into profiles/ as rNN_soak_boxK.txt).  9 cases x 4 shapes x reps launches, every
# one compared with the oracle, every second repetition executed twice and compared on the device (verify_twice).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
REPS=${1:-10000}
cd $ROOT
mkdir -p gpurun_out/soak
{
  echo "== soak on $(hostname) $(date -u +%FT%TZ), RTLFM_SOAK=$REPS, library $(sha256sum rtlsdr_amd/csrc/librtlfm_hip.so | cut -c1-16)"
  rocm-smi --showproductname 2>/dev/null | grep -i "card series" | head -1
} >> gpurun_out/soak/soak_log.txt
RTLFM_SOAK=$REPS RTLFM_SOAK_CONTINUE=1 timeout ${SOAK_TIMEOUT:-3000} python -m pytest tests/test_parity_gpu.py tests/test_soak_gpu.py -m gpu -q -p no:cacheprovider \
  ${SOAK_K:+-k "$SOAK_K"} 2>&1 | tail -40 | cut -c1-1500 | tee -a gpurun_out/soak/soak_pytest_tail.txt
tail -12 gpurun_out/soak/soak_log.txt
