#!/bin/bash
# tools/ablate.sh build|run — cost attribution of the fused kernel by leaving parts out
# (RTLFM_ABLATE, analysis only; results are wrong by construction, so bench runs with --check 0).
# Bits: k_fused 1 no atan2, 2 / 4 / 8 no passes 1-3 / pass 0 / hand-offs, 16 no PCM stores, 32 PCM
# stores to the same few lines (no write traffic leaves L2); k_boxcar_scan 1 no atan2, 2 no PCM
# stores, 4 no outputs at all, 8 PCM stores to the same few lines.  For pairs of builds measured in
# one process: tools/build_variant.sh + tools/ab_engines.py --libs.
set -e
cd "$(dirname "$0")/.."
VARIANTS="0 1 2 4 8 3 6 7 15"
if [ "$1" = build ]; then
  mkdir -p build_ablate
  for v in $VARIANTS; do
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -Iinclude -DRTLFM_ABLATE=$v \
      rtlsdr_amd/csrc/rtlfm_hip.hip rtlsdr_amd/csrc/rtlpower_hip.hip -o build_ablate/librtlfm_hip_a$v.so &
    [ $(jobs -r | wc -l) -ge 4 ] && wait -n
  done
  wait
else
  for rep in 1 2; do for v in $VARIANTS; do
    echo -n "ablate=$v: "
    RTLFM_HIP_LIB=$PWD/build_ablate/librtlfm_hip_a$v.so python bench.py --steps 40 --warmup 10 --no-cpu-baseline --check 0 2>/dev/null |
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['launch_ms'])"
  done; done
fi
