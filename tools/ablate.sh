#!/bin/bash
# tools/ablate.sh build|run — cost attribution of the front-end kernels by leaving parts out.
# The ablation points are NOT in the product headers: tools/ablate.patch adds them to a copy of
# rtlsdr_amd/csrc under build_ablate/src (git-ignored), which is then built once per variant with
# -DRTLFM_ABLATE=<bits>.  Analysis only - results are wrong by construction, bench runs with --check 0.
# Bits (round 5: the patch was re-cut for the current kernels and keeps the one point that is still asked about): 1 = no
# discriminator - atan2_q14 and fast_atan2 return a cheap function of both operands - in every front end.  (Rounds 2-4 also
# cut passes, hand-offs and the PCM stores out; those questions are answered in LAB.md II 4.2 / 4.4b, and the store side is
# what tools/write_share_probe.hip measures without a kernel around it.)  For pairs of builds measured in one process:
# tools/ab_engines.py --libs ... (LAB.md I.13).  If the patch no longer applies after a change to dsp_device.h, re-create
# the point by hand in build_ablate/src and `diff -u` it back.
set -e
cd "$(dirname "$0")/.."
VARIANTS=${VARIANTS:-"0 1"}
if [ "$1" = build ]; then
  rm -rf build_ablate/src && mkdir -p build_ablate/src/rtlsdr_amd && cp -r rtlsdr_amd/csrc build_ablate/src/rtlsdr_amd/csrc && cp -r include build_ablate/src/include
  (cd build_ablate/src && patch -p1 < ../../tools/ablate.patch)
  for v in $VARIANTS; do
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -w -Ibuild_ablate/src/include -DRTLFM_ABLATE=$v \
      build_ablate/src/rtlsdr_amd/csrc/rtlfm_hip.hip build_ablate/src/rtlsdr_amd/csrc/rtlpower_hip.hip -o build_ablate/librtlfm_hip_a$v.so &
    [ $(jobs -r | wc -l) -ge 4 ] && wait -n
  done
  wait
else
  for rep in 1 2; do for v in $VARIANTS; do
    echo -n "ablate=$v: "
    RTLFM_HIP_LIB=$PWD/build_ablate/librtlfm_hip_a$v.so python bench.py --steps 40 --warmup 10 --no-cpu-baseline --check 0 --pmc 0 --e2e 0 --also 0 --ceiling 0 --sustain 0 "${@:2}" 2>/dev/null |
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['launch_ms'])"
  done; done
fi
