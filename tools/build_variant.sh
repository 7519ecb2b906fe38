#!/bin/bash
# tools/build_variant.sh <name> [-DFLAG=...]... — an extra build of the library under build_ablate/
# (git-ignored, travels with gpurun) for interleaved A/B runs: tools/ab_engines.py --libs ...
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p build_ablate
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -w -Iinclude "$@" \
  rtlsdr_amd/csrc/rtlfm_hip.hip rtlsdr_amd/csrc/rtlpower_hip.hip rtlsdr_amd/csrc/rtlfm_place.hip -o build_ablate/lib_$name.so
echo build_ablate/lib_$name.so
