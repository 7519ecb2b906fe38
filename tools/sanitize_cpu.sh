#!/bin/bash
# tools/sanitize_cpu.sh — the CPU-side C code under AddressSanitizer + UBSan (GPU sanitizers are
# not available on the pool): the device shim (product code that runs in the caller's process) and
# the oracle's restatement (it defines "correct", so it must itself be free of undefined
# behaviour).  The regular builds are put back afterwards.
set -e
cd "$(dirname "$0")/.."
ASAN=$(gcc -print-file-name=libasan.so); UBSAN=$(gcc -print-file-name=libubsan.so)
SAN="-O1 -g -fPIC -shared -fsanitize=address,undefined -fno-omit-frame-pointer"
export ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
cp rtlsdr_amd/csrc/host/librtlsdr_file.so /tmp/librtlsdr_file.orig.so
cp oracle/liboracle.so /tmp/liboracle.orig.so
trap 'cp /tmp/librtlsdr_file.orig.so rtlsdr_amd/csrc/host/librtlsdr_file.so; cp /tmp/liboracle.orig.so oracle/liboracle.so' EXIT
gcc $SAN -Wall -Wextra -o rtlsdr_amd/csrc/host/librtlsdr_file.so rtlsdr_amd/csrc/host/rtlsdr_file.c
(cd oracle && gcc $SAN -Wall -Wextra -Wno-unused-parameter -o liboracle.so rtlfm_oracle.c rtlpower_oracle.c -lm -lpthread)
LD_PRELOAD="$ASAN $UBSAN" python -m pytest tests/test_device_shim.py tests/test_oracle_golden.py tests/test_power_oracle.py -q
