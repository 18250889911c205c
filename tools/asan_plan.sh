#!/bin/bash
# The planner and the host side of the C ABI under AddressSanitizer + UBSan (CPU only: hint_plan_check
# builds and verifies plans without a device).  Last run (round 6): 61 tests, no report.
set -e
cd "$(dirname "$0")/.."
make -C hint_amd/csrc asan
ASAN=$(find /opt/rocm/lib/llvm -name "libclang_rt.asan*x86_64*.so" | head -1)
HINT_AMD_LIB=$PWD/hint_amd/lib/libhint_amd_asan.so LD_PRELOAD=$ASAN ASAN_OPTIONS=detect_leaks=0:verify_asan_link_order=0 \
    UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 python -m pytest tests/test_plan_cpu.py tests/test_host_cpu.py -q -x
rm -f hint_amd/lib/libhint_amd_asan.so
