#!/bin/bash
# The CPU suite against the library's HOST side built with AddressSanitizer + UndefinedBehaviorSanitizer (the reference runs Miri / ASan on its
# CPU code: SURVEY.md section 5; GPU sanitizers are not available on the build pool).  What it covers: everything the C ABI does without a device -
# the automaton builders, the query parser, packing, shard arithmetic, k-merge / radix sort, argument checks, error paths.
# Usage: bash tools/run_asan_cpu.sh [pytest args]      (builds build/asan/libfrizbee_hip.so first: ~2 min)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
make -C $ROOT/frizbee_amd/csrc asan > /dev/null
RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
cd $ROOT
LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:halt_on_error=0 UBSAN_OPTIONS=print_stacktrace=1 python -c "
import sys, frizbee_amd
frizbee_amd._LIB_PATH = '$ROOT/build/asan/libfrizbee_hip.so'
import pytest
sys.exit(pytest.main(['tests', '-q', '-m', 'not gpu', '-p', 'no:cacheprovider', '--deselect', 'tests/test_cpp_facade.py', '--deselect', 'tests/test_distributed_gloo.py'] + sys.argv[1:]))
" "$@" 2>&1 | tee build/asan/cpu_suite.log | tail -3
echo "sanitizer reports: $(grep -c 'runtime error\|ERROR: AddressSanitizer' build/asan/cpu_suite.log || true)"
