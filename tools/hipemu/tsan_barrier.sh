#!/bin/bash
# The grid barrier of the persistent trainers under ThreadSanitizer (workgroups = threads; tools/hipemu/tsan/hip/hip_runtime.h says what is and is not modelled).
#   bash tools/hipemu/tsan_barrier.sh      -> builds _build/tsan_barrier and _build/tsan_barrier_broken, runs both; exit 0 iff the real barrier is clean AND the broken one is caught
set -e
cd "$(dirname "$0")"
CXX=${CXX:-/opt/rocm/lib/llvm/bin/clang++}
mkdir -p _build
FLAGS="-x c++ -std=c++17 -O1 -g -pthread -fsanitize=thread -Itsan -I../../apex_amd/csrc -I../../include -Wno-unused-function -Wno-unknown-attributes -Wno-ignored-attributes -Wno-builtin-macro-redefined"
$CXX $FLAGS tsan_barrier.cpp -o _build/tsan_barrier
$CXX $FLAGS -DTSAN_BREAK_RELEASE tsan_barrier.cpp -o _build/tsan_barrier_broken
TSAN_OPTIONS="halt_on_error=1 exitcode=66" ./_build/tsan_barrier 8 3000 2048
TSAN_OPTIONS="halt_on_error=1 exitcode=66" ./_build/tsan_barrier 3 20000 64
set +e
TSAN_OPTIONS="halt_on_error=1 exitcode=66" ./_build/tsan_barrier_broken 8 300 2048 > _build/tsan_broken.log 2>&1
rc=$?
set -e
grep -q "data race" _build/tsan_broken.log && [ $rc -eq 66 ] && echo "positive control: the barrier without its release is reported (exit $rc)"
