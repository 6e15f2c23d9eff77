#!/bin/bash
# Memory-safety pass over the kernel sources (learner side by default; TESTS=tests/test_kernel_emulation_env.py: the env kernels): the host emulation (build.sh) compiled with AddressSanitizer, the emulated test suites run with the ASan runtime
# preloaded, so that every read / write of a kernel beyond a tensor it was handed (torch's CPU allocations carry redzones then) aborts the run.  Fibers (ucontext) and ASan
# coexist with detect_stack_use_after_return=0.   usage: bash tools/hipemu/asan.sh [pytest -k expression]   -> exit code of pytest
# Fibers switch with the emulation's own register switch on 8 MB stacks (default SWITCH=-DHIPEMU_STACK_MB=8).  SWITCH=-DHIPEMU_UCONTEXT selects glibc's swapcontext, which ASan knows about - but its
# interceptor clears the shadow of the whole fiber stack at every switch: far too slow for the env kernels, which switch at every DPP operand; stack-use-after-return detection is off either way.
# SAN=ubsan bash tools/hipemu/asan.sh: the same with UndefinedBehaviorSanitizer (misaligned vector loads, indices beyond a static = LDS array, shifts, overflow, NULL arithmetic).
set -e
cd "$(dirname "$0")"
CXX=${CXX:-/opt/rocm/lib/llvm/bin/clang++}
SAN=${SAN:-asan}
if [ "$SAN" = ubsan ]; then
    RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.ubsan_standalone-x86_64.so | head -1)
    SANFLAGS="-fsanitize=undefined -fno-sanitize=vptr,function -fno-sanitize-recover=undefined"; LINKFLAGS="-fsanitize=undefined"
else
    RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
    SANFLAGS="-fsanitize=address -shared-libsan -fno-omit-frame-pointer"; LINKFLAGS="-fsanitize=address -shared-libsan"
fi
mkdir -p _build/$SAN
FLAGS="-x c++ -std=c++17 -O1 -g -fPIC -pthread $SANFLAGS -I. -I../../apex_amd/csrc -I../../include -Wno-unused-function -Wno-unused-variable -Wno-unknown-attributes -Wno-ignored-attributes ${SWITCH:--DHIPEMU_STACK_MB=8}"
for f in emul_ppo_small emul_learner emul_td3_small emul_env; do $CXX $FLAGS -c $f.cpp -o _build/$SAN/$f.o 2> /dev/null & done
wait
$CXX -shared -pthread $LINKFLAGS _build/$SAN/*.o -ldl -o _build/libapx_emul_$SAN.so
cd ../..
LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 APX_EMUL_LIB=$PWD/tools/hipemu/_build/libapx_emul_$SAN.so \
    python -m pytest ${TESTS:-tests/test_kernel_emulation_learner.py} -q ${1:+-k "$1"}
