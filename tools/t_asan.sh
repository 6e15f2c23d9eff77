#!/bin/bash
# AddressSanitizer run of the env kernels (host + device code; LLVM lowers the LDS regions of the kernels to instrumented global memory) on the GPU box:
#   make -C apex_amd/csrc VARIANT=asan      (~20 min, gfx950:xnack+)
#   bash tools/t_asan.sh [steps]            -> gpurun_out/asan/asan.log
# Needs XNACK (HSA_XNACK=1); a box whose driver refuses it makes the first launch fail with "invalid device function": that is reported, not hidden.
mkdir -p gpurun_out/asan
export HSA_XNACK=1
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:halt_on_error=0 APX_LIB=$PWD/apex_amd/lib/libapx_asan.so python tools/t_check.py ${1:-12} > gpurun_out/asan/asan.log 2>&1
echo "exit code $?" >> gpurun_out/asan/asan.log
tail -25 gpurun_out/asan/asan.log
