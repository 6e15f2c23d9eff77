#!/bin/bash
# host build of the kernel sources (learner side: learner.hip, ppo_small.hip, td3_small.hip; env side: env.hip with its headers) under the HIP emulation headers -> tools/hipemu/_build/libapx_emul.so (git-ignored)
set -e
cd "$(dirname "$0")"
mkdir -p _build
CXX=${CXX:-/opt/rocm/lib/llvm/bin/clang++}
FLAGS="-x c++ -std=c++17 -O2 -g1 -fPIC -pthread -I. -I../../apex_amd/csrc -I../../include -Wno-unused-function -Wno-unused-variable -Wno-unknown-attributes -Wno-ignored-attributes -Wno-psabi"
OUT=libapx_emul.so; OBJ=_build
if [ "${1:-}" = lockstep ]; then      # the lockstep checker (hip/hip_runtime.h): every load / store of the env kernels' translation unit reports to it
    OUT=libapx_emul_lockstep.so; OBJ=_build/lockstep; mkdir -p $OBJ
    FLAGS="$FLAGS -g -DHIPEMU_LOCKSTEP_CHECK"; ENVFLAGS="-fsanitize-coverage=trace-pc-guard,trace-loads,trace-stores"
fi
if [ "${1:-}" = check ]; then         # the env kernels' range-checked accessors (-DAPX_CHECK, env_state.h): every S(f) / S.W(i) / S.I(f) index of an emulated run is tested against its LDS region
    OUT=libapx_emul_check.so; OBJ=_build/check; mkdir -p $OBJ
    ENVFLAGS="-DAPX_CHECK"
fi
pids=()
grep -q ' fma ' /proc/cpuinfo && FLAGS="$FLAGS -mfma"      # fmaf as one instruction where the host has it (the env kernels' lane operations are fused multiply-adds)
for u in emul_ppo_small emul_learner emul_td3_small emul_env; do
    EX=; [ $u = emul_env ] && EX="${ENVFLAGS:-}"
    $CXX $FLAGS $EX -c $u.cpp -o $OBJ/$u.o &
    pids+=($!)
done
for p in "${pids[@]}"; do wait "$p"; done      # (set -e: a failed translation unit ends the script here)
$CXX -shared -pthread $OBJ/emul_ppo_small.o $OBJ/emul_learner.o $OBJ/emul_td3_small.o $OBJ/emul_env.o -ldl -o _build/$OUT
echo built tools/hipemu/_build/$OUT
