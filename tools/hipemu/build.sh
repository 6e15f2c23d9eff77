#!/bin/bash
# host build of the learner-side kernel sources under the HIP emulation header -> tools/hipemu/_build/libapx_emul.so (git-ignored)
set -e
cd "$(dirname "$0")"
mkdir -p _build
CXX=${CXX:-/opt/rocm/lib/llvm/bin/clang++}
sed 's|extern __shared__ float fls\[\];|float* fls = (float*)hipemu::g_dynsmem;|' ../../apex_amd/csrc/learner.hip > _build/learner_emul.hip
grep -q 'hipemu::g_dynsmem' _build/learner_emul.hip
FLAGS="-x c++ -std=c++17 -O2 -fPIC -pthread -I. -I../../apex_amd/csrc -I../../include -Wno-unused-function -Wno-unused-variable -Wno-unknown-attributes -Wno-ignored-attributes"
pids=()
for u in emul_ppo_small emul_learner emul_td3_small; do
    $CXX $FLAGS -c $u.cpp -o _build/$u.o &
    pids+=($!)
done
for p in "${pids[@]}"; do wait "$p"; done      # (set -e: a failed translation unit ends the script here)
$CXX -shared -pthread _build/emul_ppo_small.o _build/emul_learner.o _build/emul_td3_small.o -o _build/libapx_emul.so
echo built tools/hipemu/_build/libapx_emul.so
