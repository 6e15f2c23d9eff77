#!/bin/bash
# host build of the epoch kernel's source under the HIP emulation header -> tools/hipemu/_build/libppo_small_emul.so (git-ignored)
set -e
cd "$(dirname "$0")"
mkdir -p _build
CXX=${CXX:-/opt/rocm/lib/llvm/bin/clang++}
$CXX -x c++ -std=c++17 -O2 -fPIC -shared -pthread -I. -I../../include -Wno-unused-function -Wno-unused-variable emul_ppo_small.cpp -o _build/libppo_small_emul.so
echo built tools/hipemu/_build/libppo_small_emul.so
