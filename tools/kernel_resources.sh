#!/bin/bash
# Registers and scratch of every kernel in a built library (default the product build): `bash tools/kernel_resources.sh [lib.so]`.
# A non-zero private_segment_fixed_size on an env kernel is a regression (round 3: a struct copy under a branch put 44 B of scratch in the tree stage).
LIB=${1:-/root/repo/apex_amd/lib/libapx.so}
T=$(mktemp -d); cp "$LIB" "$T/lib.so"; (cd "$T" && /opt/rocm/lib/llvm/bin/llvm-objdump --offloading lib.so > /dev/null 2>&1)
for f in "$T"/*gfx950; do
  /opt/rocm/lib/llvm/bin/llvm-readelf --notes "$f" | grep -E '^\s+\.name:|private_segment_fixed_size|\.vgpr_count|\.agpr_count|\.sgpr_count|group_segment_fixed_size' | paste - - - - - - | sed 's/\s\+/ /g'
done
rm -rf "$T"
