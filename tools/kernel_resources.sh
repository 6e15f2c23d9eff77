#!/bin/bash
# Registers and scratch of every kernel in a built library (default the product build): `bash tools/kernel_resources.sh [lib.so]`.
# A non-zero private_segment_fixed_size on an env kernel is a regression (round 3: a struct copy under a branch put 44 B of scratch in the tree stage).
LIB=${1:-/root/repo/apex_amd/lib/libapx.so}
T=$(mktemp -d); cp "$LIB" "$T/lib.so"; (cd "$T" && /opt/rocm/lib/llvm/bin/llvm-objdump --offloading lib.so > /dev/null 2>&1)
for f in "$T"/*gfx950; do
  /opt/rocm/lib/llvm/bin/llvm-readelf --notes "$f" | grep -E '^\s+\.name:|private_segment_fixed_size|\.vgpr_count|\.agpr_count|\.sgpr_count|group_segment_fixed_size' | paste - - - - - - | sed 's/\s\+/ /g'
done
# code bytes per function of the env code object (tests/test_build_artifacts.py holds the hot kernels under a ceiling)
big=$(ls -S "$T"/*gfx950 | head -1)
/opt/rocm/lib/llvm/bin/llvm-objdump -d --mcpu=gfx950 "$big" | python3 -c '
import re, sys
name, first, last, out = None, None, None, []
for ln in sys.stdin:
    m = re.match(r"^[0-9a-f]+ <(.*)>:", ln)
    if m:
        if name: out.append((last - first, name))
        name, first = m.group(1), None; continue
    if "//" in ln and name:
        try: v = int(ln.split("//")[1].split(":")[0], 16)
        except ValueError: continue
        first = v if first is None else first; last = v
if name: out.append((last - first, name))
for n, k in sorted(out, reverse=True)[:24]: print("%8d B  %s" % (n, k[:110]))
'
rm -rf "$T"
