"""Per-function comparison of the gfx950 instruction streams of two builds of libapx.so: for every function of the env code object, how many instruction lines survive IN ORDER
(difflib) - the first filter for a change to the env kernels when no device is there to time it: a change that leaves the hot kernels' lines at 99.9 % touched only call offsets,
one that drops a kernel to 60 % has re-rolled its register allocation (the +- 1-2 % lottery of HISTORY.md section 10).
    python tools/isa_diff.py base.so [new.so = apex_amd/lib/libapx.so]"""
import difflib
import os
import re
import shutil
import subprocess
import sys
import tempfile

BIN = "/opt/rocm/lib/llvm/bin"


def functions(lib):
    d = tempfile.mkdtemp()
    try:
        shutil.copy(lib, os.path.join(d, "lib.so"))
        subprocess.run([os.path.join(BIN, "llvm-objdump"), "--offloading", "lib.so"], cwd=d, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        big = max((os.path.join(d, f) for f in os.listdir(d) if "gfx950" in f), key=os.path.getsize)
        asm = subprocess.run([os.path.join(BIN, "llvm-objdump"), "-d", "--mcpu=gfx950", big], capture_output=True, text=True).stdout
    finally:
        shutil.rmtree(d, ignore_errors=True)
    out, name = {}, None
    for ln in asm.split("\n"):
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", ln)
        if m:
            name = m.group(1); out[name] = []
        elif name and "//" in ln:
            out[name].append(ln.split("//")[0].strip())
    return out


def main():
    base = functions(sys.argv[1])
    new = functions(sys.argv[2] if len(sys.argv) > 2 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "apex_amd", "lib", "libapx.so"))
    same = 0
    for k in sorted(base):
        if k not in new:
            print("%-88s only in the base build" % k[:88]); continue
        if base[k] == new[k]:
            same += 1; continue
        keep = sum(b.size for b in difflib.SequenceMatcher(None, base[k], new[k], autojunk=False).get_matching_blocks())
        print("%-88s %6d -> %6d instructions, %6d in order (%.1f %%)" % (k[:88], len(base[k]), len(new[k]), keep, 100.0 * keep / max(len(base[k]), 1)))
    for k in sorted(set(new) - set(base)):
        print("%-88s only in the new build" % k[:88])
    print("%d of %d functions identical" % (same, len(base)))


if __name__ == "__main__":
    main()
