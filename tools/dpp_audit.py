"""Static audit of the built library for the DPP-under-exec-mask trap: clang compiles `c ? dpp(v) : r` into a DPP move inside a
short exec-mask region, and a DPP / bpermute read of a source lane that the region DISABLED returns 0 - silently.  Lists every cross-lane instruction
that sits in an exec-mask region of at most `maxlen` instructions (the predicated-select shape; long structured regions such as `if (env < n)` around a
whole stage are not reported).  Usage: python tools/dpp_audit.py [lib.so] [maxlen]; exit code 1 when something is found."""
import os, re, shutil, subprocess, sys, tempfile

BIN = "/opt/rocm/lib/llvm/bin"


def disassemble(lib):
    d = tempfile.mkdtemp()
    try:
        shutil.copy(lib, os.path.join(d, "lib.so"))
        subprocess.run([os.path.join(BIN, "llvm-objdump"), "--offloading", "lib.so"], cwd=d, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        cos = [os.path.join(d, f) for f in os.listdir(d) if "gfx950" in f]
        return "\n".join(subprocess.run([os.path.join(BIN, "llvm-objdump"), "-d", "--mcpu=gfx950", c], capture_output=True, text=True).stdout for c in cos)
    finally:
        shutil.rmtree(d, ignore_errors=True)


def audit(asm, maxlen=16):
    func, stack, hits = None, [], []
    for ln in asm.split("\n"):
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", ln)
        if m:
            func, stack = m.group(1), []
            continue
        t = ln.strip().split()
        if not t or not re.match(r"^[a-z]", t[0]):
            continue
        op = t[0]
        for r in stack:
            r["n"] += 1
            if "_dpp" in op or op.startswith("ds_bpermute") or op.startswith("ds_swizzle") or op.startswith("ds_permute"):
                r["x"].append(ln.strip()[:100])
        if "saveexec" in op:
            stack.append({"n": 0, "x": []})
        elif op.startswith("s_or_b64") and len(t) > 1 and t[1].startswith("exec") and stack:
            r = stack.pop()
            if r["x"] and r["n"] <= maxlen:
                hits.append((func, r["n"], r["x"]))
        elif op.startswith("s_mov_b64") and len(t) > 1 and t[1].startswith("exec"):
            stack = []
    return hits


if __name__ == "__main__":
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "apex_amd", "lib", "libapx.so")
    hits = audit(disassemble(lib), int(sys.argv[2]) if len(sys.argv) > 2 else 16)
    for f, n, x in hits:
        print("%s: exec-mask region of %d instructions holds" % (f, n))
        for i in x:
            print("    " + i)
    print("%d suspicious region(s)" % len(hits))
    sys.exit(1 if hits else 0)
