"""Symbolised summary of the lockstep checker's reports (tools/hipemu/hip/hip_runtime.h):
    APX_EMUL_LIB=tools/hipemu/_build/libapx_emul_lockstep.so python -m pytest tests/test_kernel_emulation_env.py -s 2> log ; python tools/hipemu/lockstep_report.py log
One line per distinct (source line of the access, source line of the other lane's access), innermost frames of apex_amd/csrc."""
import collections
import os
import re
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
LIB = os.path.join(REPO, "tools", "hipemu", "_build", "libapx_emul_lockstep.so")
SYM = "/opt/rocm/lib/llvm/bin/llvm-symbolizer"


def frames(addrs):
    out = subprocess.run([SYM, "-e", LIB, "--inlines", "-s", "-C"] + addrs, capture_output=True, text=True).stdout
    res = []
    for blk in out.strip().split("\n\n"):
        ln = [x for x in blk.split("\n") if x]
        locs = [ln[i + 1] for i in range(0, len(ln) - 1, 2)]
        locs = [x for x in locs if re.match(r"(cassie_|env|estimator|gfx950)", x)]
        res.append(" < ".join(locs[:3]) if locs else "?")
    return res


def main():
    pairs = collections.Counter()
    kind = {}
    for line in open(sys.argv[1]):
        m = re.search(r"hipemu lockstep: (.*?) on (\w+) word -?\d+: lane \d+ at \+(0x[0-9a-f]+) vs lane \d+ at \+(0x[0-9a-f]+)", line)
        if m:
            pairs[(m.group(3), m.group(4))] += 1; kind[(m.group(3), m.group(4))] = m.group(1)[:28] + " " + m.group(2)
    addrs = sorted({a for p in pairs for a in p})
    loc = dict(zip(addrs, frames(addrs))) if addrs else {}
    agg = collections.Counter()
    for (a, b), n in pairs.items():
        agg[(kind[(a, b)], loc[a], loc[b])] += n
    for (k, a, b), n in sorted(agg.items(), key=lambda kv: kv[0][1]):
        print("%-34s %s   <->   %s" % (k, a, b))
    print(len(agg), "distinct source pairs")


if __name__ == "__main__":
    main()
