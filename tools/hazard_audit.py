"""Static audit of the built library for gfx9 / gfx940 VALU data hazards around the instructions the env kernels issue through INLINE ASSEMBLY (apex_amd/csrc/gfx950/lane_ops.h):
the compiler's hazard recogniser does not look inside an asm statement, so the wait states of those sequences are kept by hand (the order of the statements, `s_nop`s, fences).
This walks the disassembly of every gfx950 code object and checks, inside each basic block:
  dpp    a VGPR read through DPP (src0 of a *_dpp instruction) was written by a VALU instruction at least 2 wait states earlier (GCNHazardRecognizer::checkDPPHazards, DppVgprWaitStates);
  dppx   a DPP instruction sits at least 5 wait states behind a VALU write of EXEC (v_cmpx*; DppExecWaitStates);
  trans  a non-transcendental VALU instruction reads the result of v_rcp / v_rsq / v_sqrt / v_exp / v_log / v_sin / v_cos no sooner than 1 wait state later (gfx940 trans forwarding);
  lane   v_readlane / v_writelane with a lane-select SGPR written by a VALU instruction at least 4 wait states earlier.
A wait state = one issued instruction; `s_nop N` counts N + 1.  Block boundaries (function entry, branch targets, the instruction behind a branch) reset the history: the
audit is about straight-line sequences, which is where the hand-kept distances live.  Usage: python tools/hazard_audit.py [lib.so]; exit code 1 when something is found."""
import os
import re
import shutil
import subprocess
import sys
import tempfile

BIN = "/opt/rocm/lib/llvm/bin"
TRANS = re.compile(r"^v_(rcp|rsq|sqrt|exp|log|sin|cos)_(f32|f16|f64|legacy_f32|iflag_f32)")
FAR = 1 << 20


def code_objects(lib):
    d = tempfile.mkdtemp()
    shutil.copy(lib, os.path.join(d, "lib.so"))
    subprocess.run([os.path.join(BIN, "llvm-objdump"), "--offloading", "lib.so"], cwd=d, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return d, [os.path.join(d, f) for f in os.listdir(d) if "gfx950" in f]


def disassemble(lib):
    d, cos = code_objects(lib)
    try:
        return "\n".join(subprocess.run([os.path.join(BIN, "llvm-objdump"), "-d", "--mcpu=gfx950", c], capture_output=True, text=True).stdout for c in cos)
    finally:
        shutil.rmtree(d, ignore_errors=True)


def regs(tok):
    """VGPR numbers named by an operand token: v12 -> [12], v[4:7] -> [4..7], anything else -> []"""
    tok = tok.strip().lstrip("-|").rstrip(",|")
    m = re.match(r"^v(\d+)$", tok)
    if m:
        return [int(m.group(1))]
    m = re.match(r"^v\[(\d+):(\d+)\]$", tok)
    if m:
        return list(range(int(m.group(1)), int(m.group(2)) + 1))
    return []


def sregs(tok):
    tok = tok.strip().rstrip(",")
    m = re.match(r"^s(\d+)$", tok)
    if m:
        return [int(m.group(1))]
    m = re.match(r"^s\[(\d+):(\d+)\]$", tok)
    if m:
        return list(range(int(m.group(1)), int(m.group(2)) + 1))
    return {"vcc": [106, 107], "vcc_lo": [106], "vcc_hi": [107]}.get(tok, [])


def parse(asm):
    """[(function, address, opcode, [operand tokens], text)] + the set of branch-target addresses"""
    ins, targets, func, base = [], set(), None, {}
    for ln in asm.split("\n"):
        m = re.match(r"^([0-9a-f]+) <(\S+)>:", ln)
        if m:
            func = m.group(2); base[func] = int(m.group(1), 16)
            continue
        if "//" not in ln or func is None:
            continue
        code, tail = ln.split("//", 1)
        t = code.strip().split(None, 1)
        if not t or not re.match(r"^[a-z]", t[0]):
            continue
        try:
            addr = int(tail.strip().split(":")[0], 16)
        except ValueError:
            continue
        ops = [o.strip() for o in re.split(r",\s*|\s+(?=row_|quad_|bank_|bound_|op_sel|neg_|clamp|mul:|div:|dst_sel|src\d_sel)", t[1])] if len(t) > 1 else []
        ins.append((func, addr, t[0], ops, code.strip()))
        if t[0].startswith("s_cbranch") or t[0] == "s_branch":
            m = re.search(r"<(\S+)\+0x([0-9a-f]+)>", tail)
            if m and m.group(1) in base:
                targets.add(base[m.group(1)] + int(m.group(2), 16))
            elif re.search(r"<(\S+)>\s*$", tail) and re.search(r"<(\S+)>\s*$", tail).group(1) in base:
                targets.add(base[re.search(r"<(\S+)>\s*$", tail).group(1)])
    return ins, targets


def audit(asm):
    ins, targets = parse(asm)
    hits = []
    vw, vtrans, sw = {}, {}, {}          # register -> wait states since a VALU wrote it (absent = far)
    since_execw = FAR
    cur = None
    for func, addr, op, ops, text in ins:
        if func != cur or addr in targets:
            vw, vtrans, sw, since_execw, cur = {}, {}, {}, FAR, func
        is_valu = op.startswith("v_") and not op.startswith("v_readlane") and not op.startswith("v_writelane") and not op.startswith("v_readfirstlane") or op.startswith("v_writelane")
        # ---- checks on this instruction's reads
        if "_dpp" in op and len(ops) >= 2:
            for r in regs(ops[1]):
                if vw.get(r, FAR) < 2:
                    hits.append(("dpp", func, addr, text, "v%d written %d wait state(s) earlier" % (r, vw[r])))
            if since_execw < 5:
                hits.append(("dppx", func, addr, text, "EXEC written by a VALU instruction %d wait state(s) earlier" % since_execw))
        if op.startswith("v_") and not TRANS.match(op):
            for o in ops[1:]:
                for r in regs(o):
                    if vtrans.get(r, FAR) < 1:
                        hits.append(("trans", func, addr, text, "v%d is the result of a transcendental issued right in front" % r))
        if (op.startswith("v_readlane") or op.startswith("v_writelane")) and len(ops) >= 3:
            for r in sregs(ops[2]):
                if sw.get(r, FAR) < 4:
                    hits.append(("lane", func, addr, text, "lane select s%d written by a VALU instruction %d wait state(s) earlier" % (r, sw[r])))
        # ---- advance the clocks
        n = 1
        if op == "s_nop" and ops:
            try:
                n = int(ops[0], 0) + 1
            except ValueError:
                n = 1
        for d_ in (vw, vtrans, sw):
            for k in list(d_):
                d_[k] += n
                if d_[k] > 8:
                    del d_[k]
        since_execw = min(FAR, since_execw + n)
        # ---- record this instruction's writes
        if op.startswith("v_") and ops:
            if op.startswith("v_cmpx"):
                since_execw = 0
            dst = regs(ops[0])
            if not op.startswith("v_readlane") and not op.startswith("v_readfirstlane"):
                for r in dst:
                    vw[r] = 0
                    if TRANS.match(op):
                        vtrans[r] = 0
                    else:
                        vtrans.pop(r, None)
            if op.startswith("v_cmp") or op.startswith("v_readlane") or op.startswith("v_readfirstlane") or "_co_" in op:      # VALU writes of SGPRs (compare masks, carry out, lane reads)
                for o in ops[:2]:
                    for r in sregs(o):
                        sw[r] = 0
        if op.startswith("s_cbranch") or op in ("s_branch", "s_setpc_b64", "s_swappc_b64", "s_endpgm"):
            vw, vtrans, sw, since_execw = {}, {}, {}, FAR
    return hits


if __name__ == "__main__":
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "apex_amd", "lib", "libapx.so")
    asm = disassemble(lib)
    hits = audit(asm)
    ndpp = sum(1 for ln in asm.split("\n") if "_dpp" in ln.split("//")[0])
    for h in hits[:40]:
        print("%-5s %s +0x%x: %s   <- %s" % (h[0], h[1][:60], h[2], h[3][:110], h[4]))
    print("%d hazard(s) in %d DPP instructions / %d lines" % (len(hits), ndpp, asm.count("\n")))
    sys.exit(1 if hits else 0)
