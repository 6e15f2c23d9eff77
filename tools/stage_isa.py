"""Instruction mix per profiled stage: splits the disassembly of env_step_kernel<false> in a -DAPX_PROF=2 build at its probes (s_memtime) and
counts, for the straight-line code in front of each probe, instructions by class; joined with the probe cycles of tools/t_prof.py this gives
cycles per instruction per stage (1 wave per SIMD: a full-rate VALU instruction issues in 4 cycles on a 16-lane SIMD... the floor is ~4-5).
usage: python tools/stage_isa.py <disassembly.s of the code object> [prof2 log]"""
import re, sys, collections

COARSE = ["io_model", "tree_walk", "factor", "pgs_tail(z~)", "finish+euler", "rows(2 legs)", "gram+warm", "pgs_sweeps"]


def main(path, log=None):
    lines = open(path).read().split("\n")
    segs = []; cur = collections.Counter(); in_k = False; getpc = []; fn = None
    for ln in lines:
        m = re.match(r"^[0-9a-f]+ <(.*)>:", ln)
        if m:
            fn = m.group(1); in_k = ("env_step_kernelILb0" in fn) or ("stage1b_tree" in fn and "Lb0" in fn) or ("tree_lane" in fn)
            if in_k: segs.append(("== " + fn[:60], None)); cur = collections.Counter(); getpc = []
            continue
        if not in_k or "//" not in ln: continue
        ins = ln.split("//")[0].strip(); addr = int(ln.split("//")[1].split(":")[0], 16)
        op = ins.split()[0]
        if op == "s_getpc_b64": getpc.append([addr + 4, None]); continue
        if op == "s_add_u32" and getpc and getpc[-1][1] is None and "0x" in ins:
            getpc[-1][1] = getpc[-1][0] + int(ins.split(",")[-1].strip(), 16); continue
        if op in ("s_memtime", "s_memrealtime"):
            segs.append((dict(cur), getpc)); cur = collections.Counter(); getpc = getpc[-1:]; continue
        cur["n"] += 1; cur["bytes"] += 8 if len(ln.split("//")[1].split(":")[1].split()) > 1 else 4
        if op.startswith("v_"):
            cur["valu"] += 1
            if "dpp" in ins or "row_" in ins or "quad_perm" in ins: cur["dpp"] += 1
            if op.startswith("v_pk_"): cur["pk"] += 1
            if op.startswith(("v_rcp", "v_rsq", "v_sqrt", "v_sin", "v_cos", "v_exp", "v_log")): cur["trans"] += 1
            if op.startswith(("v_accvgpr", "v_mov")): cur["mov"] += 1
            if op.startswith("v_cndmask"): cur["sel"] += 1
        elif op.startswith("ds_"):
            cur["lds"] += 1; cur["lds_r" if ("read" in op or "load" in op) else "lds_w"] += 1
        elif op == "s_waitcnt": cur["wait"] += 1
        elif op == "s_nop": cur["nop"] += 1; cur["nopcyc"] += int(ins.split()[1]) + 1
        elif op.startswith("s_"): cur["salu"] += 1
        elif op.startswith(("global_", "buffer_", "flat_", "scratch_")): cur["vmem"] += 1
    cyc = {}
    if log:
        for ln in open(log):
            m = re.match(r"\s+\[\s*(\d+)\]\s+(.*?)\s+(\d+)$", ln)
            if m: cyc[int(m.group(1))] = (m.group(2), int(m.group(3)))
            m = re.match(r"(io_model|tree_walk|factor|pgs_tail\(z~\)|finish\+euler|rows\(2 legs\)|gram\+warm|pgs_sweeps)\s+(\d+) cycles", ln)
            if m: cyc[COARSE.index(m.group(1))] = (m.group(1) + " (tail of the stage behind its last fine probe)", int(m.group(2)))
    allt = collections.Counter(g[1] for c, gp in segs if gp for g in gp if g[1])
    last = allt.most_common(1)[0][0]; base = min(t for t in allt if t != last and abs(t - last) < 4096) if allt else 0
    print("%-52s %6s %6s %5s %5s %5s %5s %5s %5s %5s %6s %7s %5s" % ("segment (code in front of the probe)", "instr", "valu", "dpp", "pk", "sel", "mov", "ldsR", "ldsW", "wait", "nopcy", "cycles", "c/i"))
    for k, (c, gp) in enumerate(segs):
        if gp is None: print(c); continue
        # the slot address is loaded right after the s_memtime: first getpc target of the NEXT segment's list that is not g_prof_last
        nxt = segs[k + 1][1] if k + 1 < len(segs) and segs[k + 1][1] else []
        slot = [(g[1] - base) // 8 for g in (nxt or []) if g[1] and g[1] != last]
        slot = slot[0] if slot else -1
        name, cy = cyc.get(slot, ("slot %d" % slot, 0))
        print("%-52s %6d %6d %5d %5d %5d %5d %5d %5d %5d %6d %7d %5.1f  %dB" % (("[%d] " % slot + name)[:52], c.get("n", 0), c.get("valu", 0), c.get("dpp", 0), c.get("pk", 0), c.get("sel", 0), c.get("mov", 0), c.get("lds_r", 0), c.get("lds_w", 0), c.get("wait", 0), c.get("nopcyc", 0), cy, cy / max(1, c.get("n", 0)), c.get("bytes", 0)))

if __name__ == "__main__":
    main(*sys.argv[1:])
