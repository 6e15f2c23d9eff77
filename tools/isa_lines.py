"""Where the selects, moves and hazard nops of the step kernel come from: per SOURCE LINE counts of v_cndmask / v_mov_b32 / s_nop (and the instruction total)
in env_step_kernel<false> + the tree stage, from a `-gline-tables-only` build disassembled with `llvm-objdump -d -l`.
    make -C apex_amd/csrc VARIANT=lines EXTRA=-gline-tables-only
    python tools/isa_lines.py apex_amd/lib/libapx_lines.so [top_n]
A line marker of cassie_common.h (operators, helpers) is charged to the last marker of a stage file in front of it (objdump prints the innermost inlined location only)."""
import collections, os, re, subprocess, sys, tempfile

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
STAGE_FILES = ("cassie_lane.h", "env.hip", "estimator_lane.h")


def disassemble(lib):
    t = tempfile.mkdtemp(); subprocess.check_call(["cp", lib, t + "/lib.so"])
    subprocess.call([OBJDUMP, "--offloading", "lib.so"], cwd=t, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    co = max((f for f in os.listdir(t) if f.endswith("gfx950")), key=lambda f: os.path.getsize(t + "/" + f))
    return subprocess.check_output([OBJDUMP, "-d", "-l", t + "/" + co]).decode().split("\n")


def main(lib, top=25):
    cnt = {k: collections.Counter() for k in ("sel", "mov", "nop", "all")}
    tot = collections.Counter(); in_k = False; outer = "?"
    for ln in disassemble(lib):
        m = re.match(r"^[0-9a-f]+ <(.*)>:", ln)
        if m:
            in_k = "env_step_kernelILb0" in m.group(1) or "stage1b_tree" in m.group(1); continue
        if not in_k: continue
        m = re.match(r"^; (.*):(\d+)$", ln)
        if m:
            f = os.path.basename(m.group(1))
            if f in STAGE_FILES: outer = "%s:%s" % (f, m.group(2))
            continue
        if "//" not in ln: continue
        op = ln.split()[0]
        cnt["all"][outer] += 1; tot["all"] += 1
        k = "sel" if op.startswith("v_cndmask") else "mov" if op in ("v_mov_b32_e32", "v_mov_b64_e32", "v_accvgpr_write_b32", "v_accvgpr_read_b32") else "nop" if op == "s_nop" else None
        if k: cnt[k][outer] += 1; tot[k] += 1
    print("env_step_kernel<false> + stage1b_tree_lane: %d instructions, %d v_cndmask, %d v_mov / v_accvgpr, %d s_nop" % (tot["all"], tot["sel"], tot["mov"], tot["nop"]))
    src = {}
    for k, title in (("sel", "v_cndmask"), ("mov", "v_mov_b32 / v_accvgpr moves"), ("nop", "s_nop")):
        print("\n== top %d source lines by %s" % (top, title))
        for line, n in cnt[k].most_common(top):
            f, no = line.split(":") if ":" in line else (line, "0")
            if f not in src:
                p = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "apex_amd", "csrc", f)
                src[f] = open(p).read().split("\n") if os.path.exists(p) else []
            text = src[f][int(no) - 1].strip()[:110] if 0 < int(no) <= len(src[f]) else ""
            print("%5d  (%4d instr)  %-22s %s" % (n, cnt["all"][line], line, text))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 25)
