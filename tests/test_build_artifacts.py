"""Checks on the BUILT library that need no GPU: registers / scratch of the env kernels and the presence of the divergence guards in the ISA.
Both were regressions that no parity test saw: 44 B of scratch from a struct copy under a branch (round 3), and the bit-pattern NaN tests of round 2
folded to `false` by -ffast-math.  Reads apex_amd/lib/libapx.so with the ROCm binutils (llvm-objdump / llvm-readelf under /opt/rocm/lib/llvm/bin)."""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(REPO, "apex_amd", "lib", "libapx.so")
BIN = "/opt/rocm/lib/llvm/bin"


@pytest.fixture(scope="module")
def code_objects():
    if not os.path.exists(LIB):
        pytest.skip("libapx.so not built")
    if not os.path.exists(os.path.join(BIN, "llvm-objdump")):
        pytest.skip("ROCm binutils not present")
    d = tempfile.mkdtemp()
    shutil.copy(LIB, os.path.join(d, "lib.so"))
    subprocess.run([os.path.join(BIN, "llvm-objdump"), "--offloading", "lib.so"], cwd=d, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    cos = [os.path.join(d, f) for f in os.listdir(d) if "gfx950" in f]
    assert cos, "no gfx950 code object in libapx.so"
    yield cos
    shutil.rmtree(d, ignore_errors=True)


# the out-of-line COLD functions of the env kernels (round 5): the complete-row path of a saturated forward pass, the restart of finished envs inside the one-launch
# rollout, the recurrent actor's step inside it.  They are called where nothing of the hot loop is live, keep their callee-saved registers in scratch and are the only reason a kernel has a private segment.
COLD_FUNCTIONS = ("substep_complete", "rollout_restart", "rollout_lstm_actor")
CODE_BYTES_CEILING = {"env_step_kernel": 100 * 1024, "env_substep_kernel": 100 * 1024, "env_rollout_kernel": 110 * 1024, "stage1b_tree_lane": 20 * 1024}      # shipped: 94.8 / 77.0 / 101.7 / 17.8 KB (plane variants)


def _functions(code_objects):
    """{symbol: [instruction lines]} of the env code object"""
    big = max(code_objects, key=os.path.getsize)
    asm = subprocess.run([os.path.join(BIN, "llvm-objdump"), "-d", "--mcpu=gfx950", big], capture_output=True, text=True).stdout
    out = {}
    for blk in re.split(r"\n(?=[0-9a-f]+ <)", asm):
        m = re.match(r"[0-9a-f]+ <(\S+)>:", blk)
        if m:
            out[m.group(1)] = [ln for ln in blk.split("\n")[1:] if "//" in ln]
    return out


def test_env_kernels_use_no_scratch(code_objects):
    """No scratch instruction in the BODY of any env kernel (step / substep / reset / rollout, plane and height-field variants) nor in the tree stage: scratch in the
    2 kHz substep is HBM traffic per wave and per substep, and every codegen scare so far came with it.  (Until round 4 this read
    private_segment_fixed_size == 0; since round 5 the kernels call out-of-line COLD functions, whose frames make the segment non-zero: the check moved to the
    instructions themselves.)"""
    fns = _functions(code_objects)
    seen = 0
    for name, ins in fns.items():
        if ("env_" in name and "kernel" in name) or "stage1b_tree_lane" in name:
            if "env_init" in name or "env_update_speed" in name:
                continue
            seen += 1
            n = sum(1 for ln in ins if ln.split()[0].startswith("scratch_"))
            assert n == 0, (name, n)
    assert seen >= 11
    assert any(c in n for n in fns for c in COLD_FUNCTIONS), "the cold functions are out of line"


def test_hot_kernels_stay_under_their_code_size_ceiling(code_objects):
    """Code bytes of the hot kernels (tools/kernel_resources.sh prints them): the step loop streams from L2 through a 64 KB instruction cache every substep
    (+ 5-9 % beyond 64 KB, flat to 256 KB); a cold path inlined into it by accident - the complete-row solve, a second copy of the substep -
    shows here before it shows in the launch time."""
    fns = _functions(code_objects)
    for key, ceiling in CODE_BYTES_CEILING.items():
        hits = [(n, ins) for n, ins in fns.items() if key in n and "ILb1" not in n]
        assert hits, key
        for n, ins in hits:
            a0 = int(ins[0].split("//")[1].split(":")[0], 16); a1 = int(ins[-1].split("//")[1].split(":")[0], 16)
            assert a1 - a0 < ceiling, (n, a1 - a0, ceiling)


def test_divergence_guards_survive_fast_math(code_objects):
    """The non-finite tests (c4::fbits) must be in the ISA of the step kernels: an exponent-mask compare with 0x7f800000.  A plain bitcast test is folded
    away under -ffast-math, which is how round 2 shipped without them."""
    big = max(code_objects, key=os.path.getsize)
    asm = subprocess.run([os.path.join(BIN, "llvm-objdump"), "-d", "--mcpu=gfx950", big], capture_output=True, text=True).stdout
    blocks = re.split(r"\n(?=[0-9a-f]+ <)", asm)
    step = [b for b in blocks if re.match(r"[0-9a-f]+ <_Z15env_step_kernel", b)]
    assert len(step) == 2
    for b in step:
        assert b.count("0x7f800000") >= 4, b[:80]


def test_no_cross_lane_read_under_a_select_mask():
    """tools/dpp_audit.py on the product build: no DPP / bpermute instruction inside a short exec-mask region.  clang compiles `c ? dpp(v) : r` to a DPP
    move under the mask of `c`, and a cross-lane read of a lane that the mask disabled returns 0 - silently (round 3: a pose fetch written that way produced
    accelerations off by 1000 x; the most likely cause of the two unexplained faults of round 2)."""
    import sys
    if not os.path.exists(LIB) or not os.path.exists(os.path.join(BIN, "llvm-objdump")):
        pytest.skip("library or ROCm binutils not present")
    sys.path.insert(0, os.path.join(REPO, "tools"))
    import dpp_audit
    hits = dpp_audit.audit(dpp_audit.disassemble(LIB), 16)
    assert not hits, hits[:3]


def test_dpp_audit_flags_the_shape_it_is_meant_to_find():
    """Positive control for tools/dpp_audit.py on a hand-written listing: a DPP move between s_and_saveexec and the matching s_or (the predicated select) is
    reported, the same move outside a region or inside a long structured region is not."""
    import sys
    sys.path.insert(0, os.path.join(REPO, "tools"))
    import dpp_audit
    bad = """
0000000000001000 <kern_a>:
	v_cmp_eq_u32_e32 vcc, 6, v1
	s_and_saveexec_b64 s[4:5], vcc
	v_mov_b32_dpp v21, v20 row_shr:6 row_mask:0xf bank_mask:0xf bound_ctrl:1
	s_or_b64 exec, exec, s[4:5]
	v_add_f32_e32 v2, v21, v3
"""
    good = """
0000000000002000 <kern_b>:
	v_mov_b32_dpp v21, v20 row_shr:6 row_mask:0xf bank_mask:0xf bound_ctrl:1
	s_and_saveexec_b64 s[4:5], vcc
""" + "\tv_add_f32_e32 v2, v2, v3\n" * 40 + """	v_mov_b32_dpp v22, v20 row_newbcast:3 row_mask:0xf bank_mask:0xf bound_ctrl:1
	s_or_b64 exec, exec, s[4:5]
"""
    hits = dpp_audit.audit(bad + good, 16)
    assert len(hits) == 1 and hits[0][0] == "kern_a" and "row_shr:6" in hits[0][2][0]


def test_epoch_kernel_fences_and_operand_batches(code_objects):
    """ppo_small_epoch_kernel (apx_ppo_epoch): (a) its grid barrier must carry the agent-scope release / acquire cache maintenance that makes one XCD's stores visible
    to the others (buffer_wbl2 sc1 before the arrival, buffer_inv sc1 after the wait) - a barrier without them passes every single-XCD check and reads stale L2
    lines on the real chip; (b) the operands of a K = 256 tile are fetched as ONE batch of 32 16-byte loads in front of its 64 MFMAs - left alone the scheduler sinks
    each load to just before its MFMAs and a tile costs 16 L2 round trips instead of one; (c) no scratch instruction."""
    for co in code_objects:
        asm = subprocess.run([os.path.join(BIN, "llvm-objdump"), "-d", "--mcpu=gfx950", co], capture_output=True, text=True).stdout
        if "ppo_small_epoch_kernel" in asm:
            break
    else:
        pytest.fail("ppo_small_epoch_kernel not in libapx.so")
    body = [b for b in re.split(r"\n(?=[0-9a-f]+ <)", asm) if re.match(r"[0-9a-f]+ <\S*ppo_small_epoch_kernel", b)][0]
    ins = [ln.split()[0] + " " + " ".join(ln.split("//")[0].split()[1:]) for ln in body.split("\n")[1:] if "//" in ln]
    assert sum(1 for i in ins if i.startswith("buffer_wbl2") and "sc1" in i) >= 7
    assert sum(1 for i in ins if i.startswith("buffer_inv") and "sc1" in i) >= 7
    assert not any(i.startswith("scratch_") for i in ins)
    ops = [i.split()[0] for i in ins if i.split()[0] in ("global_load_dwordx4", "global_load_dword") or i.startswith("v_mfma")]
    runs, prev, n = [], None, 0
    for o in ops + [None]:
        k = "mfma" if o and o.startswith("v_mfma") else o
        if k == prev:
            n += 1
        else:
            if prev:
                runs.append((prev, n))
            prev, n = k, 1
    batched = [i for i in range(len(runs) - 1) if runs[i] == ("global_load_dwordx4", 32) and runs[i + 1][0] == "mfma"]
    assert len(batched) >= 1, runs      # phase B: both operands contiguous in K
    assert sum(n for k, n in runs if k == "mfma") >= 200


def test_td3_updates_kernel_fences_and_no_scratch(code_objects):
    """td3_small_kernel (apx_td3_updates): the same grid barrier as the epoch kernel - agent-scope write-back before the arrival, invalidate after the wait - on each of
    its 19 barrier sites, MFMA tiles present, no scratch instruction."""
    for co in code_objects:
        asm = subprocess.run([os.path.join(BIN, "llvm-objdump"), "-d", "--mcpu=gfx950", co], capture_output=True, text=True).stdout
        if "td3_small_kernel" in asm:
            break
    else:
        pytest.fail("td3_small_kernel not in libapx.so")
    body = [b for b in re.split(r"\n(?=[0-9a-f]+ <)", asm) if re.match(r"[0-9a-f]+ <\S*td3_small_kernel", b)][0]
    ins = [ln.split()[0] + " " + " ".join(ln.split("//")[0].split()[1:]) for ln in body.split("\n")[1:] if "//" in ln]
    assert sum(1 for i in ins if i.startswith("buffer_wbl2") and "sc1" in i) >= 15
    assert sum(1 for i in ins if i.startswith("buffer_inv") and "sc1" in i) >= 15
    assert sum(1 for i in ins if i.startswith("global_atomic_add")) >= 15
    assert sum(1 for i in ins if i.startswith("v_mfma_f32_16x16x4")) >= 300
    assert not any(i.startswith("scratch_") for i in ins)


def test_no_valu_data_hazard_around_the_inline_assembly():
    """tools/hazard_audit.py on the product build: the compiler's hazard recogniser does not look inside the asm statements of gfx950/lane_ops.h (v_fmac_f32_dpp with a DPP
    source, v_rcp_f32_dpp, ...), so their wait states are kept by hand.  Every DPP read in every code object must sit at least 2 wait states behind a VALU write of the register
    it reads (the shipped env code object: 73 486 DPP instructions, 4 993 of them at exactly 2), 5 behind a VALU write of EXEC, a transcendental's result is not consumed by the
    next non-transcendental VALU instruction, a VALU-written lane select is 4 wait states old.  (What the host emulation of the kernels cannot see.)"""
    import sys
    sys.path.insert(0, os.path.join(REPO, "tools"))
    import hazard_audit
    if not os.path.exists(LIB):
        pytest.skip("libapx.so not built")
    asm = hazard_audit.disassemble(LIB)
    assert sum(1 for ln in asm.split("\n") if "_dpp" in ln.split("//")[0]) > 50000      # the audit saw the env kernels
    hits = hazard_audit.audit(asm)
    assert not hits, hits[:10]


def test_hazard_audit_flags_what_it_is_meant_to_find():
    """positive control on a hand-written listing: each of the four hazards once, next to its legal form"""
    import sys
    sys.path.insert(0, os.path.join(REPO, "tools"))
    import hazard_audit
    L = lambda a, s: "\t%s // %012X: 00000000" % (s, a)
    bad = "\n".join(["0000000000001000 <kern>:",
                     L(0x1000, "v_add_f32_e32 v5, v1, v2"), L(0x1004, "s_nop 0"), L(0x1008, "v_fmac_f32_dpp v9, v5, v3 row_newbcast:3 row_mask:0xf bank_mask:0xf"),      # 1 wait state: dpp
                     L(0x1010, "v_cmpx_gt_f32_e32 vcc, v1, v2"), L(0x1014, "s_nop 2"), L(0x1018, "v_mov_b32_dpp v7, v20 row_shr:1 row_mask:0xf bank_mask:0xf"),        # 3 wait states: dppx
                     L(0x1020, "v_rcp_f32_e32 v30, v1"), L(0x1024, "v_mul_f32_e32 v31, v30, v2"),                                                                    # trans
                     L(0x1028, "v_readfirstlane_b32 s9, v1"), L(0x102c, "s_nop 1"), L(0x1030, "v_readlane_b32 s10, v2, s9")])                                         # 2 wait states: lane
    good = "\n".join(["0000000000002000 <kern2>:",
                      L(0x2000, "v_add_f32_e32 v5, v1, v2"), L(0x2004, "s_nop 1"), L(0x2008, "v_fmac_f32_dpp v9, v5, v3 row_newbcast:3 row_mask:0xf bank_mask:0xf"),
                      L(0x2010, "v_rcp_f32_e32 v30, v1"), L(0x2014, "s_nop 0"), L(0x2018, "v_mul_f32_e32 v31, v30, v2"),
                      L(0x2020, "v_rcp_f32_e32 v40, v1"), L(0x2024, "v_rsq_f32_e32 v41, v40"),                                                                       # trans -> trans: no hazard
                      L(0x2028, "v_add_f32_e32 v50, v1, v2"), L(0x202c, "s_cbranch_scc1 3 // 00000000202C: BF850003 <kern2+0x3c>"),
                      L(0x2030, "v_mov_b32_dpp v7, v50 row_shr:1 row_mask:0xf bank_mask:0xf")])                                                                      # behind a branch: a new block
    hits = hazard_audit.audit(bad + "\n" + good)
    assert sorted(h[0] for h in hits) == ["dpp", "dppx", "lane", "trans"], hits
    assert all(h[1] == "kern" for h in hits)


def test_instruction_mix_ceilings_of_the_hot_env_kernels():
    """Ceilings on what the instruction diets of rounds 4 - 5 removed from the hot env kernels, so that a change cannot bring it back unnoticed (the kernels are VALU-issue bound:
    a select or a move costs what an FMA costs).  Shipped object: env_rollout_kernel<false,0> 16 788 instructions with 1 133 v_cndmask, 1 321 v_mov_b32, 503 s_nop and no scratch
    instruction; env_step_kernel<false> 15 275 / 1 135 / 1 205 / 442; the tree stage 2 565 with 642 packed (v_pk_*) instructions.  And a ceiling on the cold side's callee-saved
    register traffic (354 scratch instructions in rollout_restart<true>, 254 in substep_complete<false>): it may shrink, not grow."""
    import sys
    sys.path.insert(0, os.path.join(REPO, "tools"))
    import isa_diff
    if not os.path.exists(LIB):
        pytest.skip("libapx.so not built")
    F = isa_diff.functions(LIB)

    def one(frag):
        ks = [k for k in F if frag in k]
        assert len(ks) == 1, (frag, ks)
        return F[ks[0]]

    def count(ins, pred):
        return sum(1 for ln in ins if pred(ln.split()[0]))
    for frag, total, sel, mov, nop in (("env_rollout_kernelILb0ELi0", 16900, 1140, 1330, 510), ("env_step_kernelILb0", 15400, 1142, 1215, 450)):
        ins = one(frag)
        got = (len(ins), count(ins, lambda o: o.startswith("v_cndmask")), count(ins, lambda o: o.startswith("v_mov_b32")), count(ins, lambda o: o == "s_nop"))
        assert got[0] <= total and got[1] <= sel and got[2] <= mov and got[3] <= nop, (frag, got)
        assert count(ins, lambda o: o.startswith("scratch_")) == 0
    tree = one("stage1b_tree_lane")
    assert len(tree) <= 2600 and count(tree, lambda o: o.startswith("v_pk_")) >= 600
    assert count(one("rollout_restartILb1"), lambda o: o.startswith("scratch_")) <= 354 and count(one("substep_completeILb0"), lambda o: o.startswith("scratch_")) <= 254
