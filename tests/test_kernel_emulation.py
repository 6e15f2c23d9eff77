"""CPU: the SOURCE of the epoch kernel (apex_amd/csrc/ppo_small.hip) compiled for the host under a lane-exact emulation of its wave collectives
(tools/hipemu: one workgroup, 4 waves x 64 fibers, v_mfma_f32_16x16x4_f32 operand / accumulator layout, __shfl, __syncthreads) against the reference's own
outputs (golden G4b: an epoch of small-minibatch PPO steps on the 2 x 256 networks), with 1 and with 3 workgroups (a workgroup = a forked process on MAP_SHARED
buffers, so the kernel's own grid barrier, its per-workgroup partial sums and the dealing of tiles over the grid run as written).  What this pins without a GPU: every
index, tile map, guard, the loss arithmetic, the gradient layout, clip + Adam, the scalar bookkeeping and the barrier counting of the kernel.  What it cannot see:
the cache coherence between XCDs behind the barrier's fences and the compiler's gfx950 code - those are the GPU test's
(tests/test_gpu_zz_first_hardware_run.py::test_ppo_epoch_one_launch)."""
import ctypes as C
import mmap
import os
import subprocess

import numpy as np
import pytest

from golden_util import EPOCH_CASES, epoch_case_inputs, check_slim
from oracle import learner as L

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU = os.path.join(REPO, "tools", "hipemu")
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


@pytest.fixture(scope="module")
def emu():
    if not os.path.exists(CLANG):
        pytest.skip("no host clang++")
    so = os.path.join(EMU, "_build", "libapx_emul.so")
    srcs = [os.path.join(EMU, "emul_ppo_small.cpp"), os.path.join(EMU, "hip", "hip_runtime.h"), os.path.join(REPO, "apex_amd", "csrc", "ppo_small.hip")]
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(s) for s in srcs):
        subprocess.check_call(["bash", os.path.join(EMU, "build.sh")])
    from apex_amd._lib import PpoArgs
    lib = C.CDLL(so)
    lib.apx_ppo_epoch.restype = C.c_int
    lib.apx_ppo_epoch.argtypes = [C.POINTER(PpoArgs), C.c_void_p, C.c_int64, C.c_void_p]
    lib.apx_ppo_epoch_workspace_bytes.restype = C.c_size_t
    lib.apx_ppo_epoch_workspace_bytes.argtypes = [C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int]
    lib.apx_emul_last_error.restype = C.c_char_p
    return lib


def _flat(params):
    return np.ascontiguousarray(np.concatenate([np.asarray(p, np.float32).ravel() for p in params]))


_KEEP = []
_PAGE = mmap.PAGESIZE
_libc = C.CDLL(None, use_errno=True)
_libc.mprotect.argtypes = [C.c_void_p, C.c_size_t, C.c_int]


def _aligned(n, dtype):
    """zeroed MAP_SHARED buffer (the emulation's workgroups are forked processes) that ENDS at an inaccessible guard page (its size rounded up to 16 bytes: the kernel
    wants 16-byte aligned blocks): a read or write of the kernel beyond any buffer it is handed is a SIGSEGV here, as it would be a memory fault on the GPU"""
    nbytes = n * np.dtype(dtype).itemsize
    span = (max(nbytes, 1) + 15) // 16 * 16
    pages = (span + _PAGE - 1) // _PAGE
    m = mmap.mmap(-1, (pages + 1) * _PAGE, flags=mmap.MAP_SHARED | mmap.MAP_ANONYMOUS)
    _KEEP.append(m)
    whole = np.frombuffer(m, dtype=np.uint8)
    base = whole.ctypes.data
    assert _libc.mprotect(base + pages * _PAGE, _PAGE, 0) == 0, "mprotect"
    off = pages * _PAGE - span
    return whole[off:off + nbytes].view(dtype)

def _shared(a):
    out = _aligned(a.size, a.dtype).reshape(a.shape)
    out[...] = a
    return out


def _signed_perm(mirrored):
    from apex_amd.engine import signed_perm_from_mirror
    return np.ascontiguousarray(np.asarray(signed_perm_from_mirror(mirrored), np.int32))


def _run_case(lib, c, wgs):
    from apex_amd._lib import PpoArgs
    from tools.refprobe.common import MIRRORED_OBS_FULL_CLOCK, MIRRORED_ACTS
    mirror, mb, nb, adam_t0 = EPOCH_CASES[c]
    inp = epoch_case_inputs(c)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    bufs = {}
    for name, plist in (("actor", inp["actor"]), ("critic", inp["critic"])):
        flat = _flat(plist)
        for k in ("", "_m", "_v", "_grad"):
            bufs[name + k] = _aligned(flat.size, np.float32)
        bufs[name][:] = flat
    obs, act, ret, adv = (_shared(np.ascontiguousarray(inp[k])) for k in ("obs", "act", "ret", "adv"))
    mean, std = _shared(inp["obs_mean"]), _shared(inp["obs_std"])
    xn = (obs.astype(np.float32) - inp["obs_mean"]) / inp["obs_std"]
    old_mu = _shared(L.mlp_forward([np.asarray(w, np.float64) for w in inp["old"]], xn.astype(np.float64)).astype(np.float32))
    perm = _shared(np.ascontiguousarray(inp["perm"]))
    ws = _aligned(int(lib.apx_ppo_epoch_workspace_bytes(mb, nb, 50, 256, 10)), np.uint8)
    assert ws.size > 0
    scal = _aligned(nb * 6, np.float64).reshape(nb, 6)
    osp, asp = _shared(_signed_perm(MIRRORED_OBS_FULL_CLOCK)), _shared(_signed_perm(MIRRORED_ACTS))
    a = PpoArgs(actor=p(bufs["actor"]), actor_m=p(bufs["actor_m"]), actor_v=p(bufs["actor_v"]), actor_grad=p(bufs["actor_grad"]),
                critic=p(bufs["critic"]), critic_m=p(bufs["critic_m"]), critic_v=p(bufs["critic_v"]), critic_grad=p(bufs["critic_grad"]),
                D=50, H=256, A=10, obs=p(obs), act=p(act), ret=p(ret), adv=p(adv), old_mu=p(old_mu), idx=None, mb=mb,
                obs_mean=p(mean), obs_std=p(std), obs_sign_perm=p(osp) if mirror else None, clock_mask=(1 << 46) | (1 << 47),
                act_sign_perm=p(asp) if mirror else None, fixed_std=float(np.exp(-1.5)), clip=0.2, entropy_coeff=0.0, grad_clip=0.05, lr=1e-4, adam_eps=1e-5,
                mirror_coeff=0.4, adam_t=adam_t0, grad_only=0, workspace=p(ws), workspace_bytes=ws.size, scalars_out=p(scal))
    lib.apx_emul_set_workgroups(wgs)
    rc = lib.apx_ppo_epoch(C.byref(a), p(perm), nb, None)
    assert rc == 0, lib.apx_emul_last_error()
    assert lib.apx_emul_last_grid() == wgs
    return inp, scal, bufs


def _split(flat, shapes):
    out, o = [], 0
    for s in shapes:
        n = int(np.prod(s)); out.append(flat[o:o + n].reshape(s)); o += n
    return out


@pytest.mark.parametrize("c,wgs", [(c, 1) for c in range(len(EPOCH_CASES))] + [(0, 3), (4, 3), (2, 2)])
def test_epoch_kernel_source_reproduces_the_reference_g4b(emu, golden_dir, c, wgs):
    g = np.load(os.path.join(golden_dir, "g4b_epoch_h256.npz"))
    inp, scal, bufs = _run_case(emu, c, wgs)
    np.testing.assert_allclose(scal, g[f"c{c}_scalars"], rtol=1e-5, atol=2e-7)
    for name, ref_list in (("actor", inp["actor"]), ("critic", inp["critic"])):
        for i, w in enumerate(_split(bufs[name], [np.shape(x) for x in ref_list])):
            check_slim(w, g[f"c{c}_{name}1.{i}"], atol=2.5e-4, frac_tol=2e-6, frac=2e-3, err_msg=(c, name, i))


def test_guard_pages_fault_on_an_overrun():
    """positive control of the guard pages: one float beyond a buffer of this module's allocator kills the reading process"""
    import sys
    code = ("import sys, ctypes as C; sys.path[:0] = [%r, %r]; import numpy as np; import test_kernel_emulation as T; a = T._aligned(81418, np.float32); "
            "print(C.c_float.from_address(a.ctypes.data + 4 * 81417).value); sys.stdout.flush(); print(C.c_float.from_address(a.ctypes.data + 4 * 81420).value)"
            % (REPO, os.path.join(REPO, "tests")))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert r.stdout.strip() == "0.0" and r.returncode == -11, (r.returncode, r.stdout, r.stderr[-300:])


def test_epoch_kernel_reruns_with_three_workgroups_are_bit_identical(emu):
    """Three concurrent workgroups (processes the OS schedules as it likes), the same epoch three times: parameters, Adam moments and scalars must be bit-identical -
    a phase that reads what another workgroup is still writing (a missing barrier) shows up here as run-to-run noise; the fixed-order partial sums make the exact
    comparison legitimate."""
    outs = []
    for rep in range(3):
        inp, scal, bufs = _run_case(emu, 2, 3)
        outs.append((scal.copy(), {k: v.copy() for k, v in bufs.items()}))
    for scal, bufs in outs[1:]:
        assert np.array_equal(scal, outs[0][0])
        for k in bufs:
            assert np.array_equal(bufs[k], outs[0][1][k]), k


def _barrier_selftest(lib, wgs, phases, words):
    lib.apx_grid_barrier_selftest.restype = C.c_int
    lib.apx_grid_barrier_selftest.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    ws = _aligned(2 + words, np.uint32); res = _aligned(4, np.uint64)
    lib.apx_emul_set_workgroups(wgs)
    rc = lib.apx_grid_barrier_selftest(wgs, phases, words, ws.ctypes.data_as(C.c_void_p), res.ctypes.data_as(C.c_void_p), None)
    assert rc == 0, lib.apx_emul_last_error()
    return [int(x) for x in res], ws


def test_grid_barrier_selftest_with_eight_workgroups(emu):
    """The barrier's stress kernel (apx_grid_barrier_selftest, the GPU twin is tests/test_gpu_learner.py::test_grid_barrier_under_stress) with eight concurrent workgroups =
    eight processes of 256 fibers on one shared counter: 300 phases x 2 barriers, every word of every phase read back by every workgroup.  What it pins here: the arrival
    arithmetic (targets as multiples of the grid size on a counter that is never reset), writer rotation, that no workgroup runs ahead of a phase."""
    res, ws = _barrier_selftest(emu, 8, 300, 777)
    assert res == [0, 0, 300, 8 * 300], res
    assert int(ws[0]) == 8 * 2 * 300 and int(ws[1]) == 0      # arrivals counted, watchdog silent


def test_grid_barrier_selftest_with_one_workgroup(emu):
    """degenerate grid: one workgroup is its own writer in every phase; all 50 phases complete, nothing stale, the word block was written"""
    res, ws = _barrier_selftest(emu, 1, 50, 64)
    assert res == [0, 0, 50, 50]
    assert np.count_nonzero(ws[2:]) >= 60


def test_grid_barrier_under_thread_sanitizer():
    """The barrier's source with workgroups as THREADS under ThreadSanitizer (tools/hipemu/tsan_barrier.sh): 8 workgroups x 3000 phases and 3 x 20000, the word block
    written and read by plain accesses that only the barrier orders - no report; the same build with the arrival counted RELAXED instead of RELEASE is reported as a
    data race (positive control).  The shim's header states what this does not model (stand-alone fences, cache scopes: the GPU stress test's job)."""
    if not os.path.exists(CLANG):
        pytest.skip("no host clang++")
    r = subprocess.run(["bash", os.path.join(EMU, "tsan_barrier.sh")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    assert r.stdout.count("stale 0 watchdog 0") == 2 and "positive control: the barrier without its release is reported" in r.stdout, r.stdout
