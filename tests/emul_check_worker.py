"""Worker of tests/test_kernel_emulation_env.py::test_emulated_range_checked_build: the env kernels' sources compiled for the host WITH -DAPX_CHECK (every S(f) / S.W(i) /
S.I(f) index range-checked, apex_amd/csrc/env_state.h) driven through the entry points, env kinds, command profiles, a height field, restarts, the complete-row path and
the one-launch rollout; prints the first out-of-range index the kernels saw (kind 0 = none).  APX_EMUL_LIB must point at tools/hipemu/_build/libapx_emul_check.so."""
import contextlib
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "tests")]
assert "check" in os.environ.get("APX_EMUL_LIB", "")
from test_kernel_emulation_learner import _NoStream, emulated_library      # noqa: E402
from apex_amd import _lib, engine, vecenv      # noqa: E402

lib = emulated_library(); lib.apx_emul_set_workgroups(0)
_lib._lib = lib
engine._need_gpu = lambda *ts: None; engine._stream = lambda: None; vecenv._stream = lambda: None
vecenv._device = lambda i: torch.device("cpu"); vecenv._on_device = lambda t: True
torch.cuda.current_stream = lambda *a, **k: _NoStream(); torch.cuda.synchronize = lambda *a, **k: None
torch.cuda.Stream = lambda *a, **k: _NoStream(); torch.cuda.Event = lambda *a, **k: _NoStream(); torch.cuda.stream = lambda s: contextlib.nullcontext()
from apex_amd.vecenv import CassieVecEnv      # noqa: E402

torch.manual_seed(0)
dev = torch.device("cpu")
worst = None
for kw in (dict(), dict(command_profile="phase"), dict(env_name="CassieTraj-v0"), dict(dynamics_randomization=False, est_lifetime=0)):
    env = CassieVecEnv(n_envs=64, seed=9, max_traj_len=3, **kw)
    env.reset()
    for t in range(5):
        env.step(torch.randn(64, 10) * (0.2 + 0.4 * (t % 3)))
    if not kw:
        rng = np.random.RandomState(0)
        env.set_hfield(rng.rand(60, 60).astype(np.float32), size=(3.0, 3.0, 0.2))
        for t in range(2):
            env.step(torch.randn(64, 10) * 0.3)
        env.set_hfield(None)
    env.reset_for_test(full_reset=True); env.step_basic(torch.zeros(64, 10)); env.reset_for_test(); env.update_speed(1.0)
    env.apply_force(torch.tensor([50.0, 0, 0, 0, 0, 0])); env.step(torch.zeros(64, 10))
    oob = env.get_field("oob")[0, :4].numpy()
    print("variant %s: oob %s" % (kw, oob.astype(int).tolist()))
    if oob[0] != 0 and worst is None:
        worst = oob
    last = env
from tests import test_gpu_env as G      # noqa: E402
G.test_single_substep_crafted_states(dev)      # the complete-row path (third capsule end, hip-pitch capsule, pelvis sphere, leg-leg pairs)
import test_kernel_emulation_env as TE      # noqa: E402
a = TE._small_ppo(dev, 64, 3, 2, 6); a.sample()      # one-launch rollout with in-kernel restarts
oob = a.env.get_field("oob")[0, :4].numpy()
print("crafted states + one-launch rollout: oob %s" % oob.astype(int).tolist())
if oob[0] != 0 and worst is None:
    worst = oob
print("RESULT", "clean" if worst is None else worst.astype(int).tolist())
