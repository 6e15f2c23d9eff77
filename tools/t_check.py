"""Run the env kernels of the range-checked build (make -C apex_amd/csrc VARIANT=check EXTRA=-DAPX_CHECK; APX_LIB=apex_amd/lib/libapx_check.so) through every
entry point and terrain / command-profile / env-kind variant on falling robots (random actions: contacts, limits, leg-leg pairs, resets), then read the first
out-of-range S(f) / S.W(i) / S.I(f) index the kernels saw.  Prints `oob [kind, index, env, lane]`; kind 0 = none."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
LIB = os.environ.get("APX_LIB", "")
assert "check" in LIB or "asan" in LIB, "run with APX_LIB=<path to libapx_check.so> (or libapx_asan.so under tools/t_asan.sh: the sanitizer reports by itself)"
from apex_amd.vecenv import CassieVecEnv
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
torch.manual_seed(0)
worst = None
for kw in (dict(), dict(command_profile="phase"), dict(env_name="CassieTraj-v0"), dict(dynamics_randomization=False, est_lifetime=0)):
    env = CassieVecEnv(n_envs=256, seed=9, max_traj_len=40, **kw)
    env.reset()
    for t in range(steps):
        env.step(torch.randn(256, 10, device="cuda") * (0.2 + 0.4 * (t % 3)))
    if not kw:
        rng = np.random.RandomState(0)
        env.set_hfield(rng.rand(60, 60).astype(np.float32), size=(3.0, 3.0, 0.2))
        for t in range(steps // 2):
            env.step(torch.randn(256, 10, device="cuda") * 0.3)
        env.set_hfield(None)
    env.reset_for_test(full_reset=True); env.step_basic(torch.zeros(256, 10, device="cuda")); env.reset_for_test(); env.update_speed(1.0)
    env.apply_force(torch.tensor([50.0, 0, 0, 0, 0, 0])); env.step(torch.zeros(256, 10, device="cuda"))
    oob = env.get_field("oob")[0, :4].cpu().numpy() if "check" in LIB else np.zeros(4)
    print("variant %s: oob %s" % (kw, oob.astype(int).tolist()))
    if oob[0] != 0 and worst is None: worst = oob
print("RESULT", "clean" if worst is None else worst.astype(int).tolist())
