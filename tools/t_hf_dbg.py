"""Debug: the crafted heightfield substep of tests/test_gpu_env.py::test_heightfield_terrain_vs_oracle, per-dof qacc error
of the library named by APX_LIB against the fp64 oracle (and the fp32 control build when present)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from test_gpu_env import _terrain, N
import oracle.sim as S
from apex_amd.vecenv import CassieVecEnv

for kind in ("slope", "noise", "hills"):
    hf = _terrain(kind); size = (4.0, 4.0, 0.15)
    g = CassieVecEnv(n_envs=N, seed=4, dynamics_randomization=False, max_traj_len=1000)
    g.set_hfield(hf, size)
    o = [S.OracleEnv(dyn_rand=False, seed=4, env_id=i).set_hfield(hf, size) for i in range(6)]
    g.reset_for_test(); [e.reset_for_test() for e in o]
    q = g.get_field("qpos").cpu().numpy().astype(np.float64)
    rng = np.random.RandomState(3)
    for i in range(N):
        q[i, 0] = rng.uniform(-2.5, 2.5); q[i, 1] = rng.uniform(-2.5, 2.5)
        hh, _ = o[0].floor_query(q[i, 0], q[i, 1])
        q[i, 2] = 1.0 + hh + rng.uniform(-0.06, 0.0)
    g.set_field("qpos", torch.tensor(q, dtype=torch.float32)); g.set_field("qvel", torch.zeros(N, 32)); g.set_field("qacc_warm", torch.zeros(N, 32))
    g.substep()
    qa = g.get_field("qacc_warm").cpu().numpy()
    gi = g.get_field("ints").cpu().numpy() if hasattr(g, "get_field") else None
    for i, e in enumerate(o):
        e.set("qpos", q[i].astype(np.float32).astype(np.float64)); e.set("qvel", np.zeros(32)); e.set("qacc_warm", np.zeros(32))
        e.kernel_caps(True); e.substep()
        ref = e.get("qacc_warm"); scale = np.maximum(1.0, np.abs(ref))
        err = np.abs(qa[i] - ref) / scale
        k = int(np.argmax(np.where(np.isin(np.arange(32), [9, 22]), 0, err)))
        ints = e.get("ints")
        print("%s env %d ncon %d worst dof %d err %.3e (gpu %.5f ref %.5f) |ref|max %.1f  err[9,22]=%.2e %.2e" % (
            kind, i, int(ints[3]), k, err[k], qa[i, k], ref[k], np.abs(ref).max(), err[9], err[22]))
