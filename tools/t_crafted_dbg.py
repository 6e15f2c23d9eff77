"""Debug aid: the crafted single-substep comparison of tests/test_gpu_env.py::test_single_substep_crafted_states, printing per case the kernel's saturation report and
the error of qacc against the COMPLETE oracle and against the oracle switched to the lane map's caps (APX_LIB / APX_COMPLETE_ROWS select the build / the path)."""
import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from tests import test_gpu_env as T
from oracle import sim as S
np.set_printoptions(linewidth=250, precision=3, threshold=100000)
N = T.N
genv, oenv = T._mk(False, 11)
ocap = [S.OracleEnv(dyn_rand=False, seed=11, env_id=i) for i in range(12)]
genv.reset(); [e.reset() for e in oenv[:12]]; [e.reset() for e in ocap]; [e.kernel_caps(True) for e in ocap]
rng = np.random.RandomState(5)
qpos = genv.get_field("qpos").cpu().numpy().astype(np.float64); qvel = genv.get_field("qvel").cpu().numpy().astype(np.float64)
cases = []
for i in range(12):
    q = qpos[i].copy(); v = 0.05 * rng.randn(32); kind = i % 4
    if kind == 0: q[2] = 0.80 + 0.02 * rng.rand()
    elif kind == 1:
        q[2] = 0.45 + 0.05 * rng.rand(); ang = 0.9 + 0.3 * rng.rand(); q[3:7] = [np.cos(ang / 2), 0.0, np.sin(ang / 2), 0.0]
    elif kind == 2:
        q[2] = 1.2; q[7] = 0.45; q[14] = -0.60; q[20] = -0.45; q[21] = -0.45; q[28] = -2.95; q[34] = -2.50
    else:
        q[2] = 0.50 + 0.05 * rng.rand(); ang = 0.7; q[3:7] = [np.cos(ang / 2), np.sin(ang / 2), 0.0, 0.0]
    cases.append((q, v)); qpos[i] = q; qvel[i] = v
genv.set_field("qpos", torch.tensor(qpos, dtype=torch.float32)); genv.set_field("qvel", torch.tensor(qvel, dtype=torch.float32))
genv.set_field("qacc_warm", torch.zeros(N, 32))
s0, c0 = (x.cpu().numpy().copy() for x in genv.saturation())
genv.substep()
s1, c1 = (x.cpu().numpy() for x in genv.saturation())
qa = genv.get_field("qacc_warm").cpu().numpy()
for i in range(12):
    q, v = cases[i]
    out = []
    for e in (oenv[i], ocap[i]):
        e.set("qpos", q.astype(np.float32).astype(np.float64)); e.set("qvel", v.astype(np.float32).astype(np.float64)); e.set("qacc_warm", np.zeros(32))
        e.substep()
        ref_a = e.get("qacc_warm"); scale = np.maximum(1.0, np.abs(ref_a))
        err = np.abs(qa[i] - ref_a) / scale
        out.append((err.max(), int(err.argmax())))
    ii = oenv[i].get("ints")
    print("case %2d kernel sat flags %2d passes %d | oracle ncon %d nefc %d sat %d | err vs complete %.3g (dof %d), vs capped %.3g (dof %d)" % (
        i, int(s1[i]) & 31, int(c1[i] - c0[i]), int(ii[3]), int(ii[4]), int(ii[8]), out[0][0], out[0][1], out[1][0], out[1][1]))
