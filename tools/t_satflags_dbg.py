"""Debug aid: the crafted cases of tests/test_gpu_env.py::test_saturation_flags_vs_oracle_crafted with the row counts of the complete oracle and the per-dof error of qacc."""
import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from tests import test_gpu_env as T
np.set_printoptions(linewidth=250, precision=3, threshold=100000, suppress=True)
N = T.N
genv, oenv = T._mk(False, 12)
genv.reset()
q0 = genv.get_field("qpos").cpu().numpy().astype(np.float64)[0]
def case(**kw):
    q = q0.copy()
    for k, v in kw.items(): q[int(k[1:])] = v
    return q
fold = dict(q9=1.3, q23=1.3, q14=-2.4, q28=-2.4)
cases = [("standing", case()), ("airborne", case(q2=1.5)), ("pelvis sphere on the floor", case(q2=0.10, **fold)), ("crossed legs", case(q2=1.5, q7=-0.10, q21=0.10)),
         ("legs apart", case(q2=1.5, q7=-0.08, q21=0.08)), ("feet brushing: two pairs", case(q2=1.5, q7=-0.115, q21=0.115)), ("6 pairs", case(q2=1.5, q7=-0.15, q21=0.15)),
         ("two limits on one leg", case(q2=1.5, q14=-2.9, q20=-2.5)), ("one limit per leg", case(q2=1.5, q14=-2.9, q34=-2.5)), ("feet pressed in", case(q2=0.70))]
qpos = np.tile(q0, (N, 1))
for i, (_, q) in enumerate(cases): qpos[i] = q
genv.set_field("qpos", torch.tensor(qpos, dtype=torch.float32)); genv.set_field("qvel", torch.zeros(N, 32)); genv.set_field("qacc_warm", torch.zeros(N, 32))
genv.substep()
sat, cnt = (x.cpu().numpy() for x in genv.saturation())
qa = genv.get_field("qacc_warm").cpu().numpy()
for i, (name, q) in enumerate(cases):
    e = oenv[i]; e.reset()
    e.set("qpos", q.astype(np.float32).astype(np.float64)); e.set("qvel", np.zeros(32)); e.set("qacc_warm", np.zeros(32))
    e.substep()
    ii = e.get("ints"); ref = e.get("qacc_warm"); scale = np.maximum(1.0, np.abs(ref))
    err = np.abs(qa[i] - ref) / scale
    print("%-28s kernel flags %2d | oracle ncon %2d nefc %2d ncon1 %d sat %d iters %d | max err %.3g at dof %d" % (name, int(sat[i]) & 31, int(ii[3]), int(ii[4]), int(ii[9]), int(ii[8]), int(e.get("solver")[0]), err.max(), err.argmax()))
    if err.max() > 0.03: print("   err", err); print("   ref", ref)
