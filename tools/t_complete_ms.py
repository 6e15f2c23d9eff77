"""Cost of one forward pass on the complete-row path (cassie_complete.h) against the capped fast path: every env of the launch put in the same crafted pose, one substep launch timed
(hip events), complete rows on / off."""
import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from apex_amd.vecenv import CassieVecEnv
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
ITERS = int(sys.argv[2]) if len(sys.argv) > 2 else 50      # PGS sweeps (1: the cost of building the rows alone)
env = CassieVecEnv(n_envs=N, seed=3, pgs_iters=ITERS); env.reset()
q0 = env.get_field("qpos").cpu().numpy().astype(np.float64)[0]
def case(**kw):
    q = q0.copy()
    for k, v in kw.items(): q[int(k[1:])] = v
    return q
fold = dict(q9=1.3, q23=1.3, q14=-2.4, q28=-2.4)
cases = [("standing", case()), ("feet pressed in (4 ends)", case(q2=0.70)), ("crouch q2=0.45", case(q2=0.45, q9=0.9, q23=0.9, q14=-2.0, q28=-2.0)), ("kneel q2=0.30", case(q2=0.30, **fold)),
         ("pelvis sphere on the floor", case(q2=0.10, **fold)), ("two limits on one leg", case(q2=1.5, q14=-2.9, q20=-2.5)), ("6 pairs", case(q2=1.5, q7=-0.15, q21=0.15))]
for name, q in cases:
    out = []
    for comp in (False, True):
        env.set_complete_rows(comp)
        ms = []
        for rep in range(4):
            env.set_field("qpos", torch.tensor(np.tile(q, (N, 1)), dtype=torch.float32)); env.set_field("qvel", torch.zeros(N, 32)); env.set_field("qacc_warm", torch.zeros(N, 32))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record(); env.substep(); e1.record(); torch.cuda.synchronize(); ms.append(e0.elapsed_time(e1))
        sat, cnt = env.saturation()
        out.append((min(ms), int(sat[0]) & 31))
    print("%-30s capped %.3f ms (flags %d)   complete %.3f ms   ratio %.1f" % (name, out[0][0], out[0][1], out[1][0], out[1][0] / out[0][0]))
