import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from apex_amd.vecenv import CassieVecEnv
dev = torch.device("cuda", 0)
a = CassieVecEnv(n_envs=256, seed=31, max_traj_len=12); b = CassieVecEnv(n_envs=256, seed=31, max_traj_len=12)
a.reset(); b.reset()
g = torch.Generator(device=dev); g.manual_seed(4)
names = ("qpos", "qvel", "qacc_warm", "mass", "damping", "friction", "floor", "body_invweight0", "dof_invweight0", "motor_noise", "joint_noise", "snap", "cmd", "fwd", "tq_fifo", "menc", "so_tacc", "so_tvel", "so_height", "foot_prev")
for t in range(14):
    a.prepare_resets()
    act = torch.randn(256, 10, device=dev, generator=g) * 0.3
    oa, ra, da, fa = a.step(act); ob, rb, db, fb = b.step(act)
    if not torch.equal(oa, ob):
        bad = (oa != ob).any(1).nonzero().view(-1)
        print("t", t, "envs differing", bad.tolist()[:10], "done of those", da[bad].tolist()[:10], "n done", int((da != 0).sum()))
        i = int(bad[0]); print("cols", (oa[i] != ob[i]).nonzero().view(-1).tolist(), (oa[i] - ob[i]).abs().max().item())
        for nm in names:
            x, y = a.get_field(nm), b.get_field(nm)
            if not torch.equal(x, y):
                print("  field", nm, "differs on", int((x != y).any(1).sum()), "envs; max", (x - y).abs().max().item(), "cols", (x[i] != y[i]).nonzero().view(-1).tolist()[:12])
        ia, ib = a.get_field("ints_bits").view(torch.int32), b.get_field("ints_bits").view(torch.int32)
        print("  ints", ia[i].tolist(), ib[i].tolist())
        break
