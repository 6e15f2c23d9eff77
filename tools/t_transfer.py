"""Behavioural parity: a policy trained on the HIP env (apex.py ppo) is run, deterministically, in the fp64 CPU oracle.
usage: python tools/t_transfer.py <run dir with actor.pt> [speed]"""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import sim as S
path = sys.argv[1]; speed = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
policy = torch.load(os.path.join(path, "actor.pt"), weights_only=False)
policy.eval()
res = []
for i in range(6):
    e = S.OracleEnv(dyn_rand=(i % 2 == 1), seed=50, env_id=i)
    obs = e.reset()
    if i % 2 == 0:
        obs = e.reset_for_test(); e.update_speed(speed)
    ret, L = 0.0, 0
    for t in range(400):
        with torch.no_grad():
            a = policy(torch.tensor(obs, dtype=torch.float32), deterministic=True).numpy().astype(np.float64)
        obs, r, d = e.step(a); ret += r; L += 1
        if d: break
    res.append((i, "reset_for_test+speed %.1f" % speed if i % 2 == 0 else "training reset (random command, dyn. rand.)", L, round(ret, 2)))
    print(res[-1])
