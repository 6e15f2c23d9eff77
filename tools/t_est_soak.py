"""Soak of the state estimator's fp32 records: 4096 envs, random-policy rollouts with auto-reset (robots fall all the time), then a census of
non-finite values and of the filters' covariance diagonals."""
import sys, os, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from apex_amd.vecenv import CassieVecEnv
from tests.state_xfer import est_from_record
env = CassieVecEnv(n_envs=4096, seed=5)
obs = env.reset()
torch.manual_seed(0)
for t in range(int(sys.argv[1]) if len(sys.argv) > 1 else 400):
    obs, rew, done, _ = env.step(torch.randn(4096, 10, device="cuda") * 0.3)
    if t % 50 == 49:
        est = env.get_field("est").cpu().numpy()
        bad = ~np.isfinite(est).all(1)
        print("step %4d: envs with a non-finite estimator record %d, non-finite obs rows %d, |vel| max %.2f, height min/max %.2f %.2f" % (
            t + 1, bad.sum(), int((~torch.isfinite(obs).all(1)).sum()), float(obs[:, 15:18].abs().max()), float(obs[:, 0].min()), float(obs[:, 0].max())))
est = env.get_field("est").cpu().numpy()
r = [est_from_record(e) for e in est[:256]]
dg = np.array([[np.diag(x["hP"][0]).min(), np.diag(x["hP"][0]).max(), np.diag(x["zP"]).min(), np.diag(x["zP"]).max()] for x in r])
print("covariance diagonals over 256 envs: x filter min %.2e max %.2e; z filter min %.2e max %.2e" % (dg[:, 0].min(), dg[:, 1].max(), dg[:, 2].min(), dg[:, 3].max()))
