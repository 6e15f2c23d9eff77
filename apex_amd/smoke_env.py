"""smoke(): one tiny invocation of the env hot path on cuda:0, checked against the fp64 oracle (the checker, not
the thing measured)."""
import numpy as np
import torch


def run(dev):
    from apex_amd.vecenv import CassieVecEnv
    from oracle import sim as S
    env = CassieVecEnv(n_envs=64, seed=11, device=dev.index or 0)
    obs = env.reset().cpu().numpy()
    orc = [S.OracleEnv(seed=11, env_id=i) for i in range(4)]
    ref = np.stack([e.reset() for e in orc])
    assert np.allclose(obs[:4], ref, rtol=1e-4, atol=3e-4), "reset observation mismatch vs oracle"
    act = (np.random.RandomState(0).randn(64, 10) * 0.1).astype(np.float32)
    o, r, d, _ = env.step(torch.tensor(act, device=dev), auto_reset=False)
    o, r, d = o.cpu().numpy(), r.cpu().numpy(), d.cpu().numpy()
    for i, e in enumerate(orc):
        oo, rr, dd = e.step(act[i].astype(np.float64))
        assert dd == d[i], "done flag mismatch vs oracle"
        assert abs(rr - r[i]) < 0.02 and np.allclose(o[i, :15], oo[:15], atol=5e-3), "env step mismatch vs oracle"
    env.close()
