"""Child process of tests/test_gpu_learner.py::test_ppo_epoch_*: drives apx_ppo_epoch (one persistent launch per epoch, a hand-written grid barrier) in its own
process, so that a fault or a watchdog exit of that kernel costs one test and not the GPU session of the whole suite.  Prints one JSON line per check; exit code 0 =
every check passed.

    python tests/epoch_worker.py golden      apx_ppo_epoch against the reference's own per-step outputs (tests/golden/g4b_epoch_h256.npz, inputs from seeds)
    python tests/epoch_worker.py twin        apx_ppo_epoch against the per-step apx_ppo_minibatch loop: 48 steps of minibatch 64 with mirror loss, reruns bit-identical
    python tests/epoch_worker.py ppo         PPO.update with epoch_kernel on / off on the same rollout of the HIP env (the wiring of apex_amd/ppo.py)
    python tests/epoch_worker.py td3_golden  apx_td3_updates (td3_small.hip, the same kind of kernel for TD3's update block) against the reference's TD3.train outputs (G20b)
    python tests/epoch_worker.py td3_twin    apx_td3_updates against the per-launch TD3Learner.train_step loop: batch 128 and batch 1024, random replay rows
"""
import json
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "tests")]

from golden_util import EPOCH_CASES, epoch_case_inputs, check_slim      # noqa: E402

FAILED = []


def report(name, ok, **kw):
    print(json.dumps(dict(check=name, ok=bool(ok), **{k: (float(v) if isinstance(v, (np.floating, float)) else v) for k, v in kw.items()})), flush=True)
    if not ok:
        FAILED.append(name)


def make_learner(dev, inp, mirror, adam_t0=1):
    from apex_amd import engine
    from tools.refprobe.common import MIRRORED_OBS_FULL_CLOCK, MIRRORED_ACTS
    lr = engine.PPOLearner(50, 10, 256, dev, fixed_std=np.exp(-1.5), mirrored_obs=MIRRORED_OBS_FULL_CLOCK, mirrored_acts=MIRRORED_ACTS)
    lr.actor.load_list(inp["actor"]); lr.critic.load_list(inp["critic"])
    old = engine.Mlp(50, 256, 10, dev); old.load_list(inp["old"])
    lr.obs_mean.copy_(torch.tensor(inp["obs_mean"])); lr.obs_std.copy_(torch.tensor(inp["obs_std"]))
    lr.t = adam_t0 - 1
    t = lambda x, **k: torch.tensor(x, device=dev, **k)
    obs, act, ret, adv = t(inp["obs"]), t(inp["act"]), t(inp["ret"]), t(inp["adv"])
    old_mu = old.forward(obs, lr.obs_mean, lr.obs_std)
    return lr, (obs, act, ret, adv, old_mu)


def run_steps(lr, data, perm, mb, mirror):
    scal = []
    for k in range(perm.numel() // mb):
        scal.append(lr.minibatch(*data, idx=perm[k * mb:(k + 1) * mb].contiguous(), mirror=mirror))
    return np.array(scal)


def run_epoch(lr, data, perm, mb, mirror):
    scal = lr.epoch(*data, perm, mb, mirror=mirror)
    torch.cuda.synchronize()
    return scal.cpu().numpy()


def golden(dev):
    g = np.load(os.path.join(REPO, "tests", "golden", "g4b_epoch_h256.npz"))
    for c, (mirror, mb, nb, adam_t0) in enumerate(EPOCH_CASES):
        inp = epoch_case_inputs(c)
        lr, data = make_learner(dev, inp, mirror, adam_t0)
        assert lr.epoch_supported(mb)
        scal = run_epoch(lr, data, torch.tensor(inp["perm"], device=dev), mb, mirror)
        ref = g[f"c{c}_scalars"]
        err = np.abs(scal - ref) / (1e-5 * np.abs(ref) + 2e-7)      # (the actor loss is a cancelling mean: the reference's own fp32 rounding is ~2e-7 absolute there)
        report(f"golden c{c} scalars", np.isfinite(scal).all() and err.max() <= 1.0, mirror=mirror, mb=mb, nb=nb, worst_over_tolerance=err.max())
        assert lr.t == adam_t0 - 1 + nb
        try:      # post-epoch parameters: at most 2e-3 of the sampled entries off by more than 2e-6 (test_ppo_update_golden_g4's criterion), none by more than 2.5e-4
            for i, w in enumerate(lr.actor.views()):
                check_slim(w.cpu().numpy(), g[f"c{c}_actor1.{i}"], atol=2.5e-4, frac_tol=2e-6, frac=2e-3, err_msg=(c, "actor", i))
            for i, w in enumerate(lr.critic.views()):
                check_slim(w.cpu().numpy(), g[f"c{c}_critic1.{i}"], atol=2.5e-4, frac_tol=2e-6, frac=2e-3, err_msg=(c, "critic", i))
            report(f"golden c{c} parameters", True)
        except AssertionError as e:
            report(f"golden c{c} parameters", False, detail=str(e))


def twin(dev):
    rs = np.random.RandomState(77)
    mirror, mb, nb = True, 64, 48
    inp = epoch_case_inputs(0)
    B = 4096
    obs = rs.randn(B, 50).astype(np.float32); ph = rs.rand(B) * 2 * np.pi
    obs[:, 46] = np.sin(ph); obs[:, 47] = np.cos(ph)
    inp.update(obs=obs, act=(rs.randn(B, 10) * 0.3).astype(np.float32), ret=rs.randn(B).astype(np.float32), adv=rs.randn(B).astype(np.float32))
    perm = torch.tensor(rs.permutation(B)[:nb * mb].astype(np.int64), device=dev)
    la, da = make_learner(dev, inp, mirror)
    sa = run_steps(la, da, perm, mb, mirror)
    out = []
    for rep in range(2):
        lb, db = make_learner(dev, inp, mirror)
        sb = run_epoch(lb, db, perm, mb, mirror)
        out.append((sb, lb.actor.params.clone(), lb.critic.params.clone(), lb.actor_m.clone(), lb.critic_v.clone()))
    sb = out[0][0]
    err = np.abs(sb - sa) / (1e-4 * np.abs(sa) + 1e-6)
    report("twin scalars", np.isfinite(sb).all() and err.max() <= 1.0, worst_over_tolerance=err.max())
    for name, x, y in (("actor", la.actor.params, out[0][1]), ("critic", la.critic.params, out[0][2])):
        d = (x - y).abs()
        report(f"twin {name} parameters", float(d.max()) <= 5e-4 and float((d > 5e-6).float().mean()) <= 5e-3, max=float(d.max()), frac_above_5e6=float((d > 5e-6).float().mean()))
    d = (la.actor_m - out[0][3]).abs().max() / la.actor_m.abs().max()
    report("twin Adam first moment", float(d) <= 1e-3, rel=float(d))
    same = all(bool(torch.equal(a, b)) for a, b in zip(out[0][1:], out[1][1:])) and np.array_equal(out[0][0], out[1][0])
    report("epoch reruns are bit-identical", same)


def ppo(dev):
    from apex_amd.vecenv import CassieVecEnv
    from apex_amd.ppo import PPO
    N, T, mtl = 256, 8, 10
    res = []
    for ek in (False, True):
        env = CassieVecEnv(n_envs=N, seed=2, max_traj_len=mtl)
        args = dict(gamma=0.99, lam=0.95, lr=1e-4, eps=1e-5, entropy_coeff=0.0, clip=0.2, minibatch_size=64, epochs=2, num_steps=T * N, max_traj_len=mtl,
                    max_grad_norm=0.05, mirror=True, std_dev=-1.5, seed=0, epoch_kernel=ek)
        algo = PPO(args, "/tmp/apx_test_unused", env, rank=0, world_size=1, group=None)
        algo.init_networks(0)
        algo.normalization_params(2000)
        algo.trace = []
        ret, _, _ = algo.sample()
        losses, kl, epochs_run = algo.update(ret)
        torch.cuda.synchronize()
        res.append((losses, kl, epochs_run, torch.stack(algo.trace).cpu().numpy(), algo.learner.actor.params.clone(), algo.learner.t))
        env.close()
    (l0, k0, e0, t0, p0, n0), (l1, k1, e1, t1, p1, n1) = res
    report("ppo same control flow", e0 == e1 and n0 == n1 and t0.shape == t1.shape, epochs=(e0, e1), steps=(n0, n1))
    if t0.shape == t1.shape:
        err = np.abs(t1 - t0) / (1e-4 * np.abs(t0) + 1e-6)
        report("ppo per-step scalars", err.max() <= 1.0, worst_over_tolerance=err.max())
        err = np.abs(l1 - l0) / (1e-4 * np.abs(l0) + 1e-6)
        report("ppo epoch means", err.max() <= 1.0 and abs(k1 - k0) <= 1e-4 * abs(k0) + 1e-7, worst_over_tolerance=err.max())
        d = (p0 - p1).abs()
        report("ppo actor parameters", float(d.max()) <= 5e-4 and float((d > 5e-6).float().mean()) <= 5e-3, max=float(d.max()))


def td3_golden(dev):
    from tests import test_gpu_learner as G
    try:
        G._run_g20(dev, os.path.join(REPO, "tests", "golden"), "g20b_td3_h256.npz", one_launch=True)
        report("td3 one-launch updates vs G20b", True)
    except AssertionError as e:
        report("td3 one-launch updates vs G20b", False, detail=str(e)[:2000])


def td3_twin(dev):
    from tests import test_gpu_learner as G
    for B, U in ((128, 6), (1024, 8)):
        try:
            G._td3_twin(dev, B=B, U=U, cap=5000)
            report("td3 one launch == per-launch loop, batch %d" % B, True)
        except AssertionError as e:
            report("td3 one launch == per-launch loop, batch %d" % B, False, detail=str(e)[:2000])


if __name__ == "__main__":
    assert torch.cuda.is_available(), "needs a GPU"
    dev = torch.device("cuda:0")
    {"golden": golden, "twin": twin, "ppo": ppo, "td3_golden": td3_golden, "td3_twin": td3_twin}[sys.argv[1]](dev)
    sys.exit(1 if FAILED else 0)
