"""GPU parity tests for the learner kernels: HIP path (through the C ABI) vs the numpy oracle and vs the golden
vectors produced by the reference itself.  Tolerances: bit-exact indices; <=1e-5 relative on returns, advantages
and losses (BASELINE.json north_star)."""
import os

import numpy as np
import pytest
import torch

from oracle import learner as L
from tests.test_oracle_learner import ACTOR_KEYS, CRITIC_KEYS

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def test_returns_scan_vs_oracle(dev):
    from apex_amd import engine
    rng = np.random.RandomState(0)
    for (T, N) in [(1, 64), (17, 5), (32, 4096), (400, 130), (7, 1)]:
        rew = rng.randn(T, N).astype(np.float32)
        end = (rng.rand(T, N) < 0.1).astype(np.uint8)
        boot = (rng.randn(T, N) * (rng.rand(T, N) < 0.5)).astype(np.float32)
        last = rng.randn(N).astype(np.float32)
        ref = L.returns_scan_grid_boot(rew, end, boot, last, 0.99)
        got = engine.returns_scan(torch.tensor(rew, device=dev), torch.tensor(end, device=dev),
                                  torch.tensor(boot, device=dev), torch.tensor(last, device=dev), 0.99).cpu().numpy()
        np.testing.assert_array_equal(got, ref.astype(np.float32))     # fp64 recurrence on both sides: bit-exact


def test_returns_scan_golden_g1(dev, golden_dir):
    """Reference finish_path outputs, replayed through the grid kernel one trajectory per column."""
    from apex_amd import engine
    g = np.load(os.path.join(golden_dir, "g1_finish_path.npz"))
    for c in range(int(g["n_cases"])):
        lens = g[f"c{c}_lens"]; T = int(lens.max()); N = len(lens)
        rew = np.zeros((T, N), np.float32); end = np.zeros((T, N), np.uint8); boot = np.zeros((T, N), np.float32)
        off = 0
        r64 = g[f"c{c}_rewards"]
        for n, Ln in enumerate(lens):
            rew[:Ln, n] = r64[off:off + Ln]; end[Ln - 1, n] = 1; boot[Ln - 1, n] = g[f"c{c}_last_vals"][n]; off += Ln
        got = engine.returns_scan(*(torch.tensor(x, device=dev) for x in (rew, end, boot)),
                                  torch.zeros(N, device=dev), float(g[f"c{c}_gamma"])).cpu().numpy()
        off = 0
        for n, Ln in enumerate(lens):
            # rewards were rounded to fp32 on the way in -> 1e-6 relative, well inside the 1e-5 bar
            np.testing.assert_allclose(got[:Ln, n], g[f"c{c}_returns"][off:off + Ln], rtol=1e-5, atol=1e-6); off += Ln


def test_returns_scan_golden_g15a(dev, golden_dir):
    """Returns of the reference's own PPO.sample run (toy env) through the HIP scan: <= 1e-5 relative."""
    from apex_amd import engine
    from tests.test_oracle_learner import _g15a_grid
    g = np.load(os.path.join(golden_dir, "g15a_ppo_sample.npz"))
    rew, end, boot = _g15a_grid(g)
    got = engine.returns_scan(torch.tensor(rew, dtype=torch.float32, device=dev), torch.tensor(end, device=dev),
                              torch.tensor(boot, dtype=torch.float32, device=dev), torch.zeros(1, device=dev), float(g["gamma"]))
    np.testing.assert_allclose(got.cpu().numpy()[:, 0], g["returns"], rtol=1e-5, atol=1e-6)


def test_adv_norm_golden_g2(dev, golden_dir):
    from apex_amd import engine
    g = np.load(os.path.join(golden_dir, "g2_adv_norm.npz"))
    for c in range(int(g["n_cases"])):
        adv = engine.normalize_advantages(torch.tensor(g[f"c{c}_returns"], device=dev),
                                          torch.tensor(g[f"c{c}_values"], device=dev), 1e-5).cpu().numpy()
        np.testing.assert_allclose(adv, g[f"c{c}_adv"].reshape(-1), rtol=1e-5, atol=2e-6)


def test_mlp_forward_golden_g3(dev, golden_dir):
    from apex_amd import engine
    g = np.load(os.path.join(golden_dir, "g3_policy_forward.npz"))
    actor = engine.Mlp(50, 256, 10, dev); actor.load_list([g["actor." + k] for k in ACTOR_KEYS])
    critic = engine.Mlp(50, 256, 1, dev); critic.load_list([g["critic." + k] for k in CRITIC_KEYS])
    obs = torch.tensor(g["obs"], device=dev)
    om, os_ = torch.tensor(g["obs_mean"], device=dev), torch.tensor(g["obs_std"], device=dev)
    np.testing.assert_allclose(actor.forward(obs, om, os_).cpu().numpy(), g["mean"], rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(critic.forward(obs).cpu().numpy(), g["value_train"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(critic.forward(obs, om, os_).cpu().numpy(), g["value_eval"], rtol=1e-5, atol=1e-5)


def test_mlp_forward_ragged_shapes(dev):
    """Empty, single-row and non-multiple-of-tile batches; gather + mirror path vs oracle."""
    from apex_amd import engine
    from tools.refprobe.common import MIRRORED_OBS_FULL_CLOCK
    rng = np.random.RandomState(3)
    net = engine.Mlp(50, 64, 10, dev)
    W = [rng.randn(*v.shape).astype(np.float32) * 0.2 for v in net.views()]
    net.load_list(W)
    sp = torch.as_tensor(engine.signed_perm_from_mirror(MIRRORED_OBS_FULL_CLOCK), device=dev)
    Mo = L.mirror_matrix(MIRRORED_OBS_FULL_CLOCK)
    for B in [0, 1, 63, 65, 1000]:
        x = rng.randn(max(B, 1) + 5, 50).astype(np.float32)
        x[:, 46:48] = np.clip(x[:, 46:48], -1, 1)
        idx = rng.randint(0, x.shape[0], size=B)
        y = net.forward(torch.tensor(x, device=dev), idx=torch.tensor(idx, device=dev, dtype=torch.int64),
                        sign_perm=sp, clock_mask=(1 << 46) | (1 << 47))
        assert y.shape == (B, 10)
        if B:
            ref = L.mlp_forward(W, L.mirror_clock_observation(x[idx].astype(np.float64), Mo, [46, 47]))
            np.testing.assert_allclose(y.cpu().numpy(), ref, rtol=1e-4, atol=1e-5)


def _run_g4_case(dev, g, g5, c):
    from apex_amd import engine
    from tools.refprobe.common import MIRRORED_OBS_FULL_CLOCK, MIRRORED_ACTS
    p = f"c{c}_"
    H = int(g[p + "hidden"]); mirror = bool(g[p + "mirror"])
    lr = engine.PPOLearner(50, 10, H, dev, fixed_std=np.exp(-1.5), entropy_coeff=float(g[p + "entropy_coeff"]),
                           mirrored_obs=MIRRORED_OBS_FULL_CLOCK, mirrored_acts=MIRRORED_ACTS)
    lr.actor.load_list([g[p + "actor0." + k] for k in ACTOR_KEYS])
    lr.critic.load_list([g[p + "critic0." + k] for k in CRITIC_KEYS])
    old = engine.Mlp(50, H, 10, dev); old.load_list([g[p + "old." + k] for k in ACTOR_KEYS])
    lr.obs_mean.copy_(torch.tensor(g[p + "obs_mean"])); lr.obs_std.copy_(torch.tensor(g[p + "obs_std"]))
    scal = []
    for s in range(int(g[p + "nsteps"])):
        obs, act, ret, adv = (torch.tensor(g[p + f"s{s}_{k}"], device=dev) for k in ("obs", "act", "ret", "adv"))
        old_mu = old.forward(obs, lr.obs_mean, lr.obs_std)
        scal.append(lr.minibatch(obs, act, ret.view(-1), adv.view(-1), old_mu, mirror=mirror))
    return lr, np.array(scal)


def test_ppo_update_golden_g4(dev, golden_dir):
    """update_policy 6-tuple (<=1e-5 relative) and post-step parameters vs the reference's own outputs."""
    g = np.load(os.path.join(golden_dir, "g4_update_policy.npz"))
    g5 = np.load(os.path.join(golden_dir, "g5_mirror.npz"))
    for c in range(int(g["n_cases"])):
        lr, scal = _run_g4_case(dev, g, g5, c)
        np.testing.assert_allclose(scal, g[f"c{c}_scalars"], rtol=1e-5, atol=1e-7)
        for k, w in zip(ACTOR_KEYS, lr.actor.views()):
            bad = np.abs(w.cpu().numpy() - g[f"c{c}_actor1." + k]) > 2e-6
            assert bad.mean() < 2e-3, (c, k, bad.mean())
        for k, w in zip(CRITIC_KEYS, lr.critic.views()):
            bad = np.abs(w.cpu().numpy() - g[f"c{c}_critic1." + k]) > 2e-6
            assert bad.mean() < 2e-3, (c, k, bad.mean())


def test_ppo_steps_golden_g4b(dev, golden_dir):
    """An epoch of small-minibatch steps (64 / 32 / 128 / 16 rows, the 2 x 256 networks, Adam step counts 1.. / 7.. / 2049..) through the per-step launches
    against the reference's own per-step 6-tuples and post-epoch parameters (G4b; inputs regenerated from the fixture's seeds)."""
    from golden_util import EPOCH_CASES, epoch_case_inputs, check_slim
    from tests import epoch_worker as W
    g = np.load(os.path.join(golden_dir, "g4b_epoch_h256.npz"))
    for c, (mirror, mb, nb, adam_t0) in enumerate(EPOCH_CASES):
        inp = epoch_case_inputs(c)
        lr, data = W.make_learner(dev, inp, mirror, adam_t0)
        scal = W.run_steps(lr, data, torch.tensor(inp["perm"], device=dev), mb, mirror)
        np.testing.assert_allclose(scal, g[f"c{c}_scalars"], rtol=1e-5, atol=2e-7)      # (the actor loss is a cancelling mean: the reference's own fp32 rounding is ~2e-7 absolute there)
        for i, w in enumerate(lr.actor.views()):
            check_slim(w.cpu().numpy(), g[f"c{c}_actor1.{i}"], atol=2.5e-4, frac_tol=2e-6, frac=2e-3, err_msg=(c, "actor", i))
        for i, w in enumerate(lr.critic.views()):
            check_slim(w.cpu().numpy(), g[f"c{c}_critic1.{i}"], atol=2.5e-4, frac_tol=2e-6, frac=2e-3, err_msg=(c, "critic", i))


def test_ppo_update_large_minibatch_vs_oracle(dev):
    """Throughput-sized minibatch (16 384) with index gather: gradients via grad_only vs the fp64 oracle."""
    from apex_amd import engine
    from tools.refprobe.common import MIRRORED_OBS_FULL_CLOCK, MIRRORED_ACTS
    rng = np.random.RandomState(7)
    H, Btot, mb = 256, 20000, 16384          # the fused 2 x 256 forward + split-K backward of BASELINE configs[1]
    lr = engine.PPOLearner(50, 10, H, dev, fixed_std=np.exp(-1.5), mirrored_obs=MIRRORED_OBS_FULL_CLOCK,
                           mirrored_acts=MIRRORED_ACTS)
    sc = [0.1, 0.1, 0.05, 0.1, 0.02, 0.1]            # layer scales that keep |mu| ~ 0.1 and the probability ratio within [0.4, 2] at H = 256
    Wa = [rng.randn(*v.shape).astype(np.float32) * c for v, c in zip(lr.actor.views(), sc)]
    Wc = [rng.randn(*v.shape).astype(np.float32) * c for v, c in zip(lr.critic.views(), sc)]
    Wo = [w + rng.randn(*w.shape).astype(np.float32) * 0.002 for w in Wa]
    lr.actor.load_list(Wa); lr.critic.load_list(Wc)
    old = engine.Mlp(50, H, 10, dev); old.load_list(Wo)
    obs = rng.randn(Btot, 50).astype(np.float32); ph = rng.rand(Btot) * 2 * np.pi
    obs[:, 46] = np.sin(ph); obs[:, 47] = np.cos(ph)
    act = (rng.randn(Btot, 10) * 0.3).astype(np.float32); ret = rng.randn(Btot).astype(np.float32)
    adv = rng.randn(Btot).astype(np.float32); idx = rng.permutation(Btot)[:mb]
    t = lambda x, **k: torch.tensor(x, device=dev, **k)
    old_mu = old.forward(t(obs), lr.obs_mean, lr.obs_std)
    scal = lr.minibatch(t(obs), t(act), t(ret), t(adv), old_mu, idx=t(idx, dtype=torch.int64), grad_only=True)
    Mo, Ma = L.mirror_matrix(MIRRORED_OBS_FULL_CLOCK), L.mirror_matrix(MIRRORED_ACTS)

    class Rec:           # capture the (clipped) gradients the oracle hands to Adam; use a huge clip so they are raw
        def step(self, params, grads):
            self.g = grads
            return params
    ra, rc = Rec(), Rec()
    ref, _, _ = L.ppo_update(Wa, Wo, Wc, ra, rc, obs[idx], act[idx], ret[idx, None], adv[idx, None], np.zeros(50),
                             np.ones(50), np.exp(-1.5), grad_clip=1e9, M_obs=Mo, M_act=Ma)
    np.testing.assert_allclose(scal, ref, rtol=1e-5, atol=1e-7)
    ga = np.concatenate([x.reshape(-1) for x in ra.g]); gc = np.concatenate([x.reshape(-1) for x in rc.g])
    np.testing.assert_allclose(lr.actor_g.cpu().numpy(), ga, rtol=1e-3, atol=1e-6 * np.abs(ga).max() + 1e-9)
    np.testing.assert_allclose(lr.critic_g.cpu().numpy(), gc, rtol=1e-3, atol=5e-5 * np.abs(gc).max() + 1e-9)      # fp32 sums over 16 384 rows


def _gparams(g, prefix, keys, which):
    """parameter list of a golden: stored tensors (small fixtures) or regenerated from the stored seed (BASELINE-size fixtures)"""
    from golden_util import seeded_params
    if which + "_seed" in g.files:
        return seeded_params([[d for d in row if d] for row in g[which + "_shapes"]], int(g[which + "_seed"]))
    return [g[prefix + "." + str(k)] for k in keys]


@pytest.mark.parametrize("fname", ["g18_lstm.npz", "g18b_lstm_h128.npz"])
def test_lstm_forward_backward_golden_g18(dev, golden_dir, fname):
    """G18 (next row f1): the reference's Gaussian_LSTM_Actor / LSTM_V (2 LSTM cells + linear head; 64 units, and the 2 x 128 of BASELINE
    configs[3] in g18b) on a padded batch [9, 5, 50] from zero state: outputs, BPTT parameter gradients of sum(w * y), and the
    step-by-step rollout with the carried state."""
    import os
    from apex_amd import engine
    from golden_util import check_slim
    g = np.load(os.path.join(golden_dir, fname))
    big = "actor_seed" in g.files
    x = torch.tensor(g["x"], device=dev)
    xn = (x - torch.tensor(g["obs_mean"], device=dev)) / torch.tensor(g["obs_std"], device=dev)
    for name, keys, inp, yref, w, O in (("actor", g["actor_keys"], xn, g["mu"], g["wa"], 10), ("critic", g["critic_keys"], x, g["v"], g["wc"], 1)):
        net = engine.Lstm(50, int(g["hidden"]), 2, O, dev)
        net.load_list(_gparams(g, name, keys, name))
        y, x3, save = net.forward(inp, keep=True)
        np.testing.assert_allclose(y.cpu().numpy(), yref, rtol=2e-5, atol=2e-6)
        grads = torch.zeros_like(net.params)
        net.backward(grads, x3, save, torch.tensor(w, device=dev))
        for k, gv in zip(keys, net.views(grads)):
            ref = g[name + "_grad." + str(k)]
            if big:
                check_slim(gv.cpu().numpy(), ref, atol=2e-4 * max(1.0, np.abs(ref[2:]).max()), err_msg="%s %s" % (name, k))
            else:
                np.testing.assert_allclose(gv.cpu().numpy(), ref, rtol=2e-4, atol=2e-5 * max(1.0, np.abs(ref).max()), err_msg="%s %s" % (name, k))
        if name == "actor":      # rollout: one env, step by step, carried (h, c)
            hc = torch.zeros(2, 2, 1, int(g["hidden"]), device=dev)
            steps = torch.stack([net.forward(inp[t, 2:3].contiguous(), hc=hc) for t in range(inp.shape[0])])
            np.testing.assert_allclose(steps[:, 0].cpu().numpy(), g["mu_step_env2"], rtol=2e-5, atol=2e-6)
            np.testing.assert_allclose(steps[:, 0].cpu().numpy(), y[:, 2].cpu().numpy(), rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("fname", ["g19_lstm_update.npz", "g19b_lstm_update_h128.npz"])
def test_recurrent_update_policy_golden_g19(dev, golden_dir, fname):
    """G19 (next row f1): the reference's PPO.update_policy in recurrent mode on padded batches (4 trajectories of different lengths, mirror
    loss on), two consecutive steps: six scalars per step (masked actor / critic terms, unmasked ratio / KL / mirror means over the padded
    tensor) and the parameters after each step."""
    import os
    from apex_amd import engine
    from apex_amd.vecenv import MIRRORED_OBS, MIRRORED_ACTS, CLOCK_INDS
    from golden_util import check_slim, seeded_noise
    g = np.load(os.path.join(golden_dir, fname))
    big = "actor_seed" in g.files
    H = int(g["hidden"])
    L = engine.RecurrentPPOLearner(50, 10, H, 2, dev, float(g["fixed_std"]), mirrored_obs=MIRRORED_OBS, mirrored_acts=MIRRORED_ACTS, clock_inds=CLOCK_INDS)
    if big:          # old = seeded parameters, actor0 = old + seeded noise (tools/refprobe/gen_golden_lstm_update.py)
        old = _gparams(g, "old", g["actor_keys"], "actor")
        nz = seeded_noise([w.shape for w in old], int(g["pert_seed"]), float(g["pert_scale"]))
        L.old_actor.load_list(old); L.actor.load_list([w + n for w, n in zip(old, nz)])
        L.critic.load_list(_gparams(g, "critic0", g["critic_keys"], "critic"))
    else:
        L.actor.load_list([g["actor0." + str(k)] for k in g["actor_keys"]])
        L.old_actor.load_list([g["old." + str(k)] for k in g["actor_keys"]])
        L.critic.load_list([g["critic0." + str(k)] for k in g["critic_keys"]])
    L.obs_mean.copy_(torch.tensor(g["obs_mean"])); L.obs_std.copy_(torch.tensor(g["obs_std"]))
    t = lambda a: torch.tensor(a, device=dev)
    for s in range(2):
        p = "s%d_" % s
        scal = L.minibatch(t(g[p + "obs"]), t(g[p + "act"]), t(g[p + "ret"]), t(g[p + "adv"]), t(g[p + "mask"])).cpu().numpy()
        np.testing.assert_allclose(scal, g["scalars"][s], rtol=2e-4, atol=2e-6, err_msg="step %d" % s)
        for nm, net, keys in (("actor", L.actor, g["actor_keys"]), ("critic", L.critic, g["critic_keys"])):
            for k, v in zip(keys, net.views()):
                if big:
                    check_slim(v.cpu().numpy(), g[p + nm + "." + str(k)], atol=4.1e-4, frac_tol=3e-6, frac=1e-2, err_msg="%d %s %s" % (s, nm, k))
                    continue
                d = np.abs(v.cpu().numpy() - g[p + nm + "." + str(k)])
                assert (d > 3e-6).mean() < 5e-3 and d.max() < 4.1e-4, (s, nm, str(k), (d > 3e-6).mean(), d.max())


def _run_g20(dev, golden_dir, fname, one_launch=False, prepare=None):
    """G20 (next row f2): the reference's TD3.train for 4 iterations on recorded batches (target smoothing with the recorded noise, clipped
    double-Q target, two delayed policy updates, Polyak averaging): returned statistics and all four parameter sets afterwards."""
    import os
    from apex_amd import engine
    from golden_util import check_slim, seeded_params, seeded_noise
    g = np.load(os.path.join(golden_dir, fname))
    big = "seeds" in g.files
    H = int(g["hidden"])
    L = engine.TD3Learner(50, 10, H, dev, max_action=1.0, a_lr=float(g["lr"]), c_lr=float(g["lr"]))
    ak, ck = [str(k) for k in g["actor_keys"]], [str(k) for k in g["critic_keys"]]
    if big:          # 256-unit nets of BASELINE configs[4]: live nets from seeds, targets = live + seeded noise
        shp = lambda a: [[d for d in row if d] for row in a]
        sa, sc, sat, sct = (int(x) for x in g["seeds"])
        A = seeded_params(shp(g["actor_shapes"]), sa); Cq = seeded_params(shp(g["critic_shapes"]), sc)
        At = [w + n for w, n in zip(A, seeded_noise([w.shape for w in A], sat, float(g["target_noise"])))]
        Ct = [w + n for w, n in zip(Cq, seeded_noise([w.shape for w in Cq], sct, float(g["target_noise"])))]
        L.actor.load_list(A); L.actor_t.load_list(At)
        for i in range(2):
            L.q[i].load_list(Cq[6 * i:6 * i + 6]); L.q_t[i].load_list(Ct[6 * i:6 * i + 6])
    else:
        L.actor.load_list([g["actor0." + k] for k in ak]); L.actor_t.load_list([g["actor_target0." + k] for k in ak])
        for i in range(2):
            L.q[i].load_list([g["critic0." + k] for k in ck[6 * i:6 * i + 6]]); L.q_t[i].load_list([g["critic_target0." + k] for k in ck[6 * i:6 * i + 6]])
    t = lambda a: torch.tensor(a, device=dev)
    q_loss = pi_loss = avg_q1 = 0.0
    if prepare is not None:
        prepare(L)
    if one_launch:      # the same iterations as ONE launch (apx_td3_updates): the batches laid end to end as the replay, update u on rows 64 u .. 64 u + 63
        n_it = int(g["iters"])
        cat = lambda k: t(np.concatenate([g["b%d_%s" % (it, k)] for it in range(n_it)]))
        nb = g["b0_x"].shape[0]
        ind = torch.arange(n_it * nb, device=dev, dtype=torch.int64).view(n_it, nb).contiguous()
        noise = t(np.stack([g["b%d_noise" % it] for it in range(n_it)])).contiguous()
        st = L.updates(cat("x"), cat("y"), cat("u"), cat("r").view(-1).contiguous(), (1.0 - cat("d").view(-1)).contiguous(), ind, noise, 0, discount=float(g["discount"]),
                       tau=float(g["tau"]), noise_clip=float(g["noise_clip"]), policy_freq=int(g["policy_freq"])).cpu().numpy()
        assert np.isfinite(st).all()
        q_loss, avg_q1, pi_loss = st[:, 0].sum(), st[:, 1].sum() / 64, st[:, 3].sum()
    for it in range(0 if one_launch else int(g["iters"])):
        p = "b%d_" % it
        stats, pl = L.train_step(t(g[p + "x"]), t(g[p + "u"]), t(g[p + "y"]), t(g[p + "r"]).view(-1), 1.0 - t(g[p + "d"]).view(-1), t(g[p + "noise"]), it,
                                 discount=float(g["discount"]), tau=float(g["tau"]), noise_clip=float(g["noise_clip"]), policy_freq=int(g["policy_freq"]))
        s = stats.cpu().numpy()
        q_loss += s[0]; avg_q1 += s[1] / 64
        if pl is not None:
            pi_loss += float(pl)
    n = int(g["iters"])
    np.testing.assert_allclose([avg_q1 / n, q_loss / n, pi_loss / n], [float(g["ret_avg_q1"]), float(g["ret_q_loss"]), float(g["ret_pi_loss"])], rtol=2e-4, atol=2e-6)
    for nm, nets, keys in (("actor1", [L.actor], ak), ("actor_target1", [L.actor_t], ak), ("critic1", L.q, ck), ("critic_target1", L.q_t, ck)):
        views = [v for net in nets for v in net.views()]
        for k, v in zip(keys, views):
            lim = 2.1e-3 if "target" not in nm else 2e-5           # Adam at lr 1e-3: a sign tie at g ~ 0 moves a weight by 2e-3
            if big:
                check_slim(v.cpu().numpy(), g[nm + "." + k], atol=lim + 1e-9, frac_tol=2e-5, frac=1e-2, err_msg=nm + "." + k)
                continue
            d = np.abs(v.cpu().numpy() - g[nm + "." + k])
            assert (d > 2e-5).mean() < (5e-3 if "target" not in nm else 1e-9 + 5e-3) and d.max() < lim + 1e-9, (nm, k, (d > 2e-5).mean(), d.max())


def _td3_twin(dev, B=128, U=6, cap=1000, prepare=None):
    """apx_td3_updates (one launch) against the per-launch TD3Learner.train_step loop on the same replay rows and noise: statistics of every update and all four
    parameter sets + Adam moments afterwards.  Returns the worst deviations."""
    from apex_amd import engine
    from golden_util import seeded_params, seeded_noise
    rs = np.random.RandomState(5)
    ash = [(256, 50), (256,), (256, 256), (256,), (10, 256), (10,)]; csh = [(256, 60), (256,), (256, 256), (256,), (1, 256), (1,)] * 2
    A0 = seeded_params(ash, 11); C0 = seeded_params(csh, 12)
    At = [w + n for w, n in zip(A0, seeded_noise(ash, 13, 0.01))]; Ct = [w + n for w, n in zip(C0, seeded_noise(csh, 14, 0.01))]
    t = lambda a, **k: torch.tensor(a, device=dev, **k)
    rep = [t(rs.randn(cap, 50).astype(np.float32)), t(rs.randn(cap, 50).astype(np.float32)), t(rs.uniform(-1, 1, (cap, 10)).astype(np.float32)),
           t(rs.rand(cap).astype(np.float32)), t((rs.rand(cap) > 0.1).astype(np.float32))]
    ind = t(rs.randint(0, cap, (U, B)).astype(np.int64)); noise = t((rs.randn(U, B, 10) * 0.2).astype(np.float32))
    out = []
    for one in (False, True):
        L = engine.TD3Learner(50, 10, 256, dev, max_action=1.0, a_lr=1e-3, c_lr=1e-3)
        L.actor.load_list(A0); L.actor_t.load_list(At)
        for i in range(2):
            L.q[i].load_list(C0[6 * i:6 * i + 6]); L.q_t[i].load_list(Ct[6 * i:6 * i + 6])
        if one:
            if prepare is not None:
                prepare(L, B, U)
            st = L.updates(rep[0], rep[1], rep[2], rep[3], rep[4], ind, noise, 3, policy_freq=2).cpu().numpy()
        else:
            st = np.zeros((U, 4))
            for u in range(U):
                g_ = lambda x: x.index_select(0, ind[u])
                s3, pl = L.train_step(g_(rep[0]), g_(rep[2]), g_(rep[1]), g_(rep[3]), g_(rep[4]), noise[u], 3 + u, policy_freq=2)
                st[u, :3] = s3.cpu().numpy(); st[u, 3] = 0.0 if pl is None else float(pl)
        out.append((st, [x.clone() for x in (L.actor.params, L.actor_t.params, L.critic_flat, L.critic_t_flat, L.a_m, L.c_m, L.a_v, L.c_v)], (L.t_a, L.t_c)))
    (s0, p0, n0), (s1, p1, n1) = out
    assert n0 == n1, (n0, n1)
    assert np.isfinite(s1).all()
    np.testing.assert_allclose(s1, s0, rtol=2e-4, atol=2e-6)
    assert (s0[:, 3] != 0).sum() == sum(1 for u in range(U) if (3 + u) % 2 == 0)
    for nm, a, b in zip(("actor", "actor_target", "critic", "critic_target", "actor m", "critic m", "actor v", "critic v"), p0, p1):
        d = (a - b).abs()
        lim = 2.1e-3 if nm in ("actor", "critic") else (2.2e-5 if "target" in nm else float(a.abs().max()) * 2e-3 + 1e-9)      # Adam at lr 1e-3: a sign tie at g ~ 0 moves a weight by 2e-3
        assert float(d.max()) <= lim and float((d > lim / 50).float().mean()) < 1e-2, (nm, float(d.max()), float((d > lim / 50).float().mean()))


@pytest.mark.parametrize("fname", ["g20_td3.npz", "g20b_td3_h256.npz"])
def test_td3_train_golden_g20(dev, golden_dir, fname):
    """TD3.train iterations (sync_td3.py:133-209) through the per-launch path against the reference's own outputs"""
    _run_g20(dev, golden_dir, fname)


def test_full_size_properties_of_the_learner_kernels(dev):
    """BASELINE sizes (T = 32, N = 4096 -> 131 072 samples, minibatch 16 384) through size-independent properties: the return scan is linear
    in (rewards, bootstrap values) and splits at episode ends; normalised advantages have mean 0 / unbiased std 1 and are invariant to an
    affine change of the returns; the mirror loss of a mirror-symmetric policy is zero and grows when the symmetry is broken; a PPO step
    with old == new policy has ratio 1 and KL 0 on the whole minibatch."""
    from apex_amd import engine
    from apex_amd.vecenv import MIRRORED_OBS, MIRRORED_ACTS, CLOCK_INDS
    T, N = 32, 4096
    g = torch.Generator(device=dev); g.manual_seed(0)
    rnd = lambda *s: torch.randn(*s, device=dev, generator=g)
    r1, r2, b1, b2 = rnd(T, N), rnd(T, N), rnd(T, N), rnd(T, N)
    end = (torch.rand(T, N, device=dev, generator=g) < 0.05).to(torch.uint8)
    l1, l2 = rnd(N), rnd(N)
    R = lambda r, b, l: engine.returns_scan(r, end, b, l, 0.99)
    np.testing.assert_allclose(R(2 * r1 - 3 * r2, 2 * b1 - 3 * b2, 2 * l1 - 3 * l2).cpu().numpy(), (2 * R(r1, b1, l1) - 3 * R(r2, b2, l2)).cpu().numpy(), rtol=1e-4, atol=1e-4)
    ret = R(r1, b1, l1)
    e = end.bool()
    np.testing.assert_allclose(ret[e].cpu().numpy(), (r1 + 0.99 * b1)[e].cpu().numpy(), rtol=1e-6, atol=1e-6)        # an episode end sees only its own reward + bootstrap
    val = rnd(T, N)
    adv = engine.normalize_advantages(ret, val)
    assert abs(float(adv.mean())) < 1e-4 and abs(float(adv.std()) - 1) < 1e-3
    adv2 = engine.normalize_advantages(5 * ret + 7, 5 * val)
    np.testing.assert_allclose(adv2.cpu().numpy(), adv.cpu().numpy(), rtol=1e-3, atol=2e-4)
    # learner on a 16 384-row minibatch
    L = engine.PPOLearner(50, 10, 256, dev, float(np.exp(-1.5)), mirrored_obs=MIRRORED_OBS, mirrored_acts=MIRRORED_ACTS, clock_inds=CLOCK_INDS)
    L.actor.params.copy_(rnd(L.actor.n) * 0.05); L.critic.params.copy_(rnd(L.critic.n) * 0.05)
    B = 16384
    obs = rnd(B, 50) * 0.5; ph = torch.rand(B, device=dev, generator=g) * 6.28; obs[:, 46] = torch.sin(ph); obs[:, 47] = torch.cos(ph)
    act = rnd(B, 10) * 0.3; retb = rnd(B); advb = rnd(B)
    mu = L.old_means(obs)
    scal = L.minibatch(obs, act, retb, advb, mu, grad_only=True)
    assert abs(scal[3] - 1) < 1e-6 and abs(scal[4]) < 1e-9 and scal[5] > 0          # ratio 1, KL 0, a random net is not mirror-symmetric
    np.testing.assert_allclose(scal[0], -float(advb.mean()), rtol=1e-4, atol=1e-6)   # ratio 1 everywhere: actor loss = -mean(adv)
    # a mirror-symmetric policy: pi(s) := (f(s) + M_a f(M_s s)) / 2 is symmetric; the plain net is not, but its mirror loss must equal
    # 0.4 * mean((f(s) - M_a f(M_s s))^2) computed from two plain forwards
    sp = engine.signed_perm_from_mirror(MIRRORED_OBS); ap = engine.signed_perm_from_mirror(MIRRORED_ACTS)
    src = torch.as_tensor(np.where(sp >= 0, sp, -sp - 1), device=dev); sgn = torch.as_tensor(np.where(sp >= 0, 1.0, -1.0), dtype=torch.float32, device=dev)
    mobs = obs.index_select(1, src) * sgn
    for c in CLOCK_INDS:
        mobs[:, c] = torch.sin(torch.asin(mobs[:, c].clamp(-1, 1)) + np.pi)
    f_m = L.actor.forward(mobs.contiguous(), L.obs_mean, L.obs_std)
    asrc = torch.as_tensor(np.where(ap >= 0, ap, -ap - 1), device=dev); asgn = torch.as_tensor(np.where(ap >= 0, 1.0, -1.0), dtype=torch.float32, device=dev)
    mir = f_m.index_select(1, asrc) * asgn
    np.testing.assert_allclose(scal[5], 0.4 * float(((mu - mir) ** 2).mean()), rtol=2e-4)


def test_mirror_loss_uses_the_env_clock_columns_min_profile(dev):
    """with input_profile=min the clock pair sits in columns 21, 22 of the 25-entry observation (cassie.py:829-837), and the reference takes the
    columns from env.clock_inds (rl/algos/ppo.py:307-310).  The mirror term of the fused minibatch must equal 0.4 * mean((f(s) - M_a f(M_s s))^2) with the
    min-profile mirror list and THOSE columns phase-shifted; columns beyond the observation are refused."""
    from apex_amd import engine
    from apex_amd.vecenv import mirrored_obs_for, MIRRORED_ACTS, OBS_DIM_MIN
    D = OBS_DIM_MIN + 4
    clock = [D - 4, D - 3]
    mo = mirrored_obs_for("clock", "min")
    with pytest.raises(ValueError):
        engine.PPOLearner(D, 10, 256, dev, float(np.exp(-1.5)), mirrored_obs=mo, mirrored_acts=MIRRORED_ACTS, clock_inds=[46, 47])
    g = torch.Generator(device=dev); g.manual_seed(1)
    rnd = lambda *s: torch.randn(*s, device=dev, generator=g)
    L = engine.PPOLearner(D, 10, 256, dev, float(np.exp(-1.5)), mirrored_obs=mo, mirrored_acts=MIRRORED_ACTS, clock_inds=clock)
    L.actor.params.copy_(rnd(L.actor.n) * 0.05); L.critic.params.copy_(rnd(L.critic.n) * 0.05)
    B = 4096
    obs = rnd(B, D) * 0.5; ph = torch.rand(B, device=dev, generator=g) * 6.28; obs[:, clock[0]] = torch.sin(ph); obs[:, clock[1]] = torch.cos(ph)
    act = rnd(B, 10) * 0.3; retb = rnd(B); advb = rnd(B)
    mu = L.old_means(obs)
    scal = L.minibatch(obs, act, retb, advb, mu, grad_only=True)
    sp = engine.signed_perm_from_mirror(mo); ap = engine.signed_perm_from_mirror(MIRRORED_ACTS)
    src = torch.as_tensor(np.where(sp >= 0, sp, -sp - 1), device=dev); sgn = torch.as_tensor(np.where(sp >= 0, 1.0, -1.0), dtype=torch.float32, device=dev)
    mobs = obs.index_select(1, src) * sgn
    for c in clock:
        mobs[:, c] = torch.sin(torch.asin(mobs[:, c].clamp(-1, 1)) + np.pi)
    f_m = L.actor.forward(mobs.contiguous(), L.obs_mean, L.obs_std)
    asrc = torch.as_tensor(np.where(ap >= 0, ap, -ap - 1), device=dev); asgn = torch.as_tensor(np.where(ap >= 0, 1.0, -1.0), dtype=torch.float32, device=dev)
    mir = f_m.index_select(1, asrc) * asgn
    want = 0.4 * float(((mu - mir) ** 2).mean())
    np.testing.assert_allclose(scal[5], want, rtol=2e-4)
    # and it is NOT what the full-profile columns would give (no column phase-shifted at all): the test can tell the difference
    mobs0 = obs.index_select(1, src) * sgn
    mir0 = L.actor.forward(mobs0.contiguous(), L.obs_mean, L.obs_std).index_select(1, asrc) * asgn
    assert abs(0.4 * float(((mu - mir0) ** 2).mean()) - want) > 1e-3 * want


class _ToyTd3Env:
    """The toy dynamics of tools/refprobe/gen_golden_td3_loop.py (G20c) as an N-column device env with the CassieVecEnv surface the TD3 driver uses:
    reset() starts a new scripted episode in every column (episode counter in column order, like the reference's workers run one after the other),
    step(act) -> (next_obs, reward, done, final_obs); a finished column keeps stepping on harmlessly (the driver ignores it)."""
    def __init__(self, dev, n, lens):
        self.device, self.n_envs, self.obs_dim, self.lens, self.k = dev, n, 50, [int(x) for x in lens], 0

    def _obs(self):
        o = self.x.clone()
        o[:, 46] = torch.sin(0.2 * self.t); o[:, 47] = torch.cos(0.2 * self.t)
        return o.float()

    def reset(self):
        ks = torch.arange(self.k + 1, self.k + 1 + self.n_envs, dtype=torch.float64, device=self.device); self.k += self.n_envs
        self.L = torch.tensor([self.lens[(int(k) - 1) % len(self.lens)] for k in ks.tolist()], device=self.device)
        self.t = torch.zeros(self.n_envs, dtype=torch.float64, device=self.device)
        self.x = torch.cos(torch.arange(50, dtype=torch.float64, device=self.device).view(1, 50) * 0.1 * ks.view(-1, 1))
        return self._obs()

    def step(self, act):
        self.t = self.t + 1
        self.x = 0.9 * self.x + 0.1 * act.double().repeat(1, 5) + 0.01
        rew = torch.exp(-self.x.abs().mean(1)).float()
        obs = self._obs()
        done = (self.t >= self.L).to(torch.uint8)
        return obs, rew, done, obs.clone()


def test_td3_whole_loop_golden_g20c(dev, golden_dir):
    """G20c: three rounds of the reference's synchronous TD3 loop body (parallel_collect_experience -> add_parallel -> train, sync_td3.py:304-313)
    on the toy env, replayed through apex_amd.td3.TD3.reference_round with the captured exploration / sampling / smoothing streams: the
    transitions of every round (collection with the live actor: the rounds depend on each other through the updates), the returned statistics,
    the live nets after every round and the target nets at the end."""
    import os
    from apex_amd.td3 import TD3
    g = np.load(os.path.join(golden_dir, "g20c_td3_loop.npz"))
    P, B, H, mtl = int(g["procs"]), int(g["batch"]), int(g["hidden"]), int(g["max_traj_len"])
    env = _ToyTd3Env(dev, P, g["lens"])
    algo = TD3(env, "/tmp/g20c", hidden=H, a_lr=float(g["lr"]), c_lr=float(g["lr"]), discount=float(g["discount"]), tau=float(g["tau"]),
               policy_noise=float(g["policy_noise"]), noise_clip=float(g["noise_clip"]), policy_freq=int(g["policy_freq"]), act_noise=float(g["act_noise"]),
               batch_size=B, replay_size=4096)
    L = algo.learner
    ak, ck = [str(k) for k in g["actor_keys"]], [str(k) for k in g["critic_keys"]]
    L.actor.load_list([g["actor0." + k] for k in ak]); L.actor_t.load_list([g["actor0." + k] for k in ak])
    for i in range(2):
        L.q[i].load_list([g["critic0." + k] for k in ck[6 * i:6 * i + 6]]); L.q_t[i].load_list([g["critic0." + k] for k in ck[6 * i:6 * i + 6]])
    lens = [int(x) for x in g["lens"]]
    seen = 0
    for r in range(int(g["rounds"])):
        p = "r%d_" % r
        T = int(g[p + "T"])
        ep = [min(lens[(r * P + i) % len(lens)], mtl) for i in range(P)]          # worker i's episode length this round
        assert sum(ep) == T
        off = np.concatenate([[0], np.cumsum(ep)])
        ex, idx, sm = g[p + "explore"], g[p + "idx"], g[p + "smooth"]
        out = algo.reference_round(mtl, explore_fn=lambda e, t: ex[off[e] + t] if t < ep[e] else 0.0, index_fn=lambda it: idx[it], smooth_fn=lambda it: sm[it])
        assert out["transitions"] == T
        rp = algo.replay
        sl = slice(seen, seen + T)
        np.testing.assert_allclose(rp.s[sl].cpu().numpy(), g[p + "s"], atol=2e-5, err_msg="states round %d" % r)
        np.testing.assert_allclose(rp.a[sl].cpu().numpy(), g[p + "a"], atol=2e-5, err_msg="actions round %d" % r)
        np.testing.assert_allclose(rp.s2[sl].cpu().numpy(), g[p + "s2"], atol=2e-5)
        np.testing.assert_allclose(rp.r[sl].cpu().numpy(), g[p + "rew"], atol=1e-5)
        np.testing.assert_array_equal(rp.nd[sl].cpu().numpy(), 1.0 - g[p + "d"])          # done_bool, incl. the time limit
        seen += T
        np.testing.assert_allclose([out["avg_q1"], out["q_loss"], out["pi_loss"]], [float(g[p + "avg_q1"]), float(g[p + "q_loss"]), float(g[p + "pi_loss"])],
                                   rtol=5e-4, atol=5e-6, err_msg="statistics round %d" % r)
        for nm, nets, keys in (("actor", [L.actor], ak), ("critic", L.q, ck)):
            for k, v in zip(keys, [v for net in nets for v in net.views()]):
                d = np.abs(v.cpu().numpy() - g[p + nm + "." + k])
                assert (d > 5e-5).mean() < 2e-2 and d.max() < 2.1e-3 * (r + 1), (r, nm, k, (d > 5e-5).mean(), d.max())      # Adam at lr 1e-3: a sign tie at g ~ 0 moves a weight by 2e-3
    for nm, nets, keys in (("actor_target", [L.actor_t], ak), ("critic_target", L.q_t, ck)):
        for k, v in zip(keys, [v for net in nets for v in net.views()]):
            d = np.abs(v.cpu().numpy() - g[nm + "." + k])
            assert d.max() < 2e-4, (nm, k, d.max())


def test_fused_recurrent_step_equals_the_per_launch_chain(dev):
    """apx_lstm_step (the rollout's policy step as one launch: normalisation, init_hidden_state of restarted rows, two LSTMCell(128), head, action noise) against the
    chain it replaces - normalise, masked state reset, apx_lstm_forward(T = 1), mu + sigma * noise - over several steps with a carried state, for the actor (O = 10) and
    the critic (O = 1) shapes, ragged batch sizes included.  Same arithmetic up to the order of the K sum (one accumulation over [x | h] instead of two GEMMs): 2e-5."""
    from apex_amd import engine
    g = torch.Generator(device=dev); g.manual_seed(5)
    for D, O, B in ((49, 10, 2048), (49, 1, 300), (40, 10, 17), (64, 10, 33)):
        net = engine.Lstm(D, 128, 2, O, dev)
        assert net.step_supported()
        net.params.copy_(torch.randn(net.n, device=dev, generator=g) * 0.08)
        net.pack_step()
        mean = torch.randn(D, device=dev, generator=g); std = torch.rand(D, device=dev, generator=g) + 0.5
        hc_a = torch.randn(2, 2, B, 128, device=dev, generator=g) * 0.3; hc_b = hc_a.clone()
        reset = None
        for t in range(4):
            x = torch.randn(B, D, device=dev, generator=g)
            noise = torch.randn(B, O, device=dev, generator=g)
            # reference chain
            if reset is not None:
                hc_a.masked_fill_((reset != 0).view(1, 1, B, 1), 0.0)
            mu_a = net.forward(((x - mean) / std).contiguous(), hc=hc_a)
            act_a = mu_a + 0.2 * noise
            # fused
            act_b = torch.empty(B, O, device=dev)
            mu_b = net.step(x, hc_b, mean, std, reset=reset, noise=noise, sigma=0.2, act_out=act_b)
            np.testing.assert_allclose(mu_b.cpu().numpy(), mu_a.cpu().numpy(), rtol=0, atol=2e-5, err_msg="mu D=%d O=%d B=%d t=%d" % (D, O, B, t))
            np.testing.assert_allclose(act_b.cpu().numpy(), act_a.cpu().numpy(), rtol=0, atol=2e-5)
            np.testing.assert_allclose(hc_b.cpu().numpy(), hc_a.cpu().numpy(), rtol=0, atol=2e-5, err_msg="state D=%d O=%d B=%d t=%d" % (D, O, B, t))
            reset = (torch.rand(B, device=dev, generator=g) < 0.3).to(torch.uint8) * 2      # (the done flags are 0 / 1 / 2)
        # a prepared input without normalisation, no noise, no reset
        y_a = net.forward(x.contiguous(), hc=hc_a); y_b = net.step(x, hc_b)
        np.testing.assert_allclose(y_b.cpu().numpy(), y_a.cpu().numpy(), rtol=0, atol=2e-5)
    assert not engine.Lstm(50, 64, 2, 10, dev).step_supported() and not engine.Lstm(80, 128, 2, 10, dev).step_supported()


def test_rec_gather_equals_the_torch_assembly(dev):
    """apx_rec_gather (the padded recurrent minibatch in one launch) against the torch ops it replaces: index_select of the grid rows times the 0 / 1 mask,
    (x - mean) / std, SymmetricEnv.mirror_clock_observation on the PADDED observations (zero rows included: the reference's mirror term is not masked), cat along
    the batch axis.  Bit-exact: same formulas, same libm routines."""
    from apex_amd import engine
    from apex_amd.ppo import MIRRORED_OBS, MIRRORED_ACTS, CLOCK_INDS
    g = torch.Generator(device=dev); g.manual_seed(9)
    L = engine.RecurrentPPOLearner(50, 10, 128, 2, dev, 0.13, mirrored_obs=MIRRORED_OBS, mirrored_acts=MIRRORED_ACTS, clock_inds=CLOCK_INDS)
    L.obs_mean.copy_(torch.randn(50, device=dev, generator=g)); L.obs_std.copy_(torch.rand(50, device=dev, generator=g) + 0.5)
    rows = 5000
    obs = torch.randn(rows, 50, device=dev, generator=g) * 0.7
    obs[:, list(CLOCK_INDS)] = torch.sin(torch.rand(rows, 2, device=dev, generator=g) * 6.28)      # the clock columns live in [-1, 1]
    act = torch.randn(rows, 10, device=dev, generator=g); ret = torch.randn(rows, device=dev, generator=g); adv = torch.randn(rows, device=dev, generator=g)
    for T, B in ((37, 21), (1, 3), (64, 256)):
        idx = torch.randint(0, rows, (T, B), device=dev, generator=g)
        lens = torch.randint(1, T + 1, (B,), device=dev, generator=g)
        idx = torch.where(torch.arange(T, device=dev).view(T, 1) < lens.view(1, B), idx, -1).contiguous()
        for mirror in (True, False):
            o, a, r, d, m, (xn, xa) = L.gather(idx, obs, act, ret, adv, mirror=mirror)
            valid = idx >= 0; gi = idx.clamp(min=0).view(-1)
            pick = lambda x, dd: (x.view(rows, dd).index_select(0, gi) * valid.view(-1, 1)).view(T, B, dd)
            o_ref = pick(obs, 50)
            norm = lambda z: (z - L.obs_mean) / L.obs_std
            assert torch.equal(o, o_ref) and torch.equal(a, pick(act, 10)) and torch.equal(r, pick(ret, 1)) and torch.equal(d, pick(adv, 1))
            assert torch.equal(m, valid.float().unsqueeze(-1))
            assert torch.equal(xn, norm(o_ref))
            if mirror:
                ref = torch.cat([norm(o_ref), norm(L.mirror_obs(o_ref))], dim=1)
                assert xa.shape == ref.shape
                np.testing.assert_allclose(xa.cpu().numpy(), ref.cpu().numpy(), rtol=0, atol=2e-6)
            else:
                assert xa is xn
    # the index formed inside the kernel from the trajectory list (column, t0, t1) + a selection = the explicit [T_max, B] index of padded_index()
    Ngrid, Tgrid = 25, 200                                                 # rows = 5000 = Tgrid * Ngrid
    trajs = []
    for n in range(Ngrid):
        cuts = sorted(set(np.random.RandomState(n).randint(1, Tgrid, 6).tolist() + [Tgrid]))
        t0 = 0
        for t1 in cuts:
            trajs.append((n, t0, t1)); t0 = t1
    trajs = np.array(trajs, dtype=np.int64)
    sel = np.random.RandomState(1).permutation(len(trajs))[:40].astype(np.int64)
    lens = trajs[sel, 2] - trajs[sel, 1]; Tm = int(lens.max())
    idx = np.full((Tm, len(sel)), -1, dtype=np.int64)
    for b, (n, t0, t1) in enumerate(trajs[sel]):
        idx[: t1 - t0, b] = np.arange(t0, t1) * Ngrid + n
    ref = L.gather(torch.as_tensor(idx, device=dev), obs, act, ret, adv, mirror=True)
    got = L.gather(None, obs, act, ret, adv, mirror=True, traj=torch.as_tensor(trajs, device=dev), sel=torch.as_tensor(sel, device=dev), grid_cols=Ngrid, t_max=Tm)
    for a, b in zip(ref[:5] + ref[5], got[:5] + got[5]):
        assert torch.equal(a, b)


def test_lstm_forward_backward_large_batch_equals_small_batches(dev):
    """The large-shape paths of the recurrent learner - the 128 x 128 GEMM with a k-contiguous B operand and bias epilogue (upper layers' input projection), the K-chunk
    slabs + grad_reduce of the weight gradients, the prefetching sequence kernels - against the same network run over 64-column slices of the batch, which stay on the
    64 x 64 kernels and the atomic split-K: outputs 1e-5, gradients 2e-4 relative to their scale (sums over 16 384 rows in a different order)."""
    from apex_amd import engine
    g = torch.Generator(device=dev); g.manual_seed(3)
    T, B, D, O = 8, 2048, 49, 10
    net = engine.Lstm(D, 128, 2, O, dev)
    net.params.copy_(torch.randn(net.n, device=dev, generator=g) * 0.08)
    x = torch.randn(T, B, D, device=dev, generator=g); dy = torch.randn(T, B, O, device=dev, generator=g) * 0.1
    y, x3, save = net.forward(x, keep=True)
    gr = torch.zeros(net.n, device=dev); net.backward(gr, x3, save, dy)
    gr_s = torch.zeros(net.n, device=dev); ys = []
    for b0 in range(0, B, 64):
        xs = x[:, b0:b0 + 64].contiguous()
        y1, x31, save1 = net.forward(xs, keep=True); ys.append(y1)
        net.backward(gr_s, x31, save1, dy[:, b0:b0 + 64].contiguous())
    np.testing.assert_allclose(y.cpu().numpy(), torch.cat(ys, 1).cpu().numpy(), rtol=0, atol=1e-5)
    for name, a, b in zip(("Wih0", "Whh0", "bih0", "bhh0", "Wih1", "Whh1", "bih1", "bhh1", "Wo", "bo"), net.views(gr), net.views(gr_s)):
        scale = float(b.abs().max()) + 1e-6
        assert float((a - b).abs().max()) < 2e-4 * scale, (name, float((a - b).abs().max()), scale)
