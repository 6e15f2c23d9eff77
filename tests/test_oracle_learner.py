"""Pins oracle/learner.py against golden vectors produced by the reference itself (G1-G5)."""
import os

import numpy as np

from oracle import learner as L

ACTOR_KEYS = ["actor_layers.0.weight", "actor_layers.0.bias", "actor_layers.1.weight", "actor_layers.1.bias",
              "means.weight", "means.bias"]
CRITIC_KEYS = ["critic_layers.0.weight", "critic_layers.0.bias", "critic_layers.1.weight", "critic_layers.1.bias",
               "network_out.weight", "network_out.bias"]


def test_g1_finish_path(golden_dir):
    g = np.load(os.path.join(golden_dir, "g1_finish_path.npz"))
    for c in range(int(g["n_cases"])):
        ret = L.discounted_returns(g[f"c{c}_rewards"], g[f"c{c}_lens"], g[f"c{c}_last_vals"], float(g[f"c{c}_gamma"]))
        np.testing.assert_allclose(ret, g[f"c{c}_returns"], rtol=1e-12, atol=1e-12)
        assert list(np.cumsum(np.r_[0, g[f"c{c}_lens"]])) == list(g[f"c{c}_traj_idx"])   # bit-exact step indices


def test_g1_grid_layout_equals_list_layout():
    rng = np.random.RandomState(0)
    T, N = 17, 5
    rew = rng.randn(T, N); end = (rng.rand(T, N) < 0.2).astype(np.int32); boot = rng.randn(T, N) * (rng.rand(T, N) < 0.5)
    last = rng.randn(N)
    grid = L.returns_scan_grid_boot(rew, end, boot, last, 0.99)
    for n in range(N):
        lens, lv, t0 = [], [], 0
        for t in range(T):
            if end[t, n]:
                lens.append(t + 1 - t0); lv.append(boot[t, n]); t0 = t + 1
        if t0 < T:
            lens.append(T - t0); lv.append(last[n])
        ref = L.discounted_returns(rew[:, n], lens, lv, 0.99)
        np.testing.assert_allclose(grid[:, n], ref, rtol=1e-13)


def test_g2_adv_norm(golden_dir):
    g = np.load(os.path.join(golden_dir, "g2_adv_norm.npz"))
    for c in range(int(g["n_cases"])):
        adv = L.normalize_advantages(g[f"c{c}_returns"], g[f"c{c}_values"], 1e-5)
        np.testing.assert_allclose(adv, g[f"c{c}_adv"], rtol=1e-5, atol=1e-6)


def test_g3_forward(golden_dir):
    g = np.load(os.path.join(golden_dir, "g3_policy_forward.npz"))
    assert list(g["actor_keys"]) == ACTOR_KEYS and list(g["critic_keys"]) == CRITIC_KEYS
    A = [g["actor." + k] for k in ACTOR_KEYS]
    C = [g["critic." + k] for k in CRITIC_KEYS]
    mu = L.actor_mean(A, g["obs"], g["obs_mean"], g["obs_std"])
    np.testing.assert_allclose(mu, g["mean"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(L.critic_value(C, g["obs"]), g["value_train"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(L.critic_value(C, g["obs"], g["obs_mean"], g["obs_std"], training=False),
                               g["value_eval"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(L.gaussian_logp(mu, float(g["fixed_std"]), g["act"]), g["logp"], rtol=1e-5)
    assert abs(L.gaussian_entropy(float(g["fixed_std"])) - float(g["entropy"])) < 1e-6
    np.testing.assert_allclose(g["init_row_norm_l0"], 1.0, atol=1e-5)       # normc: unit rows (base.py:7-13)
    np.testing.assert_allclose(g["init_row_norm_means"], 0.01, atol=1e-6)   # mean head * 0.01 (actor.py:178)
    assert float(g["init_bias_abs_max"]) == 0.0


def test_g5_mirror(golden_dir):
    g = np.load(os.path.join(golden_dir, "g5_mirror.npz"))
    Mo = L.mirror_matrix(list(g["mirrored_obs"])); Ma = L.mirror_matrix(list(g["mirrored_acts"]))
    np.testing.assert_array_equal(Mo, g["obs_mirror_matrix"]); np.testing.assert_array_equal(Ma, g["act_mirror_matrix"])
    np.testing.assert_allclose(L.mirror_clock_observation(g["obs"].astype(np.float64), Mo, [46, 47]), g["mirror_obs"],
                               atol=1e-6)
    np.testing.assert_allclose(g["act"] @ Ma, g["mirror_act"], atol=1e-7)


def test_g4_update_policy(golden_dir):
    g = np.load(os.path.join(golden_dir, "g4_update_policy.npz"))
    g5 = np.load(os.path.join(golden_dir, "g5_mirror.npz"))
    for c in range(int(g["n_cases"])):
        p = f"c{c}_"
        actor = [g[p + "actor0." + k] for k in ACTOR_KEYS]
        old = [g[p + "old." + k] for k in ACTOR_KEYS]
        critic = [g[p + "critic0." + k] for k in CRITIC_KEYS]
        mirror = bool(g[p + "mirror"])
        oa, oc = L.Adam(actor), L.Adam(critic)
        scal_all = []
        for s in range(int(g[p + "nsteps"])):
            scal, actor, critic = L.ppo_update(
                actor, old, critic, oa, oc, g[p + f"s{s}_obs"], g[p + f"s{s}_act"], g[p + f"s{s}_ret"], g[p + f"s{s}_adv"],
                g[p + "obs_mean"], g[p + "obs_std"], np.exp(-1.5), entropy_coeff=float(g[p + "entropy_coeff"]),
                M_obs=g5["obs_mirror_matrix"] if mirror else None, M_act=g5["act_mirror_matrix"] if mirror else None)
            scal_all.append(scal)
        np.testing.assert_allclose(np.array(scal_all), g[p + "scalars"], rtol=2e-5, atol=1e-7)
        for k, w in zip(ACTOR_KEYS, actor):
            ref = g[p + "actor1." + k]
            # Adam's first steps move every weight by ~lr*sign(g); allow a vanishing fraction of sign flips at g~0
            bad = np.abs(w - ref) > 2e-6
            assert bad.mean() < 1e-3, (c, k, bad.mean())
        for k, w in zip(CRITIC_KEYS, critic):
            bad = np.abs(w - g[p + "critic1." + k]) > 2e-6
            assert bad.mean() < 1e-3, (c, k, bad.mean())


def test_g4b_epoch_of_small_minibatches(golden_dir):
    """G4b: an epoch of minibatch-64 / 32 / 128 / 16 optimiser steps on the 2 x 256 networks (the reference's CLI default minibatch, apex.py:242): the oracle's
    per-step update against the reference's per-step 6-tuples and the post-epoch parameters (inputs regenerated from the fixture's seeds)."""
    from golden_util import EPOCH_CASES, epoch_case_inputs, check_slim
    g = np.load(os.path.join(golden_dir, "g4b_epoch_h256.npz"))
    g5 = np.load(os.path.join(golden_dir, "g5_mirror.npz"))
    assert int(g["n_cases"]) == len(EPOCH_CASES)
    for c, (mirror, mb, nb, adam_t0) in enumerate(EPOCH_CASES):
        inp = epoch_case_inputs(c)
        actor, old, critic = inp["actor"], inp["old"], inp["critic"]
        oa, oc = L.Adam(actor), L.Adam(critic)
        oa.t = oc.t = adam_t0 - 1
        scal_all = []
        for k in range(nb):
            idx = inp["perm"][k * mb:(k + 1) * mb]
            scal, actor, critic = L.ppo_update(actor, old, critic, oa, oc, inp["obs"][idx], inp["act"][idx], inp["ret"][idx, None], inp["adv"][idx, None],
                                               inp["obs_mean"], inp["obs_std"], np.exp(-1.5),
                                               M_obs=g5["obs_mirror_matrix"] if mirror else None, M_act=g5["act_mirror_matrix"] if mirror else None)
            scal_all.append(scal)
        np.testing.assert_allclose(np.array(scal_all), g[f"c{c}_scalars"], rtol=3e-5, atol=2e-7)
        for i, w in enumerate(actor):
            check_slim(w, g[f"c{c}_actor1.{i}"], atol=2e-4, frac_tol=3e-6, frac=5e-3, err_msg=(c, "actor", i))
        for i, w in enumerate(critic):
            check_slim(w, g[f"c{c}_critic1.{i}"], atol=2e-4, frac_tol=3e-6, frac=5e-3, err_msg=(c, "critic", i))


def _g15a_grid(g):
    idx = g["traj_idx"]; T = int(idx[-1])
    rew = g["rewards"].reshape(T, 1); end = np.zeros((T, 1), np.uint8); boot = np.zeros((T, 1))
    for k, e in enumerate(idx[1:]):
        end[e - 1, 0] = 1; boot[e - 1, 0] = g["last_vals"][k]
    return rew, end, boot


def test_g15a_ppo_sample_control_flow(golden_dir):
    """The reference's own PPO.sample on a scripted toy env: episode boundaries bit-exact, truncation at max_traj_len
    bootstraps with V(s_T), termination with 0, returns reproduced by the grid scan."""
    g = np.load(os.path.join(golden_dir, "g15a_ppo_sample.npz"))
    lens = np.diff(g["traj_idx"])
    want = [min(int(L), int(g["max_traj_len"])) for L in g["scripted_lens"]]
    assert list(lens) == [want[k % len(want)] for k in range(len(lens))]                    # bit-exact episode-step indices
    assert list(lens) == list(g["ep_lens"])
    truncated = np.array([g["scripted_lens"][k % 8] > g["max_traj_len"] for k in range(len(lens))])
    assert ((np.abs(g["last_vals"]) > 0) == truncated).all()                                # (not done) * V(s_T), ppo.py:183-184
    rew, end, boot = _g15a_grid(g)
    ret = L.returns_scan_grid_boot(rew, end, boot, np.zeros(1), float(g["gamma"]))
    np.testing.assert_allclose(ret[:, 0], g["returns"], rtol=1e-6, atol=1e-7)
    for k in range(len(lens)):
        np.testing.assert_allclose(g["ep_returns"][k], g["rewards"][g["traj_idx"][k]:g["traj_idx"][k + 1]].sum(), rtol=1e-12)


def test_g15b_whole_train_loop(golden_dir):
    """G15b: the reference's whole PPO.train on a toy env (tools/refprobe/gen_golden_train.py).  The oracle replays the
    recorded buffers: returns from (rewards, traj_idx, bootstrap rule), normalised advantages, then all 3 x 7 minibatch
    updates per iteration in the recorded order -> every 6-tuple and the parameters after each iteration."""
    g = np.load(os.path.join(golden_dir, "g15b_ppo_train.npz"))
    g5 = np.load(os.path.join(golden_dir, "g5_mirror.npz"))
    actor = [g["actor0." + k] for k in ACTOR_KEYS]; critic = [g["critic0." + k] for k in CRITIC_KEYS]
    oa, oc = L.Adam(actor), L.Adam(critic)
    mb, mtl = int(g["minibatch"]), int(g["max_traj_len"])
    want_lens = [min(int(x), mtl) for x in g["lens"]]
    steps = 0
    for it in range(int(g["n_itr"])):
        p = "it%d." % it
        idx = g[p + "traj_idx"]; lens = np.diff(idx)
        assert list(lens) == [want_lens[(int(g[p + "k0"]) + j) % len(want_lens)] for j in range(len(lens))]   # bit-exact indices
        assert list(lens) == list(g[p + "ep_lens"])
        obs, act = g[p + "states"].astype(np.float64), g[p + "actions"].astype(np.float64)
        # values / means of the sampling policy from the oracle's forward
        np.testing.assert_allclose(L.critic_value(critic, obs).reshape(-1), g[p + "values"], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(L.actor_mean(actor, obs, g["obs_mean"], g["obs_std"]), g[p + "mu"], rtol=1e-4, atol=2e-6)
        # bootstrap rule: truncated episodes (scripted length > max_traj_len) carry V(s_T), terminated ones 0
        rew = g[p + "rewards"]; ret_ref = g[p + "returns"]
        last_vals = np.array([(ret_ref[e - 1] - rew[e - 1]) / float(g["gamma"]) for e in idx[1:]])
        trunc = np.array([int(g["lens"][(int(g[p + "k0"]) + j) % len(want_lens)]) > mtl for j in range(len(lens))])
        assert ((np.abs(last_vals) > 1e-12) == trunc).all()
        ret = L.discounted_returns(rew, lens, last_vals, float(g["gamma"]))
        np.testing.assert_allclose(ret, ret_ref, rtol=1e-7, atol=1e-7)      # last_vals are recovered from the returns
        adv = L.normalize_advantages(ret.astype(np.float32), g[p + "values"])
        old = [w.copy() for w in actor]
        scal_all = []
        for e in range(int(g[p + "epochs_run"])):
            order = g[p + "idx"][e]
            for k in range(len(order) // mb):
                ii = order[k * mb:(k + 1) * mb]
                scal, actor, critic = L.ppo_update(actor, old, critic, oa, oc, obs[ii], act[ii], ret[ii].astype(np.float32).reshape(-1, 1),
                                                   np.asarray(adv).reshape(-1, 1)[ii],
                                                   g["obs_mean"], g["obs_std"], np.exp(-1.5), M_obs=g5["obs_mirror_matrix"],
                                                   M_act=g5["act_mirror_matrix"])
                scal_all.append(scal)
        ref = g[p + "scal"].reshape(-1, 6)
        np.testing.assert_allclose(np.array(scal_all), ref, rtol=3e-5, atol=2e-7)
        for k, w in zip(ACTOR_KEYS, actor):
            d = np.abs(w - g[p + "actor." + k]); assert (d > 5e-6).mean() < 2e-3 and d.max() < 5e-4, (it, k, d.max())
        for k, w in zip(CRITIC_KEYS, critic):
            d = np.abs(w - g[p + "critic." + k]); assert (d > 5e-6).mean() < 2e-3 and d.max() < 5e-4, (it, k, d.max())
        steps += len(rew)
        assert steps == int(g["timesteps"][it])
        np.testing.assert_allclose(g["train_return"][it], g[p + "ep_returns"].mean(), rtol=1e-12)
        np.testing.assert_allclose(g["mean_eplen"][it], g[p + "ep_lens"].mean(), rtol=1e-12)
    assert len(g["scalar_names"]) == 13


def test_g21_normalization_params(golden_dir):
    """Row a12: the numpy restatement of get_normalization_params (rl/envs/normalize.py:11-48) reproduces the reference's (mean, std)
    on the toy envs of golden G21 with the captured noise."""
    g = np.load(os.path.join(golden_dir, "g21_normalization_params.npz"))
    lens = [int(x) for x in g["lens"]]

    class Toy:
        def __init__(self, w): self.k = 100 * w
        def reset(self):
            self.k += 1; self.t = 0; self.L = lens[(self.k - 1) % len(lens)]
            self.x = np.cos(np.arange(50) * 0.1 * self.k)
            return self.x.copy()
        def step(self, a):
            self.t += 1
            self.x = 0.9 * self.x + 0.1 * np.tile(a, 5) + 0.01
            return self.x.copy(), 0.0, self.t >= self.L, {}

    W = [g["actor." + k].astype(np.float64) for k in ACTOR_KEYS]
    mean, std = L.normalization_params(W, [Toy(w) for w in range(int(g["procs"]))], g["noise"].astype(np.float64), float(g["noise_std"]))
    np.testing.assert_allclose(mean, g["mean"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(std, g["std"], rtol=1e-6, atol=1e-6)
