"""CPU: the ENV kernels' sources (apex_amd/csrc/env.hip with cassie_lane.h, cassie_complete.h, estimator_lane.h: reset, env step, substep, one-launch rollout) compiled for
the host under the lane-exact wave emulation of tools/hipemu - 64 fibers per wave, every DPP operand / ds_bpermute / readlane / ballot an exchange between them, the
gfx950 inline-assembly dialect restated per lane in tools/hipemu/gfx950/lane_ops.h - and driven through the SAME host code (apex_amd/vecenv.py, the C ABI of env.hip) and
the SAME test bodies as the GPU parity tests of tests/test_gpu_env.py, against the fp64 oracle.  The redirection lives in this file only (a fixture swaps the loaded
library handle and the device hooks for the duration of a test); the product has no CPU path.

What this pins without a GPU: the lane map, every index and LDS offset, the tree / factor / sweep / finish stages as written, the contact detection, the complete-row
path, the estimator, reward, reset images, the in-kernel restart of the one-launch rollout.  What it cannot see: gfx950 code generation (hazard distances of the
hand-spaced DPP sequences, register allocation), v_rcp_f32 / fast-math rounding, speed."""
import contextlib
import os

import numpy as np
import pytest
import torch

from test_kernel_emulation_learner import CLANG, _NoStream, emulated_library

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture()
def dev(monkeypatch):
    """torch.device('cpu') with apex_amd.vecenv / engine talking to the emulated kernel sources for this test only"""
    if not os.path.exists(CLANG):
        pytest.skip("no host clang++")
    from apex_amd import _lib, engine, vecenv
    lib = emulated_library()
    lib.apx_emul_set_workgroups(0)
    monkeypatch.setattr(_lib, "_lib", lib)
    monkeypatch.setattr(engine, "_need_gpu", lambda *ts: None)
    monkeypatch.setattr(engine, "_stream", lambda: None)
    monkeypatch.setattr(vecenv, "_stream", lambda: None)
    monkeypatch.setattr(vecenv, "_device", lambda index: torch.device("cpu"))
    monkeypatch.setattr(vecenv, "_on_device", lambda t: True)
    monkeypatch.setattr(torch.cuda, "current_stream", lambda *a, **k: _NoStream())
    monkeypatch.setattr(torch.cuda, "Stream", lambda *a, **k: _NoStream())
    monkeypatch.setattr(torch.cuda, "Event", lambda *a, **k: _NoStream())
    monkeypatch.setattr(torch.cuda, "stream", lambda s: contextlib.nullcontext())
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a, **k: None)
    return torch.device("cpu")


def test_emulated_smoke_reset_and_one_teacher_forced_step(dev):
    """the body of __graft_entry__.smoke()'s env half: reset observation against the oracle, then one teacher-forced env step (kernel state overwritten with the
    oracle's) at the parity suite's fixed tolerances"""
    import __graft_entry__ as G
    G._smoke_env(dev)


def _gpu_env_tests():
    from tests import test_gpu_env as G
    return G


def test_emulated_reset_parameters_and_first_substeps(dev):
    """bodies of the GPU tests: randomised model parameters and the reset observation against the oracle (64 envs), mj_setConst's inverse weights, 20 raw substeps"""
    G = _gpu_env_tests()
    G.test_set_const_invweights(dev)
    G.test_reset_obs_and_params(dev)
    G.test_substeps_track_oracle(dev)


def test_emulated_env_steps_integers_bit_exact(dev):
    """Free-running env steps from the same reset with the same random actions (a short form of test_gpu_env.py::test_env_steps_vs_oracle: 3 steps, the first 12 of the
    64 envs compared): time / phase / cycle counter / RNG counter and done flags bit-exact, slow observation groups within the GPU test's drift bound"""
    G = _gpu_env_tests()
    genv, oenv = G._mk(True, 3)
    oenv = oenv[:12]
    genv.reset(); [e.reset() for e in oenv]
    rng = np.random.RandomState(0)
    grp = [(slice(0, 15), 3e-3), (slice(34, 40), 3e-3), (slice(46, 50), 1e-5)]
    for t in range(3):
        act = (rng.randn(64, 10) * 0.15).astype(np.float32)
        obs, rew, done, fin = genv.step(torch.tensor(act), auto_reset=False)
        obs, done = obs.numpy(), done.numpy()
        for i, e in enumerate(oenv):
            o, r, d = e.step(act[i].astype(np.float64))
            assert d == done[i], (t, i)
            for sl, tol in grp:
                np.testing.assert_allclose(obs[i, sl], o[sl], atol=tol * (t + 1), err_msg=f"t={t} env={i} obs{sl}")
        ints = genv.get_field("ints").numpy()[:12]
        np.testing.assert_array_equal(ints[:, [0, 1, 2, 3]], np.stack([e.get("ints") for e in oenv])[:, [0, 1, 2, 5]])
    genv.close()


def test_emulated_prepared_resets_are_bit_identical_to_computed_ones(dev):
    """short form of test_gpu_env.py::test_prepared_resets_are_bit_identical_to_computed_ones: two envs with the same seed, one whose reset ring is prepared ahead of every
    step and one never (its restarts compute the image on demand: the `need` rows of env_reset_kernel next to rows that only restart), 64 envs, episodes of 2 steps, 5
    steps: observations, rewards, done flags and the state fields bit-identical; every env restarted twice"""
    from apex_amd.vecenv import CassieVecEnv
    for kw in (dict(), dict(env_name="CassieTraj-v0")):
        a = CassieVecEnv(n_envs=64, seed=31, max_traj_len=2, **kw); b = CassieVecEnv(n_envs=64, seed=31, max_traj_len=2, **kw)
        a.set_refill(False); b.set_refill(False)
        oa, ob = a.reset().clone(), b.reset().clone()
        assert torch.equal(oa, ob)
        g = torch.Generator(); g.manual_seed(4)
        for t in range(5):
            if t != 3:
                a.prepare_resets()
            act = torch.randn(64, 10, generator=g) * 0.3
            oa, ra, da, fa = a.step(act); ob, rb, db, fb = b.step(act)
            assert torch.equal(oa, ob) and torch.equal(ra, rb) and torch.equal(da, db), (kw, t)
        for name in ("qpos", "qvel", "qacc_warm", "mass", "damping", "friction", "floor", "motor_noise", "joint_noise", "snap", "cmd", "fwd", "tq_fifo", "menc", "est"):
            assert torch.equal(a.get_field(name), b.get_field(name)), (kw, name)
        ia, ib = a.get_field("ints_bits").view(torch.int32), b.get_field("ints_bits").view(torch.int32)
        assert torch.equal(ia, ib) and int(ia[:, 9].min()) >= 2
        assert float(a.get_field("reset_miss")[0, 0]) == 0 and float(b.get_field("reset_miss")[0, 0]) == 0
        a.close(); b.close()


def _small_ppo(dev, n_envs, T, max_traj_len, seed):
    from apex_amd.vecenv import CassieVecEnv
    from apex_amd.ppo import PPO
    env = CassieVecEnv(n_envs=n_envs, seed=seed, max_traj_len=max_traj_len)
    args = dict(gamma=0.99, lam=0.95, lr=1e-4, eps=1e-5, entropy_coeff=0.0, clip=0.2, minibatch_size=n_envs * T, epochs=1, num_steps=n_envs * T, max_traj_len=max_traj_len,
                max_grad_norm=0.05, mirror=True, std_dev=-1.5, seed=0, prepare_resets=False)
    a = PPO(args, "/tmp/apx_test_unused", env); a.init_networks(0)
    return a


def test_emulated_one_launch_rollout_equals_the_stepwise_loop(dev):
    """env_rollout_kernel (apx_rollout: per-wave actor forward, env step and in-kernel restart of every step in ONE launch) against the step-by-step loop over
    apx_mlp_forward / apx_env_step / masked env_reset_kernel fed with the same action noise - a short form of test_gpu_ppo.py::test_apx_rollout_equals_the_stepwise_loop
    and ::test_one_launch_rollout_restarts_match_the_stepwise_resets: 64 envs, 4 steps, episodes of 2 steps (every env restarts twice inside the launch: once from a ring
    image computed in place).  Done flags bit-equal; first observation bit-equal; actions mu + sigma noise to round-off; later observations / rewards by the GPU tests'
    population rule."""
    a = _small_ppo(dev, 64, 4, 2, 6)
    a.sample()
    noise = a.noise.clone()
    b = _small_ppo(dev, 64, 4, 2, 6)
    b.noise_fn = lambda t, out: out.copy_(noise[t])
    b.sample()
    assert torch.equal(a.b_obs[0], b.b_obs[0]) and torch.equal(a.b_mu[0], b.b_mu[0])
    np.testing.assert_allclose(a.b_act.numpy(), (a.b_mu + a.fixed_std * noise).numpy(), rtol=0, atol=3e-6)
    assert torch.equal(a.b_done, b.b_done) and int((a.b_done == 2).sum()) == 64 * 2
    for t in range(1, 4):
        d = (a.b_obs[t] - b.b_obs[t]).abs().numpy()
        assert (d <= 5e-4 * t).mean() > 0.99 and d.max() < 5.0, (t, (d <= 5e-4 * t).mean(), d.max())
    dr = (a.b_rew - b.b_rew).abs().numpy()
    assert (dr <= 5e-3).mean() > 0.99 and dr.max() < 0.1, dr.max()
    assert int(a.env.get_field("reset_miss")[0, 0]) == 0 and int(a.env.get_field("ints")[:, 9].min()) >= 2
    assert torch.isfinite(a.b_obs).all() and torch.isfinite(a.b_val).all()
