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
    for kw in (dict(), dict(env_name="CassieTraj-v0")) if os.environ.get("APX_EMUL_FULL") == "1" else (dict(),):
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


# The bodies of the GPU parity tests of tests/test_gpu_env.py, unchanged, on the emulated sources.  Default suite: the ones that take seconds; APX_EMUL_FULL=1: all of them
# (results of the last full run: profiles/r06_emulation_checks.txt).
full = pytest.mark.skipif(os.environ.get("APX_EMUL_FULL") != "1", reason="a minute or more of emulation each: APX_EMUL_FULL=1")


def test_emulated_crafted_states_and_saturation_flags(dev):
    """single substeps from crafted states (joint limits, shin / tarsus / third capsule end on the floor, hip-pitch capsule, pelvis sphere, leg-leg pairs: the COMPLETE
    row path of cassie_complete.h) against the oracle's complete row set; saturation flags"""
    G = _gpu_env_tests()
    G.test_single_substep_crafted_states(dev)
    G.test_saturation_flags_vs_oracle_crafted(dev)


def test_emulated_eval_entry_points(dev):
    """update_speed / reset_for_test, step_basic, the wrench on any body, CassieTraj-v0's reset and steps, the estimator record's recovery"""
    G = _gpu_env_tests()
    G.test_update_speed_and_reset_for_test_vs_oracle(dev)
    G.test_step_basic_vs_oracle(dev)
    G.test_apply_force_on_any_body_vs_oracle(dev)
    G.test_cassie_traj_v0_reset_and_steps_vs_oracle(dev)
    G.test_estimator_record_recovers_from_non_finite_state(dev)


def test_emulated_phase_command_profile(dev):
    _gpu_env_tests().test_phase_command_profile_vs_oracle(dev, "clock", 1)


@full
@pytest.mark.parametrize("name", ["test_env_steps_vs_oracle", "test_min_input_profile_vs_oracle", "test_fractional_phase_add_vs_oracle", "test_observation_history_stack",
                                  "test_diverged_env_ends_its_episode", "test_estimator_twin_from_identical_state", "test_teacher_forced_env_steps_random_actions",
                                  "test_teacher_forced_env_steps_on_walking_states"])
def test_emulated_gpu_env_test_body(dev, name):
    getattr(_gpu_env_tests(), name)(dev)


@full
@pytest.mark.parametrize("kind", ["slope", "noise", "hills"])
def test_emulated_heightfield_terrain(dev, kind):
    _gpu_env_tests().test_heightfield_terrain_vs_oracle(dev, kind)


@full
def test_emulated_phase_command_profile_library(dev):
    _gpu_env_tests().test_phase_command_profile_vs_oracle(dev, "library_clock", 2)


@full
def test_emulated_full_reset_and_apply_force(dev, golden_dir):
    _gpu_env_tests().test_full_reset_and_apply_force_vs_oracle(dev, golden_dir)


def _scenario_names():
    from tests.tf_scenarios import SCENARIOS
    return [s.name for s in SCENARIOS]


@pytest.mark.parametrize("name", [pytest.param(n, marks=() if n == "safety_zones" else full) for n in _scenario_names()])      # (one scenario in the default suite, all 16 with APX_EMUL_FULL=1)
def test_emulated_teacher_forced_scenario(dev, name):
    """every teacher-forced scenario of the GPU suite (safety zones, coupled zone, early / max_vel rewards, pushes, height fields, phase profile, eval entry points ..):
    kernel state overwritten with the oracle's before every env step, FIXED tolerances on the identical-row-set population"""
    _gpu_env_tests().test_teacher_forced_scenario(dev, name)


def test_emulated_td3_one_launch_collection(dev, tmp_path):
    """env_rollout_kernel<MODE 2> (apx_rollout_td3: deterministic actor with tanh head, clipped scalar exploration noise, in-kernel restarts) - the assertions of
    test_gpu_ppo.py::test_td3_one_launch_collection on 64 envs x 4 steps with episodes of 2 steps, update block 4 x 1 updates of 64 rows through the emulated learner"""
    from apex_amd.vecenv import CassieVecEnv
    from apex_amd.td3 import TD3
    from apex_amd import engine
    N, T = 64, 4
    env = CassieVecEnv(n_envs=N, seed=4, max_traj_len=2)
    algo = TD3(env, str(tmp_path), hidden=256, act_noise=0.3, batch_size=64, updates_per_step=1, replay_size=N * T, seed=1)
    algo.init_networks(0)
    p0 = algo.learner.actor.params.clone()
    out = algo.collect_and_train(T)
    assert out["updates"] == T and algo.replay.size == N * T
    g, noise = algo._g, algo._noise_last
    pre = engine.Mlp(50, 256, 10, dev); pre.params.copy_(p0)
    m = torch.tanh(pre.forward(g["obs"].view(T * N, 50))).view(T, N, 10)
    d = (g["mu"] - m).abs()
    big = g["obs"].abs().amax(-1, keepdim=True)
    assert float((d / (1.0 + big)).max()) < 2e-6, float((d / (1.0 + big)).max())
    np.testing.assert_allclose(g["act"].numpy(), (g["mu"] + 0.3 * noise.view(T, N, 1)).clamp(-1, 1).numpy(), rtol=0, atol=1e-6)
    assert int((g["done"] != 0).sum()) >= 2 * N
    R = algo.replay
    assert torch.equal(R.s[:N * T].view(T, N, 50), g["obs"]) and torch.equal(R.a[:N * T].view(T, N, 10), g["act"]) and torch.equal(R.r[:N * T].view(T, N), g["rew"])
    ended = g["done"] != 0
    assert torch.equal(R.nd[:N * T].view(T, N), (~ended).float())
    nxt = torch.cat([g["obs"][1:], g["nxt"].unsqueeze(0)])
    assert torch.equal(R.s2[:N * T].view(T, N, 50), torch.where(ended.unsqueeze(-1), g["fin"], nxt))
    assert not torch.equal(algo.learner.actor.params, p0) and torch.isfinite(algo.learner.actor.params).all()


def test_emulated_one_launch_recurrent_rollout(dev, tmp_path):
    """env_rollout_kernel<MODE 1> (apx_rollout_lstm: two LSTMCell(128) + head per wave inside the rollout, hidden state zeroed where an episode ends) - the identity of
    test_gpu_ppo.py::test_one_launch_recurrent_rollout_means_match_the_sequence_pass on 64 CassieTraj-v0 envs x 6 steps with episodes of 3 steps: the learner's padded
    sequence pass from the zero state over whole trajectories cut out of the grid reproduces the means the in-kernel actor produced step by step with its carried (h, c)"""
    from apex_amd.vecenv import CassieVecEnv
    from apex_amd.ppo_recurrent import RecurrentPPO
    N, T, mtl = 64, 6, 3
    env = CassieVecEnv(n_envs=N, seed=9, max_traj_len=mtl, env_name="CassieTraj-v0")
    args = dict(gamma=0.99, lam=0.95, lr=1e-4, eps=1e-5, entropy_coeff=0.0, clip=0.2, minibatch_size=64, epochs=1, num_steps=T * N, max_traj_len=mtl,
                max_grad_norm=0.05, mirror=True, seed=3, env_name="CassieTraj-v0")
    algo = RecurrentPPO(args, str(tmp_path), env, hidden=128, layers=2)
    algo.init_networks(0)
    L = algo.learner
    ret = algo.sample()
    noise = algo._noise_all
    done = algo.b_done.numpy()
    assert np.isfinite(algo.b_obs.numpy()).all() and np.isfinite(algo.b_rew.numpy()).all() and np.isfinite(ret.numpy()).all()
    assert (done == 2).sum() == N * (T // mtl)
    np.testing.assert_allclose(algo.b_act.numpy(), (algo._b_mu + algo.fixed_std * noise).numpy(), rtol=0, atol=1e-6)
    trajs = algo.trajectories()
    assert len(trajs) == 2 * N and (trajs[:, 1] > 0).any()
    idx = algo.padded_index(trajs)
    valid = idx >= 0
    gi = idx.clamp(min=0).view(-1)
    obs_p = (algo.b_obs.view(T * N, 50).index_select(0, gi) * valid.view(-1, 1)).view(idx.shape[0], len(trajs), 50)
    mu_seq = L.actor.forward(((obs_p - L.obs_mean) / L.obs_std).contiguous())
    mu_roll = algo._b_mu.view(T * N, 10).index_select(0, gi).view(idx.shape[0], len(trajs), 10)
    d = ((mu_seq - mu_roll) * valid.unsqueeze(-1)).abs().max()
    assert float(d) < 2e-5, float(d)


@full
def test_emulated_cli_ppo_run_directory_checkpoints_and_scalars(dev, tmp_path, golden_dir):
    """BASELINE configs[0]'s analogue END TO END on the emulated kernels: `apex.py ppo --env_name Cassie-v0 --reward clock` for 2 iterations of 64 envs x 8 steps (observation
    statistics, one-launch rollouts, returns, PPO epochs through the emulated learner): the run directory <logdir>/Cassie-v0/<md5[:6]>-seed0 with experiment.info /
    experiment.pkl, whole-module actor.pt / critic.pt that load with this repo's rl.policies classes, the reference's 13 scalar names - the assertions of
    test_gpu_ppo.py::test_cli_ppo_run_directory_checkpoints_and_scalars"""
    import json
    import apex
    rc = apex.main(["ppo", "--env_name", "Cassie-v0", "--reward", "clock", "--n_envs", "64", "--num_steps", "512", "--minibatch_size", "256",
                    "--n_itr", "2", "--input_norm_steps", "128", "--eval_every", "0", "--max_traj_len", "4", "--logdir", str(tmp_path), "--seed", "0"])
    # (--eval_every 0: no 256-env evaluation episodes, Test/Return = the batch return; --max_traj_len 4: episodes end inside the 8-step rollout, so there IS a return to beat and a checkpoint)
    assert rc in (0, None)
    runs = os.listdir(os.path.join(str(tmp_path), "Cassie-v0"))
    assert len(runs) == 1 and runs[0].endswith("-seed0") and len(runs[0].split("-")[0]) == 6
    run = os.path.join(str(tmp_path), "Cassie-v0", runs[0])
    for f in ("actor.pt", "critic.pt", "experiment.info", "experiment.pkl"):
        assert os.path.exists(os.path.join(run, f)), f
    actor = torch.load(os.path.join(run, "actor.pt"), weights_only=False)
    assert type(actor).__module__ == "rl.policies.actor" and type(actor).__name__ == "Gaussian_FF_Actor"
    assert torch.is_tensor(actor.obs_mean) and actor.obs_mean.shape == (50,) and abs(float(actor.fixed_std) - np.exp(-1.5)) < 1e-6
    y = actor(torch.zeros(50), deterministic=True)
    assert y.shape[-1] == 10 and torch.isfinite(y).all()
    g = np.load(os.path.join(golden_dir, "g15b_ppo_train.npz"))
    if os.path.exists(os.path.join(run, "scalars.jsonl")):
        names = set(json.loads(line)["tag"] for line in open(os.path.join(run, "scalars.jsonl")))
        assert names == set(str(x) for x in g["scalar_names"])
    else:
        assert any(f.startswith("events.out.tfevents") for f in os.listdir(run))


@full
def test_emulated_range_checked_build():
    """the env kernels' sources with -DAPX_CHECK under the emulation (tools/hipemu/build.sh check), driven through entry points, env kinds, profiles, a height field, restarts,
    the complete-row path and the one-launch rollout (tests/emul_check_worker.py): no S(f) / S.W(i) / S.I(f) index leaves its LDS region - the CPU twin of
    test_gpu_env.py::test_range_checked_build_sees_no_out_of_range_index"""
    import subprocess
    import sys
    if not os.path.exists(CLANG):
        pytest.skip("no host clang++")
    subprocess.check_call(["bash", os.path.join(REPO, "tools", "hipemu", "build.sh"), "check"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    out = subprocess.run([sys.executable, os.path.join(REPO, "tests", "emul_check_worker.py")], capture_output=True, text=True, timeout=3000,
                         env=dict(os.environ, APX_EMUL_LIB=os.path.join(REPO, "tools", "hipemu", "_build", "libapx_emul_check.so")))
    assert out.returncode == 0, out.stderr[-2000:]
    assert "RESULT clean" in out.stdout, out.stdout[-2000:]


def test_emulated_iteration_grids_bootstrap_and_returns(dev):
    """PPO.sample + PPO.update on the emulated env AND learner kernels (short form of test_gpu_ppo.py::test_iteration_grids_bootstrap_and_returns: 64 envs x 6 steps, episodes
    of 3): time-limit truncations carry the critic's value of the recorded FINAL observation as bootstrap and nothing else does, the returns equal the oracle's scan over the
    recorded grids, episode statistics respect the limit, one update epoch runs to finite losses"""
    from oracle import learner as OL
    a = _small_ppo(dev, 64, 6, 3, 2)
    ret, ep_rets, ep_lens = a.sample()
    T, N = 6, 64
    done, end, boot = a.b_done.numpy(), a.b_end.numpy(), a.b_boot.numpy()
    assert np.array_equal(end != 0, done != 0) and set(np.unique(done)) <= {0, 1, 2} and (done == 2).sum() >= 2 * N - 8
    vfin = a.learner.critic.forward(a.b_fin.view(T * N, 50)).view(T, N).numpy()
    np.testing.assert_allclose(boot[done == 2], vfin[done == 2], rtol=1e-5, atol=1e-6)
    assert np.all(boot[done != 2] == 0)
    last_val = a.learner.critic.forward(a.obs).view(-1).numpy()
    np.testing.assert_allclose(ret.numpy(), OL.returns_scan_grid_boot(a.b_rew.numpy(), end, boot, last_val, 0.99), rtol=1e-6, atol=1e-6)
    el = ep_lens.numpy()
    assert el.size == int((done != 0).sum()) and el.max() <= 3 and el.min() >= 1
    losses, kl, epochs_run = a.update(ret)
    assert np.all(np.isfinite(losses)) and np.isfinite(kl) and epochs_run == 1


@full
@pytest.mark.parametrize("name", ["test_recurrent_ppo_iteration_on_the_hip_env", "test_td3_driver_hbm_replay_and_updates", "test_td3_one_launch_collection",
                                  "test_compute_perturbs_batched", "test_eval_commands_batched", "test_batched_deterministic_evaluation", "test_normalization_params_golden_g21"])
def test_emulated_gpu_ppo_test_body(dev, tmp_path, golden_dir, name):
    """bodies of tests/test_gpu_ppo.py on the emulated env + learner kernels: recurrent PPO iteration (rollout grids -> padded whole-trajectory minibatches -> update -> checkpoint ->
    `apex.py eval` of the recurrent checkpoint), the TD3 driver (HBM replay, twin-critic updates, Polyak), TD3's one-launch collection at the GPU test's size, the batched push sweep and
    command-schedule evaluation (apex_amd/eval.py), deterministic evaluation, the observation-statistics golden G21"""
    import inspect
    from tests import test_gpu_ppo as P
    fn = getattr(P, name)
    have = dict(dev=dev, tmp_path=tmp_path, golden_dir=golden_dir)
    fn(**{k: have[k] for k in inspect.signature(fn).parameters})


@pytest.mark.parametrize("workload,extra", [("cassie_ppo", ["--n_envs", "64", "--rollout_len", "2", "--minibatch", "64", "--epochs", "1"]),
                                            ("cassietraj_recurrent", ["--epochs", "1", "--no_cpu_baseline"]), ("cassie_td3", ["--no_cpu_baseline"])])
def test_emulated_bench_lines(dev, monkeypatch, capsys, workload, extra):
    """bench.py's three workloads executed END TO END on the emulated kernels at a tiny shape (APX_BENCH_TINY: 64 envs, 2 - 3 steps; one timed step, no warm-up): every code
    path of the timed region and the whole JSON assembly run - round 5 shipped a NameError there that no CPU test could see.  Asserted: one JSON line with the contract's
    keys, the roofline and cpu_baseline objects on the headline, value = steps / time, "tiny_test_shape" in its config (such a line is never a measurement)."""
    import json
    import sys
    import bench
    monkeypatch.setattr(bench, "TINY", True)
    full_baseline = bench.cpu_baseline
    monkeypatch.setattr(bench, "cpu_baseline", lambda **kw: full_baseline(seconds_hint=0.3, **kw))      # (the oracle's sampling leg: a fraction of a second instead of 12)
    monkeypatch.setattr(torch.cuda, "set_device", lambda *a, **k: None)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "1", "--steps", "1", "--warmup", "0", "--workload", workload] + extra)
    monkeypatch.delenv("RANK", raising=False); monkeypatch.delenv("WORLD_SIZE", raising=False)
    bench.main()
    lines = [ln for ln in capsys.readouterr().out.strip().split("\n") if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 1 and d["dtype"] == "f32" and d["vs_baseline"] is None and d["config"]["tiny_test_shape"] is True and "workload" in d["config"]
    assert d["value"] > 0 and abs(d["value"] - {"cassie_ppo": 64 * 2, "cassietraj_recurrent": 64 * 3, "cassie_td3": 64 * 2}[workload] / (d["ms_per_step"] * 1e-3)) < 0.02 * d["value"]
    if workload == "cassie_ppo":
        r = d["roofline"]
        assert r["kernel"] == "env_rollout_kernel" and r["bound"] == "valu" and r["launches_timed"] == 1 and r["frac"] > 0 and "hbm" in r and "mlp_forward_mfma" in r
        c = d["cpu_baseline"]
        assert c["kind"] == "port" and c["value"] > 0 and c["cores"] >= 1 and "sample" in c and c["end_to_end"]["value"] > 0
        assert d["optimiser_step_us"] > 0 and d["config"]["optimiser_steps_as_one_launch_per_epoch"] is False
    if workload == "cassietraj_recurrent":
        assert d["optimiser_step_us"] > 0 and d["epochs_run_per_step"] == 1.0


G24_CELL_SETS = {"a": [(0, 7), (20, 0), (30, 14), (40, 21), (50, 7), (60, 14), (80, 0), (90, 21)],
                 "b": [(10, 3), (10, 17), (30, 5), (30, 24), (50, 12), (50, 26), (70, 9), (70, 19)],
                 "c": [(0, 0), (0, 20), (20, 10), (20, 25), (40, 4), (40, 13), (60, 2), (60, 22)],
                 "d": [(80, 8), (80, 16), (90, 1), (90, 11), (10, 27), (30, 18), (50, 20), (70, 23)],
                 "e": [(0, 14), (20, 18), (40, 9), (60, 6), (80, 23), (90, 15), (10, 10), (70, 2)]}


@full
@pytest.mark.parametrize("cellset", sorted(G24_CELL_SETS))
def test_emulated_g24_push_cells_of_the_reference_policy_on_the_kernel_sources(dev, golden_dir, cellset):
    """Golden G24 on the KERNEL SOURCES: the reference's shipped, MuJoCo-trained policy on the emulated env kernel (49-entry observation of its Cassie-v0 revision, simrate 60,
    through apx_env_step / apply_force like apex_amd.eval.compute_perturbs), 8 (direction, phase) cells of the oracle's 280-cell lattice x 8 push sizes bracketing the
    oracle's result = 64 trials in lock step, the reference's protocol (tools/eval_perturb.py:36-85: two gait cycles, push for 0.2 s, survive 3 s).  The largest push survived
    per cell on the kernel sources against the fp64 oracle's (same physics, fp32, other arithmetic: the boundary is chaotic, one 10 N step either way is the resolution) and
    against MuJoCo's table."""
    import math
    from apex_amd.vecenv import CassieVecEnv
    from test_gpu_zz_first_hardware_run import _RefPolicy49
    g = np.load(os.path.join(golden_dir, "g24_ref_policy_push_sweep.npz"))
    lat = np.load(os.path.join(golden_dir, "g24_oracle_lattice_280.npz"))
    simrate, speed, wait, dur, first, incr = (float(x) for x in g["protocol"])
    dirs, phases, table = lat["directions"].astype(int), lat["phases"].astype(int), lat["oracle"].astype(np.float64)
    cells = G24_CELL_SETS[cellset]
    orc = np.array([table[list(dirs).index(a), list(phases).index(p)] for a, p in cells])
    muj = np.array([g["a_eval_perturbs"][a, p] for a, p in cells], dtype=np.float64)
    offs = np.arange(-4, 4) * incr                                    # sizes oracle - 40 .. oracle + 30
    n = 64
    a_i = np.repeat(np.arange(8), 8); s_i = np.tile(np.arange(8), 8)
    angle = -2.0 * math.pi * np.array([cells[i][0] for i in a_i]) / 100.0
    p_i = torch.tensor([cells[i][1] for i in a_i])
    size = np.maximum(orc[a_i] + offs[s_i], first)
    wrench = torch.zeros(n, 6); wrench[:, 0] = torch.tensor(size * np.cos(angle), dtype=torch.float32); wrench[:, 1] = torch.tensor(size * np.sin(angle), dtype=torch.float32)
    env = CassieVecEnv(n_envs=n, simrate=int(simrate), dynamics_randomization=False, seed=0, max_traj_len=100000)
    pol = _RefPolicy49(g, "a", dev, speed)
    dt = env.simrate * 0.0005
    n_push = int(math.ceil(dur / dt - 1e-9)); n_wait = int(math.ceil(wait / dt - 1e-9))
    obs = env.reset_for_test(full_reset=True)
    env.set_command(speed=speed)
    for _ in range(2 * 28):
        obs, _, _, _ = env.step(pol(obs), auto_reset=False)
    fell = torch.zeros(n, dtype=torch.bool); zero = torch.zeros_like(wrench)
    for k in range(27 + n_push + n_wait):
        pushing = (k >= p_i) & (k < p_i + n_push)
        env.apply_force(torch.where(pushing.view(n, 1), wrench, zero), "cassie-pelvis")
        obs, _, _, _ = env.step(pol(obs), auto_reset=False)
        waiting = (k >= p_i + n_push) & (k < p_i + n_push + n_wait)
        fell |= waiting & (env.get_field("qpos")[:, 2] < 0.4)
    fell = fell.view(8, 8).numpy()
    firstfail = np.where(fell.any(1), fell.argmax(1), 8)
    mine = np.array([size.reshape(8, 8)[c, min(f, 7)] - (incr if f < 8 else 0.0) for c, f in enumerate(firstfail)])      # (a cell that survives every size of its bracket reports the largest tried)
    print("G24CELLS", cellset, cells, "kernel", mine.astype(int).tolist(), "oracle", orc.astype(int).tolist(), "mujoco", muj.astype(int).tolist())
    d = np.abs(mine - orc)
    assert (d <= incr).sum() >= 6 and d.max() <= 40.0, (mine, orc)
    assert abs(mine.mean() - muj.mean()) < 0.12 * muj.mean() and np.corrcoef(mine, muj)[0, 1] > 0.8
    env.close()


@full
def test_emulated_g24_the_reference_policy_walks_on_the_kernel_sources(dev, golden_dir):
    """the body of tests/test_gpu_zz_first_hardware_run.py::test_g24_the_reference_policy_walks_on_the_kernel at 0.5 m/s on the emulated sources: sim-to-sim transfer of the
    reference's MuJoCo-trained policy through step_basic at simrate 60, 64 envs, 200 policy steps - nobody falls, walking height, commanded speed tracked"""
    import test_gpu_zz_first_hardware_run as Z
    Z.test_g24_the_reference_policy_walks_on_the_kernel(dev, golden_dir, 0.5, 0.10)


@full
@pytest.mark.parametrize("mode", ["golden", "twin", "ppo"])
def test_emulated_epoch_worker_modes(dev, monkeypatch, mode):
    """tests/epoch_worker.py - the script the GPU suite runs in a child process for the persistent one-launch trainers - with its own `golden`, `twin` and `ppo` bodies on the
    emulated kernels (one workgroup for the cooperative launch): `ppo` is PPO.update with the epoch kernel on / off on the same rollout of the (emulated) env"""
    import epoch_worker as W
    monkeypatch.setattr(W, "FAILED", [])
    getattr(W, mode)(dev)
    assert W.FAILED == [], W.FAILED
