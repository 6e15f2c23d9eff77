"""GPU: the batched PPO driver end to end on the HIP env (rollout grids, truncation bootstrap, returns, update)."""
import numpy as np
from tests.test_oracle_learner import ACTOR_KEYS
import pytest
import torch

from oracle import learner as OL

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def _mk(graph=False):
    from apex_amd.vecenv import CassieVecEnv
    from apex_amd.ppo import PPO
    N, T, mtl = 256, 24, 10
    env = CassieVecEnv(n_envs=N, seed=2, max_traj_len=mtl)
    args = dict(gamma=0.99, lam=0.95, lr=1e-4, eps=1e-5, entropy_coeff=0.0, clip=0.2, minibatch_size=1024, epochs=2,
                num_steps=T * N, max_traj_len=mtl, max_grad_norm=0.05, mirror=True, std_dev=-1.5, seed=0, graph=graph)
    algo = PPO(args, "/tmp/apx_test_unused", env, rank=0, world_size=1, group=None)
    algo.init_networks(0)
    algo.normalization_params(2000)
    return algo, N, T, mtl


@pytest.mark.parametrize("graph", [False, True])
def test_iteration_grids_bootstrap_and_returns(dev, graph):
    """max_traj_len = 10 forces time-limit truncations (done = 2) in every rollout: the bootstrap value must be the critic
    on the recorded final observation exactly there and 0 elsewhere, the returns must equal the oracle's scan over the
    recorded grids, and the episode statistics must respect the limit."""
    algo, N, T, mtl = _mk(graph)
    for it in range(2):
        ret, ep_rets, ep_lens = algo.sample()
        torch.cuda.synchronize()
        done = algo.b_done.cpu().numpy(); end = algo.b_end.cpu().numpy(); boot = algo.b_boot.cpu().numpy()
        assert np.array_equal(end != 0, done != 0) and set(np.unique(done)) <= {0, 1, 2}
        assert (done == 2).sum() > 0
        vfin = algo.learner.critic.forward(algo.b_fin.view(T * N, 50)).view(T, N).cpu().numpy()
        np.testing.assert_allclose(boot[done == 2], vfin[done == 2], rtol=1e-5, atol=1e-6)
        assert np.all(boot[done != 2] == 0)
        last_val = algo.learner.critic.forward(algo.obs).view(-1).cpu().numpy()
        ref = OL.returns_scan_grid_boot(algo.b_rew.cpu().numpy(), end, boot, last_val, 0.99)
        np.testing.assert_allclose(ret.cpu().numpy(), ref, rtol=1e-6, atol=1e-6)
        # the grids are one consistent trajectory: obs[t+1] is what the env returned for step t (reset obs where done)
        assert torch.isfinite(algo.b_obs).all() and torch.isfinite(algo.b_act).all() and torch.isfinite(algo.b_rew).all()
        np.testing.assert_allclose(algo.b_act.cpu().numpy() - algo.b_mu.cpu().numpy(), (algo.b_act - algo.b_mu).cpu().numpy())
        el = ep_lens.cpu().numpy()
        assert el.size == int((done != 0).sum()) and el.max() <= mtl and el.min() >= 1
        losses, kl, epochs_run = algo.update(ret)
        assert np.all(np.isfinite(losses)) and np.isfinite(kl) and epochs_run >= 1


def test_batched_deterministic_evaluation(dev):
    """apex_amd.eval.evaluate: one deterministic episode per env from reset_for_test at a commanded speed."""
    from apex_amd.eval import evaluate
    algo, N, T, mtl = _mk(False)
    out = evaluate(algo.learner.actor, algo.env, algo.learner.obs_mean, algo.learner.obs_std, speed=1.0)
    ln = out["lengths"].cpu().numpy(); rt = out["returns"].cpu().numpy()
    assert ln.min() >= 1 and ln.max() <= mtl and np.all(np.isfinite(rt)) and np.all(rt >= 0)
    assert bool((out["terminated"] ^ out["truncated"]).all())            # every env finished exactly one way
    # default dynamics, one deterministic policy, one command: the episodes differ only through what reset_for_test keeps
    # (stale pd targets, delay line, encoder filters), so the returns are close but not identical
    assert np.ptp(rt) < 0.5 * max(1.0, abs(rt).max())
    ob = evaluate(algo.learner.actor, algo.env, algo.learner.obs_mean, algo.learner.obs_std, speed=1.0, basic=True, max_steps=60)
    lb = ob["lengths"].cpu().numpy()
    assert lb.min() >= 1 and lb.max() <= 60 and bool((ob["terminated"] ^ ob["truncated"]).all())


def test_cli_eval_scores_a_saved_checkpoint(dev, tmp_path, capsys):
    """`apex.py eval --path RUN_DIR`: loads the whole-module actor.pt pickle the trainer writes and scores it on the batch."""
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import apex
    algo, N, T, mtl = _mk(False)
    algo.save_path = str(tmp_path)
    algo.save()
    rc = apex.main(["eval", "--path", str(tmp_path), "--n_envs", "64", "--speed", "0.5", "--max_traj_len", "40"])
    out = capsys.readouterr().out
    assert rc == 0 and "episodes 64" in out and "mean length" in out


class _ToyVecEnv:
    """The toy dynamics of tools/refprobe/gen_golden_train.py (G15b) as a one-column device env with the CassieVecEnv
    surface the driver uses: reset(), step(act, out=(next_obs, rew, done, final_obs)), done codes 1 = terminated,
    2 = time-limit truncation.  `ks` is the list of episode counters the reference consumed for this iteration."""
    def __init__(self, dev, lens, max_traj_len):
        self.device, self.n_envs, self.lens, self.mtl = dev, 1, [int(x) for x in lens], int(max_traj_len)
        self.ks = []

    def _start(self):
        k = self.ks.pop(0) if self.ks else 1
        self.t = 0; self.L = self.lens[(k - 1) % len(self.lens)]
        self.x = torch.cos(torch.arange(50, dtype=torch.float64, device=self.device) * 0.1 * k)

    def _obs(self):
        o = self.x.clone(); o[46] = np.sin(0.2 * self.t); o[47] = np.cos(0.2 * self.t)
        return o.float().view(1, 50)

    def reset(self):
        self._start()
        return self._obs()

    def step(self, act, out):
        nxt, rew, done, fin = out
        self.t += 1
        self.x = 0.9 * self.x + 0.1 * act.double().view(10).repeat(5) + 0.01
        rew.copy_(torch.exp(-self.x.abs().mean()).float().view(1))
        fin.copy_(self._obs())
        code = 1 if self.t >= self.L else (2 if self.t >= self.mtl else 0)       # done wins over the time limit (ppo.py:174,184)
        done.fill_(code)
        if code:
            self._start()
        nxt.copy_(self._obs())


def test_kl_early_stop_golden_g15c(dev, golden_dir):
    """G15c: the same whole-loop replay with lr = 1.2e-2 and 64-unit nets (unfused GEMM path): the reference stops iteration 0 after
    ONE epoch (KL of the last minibatch 0.046 > 0.02, ppo.py:449) and runs all three epochs of iteration 1 (0.010, 0.008 below the
    threshold).  The build must take the same decisions; the scalars agree to a few percent (each Adam step moves a weight by
    +-1.2e-2, so a sign tie at g ~ 0 is visible at this learning rate)."""
    _replay_train_golden(dev, golden_dir, "g15c_ppo_train_earlystop.npz", strict=False)


def test_whole_train_loop_golden_g15b(dev, golden_dir):
    """G15b: the reference's whole PPO.train (3 iterations: sample -> returns -> normalised advantages -> 3 epochs x 7 random
    minibatches with the mirror loss -> Adam/clip) on a toy env, replayed through the build's driver + HIP learner with the
    captured noise / minibatch-order streams.  Episode indices bit-exact; values, returns, advantages' inputs, every
    minibatch's six scalars and the parameters after each iteration within the north-star tolerance."""
    _replay_train_golden(dev, golden_dir, "g15b_ppo_train.npz")


def _replay_train_golden(dev, golden_dir, fname, strict=True):
    import os
    from apex_amd.ppo import PPO
    from rl.policies.actor import Gaussian_FF_Actor
    from rl.policies.critic import FF_V
    g = np.load(os.path.join(golden_dir, fname))
    H, mb = int(g["hidden"]), int(g["minibatch"])
    env = _ToyVecEnv(dev, g["lens"], g["max_traj_len"])
    args = dict(gamma=float(g["gamma"]), lam=0.95, lr=float(g["lr"]) if "lr" in g.files else 1e-4, eps=1e-5, entropy_coeff=0.0, clip=0.2, minibatch_size=mb,
                epochs=int(g["epochs"]), num_steps=int(g["num_steps"]), max_traj_len=int(g["max_traj_len"]), max_grad_norm=0.05,
                mirror=True, std_dev=-1.5, seed=0)
    algo = PPO(args, "/tmp/apx_test_unused", env, rank=0, world_size=1, group=None, hidden=H)
    algo.policy = Gaussian_FF_Actor(50, 10, layers=(H, H), fixed_std=np.exp(-1.5)); algo.critic = FF_V(50, layers=(H, H))
    algo.policy.load_state_dict({k: torch.as_tensor(g["actor0." + k]) for k in algo.policy.state_dict()})
    algo.critic.load_state_dict({k: torch.as_tensor(g["critic0." + k]) for k in algo.critic.state_dict()})
    algo.policy.obs_mean = torch.as_tensor(g["obs_mean"]); algo.policy.obs_std = torch.as_tensor(g["obs_std"])
    algo.upload()
    sigma = float(np.exp(-1.5))
    for it in range(int(g["n_itr"])):
        p = "it%d." % it
        B = len(g[p + "rewards"])
        assert B == algo.T
        n_ep = len(g[p + "ep_lens"])
        env.ks = [int(g[p + "k0"]) + 1 + j for j in range(n_ep)]
        noise = torch.as_tensor((g[p + "actions"].astype(np.float64) - g[p + "mu"]) / sigma, dtype=torch.float32, device=dev)
        algo.noise_fn = lambda t, out: out.copy_(noise[t].view(1, 10))
        perms = torch.as_tensor(g[p + "idx"], device=dev)
        algo.perm_fn = lambda e: perms[e]
        algo.trace = []
        algo.obs = None; algo.ep_ret.zero_(); algo.ep_len.zero_()
        ret, ep_rets, ep_lens = algo.sample()
        loose = (not strict) and it > 0          # after a high-learning-rate iteration the two parameter sets differ visibly
        tol = (lambda r, a: dict(rtol=5e-2, atol=5e-2)) if loose else (lambda r, a: dict(rtol=r, atol=a))
        # --- sample: episode-step indices bit-exact, data within fp32 round-off of the reference's buffer
        ends = np.nonzero(algo.b_end.view(-1).cpu().numpy())[0] + 1
        assert np.array_equal(ends, g[p + "traj_idx"][1:])
        assert np.array_equal(ep_lens.cpu().numpy().astype(np.int64), g[p + "ep_lens"])
        np.testing.assert_allclose(ep_rets.cpu().numpy(), g[p + "ep_returns"], **tol(2e-6, 0))
        np.testing.assert_allclose(algo.b_obs.view(B, 50).cpu().numpy(), g[p + "states"], **tol(1e-5, 2e-6))
        np.testing.assert_allclose(algo.b_act.view(B, 10).cpu().numpy(), g[p + "actions"], **tol(1e-5, 5e-6))
        np.testing.assert_allclose(algo.b_rew.view(B).cpu().numpy(), g[p + "rewards"], **tol(2e-6, 0))
        np.testing.assert_allclose(algo.b_val.view(B).cpu().numpy(), g[p + "values"], **tol(1e-5, 1e-5))
        np.testing.assert_allclose(ret.view(B).cpu().numpy(), g[p + "returns"], **(dict(rtol=5e-2, atol=0.5) if loose else dict(rtol=1e-5, atol=1e-5)))
        # --- optimise: every minibatch's (actor loss, entropy, critic loss, ratio, kl, mirror loss)
        losses, kl, epochs_run = algo.update(ret)
        assert epochs_run == int(g[p + "epochs_run"])
        scal = torch.stack(algo.trace).cpu().numpy().reshape(epochs_run, -1, 6)
        ref = g[p + "scal"]
        assert scal.shape == ref.shape
        stol = [(1e-5, 2e-6), (1e-6, 0), (1e-5, 0), (1e-5, 0), (1e-3, 2e-8), (1e-4, 1e-9)] if strict else [(0.2, 0.02), (1e-6, 0), (0.1, 0), (0.02, 0), (0.25, 2e-3), (0.25, 2e-4)]
        for c, (rt, at) in enumerate(stol):
            np.testing.assert_allclose(scal[..., c], ref[..., c], rtol=rt, atol=at, err_msg="scalar %d itr %d" % (c, it))
        # --- parameters after the iteration (21 Adam steps each)
        for nm, views, ref_p in (("actor", algo.learner.actor.views(), algo.policy), ("critic", algo.learner.critic.views(), algo.critic)):
            for k, v in zip(ref_p.state_dict(), views):
                d = np.abs(v.cpu().numpy() - g[p + nm + "." + k])
                if strict:
                    assert (d > 5e-6).mean() < 2e-3 and d.max() < 5e-4, (it, nm, k, (d > 5e-6).mean(), d.max())
                else:
                    assert d.mean() < 4e-3, (it, nm, k, d.mean())


def test_compute_perturbs_batched(dev):
    """tools/eval_perturb.py's sweep as one batch (apex_amd/eval.py::compute_perturbs) with the PD-hold "policy" (zero action =
    nominal standing pose, cassie.py:107,295; it keeps the pelvis above 0.4 m for ~1 s, enough for a short window): no push ->
    nobody falls; 1250 N for 0.1 s -> every trial falls; failures are monotone in the push size; the reported max force is
    (first failing size - increment) per (phase, direction)."""
    from apex_amd.vecenv import CassieVecEnv
    from apex_amd.eval import compute_perturbs
    mk = lambda n: CassieVecEnv(n_envs=n, max_traj_len=100000, dynamics_randomization=False)
    stand = lambda o: torch.zeros(o.shape[0], 10, device=o.device)
    mf, fell = compute_perturbs(stand, mk, wait_time=0.3, perturb_duration=0.1, perturb_size=0.0, perturb_incr=250.0, num_angles=4,
                                n_sizes=6, num_phases=2)
    assert mf.shape == (2, 4) and fell.shape == (2, 4, 6)
    assert not fell[:, :, 0].any()                   # 0 N
    assert fell[:, :, 5].all()                       # 1250 N
    assert (np.diff(fell.astype(int), axis=-1) >= 0).all()
    first = np.where(fell.any(-1), fell.argmax(-1), 6)
    np.testing.assert_allclose(mf, 250.0 * first - 250.0)
    assert (mf >= 250).all() and (mf <= 1000).all()


def test_eval_commands_batched(dev):
    """tools/test_commands.py's schedule test as one batch.  (a) the harness-side yaw rotation of the policy input equals the env's own
    orient_add rotation of the pelvis quaternion / velocity entries (cassie.py:280-291,822-835); (b) bookkeeping: with the PD-hold
    "policy" every schedule fails within the first half period, and the failure rows carry speed 0.5, no orientation change."""
    from apex_amd.vecenv import CassieVecEnv
    from apex_amd.eval import eval_commands, _yaw_unrotate_obs
    env = CassieVecEnv(n_envs=64, max_traj_len=1000, dynamics_randomization=False, seed=3)
    env.reset()
    act = torch.randn(64, 10, device=dev) * 0.2
    for _ in range(3):
        env.step(act, auto_reset=False)
    yaw = torch.linspace(-1.2, 1.2, 64, device=dev)
    st = {k: env.get_field(k) for k in ("qpos", "qvel")}
    env.set_command(orient_add=torch.zeros(64, device=dev))
    o0, _, _, _ = env.step(act, auto_reset=False); o0 = o0.clone()
    env.set_field("qpos", st["qpos"]); env.set_field("qvel", st["qvel"])          # (the estimator inputs are a function of the state)
    rot = _yaw_unrotate_obs(o0, yaw)
    from oracle import sim as OS
    e = OS.OracleEnv(dyn_rand=False)
    for i in (0, 13, 40, 63):                                                       # the oracle's rotate_to_orient is pinned by golden G8
        e.set("orient_add", [float(yaw[i])]); e.set("so_quat", o0[i, 1:5].double().cpu().numpy()); e.set("so_tvel", o0[i, 15:18].double().cpu().numpy())
        ob = e.obs()
        np.testing.assert_allclose(rot[i, 1:5].cpu().numpy(), ob[1:5], atol=2e-6)
        np.testing.assert_allclose(rot[i, 15:18].cpu().numpy(), ob[15:18], atol=2e-6)
    mk = lambda n: CassieVecEnv(n_envs=n, max_traj_len=100000, dynamics_randomization=False)
    stand = lambda o: torch.zeros(o.shape[0], 10, device=o.device)
    d = eval_commands(stand, mk, num_steps=200, num_commands=3, num_iters=64)
    assert d.shape == (64, 6) and (d[:, 0] == 0).all() and (d[:, 1] == 0).all()
    np.testing.assert_allclose(d[:, 2], 0.5); assert (d[:, 3] == 0).all() and (d[:, 5] == 0).all()


def test_cli_ppo_run_directory_checkpoints_and_scalars(dev, tmp_path, golden_dir):
    """BASELINE configs[0] analogue on the GPU: `apex.py ppo --env_name Cassie-v0 --reward clock ...` end to end for 2 iterations:
    run directory <logdir>/Cassie-v0/<md5[:6]>-seed0 with experiment.info / experiment.pkl (util/log.py:11-70), whole-module
    actor.pt / critic.pt that load with `torch.load` and this repo's rl.policies classes (ppo.py:129-137), and the reference's 13
    scalar names (ppo.py:476-491; the golden G15b run emitted exactly these)."""
    import json, os, sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import apex
    rc = apex.main(["ppo", "--env_name", "Cassie-v0", "--reward", "clock", "--n_envs", "256", "--num_steps", "4096", "--minibatch_size", "1024",
                    "--n_itr", "2", "--input_norm_steps", "512", "--logdir", str(tmp_path), "--seed", "0"])
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
    if os.path.exists(os.path.join(run, "scalars.jsonl")):          # no TensorBoard in this image: same tags, one JSON object per line
        names = set(json.loads(line)["tag"] for line in open(os.path.join(run, "scalars.jsonl")))
        assert names == set(str(x) for x in g["scalar_names"])
    else:
        assert any(f.startswith("events.out.tfevents") for f in os.listdir(run))


def test_recurrent_ppo_iteration_on_the_hip_env(dev, tmp_path):
    """Row f1: recurrent PPO (LSTM 2 x 128 actor / critic) on the HIP env.  Rollout grids -> whole trajectories (cut at the grid end) ->
    padded [T_max, B] minibatches with the layout of torch's pad_sequence -> update; a short horizon forces time-limit truncations,
    whose bootstrap uses the critic's carried hidden state; the checkpoint is a whole-module pickle of Gaussian_LSTM_Actor."""
    from apex_amd.vecenv import CassieVecEnv
    from apex_amd.ppo_recurrent import RecurrentPPO
    N, T, mtl = 64, 30, 12
    env = CassieVecEnv(n_envs=N, seed=5, max_traj_len=mtl)
    args = dict(gamma=0.99, lam=0.95, lr=1e-4, eps=1e-5, entropy_coeff=0.0, clip=0.2, minibatch_size=32, epochs=2, num_steps=T * N, max_traj_len=mtl,
                max_grad_norm=0.05, mirror=True, seed=0)
    algo = RecurrentPPO(args, str(tmp_path), env)
    algo.init_networks(0)
    algo.normalization_params(1000)
    p0 = algo.learner.actor.params.clone()
    ret = algo.sample()
    done = algo.b_done.cpu().numpy(); boot = algo.b_boot.cpu().numpy(); end = algo.b_end.cpu().numpy()
    assert (done == 2).sum() > 0 and np.all(end[-1] == 1) and np.all(boot[(done == 1)] == 0) and np.all(boot[done == 2] != 0)
    trajs = algo.trajectories()
    lens = trajs[:, 2] - trajs[:, 1]
    assert lens.sum() == T * N and lens.max() <= mtl and (lens > 0).all()
    # the padded index equals pad_sequence of the per-trajectory index lists
    sel = trajs[[0, 5, 9, len(trajs) - 1]]
    idx = algo.padded_index(sel).cpu()
    ref = torch.nn.utils.rnn.pad_sequence([torch.arange(t0, t1) * N + n for n, t0, t1 in sel], batch_first=False, padding_value=-1)
    assert torch.equal(idx, ref)
    # returns: bootstrap at truncations / the grid end, zero at terminations (oracle scan on the recorded grids)
    r = OL.returns_scan_grid_boot(algo.b_rew.cpu().numpy(), end, boot, np.zeros(N), 0.99)
    np.testing.assert_allclose(ret.cpu().numpy(), r, rtol=1e-6, atol=1e-6)
    losses, kl, epochs_run = algo.update(ret)
    assert np.all(np.isfinite(losses)) and epochs_run >= 1
    assert (algo.learner.actor.params - p0).abs().max() > 1e-5
    out = algo.iteration()
    assert np.isfinite(out["losses"]).all() and out["ep_lens"].numel() > 0
    algo.save()
    pol = torch.load(str(tmp_path / "actor.pt"), weights_only=False)
    assert type(pol).__name__ == "Gaussian_LSTM_Actor" and pol.is_recurrent
    pol.init_hidden_state()
    assert pol(torch.zeros(50), deterministic=True).shape[-1] == 10
    # the recurrent checkpoint is scored by the same CLI: one carried (h, c) per env through the HIP LSTM
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import apex
    assert apex.main(["eval", "--path", str(tmp_path), "--n_envs", "64", "--speed", "0.5", "--max_traj_len", "30"]) == 0
    from apex_amd.eval import RecurrentActor
    ra = RecurrentActor(algo.learner.actor, algo.learner.obs_mean, algo.learner.obs_std)
    o = torch.randn(5, 64, 50, device=dev)
    steps = torch.stack([ra(o[t]) for t in range(5)])
    seq = algo.learner.actor.forward(((o - algo.learner.obs_mean) / algo.learner.obs_std).contiguous())
    np.testing.assert_allclose(steps.cpu().numpy(), seq.cpu().numpy(), rtol=1e-5, atol=1e-6)       # stepping == the sequence pass from zero state


@pytest.mark.parametrize("fname", ["g15d_ppo_train_recurrent.npz", "g15e_ppo_train_recurrent_h128.npz"])
def test_whole_train_loop_recurrent_golden_g15d(dev, golden_dir, fname):
    """G15d: the reference's whole PPO.train in RECURRENT mode (LSTM 2 x 32; G15e: the 2 x 128 of BASELINE configs[3]; 2 iterations, 16
    trajectories per batch, minibatches of 4 whole trajectories padded by pad_sequence, mirror loss, truncation bootstrap with the
    critic's carried hidden state) on the toy env, replayed through apex_amd.ppo_recurrent.RecurrentPPO with the captured noise /
    trajectory-order streams."""
    import os
    from apex_amd.ppo_recurrent import RecurrentPPO
    from rl.policies.actor import Gaussian_LSTM_Actor
    from rl.policies.critic import LSTM_V
    from golden_util import check_slim, seeded_params
    g = np.load(os.path.join(golden_dir, fname))
    big = "actor_seed" in g.files
    H, mb = int(g["hidden"]), int(g["minibatch"])
    env = _ToyVecEnv(dev, g["lens"], g["max_traj_len"])
    args = dict(gamma=float(g["gamma"]), lam=0.95, lr=float(g["lr"]), eps=1e-5, entropy_coeff=0.0, clip=0.2, minibatch_size=mb, epochs=int(g["epochs"]),
                num_steps=int(g["num_steps"]), max_traj_len=int(g["max_traj_len"]), max_grad_norm=0.05, mirror=True, seed=0,
                std_dev=float(np.log(g["fixed_std"])))
    algo = RecurrentPPO(args, "/tmp/apx_test_unused", env, hidden=H, layers=2)
    algo.policy = Gaussian_LSTM_Actor(50, 10, layers=(H, H), fixed_std=float(g["fixed_std"])); algo.critic = LSTM_V(50, layers=(H, H))
    if big:
        for net, which in ((algo.policy, "actor"), (algo.critic, "critic")):
            sd = net.state_dict()
            assert list(sd.keys()) == [str(k) for k in g[which + "_keys"]]
            net.load_state_dict({k: torch.as_tensor(w) for k, w in zip(sd.keys(), seeded_params([v.shape for v in sd.values()], int(g[which + "_seed"])))})
    else:
        algo.policy.load_state_dict({k: torch.as_tensor(g["actor0." + k]) for k in algo.policy.state_dict()})
        algo.critic.load_state_dict({k: torch.as_tensor(g["critic0." + k]) for k in algo.critic.state_dict()})
    algo.policy.obs_mean = torch.as_tensor(g["obs_mean"]); algo.policy.obs_std = torch.as_tensor(g["obs_std"])
    algo.upload()
    sigma = float(g["fixed_std"])
    for it in range(int(g["n_itr"])):
        p = "it%d." % it
        B = len(g[p + "rewards"])
        assert B == algo.T
        env.ks = [int(g[p + "k0"]) + 1 + j for j in range(len(g[p + "ep_lens"]))]
        noise = torch.as_tensor((g[p + "actions"].astype(np.float64) - g[p + "mu"]) / sigma, dtype=torch.float32, device=dev)
        algo.noise_fn = lambda t, out: out.copy_(noise[t].view(1, 10))
        order = g[p + "idx"]
        algo.perm_fn = lambda e: order[e]
        algo.trace = []
        ret = algo.sample()
        ends = np.nonzero(algo.b_end.view(-1).cpu().numpy())[0] + 1
        assert np.array_equal(ends, g[p + "traj_idx"][1:])                                  # bit-exact episode-step indices
        assert np.array_equal(algo.trajectories()[:, 2] - algo.trajectories()[:, 1], g[p + "ep_lens"])
        np.testing.assert_allclose(algo.b_obs.view(B, 50).cpu().numpy(), g[p + "states"], rtol=1e-5, atol=3e-6)
        np.testing.assert_allclose(algo.b_act.view(B, 10).cpu().numpy(), g[p + "actions"], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(algo.b_val.view(B).cpu().numpy(), g[p + "values"], rtol=1e-4, atol=2e-5)      # hidden state carried per episode
        np.testing.assert_allclose(ret.view(B).cpu().numpy(), g[p + "returns"], rtol=1e-4, atol=2e-5)           # incl. carried-state bootstrap
        losses, kl, epochs_run = algo.update(ret)
        assert epochs_run == int(g[p + "epochs_run"])
        scal = torch.stack(algo.trace).cpu().numpy().reshape(epochs_run, -1, 6)
        ref = g[p + "scal"]
        assert scal.shape == ref.shape
        for c, (rt, at) in enumerate([(2e-4, 5e-6), (1e-6, 0), (2e-4, 0), (1e-4, 0), (5e-3, 1e-7), (1e-3, 1e-8)]):
            np.testing.assert_allclose(scal[..., c], ref[..., c], rtol=rt, atol=at, err_msg="scalar %d itr %d" % (c, it))
        for nm, views, ref_p in (("actor", algo.learner.actor.views(), algo.policy), ("critic", algo.learner.critic.views(), algo.critic)):
            for k, v in zip(ref_p.state_dict(), views):
                if big:
                    check_slim(v.cpu().numpy(), g[p + nm + "." + k], atol=5e-4, frac_tol=5e-6, frac=2e-2, err_msg="%d %s %s" % (it, nm, k))
                    continue
                d = np.abs(v.cpu().numpy() - g[p + nm + "." + k])
                assert (d > 5e-6).mean() < 1e-2 and d.max() < 5e-4, (it, nm, k, (d > 5e-6).mean(), d.max())


class _ToyColumns:
    """N independent toy envs of tools/refprobe/gen_golden_normparams.py (G21) as one lock-step device env: column w is the env
    worker w of the reference builds (episode counter starts at 100 w)."""
    def __init__(self, dev, n, lens):
        self.device, self.n_envs, self.lens = dev, n, torch.tensor([int(x) for x in lens], device=dev)
        self.k = 100 * torch.arange(n, device=dev)
        self.x = torch.zeros(n, 50, dtype=torch.float64, device=dev)
        self.t = torch.zeros(n, dtype=torch.long, device=dev); self.L = torch.zeros(n, dtype=torch.long, device=dev)

    def _start(self, m):
        self.k = torch.where(m, self.k + 1, self.k)
        self.t = torch.where(m, torch.zeros_like(self.t), self.t)
        self.L = torch.where(m, self.lens[(self.k - 1) % len(self.lens)], self.L)
        fresh = torch.cos(torch.arange(50, dtype=torch.float64, device=self.device).view(1, 50) * 0.1 * self.k.double().view(-1, 1))
        self.x = torch.where(m.view(-1, 1), fresh, self.x)

    def reset(self):
        self._start(torch.ones(self.n_envs, dtype=torch.bool, device=self.device))
        return self.x.float()

    def step(self, act):
        self.t = self.t + 1
        self.x = 0.9 * self.x + 0.1 * act.double().repeat(1, 5) + 0.01
        done = self.t >= self.L
        self._start(done)
        return self.x.float(), None, done, None


def test_normalization_params_golden_g21(dev, golden_dir):
    """Row a12, golden G21: the reference's get_normalization_params (rl/envs/normalize.py:11-48; 4 workers x 60 steps, actions =
    un-normalised policy(state) + N(0, 1), reset on done with the terminal state dropped) replayed through PPO.normalization_params
    with the captured noise: mean and sqrt(var + 1e-8) of the raw observations."""
    import os
    from apex_amd.ppo import PPO
    g = np.load(os.path.join(golden_dir, "g21_normalization_params.npz"))
    H, procs, iters = int(g["hidden"]), int(g["procs"]), int(g["iters"])
    env = _ToyColumns(dev, procs, g["lens"])
    args = dict(gamma=0.99, lam=0.95, lr=1e-4, eps=1e-5, entropy_coeff=0.0, clip=0.2, minibatch_size=64, epochs=3, num_steps=procs * 8, max_traj_len=400,
                max_grad_norm=0.05, mirror=True, std_dev=-1.5, seed=0)
    algo = PPO(args, "/tmp/apx_test_unused", env, hidden=H)
    algo.learner.actor.load_list([g["actor." + k] for k in ACTOR_KEYS])
    noise = torch.as_tensor(g["noise"], device=dev)                     # [procs, steps, 10]
    algo.noise_fn = lambda t, out: out.copy_(noise[:, t])
    algo.normalization_params(iters, noise_std=float(g["noise_std"]))
    np.testing.assert_allclose(algo.learner.obs_mean.cpu().numpy(), g["mean"], rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(algo.learner.obs_std.cpu().numpy(), g["std"], rtol=1e-5, atol=2e-6)


def test_td3_driver_hbm_replay_and_updates(dev, tmp_path):
    """Row f2: TD3 on the HIP env with the replay buffer in HBM.  Ring semantics (overwrite the oldest, remote_replay.py:70-74), the
    stored transitions (s' of a finished env is its FINAL observation, done_bool also at the time limit, sync_td3.py:82), uniform sampling
    with replacement, and a few collect + update rounds: finite statistics, moving weights, Polyak-averaged targets, checkpoint classes."""
    from apex_amd.vecenv import CassieVecEnv
    from apex_amd.td3 import TD3, HbmReplay
    rb = HbmReplay(100, 3, 2, dev)
    for k in range(3):
        n = 40
        base = torch.arange(n, device=dev).float() + 40 * k
        rb.add(base.view(n, 1).repeat(1, 3), base.view(n, 1).repeat(1, 3) + 0.5, base.view(n, 1).repeat(1, 2), base, torch.ones(n, device=dev))
    assert rb.size == 100 and rb.ptr == 20
    np.testing.assert_allclose(rb.r.cpu().numpy()[:20], np.arange(100, 120)); np.testing.assert_allclose(rb.r.cpu().numpy()[20:], np.arange(20, 100))
    g = torch.Generator(device=dev); g.manual_seed(0)
    s, s2, a, r, nd = rb.sample(4096, g)
    assert s.shape == (4096, 3) and torch.equal(s[:, 0], r) and torch.equal(s2[:, 0], r + 0.5) and len(torch.unique(r)) > 90      # with replacement, whole ring
    env = CassieVecEnv(n_envs=64, seed=3, max_traj_len=15)
    algo = TD3(env, str(tmp_path), hidden=64, batch_size=128, updates_per_step=2, replay_size=5000, seed=1)
    algo.init_networks(0)
    a0 = algo.learner.actor.params.clone(); t0 = algo.learner.actor_t.params.clone()
    out = algo.collect_and_train(20)
    assert out["updates"] > 20 and np.isfinite([out["q_loss"], out["avg_q1"], out["avg_q2"]]).all()
    assert algo.replay.size == 20 * 64 and (algo.replay.nd[:1280] == 0).sum() > 0           # episodes ended (height or the 15-step limit)
    assert (algo.replay.a[:1280].abs() <= 1).all()
    da = (algo.learner.actor.params - a0).abs().max(); dt = (algo.learner.actor_t.params - t0).abs().max()
    assert da > 1e-4 and 0 < dt < da                                                          # targets trail the live actor (tau = 0.005)
    ret, eplen = algo.evaluate(n_envs=64, max_traj_len=30)
    assert np.isfinite(ret) and 1 <= eplen <= 30
    algo.save()
    pol = torch.load(str(tmp_path / "actor.pt"), weights_only=False)
    assert type(pol).__name__ == "FF_Actor" and pol(torch.zeros(50)).abs().max() <= 1
    # parameter-space noise (sync_td3.py:113-128, rl/utils/param_noise.py): perturbed actor = actor + N(0, stddev); the stddev adapts towards the
    # desired action-space distance
    pn = TD3(env, str(tmp_path), hidden=64, batch_size=128, updates_per_step=1, replay_size=5000, seed=2, param_noise=True, noise_scale=0.3)
    pn.init_networks(0)
    s0 = pn.pn_std
    pn.collect_and_train(4)
    d = (pn.actor_perturbed.params - pn.learner.actor.params)
    assert 0.5 * s0 < float(d.std()) < 2.5 * s0 and pn.pn_std in (s0 * 1.05, s0 / 1.05)
    obs = pn.obs
    dist = pn.adapt_param_noise(obs)
    assert 0 < dist < 1


def test_td3_one_launch_collection(dev, tmp_path):
    """apx_rollout_td3 (round 5: TD3's collection phase as one env_rollout_kernel launch, policy fixed as in sync_td3.py's collect_experience) on the recorded grid: the
    actions are clip(max_action tanh(actor(obs)) + act_noise * scalar noise, -1, 1) of the learner's own forward on the recorded observations, the replay holds exactly the
    grid's transitions (terminal rows carry the episode's own final observation and not-done 0), and the update block ran steps x updates_per_step times."""
    from apex_amd.vecenv import CassieVecEnv
    from apex_amd.td3 import TD3
    N, T = 256, 12
    env = CassieVecEnv(n_envs=N, seed=4, max_traj_len=8)
    from apex_amd import engine
    algo = TD3(env, str(tmp_path), hidden=256, act_noise=0.3, batch_size=256, updates_per_step=2, replay_size=N * T, seed=1)
    algo.init_networks(0)
    p0 = algo.learner.actor.params.clone()
    env.kernel_timing(True); env.kernel_timing_read(reset=True)
    out = algo.collect_and_train(T)
    _, launches = env.kernel_timing_read(reset=True); env.kernel_timing(False)
    assert launches == 1 and out["updates"] == T * 2 and algo.replay.size == N * T
    g, noise = algo._g, algo._noise_last
    L = algo.learner
    pre = engine.Mlp(50, 256, 10, dev); pre.params.copy_(p0)
    m = torch.tanh(pre.forward(g["obs"].view(T * N, 50))).view(T, N, 10)
    # (TD3 feeds the RAW observation - no normalisation, sync_td3.py - so the first layer sums entries of O(10): the per-wave sequential-k fp32 sum and the MFMA sum of the
    # learner differ by round-off in proportion to the row's largest entry, where the normalised PPO input stays within 3e-6)
    d = (g["mu"] - m).abs()
    big = g["obs"].abs().amax(-1, keepdim=True)                                  # the row's largest observation entry (a falling robot's velocities: 10 - 50)
    assert float((d / (1.0 + big)).max()) < 2e-6 and float((d <= 5e-6).float().mean()) > 0.99, (float(d.max()), float((d / (1.0 + big)).max()))
    np.testing.assert_allclose(g["act"].cpu().numpy(), (g["mu"] + 0.3 * noise.view(T, N, 1)).clamp(-1, 1).cpu().numpy(), rtol=0, atol=1e-6)
    assert float(g["act"].abs().max()) <= 1.0 and int((g["done"] != 0).sum()) >= N      # max_traj_len 8 < T: every env restarted inside the launch
    R = algo.replay
    assert torch.equal(R.s[:N * T].view(T, N, 50), g["obs"]) and torch.equal(R.a[:N * T].view(T, N, 10), g["act"]) and torch.equal(R.r[:N * T].view(T, N), g["rew"])
    ended = g["done"] != 0
    assert torch.equal(R.nd[:N * T].view(T, N), (~ended).float())
    nxt = torch.cat([g["obs"][1:], g["nxt"].unsqueeze(0)])
    assert torch.equal(R.s2[:N * T].view(T, N, 50), torch.where(ended.unsqueeze(-1), g["fin"], nxt))
    assert not torch.equal(L.actor.params, p0) and torch.isfinite(L.actor.params).all()


def test_one_launch_recurrent_rollout_means_match_the_sequence_pass(dev, tmp_path):
    """apx_rollout_lstm (round 5: the recurrent rollout as ONE env_rollout_kernel launch, the two LSTM cells and the head evaluated per wave inside it, hidden state zeroed
    where an episode ends) against the learner's own sequence pass: on whole trajectories cut out of the recorded grid, the padded pass from the zero state reproduces the
    means the in-kernel actor produced step by step with its carried (h, c) - the identity the padded update relies on - and the recorded actions are those means plus the
    drawn noise.  Episodes end inside the call (max_traj_len 16 < T = 40, and falls), so the in-kernel hidden-state reset is on the path.  The critic's values over the
    stored grid equal a step-by-step replay with a fresh carried state."""
    from apex_amd.vecenv import CassieVecEnv
    from apex_amd.ppo_recurrent import RecurrentPPO
    N, T, mtl = 512, 40, 16
    env = CassieVecEnv(n_envs=N, seed=9, max_traj_len=mtl, env_name="CassieTraj-v0")
    args = dict(gamma=0.99, lam=0.95, lr=1e-4, eps=1e-5, entropy_coeff=0.0, clip=0.2, minibatch_size=512, epochs=1, num_steps=T * N, max_traj_len=mtl,
                max_grad_norm=0.05, mirror=True, seed=3, env_name="CassieTraj-v0")
    algo = RecurrentPPO(args, str(tmp_path), env, hidden=128, layers=2)
    algo.init_networks(0)
    algo.normalization_params(N * 50)
    L = algo.learner
    env.kernel_timing(True); env.kernel_timing_read(reset=True)
    ret = algo.sample()
    _, launches = env.kernel_timing_read(reset=True); env.kernel_timing(False)
    assert launches == 1                                     # the one-launch path ran (the per-step path times T env_step_kernel launches)
    noise = algo._noise_all
    done = algo.b_done.cpu().numpy()
    assert np.isfinite(algo.b_obs.cpu().numpy()).all() and np.isfinite(algo.b_rew.cpu().numpy()).all() and np.isfinite(ret.cpu().numpy()).all()
    assert (done == 2).sum() > 0 and (done != 0).sum() >= N * (T // mtl)
    np.testing.assert_allclose(algo.b_act.cpu().numpy(), (algo._b_mu + algo.fixed_std * noise).cpu().numpy(), rtol=0, atol=1e-6)
    trajs = algo.trajectories()
    sel = trajs[np.random.RandomState(0).permutation(len(trajs))[:128]]
    idx = algo.padded_index(sel)
    valid = idx >= 0
    gi = idx.clamp(min=0).view(-1)
    obs_p = (algo.b_obs.view(T * N, 50).index_select(0, gi) * valid.view(-1, 1)).view(idx.shape[0], 128, 50)
    mu_seq = L.actor.forward(((obs_p - L.obs_mean) / L.obs_std).contiguous())
    mu_roll = algo._b_mu.view(T * N, 10).index_select(0, gi).view(idx.shape[0], 128, 10)
    d = ((mu_seq - mu_roll) * valid.unsqueeze(-1)).abs().max()
    assert float(d) < 2e-5, float(d)
    # a trajectory that does NOT start at t = 0 is in the sample (its first step follows an in-kernel hidden-state reset)
    assert (sel[:, 1] > 0).any()
    # the critic's values: step-by-step replay over the grid
    hc = torch.zeros(2, 2, N, 128, device=dev)
    for t in range(T):
        v = torch.empty(N, 1, device=dev)
        L.critic.step(algo.b_obs[t], hc, reset=algo.b_done[t - 1] if t > 0 else None, y_out=v)
        assert torch.equal(v.view(-1), algo.b_val[t])


def test_full_size_config3_recurrent_iteration(dev, tmp_path):
    """BASELINE configs[3] at full size: CassieTraj-v0, 2048 envs, LSTM 2 x 128 actor / critic, one whole iteration (rollout -> returns ->
    padded whole-trajectory minibatches -> update).  Size-independent properties: the returns equal the oracle's scan on the recorded
    grids; every trajectory tiles its column; the padded index equals torch's pad_sequence; the sequence pass of the OLD policy from zero
    state reproduces the means the step-by-step rollout produced with its carried (h, c) (the identity the padded update relies on);
    a first minibatch has ratio 1 / KL 0; the update moves the weights and keeps everything finite."""
    from apex_amd.vecenv import CassieVecEnv
    from apex_amd.ppo_recurrent import RecurrentPPO
    N, T, mtl = 2048, 24, 16
    env = CassieVecEnv(n_envs=N, seed=9, max_traj_len=mtl, env_name="CassieTraj-v0")
    args = dict(gamma=0.99, lam=0.95, lr=1e-4, eps=1e-5, entropy_coeff=0.0, clip=0.2, minibatch_size=512, epochs=1, num_steps=T * N, max_traj_len=mtl,
                max_grad_norm=0.05, mirror=True, seed=3, env_name="CassieTraj-v0")
    algo = RecurrentPPO(args, str(tmp_path), env, hidden=128, layers=2)
    algo.init_networks(0)
    algo.normalization_params(N * 50)
    L = algo.learner
    assert L.actor.H == 128 and L.actor.L == 2 and algo.T == T
    noises = []
    algo.noise_fn = lambda t, out: (out.normal_(generator=algo.gen), noises.append(out.clone()))[0]
    ret = algo.sample()
    done = algo.b_done.cpu().numpy(); boot = algo.b_boot.cpu().numpy(); end = algo.b_end.cpu().numpy()
    assert np.isfinite(algo.b_obs.cpu().numpy()).all() and np.isfinite(algo.b_rew.cpu().numpy()).all()
    assert (done == 2).sum() > 0 and np.all(boot[done == 1] == 0)
    r = OL.returns_scan_grid_boot(algo.b_rew.cpu().numpy(), end, boot, np.zeros(N), 0.99)
    np.testing.assert_allclose(ret.cpu().numpy(), r, rtol=1e-6, atol=1e-6)
    trajs = algo.trajectories()
    lens = trajs[:, 2] - trajs[:, 1]
    assert lens.sum() == T * N and lens.max() <= mtl and lens.min() >= 1 and len(trajs) >= N * (T // mtl)
    sel = trajs[np.random.RandomState(0).permutation(len(trajs))[:64]]
    idx = algo.padded_index(sel)
    ref = torch.nn.utils.rnn.pad_sequence([torch.arange(t0, t1) * N + n for n, t0, t1 in sel], batch_first=False, padding_value=-1)
    assert torch.equal(idx.cpu(), ref)
    # rollout means (carried state, episode by episode) == padded sequence pass from zero state, on 64 whole trajectories
    valid = idx >= 0
    gi = idx.clamp(min=0).view(-1)
    obs_p = (algo.b_obs.view(T * N, 50).index_select(0, gi) * valid.view(-1, 1)).view(idx.shape[0], 64, 50)
    mu_seq = L.actor.forward(((obs_p - L.obs_mean) / L.obs_std).contiguous())
    noise = torch.stack(noises)                                              # [T, N, 10]
    mu_roll = (algo.b_act - noise * algo.fixed_std).view(T * N, 10).index_select(0, gi).view(idx.shape[0], 64, 10)
    d = ((mu_seq - mu_roll) * valid.unsqueeze(-1)).abs().max()
    assert float(d) < 2e-5, float(d)
    p0 = L.actor.params.clone(); c0 = L.critic.params.clone()
    algo.trace = []
    losses, kl, epochs_run = algo.update(ret)
    first = algo.trace[0].cpu().numpy()
    assert abs(first[3] - 1) < 1e-5 and abs(first[4]) < 1e-8                  # old == new on the first minibatch
    assert np.isfinite(losses).all() and epochs_run == 1
    assert float((L.actor.params - p0).abs().max()) > 1e-6 and float((L.critic.params - c0).abs().max()) > 1e-6
    assert torch.isfinite(L.actor.params).all() and torch.isfinite(L.critic.params).all()
    env.close()


def test_full_size_config4_td3_million_ring(dev, tmp_path):
    """BASELINE configs[4] at full size: Cassie-v0 TD3, 4096 envs, 256-unit actor / twin critics, 10^6-transition replay in HBM.  250
    lock-step steps wrap the ring once (1 024 000 transitions): ring pointer / size arithmetic (remote_replay.py:70-74), the slots
    hold what was added last, done_bool semantics, then updates on the full ring: finite statistics, the clipped double-Q target
    identity on a sampled batch (recomputed with plain torch ops from the same forwards), exact Polyak averaging."""
    from apex_amd.vecenv import CassieVecEnv
    from apex_amd.td3 import TD3
    N = 4096
    env = CassieVecEnv(n_envs=N, seed=4, max_traj_len=400)
    algo = TD3(env, str(tmp_path), hidden=256, batch_size=64, updates_per_step=0, replay_size=1_000_000, seed=2)
    algo.init_networks(0)
    L = algo.learner
    assert L.actor.H == 256
    algo.collect_and_train(249)
    rb = algo.replay
    assert rb.size == 1_000_000 and rb.ptr == (249 * N) % 1_000_000 and rb.s.shape == (1_000_000, 50)
    obs_before = algo.obs.clone()
    algo.collect_and_train(1)
    lo = (249 * N) % 1_000_000
    assert rb.ptr == (250 * N) % 1_000_000
    assert torch.equal(rb.s[lo:lo + N], obs_before)                           # the newest block sits right behind the old pointer
    assert set(torch.unique(rb.nd).cpu().tolist()) <= {0.0, 1.0} and float((rb.nd == 0).float().mean()) > 1e-3
    assert torch.isfinite(rb.s).all() and torch.isfinite(rb.s2).all() and torch.isfinite(rb.r).all() and float(rb.a.abs().max()) <= 1.0
    # ---- one update on a sampled batch, checked against plain torch ops on the SAME HIP forwards
    algo.batch_size = 64
    s, sn, ac, r, nd = rb.sample(64, algo.gen)
    noise = torch.randn(64, 10, device=dev, generator=algo.gen) * 0.2
    at0 = L.actor_t.params.clone(); a0 = L.actor.params.clone(); ct0 = L.critic_t_flat.clone()
    na = (torch.tanh(L.actor_t.forward(sn)) + noise.clamp(-0.5, 0.5)).clamp(-1, 1)
    q1t = L.q_t[0].forward(torch.cat([sn, na], 1).contiguous()).view(-1); q2t = L.q_t[1].forward(torch.cat([sn, na], 1).contiguous()).view(-1)
    target = r + nd * 0.99 * torch.minimum(q1t, q2t)
    q1 = L.q[0].forward(torch.cat([s, ac], 1).contiguous()).view(-1); q2 = L.q[1].forward(torch.cat([s, ac], 1).contiguous()).view(-1)
    ref_loss = float(((q1 - target) ** 2).mean() + ((q2 - target) ** 2).mean())
    stats, pl = L.train_step(s, ac, sn, r, nd, noise, 0, 0.99, 0.005, 0.5, 2)
    st = stats.cpu().numpy()
    np.testing.assert_allclose(st[0], ref_loss, rtol=2e-4)
    np.testing.assert_allclose(st[1] / 64, float(q1.mean()), rtol=2e-4, atol=1e-5)
    assert pl is not None                                                     # iteration 0 is a delayed-policy-update iteration
    np.testing.assert_allclose(L.actor_t.params.cpu().numpy(), (0.005 * L.actor.params + 0.995 * at0).cpu().numpy(), rtol=0, atol=2e-7)
    np.testing.assert_allclose(L.critic_t_flat.cpu().numpy(), (0.005 * L.critic_flat + 0.995 * ct0).cpu().numpy(), rtol=0, atol=2e-7)
    assert float((L.actor.params - a0).abs().max()) > 1e-5
    # ---- and a few collect + update rounds at the reference's batch size (apex.py:145)
    algo.updates_per_step = 2
    out = algo.collect_and_train(5)
    assert out["updates"] == 10 and np.isfinite([out["q_loss"], out["avg_q1"], out["avg_q2"]]).all()
    env.close()


def test_apx_rollout_equals_the_stepwise_loop(dev):
    """apx_rollout (the T-step PPO.sample loop as one C-ABI call) against the step-by-step Python loop over
    apx_mlp_forward / apx_env_step fed with the same action noise.  Same kernels in the same order; only the action mu + sigma * noise is
    rounded differently (fused multiply-add in the C path), so the first steps agree to round-off and the contact dynamics amplify it later."""
    from apex_amd.vecenv import CassieVecEnv
    from apex_amd.ppo import PPO
    def mk():
        env = CassieVecEnv(n_envs=256, seed=6, max_traj_len=10)
        args = dict(gamma=0.99, lam=0.95, lr=1e-4, eps=1e-5, entropy_coeff=0.0, clip=0.2, minibatch_size=1024, epochs=1, num_steps=256 * 24, max_traj_len=10,
                    max_grad_norm=0.05, mirror=True, std_dev=-1.5, seed=0)
        a = PPO(args, "/tmp/apx_test_unused", env); a.init_networks(0); a.normalization_params(256 * 50)
        return a
    a = mk()
    a.sample()                                              # one C call for the 24 steps
    noise = a.noise.clone()
    b = mk()
    b.noise_fn = lambda t, out: out.copy_(noise[t])          # the Python loop, replaying the same draws
    b.sample()
    assert torch.equal(a.b_obs[0], b.b_obs[0]) and torch.equal(a.b_mu[0], b.b_mu[0])      # (b_mu: the learner's forward over the recorded observations on both sides)
    # the action inside the one-launch rollout comes from the per-wave forward (sequential-k fp32, not the MFMA summation order): mean + sigma * noise to round-off
    np.testing.assert_allclose(a.b_act.cpu().numpy(), (a.b_mu + a.fixed_std * noise).cpu().numpy(), rtol=0, atol=3e-6)
    for t in range(1, 3):      # (a flipped encoder count moves a FIR-filtered motor velocity by a few 1e-2: allow a handful of such entries)
        close = np.isclose(a.b_obs[t].cpu().numpy(), b.b_obs[t].cpu().numpy(), rtol=0, atol=5e-4 * t)
        # (the largest single entry is an acceleration or a FIR velocity of a robot whose contact switched one substep earlier: m/s^2 scale - the differing-row-set
        # population of the teacher-forced tests reaches 1.2 m/s^2 on random-action rollouts, tests/test_gpu_env.py; ceiling 5 like there)
        assert close.mean() > 0.995 and np.abs(a.b_obs[t].cpu().numpy() - b.b_obs[t].cpu().numpy()).max() < 5.0, (t, close.mean())
        dr = np.abs(a.b_rew[t - 1].cpu().numpy() - b.b_rew[t - 1].cpu().numpy())      # same population rule as the observations: the env whose contact switched is the outlier
        assert (dr <= 5e-3).mean() > 0.99 and dr.max() < 0.1, (t, (dr <= 5e-3).mean(), dr.max())
    assert torch.equal(a.b_done[:2], b.b_done[:2])
    na, nb = int((a.b_done != 0).sum()), int((b.b_done != 0).sum())
    assert na > 256 and abs(na - nb) <= 0.05 * na                                # episodes ended and restarted inside the call, at the same rate
    assert torch.isfinite(a.b_obs).all() and torch.isfinite(a.b_val).all()


def test_one_launch_rollout_restarts_match_the_stepwise_resets(dev):
    """The auto-reset INSIDE env_rollout_kernel (round 5: the whole T-step rollout is one launch; a finished env restarts on its own wave - ring image or, when the ring does
    not hold the episode, the image computed in place - instead of in a masked env_reset_kernel launch) against the per-step launches (APX_ROLLOUT_STEPWISE=1) on a
    horizon of ONE step: every env restarts after every step, so every step of both runs starts one settle substep behind a reset of the same (seed, env, episode) and the
    two runs stay together (up to what a reset does not reset).  No image is prepared: from the third restart of an env on, both ring slots are stale and the in-kernel image path runs.  Done flags
    bit-equal, observations and rewards equal up to the round-off of two inlined copies of the substep (population rule for the FIR velocities: one encoder count)."""
    import os, subprocess, sys
    from apex_amd.vecenv import CassieVecEnv
    from apex_amd.ppo import PPO
    def run():
        env = CassieVecEnv(n_envs=256, seed=9, max_traj_len=1)
        args = dict(gamma=0.99, lam=0.95, lr=1e-4, eps=1e-5, entropy_coeff=0.0, clip=0.2, minibatch_size=1024, epochs=1, num_steps=256 * 8, max_traj_len=1,
                    max_grad_norm=0.05, mirror=True, std_dev=-1.5, seed=0)
        a = PPO(args, "/tmp/apx_test_unused", env); a.init_networks(0)
        _normalise_by_the_step_loop(a, 256 * 20)
        a.prepare_resets = False
        a.sample()
        return a
    a = run()
    assert (a.b_done == 2).all() and int(a.env.get_field("reset_miss")[0, 0]) == 0
    ep = a.env.get_field("ints")[:, 9]
    assert int(ep.min()) >= 8                                   # eight restarts per env: the ring (2 slots) was outrun
    # the stepwise twin in a fresh process (the switch is read once per process)
    code = ("import sys, torch; sys.path.insert(0, %r); import tests.test_gpu_ppo as T; a = T._stepwise_twin(); torch.save(dict(obs=a.b_obs.cpu(), rew=a.b_rew.cpu(), done=a.b_done.cpu(), act=a.b_act.cpu()), sys.argv[1])"
            % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    out = "/tmp/apx_stepwise_twin.pt"
    r = subprocess.run([sys.executable, "-c", code, out], env=dict(os.environ, APX_ROLLOUT_STEPWISE="1"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    b = torch.load(out)
    assert torch.equal(a.b_done.cpu(), b["done"])
    for t in range(a.T):
        d = (a.b_obs[t].cpu() - b["obs"][t]).abs().numpy()
        close = d <= 3e-4
        # (a flipped encoder count moves the FIR velocities of that env, and the torque delay line / encoder filters / estimator are NOT part of a reset: they carry such a
        # flip into the next episodes - 0.5 % of the entries at t = 3, 1.1 % at t = 7; a wrong image or settle step would move every entry of the env)
        assert close.mean() > 0.98 and d.max() < 0.2, (t, close.mean(), d.max())
        dr = (a.b_rew[t].cpu() - b["rew"][t]).abs().numpy()
        assert (dr <= 2e-3).mean() > 0.99 and dr.max() < 0.05, (t, dr.max())


def _stepwise_twin():
    from apex_amd.vecenv import CassieVecEnv
    from apex_amd.ppo import PPO
    env = CassieVecEnv(n_envs=256, seed=9, max_traj_len=1)
    args = dict(gamma=0.99, lam=0.95, lr=1e-4, eps=1e-5, entropy_coeff=0.0, clip=0.2, minibatch_size=1024, epochs=1, num_steps=256 * 8, max_traj_len=1,
                max_grad_norm=0.05, mirror=True, std_dev=-1.5, seed=0)
    a = PPO(args, "/tmp/apx_test_unused", env); a.init_networks(0)
    _normalise_by_the_step_loop(a, 256 * 20)
    a.prepare_resets = False
    a.sample()
    return a


def _normalise_by_the_step_loop(a, iters):
    """Observation statistics through the per-step Python loop (the replayed-noise path of PPO.normalization_params) in BOTH twins: the default path is itself a rollout
    launch as soon as normalization_params is routed through apx_rollout: its one-launch and per-step forms agree to round-off only, the twins would then start
    from slightly different normalisation constants."""
    a.noise_fn = lambda t, out: out.normal_(generator=a.gen)
    a.normalization_params(iters)
    a.noise_fn = None


def test_td3_async_collection_and_updates_on_two_streams(dev, tmp_path):
    """rl/algos/async_td3.py on the batched env (TD3.collect_and_train_async): collection with a periodically re-loaded behaviour copy and per-dimension action noise on
    one stream, the updates on a second one next to the following env step.  Same bookkeeping as the synchronous loop (one transition per env and step in the ring, as
    many updates as asked, moving weights, trailing targets), the behaviour copy equals the learner's actor as of its last re-load and lags the live actor, and - the
    race check - two runs from the same seed end with bit-identical parameters and replay contents although every update overlapped an env step and a replay write."""
    from apex_amd.vecenv import CassieVecEnv
    from apex_amd.td3 import TD3

    def run():
        env = CassieVecEnv(n_envs=256, seed=3, max_traj_len=15)
        algo = TD3(env, str(tmp_path), hidden=64, batch_size=128, updates_per_step=3, replay_size=3000, seed=1)      # 3000 < 24 x 256: the ring wraps twice
        algo.init_networks(0)
        a0 = algo.learner.actor.params.clone()
        snaps = []
        out = None
        for _ in range(3):
            out = algo.collect_and_train_async(8, load_freq=4)
            snaps.append(algo.behav[algo._cur].params.clone())
        torch.cuda.synchronize()
        res = (algo.learner.actor.params.clone(), algo.learner.critic_flat.clone(), algo.replay.s.clone(), algo.replay.a.clone(), algo.replay.r.clone(), snaps, out, a0, algo)
        return res
    actor, critic, rs, ra, rr, snaps, out, a0, algo = run()
    assert out["updates"] == 8 * 3 and np.isfinite([out["q_loss"], out["avg_q1"], out["avg_q2"]]).all()
    assert algo.replay.size == 3000 and algo.total_steps == 24 * 256 and algo.it == 24 * 3
    assert (ra.abs() <= 1).all() and (actor - a0).abs().max() > 1e-4
    # per-dimension exploration noise (async_td3.py:253-256): the stored actions of one step are not a common shift of tanh(behaviour(s))
    pre = torch.tanh(algo.behav[algo._cur].forward(rs[:256]))
    d = ra[:256] - pre
    assert d.std(dim=1).mean() > 0.05
    # the behaviour copy is a past state of the actor: it moved between re-loads and differs from the live actor
    assert (snaps[0] - a0).abs().max() > 0 and (snaps[2] - snaps[0]).abs().max() > 0 and (snaps[2] - actor).abs().max() > 0
    actor2, critic2, rs2, ra2, rr2, _, _, _, algo2 = run()
    assert torch.equal(actor, actor2) and torch.equal(critic, critic2) and torch.equal(rs, rs2) and torch.equal(ra, ra2) and torch.equal(rr, rr2)
    algo.env.close(); algo2.env.close()
