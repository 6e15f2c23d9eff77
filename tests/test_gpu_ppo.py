"""GPU: the batched PPO driver end to end on the HIP env (rollout grids, truncation bootstrap, returns, update)."""
import numpy as np
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
