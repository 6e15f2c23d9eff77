"""Golden G24: what the reference ships of a TRAINED policy and of its evaluation UNDER MUJOCO.

trained_models/nodelta_neutral_StateEst_symmetry_speed0-3_freq1-2/ and trained_models/5k_retrain/ hold actor.pt (Gaussian_FF_Actor 49-256-256-10, trained by the reference's
PPO on Cassie-v0 in MuJoCo, no dynamics randomisation, simrate 60) and eval_perturbs.npy = the output of the reference's own push sweep (test_policy.py:30-35,78-89 ->
tools/eval_perturb.py:97-160: for 100 push directions x 28 gait phases the largest 0.2 s pelvis push, in 10 N steps from 50 N, after which the robot is still up 3 s later),
i.e. a 100 x 28 table of closed-loop results of (policy, MuJoCo physics, estimator, PD loop).  This script stores the policy's tensors (state_dict order), its input
normaliser and that table; tests/test_oracle_env.py replays policy and protocol on the ORACLE's physics and compares with the table (sim-to-sim transfer)."""
from common import setup_reference_path, REF, GOLD
setup_reference_path()
import os
import numpy as np
import torch

out = {}
for tag, name in (("a", "nodelta_neutral_StateEst_symmetry_speed0-3_freq1-2"), ("b", "5k_retrain")):
    d = os.path.join(REF, "trained_models", name)
    pol = torch.load(os.path.join(d, "actor.pt"), weights_only=False)
    pol.eval()
    W = [p.detach().numpy().astype(np.float32) for p in pol.parameters()]
    assert [w.shape for w in W] == [(256, 49), (256,), (256, 256), (256,), (10, 256), (10,)]
    mean, std = pol.obs_mean.numpy().astype(np.float32), pol.obs_std.numpy().astype(np.float32)
    x = torch.randn(5, 49)
    h = (x.numpy() - mean) / std
    for k in (0, 2):
        h = np.maximum(h @ W[k].T + W[k + 1], 0.0)
    np.testing.assert_allclose(h @ W[4].T + W[5], pol(x, True).detach().numpy(), rtol=1e-4, atol=1e-5)      # deterministic action = the mean: Linear-ReLU-Linear-ReLU-Linear on the normalised input
    for i, w in enumerate(W):
        out[f"{tag}_w{i}"] = w
    out[f"{tag}_obs_mean"], out[f"{tag}_obs_std"] = mean, std
    ev = np.load(os.path.join(d, "eval_perturbs.npy"))
    assert ev.shape == (100, 28) and np.all(ev == np.round(ev / 10) * 10)
    out[f"{tag}_eval_perturbs"] = ev.astype(np.int16)
    out[f"{tag}_name"] = name
    ec = np.load(os.path.join(d, "eval_commands.npy"))      # tools/test_commands.py under MuJoCo: 10 000 random command schedules; rows (passed, half-period kind, speed, yaw offset, last speed step, last yaw step)
    assert ec.shape == (10000, 6)
    out[f"{tag}_eval_commands"] = ec.astype(np.float32)
# the "5k" stress test of policy a under MuJoCo (5k_test.py:19-74; 5k_test.pkl = five pickled lists: pass, terrain, mission, friction, foot mass of 17 328 trials = 2 terrains x
# 4 missions x 6 mission speeds x 19 frictions x 19 foot masses; the speed of a trial was not stored): the pass fractions on the flat terrain, and the missions' command files
import pickle
with open(os.path.join(REF, "trained_models", "nodelta_neutral_StateEst_symmetry_speed0-3_freq1-2", "5k_test.pkl"), "rb") as fh:
    p5, terr, mis, fric, mass = (pickle.load(fh) for _ in range(5))
p5, flat = np.array(p5), np.array([t.endswith("cassie.xml") for t in terr])
assert len(p5) == 17328 == 2 * 4 * 6 * 19 * 19
MISSIONS, MSPEEDS = ("straight", "curvy", "90_left", "90_right"), (0.5, 0.9, 1.4, 1.9, 2.3, 2.8)
out["k5_missions"] = np.array(MISSIONS)
out["k5_flat_pass"] = np.array([p5[flat & (np.array(mis) == m)].mean() for m in MISSIONS])
out["k5_mission_speeds"] = np.array(MSPEEDS)
for m in MISSIONS:
    for sp in MSPEEDS:
        with open(os.path.join(REF, "cassie", "missions", m, "command_trajectory_%s.pkl" % sp), "rb") as fh:
            d = pickle.load(fh)
        out["mission_%s_%s_speed" % (m, sp)] = np.asarray(d["speed"], np.float32); out["mission_%s_%s_orient" % (m, sp)] = np.asarray(d["orient"], np.float32)
out["protocol"] = np.array([60, 0.5, 3.0, 0.2, 50.0, 10.0])      # simrate, commanded speed, wait [s], push duration [s], first push [N], increment [N] (test_policy.py:30-35, experiment.pkl)
np.savez_compressed(os.path.join(GOLD, "g24_ref_policy_push_sweep.npz"), **out)
print("wrote g24_ref_policy_push_sweep.npz")
