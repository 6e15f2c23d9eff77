"""GPU parity: the HIP Cassie-v0 env (through the C ABI) vs the fp64 C++ oracle on the same seeds.

Bit-exact: episode-step indices, phase, done flags, RNG counters (integer work).  Floating point: fp32 lane vs fp64
host — tolerances are written next to each check; they widen with the number of 2 kHz substeps because the contact
dynamics amplify round-off (the physics itself is parity-unpinned against MuJoCo)."""
import numpy as np
import pytest
import torch

from oracle import sim as S

pytestmark = pytest.mark.gpu
N = 64


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def _mk(dyn_rand, seed, n=N):
    from apex_amd.vecenv import CassieVecEnv
    genv = CassieVecEnv(n_envs=n, dynamics_randomization=dyn_rand, seed=seed)
    oenv = [S.OracleEnv(dyn_rand=dyn_rand, seed=seed, env_id=i) for i in range(n)]
    return genv, oenv


def test_set_const_invweights(dev):
    genv, oenv = _mk(False, 0, 64)
    biw = genv.get_field("body_invweight0").cpu().numpy()[0]
    diw = genv.get_field("dof_invweight0").cpu().numpy()[0]
    ref_b = oenv[0].get("body_invweight0").reshape(26, 2)[:, 0]
    ref_d = oenv[0].get("dof_invweight0")
    cb = [b + 12 * leg for leg in (0, 1) for b in (5, 8, 9, 10, 12, 13)]     # bodies that carry constraints (connects, contact capsules)
    np.testing.assert_allclose(biw[cb], ref_b[cb], rtol=2e-4, atol=1e-7)
    lim = diw != 0                      # the kernel only computes it for limited joints
    assert lim.sum() == 16
    np.testing.assert_allclose(diw[lim], ref_d[lim], rtol=2e-4)


def test_reset_obs_and_params(dev):
    genv, oenv = _mk(True, 7)
    obs = genv.reset().cpu().numpy()
    ref = np.stack([e.reset() for e in oenv])
    ints = genv.get_field("ints").cpu().numpy()
    oints = np.stack([e.get("ints") for e in oenv])
    np.testing.assert_array_equal(ints[:, [0, 1, 2, 3]], oints[:, [0, 1, 2, 5]])          # time, phase, counter, rng ctr
    for name in ("mass", "damping", "friction", "motor_noise", "joint_noise"):
        np.testing.assert_allclose(genv.get_field(name).cpu().numpy(), np.stack([e.get(name) for e in oenv]), rtol=2e-6, atol=1e-7)
    # one substep of free fall from the init pose: fp32 round-off only
    np.testing.assert_allclose(obs, ref, rtol=1e-4, atol=2e-4)


def test_substeps_track_oracle(dev):
    """20 raw 2 kHz substeps after reset (still airborne, loop-closure + spring dynamics): tight agreement."""
    genv, oenv = _mk(False, 1)
    genv.reset(); [e.reset() for e in oenv]
    for _ in range(20):
        genv.substep()
    for e in oenv[:4]:
        for _ in range(20):
            e.substep()
    q = genv.get_field("qpos").cpu().numpy(); v = genv.get_field("qvel").cpu().numpy()
    for i, e in enumerate(oenv[:4]):
        np.testing.assert_allclose(q[i], e.get("qpos"), atol=2e-5)
        np.testing.assert_allclose(v[i], e.get("qvel"), atol=5e-3, rtol=1e-3)


def test_env_steps_vs_oracle(dev):
    """Free-running env steps (50 substeps each, contacts, reward, termination, command resampling) from the same reset with the same random actions: the integer
    bookkeeping (time, phase, cycle counter, RNG counter) and the done flags stay bit-exact, the SLOW observation groups (height, orientation, motor / joint
    positions, clock, commands) stay together with a drift bound per step.  The stiff groups (velocities, accelerations, reward) are NOT compared here: two free
    trajectories separate, and a tolerance that grows with t proves little - they are held to FIXED tolerances from identical states in
    test_teacher_forced_env_steps_random_actions / ..._on_walking_states."""
    genv, oenv = _mk(True, 3)
    genv.reset(); [e.reset() for e in oenv]
    rng = np.random.RandomState(0)
    grp = [(slice(0, 15), 3e-3), (slice(34, 40), 3e-3), (slice(46, 50), 1e-5)]
    for t in range(12):
        act = (rng.randn(N, 10) * 0.15).astype(np.float32)
        obs, rew, done, fin = genv.step(torch.tensor(act, device=dev), auto_reset=False)
        obs, rew, done = obs.cpu().numpy(), rew.cpu().numpy(), done.cpu().numpy()
        for i in range(N):
            o, r, d = oenv[i].step(act[i].astype(np.float64))
            assert d == done[i], (t, i)
            for sl, tol in grp:
                np.testing.assert_allclose(obs[i, sl], o[sl], atol=tol * (t + 1), err_msg=f"t={t} env={i} obs{sl}")
        ints = genv.get_field("ints").cpu().numpy()
        oints = np.stack([e.get("ints") for e in oenv])
        np.testing.assert_array_equal(ints[:, [0, 1, 2, 3]], oints[:, [0, 1, 2, 5]])


def test_teacher_forced_env_steps_random_actions(dev):
    """The teacher-forced comparison (kernel state overwritten with the oracle's before every env step) on RANDOM-action rollouts from reset: robots that stumble and
    fall - joint limits, shin / tarsus contacts, safety zones - instead of the trained gait.  Same binning by identical constraint-row sets and the same FIXED
    tolerances as test_teacher_forced_env_steps_on_walking_states."""
    E, same = _teacher_forced_rows(dev, n_steps=30, active=48, actions="random")
    Es, Ed = E[same], E[~same]
    for k, nm in enumerate(TF_NAMES):
        print("teacher-forced/random %-12s same sets (n=%d): median %.2e p99 %.2e max %.2e | differing sets (n=%d): max %.2e" % (
            nm, len(Es), np.median(Es[:, k]), np.percentile(Es[:, k], 99), Es[:, k].max(), len(Ed), Ed[:, k].max() if len(Ed) else 0))
    frac = 1.0 - same.mean()
    print("teacher-forced/random: differing row sets in %.4f of the (env, step) pairs" % frac)
    assert frac < 0.08, frac      # falling robots switch contacts more often than a gait does (measured 4.8 % of 1440 pairs)
    assert np.all(Es.max(0) <= TF_TOL_SAME), (Es.max(0), TF_TOL_SAME)


# (the safety-zone, coupled pitch-knee zone and early / max_vel reward comparisons of rounds 1-4 ran free with tolerances growing like (t + 1); they are scenarios of
# test_teacher_forced_scenario now: same action generators, fixed tolerances from identical states, "the zones were actually reached" asserted there)


def test_step_invariants_full_size(dev):
    """BASELINE size (4096 envs): finite outputs, quaternion norms, loop closure, reward range, auto-reset bookkeeping."""
    from apex_amd.vecenv import CassieVecEnv
    env = CassieVecEnv(n_envs=4096, seed=5, max_traj_len=30)
    env.reset()
    g = torch.Generator(device=dev).manual_seed(0)
    lens = torch.zeros(4096, device=dev)
    n_done = 0
    for t in range(40):
        act = torch.randn(4096, 10, device=dev, generator=g) * 0.2
        obs, rew, done, fin = env.step(act)
        assert torch.isfinite(obs).all() and torch.isfinite(rew).all()
        assert (rew > -0.5).all() and (rew < 1.0).all()
        lens += 1
        assert (lens[done == 2] == 30).all()                     # truncation exactly at max_traj_len
        lens[done != 0] = 0
        n_done += int((done != 0).sum())
        q = env.get_field("qpos")
        for a in (3, 10, 24):
            assert (q[:, a:a + 4].norm(dim=1) - 1).abs().max() < 1e-5
        ints = env.get_field("ints")
        assert (ints[:, 0] == lens).all()                        # time counter == steps since last reset
    assert n_done >= 4096                                        # every env finished at least once (horizon 30 < 40 steps)


def test_single_substep_crafted_states(dev):
    """One 2 kHz substep from crafted states, HIP vs oracle: low / pitched pelvis poses put foot, tarsus AND shin capsule ends
    into the floor (every contact-geometry branch of the row stage), knee / hip-roll / foot joints start past their range
    (limit rows).  A single substep keeps the comparison free of chaotic divergence: qacc and the integrated qvel must agree to
    fp32 solver accuracy."""
    genv, oenv = _mk(False, 11)
    genv.reset(); [e.reset() for e in oenv[:12]]
    # (these states need more rows than the lane map of the fast constraint stage keeps per leg - 3 to 4 capsule ends of a leg on the floor, 3 limits per leg: rounds 1-4
    # switched the oracle to the kernel's caps here; round 5 compares with the COMPLETE oracle, the kernel solves such a pass with its complete row set, cassie_complete.h)
    sat0 = genv.saturation()[0].cpu().numpy()
    rng = np.random.RandomState(5)
    qpos = genv.get_field("qpos").cpu().numpy().astype(np.float64)
    qvel = genv.get_field("qvel").cpu().numpy().astype(np.float64)
    cases = []
    for i in range(12):
        q = qpos[i].copy(); v = 0.05 * rng.randn(32)
        kind = i % 4
        if kind == 0:                       # standing low: both feet pressed into the floor
            q[2] = 0.80 + 0.02 * rng.rand()
        elif kind == 1:                     # pelvis pitched forward and low: tarsus / shin ends reach the floor
            q[2] = 0.45 + 0.05 * rng.rand()
            ang = 0.9 + 0.3 * rng.rand()
            q[3:7] = [np.cos(ang / 2), 0.0, np.sin(ang / 2), 0.0]
        elif kind == 2:                     # joint limits: hip roll, knee, foot beyond their ranges
            q[2] = 1.2
            q[7] = 0.45; q[14] = -0.60; q[20] = -0.45; q[21] = -0.45; q[28] = -2.95; q[34] = -2.50
        else:                               # rolled and low: one-sided contacts with mixed geoms
            q[2] = 0.50 + 0.05 * rng.rand()
            ang = 0.7
            q[3:7] = [np.cos(ang / 2), np.sin(ang / 2), 0.0, 0.0]
        cases.append((q, v))
        qpos[i] = q; qvel[i] = v
    genv.set_field("qpos", torch.tensor(qpos, dtype=torch.float32))
    genv.set_field("qvel", torch.tensor(qvel, dtype=torch.float32))
    genv.set_field("qacc_warm", torch.zeros(N, 32))
    genv.substep()
    qa = genv.get_field("qacc_warm").cpu().numpy(); qv = genv.get_field("qvel").cpu().numpy()
    ncon_seen, nlim_cases = 0, 0
    for i, e in enumerate(oenv[:12]):
        q, v = cases[i]
        e.set("qpos", q.astype(np.float32).astype(np.float64)); e.set("qvel", v.astype(np.float32).astype(np.float64)); e.set("qacc_warm", np.zeros(32))
        e.substep()
        ints = e.get("ints")
        ncon_seen = max(ncon_seen, int(ints[3]))
        nlim_cases += int(i % 4 == 2)
        ref_a, ref_v = e.get("qacc_warm"), e.get("qvel")
        scale = np.maximum(1.0, np.abs(ref_a))
        assert np.all(np.isfinite(qa[i]))
        tol = np.full(32, 3e-2); tol[[9, 22]] = 0.25      # spin of the achilles rods about their own axis: inertia 3.8e-6, fp32 solver noise
        # ... whose ABSOLUTE level follows the largest acceleration of the env: the rod's coupling entries M[spin][ancestor] are O(1e-9) remainders of O(1e-2) terms (inertia
        # about the pelvis origin), times ancestor accelerations of 2.5e5 rad/s^2 in the limit cases, over 3.8e-6.  The fp32 build of the ORACLE shows 9.1e-5 * max|qacc| on
        # these two dofs from the same states (tests/test_oracle_env.py::test_fp32_control_of_the_crafted_substep_tolerance, CPU suite): the floor below is 2.2 x that.  The
        # round-4 kernel sat at 0.237 of the per-dof 0.25 in case 10 by luck of its rounding order; the round-5 kernel (packed tree stage) at 3e-5 * max|qacc|.
        scale[[9, 22]] = np.maximum(scale[[9, 22]], 8e-4 * np.abs(ref_a).max())
        assert np.all(np.abs(qa[i] - ref_a) / scale <= tol), ("case %d" % i, np.abs(qa[i] - ref_a) / scale)
        np.testing.assert_allclose(qv[i], ref_v, atol=2e-3 + 5e-4 * np.abs(ref_a).max(), rtol=2e-3, err_msg="case %d" % i)
    assert ncon_seen >= 3          # more than the two foot ends: tarsus / shin geometry took part
    # the kernel reports exactly the saturation the oracle's complete collision pass sees in that substep
    sat = genv.saturation()[0].cpu().numpy()
    for i, e in enumerate(oenv[:12]):
        assert int(sat[i]) == (int(e.get("ints")[8]) | int(sat0[i])), (i, sat[i], e.get("ints")[8])
    assert (sat[:12] != 0).any()


def test_random_policy_statistics_match_oracle(dev):
    """Statistical parity at BASELINE size: random-policy rollouts with auto-reset, 4096 HIP envs vs 40 fp64 oracle envs on
    other streams of the same generator.  Episode-length and reward statistics must agree within sampling error (the
    trajectories themselves are chaotic, so this is the long-horizon counterpart of the per-step comparisons above)."""
    from apex_amd.vecenv import CassieVecEnv
    T, n, no = 240, 4096, 40
    g = CassieVecEnv(n_envs=n, seed=21)
    g.reset()
    gen = torch.Generator(device=dev); gen.manual_seed(5)
    ep_len = torch.zeros(n, device=dev); lens = []; rsum = 0.0; bad = 0
    for t in range(T):
        obs, rew, done, _ = g.step(0.2 * torch.randn(n, 10, device=dev, generator=gen))
        ep_len += 1
        d = done != 0
        bad += int((~torch.isfinite(obs)).sum() + (~torch.isfinite(rew)).sum())
        rsum += float(rew.sum())
        if bool(d.any()):
            lens.append(ep_len[d].cpu().numpy()); ep_len[d] = 0
    lens = np.concatenate(lens)
    assert bad == 0 and lens.size > 10000
    # the kernel's constraint caps (2 floor contacts + 1 limit per leg, 2 leg-leg rows): even on random-policy
    # rollouts, where robots fall all the time, fewer than 1 in 10^4 forward passes would have needed more (measured: 6.5e-7)
    flags, cnt = g.saturation()
    assert float(cnt.sum()) / (T * n * 50) < 1e-4, (float(cnt.sum()), flags.unique())
    rng = np.random.RandomState(7); olens = []; orsum = 0.0; osteps = 0
    for i in range(no):
        e = S.OracleEnv(dyn_rand=True, seed=21, env_id=i); e.reset(); L = 0
        for t in range(T // 2):
            _, r, d = e.step(0.2 * rng.randn(10)); L += 1; orsum += r; osteps += 1
            if d:
                olens.append(L); L = 0; e.reset()
    olens = np.array(olens)
    se = np.sqrt(lens.var() / lens.size + olens.var() / olens.size)
    assert abs(lens.mean() - olens.mean()) < 4.0 * se + 0.3, (lens.mean(), olens.mean(), se)
    assert abs(rsum / (T * n) - orsum / osteps) < 0.01, (rsum / (T * n), orsum / osteps)


def test_update_speed_and_reset_for_test_vs_oracle(dev):
    """Evaluation-side API (next row f3): CassieEnv.update_speed and reset_for_test through the C ABI vs the oracle (which is
    pinned to the reference by golden G16).  Both calls start from the ORACLE's state (teacher forcing, tests/state_xfer.py): what is compared is the call, not the
    fp32-vs-fp64 drift of the steps in front of it; the steps behind reset_for_test are the scenarios eval_step / eval_step_basic of test_teacher_forced_scenario."""
    from tests.state_xfer import oracle_to_kernel
    genv, oenv = _mk(True, 13)
    K = 16
    genv.reset(); [e.reset() for e in oenv[:K]]
    rng = np.random.RandomState(4)
    act = (rng.randn(N, 10) * 0.1).astype(np.float32)
    for t in range(2):
        [e.step(act[i].astype(np.float64)) for i, e in enumerate(oenv[:K])]
    oracle_to_kernel(genv, oenv[:K])
    # ---- update_speed: commands beyond the clip range included; phase rescale is integer work
    ns = rng.uniform(-0.8, 4.6, N).astype(np.float32); nd = rng.uniform(-0.5, 0.5, N).astype(np.float32)
    ns[K:] = ns[np.arange(K, N) % K]; nd[K:] = nd[np.arange(K, N) % K]
    genv.update_speed(torch.tensor(ns), torch.tensor(nd))
    ints = genv.get_field("ints").cpu().numpy()
    mism = 0
    for i, e in enumerate(oenv[:K]):
        e.update_speed(float(ns[i]), float(nd[i]))
        mism += int(ints[i, 1] != int(e.get("ints")[1]))
    assert mism <= 1                                           # the old cycle length is fp32 state: a tie within fp32 rounding may flip
    acts = act[np.arange(N) % K]
    obs, rew, done, _ = genv.step(torch.tensor(acts, device=dev), auto_reset=False)
    obs = obs.cpu().numpy()
    for i, e in enumerate(oenv[:K]):
        o, r, d = e.step(acts[i].astype(np.float64))
        np.testing.assert_allclose(obs[i, 48:50], o[48:50], atol=1e-6)                    # clipped commands
        np.testing.assert_allclose(obs[i, 46:48], o[46:48], atol=0.3 if mism else 1e-4)   # clock input sin / cos(2 pi phase / phaselen)
    # ---- reset_for_test, from the oracle's state again: the observation behind it is ONE settle substep away from identical states
    oracle_to_kernel(genv, oenv[:K])
    gobs = genv.reset_for_test().cpu().numpy()
    grp = [slice(0, 5), slice(5, 15), slice(15, 18), slice(18, 21), slice(21, 31), slice(31, 34), slice(34, 40), slice(40, 46), slice(46, 50)]
    for i, e in enumerate(oenv[:K]):
        o = e.reset_for_test()
        err = np.array([np.abs(gobs[i, sl] - o[sl]).max() for sl in grp])
        assert np.all(err <= TF_TOL_SAME[:9]), ("env %d" % i, err)
    gi = genv.get_field("ints").cpu().numpy()
    assert np.all(gi[:, :3] == 0)
    d0 = S.OracleEnv(dyn_rand=False, seed=0, env_id=0)
    np.testing.assert_allclose(genv.get_field("mass").cpu().numpy()[:4], np.tile(d0.get("mass"), (4, 1)), rtol=1e-6)
    np.testing.assert_allclose(genv.get_field("damping").cpu().numpy()[:4], np.tile(d0.get("damping"), (4, 1)), rtol=1e-6)
    assert np.all(genv.get_field("motor_noise").cpu().numpy() == 0) and np.all(genv.get_field("joint_noise").cpu().numpy() == 0)
    np.testing.assert_allclose(genv.get_field("friction").cpu().numpy()[:4, 0], 1.0)


def test_step_basic_vs_oracle(dev):
    """CassieEnv.step_basic through the C ABI vs the oracle: observation, counters; commands and RNG counter untouched."""
    genv, oenv = _mk(True, 17)
    genv.reset(); [e.reset() for e in oenv[:8]]
    genv.reset_for_test(); [e.reset_for_test() for e in oenv[:8]]
    genv.update_speed(torch.full((N,), 1.0)); [e.update_speed(1.0) for e in oenv[:8]]
    rng = np.random.RandomState(8)
    rc0 = genv.get_field("ints").cpu().numpy()[:, 3].copy()
    for t in range(4):
        act = (rng.randn(N, 10) * 0.1).astype(np.float32)
        obs = genv.step_basic(torch.tensor(act, device=dev)).cpu().numpy()
        gi = genv.get_field("ints").cpu().numpy()
        for i, e in enumerate(oenv[:8]):
            o = e.step_basic(act[i].astype(np.float64))      # (the observation itself: scenario eval_step_basic of test_teacher_forced_scenario, from identical states)
            oi = e.get("ints")
            assert (gi[i, 0], gi[i, 1], gi[i, 2]) == (oi[0], oi[1], oi[2])
            assert obs[i, 48] == 1.0 and obs[i, 49] == 0.0                       # the commanded speed stays put
    assert np.array_equal(genv.get_field("ints").cpu().numpy()[:, 3], rc0)       # no random draws


def test_full_reset_and_apply_force_vs_oracle(dev, golden_dir):
    """Row f3: reset_for_test(full_reset=True) and CassieSim.apply_force on the pelvis (tools/eval_perturb.py:31,62,70).
    The observation after the full reset is the reference's own (golden G16, real get_full_state on reset_cassie_state); after
    it every env is in the SAME state (init pose, default dynamics, zero delay line), so the pushed trajectories compare
    kernel vs oracle without accumulated history: per-env wrenches of different size / direction, 8 env steps of push, then
    release."""
    import os
    g = np.load(os.path.join(golden_dir, "g16_eval_api.npz"))
    genv, oenv = _mk(True, 21)
    K = 12
    genv.reset(); [e.reset() for e in oenv[:K]]
    rng = np.random.RandomState(9)
    act = (rng.randn(N, 10) * 0.1).astype(np.float32)
    for t in range(2):
        genv.step(torch.tensor(act, device=dev), auto_reset=False)
        [e.step(act[i].astype(np.float64)) for i, e in enumerate(oenv[:K])]
    genv.set_command(side_speed=-0.1)
    genv.apply_force(torch.tensor([30.0, 0, 0, 0, 0, 0]))                 # must be cleared by the full reset
    gobs = genv.reset_for_test(full_reset=True).cpu().numpy()
    np.testing.assert_allclose(gobs, np.tile(g["rft_full_obs"], (N, 1)), atol=1e-6)
    assert np.all(genv.get_field("xfrc").cpu().numpy() == 0) and np.all(genv.get_field("tq_fifo").cpu().numpy() == 0)
    for e in oenv[:K]:
        e.set("side_speed", -0.1); e.reset_for_test(full_reset=True)
    # plain attribute writes of the harness (env.speed = 0.5): no clock rebuild
    genv.set_command(speed=0.5)
    [e.set("speed", 0.5) for e in oenv[:K]]
    xfrc = np.zeros((N, 6), np.float32)
    ang = -2 * np.pi * np.arange(N) / 8.0
    size = 20.0 + 15.0 * (np.arange(N) % 7)
    xfrc[:, 0] = size * np.cos(ang); xfrc[:, 1] = size * np.sin(ang)
    xfrc[1::3, 3:] = rng.uniform(-5, 5, (len(xfrc[1::3]), 3))             # some envs also get a torque
    genv.apply_force(torch.tensor(xfrc))
    [e.apply_force(xfrc[i].astype(np.float64)) for i, e in enumerate(oenv[:K])]
    zero = torch.zeros(N, 10, device=dev)
    tol = np.full(50, 1e-2); tol[21:31] = 0.15; tol[31:34] = 0.3; tol[40:46] = 0.15
    for t in range(10):
        if t == 8:
            genv.apply_force(torch.zeros(6)); [e.apply_force(np.zeros(6)) for e in oenv[:K]]
        obs, rew, done, _ = genv.step(zero, auto_reset=False)
        obs = obs.cpu().numpy(); qp = genv.get_field("qpos").cpu().numpy(); qv = genv.get_field("qvel").cpu().numpy()
        for i, e in enumerate(oenv[:K]):
            o, r, d = e.step(np.zeros(10))
            sc = 1.0 + t
            np.testing.assert_allclose(qp[i, :3], e.get("qpos")[:3], atol=2e-3 * sc, err_msg="pelvis position env %d step %d" % (i, t))
            np.testing.assert_allclose(qv[i, :3], e.get("qvel")[:3], atol=3e-2 * sc, err_msg="pelvis velocity env %d step %d" % (i, t))
            assert np.all(np.abs(obs[i] - o) <= tol * sc + 5e-3 * np.abs(o)), ("env %d step %d" % (i, t), np.abs(obs[i] - o).max())
    # the state estimator (restarted by the full reset, then 10 env steps = 500 filter updates in both implementations): filter states of the
    # kernel's fp32 records against the oracle's fp64 object (positions / foot states / load share, velocities, heel springs, terrain)
    from tests.state_xfer import est_from_record
    est = genv.get_field("est").cpu().numpy()
    for i, e in enumerate(oenv[:K]):
        r = est_from_record(est[i])
        assert r["inited"] == 1.0 and e.get("est_flags")[0] == 1
        np.testing.assert_allclose(r["heel"], e.get("est_heel"), atol=2e-4, err_msg="heel springs env %d" % i)
        np.testing.assert_allclose(r["hx"][:, [0, 2, 3, 4]], e.get("est_hx").reshape(2, 6)[:, [0, 2, 3, 4]], atol=2e-2, err_msg="horizontal filter positions env %d" % i)
        np.testing.assert_allclose(r["hx"][:, 1], e.get("est_hx").reshape(2, 6)[:, 1], atol=0.3, err_msg="horizontal filter velocity env %d" % i)
        np.testing.assert_allclose(r["zx"][:4], e.get("est_zx")[:4], atol=3e-2, err_msg="vertical filter env %d" % i)
        assert abs(r["terrain"] - e.get("est_terrain")[0]) < 5e-3
    # the push actually moved the robots apart: lateral pelvis velocity follows the direction of the wrench
    vy = genv.get_field("qvel").cpu().numpy()[:, 1]
    assert np.corrcoef(vy[:64], xfrc[:64, 1])[0, 1] > 0.5


def test_cassie_traj_v0_reset_and_steps_vs_oracle(dev):
    """CassieTraj-v0 with the CLI defaults (next row f1, env half; golden G17 pins the oracle to the reference): same Philox streams
    in kernel and oracle -> same start phase / first speed, the robot starts from the reference trajectory's pose of that phase
    (not from the init pose), and the following env steps agree like Cassie-v0's."""
    from apex_amd.vecenv import CassieVecEnv
    genv = CassieVecEnv(n_envs=N, dynamics_randomization=True, seed=17, env_name="CassieTraj-v0")
    K = 12
    oenv = [S.OracleEnv(dyn_rand=True, seed=17, env_id=i, env_kind=1) for i in range(K)]
    gobs = genv.reset().cpu().numpy()
    oobs = np.stack([e.reset() for e in oenv])
    gi = genv.get_field("ints").cpu().numpy(); gq = genv.get_field("qpos").cpu().numpy(); gv = genv.get_field("qvel").cpu().numpy()
    cmd = genv.get_field("cmd").cpu().numpy()
    for i, e in enumerate(oenv):
        assert int(gi[i, 1]) == int(e.get("ints")[1])                                    # start phase: integer draw, bit-exact
        assert abs(cmd[i, 5] - e.get("phaselen")[0]) < 1e-4                              # clock from the randint speed
        np.testing.assert_allclose(gq[i], e.get("qpos"), atol=2e-4); np.testing.assert_allclose(gv[i], e.get("qvel"), atol=3e-2)
        np.testing.assert_allclose(gobs[i, 46:50], oobs[i, 46:50], atol=1e-5)
    assert np.abs(gq[:, 2] - 1.01).max() > 5e-3 and len(np.unique(np.round(gq[:, 7], 3))) > 8        # poses differ from env to env: not the init pose
    # (the env steps behind the reset: scenario cassie_traj of test_teacher_forced_scenario, from identical states with fixed tolerances)
    # the masked restart inside apx_env_step uses the same trajectory-pose reset
    genv2 = CassieVecEnv(n_envs=N, dynamics_randomization=False, seed=4, env_name="CassieTraj-v0", max_traj_len=2)
    genv2.reset()
    z = torch.zeros(N, 10, device=dev)
    genv2.step(z); _, _, done, _ = genv2.step(z)
    assert (done == 2).all()
    q = genv2.get_field("qpos").cpu().numpy()
    assert np.abs(q[:, 2] - 1.01).max() > 5e-3 and np.isfinite(q).all()


def test_edge_cases_simrate_empty_mask_and_bad_arguments(dev):
    """Edge cases through the C ABI: a non-default simrate (apex.py:18 --simrate; phase length and the clock follow 2000 // simrate) against
    the oracle; a reset with an all-zero mask is a no-op; invalid configurations fail loudly with a message instead of launching."""
    from apex_amd.vecenv import CassieVecEnv
    from apex_amd import _lib
    genv = CassieVecEnv(n_envs=N, dynamics_randomization=True, seed=8, simrate=40)
    oenv = [S.OracleEnv(simrate=40, dyn_rand=True, seed=8, env_id=i) for i in range(6)]
    gobs = genv.reset().cpu().numpy()
    for i, e in enumerate(oenv):
        np.testing.assert_allclose(gobs[i, 46:50], e.reset()[46:50], atol=1e-5)
        assert abs(genv.get_field("cmd").cpu().numpy()[i, 5] - e.get("phaselen")[0]) < 1e-4          # 2 (swing + stance) * (2000 // 40)
    act = (np.random.RandomState(1).randn(N, 10) * 0.1).astype(np.float32)
    obs, rew, done, _ = genv.step(torch.tensor(act, device=dev), auto_reset=False)
    obs, rew = obs.cpu().numpy(), rew.cpu().numpy()
    tol = np.full(50, 1.2e-2); tol[21:31] = 0.3; tol[31:34] = 0.6; tol[40:46] = 0.3
    for i, e in enumerate(oenv):
        o, r, d = e.step(act[i].astype(np.float64))
        assert np.all(np.abs(obs[i] - o) <= tol + 5e-3 * np.abs(o)), (i, np.abs(obs[i] - o).max())
        assert abs(rew[i] - r) < 0.02
    # reset with an empty mask: nothing changes
    q0 = genv.get_field("qpos").clone(); i0 = genv.get_field("ints").clone()
    genv.reset(mask=torch.zeros(N, dtype=torch.uint8, device=dev))
    assert torch.equal(genv.get_field("qpos"), q0) and torch.equal(genv.get_field("ints"), i0)
    # loud failures
    for kw, frag in ((dict(n_envs=100), "multiple of 64"), (dict(n_envs=64, env_name="CassieTraj-v0", simrate=40), "simrate")):
        with pytest.raises((_lib.ApxError, NotImplementedError)) as ei:
            CassieVecEnv(**kw)
        assert frag in str(ei.value)
    with pytest.raises(_lib.ApxError):
        genv.step(torch.zeros(N, 10))                                       # host tensor: no CPU path


def test_saturation_flags_vs_oracle_crafted(dev):
    """Row 'contact / limit set of cassie.xml': the fast path of the kernel instantiates 2 floor contacts + 1 limit per leg and three capsule pairs; everything else
    cassie.xml can produce (pelvis sphere :87, hip-pitch capsules :101,164, left-right capsule pairs :119-144, further capsule ends / limits) is detected, counted in
    I_SAT and - since round 5 - solved out of line with the complete row set (cassie_complete.h; the folded robot on the floor has 17 contacts = 63 basis records, i.e. the
    HBM tier of the record pool is on the path).  Crafted single-forward-pass states, one per geom pair class: the kernel's flags equal the flags of the oracle's complete
    collision pass, and kernel and (complete) oracle accelerations agree, saturated or not."""
    genv, oenv = _mk(False, 12)
    genv.reset()
    q0 = genv.get_field("qpos").cpu().numpy().astype(np.float64)[0]
    def case(**kw):
        q = q0.copy()
        for k, v in kw.items():
            q[int(k[1:])] = v
        return q
    fold = dict(q9=1.3, q23=1.3, q14=-2.4, q28=-2.4)
    cases = [("standing", case(), 0),
             ("airborne", case(q2=1.5), 0),
             ("pelvis sphere on the floor", case(q2=0.10, **fold), 4),
             ("crossed legs: one left-right pair in contact, a row in the kernel", case(q2=1.5, q7=-0.10, q21=0.10), 0),
             ("legs apart", case(q2=1.5, q7=-0.08, q21=0.08), 0),
             ("feet brushing: two pairs", case(q2=1.5, q7=-0.115, q21=0.115), None),
             ("legs pushed through each other: 6 pairs > 3 rows", case(q2=1.5, q7=-0.15, q21=0.15), 8),
             ("two limits on one leg", case(q2=1.5, q14=-2.9, q20=-2.5), 2),
             ("one limit per leg", case(q2=1.5, q14=-2.9, q34=-2.5), 0),
             ("feet pressed in, toes and tarsus down", case(q2=0.70), None)]
    qpos = np.tile(q0, (N, 1)); 
    for i, (_, q, _) in enumerate(cases):
        qpos[i] = q
    genv.set_field("qpos", torch.tensor(qpos, dtype=torch.float32)); genv.set_field("qvel", torch.zeros(N, 32)); genv.set_field("qacc_warm", torch.zeros(N, 32))
    s0 = genv.saturation()[0].cpu().numpy()
    genv.substep()
    sat, cnt = (x.cpu().numpy() for x in genv.saturation())
    qa = genv.get_field("qacc_warm").cpu().numpy()
    for i, (name, q, want) in enumerate(cases):
        e = oenv[i]; e.reset()
        e.set("qpos", q.astype(np.float32).astype(np.float64)); e.set("qvel", np.zeros(32)); e.set("qacc_warm", np.zeros(32))
        f0 = int(e.get("ints")[8])
        e.substep()
        oflag = int(e.get("ints")[8]) & ~f0 if f0 == 0 else None
        st = e.get("ints")
        cur = int(sat[i]) & ~int(s0[i]) if s0[i] == 0 else int(sat[i])
        if want is not None:
            assert cur & want == want and (want != 0 or cur == 0), (name, cur, want)
        if oflag is not None:
            assert cur == oflag, (name, cur, oflag)
        # saturated or not: the accelerations are those of the COMPLETE row set (round 5: a saturated pass is solved out of line with every row, cassie_complete.h).
        # Left out: the legs pushed THROUGH each other - seven redundant capsule-pair rows, an ill-conditioned dual on which 50 Gauss-Seidel sweeps end at a different
        # vertex for every rounding: the fp32 build of the ORACLE differs from the fp64 oracle by 13.8 (relative) / 2 750 rad/s^2 there, with other pairs active
        if name.startswith("legs pushed through"):
            continue
        ref = e.get("qacc_warm"); scale = np.maximum(1.0, np.abs(ref))
        tol = np.full(32, 3e-2); tol[[9, 22]] = 0.25
        scale[[9, 22]] = np.maximum(scale[[9, 22]], 8e-4 * np.abs(ref).max())      # (the rod-spin dofs: see test_single_substep_crafted_states)
        # a dof of a few rad/s^2 inside a vector of 1e5 - 7e5 rad/s^2 (17 contacts: the folded robot on the floor) carries the rounding of the large ones:
        # floor of 2e-5 * max|qacc| absolute under the relative bound, as in test_heightfield_terrain_vs_oracle (fp32 control there)
        err = np.abs(qa[i] - ref)
        assert np.all((err / scale <= tol) | (err <= 2e-5 * np.abs(ref).max())), (name, cur, err / scale)
    assert cnt[2] >= 1 and cnt[3] == 0 and cnt[0] == cnt[1] == 0
    assert int(oenv[3].get("ints")[9]) >= 1          # the crossed-legs case really had a leg-leg row in the oracle


def test_trained_policy_never_saturates_the_constraint_caps(dev):
    """The kept checkpoint (trained_models/r04_cassie_v0_clock) walking for 300 steps on 256 envs with dynamics randomisation, speeds 0-3 m/s:
    no forward pass of an env that stays up needs a constraint row the kernel does not instantiate (I_SAT stays 0); envs that fall may
    saturate only in the steps right before termination.  The same counters stay 0 through a push-recovery trial that is survived."""
    import os
    from apex_amd.vecenv import CassieVecEnv
    sys_path = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import sys; sys.path.insert(0, sys_path)
    import apex
    env = CassieVecEnv(n_envs=256, seed=31, max_traj_len=300)
    actor, mean, std = apex._load_actor(os.path.join(sys_path, "trained_models", "r04_cassie_v0_clock"), env.device)
    obs = env.reset()
    alive = torch.ones(256, dtype=torch.bool, device=dev)
    sat_alive = torch.zeros(256, dtype=torch.int64, device=dev)
    for t in range(300):
        obs, rew, done, _ = env.step(actor.forward(obs, mean, std), auto_reset=False)
        if t % 10 == 9:
            flags, cnt = env.saturation()
            sat_alive = torch.where(alive & (done == 0), cnt, sat_alive)     # counters of envs that are still walking
        alive &= done != 1
    flags, cnt = env.saturation()
    walked = alive
    assert int(walked.sum()) >= 200, int(walked.sum())                       # the policy walks (dynamics randomisation, random commands)
    assert int(cnt[walked].max()) == 0, (flags[walked].unique(), int(cnt[walked].max()))
    env.close()
    # push recovery (tools/eval_perturb.py semantics): 60 N for 0.24 s on the pelvis, survived -> still no saturation
    env = CassieVecEnv(n_envs=64, seed=32, max_traj_len=100000, dynamics_randomization=False)
    obs = env.reset_for_test(full_reset=True); env.set_command(speed=0.5)
    push = torch.zeros(64, 6, device=dev); ang = torch.linspace(0, 6.28, 65, device=dev)[:64]
    push[:, 0] = 60 * torch.cos(ang); push[:, 1] = 60 * torch.sin(ang)
    for t in range(200):
        if t == 60: env.apply_force(push)
        if t == 68: env.apply_force(torch.zeros(64, 6, device=dev))
        obs = env.step_basic(actor.forward(obs, mean, std))
    z = env.get_field("qpos")[:, 2]
    up = z > 0.6
    flags, cnt = env.saturation()
    assert int(up.sum()) >= 32 and int(cnt[up].max()) == 0, (int(up.sum()), flags[up].unique())


def test_physics_invariants_on_the_hip_side(dev):
    """The invariant checks of tests/test_oracle_env.py run on the KERNEL at 4096 envs: (a) free fall with joint damping off conserves
    kinetic + potential energy and follows z = z0 - g t^2 / 2; (b) the loop closures stay closed; (c) quaternions stay normalised;
    (d) contact complementarity on standing robots: foot force >= 0 and only with a foot on the ground, the feet carry the weight."""
    from apex_amd.vecenv import CassieVecEnv
    n = 4096
    env = CassieVecEnv(n_envs=n, dynamics_randomization=False, seed=2)
    env.reset()
    q = env.get_field("qpos"); q[:, 2] = 2.0 + torch.rand(n, device=dev)
    env.set_field("qpos", q); env.set_field("qvel", torch.zeros(n, 32, device=dev)); env.set_field("damping", torch.zeros(n, 32, device=dev))
    env.set_field("pd_target", env.get_field("so_mpos"))        # hold the current motor positions: the PD block sees no error, tiny torques
    z0 = q[:, 2].clone()
    orc = S.OracleEnv(dyn_rand=False); orc.reset()
    def energy(qv, vv):                                          # kinetic + potential of the first 8 envs from the fp64 oracle's formulas
        out = []
        for i in range(8):
            orc.set("qpos", qv[i].double().cpu().numpy()); orc.set("qvel", vv[i].double().cpu().numpy()); out.append(orc.energy())
        return np.array(out)
    e0 = energy(env.get_field("qpos"), env.get_field("qvel"))
    for _ in range(400):                                         # 0.2 s
        env.substep()
    q1, v1 = env.get_field("qpos"), env.get_field("qvel")
    assert torch.isfinite(q1).all() and torch.isfinite(v1).all()
    np.testing.assert_allclose((q1[:, 2] - (z0 - 0.5 * 9.81 * 0.2 ** 2)).abs().max().item(), 0, atol=4e-3)
    e1 = energy(q1, v1)
    assert np.abs(e1 - e0).max() / np.abs(e0).max() < 5e-3, (e0, e1)     # fp32 state, PD hold on armature'd motors: 0.5 %
    for a in (3, 10, 24):
        assert (q1[:, a:a + 4].norm(dim=1) - 1).abs().max() < 1e-5
    viol = []
    for i in range(8):
        orc.set("qpos", q1[i].double().cpu().numpy()); orc.set("qvel", v1[i].double().cpu().numpy()); orc.phys_forward(); viol.append(orc.violation())
    assert max(viol) < 2e-4, viol                                # soft constraints pull the loops closed (6.6 mm at the init pose)
    env.close()
    # (d) standing robots under the PD hold
    env = CassieVecEnv(n_envs=n, dynamics_randomization=False, seed=3)
    env.reset()
    fz_sum = torch.zeros(n, device=dev)
    for t in range(8):
        env.step(torch.zeros(n, 10, device=dev), auto_reset=False)
        fwd = env.get_field("fwd")                               # [:, 0:2] world-z contact force on the left / right foot body of the last substep
        assert (fwd[:, :2] >= -1e-3).all()
        foot_z = fwd[:, [12, 15]]                                # sole height of the feet
        assert (fwd[:, :2][foot_z > 0.10] == 0).all()            # no force on a foot that is clearly off the ground
        fz_sum += fwd[:, :2].sum(1)
    w = 33.3 * 9.81
    assert 0.5 * w < float(fz_sum.mean()) / 8 < 1.5 * w          # the feet carry the robot's weight on average


@pytest.mark.parametrize("reward,cp", [("clock", 1), ("library_clock", 2)])
def test_phase_command_profile_vs_oracle(dev, reward, cp):
    """Row f4: command_profile="phase" (cassie.py:266-271,529-545,805-808).  Per-reset swing / stance duration and stance mode (integer
    draws: bit-exact vs the oracle, which golden G22 pins against the reference), the 55-entry observation with the 9-entry command tail,
    per-env clocks in the reward, update_speed that keeps the clock."""
    from apex_amd.vecenv import CassieVecEnv
    g = CassieVecEnv(n_envs=N, seed=13, command_profile="phase", reward=reward)
    assert g.obs_dim == 55 and g.observation_space.shape == (55,) and len(g.mirrored_obs) == 55
    o = [S.OracleEnv(dyn_rand=True, seed=13, env_id=i, command_profile=cp) for i in range(8)]
    obs = g.reset().cpu().numpy()
    assert obs.shape == (N, 55)
    cmd = g.get_field("cmd").cpu().numpy(); ints = g.get_field("ints").cpu().numpy()
    modes = set()
    for i, e in enumerate(o):
        ob = e.reset()
        sw = e.get("swing_stance")
        np.testing.assert_allclose(cmd[i, 3:5], sw, rtol=1e-6)                  # (k / 100) in fp32
        assert abs(cmd[i, 5] - e.get("phaselen")[0]) < 1e-4 and int(ints[i, 1]) == int(e.get("ints")[1])
        np.testing.assert_allclose(obs[i, 46:], ob[46:], atol=2e-6)               # clock, swing, stance, one-hot mode, speeds
        np.testing.assert_allclose(obs[i, :46], ob[:46], rtol=1e-4, atol=3e-4)
        assert obs[i, 50:53].sum() == 1.0
        modes.add(int(cmd[i, 6]))
    assert set(np.unique(cmd[:, 6]).astype(int)) == {0, 1, 2}                     # all three stance modes occur over the batch
    rng = np.random.RandomState(1)
    for t in range(6):
        act = (rng.randn(N, 10) * 0.15).astype(np.float32)
        ob, rew, done, _ = g.step(torch.tensor(act, device=dev), auto_reset=False)
        ob, rew, done = ob.cpu().numpy(), rew.cpu().numpy(), done.cpu().numpy()
        for i, e in enumerate(o):
            oo, rr, dd = e.step(act[i].astype(np.float64))
            assert dd == done[i]
            np.testing.assert_allclose(ob[i, 46:], oo[46:], atol=2e-6)      # integer phase, durations, mode, commands: exact whatever the dynamics do
            # (the 46 dynamics entries and the reward: scenarios phase_clock / phase_library of test_teacher_forced_scenario)
    g.update_speed(1.5, 0.1)
    c2 = g.get_field("cmd").cpu().numpy()
    np.testing.assert_allclose(c2[:, :2], [[1.5, 0.1]] * N); np.testing.assert_allclose(c2[:, 3:7], cmd[:, 3:7])      # durations / mode kept


def test_ppo_iteration_on_the_phase_profile(dev, tmp_path):
    """The driver + HIP learner with the 55-entry observation of command_profile="phase": mirror lists of length 55, fused 2 x 256 forward
    with D = 55, one whole iteration, checkpoint classes with 55 inputs."""
    from apex_amd.vecenv import CassieVecEnv
    from apex_amd.ppo import PPO
    env = CassieVecEnv(n_envs=256, seed=2, command_profile="phase", reward="clock", max_traj_len=20)
    args = dict(gamma=0.99, lam=0.95, lr=1e-4, eps=1e-5, entropy_coeff=0.0, clip=0.2, minibatch_size=1024, epochs=2, num_steps=256 * 16, max_traj_len=20,
                max_grad_norm=0.05, mirror=True, std_dev=-1.5, seed=0)
    algo = PPO(args, str(tmp_path), env)
    algo.init_networks(0); algo.normalization_params(256 * 50)
    assert algo.D == 55 and algo.learner.actor.D == 55 and algo.b_obs.shape == (16, 256, 55)
    out = algo.iteration()
    assert np.isfinite(out["losses"]).all() and out["losses"][3] > 0.5 and out["losses"][5] >= 0
    # mirror loss of the phase-profile lists: a policy is compared with itself on M_s s; the command tail maps onto itself
    o = algo.b_obs.view(-1, 55)[:64]
    sp = algo.learner.obs_sp.cpu().numpy()
    assert (sp[46:] == np.arange(46, 55)).all()
    algo.save()
    pol = torch.load(str(tmp_path / "actor.pt"), weights_only=False)
    assert pol(torch.zeros(55), deterministic=True).shape[-1] == 10


def _terrain(kind, n=41, seed=0):
    """small synthetic height fields over [-4, 4] x [-4, 4] (0.2 m cells like cassie_hfield.xml's 500 x 500 over 100 m): raw values, scale 0.15"""
    xs = np.linspace(-4, 4, n)
    X, Y = np.meshgrid(xs, xs)                       # rows along y, columns along x
    if kind == "slope":
        return (0.5 + 0.6 * X / 4).astype(np.float32)                     # 2.25 % grade in x after the 0.15 scale
    if kind == "noise":                                                   # like terrains/noise*.npy: values in [0, 0.25] -> bumps up to 3.75 cm
        return (0.25 * np.random.RandomState(seed).rand(n, n)).astype(np.float32)
    return (0.3 + 0.3 * np.sin(1.3 * X) * np.cos(0.9 * Y)).astype(np.float32)      # rolling hills, +-4.5 cm


@pytest.mark.parametrize("kind", ["slope", "noise", "hills"])
def test_heightfield_terrain_vs_oracle(dev, kind):
    """Row f4, terrain: CassieSim("cassie_hfield.xml") + set_hfield_data (util/eval.py:73-76) as apx_env_set_hfield.  The foot / tarsus /
    shin capsule ends collide with the grid triangle under them, each contact in its own frame.  HIP vs the fp64 oracle (same triangle
    rule): single substeps on crafted poses (qacc), then env steps from reset_for_test with the PD hold."""
    from apex_amd.vecenv import CassieVecEnv
    hf = _terrain(kind)
    size = (4.0, 4.0, 0.15)
    g = CassieVecEnv(n_envs=N, seed=4, dynamics_randomization=False, max_traj_len=1000)
    g.set_hfield(hf, size)
    o = [S.OracleEnv(dyn_rand=False, seed=4, env_id=i).set_hfield(hf, size) for i in range(6)]
    g.reset_for_test(); [e.reset_for_test() for e in o]
    # the robot starts at the origin: lift / shift it so that the feet meet the terrain at different places
    q = g.get_field("qpos").cpu().numpy().astype(np.float64)
    rng = np.random.RandomState(3)
    for i in range(N):
        q[i, 0] = rng.uniform(-2.5, 2.5); q[i, 1] = rng.uniform(-2.5, 2.5)
        hh, _ = o[0].floor_query(q[i, 0], q[i, 1])
        q[i, 2] = 1.0 + hh + rng.uniform(-0.06, 0.0)                     # feet 1-6 cm into the surface
    g.set_field("qpos", torch.tensor(q, dtype=torch.float32)); g.set_field("qvel", torch.zeros(N, 32)); g.set_field("qacc_warm", torch.zeros(N, 32))
    g.substep()
    qa = g.get_field("qacc_warm").cpu().numpy()
    ncon = 0
    for i, e in enumerate(o):
        e.set("qpos", q[i].astype(np.float32).astype(np.float64)); e.set("qvel", np.zeros(32)); e.set("qacc_warm", np.zeros(32))
        e.substep()                       # (the COMPLETE oracle: a third capsule end of a leg on a bump is solved by the kernel's complete-row path, cassie_complete.h)
        ncon += int(e.get("ints")[3])
        ref = e.get("qacc_warm"); scale = np.maximum(1.0, np.abs(ref))
        tol = np.full(32, 3e-2); tol[[9, 22]] = 0.25
        # a dof whose acceleration is a few rad/s^2 inside a vector of 2 000 - 10 000 rad/s^2 (the crafted penetration) carries the
        # rounding of the large ones: the fp32 build of the ORACLE (make -C oracle f32) already moves such a dof by 1.3e-2 relative
        # (slope, env 0) and the stiff ones by 2.4e-4 of max|qacc|; floor of 2e-5 * max|qacc| absolute under the relative bound
        err = np.abs(qa[i] - ref)
        ok = (err / scale <= tol) | (err <= 2e-5 * np.abs(ref).max())
        assert np.all(ok), (kind, i, err / scale, err / np.abs(ref).max())
    assert ncon >= 6
    # env steps with the PD hold from those states (kernel vs oracle: scenarios hfield_* of test_teacher_forced_scenario)
    for t in range(4):
        g.step_basic(torch.zeros(N, 10, device=dev))
    assert np.isfinite(g.get_field("qpos").cpu().numpy()).all()
    g.set_hfield(None)                                                   # back to the plane: a robot at the origin stands at the usual height
    g.reset_for_test()
    for _ in range(3):
        g.step_basic(torch.zeros(N, 10, device=dev))
    assert abs(float(g.get_field("qpos")[:, 2].mean()) - 0.95) < 0.08


def test_observation_history_stack(dev):
    """--history h (cassie.py:51-55,565,856-859): observation = newest frame followed by the h previous frames of the episode, zeros before
    its start; a finished env's final observation ends the OLD stack, its auto-reset observation starts a new one.  Frames come from the same
    kernel: a history-0 twin env on the same seed produces them."""
    from apex_amd.vecenv import CassieVecEnv
    h = 2
    a = CassieVecEnv(n_envs=N, seed=8, max_traj_len=6, history=h)
    b = CassieVecEnv(n_envs=N, seed=8, max_traj_len=6)
    assert a.obs_dim == 150 and a.observation_space.shape == (150,)
    oa, ob = a.reset().clone(), b.reset().clone()
    assert torch.equal(oa[:, :50], ob) and float(oa[:, 50:].abs().max()) == 0
    frames = [ob.clone()]
    g = torch.Generator(device=dev).manual_seed(1)
    for t in range(9):
        act = torch.randn(N, 10, device=dev, generator=g) * 0.1
        oa, ra, da, fa = a.step(act); ob, rb, db, fb = b.step(act)
        oa, fa, ob, fb = oa.clone(), fa.clone(), ob.clone(), fb.clone()
        assert torch.equal(da, db) and torch.equal(ra, rb) and torch.equal(oa[:, :50], ob)
        ended = (da != 0)
        prev1, prev2 = frames[-1], (frames[-2] if len(frames) > 1 else torch.zeros_like(ob))
        if len(frames) > 1:
            prev2 = torch.where(last_ended.view(-1, 1), torch.zeros_like(prev2), prev2)      # an episode that began one step ago has one old frame only
        # running envs: [new, prev1, prev2]; finished + restarted envs: [new, 0, 0] and the final observation [final frame, prev1, prev2]
        exp = torch.cat([ob, torch.where(ended.view(-1, 1), torch.zeros_like(prev1), prev1), torch.where(ended.view(-1, 1), torch.zeros_like(prev2), prev2)], 1)
        assert torch.equal(oa, exp), t
        if bool(ended.any()):
            expf = torch.cat([fb, prev1, prev2], 1)
            assert torch.equal(fa[ended], expf[ended])
        frames.append(ob.clone()); last_ended = ended
    assert int((torch.stack([f for f in frames]).abs().sum() > 0)) == 1


def test_estimator_twin_from_identical_state(dev):
    """The restated reference estimator (state_output_step; golden G11 pins the fp64 oracle to the binary) in its lane form: both sides start ONE env
    step from the same state (teacher forcing, tests/state_xfer.py), so the 50 filter updates see the same sensors up to the fp32 physics of one step
    (pairs whose signature of constraint rows and estimator load switches differs in some substep - a foot load within round-off of the filters' 50 N threshold flips a
    gain - are counted, bounded and left out, like in every teacher-forced test).
    Checked on the estimator's own state: heel springs (two Newton steps vs the reference's Levenberg-Marquardt), all three filters' states and
    covariances, terrain; and on the three observation groups it produces.  Tolerances are fixed (no growth with the rollout length)."""
    from tests.state_xfer import oracle_to_kernel, est_from_record
    import os
    n = 64
    genv, oenv = _mk(True, 21, n)
    genv.reset(); [e.reset() for e in oenv]
    policy = torch.load(os.path.join(os.path.dirname(__file__), "..", "trained_models", "r04_cassie_v0_clock", "actor.pt"), weights_only=False).eval()
    obs_o = np.stack([e.obs() for e in oenv])
    worst = np.zeros(8); n_pairs = n_diff = 0
    for t in range(30):
        with torch.no_grad():
            act = policy(torch.tensor(obs_o, dtype=torch.float32), deterministic=True).numpy()
        act = (act + np.random.RandomState(t).randn(n, 10) * 0.05).astype(np.float32)
        oracle_to_kernel(genv, oenv)
        obs, rew, done, _ = genv.step(torch.tensor(act, device=dev), auto_reset=False)
        obs = obs.cpu().numpy(); est = genv.get_field("est").cpu().numpy(); kh = _kernel_hash(genv)
        for i, e in enumerate(oenv):
            o, r, d = e.step(act[i].astype(np.float64)); obs_o[i] = o
            n_pairs += 1
            if int(kh[i]) != _oracle_hash(e):      # a constraint row or one of the estimator's load switches (50 N: process noise of a foot state) differed in some substep: not round-off
                n_diff += 1
                if d:
                    e.reset(); obs_o[i] = e.obs()
                continue
            k = est_from_record(est[i])
            ohx, ohP, ozx, ozP = e.get("est_hx").reshape(2, 6), e.get("est_hP").reshape(2, 6, 6), e.get("est_zx"), e.get("est_zP").reshape(5, 5)
            errs = np.array([np.abs(k["heel"] - e.get("est_heel")).max(), np.abs(k["hx"][:, [0, 2, 3]] - ohx[:, [0, 2, 3]]).max(), np.abs(k["hx"][:, 1] - ohx[:, 1]).max(),
                             np.abs(k["hx"][:, 4] - ohx[:, 4]).max(), np.abs(k["zx"][:4] - ozx[:4]).max(), abs(k["terrain"] - e.get("est_terrain")[0]),
                             np.abs(k["hP"] - ohP)[:, :5, :5].max() / np.abs(ohP[:, :5, :5]).max(), np.abs(k["zP"][:4, :4] - ozP[:4, :4]).max() / np.abs(ozP[:4, :4]).max()])
            worst = np.maximum(worst, errs)
            if d:
                e.reset(); obs_o[i] = e.obs()
    print("estimator twin, worst over the %d of %d pairs with identical row-set / estimator-switch signatures [heel, positions, velocity, load share, vertical, terrain, rel P(h), rel P(z)]:" % (n_pairs - n_diff, n_pairs), worst)
    assert n_diff < TF_MAX_DIFFERING_FRACTION * n_pairs, (n_diff, n_pairs)
    assert worst[0] < 6e-4 and worst[1] < 2e-3 and worst[2] < 2e-2 and worst[3] < 2e-2 and worst[4] < 2e-3 and worst[5] < 1e-4 and worst[6] < 2e-2 and worst[7] < 2e-2, worst


def _kernel_hash(genv):
    """I_ROWSET of every env: hash of the active constraint-row sets over the 50 forward passes of the most recent env step (bit-exact integer view)"""
    return genv.get_field("ints_bits").view(torch.int32)[:, 8].cpu().numpy().astype(np.int64) & 0xffffffff


def _oracle_hash(e):
    ii = e.get("ints")
    return int(ii[10]) | int(ii[11]) << 16


from tests.state_xfer import TF_NAMES, TF_TOL_SAME, TF_TOL_SAME_P99, TF_MAX_DIFFERING_FRACTION      # fixed tolerances, shared with the fp32 control test (CPU suite)


def _teacher_forced_rows(dev, n_steps=100, active=32, actions="policy"):
    from tests.state_xfer import oracle_to_kernel
    import os
    n = 64
    genv, oenv = _mk(True, 22, n)
    genv.reset(); [e.reset() for e in oenv]
    policy = torch.load(os.path.join(os.path.dirname(__file__), "..", "trained_models", "r04_cassie_v0_clock", "actor.pt"), weights_only=False).eval()
    obs_o = np.stack([e.obs() for e in oenv])
    grp = [slice(0, 5), slice(5, 15), slice(15, 18), slice(18, 21), slice(21, 31), slice(31, 34), slice(34, 40), slice(40, 46), slice(46, 50)]
    rows, same = [], []
    rng = np.random.RandomState(5)
    for t in range(n_steps):
        if actions == "policy":
            with torch.no_grad():
                act = policy(torch.tensor(obs_o, dtype=torch.float32), deterministic=True).numpy().astype(np.float32)
        else:
            act = (rng.randn(n, 10) * 0.15).astype(np.float32)
        oracle_to_kernel(genv, oenv)
        obs, rew, done, _ = genv.step(torch.tensor(act, device=dev), auto_reset=False)
        obs, rew, done = obs.cpu().numpy(), rew.cpu().numpy(), done.cpu().numpy()
        qp, qv = genv.get_field("qpos").cpu().numpy(), genv.get_field("qvel").cpu().numpy()
        ints = genv.get_field("ints").cpu().numpy()
        kh = _kernel_hash(genv)
        for i, e in enumerate(oenv[:active]):
            o, r, d = e.step(act[i].astype(np.float64)); obs_o[i] = o
            assert d == done[i], (t, i, d, done[i])
            np.testing.assert_array_equal(ints[i, [0, 1, 2, 3]], e.get("ints")[[0, 1, 2, 5]])
            rows.append([np.abs(obs[i, sl] - o[sl]).max() for sl in grp] + [abs(rew[i] - r), np.abs(qp[i] - e.get("qpos")).max(), np.abs(qv[i] - e.get("qvel")).max()])
            same.append(int(kh[i]) == _oracle_hash(e))
            if d:
                e.reset(); obs_o[i] = e.obs()
        for i in range(active, n):                         # keep the rest of the batch defined (they mirror env 0's action stream)
            o, r, d = oenv[i].step(act[i].astype(np.float64)); obs_o[i] = o
            if d:
                oenv[i].reset(); obs_o[i] = oenv[i].obs()
    return np.array(rows), np.array(same)


def test_teacher_forced_env_steps_on_walking_states(dev):
    """One env step of the kernel against one env step of the oracle FROM THE SAME STATE, on the states a trained policy visits (walking, contact switches
    every step), 100 steps x 32 envs: the kernel's whole state is overwritten with the oracle's before every step (tests/state_xfer.py).  Every (env, step)
    pair is binned by whether kernel and oracle had the SAME active constraint-row sets (limits, capsule ends, body-floor, leg-leg pairs) in all 50 substeps
    (I_ROWSET against the oracle's hash of the same signature words).  On the identical-set population every tolerance is FIXED and is the fp32 level that the
    fp32 control build of the oracle itself shows against the fp64 oracle; the differing-set population (a contact that switches a substep earlier or later) is
    bounded in SIZE.  Integer bookkeeping bit-exact.  Negative control: test_teacher_forced_test_fails_on_a_one_contact_kernel."""
    E, same = _teacher_forced_rows(dev)
    Es, Ed = E[same], E[~same]
    for k, nm in enumerate(TF_NAMES):
        print("teacher-forced %-12s same sets (n=%d): median %.2e p99 %.2e max %.2e | differing sets (n=%d): median %.2e max %.2e" % (
            nm, len(Es), np.median(Es[:, k]), np.percentile(Es[:, k], 99), Es[:, k].max(), len(Ed), np.median(Ed[:, k]) if len(Ed) else 0, Ed[:, k].max() if len(Ed) else 0))
    frac = 1.0 - same.mean()
    print("teacher-forced: differing row sets in %.4f of the (env, step) pairs" % frac)
    assert frac < TF_MAX_DIFFERING_FRACTION, frac
    assert np.all(Es.max(0) <= TF_TOL_SAME), (Es.max(0), TF_TOL_SAME)
    assert np.all(np.percentile(Es, 99, axis=0) <= TF_TOL_SAME_P99), (np.percentile(Es, 99, axis=0), TF_TOL_SAME_P99)
    # the differing-set pairs: one contact / limit row more or less for a substep or two moves the stiff signals of that step (accelerations, FIR motor velocities);
    # a ceiling only, far above the identical-set level and far below a wrong model (the one-contact kernel breaks it)
    if len(Ed):
        assert np.all(Ed.max(0) <= np.array([1e-3, 5e-3, 2e-2, 0.1, 1.0, 5.0, 1e-2, 1.0, 1e-5, 5e-2, 1e-2, 2.0])), Ed.max(0)


def test_teacher_forced_test_fails_on_a_one_contact_kernel(dev):
    """NEGATIVE CONTROL: the same test against a kernel that instantiates only ONE floor contact per leg (libapx_maxc1.so: make VARIANT=maxc1
    EXTRA=-DAPX_NEG_MAXC1, built by __graft_entry__.build()) must FAIL - the tolerances above can tell a wrong constraint set from round-off."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, "apex_amd", "lib", "libapx_maxc1.so")
    if not os.path.exists(lib):
        pytest.skip("build it with: make -C apex_amd/csrc VARIANT=maxc1 EXTRA=-DAPX_NEG_MAXC1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_env.py"), "-x", "-q", "-m", "gpu", "-k", "test_teacher_forced_env_steps_on_walking_states"],
                       env=dict(os.environ, APX_LIB=lib), capture_output=True, text=True, timeout=1200, cwd=root)
    assert r.returncode != 0 and "1 failed" in r.stdout, r.stdout[-1500:]
    assert "AssertionError" in r.stdout or "assert" in r.stdout


def test_range_checked_build_sees_no_out_of_range_index(dev):
    """The env kernels compiled with -DAPX_CHECK (every S(f) / S.W(i) / S.I(f) index range-checked, apex_amd/csrc/env_state.h) driven through all entry
    points, terrains, command profiles and env kinds on falling robots (tools/t_check.py): no index leaves its LDS region.  DESIGN.md section 4.1 records
    two faults of round 2 that were never explained; this build is the standing check that the shipped sources index inside their regions."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, "apex_amd", "lib", "libapx_check.so")
    if not os.path.exists(lib):
        pytest.skip("build it with: make -C apex_amd/csrc VARIANT=check EXTRA=-DAPX_CHECK")
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "t_check.py"), "40"], env=dict(os.environ, APX_LIB=lib), capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "RESULT clean" in out.stdout, out.stdout[-2000:]


def test_min_input_profile_vs_oracle(dev):
    """input_profile "min" (cassie.py:246-256,829-837): the observation is built from the state estimator's foot positions / orientations (leftFoot / rightFoot of
    state_out_t: golden G11 pins them in the oracle, G8 the packing and the mirror list) instead of the 46 joint-level entries.  Kernel vs oracle over a
    reset and 6 env steps, clock and phase command profiles; 25 / 30 entries."""
    from apex_amd.vecenv import CassieVecEnv
    for cp, cpi, dim in (("clock", 0, 25), ("phase", 1, 30)):
        genv = CassieVecEnv(n_envs=64, dynamics_randomization=True, seed=12, input_profile="min", command_profile=cp)
        assert genv.obs_dim == dim and len(genv.mirrored_obs) == dim
        oenv = [S.OracleEnv(dyn_rand=True, seed=12, env_id=i, input_profile=1, command_profile=cpi) for i in range(8)]
        obs = genv.reset().cpu().numpy()
        ref = np.stack([e.reset() for e in oenv])
        np.testing.assert_allclose(obs[:8], ref, atol=3e-4)
        rng = np.random.RandomState(3)
        for t in range(6):
            act = (rng.randn(64, 10) * 0.1).astype(np.float32)
            obs, rew, done, _ = genv.step(torch.tensor(act, device=dev), auto_reset=False)
            obs = obs.cpu().numpy()
            for i, e in enumerate(oenv):
                o, r, d = e.step(act[i].astype(np.float64))
                assert d == done[i]
                np.testing.assert_allclose(obs[i, 21:], o[21:], atol=1e-5)      # clock + commands; the foot entries: scenarios min_clock / min_phase of test_teacher_forced_scenario
        genv.close()


def test_fractional_phase_add_vs_oracle(dev):
    """self.phase_add = 1.5 of the command harness (tools/test_commands.py:86, cassie.py:448,511): the phase advances by 1.5 per env step and wraps on the float
    comparison; clock entries of the observation and the clock reward follow the float phase.  40 steps (two wraps) against the oracle, mixed 1 / 1.5 batch."""
    genv, oenv = _mk(False, 41, 64)
    genv.reset_for_test(); [e.reset_for_test() for e in oenv[:8]]
    pa = torch.tensor([1.5 if i % 2 == 0 else 1.0 for i in range(64)])
    genv.set_command(speed=1.6, phase_add=pa)
    for i, e in enumerate(oenv[:8]):
        e.set("speed", [1.6]); e.set("phase_add", [float(pa[i]), 0])
    zero = torch.zeros(64, 10, device=dev)
    halves = 0
    for t in range(40):
        obs, rew, done, _ = genv.step(zero, auto_reset=False)
        obs, rew = obs.cpu().numpy(), rew.cpu().numpy(); ints = genv.get_field("ints").cpu().numpy()
        for i, e in enumerate(oenv[:8]):
            o, r, d = e.step(np.zeros(10))
            oi = e.get("ints"); half = int(e.get("phase_add")[1]); halves += half
            assert int(ints[i, 1]) == int(oi[1]) and (int(ints[i, 4]) >> 5) & 1 == half and int(ints[i, 2]) == int(oi[2]), (t, i)
            np.testing.assert_allclose(obs[i, 46:50], o[46:50], atol=2e-6)      # (the reward along the float phase: scenario phase_add of test_teacher_forced_scenario)
    assert halves > 50 and int(oenv[0].get("ints")[2]) >= 1 and int(oenv[0].get("ints")[2]) > int(oenv[1].get("ints")[2]) - 1      # half phases occurred, the 1.5 envs wrapped


def test_estimator_record_recovers_from_non_finite_state(dev):
    """A diverged env must not poison its persistent estimator record (fp32 filters have no way back from a NaN): the kernel restarts the estimator of such
    an env (state_output_setup) and the next update initialises it again; the other envs are untouched."""
    from tests.state_xfer import est_from_record
    genv, _ = _mk(False, 51, 64)
    genv.reset()
    for _ in range(3):
        genv.step(torch.zeros(64, 10, device=dev), auto_reset=False)
    est = genv.get_field("est")
    ref = est.clone()
    est[3, 18] = float("nan"); est[7, 144] = float("inf"); est[9, 0] = float("nan")      # a filter state, a heel spring, a covariance entry
    genv.set_field("est", est)
    genv.substep()
    e1 = genv.get_field("est").cpu().numpy()
    assert np.isfinite(e1).all()
    for i in (3, 7, 9):      # restarted from state_output_setup by this update: covariances at their initial 1e-6 level, the horizontal pelvis position back at 0
        k = est_from_record(e1[i])
        assert k["inited"] == 1.0 and np.abs(k["hP"][:, :5, :5]).max() < 1e-4 and np.abs(k["zP"][:4, :4]).max() < 1e-4 and np.abs(k["hx"][:, 0]).max() < 1e-2, (i, k)
    assert est_from_record(e1[4])["inited"] == 1.0 and np.abs(e1[4][:21] - ref[4][:21].cpu().numpy()).max() < 1e-3
    genv.substep()
    e2 = genv.get_field("est").cpu().numpy()
    assert all(est_from_record(e2[i])["inited"] == 1.0 for i in (3, 7, 9)) and np.isfinite(e2).all()
    # and the other way round: non-finite SENSORS (a NaN joint position) leave a zeroed record, restarted on the next finite update
    qp = genv.get_field("qpos"); qp[12, 9] = float("nan"); genv.set_field("qpos", qp)
    genv.substep(); genv.substep()      # (the estimator reads the sensor snapshot taken at the end of the previous substep)
    e3 = genv.get_field("est").cpu().numpy()
    assert np.isfinite(e3).all() and est_from_record(e3[12])["inited"] == 0.0 and np.abs(e3[12][:144]).max() == 0.0


def test_diverged_env_ends_its_episode(dev):
    """The product build is -ffast-math: the divergence guard (non-finite height or reward -> done, reward 0) has to be a bit test the optimiser cannot
    see through (c4::fbits; a plain bitcast test is folded to `false`, found in round 3 by reading the disassembly).  A NaN / inf pelvis height in two
    envs: exactly those end their episode with reward 0 and finite observations after the auto-reset; the other envs step on, finite."""
    genv, _ = _mk(False, 52, 64)
    genv.reset()
    zero = torch.zeros(64, 10, device=dev)
    for _ in range(3):
        genv.step(zero, auto_reset=False)
    qp = genv.get_field("qpos")
    qp[5, 2] = float("nan"); qp[11, 2] = float("inf")
    genv.set_field("qpos", qp)
    obs, rew, done, _ = genv.step(zero, auto_reset=True)
    obs, rew, done = obs.cpu().numpy(), rew.cpu().numpy(), done.cpu().numpy()
    assert done[5] == 1 and done[11] == 1 and rew[5] == 0.0 and rew[11] == 0.0
    assert done.sum() == 2 and np.isfinite(rew).all()
    for _ in range(3):
        obs, rew, done, _ = genv.step(zero, auto_reset=True)
    assert np.isfinite(obs.cpu().numpy()).all() and np.isfinite(rew.cpu().numpy()).all()
    assert np.isfinite(genv.get_field("est").cpu().numpy()).all() and np.isfinite(genv.get_field("qpos").cpu().numpy()).all()


def test_apply_force_on_any_body_vs_oracle(dev):
    """CassieSim.apply_force(xfrc, body_name) for bodies other than the pelvis (tools/eval_perturb.py's perturb_body; mjData.xfrc_applied row of that
    body, J^T over its ancestor chain): single substeps from identical states, kernel vs oracle, for a foot, a tarsus, a hip-pitch and a plantar rod on
    either side.  The pushed body must matter: the same substep without the wrench differs by much more than the parity tolerance."""
    from tests.state_xfer import oracle_to_kernel
    n = 64
    genv, oenv = _mk(False, 61, n)
    genv.reset(); [e.reset() for e in oenv]
    zero = torch.zeros(n, 10, device=dev)
    for _ in range(2):
        genv.step(zero, auto_reset=False); [e.step(np.zeros(10)) for e in oenv]
    rng = np.random.RandomState(3)
    for body in ("left-foot", "right-tarsus", "left-hip-pitch", "right-plantar-rod", "right-foot", "cassie-pelvis"):
        xfrc = np.concatenate([rng.uniform(-40, 40, (n, 3)), rng.uniform(-3, 3, (n, 3))], 1).astype(np.float32)
        oracle_to_kernel(genv, oenv)
        q0 = np.stack([e.get("qvel") for e in oenv])
        genv.apply_force(torch.tensor(xfrc), body)
        [e.apply_force(xfrc[i].astype(np.float64), body) for i, e in enumerate(oenv)]
        for _ in range(4):
            genv.substep(); [e.substep() for e in oenv]
        qv = genv.get_field("qvel").cpu().numpy(); qo = np.stack([e.get("qvel") for e in oenv])
        moved = np.abs(qo - q0).max()
        err = np.abs(qv - qo).max()
        assert err < 2e-2 and err < 0.05 * moved, (body, err, moved)
        # without the wrench the same substeps end somewhere else (the wrench is not a no-op on this body)
        genv.apply_force(torch.zeros(6), body); [e.apply_force(np.zeros(6), body) for e in oenv]
    genv.apply_force(torch.zeros(6))
    assert int(genv.get_field("ints")[0, 7]) == 1      # back on the pelvis row


def test_prepared_resets_are_bit_identical_to_computed_ones(dev):
    """apx_env_prepare_resets: the draws, set_const and forward pass of the next two resets of every env are computed ahead of time into a per-env
    ring, keyed by the episode index; an auto-reset that finds its episode there copies it and only runs the settle step.  Two envs with the same seed, one prepared before
    every step and one never, must stay BIT-identical over rollouts full of resets (random actions, short episodes: envs that end three times between two
    preparations exercise the fallback as well), for both env kinds and the phase command profile."""
    from apex_amd.vecenv import CassieVecEnv
    # (+ apx_env_set_refill: env a also refills the ring of the envs that just restarted on its own side stream, next to the following env step; env b never does)
    for kw, prep_every, refill in ((dict(), 1, False), (dict(), 25, False), (dict(env_name="CassieTraj-v0"), 3, False), (dict(command_profile="phase"), 2, False),
                                   (dict(dynamics_randomization=False), 1, False), (dict(), 1000, True), (dict(env_name="CassieTraj-v0"), 7, True),
                                   (dict(command_profile="phase"), 1000, True), (dict(dynamics_randomization=False), 1000, True)):
        a = CassieVecEnv(n_envs=256, seed=31, max_traj_len=12, **kw); b = CassieVecEnv(n_envs=256, seed=31, max_traj_len=12, **kw)
        a.set_refill(refill); b.set_refill(False)
        oa, ob = a.reset().clone(), b.reset().clone()
        assert torch.equal(oa, ob)
        g = torch.Generator(device=dev); g.manual_seed(4)
        nres = 0
        for t in range(40):
            if t % prep_every == 0 and (t > 0 or not refill):
                a.prepare_resets()
            act = torch.randn(256, 10, device=dev, generator=g) * 0.3
            oa, ra, da, fa = a.step(act); ob, rb, db, fb = b.step(act)
            assert torch.equal(oa, ob) and torch.equal(ra, rb) and torch.equal(da, db), (kw, t)
            nres += int((da != 0).sum())
        for name in ("qpos", "qvel", "qacc_warm", "mass", "damping", "friction", "floor", "body_invweight0", "dof_invweight0", "motor_noise", "joint_noise", "snap", "cmd", "fwd", "tq_fifo", "menc"):
            assert torch.equal(a.get_field(name), b.get_field(name)), (kw, name)
        ia, ib = a.get_field("ints_bits").view(torch.int32), b.get_field("ints_bits").view(torch.int32)
        assert torch.equal(ia, ib)
        assert torch.equal(a.get_field("est"), b.get_field("est"))
        assert float(a.get_field("reset_miss")[0, 0]) == 0 and float(b.get_field("reset_miss")[0, 0]) == 0
        assert nres > 256 * 2 and int(ia[:, 9].max()) >= 4          # every env restarted several times: ring hits and (with rare preparation) fallbacks
        a.close(); b.close()


from tests.tf_scenarios import SCENARIOS as _SCENARIOS, G50 as _G50, TOL_TORQUE, TOL_MIN_FOOT_POS, TOL_MIN_FOOT_ORI


@pytest.mark.parametrize("name", [s.name for s in _SCENARIOS])
def test_teacher_forced_scenario(dev, name):
    """Every env configuration / action regime that round 4 still compared free-running with tolerances growing like (t + 1): safety zones, coupled pitch-knee zone, early /
    max_vel rewards, evaluation API (reset_for_test + update_speed, step and step_basic), CassieTraj-v0, the phase command profile, height fields, the min input profile,
    fractional phase_add (tests/tf_scenarios.py).  TEACHER-FORCED like test_teacher_forced_env_steps_on_walking_states: the kernel's whole state is overwritten with the
    fp64 oracle's before EVERY step, every (env, step) pair is binned by identical constraint-row sets in all 50 substeps, and the identical-set population is held to the
    FIXED tolerances of tests/state_xfer.py (observation groups, reward, qpos, qvel) plus a fixed drive-torque tolerance and, for the min profile, fixed foot-entry
    tolerances - all shown to be the fp32 level by the CPU-side control of the SAME scenarios (tests/test_oracle_env.py::test_fp32_control_of_the_scenarios).  Done flags
    and the integer bookkeeping (time, phase, cycle counter, RNG counter) bit-exact on every pair."""
    from apex_amd.vecenv import CassieVecEnv
    from tests.state_xfer import oracle_to_kernel
    from tests.tf_scenarios import BY_NAME
    sc = BY_NAME[name]
    genv = sc.make_kernel(CassieVecEnv)
    oenv = sc.make_oracle(S)
    n, K = genv.n_envs, len(oenv)
    rng = np.random.RandomState(sc.rng_seed)
    grp = sc.obs_groups or _G50
    rows, same, reached = [], [], 0
    for t in range(sc.n_steps):
        act = sc.act(t, rng, K).astype(np.float32)
        oracle_to_kernel(genv, oenv)
        ak = torch.tensor(act[np.arange(n) % K], device=dev)
        if sc.stepper == "step":
            obs, rew, done, _ = genv.step(ak, auto_reset=False)
            rew, done = rew.cpu().numpy(), done.cpu().numpy()
        else:
            obs = genv.step_basic(ak); rew = np.zeros(n, dtype=np.float32); done = np.zeros(n, dtype=np.int64)
        obs = obs.cpu().numpy()
        qp, qv = genv.get_field("qpos").cpu().numpy(), genv.get_field("qvel").cpu().numpy()
        tq, mp = genv.get_field("so_torque").cpu().numpy(), genv.get_field("so_mpos").cpu().numpy()
        ints = genv.get_field("ints").cpu().numpy(); kh = _kernel_hash(genv)
        for i, e in enumerate(oenv):
            o, r, d = sc.step_oracle(e, act[i].astype(np.float64))
            assert d == done[i], (name, t, i, d, done[i])
            np.testing.assert_array_equal(ints[i, [0, 1, 2, 3]], e.get("ints")[[0, 1, 2, 5]], err_msg="%s t=%d env=%d" % (name, t, i))
            pa = e.get("phase_add")
            assert (int(ints[i, 4]) >> 5) & 1 == int(pa[1]), (name, t, i)                                            # the half phase of phase_add = 1.5
            rows.append([np.abs(obs[i, sl] - o[sl]).max() for sl in grp] + [abs(rew[i] - r), np.abs(qp[i] - e.get("qpos")).max(), np.abs(qv[i] - e.get("qvel")).max(),
                                                                             np.abs(tq[i] - e.get("so_torque")).max(), np.abs(mp[i] - e.get("so_mpos")).max()])
            same.append(int(kh[i]) == _oracle_hash(e))
        if sc.reaches is not None:
            reached += sc.reaches(oenv)
        for e in oenv:
            if sc.stepper == "step" and (e.get("qpos")[2] < 0.4 or int(e.get("ints")[0]) >= sc.okw.get("max_traj_len", 400)):
                e.reset()
    E, same = np.array(rows), np.array(same)
    G = len(grp)
    Es, frac = E[same], 1.0 - same.mean()
    print("teacher-forced %-16s pairs %d differing sets %.3f | same sets: obs groups max %s | reward %.1e qpos %.1e qvel %.1e torque %.2e mpos %.1e" % (
        name, len(E), frac, np.array2string(Es[:, :G].max(0), precision=1, max_line_width=300), Es[:, G].max(), Es[:, G + 1].max(), Es[:, G + 2].max(), Es[:, G + 3].max(), Es[:, G + 4].max()))
    assert frac <= sc.differing_max, (name, frac)
    if sc.reaches is not None:
        assert reached > 0, name                                     # the zones were actually reached
    if sc.min_profile:
        assert Es[:, 0].max() <= TOL_MIN_FOOT_POS and Es[:, 3].max() <= TOL_MIN_FOOT_ORI and Es[:, 1].max() <= TF_TOL_SAME[0] and Es[:, 2].max() <= TF_TOL_SAME[3] and Es[:, 4].max() <= 1e-5, (name, Es[:, :G].max(0))
    else:
        assert np.all(Es[:, :G].max(0) <= TF_TOL_SAME[:G]), (name, Es[:, :G].max(0), TF_TOL_SAME[:G])
    assert Es[:, G].max() <= TF_TOL_SAME[9] and Es[:, G + 1].max() <= TF_TOL_SAME[10] and Es[:, G + 2].max() <= TF_TOL_SAME[11], (name, Es[:, G:].max(0))
    assert Es[:, G + 3].max() <= TOL_TORQUE and Es[:, G + 4].max() <= TF_TOL_SAME[1], (name, Es[:, G + 3].max(), Es[:, G + 4].max())
    genv.close()
