"""Pins the env-logic half of the C++ oracle against golden vectors produced by the reference's own functions
(G6 clock splines, G7 clock_reward, G8 get_full_state; generator tools/refprobe/gen_golden_env.py), and checks the
physics restatement through invariants (the MuJoCo-backed step itself is parity-unpinned)."""
import os

import numpy as np
import pytest

from oracle import sim as S


def test_g6_clock_splines(golden_dir):
    g = np.load(os.path.join(golden_dir, "g6_clock_splines.npz"))
    for c in range(int(g["n_cases"])):
        swing, stance, relax, mode, inc, freq = g[f"c{c}_params"]
        vals, pl = S.clock_eval(swing, stance, relax, int(mode), bool(inc), int(freq), g[f"c{c}_phases"])
        assert abs(pl - float(g[f"c{c}_phaselen"])) < 1e-12
        np.testing.assert_allclose(vals, g[f"c{c}_vals"], atol=1e-12, err_msg=f"case {c} mode {mode} inc {inc}")


def test_g7_clock_reward(golden_dir):
    g = np.load(os.path.join(golden_dir, "g7_clock_reward.npz"))
    e = S.OracleEnv()
    for c in range(int(g["n_cases"])):
        p = f"c{c}_"
        r = e.clock_reward_eval(g[p + "qpos"], g[p + "qvel"], g[p + "scal"], g[p + "foot_vel"], g[p + "rotvel"], g[p + "tacc"],
                                g[p + "torque"], g[p + "prev_torque"], g[p + "prev_action"], g[p + "action"])
        assert abs(r - float(g[p + "reward"])) < 1e-12, (c, r, float(g[p + "reward"]))
    e1 = S.OracleEnv(reward_kind=1)             # early_clock_reward (clock_rewards.py:119-223)
    for c in range(int(g["n_cases"])):
        p = f"c{c}_"
        r = e1.clock_reward_eval(g[p + "qpos"], g[p + "qvel"], g[p + "scal"], g[p + "foot_vel"], g[p + "rotvel"], g[p + "tacc"],
                                 g[p + "torque"], g[p + "prev_torque"], g[p + "prev_action"], g[p + "action"])
        assert abs(r - float(g[p + "reward_early"])) < 1e-12, (c, r)
    e2 = S.OracleEnv(reward_kind=2)             # max_vel_clock_reward (clock_rewards.py:416-480)
    for c in range(int(g["n_cases"])):
        p = f"c{c}_"
        r = e2.clock_reward_eval(g[p + "qpos"], g[p + "qvel"], g[p + "scal"], g[p + "foot_vel"], g[p + "rotvel"], g[p + "tacc"],
                                 g[p + "torque"], g[p + "prev_torque"], g[p + "prev_action"], g[p + "action"])
        assert abs(r - float(g[p + "reward_max_vel"])) < 1e-12, (c, r)


def test_g8_full_state(golden_dir):
    g = np.load(os.path.join(golden_dir, "g8_full_state.npz"))
    e = S.OracleEnv()
    for c in range(int(g["n_cases"])):
        p = f"c{c}_"
        phase, phaselen, speed, side, orient, pz, th = g[p + "scal"]
        ints = e.get("ints"); ints[1] = phase; e.set("ints", ints)
        e.set("phaselen", [phaselen]); e.set("speed", [speed]); e.set("side_speed", [side]); e.set("orient_add", [orient])
        e.set("so_height", [pz - th]); e.set("so_quat", g[p + "quat"]); e.set("so_rotvel", g[p + "rotvel"])
        e.set("so_tvel", g[p + "tvel"]); e.set("so_tacc", g[p + "tacc"]); e.set("so_mpos", g[p + "mpos"])
        e.set("so_mvel", g[p + "mvel"]); e.set("so_jpos", g[p + "jpos"]); e.set("so_jvel", g[p + "jvel"])
        e.set("motor_noise", g[p + "mnoise"]); e.set("joint_noise", g[p + "jnoise"])
        np.testing.assert_allclose(e.obs(), g[p + "obs"], atol=1e-12)


def test_g9_pd_law(golden_dir):
    """pd_input_step of the reference binary: tau = ff + P (pTarget - q) + D (dTarget - qd), no clamp (bit-exact)."""
    g = np.load(os.path.join(golden_dir, "g9_pd_input.npz"))
    tau = g["ff"] + g["P"] * (g["pTarget"] - g["q"]) + g["D"] * (g["dTarget"] - g["v"])
    np.testing.assert_array_equal(tau, g["tau"])


def test_g10_core_sim_safety(golden_dir):
    """cassie_core_sim_step of the reference binary on 400 random multi-joint states (zones, clamp, radio gate)."""
    g = np.load(os.path.join(golden_dir, "g10_core_sim.npz"))
    for k in range(g["q"].shape[0]):
        out = S.core_safety(g["q"][k], g["v"][k], g["cmd"][k], g["radio"][k])
        np.testing.assert_allclose(out, g["tau"][k], rtol=1e-12, atol=1e-10)


def test_g10b_core_sim_coupled_pitch_knee_zone(golden_dir):
    """The coupled zone of cassie_core_sim_step (hip pitch + knee < -135 deg): 300 states of the reference binary, 269 inside."""
    g = np.load(os.path.join(golden_dir, "g10b_core_sim_coupled.npz"))
    inside = 0
    for k in range(g["q"].shape[0]):
        out = S.core_safety(g["q"][k], g["v"][k], g["cmd"][k], 1.0)
        np.testing.assert_allclose(out, g["tau"][k], rtol=1e-12, atol=1e-10)
        inside += int(g["q"][k, 2] + g["q"][k, 3] < -0.75 * np.pi or g["q"][k, 7] + g["q"][k, 8] < -0.75 * np.pi)
    assert inside > 200


def test_g11_estimator_lite_vs_reference_filter(golden_dir):
    """The 7 filtered estimator outputs: our closed-form estimator-lite vs the reference's state_output_step run on this
    simulator's own sensor stream (a falling robot under random actions).  The reference filter is a stateful black box,
    so this pins the FRAMES and offsets with stated tolerances, not bit parity; the 39 pass-through fields are exact."""
    g = np.load(os.path.join(golden_dir, "g11_estimator.npz"))
    def q2m(q):
        w, x, y, z = q
        return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                         [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                         [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
    Rm = np.stack([q2m(q) for q in g["quat"]])
    # acceleration: restated exactly (round 2): accelerometer - R^T (0, 0, 9.806) - w x (w x r_imu); the round-1 form (g = 9.81, no
    # centripetal term) was off by 0.09 .. 0.24 m/s^2 on the same stream
    r_imu = np.array([0.03155, 0.0, -0.07996])
    tacc = g["acc"] - np.einsum("nji,j->ni", Rm, np.array([0, 0, 9.806])) - np.cross(g["gyro"], np.cross(g["gyro"], r_imu))
    tvel = np.einsum("nji,nj->ni", Rm, g["v_world"])
    assert np.abs(tacc - g["ref_tacc"]).max() < 1e-3                        # m/s^2
    assert np.abs(tvel[:, :2] - g["ref_tvel"][:, :2]).mean(0).max() < 0.06  # m/s, x and y (z is leg-kinematics based in the filter)
    assert abs(np.mean(g["z"] - 0.0818 - g["ref_height"])) < 0.005 and np.std(g["z"] - 0.0818 - g["ref_height"]) < 0.03
    np.testing.assert_allclose(g["ref_quat"], g["quat"], atol=1e-12); np.testing.assert_allclose(g["ref_rotvel"], g["gyro"], atol=1e-12)


def test_philox_known_answer():
    # Philox4x32-10 known-answer vectors (Random123 kat_vectors): counter 0, key 0 -> 6627e8d5 ...
    import ctypes
    # our stream uses counter (ctr, env, TAG, 0); check determinism + distinctness, and the raw round function below
    a = [S.philox(0, 0, i) for i in range(4)]
    assert len(set(a)) == 4 and a == [S.philox(0, 0, i) for i in range(4)]
    assert S.philox(1, 0, 0) != a[0] and S.philox(0, 1, 0) != a[0]


def test_physics_invariants():
    e = S.OracleEnv(dyn_rand=False)
    e.phys_forward()
    assert e.violation() < 8e-3                     # init pose baked in the binary closes the loops to ~6.6 mm
    q = e.get("qpos"); q[2] = 2.0; e.set("qpos", q)
    e.set("damping", np.zeros(32))
    e0 = e.energy()
    e.phys_step(np.zeros(10), 400)                  # 0.2 s of free fall, no damping, no contact
    assert abs(e.energy() - e0) / abs(e0) < 2e-3
    assert abs(e.get("qpos")[2] - (2.0 - 0.5 * 9.81 * 0.2 ** 2)) < 2e-3
    assert e.violation() < 1e-5                     # soft constraints pull the loops closed
    for j0 in (3, 10, 24):                          # quaternions stay normalised
        assert abs(np.linalg.norm(e.get("qpos")[j0:j0 + 4]) - 1) < 1e-12


def test_contact_complementarity_and_weight():
    """Robot dropped on its feet with a stiff PD hold: contact forces are non-negative, only on penetrating points,
    and the vertical impulse over a window matches the momentum change + weight."""
    e = S.OracleEnv(dyn_rand=False)
    e.reset()
    fz = []
    for _ in range(10):
        e.step(np.zeros(10))
        ints = e.get("ints"); ncon, nefc = int(ints[3]), int(ints[4])
        assert nefc == 12 + 4 * ncon                     # no limit rows in this scenario
        ff = e.get("efc_force")[:nefc].reshape(2, -1) if ncon % 2 == 0 else None
        if ff is not None:                               # leg-major rows: [6 eq | 4 per contact] per leg, same count per leg here
            assert (ff[:, 6:] >= 0).all()
        assert (e.get("foot_force")[[2, 5]] >= 0).all()
        fz.append(e.get("foot_force")[[2, 5]].sum())
    assert max(fz) > 100.0                           # feet carry a sizeable share of the 327 N weight


def test_env_episode_runs_and_terminates():
    e = S.OracleEnv(seed=3)
    obs = e.reset()
    assert obs.shape == (50,) and np.isfinite(obs).all()
    rng = np.random.RandomState(0)
    done, t = 0, 0
    while not done and t < 400:
        obs, r, done = e.step(rng.randn(10) * 0.2)
        assert np.isfinite(obs).all() and np.isfinite(r) and -0.5 < r < 1.0
        t += 1
    assert done in (1, 2)


def test_g14_reset_draws_and_randomisation_tables(golden_dir):
    """G14: the reference's CassieEnv.__init__ + reset run on a recording CassieSim stand-in with intercepted RNG
    (tools/refprobe/gen_golden_dynrand.py).  (1) the reference's draw order and ranges have the structure the env code
    assumes; (2) the oracle's reset, re-derived draw by draw from its Philox stream with the REFERENCE's range factors,
    reproduces the oracle's parameters exactly (the GPU kernel is tied to the oracle by tests/test_gpu_env.py)."""
    g = np.load(os.path.join(golden_dir, "g14_dynrand.npz"))
    dr = g["draws"]                                            # rows: kind (0 uniform, 1 randint), lo, hi, count
    assert len(dr) == 145 and int(dr[:, 3].sum()) == 159
    c = g["consts"]                                            # damping lo/hi, mass lo/hi, fric lo/hi, roll, pitch, enc noise, speed, side speed
    np.testing.assert_allclose(c, [0.3, 5.0, 0.5, 1.5, 0.4, 1.1, 0.03, 0.03, 0.01, -0.3, 4.0, -0.3, 0.3])
    # ---- (1) structure of the reference's reset
    assert tuple(dr[0][:3]) == (0, -0.3, 4.0) and tuple(dr[1][:3]) == (0, -0.3, 0.3)                 # speed, side speed
    assert dr[2][0] == 1 and dr[2][1] == 0 and dr[2][2] == np.floor(g["phase"][1])                  # random.randint(0, floor(phaselen))
    dd, dm = g["default_damping"], g["default_mass"]
    # quirk (cassie.py:595-597, 619-622): the right leg's ranges are the LEFT leg's list appended a second time, i.e. they are
    # built from the left defaults; on cassie.xml left and right defaults are equal, so the env code uses each dof's own default
    dd_ref = np.concatenate([dd[:19], dd[6:19]]); dm_ref = np.concatenate([dm[:14], dm[2:14]])
    d_lo, d_hi = dr[3:35, 1] / dd_ref, dr[3:35, 2] / dd_ref
    vary = np.array([1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 1, 0, 1], bool)                                  # heel-spring (k=9), plantar-rod (k=11) fixed
    exp_lo = np.concatenate([np.ones(6), np.where(vary, 0.3, 1.0), np.where(vary, 0.3, 1.0)])
    exp_hi = np.concatenate([np.ones(6), np.where(vary, 5.0, 1.0), np.where(vary, 5.0, 1.0)])
    np.testing.assert_allclose(d_lo, exp_lo, rtol=1e-12); np.testing.assert_allclose(d_hi, exp_hi, rtol=1e-12)
    assert tuple(dr[35][1:3]) == (0.0, 0.0)                                                          # world body
    np.testing.assert_allclose(dr[36:61, 1] / dm_ref[1:], 0.5, rtol=1e-12); np.testing.assert_allclose(dr[36:61, 2] / dm_ref[1:], 1.5, rtol=1e-12)
    assert np.all(dr[61:136, 1] == dr[61:136, 2])                                                    # 75 COM draws with delta = 0: consumed, no effect
    np.testing.assert_allclose(dr[136:139, 1:3], [[0.4, 1.1], [1e-4, 5e-4], [1e-4, 2e-4]])
    fr = g["set_friction"].reshape(-1, 3)
    assert np.all(fr == fr[0])                                                                       # one triple for every geom
    np.testing.assert_allclose(dr[139:141, 1:3], [[-0.03, 0.03], [-0.03, 0.03]])                     # roll, pitch
    assert tuple(dr[141]) == (0, -0.01, 0.01, 10) and tuple(dr[142]) == (0, -0.01, 0.01, 6)          # encoder noise
    assert tuple(dr[143][:3]) == (0, -0.3, 4.0) and tuple(dr[144][:3]) == (0, -0.3, 0.3)             # commands redrawn after the settle step
    assert int(g["n_set_const"][0]) == 1 and int(g["n_step_pd"][0]) == 1                             # one sim.set_const (:660), one step_pd (:665)
    # floor quaternion = euler2quat(z=0, y=pitch, x=roll) of the two slope draws; our closed form (cx cy, cy sx, cx sy, sx sy)
    def unit(k): return ((k + 1) * 0.61803398875) % 1.0
    roll, pitch = -0.03 + 0.06 * unit(139), -0.03 + 0.06 * unit(140)                                 # draws 139, 140 (all earlier draws are scalars)
    cx, sx, cy, sy = np.cos(roll / 2), np.sin(roll / 2), np.cos(pitch / 2), np.sin(pitch / 2)
    np.testing.assert_allclose(g["set_geom_quat"][:4], [cx * cy, cy * sx, cx * sy, sx * sy], atol=1e-12)
    # clipping at 0 and the clock from the FIRST speed draw
    assert np.all(g["set_damping"] >= 0) and np.all(g["set_mass"] >= 0)
    sp0 = -0.3 + 4.3 * unit(0)
    total = (0.9 - 0.25 / 3.0 * abs(sp0)) / 2
    np.testing.assert_allclose(g["swing_stance"], [(0.30 + (0.40 / 3) * abs(sp0)) * total, (0.70 - (0.40 / 3) * abs(sp0)) * total], rtol=1e-12)
    # ---- (2) the oracle's reset, draw by draw, with the reference's factors
    for seed, eid in ((3, 0), (3, 5), (11, 2)):
        e = S.OracleEnv(dyn_rand=True, seed=seed, env_id=eid)
        d0, m0 = e.get("damping").copy(), e.get("mass").copy()
        e.reset()
        u01 = lambda k: ((S.philox(seed, eid, k) >> 8) + 0.5) / 16777216.0
        uni = lambda k, a, b: a + (b - a) * u01(k)
        np.testing.assert_allclose(e.get("damping"), [d0[d] * (d_lo[d] + (d_hi[d] - d_lo[d]) * u01(3 + d)) for d in range(32)], rtol=1e-12)
        exp_m = [0.0] + [m0[b] * uni(35 + b, 0.5, 1.5) for b in range(1, 26)]
        np.testing.assert_allclose(e.get("mass"), exp_m, rtol=1e-12)
        np.testing.assert_allclose(e.get("friction")[0], uni(61, *dr[136, 1:3]), rtol=1e-12)
        r_, p_ = uni(64, -0.03, 0.03), uni(65, -0.03, 0.03)
        cx, sx, cy, sy = np.cos(r_ / 2), np.sin(r_ / 2), np.cos(p_ / 2), np.sin(p_ / 2)
        np.testing.assert_allclose(e.get("floor_quat"), [cx * cy, cy * sx, cx * sy, sx * sy], atol=1e-12)
        np.testing.assert_allclose(e.get("motor_noise"), [uni(66 + u, -0.01, 0.01) for u in range(10)], rtol=1e-12)
        np.testing.assert_allclose(e.get("joint_noise"), [uni(76 + k, -0.01, 0.01) for k in range(6)], rtol=1e-12)
        np.testing.assert_allclose([e.get("speed")[0], e.get("side_speed")[0]], [uni(82, -0.3, 4.0), uni(83, -0.3, 0.3)], rtol=1e-12)
        assert int(e.get("ints")[5]) == 84                                                           # draws consumed by one reset


def test_g16_update_speed_and_reset_for_test(golden_dir):
    """G16 (next row f3): CassieEnv.update_speed on 200 random (first speed, phase, new command) states and the call order /
    field values of CassieEnv.reset_for_test, both from the reference (tools/refprobe/gen_golden_evalapi.py)."""
    g = np.load(os.path.join(golden_dir, "g16_eval_api.npz"))
    e = S.OracleEnv(dyn_rand=True, seed=0, env_id=0)
    for sp0, ph0, pl0, ns, nside, sp, side, swing, stance, ph1 in g["update_speed"]:
        e.set_command(sp0, int(ph0))
        assert abs(e.get("phaselen")[0] - pl0) < 1e-9
        e.update_speed(ns, nside)
        assert abs(e.get("speed")[0] - sp) < 1e-12 and abs(e.get("side_speed")[0] - side) < 1e-12
        assert abs(e.get("phaselen")[0] - (2 * swing + 2 * stance) * 40) < 1e-9
        assert int(e.get("ints")[1]) == int(ph1)                                   # phase rescaled with int() truncation: bit exact
    # reset_for_test: one step_pd with the stale targets FIRST, then defaults + set_const, then the floor
    assert list(g["rft_order"]) == ["step_pd", "damping", "mass", "ipos", "friction", "set_const", "geom_quat:floor"]
    ph, tm, cnt, oadd, sp, side, swing, stance, plen, padd = g["rft_scalars"]
    assert (ph, tm, cnt, oadd, sp, swing, stance, plen, padd) == (0, 0, 0, 0, 0, 0.15, 0.25, 32.0, 1) and side == 0.2   # side speed is NOT reset
    assert str(g["rft_stance_mode"][0]) == "grounded" and np.all(g["rft_noise"] == 0)
    np.testing.assert_allclose(g["rft_floor"], [1, 0, 0, 0])
    e = S.OracleEnv(dyn_rand=True, seed=5, env_id=1)
    e.reset()
    d0 = S.OracleEnv(dyn_rand=False, seed=0, env_id=0)
    e.set("side_speed", 0.2)
    for _ in range(3):
        e.step(np.zeros(10))
    obs = e.reset_for_test()
    ints = e.get("ints")
    assert (ints[0], ints[1], ints[2]) == (0, 0, 0) and e.get("speed")[0] == 0 and e.get("side_speed")[0] == 0.2 and e.get("orient_add")[0] == 0
    assert abs(e.get("phaselen")[0] - 32.0) < 1e-12
    np.testing.assert_allclose(e.get("damping"), d0.get("damping")); np.testing.assert_allclose(e.get("mass"), d0.get("mass"))
    np.testing.assert_allclose(e.get("friction"), d0.get("friction")); np.testing.assert_allclose(e.get("floor_quat"), [1, 0, 0, 0])
    assert np.all(e.get("motor_noise") == 0) and np.all(e.get("joint_noise") == 0)
    assert np.all(np.isfinite(obs)) and obs[48] == 0 and abs(obs[49] - 0.2) < 1e-12 and obs[46] == 0 and obs[47] == 1     # commands, clock at phase 0


def test_g16_reset_for_test_full_reset(golden_dir):
    """reset_for_test(full_reset=True) (tools/eval_perturb.py:31): call order full_reset -> defaults -> set_const -> floor, and the
    observation the REAL get_full_state builds from reset_cassie_state's constants (side speed is NOT reset), from the reference."""
    g = np.load(os.path.join(golden_dir, "g16_eval_api.npz"))
    assert list(g["rft_full_order"]) == ["full_reset", "damping", "mass", "ipos", "friction", "set_const", "geom_quat:floor"]
    e = S.OracleEnv(dyn_rand=True, seed=3, env_id=2)
    e.reset()
    for _ in range(4):
        e.step(np.full(10, 0.3))
    e.set("side_speed", -0.1)
    e.apply_force([50.0, 0, 0, 0, 0, 0])
    obs = e.reset_for_test(full_reset=True)
    np.testing.assert_allclose(obs, g["rft_full_obs"], atol=1e-12)
    ph, tm, cnt, oadd, sp, side, swing, stance, plen = g["rft_full_scalars"]
    ints = e.get("ints")
    assert (ints[0], ints[1], ints[2]) == (tm, ph, cnt) and e.get("speed")[0] == sp and e.get("side_speed")[0] == side
    assert abs(e.get("phaselen")[0] - plen) < 1e-12
    d0 = S.OracleEnv(dyn_rand=False, seed=0, env_id=0)
    np.testing.assert_allclose(e.get("qpos"), d0.get("qpos")); assert np.all(e.get("qvel") == 0)       # init pose (set_const follows)
    # the wrench was cleared by the full reset: the next steps equal those of an env that never had one
    e2 = S.OracleEnv(dyn_rand=True, seed=3, env_id=2)
    e2.reset()
    for _ in range(4):
        e2.step(np.full(10, 0.3))
    e2.set("side_speed", -0.1)
    e2.reset_for_test(full_reset=True)
    for _ in range(3):
        o1 = e.step_basic(np.zeros(10)); o2 = e2.step_basic(np.zeros(10))
    np.testing.assert_allclose(o1, o2, atol=1e-12)


def test_apply_force_on_the_pelvis():
    """CassieSim.apply_force([fx, fy, 0, 0, 0, 0], "cassie-pelvis") (tools/eval_perturb.py:62): momentum balance on the whole
    robot.  While airborne (first substeps after set_const the feet are off the ground) the only external forces are gravity and
    the wrench, so the COM acceleration must be g + f / M; and a pure torque must not move the COM at all."""
    e = S.OracleEnv(dyn_rand=False, seed=0, env_id=0)
    e.reset_for_test(full_reset=True)
    M = e.get("mass").sum()
    def com_vel(env):
        return np.asarray(env.com_velocity())
    for xfrc in ([60.0, -25.0, 10.0, 0, 0, 0], [0, 0, 0, 3.0, -2.0, 4.0], [40.0, 10.0, 0, 1.0, 0.5, -0.5]):
        e = S.OracleEnv(dyn_rand=False, seed=0, env_id=0)
        e.reset_for_test(full_reset=True)
        e.apply_force(xfrc)
        v0 = com_vel(e)
        n = 6
        e.phys_step(np.zeros(10), n)
        a = (com_vel(e) - v0) / (n * 0.0005)
        # semi-implicit Euler with a configuration-dependent M conserves momentum to O(h): 5e-5 relative here
        np.testing.assert_allclose(a, np.array([0, 0, -9.81]) + np.array(xfrc[:3]) / M, atol=5e-4)


def test_g16_step_basic_bookkeeping(golden_dir):
    """CassieEnv.step_basic (cassie.py:498-521): 70 calls after reset_for_test on the 32-step clock: time, phase wrap (phase > phaselen),
    counter, and simrate step_pd calls per step, from the reference; the oracle reproduces the bookkeeping."""
    g = np.load(os.path.join(golden_dir, "g16_eval_api.npz"))
    assert int(g["step_basic_pd_calls"][0]) == 70 * 50
    e = S.OracleEnv(dyn_rand=True, seed=2, env_id=0)
    e.reset(); e.reset_for_test()
    for k in range(40):
        obs = e.step_basic(np.zeros(10))
        ints = e.get("ints")
        assert (ints[0], ints[1], ints[2]) == tuple(g["step_basic"][k]), k
        if e.get("qpos")[2] < 0.3:
            break
    assert np.all(np.isfinite(obs))


def test_g11c_estimator_height_model(golden_dir):
    """G11c: the reference filter's height output (pelvis.position[2] - terrain.height, observation entry 0) on our sensor stream while
    a 1000-iteration policy trained with this build (a predecessor of trained_models/r02_cassie_v0_clock) walks for 3 s (6000 substeps of 2 kHz), with the true pelvis z and the lowest sole
    height per substep (tools/refprobe/gen_golden_estheight.py).  The build's model, height = z - L with L a first-order low-pass
    (EST_TAU) of the lowest sole height started at EST_L0 by state_output_setup, stays within 1.2 cm of the reference over the whole
    stream (the former constant offset z - 0.0818 is off by up to 8.4 cm on it); the oracle env implements exactly this recursion."""
    g = np.load(os.path.join(golden_dir, "g11c_estimator_height.npz"))
    z, sole, ref = g["z"].astype(np.float64), g["sole_low"].astype(np.float64), g["ref_height"].astype(np.float64)
    tau, L0 = 0.86, 0.126
    L = L0; err = np.zeros(len(z))
    for i in range(len(z)):
        L += 0.0005 / tau * (sole[i] - L)
        err[i] = z[i] - L - ref[i]
    assert np.abs(err[10:]).max() < 0.012 and np.abs(err[10:]).mean() < 0.008          # first 10 substeps: the filter's own start-up
    assert np.abs(z - 0.0818 - ref)[10:].max() > 0.08                                   # what the constant offset did on this stream
    e = S.OracleEnv(dyn_rand=False, seed=1)
    assert e.get("est_L")[0] == L0
    e.reset()
    rng = np.random.RandomState(0)
    for t in range(3):
        e.set("pd_target", rng.randn(10) * 0.1 + np.array([0.0045, 0, 0.4973, -1.1997, -1.5968] * 2))
        for sub in range(20):
            L_old, sole_old, z_old = e.get("est_L")[0], e.get("snap_sole")[0], e.get("snap_pz")[0]
            e.substep()
            L_new = L_old + 0.0005 / tau * (sole_old - L_old)
            assert abs(e.get("est_L")[0] - L_new) < 1e-15 and abs(e.get("so_height")[0] - (z_old - L_new)) < 1e-15
    assert -0.02 < e.get("snap_sole")[0] < 0.2
    L_before = e.get("est_L")[0]
    e.reset()                                                                            # a training reset keeps the filter state
    assert abs(e.get("est_L")[0] - L_before) < 0.01 * abs(L_before) + 1e-3
    e.reset_for_test(full_reset=True)                                                    # state_output_setup restarts it
    assert e.get("est_L")[0] == L0


def test_g11b_estimator_lite_on_a_walking_stream(golden_dir):
    """G11b: the reference's state_output_step on our sensor stream while a TRAINED policy stands / steps for 200 env steps
    (tools/refprobe/probe_estimator_walk.py).  Velocity and acceleration frames hold on this stream too.  The height entry shows
    why a constant offset is not enough: once the feet are on the ground the reference filter's terrain estimate converges (time
    constant about 1 s) and its height tends to the pelvis z itself; the height model that follows this is pinned by G11c."""
    g = np.load(os.path.join(golden_dir, "g11b_estimator_walk.npz"))
    def q2m(q):
        w, x, y, z = q
        return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                         [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                         [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
    Rm = np.stack([q2m(q) for q in g["quat"]])
    # acceleration: the exact restatement (see G11) holds on this stream too: 5e-5 m/s^2 max (the round-1 form: 0.03 mean, 0.12 max)
    tacc = g["acc"] - np.einsum("nji,j->ni", Rm, np.array([0, 0, 9.806])) - np.cross(g["gyro"], np.cross(g["gyro"], np.array([0.03155, 0.0, -0.07996])))
    tvel = np.einsum("nji,nj->ni", Rm, g["v_world"])
    assert np.abs(tacc - g["ref_tacc"]).max() < 1e-3                   # m/s^2, signal std 1.0 .. 2.5
    assert np.abs(tvel[:, :2] - g["ref_tvel"][:, :2]).mean(0).max() < 0.08
    d = g["z"] - g["ref_height"]
    assert abs(d[120:].mean()) < 0.02 and d[0] > 0.08                  # reference: offset decays to ~0 on the ground
    assert np.abs(g["z"] - 0.0818 - g["ref_height"]).mean() < 0.09     # a constant offset is off by centimetres here (see G11c)


def test_g17_traj_env_reset_and_ref_state(golden_dir):
    """G17 (next row f1, env half): CassieTrajEnv.get_ref_state on 300 random (phase, phaselen, speed, counter) of the reference
    (bit-exact in the oracle: same table, same arithmetic), and CassieTrajEnv.reset on a recording sim: set_const -> set_qpos ->
    set_qvel -> step_pd, the first speed is randint(0, 40) / 10, the pose written is get_ref_state(start phase) with that speed."""
    g = np.load(os.path.join(golden_dir, "g17_traj_env.npz"))
    for k in range(len(g["inp"])):
        ph, pl, sp, cn = g["inp"][k]
        q, v = S.traj_ref_state(ph, pl, sp, int(cn))
        assert np.array_equal(q, g["pos"][k]) and np.array_equal(v, g["vel"][k]), k
    assert list(g["reset_order"]) == ["set_const", "set_qpos", "set_qvel", "step_pd"]
    (a0, b0, v0), (a1, b1, v1) = g["reset_randint"]
    assert (a0, b0) == (0, 40) and (a1, b1) == (0, np.floor(g["reset_phaselen"][0])) and int(g["reset_phase"][0]) == int(v1)
    assert abs(g["reset_phaselen"][0] - g["reset_phaselen"][1]) < 1e-12                 # the clock is built from the randint speed
    np.testing.assert_array_equal(g["reset_qpos"], g["reset_ref_qpos"]); np.testing.assert_array_equal(g["reset_qvel"], g["reset_ref_qvel"])
    assert -0.3 <= g["reset_speed_after"][0] <= 4.0 and g["reset_speed_after"][2] == 0
    np.testing.assert_allclose(g["offset"], [0.0045, 0.0, 0.4973, -1.1997, -1.5968] * 2)   # no_delta: the PD offset is Cassie-v0's
    # the oracle env in CassieTraj mode: after reset the robot sits one substep after the reference pose of its start phase, and the
    # first speeds it saw are multiples of 0.1 in [0, 4] (the commanded speed itself is redrawn from U[-0.3, 4] afterwards)
    for seed in range(6):
        e = S.OracleEnv(dyn_rand=True, seed=seed, env_id=3, env_kind=1)
        e.reset()
        ph, pl = int(e.get("ints")[1]), e.get("phaselen")[0]
        # recover the first speed from the clock: phaselen = 40 (0.9 - 0.25 / 3 |s|)
        s0 = (0.9 - pl / 40.0) * 3 / 0.25
        assert abs(s0 * 10 - round(s0 * 10)) < 1e-9 and -1e-9 <= s0 <= 4.0 + 1e-9
        q, v = S.traj_ref_state(ph, pl, s0, 0)
        qe, ve = e.get("qpos"), e.get("qvel")
        assert np.abs(qe[[2, 7, 8, 9, 14, 20]] - q[[2, 7, 8, 9, 14, 20]]).max() < 5e-3       # one 0.5 ms substep away from the pose
        assert abs(qe[0] - q[0]) < 5e-3 and abs(qe[1]) < 1e-3
        assert -0.3 <= e.get("speed")[0] <= 4.0
    e0 = S.OracleEnv(dyn_rand=True, seed=1, env_id=3)                                      # Cassie-v0 for contrast: the init pose
    e0.reset()
    assert abs(e0.get("qpos")[2] - 1.01) < 2e-3


def _flags(e):
    return int(e.get("ints")[8])


def test_full_collision_set_of_cassie_xml():
    """The oracle instantiates every constraint cassie.xml can produce (cassie.xml:18-35,87,101,119-144 + all limited joints), not only
    what the HIP kernel keeps per leg: pelvis sphere and hip-pitch capsules against the floor, the 3 x 3 left-right capsule pairs
    (condim 1), every active limit, every penetrating capsule end; `sat` says when a forward pass needed more than the kernel's caps."""
    SAT_CONTACTS, SAT_LIMITS, SAT_BODY_FLOOR, SAT_LEG_LEG = 1, 2, 4, 8
    # --- standing: nothing beyond the caps
    e = S.OracleEnv(dyn_rand=False); e.reset()
    for _ in range(3):
        e.step(np.zeros(10))
    assert _flags(e) == 0 and int(e.get("ints")[9]) == 0
    # --- pelvis sphere (r = 0.15) pushed into the floor plane (z = -0.01): a pyramidal floor contact on body 1 that pushes the pelvis up
    e = S.OracleEnv(dyn_rand=False)
    q = e.get("qpos"); q[2] = 0.10
    q[[9, 23]] = 1.3; q[[14, 28]] = -2.4                                   # fold the legs up so that the feet are not the deeper contact
    e.set("qpos", q); e.phys_forward()
    ints = e.get("ints")
    assert _flags(e) & SAT_BODY_FLOOR
    acc_with = e.get("qacc")[2]
    q[2] = 0.30; e2 = S.OracleEnv(dyn_rand=False); e2.set("qpos", q); e2.phys_forward()      # same pose, sphere clear of the floor
    assert acc_with > e2.get("qacc")[2] + 50.0                              # the sphere contact decelerates the fall strongly
    nefc = int(ints[4]); f, ty = e.get("efc_force")[:nefc], e.get("efc_type")[:nefc]
    assert (f[ty != 0] >= 0).all() and (ty == 2).sum() == 4 * int(ints[3])
    # --- crossed legs: both hip rolls inward -> a left-right capsule pair penetrates; one frictionless row that pushes the legs apart
    e = S.OracleEnv(dyn_rand=False)
    q = e.get("qpos"); q[2] = 1.5; q[7] = -0.10; q[21] = 0.10
    e.set("qpos", q); e.phys_forward()
    ints = e.get("ints")
    assert int(ints[9]) >= 1 and not (_flags(e) & SAT_LEG_LEG)               # up to 3 pairs are rows in the kernel too: not a saturation
    nefc, n1 = int(ints[4]), int(ints[9])
    f = e.get("efc_force")[:nefc]
    assert (f[nefc - n1:] >= 0).all() and f[nefc - n1:].sum() > 10 and (e.get("efc_type")[nefc - n1:nefc] == 2).all()      # unilateral rows, placed last, pushing
    qa = e.get("qacc")
    q0 = e.get("qpos"); q0[7] = -0.08; q0[21] = 0.08
    e3 = S.OracleEnv(dyn_rand=False); e3.set("qpos", q0); e3.phys_forward()
    assert int(e3.get("ints")[9]) == 0 and not (_flags(e3) & SAT_LEG_LEG)
    assert qa[6] - e3.get("qacc")[6] > 1.0 and e3.get("qacc")[19] - qa[19] > 1.0      # left roll accelerates outward (+), right roll outward (-)
    q0[7] = -0.15; q0[21] = 0.15                                            # legs pushed through each other: 6 pairs, beyond the kernel's 3 rows
    e4 = S.OracleEnv(dyn_rand=False); e4.set("qpos", q0); e4.phys_forward()
    assert int(e4.get("ints")[9]) > 3 and _flags(e4) & SAT_LEG_LEG
    # --- two limits of one leg at once (knee and foot beyond their ranges): two limit rows, flagged
    e = S.OracleEnv(dyn_rand=False)
    q = e.get("qpos"); q[2] = 1.5; q[14] = -2.9; q[20] = -2.5
    e.set("qpos", q); e.phys_forward()
    assert _flags(e) & SAT_LIMITS
    nefc = int(e.get("ints")[4])
    assert nefc >= 12 + 2
    # --- more than two penetrating capsule ends on a leg (feet pressed 8 cm into the floor with the toes down): flagged, all instantiated
    e = S.OracleEnv(dyn_rand=False)
    q = e.get("qpos"); q[2] = 0.78
    e.set("qpos", q); e.phys_forward()
    ints = e.get("ints")
    if int(ints[3]) > 4:
        assert _flags(e) & SAT_CONTACTS
    assert int(ints[4]) == 12 + 4 * int(ints[3]) + int(ints[9]) or (_flags(e) & SAT_LIMITS)


def test_g22_phase_command_profile(golden_dir):
    """Row f4, command_profile="phase" (golden G22, tools/refprobe/gen_golden_phase.py): (a) state space 55 / clock indices / mirror list;
    (b) the reference's draw order and ranges at reset (plain and "library"), and the oracle's reset re-derived draw by draw from its
    Philox stream with those ranges; (c) the clock splines for phase-profile durations; (d) get_full_state with the 9-entry command tail."""
    from apex_amd import vecenv as V
    g = np.load(os.path.join(golden_dir, "g22_phase_profile.npz"))
    # (a)
    assert int(g["obs_size"]) == V.OBS_DIM_PHASE == 55 and list(g["clock_inds"]) == V.CLOCK_INDS
    np.testing.assert_allclose(g["mirrored_obs"], V.MIRRORED_OBS_PHASE); np.testing.assert_allclose(g["mirrored_acts"], V.MIRRORED_ACTS)
    assert str(g["plain_reward_func"]) == "clock" and str(g["library_reward_func"]) == "clock"
    # (b) structure of the reference's reset (kind 0 uniform, 1 randint inclusive, 2 choice of n)
    dp, dl = g["plain_draws"], g["library_draws"]
    assert [tuple(r[:3]) for r in dp[:5]] == [(0, -0.3, 4.0), (0, -0.3, 0.3), (1, 1, 50), (1, 1, 30), (2, 0, 3)]
    assert dp[5][0] == 1 and dp[5][1] == 0 and dp[5][2] == np.floor(g["plain_phase"][1])
    assert [tuple(r[:3]) for r in dl[:6]] == [(0, -0.3, 4.0), (0, -0.3, 0.3), (1, 0, 30), (1, 3, 6), (1, 2, 8), (2, 0, 3)]
    assert dl[6][0] == 1 and dl[6][2] == np.floor(g["library_phase"][1])
    assert tuple(dp[-2][:3]) == (0, -0.3, 4.0) and tuple(dp[-1][:3]) == (0, -0.3, 0.3)             # commands redrawn after the settle step
    np.testing.assert_allclose(g["plain_phase"][1], (2 * g["plain_swing_stance"].sum()) * 40)         # phaselen = (2 swing + 2 stance) * FREQ
    # the oracle's reset, draw by draw
    for cp, seed, eid in ((1, 3, 0), (1, 9, 4), (2, 3, 1), (2, 5, 7)):
        e = S.OracleEnv(dyn_rand=False, seed=seed, env_id=eid, command_profile=cp)
        obs = e.reset()
        assert obs.shape == (55,)
        u01 = lambda k: ((S.philox(seed, eid, k) >> 8) + 0.5) / 16777216.0
        ri = lambda k, n: (S.philox(seed, eid, k) * n) >> 32
        if cp == 1:
            swing, stance, pick, k = (1 + ri(2, 50)) / 100, (1 + ri(3, 30)) / 100, ri(4, 3), 5
        else:
            total, ratio = (3 + ri(3, 4)) / 10, (2 + ri(4, 7)) / 10
            swing, stance, pick, k = total * ratio, total - total * ratio, ri(5, 3), 6
        np.testing.assert_allclose(e.get("swing_stance"), [swing, stance], rtol=1e-12)
        plen = (2 * swing + 2 * stance) * 40
        assert abs(e.get("phaselen")[0] - plen) < 1e-12 and int(e.get("ints")[1]) == ri(k, int(np.floor(plen)) + 1)       # random.randint(0, floor(phaselen))
    # (c)
    for c in range(int(g["n_cases"])):
        swing, stance, relax, mode, inc, freq = g[f"c{c}_params"]
        vals, pl = S.clock_eval(swing, stance, relax, int(mode), bool(inc), int(freq), g[f"c{c}_phases"])
        assert abs(pl - float(g[f"c{c}_phaselen"])) < 1e-12
        np.testing.assert_allclose(vals, g[f"c{c}_vals"], atol=1e-12, err_msg=f"case {c}")
    # (d)
    for k in range(int(g["n_obs_cases"])):
        p = f"o{k}_"
        phase, phaselen, speed, side, orient, pz, th, swing, stance, mode = g[p + "scal"]
        e = S.OracleEnv(command_profile=1, stance_mode=int(mode))
        ints = e.get("ints"); ints[1] = phase; e.set("ints", ints)
        e.set("phaselen", [phaselen]); e.set("speed", [speed]); e.set("side_speed", [side]); e.set("orient_add", [orient]); e.set("swing_stance", [swing, stance])
        e.set("so_height", [pz - th]); e.set("so_quat", g[p + "quat"]); e.set("so_rotvel", g[p + "rotvel"])
        e.set("so_tvel", g[p + "tvel"]); e.set("so_tacc", g[p + "tacc"]); e.set("so_mpos", g[p + "mpos"])
        e.set("so_mvel", g[p + "mvel"]); e.set("so_jpos", g[p + "jpos"]); e.set("so_jvel", g[p + "jvel"])
        e.set("motor_noise", g[p + "mnoise"]); e.set("joint_noise", g[p + "jnoise"])
        np.testing.assert_allclose(e.obs(), g[p + "obs"], atol=1e-12)


def test_solver_tolerance_knob():
    """mjOption.tolerance (MuJoCo default 1e-8): with the early exit of mj_solPGS the solver uses fewer than the 50 sweeps the oracle and the
    HIP kernel always run, and a standing-start trajectory under a fixed action sequence stays within 1e-3 of the 50-sweep one."""
    rs = np.random.RandomState(5)
    acts = rs.randn(12, 10) * 0.1
    out = []
    for tol in (0.0, 1e-8):
        e = S.OracleEnv(dyn_rand=False, seed=3, env_id=0)
        e.reset_for_test()
        sv = e.get("solver"); sv[3] = tol; e.set("solver", sv)
        assert e.get("solver")[3] == tol and 1.0 < e.get("solver")[4] < 10.0            # meaninertia = trace(M(qpos0)) / nv
        n0 = e.get("solver")[1:3].copy()
        obs = [e.step(a)[0] for a in acts]
        s = e.get("solver")
        out.append((np.asarray(obs), (s[1] - n0[0]) / (s[2] - n0[1])))
    assert out[0][1] == 50.0 and 5.0 < out[1][1] < 50.0, (out[0][1], out[1][1])
    assert np.abs(out[0][0] - out[1][0]).max() < 1e-3
