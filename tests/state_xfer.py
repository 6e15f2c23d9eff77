"""Teacher forcing for the GPU parity tests: copy the WHOLE state of fp64 oracle envs into the HIP env through the C ABI's field setters, so that
one env step of both can be compared with tolerances that do not grow with the length of the rollout (tests/test_gpu_env.py)."""
import numpy as np
import torch

EST_REC = 168


def est_record_from_oracle(e):
    """oracle StateOutput -> the kernel's env-major estimator record (apex_amd/csrc/estimator_lane.h): lane r < 6 owns Px[6] Py[6] Pz[6] xX xY xZ
    at [24 r, 24 r + 21); lane 6 owns heelL heelR terrain inited at [144, 148)"""
    rec = np.zeros(EST_REC)
    hx, hP = e.get("est_hx").reshape(2, 6), e.get("est_hP").reshape(2, 6, 6)
    zx, zP = e.get("est_zx"), e.get("est_zP").reshape(5, 5)
    for r in range(6):
        rec[24 * r: 24 * r + 6] = hP[0, r]; rec[24 * r + 6: 24 * r + 12] = hP[1, r]
        if r < 5:
            rec[24 * r + 12: 24 * r + 17] = zP[r]; rec[24 * r + 20] = zx[r]
        rec[24 * r + 18] = hx[0, r]; rec[24 * r + 19] = hx[1, r]
    rec[144:146] = e.get("est_heel"); rec[146] = e.get("est_terrain")[0]; rec[147] = e.get("est_flags")[0]
    return rec


def est_from_record(rec):
    """the inverse view, for comparisons: dict(hx [2, 6], hP [2, 6, 6], zx [5], zP [5, 5], heel [2], terrain, inited)"""
    rec = np.asarray(rec, dtype=np.float64)
    hx = np.zeros((2, 6)); hP = np.zeros((2, 6, 6)); zx = np.zeros(5); zP = np.zeros((5, 5))
    for r in range(6):
        hP[0, r] = rec[24 * r: 24 * r + 6]; hP[1, r] = rec[24 * r + 6: 24 * r + 12]
        hx[0, r] = rec[24 * r + 18]; hx[1, r] = rec[24 * r + 19]
        if r < 5:
            zP[r] = rec[24 * r + 12: 24 * r + 17]; zx[r] = rec[24 * r + 20]
    return dict(hx=hx, hP=hP, zx=zx, zP=zP, heel=rec[144:146].copy(), terrain=rec[146], inited=rec[147])


def _floor_frame(fq):
    w, x, y, z = fq
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)], [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                  [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
    n = R[:, 2]
    t1 = np.array([0.0, 1.0, 0.0]) if abs(n[1]) < 0.5 else np.array([0.0, 0.0, 1.0])
    t1 = t1 - n * (n @ t1); t1 /= np.linalg.norm(t1)
    return np.concatenate([n, t1, np.cross(n, t1)])


def oracle_to_kernel(genv, oenvs, model_params=True):
    """set every persistent field of the HIP env (all envs) from the list of oracle envs; fewer oracle envs than kernel envs are tiled over the batch (kernel env i
    takes oracle env i % len(oenvs): the tail of the batch stays a defined state that nobody compares)"""
    assert 0 < len(oenvs) <= genv.n_envs
    tile = np.arange(genv.n_envs) % len(oenvs)
    G = lambda name: np.stack([e.get(name) for e in oenvs])[tile]
    T = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float32)
    if model_params:
        for name in ("mass", "damping", "friction", "motor_noise", "joint_noise"):
            genv.set_field(name, T(G(name)))
        genv.set_field("floor", T(np.stack([_floor_frame(e.get("floor_quat")) for e in oenvs])[tile]))
        genv.set_field("set_const")                                     # invweights in fp32 from the copied masses (mj_setConst)
    for name in ("qpos", "qvel", "qacc_warm", "pd_target", "tq_fifo", "so_mpos", "so_mvel", "so_torque", "so_jpos", "so_jvel", "so_quat", "so_rotvel", "so_tvel", "so_tacc",
                 "so_height", "prev_action", "prev_torque", "jenc_x"):
        genv.set_field(name, T(G(name)))
    genv.set_field("menc", T(G("menc_hist")))
    genv.set_field("jenc_y", T(G("jenc_y").reshape(-1, 6, 3)[:, :, :2].reshape(-1, 12)))
    genv.set_field("snap", T(np.concatenate([G("snap_mpos"), G("snap_jpos"), G("snap_quat"), G("snap_gyro"), G("snap_acc")], 1)))
    genv.set_field("foot_prev", T(G("foot_pos_prev")))
    genv.set_field("foot_vel", T(np.concatenate([G("l_foot_vel"), G("r_foot_vel")], 1)))
    cmd = np.stack([np.concatenate([e.get("speed"), e.get("side_speed"), e.get("orient_add"), e.get("swing_stance"), e.get("phaselen"), e.get("stance_mode")]) for e in oenvs])[tile]
    genv.set_field("cmd", T(cmd))
    genv.set_field("est", T(np.stack([est_record_from_oracle(e) for e in oenvs])[tile]))
    ints = genv.get_field("ints").cpu().numpy()
    for i in range(genv.n_envs):
        e = oenvs[tile[i]]
        oi = e.get("ints"); pr = e.get("enc_primed"); pa = e.get("phase_add")
        ints[i, 0:3] = oi[0:3]; ints[i, 3] = oi[5]
        ints[i, 4] = int(pr[0]) | int(pr[1]) << 1 | int(oi[6]) << 2 | int(oi[7]) << 3 | 16 | int(pa[1]) << 5 | int(pa[0] > 1.25) << 6      # (half phase, phase_add = 1.5: tools/test_commands.py:86)
        ints[i, 5] = int(e.get("est_age")[0])
        ints[i, 9] = int(e.get("episode")[0])
    genv.set_field("ints", T(ints))


# every field of the oracle's Env that persists from one env step to the next (oracle/cassie_capi.cpp `field`), in an order that can be replayed with set()
ORACLE_STATE_FIELDS = ("mass", "damping", "friction", "floor_quat", "motor_noise", "joint_noise", "qpos", "qvel", "qacc_warm", "pd_target", "pd_P", "pd_D", "tq_fifo",
                       "menc_hist", "jenc_x", "jenc_y", "enc_primed", "snap_mpos", "snap_jpos", "snap_quat", "snap_gyro", "snap_acc", "so_mpos", "so_mvel", "so_torque", "so_jpos",
                       "so_jvel", "so_quat", "so_rotvel", "so_tvel", "so_tacc", "so_height", "est_heel", "est_hx", "est_hP", "est_zx", "est_zP", "est_terrain", "est_flags",
                       "l_foot_vel", "r_foot_vel", "foot_pos_prev", "prev_action", "prev_torque", "speed", "side_speed", "orient_add", "swing_stance", "stance_mode", "phase_add", "est_age", "episode")


def oracle_state(e):
    """the persistent state of one oracle env as {field: float64 array} (+ the integer words)"""
    d = {k: np.asarray(e.get(k), dtype=np.float64).copy() for k in ORACLE_STATE_FIELDS}
    d["ints"] = np.asarray(e.get("ints"), dtype=np.float64)[:8].copy()
    return d


def oracle_load_state(e, d, set_const=True):
    """teacher forcing oracle -> oracle (the fp32 control build takes the fp64 oracle's state): model parameters, set_const in the target's own arithmetic
    (like the kernel, which recomputes its invweights in fp32), then every persistent field, the integer words and the clock tables"""
    for k in ORACLE_STATE_FIELDS:
        e.set(k, d[k])
    if set_const:
        e.set_const()
    e.set("ints", d["ints"])
    e.set("clock_rebuild", [0.0])


# ---- fixed tolerances of ONE env step from an identical state, on the (env, step) pairs whose active constraint-row sets were the same in all 50 substeps of
# both sides (row-set hash: I_ROWSET in the kernel, Env::rowset_hash in the oracle).  They are the fp32 level: the fp32 CONTROL (the oracle's own sources compiled in
# fp32 against the fp64 oracle from identical states, tests/test_oracle_env.py::test_fp32_control_of_the_parity_tolerances) shows acceleration max 9.2e-2 / p99 5.3e-2
# m/s^2, motor velocity 4.2e-2 rad/s (the 9-tap FIR on truncated encoder counts: one count is 0.03 rad/s on the foot drive), reward 9.2e-4 on 960 walking pairs; the
# kernel 8.6e-2 / 6.3e-2, 5.4e-2, 1.9e-3 on 3150.  max tolerance = about 1.5 - 2 x those.
TF_NAMES = ["height+quat", "motor pos", "tvel", "gyro", "motor vel", "tacc", "joint pos", "joint vel", "clock+cmd", "reward", "qpos", "qvel"]
TF_TOL_SAME = np.array([1e-4, 2e-4, 1e-3, 2e-3, 1e-1, 1.5e-1, 2e-3, 6e-2, 1e-5, 3e-3, 5e-5, 1e-2])
TF_TOL_SAME_P99 = np.array([5e-6, 1e-4, 2e-4, 1e-3, 5e-2, 8e-2, 1e-4, 2e-2, 1e-5, 1.2e-3, 2e-5, 6e-3])
TF_MAX_DIFFERING_FRACTION = 0.04       # kernel: 1.6 % of 3200 walking pairs; fp32 control: 1.5 % of 960
