"""ctypes front-end of the fp64 C++ oracle (oracle/_build/liboracle.so).  TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# ORC_REAL=float selects the fp32 CONTROL build of the same sources (oracle/Makefile target `f32`: `double` -> `float`, single-precision literals): every array
# that crosses the C API is then float32.  One variant per process; the parity tests run the control in a subprocess (tests/test_oracle_env.py).
_F32 = os.environ.get("ORC_REAL") == "float"
_REAL = np.float32 if _F32 else np.float64
_CR = C.c_float if _F32 else C.c_double
_SO = os.path.join(_HERE, "_build", "liboracle_f32.so" if _F32 else "liboracle.so")
_lib = None


# MuJoCo body ids of cassie.xml (world = 0): the name argument of CassieSim.apply_force
BODY_NAMES = ["world", "cassie-pelvis"] + [side + "-" + n for side in ("left", "right") for n in (
    "hip-roll", "hip-yaw", "hip-pitch", "achilles-rod", "knee", "knee-spring", "shin", "tarsus", "heel-spring", "foot-crank", "plantar-rod", "foot")]

def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            subprocess.check_call(["make", "-C", _HERE] + (["f32"] if _F32 else []))
        L = C.CDLL(_SO)
        L.orc_env_new.restype = C.c_void_p
        L.orc_env_new.argtypes = [C.c_int] * 7 + [C.c_uint64, C.c_uint32]
        L.orc_env_free.argtypes = [C.c_void_p]
        L.orc_env_reset.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_env_step.restype = C.c_int
        L.orc_env_step.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_env_substep.argtypes = [C.c_void_p]
        L.orc_env_update_speed.argtypes = [C.c_void_p, _CR, _CR]
        L.orc_env_step_basic.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_env_reset_for_test.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.orc_env_apply_force.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_env_apply_force_body.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.orc_env_set_kind.argtypes = [C.c_void_p, C.c_int]
        L.orc_env_set_command_profile.argtypes = [C.c_void_p, C.c_int]
        L.orc_env_set_input_profile.argtypes = [C.c_void_p, C.c_int]
        L.orc_traj_ref_state.argtypes = [_CR, _CR, _CR, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_env_set_command.argtypes = [C.c_void_p, _CR, C.c_int]
        L.orc_env_obs.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_phys_step.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.orc_com_velocity.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_phys_forward.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_constraint_violation.restype = _CR
        L.orc_constraint_violation.argtypes = [C.c_void_p]
        L.orc_momentum.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_inverse_dynamics.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_efc_rows.restype = C.c_int
        L.orc_efc_rows.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_total_energy.restype = _CR
        L.orc_total_energy.argtypes = [C.c_void_p]
        L.orc_env_get.restype = C.c_int
        L.orc_env_get.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p]
        L.orc_env_set.restype = C.c_int
        L.orc_env_set.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p]
        L.orc_env_set_const.argtypes = [C.c_void_p]
        L.orc_flops.restype = C.c_uint64
        L.orc_flops.argtypes = [C.c_int]
        L.orc_env_set_kernel_caps.argtypes = [C.c_void_p, C.c_int]
        L.orc_env_set_hfield.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, _CR, _CR, _CR]
        L.orc_floor_query.argtypes = [C.c_void_p, _CR, _CR, C.c_void_p]
        L.orc_clock_eval.argtypes = [_CR, _CR, _CR, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                     C.c_void_p, C.c_void_p]
        L.orc_clock_reward_eval.restype = _CR
        L.orc_clock_reward_eval.argtypes = [C.c_void_p] * 11
        L.orc_core_safety.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, _CR, C.c_void_p]
        L.orc_philox.restype = C.c_uint32
        L.orc_philox.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32]
        L.orc_est_create.restype = C.c_void_p
        L.orc_est_destroy.argtypes = [C.c_void_p]
        L.orc_est_setup.argtypes = [C.c_void_p]
        L.orc_est_step.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_est_heel_residual.restype = _CR
        L.orc_est_heel_residual.argtypes = [_CR] * 4 + [C.c_void_p]
        L.orc_est_mldivide23.argtypes = [C.c_void_p] * 3
        L.orc_est_hfilter_step.argtypes = [C.c_void_p, C.c_void_p] + [_CR] * 5
        L.orc_est_zfilter_step.argtypes = [C.c_void_p, C.c_void_p] + [_CR] * 4
        L.orc_rollout_bench.restype = _CR
        L.orc_rollout_bench.argtypes = [C.c_int, C.c_int, C.c_int, C.c_uint64, _CR]
        _lib = L
    return _lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class OracleEnv:
    def __init__(self, simrate=50, dyn_rand=True, reward_kind=0, stance_mode=0, incentive=True, max_traj_len=400,
                 pgs_iters=50, seed=0, env_id=0, env_kind=0, command_profile=0, est_lifetime=169, input_profile=0):
        self.h = lib().orc_env_new(simrate, int(dyn_rand), reward_kind, stance_mode, int(incentive), max_traj_len,
                                   pgs_iters, seed, env_id)
        self.obs_dim = (46 if input_profile == 0 else 21) + (4 if command_profile == 0 else 9)
        if input_profile:        # 1 = min
            lib().orc_env_set_input_profile(self.h, int(input_profile))
        if env_kind:
            lib().orc_env_set_kind(self.h, int(env_kind))
        if command_profile:      # 1 phase, 2 phase with the "library" draws
            lib().orc_env_set_command_profile(self.h, int(command_profile))
        if est_lifetime != 169:
            self.set("est_age", [0, est_lifetime])

    def __del__(self):
        try:
            if getattr(self, "h", None):
                lib().orc_env_free(self.h)
                self.h = None
        except Exception:      # interpreter shutdown: the module globals may already be gone
            pass

    def reset(self):
        obs = np.zeros(self.obs_dim, dtype=_REAL)
        lib().orc_env_reset(self.h, _ptr(obs))
        return obs

    def step(self, action):
        a = np.ascontiguousarray(action, dtype=_REAL)
        obs = np.zeros(self.obs_dim, dtype=_REAL); rew = np.zeros(1, dtype=_REAL)
        done = lib().orc_env_step(self.h, _ptr(a), _ptr(obs), _ptr(rew))
        return obs, float(rew[0]), done

    def substep(self):
        lib().orc_env_substep(self.h)

    def step_basic(self, action):
        a = np.ascontiguousarray(action, dtype=_REAL); obs = np.zeros(self.obs_dim, dtype=_REAL)
        lib().orc_env_step_basic(self.h, _ptr(a), _ptr(obs))
        return obs

    def update_speed(self, speed, side_speed=0.0):
        lib().orc_env_update_speed(self.h, float(speed), float(side_speed))

    def set_command(self, speed0, phase):
        lib().orc_env_set_command(self.h, float(speed0), int(phase))

    def reset_for_test(self, full_reset=False):
        obs = np.zeros(self.obs_dim, dtype=_REAL)
        lib().orc_env_reset_for_test(self.h, _ptr(obs), int(bool(full_reset)))
        return obs

    def apply_force(self, xfrc, body_name="cassie-pelvis"):
        """CassieSim.apply_force (cassiemujoco.py:99-103): one row of mjData.xfrc_applied; one pushed body at a time."""
        x = np.ascontiguousarray(xfrc, dtype=_REAL)
        assert x.shape == (6,)
        lib().orc_env_apply_force_body(self.h, _ptr(x), BODY_NAMES.index(body_name))

    def obs(self):
        o = np.zeros(self.obs_dim, dtype=_REAL)
        lib().orc_env_obs(self.h, _ptr(o))
        return o

    def com_velocity(self):
        v = np.zeros(3, dtype=_REAL)
        lib().orc_com_velocity(self.h, _ptr(v))
        return v

    def phys_step(self, ctrl, n=1):
        c = np.ascontiguousarray(ctrl, dtype=_REAL)
        lib().orc_phys_step(self.h, _ptr(c), n)

    def phys_forward(self, ctrl=None):
        c = np.zeros(10, dtype=_REAL) if ctrl is None else np.ascontiguousarray(ctrl, dtype=_REAL)
        lib().orc_phys_forward(self.h, _ptr(c))

    def get(self, name):
        buf = np.zeros(128, dtype=_REAL)
        n = lib().orc_env_get(self.h, name.encode(), _ptr(buf))
        if n < 0:
            raise KeyError(name)
        return buf[:n].copy()

    def set(self, name, val):
        v = np.ascontiguousarray(np.asarray(val, dtype=_REAL).reshape(-1))
        buf = np.zeros(128, dtype=_REAL); buf[:v.size] = v
        n = lib().orc_env_set(self.h, name.encode(), _ptr(buf))
        if n < 0:
            raise KeyError(name)

    def clock_reward_eval(self, qpos, qvel, scal, foot_vel, rotvel, tacc, torque, prev_torque, prev_action, action):
        arrs = [np.ascontiguousarray(a, dtype=_REAL) for a in
                (qpos, qvel, scal, foot_vel, rotvel, tacc, torque, prev_torque, prev_action, action)]
        return lib().orc_clock_reward_eval(self.h, *[_ptr(a) for a in arrs])

    def kernel_caps(self, on=True):
        """Instantiate only the constraint rows the HIP kernel instantiates (first 2 floor contacts / first limit per leg); the
        saturation flags (`get("ints")[8]`) still report everything cassie.xml would add."""
        lib().orc_env_set_kernel_caps(self.h, int(bool(on)))
        return self

    def set_hfield(self, data, size=(50.0, 50.0, 0.15)):
        """CassieSim("cassie_hfield.xml").set_hfield_data(data) (util/eval.py:73-76): data [nrow, ncol] raw elevations (x size[2]), rows along y."""
        if data is None:
            lib().orc_env_set_hfield(self.h, None, 0, 0, 0.0, 0.0, 0.0); return self
        d = np.ascontiguousarray(data, dtype=np.float32)
        lib().orc_env_set_hfield(self.h, _ptr(d), d.shape[0], d.shape[1], float(size[0]), float(size[1]), float(size[2]))
        return self

    def floor_query(self, x, y):
        out = np.zeros(4, dtype=_REAL)
        lib().orc_floor_query(self.h, float(x), float(y), _ptr(out))
        return out[0], out[1:]

    def set_const(self):
        lib().orc_env_set_const(self.h)

    def violation(self):
        return lib().orc_constraint_violation(self.h)

    def momentum(self):
        """dict of the tree's total momentum: p [3], L about the pelvis origin [3] (world axes), origin, mass, com, foot capsule (centre, axis) x 2"""
        out = np.zeros(25, dtype=_REAL)
        lib().orc_momentum(self.h, _ptr(out))
        return dict(p=out[0:3], L=out[3:6], o=out[6:9], mass=float(out[9]), com=out[10:13], foot=[(out[13:16], out[16:19]), (out[19:22], out[22:25])])

    def inverse_dynamics(self, qacc):
        """M(q) qacc + bias - passive of the unconstrained tree at the current (qpos, qvel): what actuators and constraint forces must add up to"""
        qa = np.ascontiguousarray(qacc, dtype=_REAL); out = np.zeros(32, dtype=_REAL)
        lib().orc_inverse_dynamics(self.h, _ptr(qa), _ptr(out))
        return out

    def efc_rows(self):
        """(J [nefc, 32], type [nefc]) of the constraint rows of the most recent phys_forward (0 equality, 1 limit, 2 contact)"""
        J = np.zeros((200, 32), dtype=_REAL); ty = np.zeros(200, dtype=np.int32)
        n = lib().orc_efc_rows(self.h, _ptr(J), ty.ctypes.data_as(C.c_void_p))
        return J[:n].copy(), ty[:n].copy()

    def energy(self):
        return lib().orc_total_energy(self.h)


def clock_eval(swing, stance, relax, mode, inc, freq, phases):
    ph = np.ascontiguousarray(phases, dtype=_REAL)
    out = np.zeros((len(ph), 4), dtype=_REAL); pl = np.zeros(1, dtype=_REAL)
    lib().orc_clock_eval(swing, stance, relax, mode, int(inc), freq, len(ph), _ptr(ph), _ptr(out), _ptr(pl))
    return out, float(pl[0])


def core_safety(q, qd, cmd, radio=1.0):
    a = [np.ascontiguousarray(x, dtype=_REAL) for x in (q, qd, cmd)]
    out = np.zeros(10, dtype=_REAL)
    lib().orc_core_safety(_ptr(a[0]), _ptr(a[1]), _ptr(a[2]), float(radio), _ptr(out))
    return out


def philox(seed, env, ctr, dom=0):
    """one Philox4x32-10 draw of stream (seed, env, dom): dom 0 = the per-step command draws (ctr = the env's running counter), dom 1 = reset draws (ctr = 128 * episode + k)"""
    return lib().orc_philox(seed, env, ctr, dom)


def rollout_bench(n_envs, n_steps, threads, seed=0, act_std=0.2):
    return lib().orc_rollout_bench(n_envs, n_steps, threads, seed, act_std)


def traj_ref_state(phase, phaselen, speed, counter=0):
    """CassieTrajEnv.get_ref_state of the oracle (walking trajectory, simrate 50) -> (qpos[35], qvel[32])."""
    q, v = np.zeros(35, dtype=_REAL), np.zeros(32, dtype=_REAL)
    lib().orc_traj_ref_state(float(phase), float(phaselen), float(speed), int(counter), _ptr(q), _ptr(v))
    return q, v


def count_flops(n_steps=40, act_std=0.2, seed=0):
    """Instrumented floating-point operation count of the restatement: multiplies + adds per ENV STEP (50 substeps + env
    logic is negligible), averaged over `n_steps` steps of a random-action rollout with resets, structural zeros of the dense loops skipped."""
    rng = np.random.RandomState(seed)
    e = OracleEnv(seed=seed, env_id=0)
    e.reset()
    lib().orc_flops(1)
    done_steps = 0
    for _ in range(n_steps):
        _, _, d = e.step(act_std * rng.randn(10))
        done_steps += 1
        if d:
            f = lib().orc_flops(0); e.reset(); lib().orc_flops(1); lib().orc_flops(0)      # do not count the reset's own forward passes
            # restore the running total (reset zeroed it): add back
            count_flops._acc = getattr(count_flops, "_acc", 0) + f
    total = getattr(count_flops, "_acc", 0) + lib().orc_flops(1)
    count_flops._acc = 0
    return total / done_steps


class StateEstimator:
    """The restated reference estimator on its own (oracle/cassie_estimator.h): step(mpos10, jpos6, quat4, gyro3, acc3) ->
    dict(pos, vel, tacc, terrain, foot_rel, foot_force, heel, lm_iters)."""
    def __init__(self):
        self.h = lib().orc_est_create()

    def __del__(self):
        try:
            lib().orc_est_destroy(self.h)
        except Exception:
            pass

    def setup(self):
        lib().orc_est_setup(self.h)

    def step(self, sensors26):
        x = np.ascontiguousarray(sensors26, dtype=_REAL)
        assert x.size == 26
        o = np.zeros(33, dtype=_REAL)
        lib().orc_est_step(self.h, _ptr(x), _ptr(o))
        return dict(pos=o[0:3], vel=o[3:6], tacc=o[6:9], terrain=o[9], foot_rel=o[10:16].reshape(2, 3), foot_force=o[16:22].reshape(2, 3), heel=o[22:24], lm_iters=int(o[24]), foot_quat=o[25:33].reshape(2, 4))


def heel_residual(knee, shin, tarsus, heel):
    g = np.zeros(4, dtype=_REAL)
    r = lib().orc_est_heel_residual(float(knee), float(shin), float(tarsus), float(heel), _ptr(g))
    return r, g


def hfilter_step(x, P, zL, zR, fl, fr, acc):
    """one step of the estimator's horizontal filter, oracle/cassie_estimator.cpp hfilter_step (the C++ the env runs) -> (x', P')"""
    xx = np.array(x, dtype=_REAL).reshape(6).copy(); PP = np.array(P, dtype=_REAL).reshape(36).copy()
    lib().orc_est_hfilter_step(_ptr(xx), _ptr(PP), float(zL), float(zR), float(fl), float(fr), float(acc))
    return xx, PP.reshape(6, 6)


def zfilter_step(x, P, zL, zR, fl, fr):
    xx = np.array(x, dtype=_REAL).reshape(5).copy(); PP = np.array(P, dtype=_REAL).reshape(25).copy()
    lib().orc_est_zfilter_step(_ptr(xx), _ptr(PP), float(zL), float(zR), float(fl), float(fr))
    return xx, PP.reshape(5, 5)


def mldivide23(M, tau):
    Mm = np.ascontiguousarray(M, dtype=_REAL).reshape(6); t = np.ascontiguousarray(tau, dtype=_REAL); x = np.zeros(3, dtype=_REAL)
    lib().orc_est_mldivide23(_ptr(Mm), _ptr(t), _ptr(x))
    return x
