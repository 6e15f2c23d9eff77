"""Scenarios of the teacher-forced parity tests (round 5): every env configuration / action regime that round 4 still compared free-running with tolerances growing like
(t + 1) - safety zones, the coupled pitch-knee zone, the early / max_vel rewards, the evaluation API, CassieTraj-v0, the phase command profile, height fields, the min
input profile, the fractional phase_add.  One definition serves both sides:
  * tests/test_gpu_env.py::test_teacher_forced_scenario        kernel vs fp64 oracle, kernel state overwritten with the oracle's before EVERY step (tests/state_xfer.py)
  * tests/test_oracle_env.py::test_fp32_control_of_the_scenarios   the oracle's own sources compiled in fp32 vs the fp64 oracle from the same states (CPU suite): shows that
    the fixed tolerances are the fp32 level of each scenario (and derives the two tolerances state_xfer.py did not have: drive torque, min-profile foot entries)
A scenario = oracle constructor arguments, kernel constructor arguments, a preparation of the oracle envs (the kernel only needs what is NOT state: the terrain), an
action generator, the stepper (env.step or env.step_basic) and the number of steps."""
import numpy as np

N_ORACLE = 8      # oracle envs per scenario (the kernel batch is 64: kernel env i carries the state of oracle env i % 8)


def _terrain(kind, n=41, seed=0):
    """small synthetic height fields over [-4, 4] x [-4, 4] (0.2 m cells like cassie_hfield.xml's 500 x 500 over 100 m): raw values, scale 0.15"""
    xs = np.linspace(-4, 4, n)
    X, Y = np.meshgrid(xs, xs)                       # rows along y, columns along x
    if kind == "slope":
        return (0.5 + 0.6 * X / 4).astype(np.float32)                     # 2.25 % grade in x after the 0.15 scale
    if kind == "noise":                                                   # like terrains/noise*.npy: values in [0, 0.25] -> bumps up to 3.75 cm
        return (0.25 * np.random.RandomState(seed).rand(n, n)).astype(np.float32)
    return (0.3 + 0.3 * np.sin(1.3 * X) * np.cos(0.9 * Y)).astype(np.float32)      # rolling hills, +-4.5 cm


HF_SIZE = (4.0, 4.0, 0.15)


class Scenario:
    def __init__(self, name, okw, gkw, act, n_steps, rng_seed, stepper="step", prep=None, hfield=None, obs_groups=None, min_profile=False, differing_max=0.10, reaches=None, prep_kernel=None):
        self.name, self.okw, self.gkw, self.act, self.n_steps, self.rng_seed, self.stepper = name, okw, gkw, act, n_steps, rng_seed, stepper
        self.prep, self.hfield, self.obs_groups, self.min_profile, self.differing_max, self.reaches, self.prep_kernel = prep, hfield, obs_groups, min_profile, differing_max, reaches, prep_kernel

    def make_oracle(self, S, n=N_ORACLE):
        envs = [S.OracleEnv(env_id=i, **self.okw) for i in range(n)]
        if self.hfield is not None:
            for e in envs: e.set_hfield(_terrain(self.hfield), HF_SIZE)
        for e in envs: e.reset()
        if self.prep is not None:
            self.prep(envs)
        return envs

    def make_kernel(self, CassieVecEnv, n=64):
        g = CassieVecEnv(n_envs=n, **self.gkw)
        if self.hfield is not None:
            g.set_hfield(_terrain(self.hfield), HF_SIZE)
        g.reset()
        if self.prep_kernel is not None:
            self.prep_kernel(g)
        return g

    def step_oracle(self, e, a):
        if self.stepper == "step":
            return e.step(a)
        return e.step_basic(a), 0.0, 0


# ---- action generators (t, rng, n) -> [n, 10]
def _act_safety(t, rng, n):      # large hip-roll / hip-yaw / foot targets: cassie_core_sim_step's soft joint-limit zones (golden G10)
    a = rng.randn(n, 10) * 0.05
    a[:, [0, 1, 5, 6]] += rng.choice([-0.5, 0.5], size=(n, 4))       # roll / yaw far past +-0.2
    a[:, [4, 9]] += 0.9                                               # foot past its -35 deg limit
    return a


def _act_coupled(t, rng, n):     # deep-crouch targets: hip pitch + knee below -135 deg, the coupled zone (golden G10b)
    a = rng.randn(n, 10) * 0.03
    a[:, [2, 7]] -= 1.1; a[:, [3, 8]] -= 1.3
    return a


def _gauss(std):
    return lambda t, rng, n: rng.randn(n, 10) * std


def _zeros(t, rng, n):
    return np.zeros((n, 10))


# ---- preparations of the oracle envs (after reset)
def _prep_eval(envs):            # reset_for_test + update_speed(1.0): the evaluation harnesses' starting point (util/eval.py, cassie.py:682-768)
    for e in envs:
        e.reset_for_test(); e.update_speed(1.0)


def _prep_phase_add(envs):       # tools/test_commands.py:86: phase_add 1.5 on every other env
    for i, e in enumerate(envs):
        e.reset_for_test(); e.set("speed", [1.6]); e.set("phase_add", [1.5 if i % 2 == 0 else 1.0, 0])


def _prep_hfield(envs):          # the robot starts at the origin: shift / lift it so that the feet meet the terrain at different places, 1-6 cm into the surface
    rng = np.random.RandomState(3)
    for e in envs:
        e.reset_for_test()
        q = e.get("qpos").copy()
        q[0] = rng.uniform(-2.5, 2.5); q[1] = rng.uniform(-2.5, 2.5)
        hh, _ = e.floor_query(q[0], q[1])
        q[2] = 1.0 + hh + rng.uniform(-0.06, 0.0)
        e.set("qpos", q.astype(np.float32).astype(np.float64)); e.set("qvel", np.zeros(32)); e.set("qacc_warm", np.zeros(32))


def _push_forces(n):            # a pelvis push per env, 150 - 300 N in a random horizontal direction (tools/eval_perturb.py:62 pushes up to a few hundred N)
    rng = np.random.RandomState(17)
    ang = rng.uniform(0, 2 * np.pi, n); mag = rng.uniform(150.0, 300.0, n)
    f = np.zeros((n, 6)); f[:, 0] = mag * np.cos(ang); f[:, 1] = mag * np.sin(ang)
    return f


def _prep_push(envs):            # falling robots: wild actions (the scenario's generator) + a standing pelvis push; the external wrench is not part of a reset on either side
    f = _push_forces(len(envs))
    for e, x in zip(envs, f):
        e.apply_force(x)


def _prep_push_kernel(g):
    import torch
    f = _push_forces(N_ORACLE)
    g.apply_force(torch.tensor(f[np.arange(g.n_envs) % N_ORACLE], dtype=torch.float32))


def _saturated(envs):            # forward passes beyond the lane map's row caps seen so far (oracle/cassie_phys.h SatFlag, accumulated): the complete-row path ran
    return int(sum(int(e.get("ints")[8]) != 0 for e in envs))


def _zones_hit(envs):
    q = np.stack([e.get("so_mpos") for e in envs])
    return int(((q[:, 0] > 0.2) | (q[:, 0] < -0.112) | (np.abs(q[:, 1]) > 0.234) | (q[:, 5] < -0.2) | (q[:, 5] > 0.112) | (np.abs(q[:, 6]) > 0.234) | (q[:, 4] > -0.761)).sum())


def _coupled_hit(envs):
    q = np.stack([e.get("so_mpos") for e in envs])
    return int(((q[:, 2] + q[:, 3] < -0.75 * np.pi) | (q[:, 7] + q[:, 8] < -0.75 * np.pi)).sum())


G50 = [slice(0, 5), slice(5, 15), slice(15, 18), slice(18, 21), slice(21, 31), slice(31, 34), slice(34, 40), slice(40, 46), slice(46, 50)]
G55 = G50[:8] + [slice(46, 55)]
# min input profile (cassie.py:829-837): foot positions 0:6, pelvis quat 6:10, rotational velocity 10:13, foot orientations 13:21, clock + commands from 21
GMIN = lambda dim: [slice(0, 6), slice(6, 10), slice(10, 13), slice(13, 21), slice(21, dim)]

SCENARIOS = [
    Scenario("safety_zones", dict(dyn_rand=False, seed=4), dict(dynamics_randomization=False, seed=4), _act_safety, 6, 1, reaches=_zones_hit, differing_max=0.25),
    Scenario("coupled_zone", dict(dyn_rand=False, seed=9), dict(dynamics_randomization=False, seed=9), _act_coupled, 8, 3, reaches=_coupled_hit, differing_max=0.25),
    Scenario("early_clock", dict(dyn_rand=True, seed=6, reward_kind=1), dict(dynamics_randomization=True, seed=6, reward="early_clock"), _gauss(0.1), 8, 2),
    Scenario("max_vel_clock", dict(dyn_rand=True, seed=6, reward_kind=2), dict(dynamics_randomization=True, seed=6, reward="max_vel_clock"), _gauss(0.1), 8, 2),
    Scenario("eval_step", dict(dyn_rand=True, seed=13), dict(dynamics_randomization=True, seed=13), _gauss(0.1), 6, 4, prep=_prep_eval),
    Scenario("eval_step_basic", dict(dyn_rand=True, seed=17), dict(dynamics_randomization=True, seed=17), _gauss(0.1), 6, 8, stepper="step_basic", prep=_prep_eval),
    Scenario("cassie_traj", dict(dyn_rand=True, seed=17, env_kind=1), dict(dynamics_randomization=True, seed=17, env_name="CassieTraj-v0"), _gauss(0.1), 8, 3),
    Scenario("phase_clock", dict(dyn_rand=True, seed=13, command_profile=1), dict(seed=13, command_profile="phase", reward="clock"), _gauss(0.15), 8, 1, obs_groups=G55),
    Scenario("phase_library", dict(dyn_rand=True, seed=13, command_profile=2), dict(seed=13, command_profile="phase", reward="library_clock"), _gauss(0.15), 8, 1, obs_groups=G55),
    Scenario("hfield_slope", dict(dyn_rand=False, seed=4), dict(dynamics_randomization=False, seed=4, max_traj_len=1000), _zeros, 5, 0, stepper="step_basic", prep=_prep_hfield, hfield="slope", differing_max=0.3),
    Scenario("hfield_noise", dict(dyn_rand=False, seed=4), dict(dynamics_randomization=False, seed=4, max_traj_len=1000), _zeros, 5, 0, stepper="step_basic", prep=_prep_hfield, hfield="noise", differing_max=0.3),
    Scenario("hfield_hills", dict(dyn_rand=False, seed=4), dict(dynamics_randomization=False, seed=4, max_traj_len=1000), _zeros, 5, 0, stepper="step_basic", prep=_prep_hfield, hfield="hills", differing_max=0.3),
    Scenario("min_clock", dict(dyn_rand=True, seed=12, input_profile=1), dict(dynamics_randomization=True, seed=12, input_profile="min", command_profile="clock"), _gauss(0.1), 8, 3,
             obs_groups=GMIN(25), min_profile=True),
    Scenario("min_phase", dict(dyn_rand=True, seed=12, input_profile=1, command_profile=1), dict(dynamics_randomization=True, seed=12, input_profile="min", command_profile="phase"), _gauss(0.1), 8, 3,
             obs_groups=GMIN(30), min_profile=True),
    Scenario("phase_add", dict(dyn_rand=False, seed=41), dict(dynamics_randomization=False, seed=41), _zeros, 40, 0, prep=_prep_phase_add),
    # falling robots against the COMPLETE oracle: wild actions and a pelvis push - several limits of a leg at once, third / fourth capsule ends on the
    # floor, hip-pitch capsules: passes beyond the lane map's caps, which the kernel solves with the complete row set (cassie_complete.h)
    Scenario("falling_pushed", dict(dyn_rand=True, seed=31), dict(dynamics_randomization=True, seed=31), _gauss(1.0), 30, 9, prep=_prep_push, prep_kernel=_prep_push_kernel,
             reaches=_saturated, differing_max=0.35),
]
BY_NAME = {s.name: s for s in SCENARIOS}

# Fixed tolerances on the identical-row-set population of a scenario, columns [reward, qpos, qvel, drive torque (N m), motor position]; the 50-entry observation groups take
# tests/state_xfer.py TF_TOL_SAME.  Derived from the fp32 control like those (test_fp32_control_of_the_scenarios prints the control's maxima per scenario and asserts
# that it stays below them and reaches a tenth of the torque one somewhere): about 2 x the control's worst scenario.
TOL_TORQUE = 1.2          # N m at the joint.  One encoder count of motor position (4.8e-5 rad; the quantiser truncates, so a 1e-7 difference can flip a count) is 0.39 - 0.41 N m through
                          # the PD gain and the gear; inside a safety zone the zone's own stiffness rides on top: fp32 control 0.73 N m (coupled zone), 0.40 elsewhere
TOL_MIN_FOOT_POS = 1e-4   # m, estimator foot positions of the min profile (observation entries 0:6); fp32 control 2.0e-5
TOL_MIN_FOOT_ORI = 1e-4   # estimator foot orientation quaternions (entries 13:21); fp32 control 1.2e-5
