"""Batched Cassie-v0 behind the Python env seam of the reference.

`CassieVecEnv` exposes what rl/algos/ppo.py expects from `env_fn()` — reset(), step(action), observation_space,
action_space, mirrored_obs, mirrored_acts, clock_inds, clock_based, simrate (util/env.py:8-52,
rl/envs/wrappers.py:5-67, cassie/cassie.py:44-69,234-278) — for N environments at once, with every array a device
tensor.  The work happens in libapx.so (apx_env_* in include/apx.h); there is no CPU path.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import check
from .engine import _p, _stream

OBS_DIM, ACT_DIM = 50, 10


# The two places this module touches the device runtime.  (tests/test_kernel_emulation_env.py points them - and the library handle - at the kernel SOURCES compiled
# for the host, for the duration of a test; the product has no CPU path and raises here.)
def _device(index):
    if not torch.cuda.is_available():
        raise _lib.ApxError("CassieVecEnv needs a GPU (there is no CPU fallback)")
    d = torch.device("cuda", index)
    torch.cuda.set_device(d)
    return d


def _on_device(t):
    return t.is_cuda

# cassie/cassie.py:244 (input_profile "full") + :262-265 (command_profile "clock"); cassie/cassie.py:69
MIRRORED_OBS = [0.1, 1, -2, 3, -4, -10, -11, 12, 13, 14, -5, -6, 7, 8, 9, 15, -16, 17, -18, 19, -20, -26, -27, 28, 29, 30,
                -21, -22, 23, 24, 25, 31, -32, 33, 37, 38, 39, 34, 35, 36, 43, 44, 45, 40, 41, 42, 46, 47, 48, 49]
MIRRORED_ACTS = [-5, -6, 7, 8, 9, -0.1, -1, 2, 3, 4]
CLOCK_INDS = [46, 47]
OBS_DIM_PHASE = 55
# command_profile "phase" (cassie/cassie.py:266-271): the 46 base entries, then clock (2), swing / stance duration + one-hot stance mode (5),
# speed / side speed (2) mirror onto themselves
MIRRORED_OBS_PHASE = MIRRORED_OBS[:46] + list(range(46, 55))
# input_profile "min" (cassie/cassie.py:246-256): L / R foot position, pelvis quaternion, pelvis rotational velocity, L / R foot quaternion
MIRRORED_OBS_MIN = [3, 4, 5, 0.1, 1, 2, 6, -7, 8, -9, -10, 11, -12, 17, -18, 19, -20, 13, -14, 15, -16]
OBS_DIM_MIN = 21


def mirrored_obs_for(command_profile, input_profile):
    """CassieEnv.set_up_state_space (cassie/cassie.py:234-278): the mirror index list of one observation frame"""
    base = MIRRORED_OBS[:46] if input_profile == "full" else MIRRORED_OBS_MIN
    n_ext = 4 if command_profile == "clock" else 9
    return list(base) + [len(base) + i for i in range(n_ext)]


def parse_phase_reward(reward):
    """cassie/cassie.py:186-199 (set_up_phase_reward, command_profile "phase"): "early" -> early_clock_reward, "library" -> the library
    draws of reset (:531-539), "no_speed" -> no_speed_clock_reward (not built), anything else -> clock_reward."""
    if reward is None:
        raise TypeError("argument of type 'NoneType' is not iterable")
    if "no_speed" in reward:
        raise NotImplementedError("no_speed_clock_reward (cassie/rewards/clock_rewards.py) is not built")
    return dict(reward_kind=1 if "early" in reward else 0, stance_mode=0, have_incentive=int("no_incentive" not in reward), library="library" in reward)


def parse_reward(reward):
    """cassie/cassie.py:91,202-232,770-785: `--reward` is a substring-matched spec."""
    if reward is None:
        raise TypeError("argument of type 'NoneType' is not iterable")     # the reference crashes at cassie.py:91 too
    have_incentive = "no_incentive" not in reward
    early = "early" in reward
    if "load" in reward:
        raise NotImplementedError("load_clock rewards: no shipped clock file matches (SURVEY.md §8 a5b)")
    stance = 1 if "grounded" in reward else 2 if "aerial" in reward else 0
    # cassie.py:223-224 + :781-783: "max_vel" selects max_vel_clock_reward whatever the `early` flag says; "switch" only sets an
    # unused switch_speed (the switch_clock branch at :551 is unreachable) and stays on clock_reward
    kind = 2 if "max_vel" in reward else 1 if early else 0
    return dict(reward_kind=kind, stance_mode=stance, have_incentive=int(have_incentive))


# MuJoCo body ids of cassie.xml (world = 0): the name argument of CassieSim.apply_force (cassiemujoco.py:99)
BODY_NAMES = ["world", "cassie-pelvis"] + [side + "-" + n for side in ("left", "right") for n in (
    "hip-roll", "hip-yaw", "hip-pitch", "achilles-rod", "knee", "knee-spring", "shin", "tarsus", "heel-spring", "foot-crank", "plantar-rod", "foot")]


class CassieVecEnv:
    clock_based = True
    clock_inds = CLOCK_INDS
    mirrored_obs = MIRRORED_OBS
    mirrored_acts = MIRRORED_ACTS

    def __init__(self, n_envs=4096, simrate=50, dynamics_randomization=True, reward="clock", max_traj_len=400, seed=0,
                 device=0, pgs_iters=50, env_id_base=0, command_profile="clock", input_profile="full", history=0, learn_gains=False,
                 env_name="Cassie-v0", traj="walking", no_delta=True, ik_baseline=False, est_lifetime=169):
        if command_profile not in ("clock", "phase") or input_profile not in ("full", "min") or learn_gains or history < 0:
            raise NotImplementedError("command_profile clock / phase with input_profile full / min are built (the traj command profile and learn_gains are not)")
        if command_profile == "phase" and env_name != "Cassie-v0":
            raise NotImplementedError("command_profile=phase is built for Cassie-v0")
        # util/env.py:22-32: Cassie-v0 -> CassieEnv; CassieTraj-v0 -> CassieTrajEnv, which with the CLI defaults (traj=walking,
        # command_profile=clock, no_delta) has Cassie-v0's step and observation and resets to the reference trajectory's pose
        if env_name not in ("Cassie-v0", "CassieTraj-v0"):
            raise NotImplementedError("env_name %r: only Cassie-v0 and CassieTraj-v0 are built" % (env_name,))
        if env_name == "CassieTraj-v0" and (traj != "walking" or not no_delta or ik_baseline or simrate != 50):
            raise NotImplementedError("CassieTraj-v0 is built for traj=walking, no_delta, simrate 50 (the CLI defaults)")
        self.env_name = env_name
        self.device = _device(device)
        lib = _lib.load()
        cfg = _lib.EnvCfg()
        lib.apx_env_default_cfg(C.byref(cfg))
        r = parse_reward(reward) if command_profile == "clock" else parse_phase_reward(reward)
        cfg.command_profile = 0 if command_profile == "clock" else (2 if r["library"] else 1)
        self.command_profile = command_profile
        cfg.input_profile = 0 if input_profile == "full" else 1
        self.input_profile = input_profile
        self.frame_dim = (46 if input_profile == "full" else OBS_DIM_MIN) + (4 if command_profile == "clock" else 9)
        self.clock_inds = [self.frame_dim - (4 if command_profile == "clock" else 9) + i for i in range(2)]
        # --history h (cassie.py:51-55,565,856-859): the observation is the newest frame followed by the h previous ones of the episode (zeros
        # before its start); the kernel writes frames, the stack is kept here.  The reference's mirror lists cover one frame only.
        self.history = int(history)
        self.obs_dim = self.frame_dim * (self.history + 1)
        self.mirrored_obs = mirrored_obs_for(command_profile, input_profile)
        cfg.n_envs, cfg.simrate, cfg.dynamics_randomization = n_envs, simrate, int(dynamics_randomization)
        cfg.reward_kind, cfg.stance_mode, cfg.have_incentive = r["reward_kind"], r["stance_mode"], r["have_incentive"]
        cfg.max_traj_len, cfg.seed, cfg.device, cfg.pgs_iters = max_traj_len, seed, device, pgs_iters
        cfg.env_id_base = env_id_base
        cfg.est_lifetime = int(est_lifetime)      # env steps served by one estimator object (one PPO.sample call of the reference builds one CassieEnv); 0 = never restarted
        cfg.env_kind = 1 if env_name == "CassieTraj-v0" else 0
        self._h = C.c_void_p()
        check(lib.apx_env_create(C.byref(cfg), C.byref(self._h)))
        self.n_envs, self.simrate, self.max_traj_len = n_envs, simrate, max_traj_len
        self.observation_space = np.zeros(self.obs_dim)
        self.action_space = np.zeros(ACT_DIM)
        f32 = dict(dtype=torch.float32, device=self.device)
        self.obs = torch.zeros(n_envs, self.obs_dim, **f32)
        self.final_obs = torch.zeros(n_envs, self.obs_dim, **f32)
        if self.history:
            self._frame = torch.zeros(n_envs, self.frame_dim, **f32); self._fin_frame = torch.zeros(n_envs, self.frame_dim, **f32)
        self.reward = torch.zeros(n_envs, **f32)
        self.done = torch.zeros(n_envs, dtype=torch.uint8, device=self.device)

    def close(self):
        if getattr(self, "_h", None):
            _lib.load().apx_env_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self, mask=None):
        """CassieEnv.reset for every env (or those with mask != 0); returns the [N, 50] observation tensor."""
        if not self.history:
            check(_lib.load().apx_env_reset(self._h, _p(mask), _p(self.obs), _stream()))
            return self.obs
        check(_lib.load().apx_env_reset(self._h, _p(mask), _p(self._frame), _stream()))
        sel = torch.ones(self.n_envs, dtype=torch.bool, device=self.device) if mask is None else mask != 0
        self._push(self._frame, sel, self.obs)
        return self.obs

    def prepare_resets(self):
        """apx_env_prepare_resets: precompute the next two resets of every env (draws, set_const, forward pass) off the rollout's critical path"""
        check(_lib.load().apx_env_prepare_resets(self._h, _stream()))

    def set_complete_rows(self, on=True):
        """apx_env_set_complete_rows: solve a forward pass beyond the lane map's row caps with its complete row set (default on) or cap it as in rounds 1-4"""
        check(_lib.load().apx_env_set_complete_rows(self._h, int(bool(on))))

    def set_refill(self, on):
        """apx_env_set_refill: refill the reset ring of the envs that just restarted next to the following env step (default: on up to 2048 envs)"""
        check(_lib.load().apx_env_set_refill(self._h, int(bool(on))))

    def _push(self, frame, restart, out):
        """state_history.insert(0, state)[:history + 1] (cassie.py:856-859); envs in `restart` begin a new episode: zero history first (:565)"""
        D = self.frame_dim
        old = torch.where(restart.view(-1, 1), torch.zeros_like(self.obs[:, :-D]), self.obs[:, :-D].clone())
        new = torch.cat([frame, old], 1)
        out.copy_(new)
        if out is not self.obs:
            self.obs.copy_(new)

    def update_speed(self, new_speed, new_side_speed=None):
        """CassieEnv.update_speed (cassie.py:757-775) for every env; arguments are [N] tensors (or floats)."""
        f = lambda v: None if v is None else torch.as_tensor(v, dtype=torch.float32, device=self.device).expand(self.n_envs).contiguous()
        sp, sd = f(new_speed), f(new_side_speed)
        check(_lib.load().apx_env_update_speed(self._h, _p(sp), _p(sd), _stream()))

    def reset_for_test(self, full_reset=False):
        """CassieEnv.reset_for_test(full_reset) (cassie.py:682-742) for every env; returns the [N, 50] observation."""
        if not self.history:
            check(_lib.load().apx_env_reset_for_test(self._h, _p(self.obs), int(bool(full_reset)), _stream()))
            return self.obs
        check(_lib.load().apx_env_reset_for_test(self._h, _p(self._frame), int(bool(full_reset)), _stream()))
        self._push(self._frame, torch.ones(self.n_envs, dtype=torch.bool, device=self.device), self.obs)      # cassie.py:689: history cleared
        return self.obs

    def apply_force(self, xfrc, body_name="cassie-pelvis"):
        """CassieSim.apply_force (cassiemujoco.py:99-103) for every env: xfrc [N, 6] (or [6]) = world-frame force xyz + torque xyz on
        body `body_name` (a cassie.xml body name, tools/eval_perturb.py's perturb_body), acting at that body's centre of mass; stays applied until
        overwritten (tools/eval_perturb.py:62,70).  One pushed body at a time: a call replaces the previous wrench whatever body it was on."""
        if body_name not in BODY_NAMES[1:]:
            raise ValueError("unknown body %r (cassie.xml bodies: %s)" % (body_name, ", ".join(BODY_NAMES[1:])))
        x = torch.as_tensor(xfrc, dtype=torch.float32, device=self.device).expand(self.n_envs, 6).contiguous()
        check(_lib.load().apx_env_apply_force_body(self._h, _p(x), BODY_NAMES.index(body_name), _stream()))

    def set_command(self, speed=None, side_speed=None, orient_add=None, phase=None, phase_add=None):
        """Plain attribute writes `env.speed = ...`, `env.orient_add = ...`, `env.phase = ...` of the reference's test harnesses
        (tools/eval_perturb.py:32, tools/test_commands.py:66-120): no clipping, the clock is NOT rebuilt (unlike update_speed)."""
        if speed is not None or side_speed is not None or orient_add is not None:
            cmd = self.get_field("cmd")
            for col, v in ((0, speed), (1, side_speed), (2, orient_add)):
                if v is not None:
                    cmd[:, col] = torch.as_tensor(v, dtype=torch.float32, device=self.device)
            self.set_field("cmd", cmd)
        if phase is not None or phase_add is not None:      # the phase is an integer + a half bit, phase_add 1 or 1.5 a flag (I_FLAGS bits 5, 6; tools/test_commands.py:86)
            ints = self.get_field("ints")
            flags = ints[:, 4].to(torch.int64)
            if phase is not None:
                ph = torch.as_tensor(phase, dtype=torch.float32, device=self.device).expand(self.n_envs)
                ints[:, 1] = torch.floor(ph)
                flags = (flags & ~32) | ((ph - torch.floor(ph) >= 0.5).to(torch.int64) << 5)
            if phase_add is not None:
                pa = torch.as_tensor(phase_add, dtype=torch.float32, device=self.device).expand(self.n_envs)
                assert bool(((pa == 1.0) | (pa == 1.5)).all()), "phase_add: 1 or 1.5 (tools/test_commands.py:86-88)"
                flags = (flags & ~64) | ((pa > 1.25).to(torch.int64) << 6)
            ints[:, 4] = flags.to(torch.float32)
            self.set_field("ints", ints)

    def step_basic(self, action):
        """CassieEnv.step_basic (cassie.py:498-521) for every env: no reward / termination / command resampling; returns obs."""
        action = action.contiguous()
        assert action.shape == (self.n_envs, ACT_DIM) and action.dtype == torch.float32 and _on_device(action)
        if not self.history:
            check(_lib.load().apx_env_step_basic(self._h, _p(action), _p(self.obs), _stream()))
            return self.obs
        check(_lib.load().apx_env_step_basic(self._h, _p(action), _p(self._frame), _stream()))
        self._push(self._frame, torch.zeros(self.n_envs, dtype=torch.bool, device=self.device), self.obs)
        return self.obs

    def step(self, action, auto_reset=True, f_term=0, out=None):
        """CassieEnv.step for every env.  `f_term` is accepted and ignored exactly like cassie/cassie.py:389.
        Returns (obs, reward, done, final_obs): done 1 = terminated, 2 = truncated at max_traj_len; with auto_reset the
        finished envs restart inside the same launch and `final_obs` holds their last observation (rows of envs that did not
        finish are left untouched).  `out` = (obs, reward, done, final_obs) lets the kernels write straight into caller-owned
        contiguous device buffers (e.g. slices of a rollout grid) instead of the env's own."""
        if not _on_device(action):
            raise _lib.ApxError("CassieVecEnv.step needs a device tensor (there is no CPU path)")
        action = action.contiguous()
        assert action.shape == (self.n_envs, ACT_DIM) and action.dtype == torch.float32
        obs, rew, done, fin = out if out is not None else (self.obs, self.reward, self.done, self.final_obs)
        if out is not None:
            assert obs.is_contiguous() and rew.is_contiguous() and done.is_contiguous() and fin.is_contiguous()
            assert obs.shape == (self.n_envs, self.obs_dim) and fin.shape == (self.n_envs, self.obs_dim) and done.dtype == torch.uint8
        if not self.history:
            check(_lib.load().apx_env_step(self._h, _p(action), _p(obs), _p(rew), _p(done), _p(fin), int(auto_reset), _stream()))
            return obs, rew, done, fin
        # history: the kernel writes frames; the stacks are composed here.  A finished env's final observation is the stack that ends with
        # its last frame; with auto_reset its new observation starts a fresh stack.
        check(_lib.load().apx_env_step(self._h, _p(action), _p(self._frame), _p(rew), _p(done), _p(self._fin_frame), int(auto_reset), _stream()))
        D = self.frame_dim
        ended = done != 0
        prev = self.obs[:, :-D].clone()
        fin.copy_(torch.where(ended.view(-1, 1), torch.cat([self._fin_frame, prev], 1), fin))
        self._push(self._frame, ended & bool(auto_reset), obs)
        return obs, rew, done, fin

    # ---- raw state access (tests, tools) ----
    def get_field(self, name, count=None):
        lib = _lib.load()
        buf = torch.zeros(self.n_envs, 192, dtype=torch.float32, device=self.device)
        n = lib.apx_env_get_field(self._h, name.encode(), _p(buf), _stream())
        if n < 0:
            check(n)
        return buf.view(-1)[: self.n_envs * n].view(self.n_envs, n).clone() if n else None

    def set_field(self, name, value=None):
        lib = _lib.load()
        v = None if value is None else value.to(self.device, torch.float32).contiguous()
        n = lib.apx_env_set_field(self._h, name.encode(), _p(v), _stream())
        if n < 0:
            check(n)

    def set_hfield(self, data, size=(50.0, 50.0, 0.15)):
        """CassieSim("cassie_hfield.xml") + sim.set_hfield_data(data) (util/eval.py:73-76) for every env of the batch: data [nrow, ncol] raw
        elevations (numpy / tensor; x size[2] = metres), rows along y, columns along x, over [-size[0], size[0]] x [-size[1], size[1]]
        (cassie_hfield.xml:69: 500 x 500 over 100 m x 100 m, scale 0.15).  None restores the floor plane."""
        lib = _lib.load()
        if data is None:
            check(lib.apx_env_set_hfield(self._h, None, 0, 0, None, _stream())); return
        d = torch.as_tensor(np.asarray(data, dtype=np.float32) if not torch.is_tensor(data) else data, dtype=torch.float32).contiguous().cpu()
        sz = (C.c_float * 3)(*[float(x) for x in size])
        check(lib.apx_env_set_hfield(self._h, C.c_void_p(d.data_ptr()), int(d.shape[0]), int(d.shape[1]), C.cast(sz, C.c_void_p), _stream()))

    def kernel_timing(self, enable=True):
        """switch the hipEvent bracketing of env_step_kernel launches on / off (include/apx.h apx_env_timing)"""
        check(_lib.load().apx_env_timing(self._h, 1 if enable else 0))

    def kernel_timing_read(self, reset=True):
        """(total ms, launches) of the env_step_kernel launches recorded since the last reset"""
        ms, n = C.c_double(0.0), C.c_int64(0)
        check(_lib.load().apx_env_timing_read(self._h, C.byref(ms), C.byref(n), 1 if reset else 0))
        return ms.value, n.value

    def saturation(self):
        """(flags [N] int64, passes [N] int64): SAT_* bits (1 = more than 2 penetrating capsule ends on a leg, 2 = more than 1 active joint
        limit on a leg, 4 = pelvis sphere / hip-pitch capsule on the floor, 8 = a left-right capsule pair in contact) a forward pass of the
        env has needed beyond the constraint rows the kernel instantiates since the env was created, and how many such passes there were."""
        sat = self.get_field("ints_bits").view(torch.int32)[:, 6].to(torch.int64)      # bit-exact integer words (the float view of "ints" rounds above 2^24)
        return sat & 0xFF, sat >> 8

    def substep(self):
        self.get_field("substep")
