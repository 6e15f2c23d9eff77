"""Tensor-level wrappers over the C ABI (include/apx.h).  PyTorch-ROCm is plumbing here: device memory, streams,
torch.distributed.  Every function launches HIP kernels from libapx.so on the current torch stream."""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import check


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _need_gpu(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _lib.ApxError("apex_amd needs device tensors (no CPU path exists)")


def signed_perm_from_mirror(mirrored):
    """Turn apex's mirror index list (cassie/cassie.py:69,244; 0.1 = '+index 0') into the column-gather table the
    kernels use: (x @ M)[c] = sign_i * x[i] for the i with |mirrored[i]| == c (rl/envs/wrappers.py:70-77).
    Encoding: entry >= 0 takes +x[entry]; entry < 0 takes -x[-entry-1]."""
    n = len(mirrored)
    out = np.zeros(n, dtype=np.int32)
    seen = set()
    for i, m in enumerate(mirrored):
        c = int(abs(int(m)))
        assert c not in seen, "mirror list must be a permutation"
        seen.add(c)
        out[c] = i if np.sign(m) > 0 else -(i + 1)
    return out


def returns_scan(rew, end, boot, last_val, gamma):
    """[T,N] backward discounted-return scan (PPOBuffer.finish_path, rl/algos/ppo.py:73-89)."""
    _need_gpu(rew, end, boot, last_val)
    T, N = rew.shape
    assert rew.dtype == torch.float32 and end.dtype == torch.uint8 and boot.dtype == torch.float32
    rew, end, boot, last_val = rew.contiguous(), end.contiguous(), boot.contiguous(), last_val.contiguous()
    ret = torch.empty_like(rew)
    check(_lib.load().apx_returns_scan(_p(rew), _p(end), _p(boot), _p(last_val), float(gamma), T, N, _p(ret), _stream()))
    return ret


def adv_moments(ret, val):
    _need_gpu(ret, val)
    mom = torch.empty(3, dtype=torch.float64, device=ret.device)
    check(_lib.load().apx_adv_moments(_p(ret), _p(val), ret.numel(), _p(mom), _stream()))
    return mom


def normalize_advantages(ret, val, eps=1e-5, group=None):
    """rl/algos/ppo.py:395-396.  With a process group the three moments are all-reduced (RCCL) first so that every
    rank normalises with the statistics of the union batch."""
    ret, val = ret.contiguous().view(-1), val.contiguous().view(-1)
    from . import dist as adist
    mom = adv_moments(ret, val)
    if group is not None:
        adist.allreduce_moments(mom, group=group)
    adv = torch.empty_like(ret)      # mean / unbiased std are derived from the moments on the device: no host sync here
    check(_lib.load().apx_adv_apply_moments(_p(ret), _p(val), ret.numel(), _p(mom), eps, _p(adv), _stream()))
    return adv


class Mlp:
    """Flat fp32 parameter block of a 3-layer ReLU MLP in state_dict order (W0,b0,W1,b1,W2,b2; torch [out,in])."""

    def __init__(self, D, H, O, device):
        self.D, self.H, self.O = D, H, O
        self.n = int(_lib.load().apx_mlp_param_count(D, H, O))
        self.params = torch.zeros(self.n, dtype=torch.float32, device=device)

    def views(self, flat=None):
        flat = self.params if flat is None else flat
        D, H, O = self.D, self.H, self.O
        shapes = [(H, D), (H,), (H, H), (H,), (O, H), (O,)]
        out, off = [], 0
        for s in shapes:
            k = int(np.prod(s))
            out.append(flat[off:off + k].view(*s))
            off += k
        return out

    def load_list(self, tensors):
        for v, t in zip(self.views(), tensors):
            v.copy_(torch.as_tensor(np.asarray(t), dtype=torch.float32))

    def forward(self, x, obs_mean=None, obs_std=None, idx=None, sign_perm=None, clock_mask=0, keep=False, out=None):
        _need_gpu(x)
        x = x.contiguous()
        assert x.shape[-1] == self.D, "input width %d, the network was built for %d" % (x.shape[-1], self.D)      # a wrong row stride reads out of bounds on the device
        assert obs_mean is None or obs_mean.numel() == self.D, "observation statistics of %d entries for a %d-input network" % (obs_mean.numel(), self.D)
        B = x.shape[0] if idx is None else idx.numel()
        dev = x.device
        if keep or not (self.H == 256 and self.D <= 64 and self.O <= 128):
            xn = torch.empty(B, self.D, dtype=torch.float32, device=dev)
            a1 = torch.empty(B, self.H, dtype=torch.float32, device=dev)
            a2 = torch.empty(B, self.H, dtype=torch.float32, device=dev)
        else:
            xn = a1 = a2 = None              # inference on the fused shapes: nothing but y leaves the kernel
        if out is not None:
            assert out.is_contiguous() and out.numel() == B * self.O and out.dtype == torch.float32 and out.device == dev
            y = out
        else:
            y = torch.empty(B, self.O, dtype=torch.float32, device=dev)
        check(_lib.load().apx_mlp_forward(_p(self.params), self.D, self.H, self.O, _p(x), B, _p(idx), _p(sign_perm),
                                          int(clock_mask), _p(obs_mean), _p(obs_std), _p(xn), _p(a1), _p(a2), _p(y), _stream()))
        return (y, xn, a1, a2) if keep else y


class Lstm:
    """Flat fp32 parameter block of L stacked LSTM cells + a linear head in the state_dict order of Gaussian_LSTM_Actor / LSTM_V
    (rl/policies/actor.py:218-311, critic.py:236-296)."""

    def __init__(self, D, H, L, O, device):
        self.D, self.H, self.L, self.O = D, H, L, O
        self.n = int(_lib.load().apx_lstm_param_count(D, H, L, O))
        self.params = torch.zeros(self.n, dtype=torch.float32, device=device)
        self.device = device
        self._packed = None          # apx_lstm_step's layout of the parameters ...
        self._packed_version = -1    # ... and the torch version counter of `params` it was made from (every in-place write bumps it)

    def views(self, flat=None):
        flat = self.params if flat is None else flat
        H, out, off = self.H, [], 0
        shapes = []
        for l in range(self.L):
            i = self.D if l == 0 else H
            shapes += [(4 * H, i), (4 * H, H), (4 * H,), (4 * H,)]
        shapes += [(self.O, H), (self.O,)]
        for sh in shapes:
            k = int(np.prod(sh)); out.append(flat[off:off + k].view(*sh)); off += k
        return out

    def load_list(self, tensors):
        for v, t in zip(self.views(), tensors):
            v.copy_(torch.as_tensor(np.asarray(t), dtype=torch.float32))

    def forward(self, x, hc=None, keep=False):
        """x [T, B, D] prepared input (or [B, D] for one step with the carried state hc [L, 2, B, H], updated in place)."""
        _need_gpu(x)
        step = x.dim() == 2
        if keep and step and hc is not None:
            raise ValueError("Lstm.forward(keep=True) with a carried one-step state: the in-place (h, c) path does not fill the backward workspace; "
                             "run the sequence form [T, B, D] from a zero state for BPTT")
        x3 = (x.unsqueeze(0) if step else x).contiguous()
        T, B, _ = x3.shape
        assert x3.shape[-1] == self.D, "input width %d, the network was built for %d" % (x3.shape[-1], self.D)
        lib = _lib.load()
        save = torch.empty(int(lib.apx_lstm_workspace_floats(T, B, self.H, self.L)), dtype=torch.float32, device=x.device)
        y = torch.empty(T, B, self.O, dtype=torch.float32, device=x.device)
        check(lib.apx_lstm_forward(_p(self.params), self.D, self.H, self.L, self.O, _p(x3), T, B, _p(hc), _p(save), _p(y), _stream()))
        y = y[0] if step else y
        return (y, x3, save) if keep else y

    # ---- the rollout's one-step pass as one launch (apx_lstm_step): normalisation, hidden-state reset, both cells, head and the action noise
    def step_supported(self):
        return int(_lib.load().apx_lstm_step_pack_floats(self.D, self.H, self.L, self.O)) > 0

    def pack_step(self):
        """Re-lay the parameters for apx_lstm_step ([W_ih | W_hh] per cell, summed biases); call again whenever the parameters changed."""
        lib = _lib.load()
        n = int(lib.apx_lstm_step_pack_floats(self.D, self.H, self.L, self.O))
        if n == 0:
            raise ValueError("apx_lstm_step needs L = 2, H = 128, D <= 64, O <= 16")
        if self._packed is None or self._packed.numel() != n:
            self._packed = torch.empty(n, dtype=torch.float32, device=self.params.device)
        check(lib.apx_lstm_step_pack(_p(self.params), self.D, self.H, self.L, self.O, _p(self._packed), _stream()))
        self._packed_version = self.params._version
        return self._packed

    def step(self, x, hc, obs_mean=None, obs_std=None, reset=None, noise=None, sigma=0.0, act_out=None, y_out=None):
        """One rollout step on raw observations x [B, D]: y = head(LSTM(normalise(x))), hc [L, 2, B, H] updated in place; the rows whose `reset` byte (uint8 [B]) is
        non-zero start from a zero state; act_out (if given) = y + sigma * noise.  The packed copy of the parameters is refreshed here when a torch-side write changed
        `params` since pack_step() (its version counter moved); a writer torch cannot see - the C-ABI optimiser step, a collective on the raw pointer - must call
        pack_step() itself, as RecurrentPPO.sample does once per rollout."""
        _need_gpu(x)
        B = x.shape[0]
        assert x.is_contiguous() and hc.is_contiguous() and x.shape[1] == self.D
        if self._packed is None or self._packed_version != self.params._version:
            self.pack_step()
        y = y_out if y_out is not None else torch.empty(B, self.O, dtype=torch.float32, device=x.device)
        check(_lib.load().apx_lstm_step(_p(self._packed), self.D, self.H, self.L, self.O, _p(x), _p(obs_mean), _p(obs_std), _p(reset), _p(hc), B, _p(y),
                                        _p(act_out), _p(noise), float(sigma), _stream()))
        return y

    def backward(self, grads, x3, save, dy):
        """grads (flat, same layout) += d(loss)/d(params) for dy [T, B, O]; x3 / save from forward(keep=True) with a zero start state."""
        T, B, _ = x3.shape
        lib = _lib.load()
        scratch = torch.empty(int(lib.apx_lstm_bwd_scratch_floats(T, B, self.D, self.H)), dtype=torch.float32, device=x3.device)
        check(lib.apx_lstm_backward(_p(self.params), _p(grads), self.D, self.H, self.L, self.O, _p(x3), T, B, _p(save), _p(dy.contiguous()),
                                    _p(scratch), _stream()))


class PPOLearner:
    """Device-resident actor/critic + Adam state; one call = one PPO.update_policy (rl/algos/ppo.py:276-345)."""

    def __init__(self, obs_dim, act_dim, hidden, device, fixed_std, lr=1e-4, eps=1e-5, clip=0.2, entropy_coeff=0.0,
                 grad_clip=0.05, mirrored_obs=None, mirrored_acts=None, clock_inds=(46, 47), mirror_coeff=0.4):
        self.device = device
        self.actor = Mlp(obs_dim, hidden, act_dim, device)
        self.critic = Mlp(obs_dim, hidden, 1, device)
        z = lambda m: torch.zeros(m.n, dtype=torch.float32, device=device)
        self.actor_m, self.actor_v = z(self.actor), z(self.actor)
        self.critic_m, self.critic_v = z(self.critic), z(self.critic)
        # both gradients in ONE flat buffer so that N>1 ranks need a single RCCL all-reduce per optimiser step
        self.grad_flat = torch.zeros(self.actor.n + self.critic.n, dtype=torch.float32, device=device)
        self.actor_g, self.critic_g = self.grad_flat[:self.actor.n], self.grad_flat[self.actor.n:]
        self.obs_mean = torch.zeros(obs_dim, dtype=torch.float32, device=device)
        self.obs_std = torch.ones(obs_dim, dtype=torch.float32, device=device)
        self.fixed_std, self.lr, self.eps, self.clip = float(fixed_std), lr, eps, clip
        self.entropy_coeff, self.grad_clip, self.mirror_coeff = entropy_coeff, grad_clip, mirror_coeff
        self.t = 0
        self.obs_sp = self.act_sp = None
        self.clock_mask = 0
        if mirrored_obs is not None:
            self.obs_sp = torch.as_tensor(signed_perm_from_mirror(mirrored_obs), device=device)
            self.act_sp = torch.as_tensor(signed_perm_from_mirror(mirrored_acts), device=device)
            for c in clock_inds:
                if not 0 <= int(c) < obs_dim:      # the reference takes them from env.clock_inds (rl/algos/ppo.py:307-310); a column beyond the observation would silently never be phase-shifted
                    raise ValueError("clock index %d outside the %d-entry observation" % (int(c), obs_dim))
                self.clock_mask |= 1 << int(c)
        self._ws = None
        self._ews = None
        self._scal = torch.zeros(6, dtype=torch.float64, device=device)

    def old_means(self, obs):
        """old_policy.distribution(obs) means for the whole buffer (ppo.py:284-285; old == current at iteration start)."""
        return self.actor.forward(obs, self.obs_mean, self.obs_std)

    def values(self, obs):
        return self.critic.forward(obs)          # training mode: raw obs (critic.py:66-67)

    def minibatch(self, obs, act, ret, adv, old_mu, idx=None, mirror=True, grad_only=False, sync=True):
        lib = _lib.load()
        mb = obs.shape[0] if idx is None else idx.numel()
        D, H, A = self.actor.D, self.actor.H, self.actor.O
        need = int(lib.apx_ppo_workspace_bytes(mb, D, H, A))
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        if not grad_only:
            self.t += 1
        use_mirror = mirror and self.obs_sp is not None
        a = _lib.PpoArgs(
            actor=_p(self.actor.params), actor_m=_p(self.actor_m), actor_v=_p(self.actor_v), actor_grad=_p(self.actor_g),
            critic=_p(self.critic.params), critic_m=_p(self.critic_m), critic_v=_p(self.critic_v),
            critic_grad=_p(self.critic_g), D=D, H=H, A=A, obs=_p(obs), act=_p(act), ret=_p(ret), adv=_p(adv),
            old_mu=_p(old_mu), idx=_p(idx), mb=mb, obs_mean=_p(self.obs_mean), obs_std=_p(self.obs_std),
            obs_sign_perm=_p(self.obs_sp) if use_mirror else None, clock_mask=self.clock_mask,
            act_sign_perm=_p(self.act_sp) if use_mirror else None, fixed_std=self.fixed_std, clip=self.clip,
            entropy_coeff=self.entropy_coeff, grad_clip=self.grad_clip, lr=self.lr, adam_eps=self.eps,
            mirror_coeff=self.mirror_coeff, adam_t=max(self.t, 1), grad_only=int(grad_only),
            workspace=_p(self._ws), workspace_bytes=self._ws.numel(), scalars_out=_p(self._scal))
        check(lib.apx_ppo_minibatch(C.byref(a), _stream()))
        return self._scal.cpu().numpy().copy() if sync else self._scal

    def epoch_supported(self, mb):
        """Whether epoch() serves this minibatch size (the 2 x 256 networks, mb a multiple of 16 up to 1024)."""
        return bool(_lib.load().apx_ppo_epoch_supported(int(mb), self.actor.D, self.actor.H, self.actor.O))

    def epoch(self, obs, act, ret, adv, old_mu, perm, mb, mirror=True):
        """nb = len(perm) // mb optimiser steps (the minibatch loop of rl/algos/ppo.py:417-438) as ONE launch: step k = minibatch(idx=perm[k mb:(k + 1) mb]).
        Returns the [nb, 6] f64 device tensor of the steps' scalars (no host sync)."""
        lib = _lib.load()
        nb = int(perm.numel()) // int(mb)
        assert nb >= 1 and perm.dtype == torch.int64 and perm.is_contiguous()
        D, H, A = self.actor.D, self.actor.H, self.actor.O
        need = int(lib.apx_ppo_epoch_workspace_bytes(int(mb), nb, D, H, A))
        if need == 0:
            raise _lib.ApxError("apx_ppo_epoch does not serve minibatch %d of a %d-%d-%d network" % (mb, D, H, A))
        if self._ews is None or self._ews.numel() < need:
            self._ews = torch.empty(need, dtype=torch.uint8, device=self.device)
        scal = torch.empty(nb, 6, dtype=torch.float64, device=self.device)
        use_mirror = mirror and self.obs_sp is not None
        a = _lib.PpoArgs(
            actor=_p(self.actor.params), actor_m=_p(self.actor_m), actor_v=_p(self.actor_v), actor_grad=_p(self.actor_g),
            critic=_p(self.critic.params), critic_m=_p(self.critic_m), critic_v=_p(self.critic_v),
            critic_grad=_p(self.critic_g), D=D, H=H, A=A, obs=_p(obs), act=_p(act), ret=_p(ret), adv=_p(adv),
            old_mu=_p(old_mu), idx=None, mb=int(mb), obs_mean=_p(self.obs_mean), obs_std=_p(self.obs_std),
            obs_sign_perm=_p(self.obs_sp) if use_mirror else None, clock_mask=self.clock_mask,
            act_sign_perm=_p(self.act_sp) if use_mirror else None, fixed_std=self.fixed_std, clip=self.clip,
            entropy_coeff=self.entropy_coeff, grad_clip=self.grad_clip, lr=self.lr, adam_eps=self.eps,
            mirror_coeff=self.mirror_coeff, adam_t=self.t + 1, grad_only=0,
            workspace=_p(self._ews), workspace_bytes=self._ews.numel(), scalars_out=_p(scal))
        check(lib.apx_ppo_epoch(C.byref(a), _p(perm), nb, _stream()))
        self.t += nb
        return scal

    def apply_grads(self, scale=1.0):
        """clip + Adam on (all-reduced) gradients after minibatch(grad_only=True)."""
        lib = _lib.load()
        self.t += 1
        sc = torch.zeros(2, dtype=torch.float64, device=self.device)
        check(lib.apx_clip_adam(_p(self.actor.params), _p(self.actor_m), _p(self.actor_v), _p(self.actor_g), self.actor.n,
                                scale, self.grad_clip, self.lr, self.eps, self.t, _p(sc[0:1]), _stream()))
        check(lib.apx_clip_adam(_p(self.critic.params), _p(self.critic_m), _p(self.critic_v), _p(self.critic_g),
                                self.critic.n, scale, self.grad_clip, self.lr, self.eps, self.t, _p(sc[1:2]), _stream()))


class RecurrentPPOLearner:
    """PPO.update_policy in recurrent mode (rl/algos/ppo.py:276-345 on the padded [T_max, B, .] batch of :411-430): Gaussian_LSTM_Actor +
    LSTM_V through apx_lstm_forward / apx_ppo_loss / apx_lstm_backward / apx_clip_adam.  The old policy is a frozen copy of the actor's
    parameter block (ppo.py:400 old_policy.load_state_dict)."""

    def __init__(self, obs_dim, act_dim, hidden, layers, device, fixed_std, lr=1e-4, eps=1e-5, clip=0.2, grad_clip=0.05,
                 mirrored_obs=None, mirrored_acts=None, clock_inds=(46, 47), mirror_coeff=0.4):
        self.device = device
        self.actor = Lstm(obs_dim, hidden, layers, act_dim, device)
        self.critic = Lstm(obs_dim, hidden, layers, 1, device)
        self.old_actor = Lstm(obs_dim, hidden, layers, act_dim, device)
        z = lambda m: torch.zeros(m.n, dtype=torch.float32, device=device)
        self.actor_m, self.actor_v, self.critic_m, self.critic_v = z(self.actor), z(self.actor), z(self.critic), z(self.critic)
        self.grad_flat = torch.zeros(self.actor.n + self.critic.n, dtype=torch.float32, device=device)
        self.actor_g, self.critic_g = self.grad_flat[:self.actor.n], self.grad_flat[self.actor.n:]
        self.obs_mean = torch.zeros(obs_dim, dtype=torch.float32, device=device); self.obs_std = torch.ones(obs_dim, dtype=torch.float32, device=device)
        self.fixed_std, self.lr, self.eps, self.clip, self.grad_clip, self.mirror_coeff = float(fixed_std), lr, eps, clip, grad_clip, mirror_coeff
        self.t = 0
        self.act_sp = None
        if mirrored_obs is not None:
            sp = signed_perm_from_mirror(mirrored_obs)
            self.obs_src = torch.as_tensor(np.where(sp >= 0, sp, -sp - 1), dtype=torch.long, device=device)
            self.obs_sgn = torch.as_tensor(np.where(sp >= 0, 1.0, -1.0), dtype=torch.float32, device=device)
            self.act_sp = torch.as_tensor(signed_perm_from_mirror(mirrored_acts), device=device)
            self.obs_sp = torch.as_tensor(np.asarray(sp, dtype=np.int32), device=device)
            self.clock_cols = [int(c) for c in clock_inds]
            if any(not 0 <= c < obs_dim for c in self.clock_cols):
                raise ValueError("clock indices %r outside the %d-entry observation" % (self.clock_cols, obs_dim))
            self.clock_mask = sum(1 << c for c in self.clock_cols)
        self._scal = torch.zeros(6, dtype=torch.float64, device=device)
        self._acc = torch.zeros(8, dtype=torch.float64, device=device)

    def sync_old(self):
        self.old_actor.params.copy_(self.actor.params)

    def mirror_obs(self, obs):
        """SymmetricEnv.mirror_clock_observation (rl/envs/wrappers.py:59-67) on [..., D] raw observations."""
        m = obs.index_select(-1, self.obs_src) * self.obs_sgn
        for c in self.clock_cols:
            m[..., c] = torch.sin(torch.asin(m[..., c]) + np.pi)
        return m

    def gather(self, idx, obs, act, ret, adv, mirror=True, traj=None, sel=None, grid_cols=0, t_max=0):
        """The padded minibatch of ppo.py:411-430 out of the rollout grid in one launch (apx_rec_gather).  Either idx [T, B] int64 flat grid rows, -1 = padded, or
        (idx None) traj [n_traj, 3] int64 (column, t0, t1) + sel [B] int64 trajectory numbers + grid_cols + t_max = the longest selected trajectory (the index is then
        formed inside the kernel); obs [rows, D], act [rows, A], ret / adv [rows].  Returns (obs [T, B, D], act, ret, adv, mask [T, B, 1], prepared) for
        minibatch(..., prepared=prepared)."""
        _need_gpu(obs)
        if idx is not None:
            T, B = idx.shape
            assert idx.is_contiguous()
        else:
            T, B = int(t_max), int(sel.shape[0])
            assert traj.is_contiguous() and sel.is_contiguous() and traj.dtype == torch.int64 and sel.dtype == torch.int64 and grid_cols > 0 and T > 0
        D, A = self.actor.D, self.actor.O
        dev = obs.device
        use_mirror = mirror and self.act_sp is not None
        e = lambda *shape: torch.empty(*shape, dtype=torch.float32, device=dev)
        obs_raw, xn, act_p, ret_p, adv_p, mask = e(T, B, D), e(T, B, D), e(T, B, A), e(T, B, 1), e(T, B, 1), e(T, B, 1)
        xa = e(T, 2 * B, D) if use_mirror else None
        assert obs.is_contiguous() and act.is_contiguous() and ret.is_contiguous() and adv.is_contiguous()
        check(_lib.load().apx_rec_gather(_p(idx), _p(traj), _p(sel), int(grid_cols), T, B, D, A, _p(obs), _p(act), _p(ret), _p(adv), _p(self.obs_sp) if use_mirror else None,
                                         self.clock_mask if use_mirror else 0, _p(self.obs_mean), _p(self.obs_std), _p(obs_raw), _p(xn), _p(xa), _p(act_p), _p(ret_p),
                                         _p(adv_p), _p(mask), _stream()))
        return obs_raw, act_p, ret_p, adv_p, mask, (xn, xa if use_mirror else xn)

    def minibatch(self, obs, act, ret, adv, mask, mirror=True, grad_only=False, prepared=None):
        """obs [T, B, D], act [T, B, A], ret / adv / mask [T, B, 1] padded like torch's pad_sequence.  Returns the six scalars (device f64).
        The three sequence passes that do not depend on each other (old policy, new policy, critic) are latency-bound chains of small
        launches; they run on three HIP streams side by side, and so do the actor's and the critic's backward passes.
        prepared = the (xn, xa) pair of gather(): the normalised / mirrored inputs are not rebuilt."""
        lib = _lib.load()
        T, B, _ = obs.shape
        A = self.actor.O
        norm = lambda o: ((o - self.obs_mean) / self.obs_std).contiguous()
        obs_c = obs.contiguous()
        use_mirror = mirror and self.act_sp is not None
        if prepared is not None:
            xn, xa = prepared
        else:
            xn = norm(obs)
            xa = torch.cat([xn, norm(self.mirror_obs(obs))], dim=1) if use_mirror else xn      # pi(s) and pi(M s) share the weights: one pass over 2B columns
        main = torch.cuda.current_stream()
        if getattr(self, "_side", None) is None:
            self._side = (torch.cuda.Stream(device=self.device), torch.cuda.Stream(device=self.device))
        s_old, s_cri = self._side
        s_old.wait_stream(main); s_cri.wait_stream(main)
        with torch.cuda.stream(s_old):
            old_mu = self.old_actor.forward(xn)
        with torch.cuda.stream(s_cri):
            v, xc3, save_c = self.critic.forward(obs_c, keep=True)               # LSTM_V in train mode: raw inputs (critic.py:262-263)
        y, x3, save = self.actor.forward(xa, keep=True)
        mu, mum = (y[:, :B].contiguous(), y[:, B:].contiguous()) if use_mirror else (y, None)
        main.wait_stream(s_old); main.wait_stream(s_cri)
        rows = T * B
        dmu = torch.empty(rows, A, dtype=torch.float32, device=self.device); dv = torch.empty(rows, dtype=torch.float32, device=self.device)
        dmum = torch.empty(rows, A, dtype=torch.float32, device=self.device) if use_mirror else None
        check(lib.apx_ppo_loss(_p(mu), _p(mum) if use_mirror else None, _p(v), _p(act.contiguous()), _p(ret.contiguous()), _p(adv.contiguous()),
                               _p(old_mu), _p(mask.contiguous()) if mask is not None else None, _p(self.act_sp) if use_mirror else None, rows, A,
                               self.fixed_std, self.clip, self.mirror_coeff, _p(dmu), _p(dmum), _p(dv), _p(self._scal), _p(self._acc), _stream()))
        self.grad_flat.zero_()
        dy = torch.cat([dmu.view(T, B, A), dmum.view(T, B, A)], dim=1) if use_mirror else dmu.view(T, B, A)
        s_cri.wait_stream(main)
        with torch.cuda.stream(s_cri):
            self.critic.backward(self.critic_g, xc3, save_c, dv.view(T, B, 1))
        self.actor.backward(self.actor_g, x3, save, dy)
        main.wait_stream(s_cri)
        if not grad_only:
            self.apply_grads()
        return self._scal

    def apply_grads(self, scale=1.0):
        lib = _lib.load()
        self.t += 1
        sc = torch.zeros(2, dtype=torch.float64, device=self.device)
        check(lib.apx_clip_adam(_p(self.actor.params), _p(self.actor_m), _p(self.actor_v), _p(self.actor_g), self.actor.n, scale, self.grad_clip,
                                self.lr, self.eps, self.t, _p(sc[0:1]), _stream()))
        check(lib.apx_clip_adam(_p(self.critic.params), _p(self.critic_m), _p(self.critic_v), _p(self.critic_g), self.critic.n, scale,
                                self.grad_clip, self.lr, self.eps, self.t, _p(sc[1:2]), _stream()))


class TD3Learner:
    """TD3.train (rl/algos/sync_td3.py:133-209): FF_Actor with tanh head + Dual_Q_Critic (two 3-layer ReLU MLPs on cat(state, action)),
    target networks, target-policy smoothing, clipped double-Q target, delayed policy update, Polyak averaging - on apx_mlp_forward /
    apx_mlp_backward / the apx_td3_* kernels / apx_clip_adam (clip disabled, Adam eps 1e-8 like torch.optim.Adam's default)."""

    def __init__(self, obs_dim, act_dim, hidden, device, max_action=1.0, a_lr=1e-3, c_lr=1e-3):
        self.D, self.A, self.H, self.device, self.max_action = obs_dim, act_dim, hidden, device, float(max_action)
        mk = lambda d, o: Mlp(d, hidden, o, device)
        self.actor, self.actor_t = mk(obs_dim, act_dim), mk(obs_dim, act_dim)
        # Dual_Q_Critic's state_dict order is q1 (W0 b0 W1 b1 Wout bout) then q2: one flat block, two views
        n1 = int(_lib.load().apx_mlp_param_count(obs_dim + act_dim, hidden, 1))
        self.critic_flat = torch.zeros(2 * n1, dtype=torch.float32, device=device); self.critic_t_flat = torch.zeros_like(self.critic_flat)
        self.q = [mk(obs_dim + act_dim, 1), mk(obs_dim + act_dim, 1)]; self.q_t = [mk(obs_dim + act_dim, 1), mk(obs_dim + act_dim, 1)]
        for i in range(2):
            self.q[i].params = self.critic_flat[i * n1:(i + 1) * n1]; self.q_t[i].params = self.critic_t_flat[i * n1:(i + 1) * n1]
        z = lambda n: torch.zeros(n, dtype=torch.float32, device=device)
        self.a_m, self.a_v, self.a_g = z(self.actor.n), z(self.actor.n), z(self.actor.n)
        self.c_m, self.c_v, self.c_g = z(2 * n1), z(2 * n1), z(2 * n1)
        self.n1, self.a_lr, self.c_lr = n1, a_lr, c_lr
        self.t_a = self.t_c = 0
        self._acc = torch.zeros(3, dtype=torch.float64, device=device)
        self._nrm = torch.zeros(2, dtype=torch.float64, device=device)

    def _cat(self, state, pre, noise=None, noise_clip=0.0):
        B = state.shape[0]
        out = torch.empty(B, self.D + self.A, dtype=torch.float32, device=self.device)
        check(_lib.load().apx_td3_cat_action(_p(state), _p(pre), _p(noise), float(noise_clip), self.max_action, B, self.D, self.A, _p(out), _stream()))
        return out

    def _adam(self, p, m, v, g, lr, t):
        check(_lib.load().apx_clip_adam(_p(p), _p(m), _p(v), _p(g), p.numel(), 1.0, 1e30, lr, 1e-8, t, _p(self._nrm[0:1]), _stream()))

    def act(self, state):
        """FF_Actor.forward: max_action * tanh(network_out(...))."""
        return self.max_action * torch.tanh(self.actor.forward(state))

    def updates_supported(self, batch):
        return bool(_lib.load().apx_td3_updates_supported(int(batch), self.D, self.H, self.A))

    def updates(self, state, next_state, action, reward, notdone, ind, noise, it0, discount=0.99, tau=0.005, noise_clip=0.5, policy_freq=2):
        """U = ind.shape[0] iterations of TD3.train's loop body (sync_td3.py:133-209) as ONE launch (apx_td3_updates): update u works on the replay rows ind[u] [B] of
        (state, next_state, action, reward, notdone) with the smoothing noise noise[u] [B, A] ~ N(0, policy_noise) (clamped inside) and iteration counter it0 + u.
        Returns the [U, 4] f64 device tensor (critic loss, sum q1, sum q2, actor loss or 0) - no host sync."""
        lib = _lib.load()
        U, B = ind.shape
        assert ind.dtype == torch.int64 and ind.is_contiguous() and noise.is_contiguous() and tuple(noise.shape) == (U, B, self.A)
        for t in (state, next_state, action, reward, notdone):
            assert t.is_contiguous() and t.dtype == torch.float32
        need = int(lib.apx_td3_updates_workspace_bytes(B, U, self.D, self.H, self.A))
        if need == 0:
            raise _lib.ApxError("apx_td3_updates does not serve batch %d of a %d+%d-%d network" % (B, self.D, self.A, self.H))
        if getattr(self, "_uws", None) is None or self._uws.numel() < need:
            self._uws = torch.empty(need, dtype=torch.uint8, device=self.device)
        stats = torch.empty(U, 4, dtype=torch.float64, device=self.device)
        a = _lib.Td3Args(actor=_p(self.actor.params), actor_t=_p(self.actor_t.params), actor_m=_p(self.a_m), actor_v=_p(self.a_v),
                         critic=_p(self.critic_flat), critic_t=_p(self.critic_t_flat), critic_m=_p(self.c_m), critic_v=_p(self.c_v),
                         D=self.D, H=self.H, A=self.A, state=_p(state), next_state=_p(next_state), action=_p(action), reward=_p(reward), notdone=_p(notdone),
                         ind=_p(ind), noise=_p(noise), B=B, U=U, it0=int(it0), policy_freq=int(policy_freq), max_action=self.max_action, noise_clip=float(noise_clip),
                         discount=float(discount), tau=float(tau), a_lr=self.a_lr, c_lr=self.c_lr, adam_eps=1e-8, t_a=self.t_a, t_c=self.t_c,
                         workspace=_p(self._uws), workspace_bytes=self._uws.numel(), stats_out=_p(stats))
        check(lib.apx_td3_updates(C.byref(a), _stream()))
        self.t_c += U
        self.t_a += sum(1 for u in range(U) if (int(it0) + u) % int(policy_freq) == 0)
        return stats

    def train_step(self, state, action, next_state, reward, notdone, noise, it, discount=0.99, tau=0.005, noise_clip=0.5, policy_freq=2):
        """One iteration of TD3.train's loop on a sampled batch; noise [B, A] ~ N(0, policy_noise) (clamped inside).  Returns
        (critic loss, mean q1, mean q2, actor loss or None) as host floats only when asked via .item() by the caller: device tensors."""
        lib = _lib.load()
        B = state.shape[0]
        state, action, next_state = state.contiguous(), action.contiguous(), next_state.contiguous()
        # the twin critics are independent, latency-bound chains at these batch sizes: Q2's passes run on a second HIP stream
        main = torch.cuda.current_stream()
        if getattr(self, "_side", None) is None:
            self._side = torch.cuda.Stream(device=self.device)
        side = self._side
        xin_t = self._cat(next_state, self.actor_t.forward(next_state), noise.contiguous(), noise_clip)
        xin = torch.cat([state, action], 1).contiguous()
        side.wait_stream(main)
        with torch.cuda.stream(side):
            tq2 = self.q_t[1].forward(xin_t); k2 = self.q[1].forward(xin, keep=True)
        tq1 = self.q_t[0].forward(xin_t); k1 = self.q[0].forward(xin, keep=True)
        main.wait_stream(side)
        kept = [k1, k2]
        dq = [torch.empty(B, dtype=torch.float32, device=self.device) for _ in range(2)]
        check(lib.apx_td3_critic_loss(_p(kept[0][0]), _p(kept[1][0]), _p(tq1), _p(tq2), _p(reward.contiguous()), _p(notdone.contiguous()), float(discount), B,
                                      _p(dq[0]), _p(dq[1]), _p(self._acc), _stream()))
        stats = self._acc.clone()
        self.c_g.zero_()
        scratch = torch.empty(2 * B * self.H, dtype=torch.float32, device=self.device); scratch2 = torch.empty_like(scratch)
        side.wait_stream(main)
        for i, (strm, scr) in enumerate(((main, scratch), (side, scratch2))):
            y, xn, a1, a2 = kept[i]
            with torch.cuda.stream(strm):
                check(lib.apx_mlp_backward(_p(self.q[i].params), _p(self.c_g[i * self.n1:(i + 1) * self.n1]), self.D + self.A, self.H, 1, _p(xn), _p(a1), _p(a2),
                                           _p(dq[i]), B, None, _p(scr), _stream()))
        main.wait_stream(side)
        self.t_c += 1
        self._adam(self.critic_flat, self.c_m, self.c_v, self.c_g, self.c_lr, self.t_c)
        pi_loss = None
        if it % policy_freq == 0:
            pre, xs, s1, s2 = self.actor.forward(state, keep=True)
            xin_pi = self._cat(state, pre)
            q1, xq, q11, q12 = self.q[0].forward(xin_pi, keep=True)
            pi_loss = -q1.mean()
            dy = torch.full((B,), -1.0 / B, dtype=torch.float32, device=self.device)
            dx = torch.empty(B, self.D + self.A, dtype=torch.float32, device=self.device)
            check(lib.apx_mlp_backward(_p(self.q[0].params), None, self.D + self.A, self.H, 1, _p(xq), _p(q11), _p(q12), _p(dy), B, _p(dx), _p(scratch), _stream()))
            dpre = torch.empty(B, self.A, dtype=torch.float32, device=self.device)
            check(lib.apx_td3_actor_grad(_p(dx), _p(pre), self.max_action, B, self.D, self.A, _p(dpre), _stream()))
            self.a_g.zero_()
            check(lib.apx_mlp_backward(_p(self.actor.params), _p(self.a_g), self.D, self.H, self.A, _p(xs), _p(s1), _p(s2), _p(dpre), B, None, _p(scratch), _stream()))
            self.t_a += 1
            self._adam(self.actor.params, self.a_m, self.a_v, self.a_g, self.a_lr, self.t_a)
            check(lib.apx_polyak(_p(self.critic_t_flat), _p(self.critic_flat), self.critic_flat.numel(), float(tau), _stream()))
            check(lib.apx_polyak(_p(self.actor_t.params), _p(self.actor.params), self.actor.n, float(tau), _stream()))
        return stats, pi_loss
