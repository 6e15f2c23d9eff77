"""ctypes binding of libapx.so (include/apx.h).  Fails loudly when the HIP library is missing."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# APX_LIB selects an A/B build of the SAME HIP library (make VARIANT=...; kernel experiments under tools/); there is no non-HIP path
LIB_PATH = os.environ.get("APX_LIB") or os.path.join(_HERE, "lib", "libapx.so")

c_f32p = C.c_void_p
c_ptr = C.c_void_p


class PpoArgs(C.Structure):
    """struct apx_ppo_args (include/apx.h)."""
    _fields_ = [
        ("actor", c_ptr), ("actor_m", c_ptr), ("actor_v", c_ptr), ("actor_grad", c_ptr),
        ("critic", c_ptr), ("critic_m", c_ptr), ("critic_v", c_ptr), ("critic_grad", c_ptr),
        ("D", C.c_int), ("H", C.c_int), ("A", C.c_int),
        ("obs", c_ptr), ("act", c_ptr), ("ret", c_ptr), ("adv", c_ptr), ("old_mu", c_ptr), ("idx", c_ptr),
        ("mb", C.c_int64),
        ("obs_mean", c_ptr), ("obs_std", c_ptr),
        ("obs_sign_perm", c_ptr), ("clock_mask", C.c_uint64), ("act_sign_perm", c_ptr),
        ("fixed_std", C.c_float), ("clip", C.c_float), ("entropy_coeff", C.c_float), ("grad_clip", C.c_float),
        ("lr", C.c_float), ("adam_eps", C.c_float), ("mirror_coeff", C.c_float),
        ("adam_t", C.c_int), ("grad_only", C.c_int),
        ("workspace", c_ptr), ("workspace_bytes", C.c_size_t),
        ("scalars_out", c_ptr),
    ]


class Td3Args(C.Structure):
    """struct apx_td3_args (include/apx.h)."""
    _fields_ = [
        ("actor", c_ptr), ("actor_t", c_ptr), ("actor_m", c_ptr), ("actor_v", c_ptr),
        ("critic", c_ptr), ("critic_t", c_ptr), ("critic_m", c_ptr), ("critic_v", c_ptr),
        ("D", C.c_int), ("H", C.c_int), ("A", C.c_int),
        ("state", c_ptr), ("next_state", c_ptr), ("action", c_ptr), ("reward", c_ptr), ("notdone", c_ptr),
        ("ind", c_ptr), ("noise", c_ptr),
        ("B", C.c_int64), ("U", C.c_int64),
        ("it0", C.c_int), ("policy_freq", C.c_int),
        ("max_action", C.c_float), ("noise_clip", C.c_float), ("discount", C.c_float), ("tau", C.c_float),
        ("a_lr", C.c_float), ("c_lr", C.c_float), ("adam_eps", C.c_float),
        ("t_a", C.c_int), ("t_c", C.c_int),
        ("workspace", c_ptr), ("workspace_bytes", C.c_size_t),
        ("stats_out", c_ptr),
    ]


class EnvCfg(C.Structure):
    """struct apx_env_cfg (include/apx.h)."""
    _fields_ = [
        ("n_envs", C.c_int), ("simrate", C.c_int), ("dynamics_randomization", C.c_int), ("reward_kind", C.c_int),
        ("stance_mode", C.c_int), ("have_incentive", C.c_int), ("max_traj_len", C.c_int),
        ("seed", C.c_uint64), ("device", C.c_int), ("pgs_iters", C.c_int), ("env_id_base", C.c_int), ("env_kind", C.c_int), ("command_profile", C.c_int), ("est_lifetime", C.c_int), ("input_profile", C.c_int), ("reserved", C.c_int * 2),
    ]


# name -> (restype, argtypes); every symbol include/apx.h declares
SIGNATURES = {
    "apx_version": (C.c_int, []),
    "apx_last_error": (C.c_char_p, []),
    "apx_device_count": (C.c_int, []),
    "apx_returns_scan": (C.c_int, [c_ptr, c_ptr, c_ptr, c_ptr, C.c_double, C.c_int, C.c_int, c_ptr, c_ptr]),
    "apx_adv_moments": (C.c_int, [c_ptr, c_ptr, C.c_int64, c_ptr, c_ptr]),
    "apx_adv_apply": (C.c_int, [c_ptr, c_ptr, C.c_int64, C.c_double, C.c_double, C.c_double, c_ptr, c_ptr]),
    "apx_adv_apply_moments": (C.c_int, [c_ptr, c_ptr, C.c_int64, c_ptr, C.c_double, c_ptr, c_ptr]),
    "apx_mlp_param_count": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "apx_mlp_backward": (C.c_int, [c_ptr, c_ptr, C.c_int, C.c_int, C.c_int, c_ptr, c_ptr, c_ptr, c_ptr, C.c_int64, c_ptr, c_ptr, c_ptr]),
    "apx_polyak": (C.c_int, [c_ptr, c_ptr, C.c_int64, C.c_float, c_ptr]),
    "apx_td3_cat_action": (C.c_int, [c_ptr, c_ptr, c_ptr, C.c_float, C.c_float, C.c_int64, C.c_int, C.c_int, c_ptr, c_ptr]),
    "apx_td3_critic_loss": (C.c_int, [c_ptr] * 6 + [C.c_float, C.c_int64, c_ptr, c_ptr, c_ptr, c_ptr]),
    "apx_td3_actor_grad": (C.c_int, [c_ptr, c_ptr, C.c_float, C.c_int64, C.c_int, C.c_int, c_ptr, c_ptr]),
    "apx_ppo_loss": (C.c_int, [c_ptr] * 9 + [C.c_int64, C.c_int, C.c_float, C.c_float, C.c_float] + [c_ptr] * 6),
    "apx_lstm_param_count": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "apx_lstm_workspace_floats": (C.c_size_t, [C.c_int, C.c_int64, C.c_int, C.c_int]),
    "apx_lstm_bwd_scratch_floats": (C.c_size_t, [C.c_int, C.c_int64, C.c_int, C.c_int]),
    "apx_lstm_forward": (C.c_int, [c_ptr, C.c_int, C.c_int, C.c_int, C.c_int, c_ptr, C.c_int, C.c_int64, c_ptr, c_ptr, c_ptr, c_ptr]),
    "apx_lstm_step_pack_floats": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "apx_lstm_step_pack": (C.c_int, [c_ptr, C.c_int, C.c_int, C.c_int, C.c_int, c_ptr, c_ptr]),
    "apx_lstm_step": (C.c_int, [c_ptr, C.c_int, C.c_int, C.c_int, C.c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, C.c_int64, c_ptr, c_ptr, c_ptr, C.c_float, c_ptr]),
    "apx_rec_gather": (C.c_int, [c_ptr, c_ptr, c_ptr, C.c_int64, C.c_int, C.c_int64, C.c_int, C.c_int] + [c_ptr] * 5 + [C.c_uint64] + [c_ptr] * 10),
    "apx_lstm_backward": (C.c_int, [c_ptr, c_ptr, C.c_int, C.c_int, C.c_int, C.c_int, c_ptr, C.c_int, C.c_int64, c_ptr, c_ptr, c_ptr, c_ptr]),
    "apx_mlp_forward": (C.c_int, [c_ptr, C.c_int, C.c_int, C.c_int, c_ptr, C.c_int64, c_ptr, c_ptr, C.c_uint64, c_ptr,
                                  c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr]),
    "apx_ppo_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int, C.c_int, C.c_int]),
    "apx_ppo_minibatch": (C.c_int, [C.POINTER(PpoArgs), c_ptr]),
    "apx_ppo_epoch_supported": (C.c_int, [C.c_int64, C.c_int, C.c_int, C.c_int]),
    "apx_ppo_epoch_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int]),
    "apx_ppo_epoch": (C.c_int, [C.POINTER(PpoArgs), c_ptr, C.c_int64, c_ptr]),
    "apx_td3_updates_supported": (C.c_int, [C.c_int64, C.c_int, C.c_int, C.c_int]),
    "apx_td3_updates_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int]),
    "apx_td3_updates": (C.c_int, [C.POINTER(Td3Args), c_ptr]),
    "apx_grid_barrier_selftest": (C.c_int, [C.c_int, C.c_int, C.c_int, c_ptr, c_ptr, c_ptr]),
    "apx_clip_adam": (C.c_int, [c_ptr, c_ptr, c_ptr, c_ptr, C.c_int64, C.c_float, C.c_float, C.c_float, C.c_float,
                                C.c_int, c_ptr, c_ptr]),
    "apx_env_default_cfg": (None, [C.POINTER(EnvCfg)]),
    "apx_env_create": (C.c_int, [C.POINTER(EnvCfg), C.POINTER(c_ptr)]),
    "apx_env_destroy": (C.c_int, [c_ptr]),
    "apx_env_set_hfield": (C.c_int, [c_ptr, c_ptr, C.c_int, C.c_int, c_ptr, c_ptr]),
    "apx_env_reset": (C.c_int, [c_ptr, c_ptr, c_ptr, c_ptr]),
    "apx_env_prepare_resets": (C.c_int, [c_ptr, c_ptr]),
    "apx_env_set_refill": (C.c_int, [c_ptr, C.c_int]),
    "apx_env_set_complete_rows": (C.c_int, [c_ptr, C.c_int]),
    "apx_env_update_speed": (C.c_int, [c_ptr, c_ptr, c_ptr, c_ptr]),
    "apx_env_reset_for_test": (C.c_int, [c_ptr, c_ptr, C.c_int, c_ptr]),
    "apx_env_apply_force": (C.c_int, [c_ptr, c_ptr, c_ptr]),
    "apx_env_apply_force_body": (C.c_int, [c_ptr, c_ptr, C.c_int, c_ptr]),
    "apx_env_step_basic": (C.c_int, [c_ptr, c_ptr, c_ptr, c_ptr]),
    "apx_env_timing": (C.c_int, [c_ptr, C.c_int]),
    "apx_env_timing_read": (C.c_int, [c_ptr, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.c_int]),
    "apx_rollout": (C.c_int, [c_ptr, c_ptr, C.c_int, c_ptr, c_ptr, C.c_float, c_ptr, C.c_int] + [c_ptr] * 8),
    "apx_rollout_td3": (C.c_int, [c_ptr, c_ptr, C.c_int, C.c_float, C.c_float, c_ptr, C.c_int, C.c_int] + [c_ptr] * 8),
    "apx_rollout_lstm": (C.c_int, [c_ptr, c_ptr, C.c_int, C.c_int, c_ptr, c_ptr, C.c_float, c_ptr, C.c_int] + [c_ptr] * 8),
    "apx_env_step": (C.c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, C.c_int, c_ptr]),
    "apx_env_get_state": (C.c_int, [c_ptr, c_ptr, c_ptr, c_ptr]),
    "apx_env_set_state": (C.c_int, [c_ptr, c_ptr, c_ptr, c_ptr]),
    "apx_env_get_field": (C.c_int, [c_ptr, C.c_char_p, c_ptr, c_ptr]),
    "apx_env_set_field": (C.c_int, [c_ptr, C.c_char_p, c_ptr, c_ptr]),
}

_lib = None


class ApxError(RuntimeError):
    pass


def load():
    """Load libapx.so and bind every declared symbol; raises if the library or a symbol is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ApxError(f"{LIB_PATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                       f"(there is no CPU fallback)")
    import torch  # noqa: F401  (first: libapx.so must bind to the HIP runtime torch ships - loaded the other way round, the process holds two runtimes and the library's sees no device)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        if os.environ.get("APX_LIB_OLD_ABI") == "1" and not hasattr(lib, name):      # tools only: A/B timing against a library built from an older commit (APX_LIB=...), which lacks the newer entry points
            continue
        fn = getattr(lib, name)          # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        raise ApxError(f"libapx error {rc}: {load().apx_last_error().decode()}")
