"""apex_amd — MI355X-native engine for osudrl/apex's Cassie-v0 PPO hot path.

The compute lives in lib/libapx.so (hand-written HIP for gfx950 behind the C ABI of include/apx.h).  This package
is the thin host side: ctypes bindings (`_lib`), tensor-level wrappers (`engine`), the batched env seam
(`vecenv`), the PPO driver mirroring rl/algos/ppo.py (`ppo`) and the run-dir layout (`log`).
There is no CPU fallback: every compute call raises if libapx.so or a GPU is missing.
"""
__version__ = "0.1.0"
