// apex_amd/csrc/learner.hip compiled for the HOST under tools/hipemu/hip/hip_runtime.h (workgroups one after the other, wave collectives emulated lane-exactly): the
// C ABI of the learner half (apx_returns_scan, apx_adv_*, apx_mlp_forward / backward, apx_ppo_minibatch, apx_clip_adam, apx_lstm_*, apx_td3_*) on host pointers, for
// tests/test_kernel_emulation.py.  The include is tools/hipemu/_build/learner_emul.hip = the product source with ONE line rewritten by build.sh: the dynamic LDS
// declaration `extern __shared__ float fls[];` has no host spelling and becomes a pointer to the emulation's LDS segment.  Test infrastructure only.
void apx_set_error(const char* fmt, ...);
#include "_build/learner_emul.hip"
