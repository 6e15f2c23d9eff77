// apex_amd/csrc/learner.hip compiled for the HOST under tools/hipemu/hip/hip_runtime.h (workgroups one after the other, wave collectives emulated lane-exactly): the
// C ABI of the learner half (apx_returns_scan, apx_adv_*, apx_mlp_forward / backward, apx_ppo_minibatch, apx_clip_adam, apx_lstm_*, apx_td3_*) on host pointers, for
// tests/test_kernel_emulation_learner.py.  The product source as it is (round 6: its one dynamic-LDS declaration goes through APX_DYNAMIC_LDS, gfx950/dynamic_lds.h, whose host
// twin hands out the emulation's LDS segment; until then build.sh rewrote that line with sed).  Test infrastructure only.
void apx_set_error(const char* fmt, ...);
#include "../../apex_amd/csrc/learner.hip"
