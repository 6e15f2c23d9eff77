// apex_amd/csrc/env.hip (the env step / reset / one-launch rollout kernels with cassie_lane.h, cassie_complete.h, estimator_lane.h) compiled for the HOST under
// tools/hipemu/hip/hip_runtime.h and tools/hipemu/gfx950/lane_ops.h: the kernels' own source, every wave collective emulated lane-exactly.  Exports the same apx_env_*
// / apx_rollout* entry points on host pointers (tests/test_kernel_emulation_env.py drives them through apex_amd/vecenv.py).  Test infrastructure only.
void apx_set_error(const char* fmt, ...);
#include "../../apex_amd/csrc/env.hip"
