// apex_amd/csrc/td3_small.hip compiled for the HOST under tools/hipemu/hip/hip_runtime.h (see emul_ppo_small.cpp).  Test infrastructure only.
void apx_set_error(const char* fmt, ...);
#include "../../apex_amd/csrc/td3_small.hip"
