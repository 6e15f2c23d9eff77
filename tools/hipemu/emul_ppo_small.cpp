// apex_amd/csrc/ppo_small.hip compiled for the HOST under tools/hipemu/hip/hip_runtime.h: the kernel's own source, one workgroup, wave collectives emulated
// lane-exactly.  Exports the same apx_ppo_epoch entry points on host pointers (tests/test_kernel_emulation.py drives them through ctypes against the reference's
// golden G4b).  Build: tools/hipemu/build.sh.  Test infrastructure only.
#include <cstdarg>
#include <cstdio>
static char g_err[512];
void apx_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap); }
extern "C" const char* apx_emul_last_error() { return g_err; }
#include "../../apex_amd/csrc/ppo_small.hip"
#include "../../apex_amd/csrc/barrier_selftest.hip"      // the grid barrier under stress, same emulation (forked workgroups on MAP_SHARED buffers)
// the grid of the next launches (the kernel's host code reads APX_PPO_EPOCH_WGS once; the emulation overrides the launch instead)
extern "C" void apx_emul_set_workgroups(int g) { hipemu::g_force_grid = g; }
extern "C" int apx_emul_last_grid() { return (int)hipemu::g_grid.x; }
// launches so far of the kernels whose launch-site expression contains `part` (e.g. "gemm_f32_128_kernel<EPI_MASK>")
extern "C" long apx_emul_launches(const char* part) {
    long n = 0;
    for (const auto& kv : hipemu::g_launches) if (kv.first.find(part) != std::string::npos) n += kv.second;
    return n;
}
