// ThreadSanitizer harness of the grid barrier (apex_amd/csrc/mlp_tiles.h grid_barrier through barrier_selftest.hip): workgroups = threads, see tsan/hip/hip_runtime.h.
//   tsan_barrier <workgroups> <phases> <words> [break]     break = 1: a positive control - the run uses a barrier whose arrival is counted RELAXED (no release), which TSan
//   must report as a data race on the word block (exit code 66).
#include <cstdarg>
#include <cstdio>
static char g_err[512];
void apx_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap); }
#ifdef TSAN_BREAK_RELEASE
#define __ATOMIC_RELEASE_SAVED __ATOMIC_RELEASE
#undef __ATOMIC_RELEASE
#define __ATOMIC_RELEASE __ATOMIC_RELAXED
#endif
#include "../../apex_amd/csrc/barrier_selftest.hip"
int main(int argc, char** argv) {
    const int G = argc > 1 ? atoi(argv[1]) : 8, phases = argc > 2 ? atoi(argv[2]) : 2000, words = argc > 3 ? atoi(argv[3]) : 4096;
    std::vector<unsigned> ws(2 + words);
    unsigned long long res[4];
    const int rc = apx_grid_barrier_selftest(G, phases, words, ws.data(), res, nullptr);
    printf("rc %d stale %llu watchdog %llu phases %llu sum %llu (expect 0 0 %d %lld)\n", rc, res[0], res[1], res[2], res[3], phases, (long long)G * phases);
    return rc == 0 && res[0] == 0 && res[1] == 0 && res[2] == (unsigned long long)phases && res[3] == (unsigned long long)G * phases ? 0 : 1;
}
