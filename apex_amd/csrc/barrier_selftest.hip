// apx_grid_barrier_selftest: the grid barrier of the persistent trainers (mlp_tiles.h grid_barrier, used by apx_ppo_epoch and apx_td3_updates) under stress, on purpose
// instead of by the few hundred barriers of a golden run.  Phase p: ONE workgroup overwrites a block of words with a pattern of p by plain stores (they land in the L2
// of its XCD), every workgroup passes the barrier, every workgroup reads the block back by plain loads (through the L2 of ITS XCD, which still holds the lines of the
// phase before) and counts the words that are not the pattern of p, second barrier (the next writer may not start before the last reader is done).  A barrier that
// counts wrongly hangs (the watchdog ends it), one whose release / acquire cache maintenance is missing or too narrow reads stale words.  gfx950 only.
#include "mlp_tiles.h"

namespace {

struct StressArgs {
    unsigned* bar;                 // [2]: arrival counter, watchdog flag (grid_barrier)
    unsigned* words;               // [n]
    unsigned long long* result;    // [4]: stale words seen, watchdog flag, phases completed by workgroup 0, sum over workgroups of phases completed
    int n, phases;
};

__device__ __forceinline__ unsigned pattern(unsigned p, unsigned i) { return (p + 1u) * 2654435761u ^ (i * 40503u + 0x9e3779b9u); }
// the writer of a phase: consecutive phases on different XCDs (workgroups are dealt round-robin over the 8 XCDs, so workgroup w sits on XCD w % 8)
__device__ __forceinline__ unsigned writer_of(unsigned p, unsigned G) { return (p * 37u + (p >> 3)) % G; }

__global__ __launch_bounds__(256) void barrier_stress_kernel(StressArgs S) {
    const unsigned G = gridDim.x, tid = threadIdx.x;
    unsigned arrivals = 0, stale = 0;
    int p = 0;
    for (; p < S.phases; ++p) {
        if (blockIdx.x == writer_of((unsigned)p, G))
            for (int i = (int)tid; i < S.n; i += 256) S.words[i] = pattern((unsigned)p, (unsigned)i);
        arrivals += G; tiles::grid_barrier(S.bar, arrivals);
        for (int i = (int)tid; i < S.n; i += 256) stale += S.words[i] != pattern((unsigned)p, (unsigned)i);
        arrivals += G; tiles::grid_barrier(S.bar, arrivals);
        if (tiles::ld_agent(S.bar + 1) != 0u) break;      // the watchdog fired: the counts from here on mean nothing
    }
    if (stale) atomicAdd(S.result, (unsigned long long)stale);
    if (tid == 0) {
        atomicAdd(S.result + 3, (unsigned long long)p);
        if (blockIdx.x == 0) { S.result[1] = tiles::ld_agent(S.bar + 1); S.result[2] = (unsigned long long)p; }
    }
}

}  // namespace

extern "C" int apx_grid_barrier_selftest(int workgroups, int phases, int n_words, unsigned* workspace, unsigned long long* result, void* stream) {
    APX_REQUIRE(workgroups >= 1 && workgroups <= 256 && phases >= 1 && phases <= (1 << 22) && n_words >= 1, "1..256 workgroups, 1..2^22 phases (a 32-bit arrival counter), >= 1 word");
    APX_REQUIRE(workspace && result, "workspace [2 + n_words] u32, result [4] u64 (device)");
    hipStream_t s = (hipStream_t)stream;
    APX_HIP(hipMemsetAsync(workspace, 0, (size_t)(2 + n_words) * sizeof(unsigned), s));
    APX_HIP(hipMemsetAsync(result, 0, 4 * sizeof(unsigned long long), s));
    StressArgs S;
    S.bar = workspace; S.words = workspace + 2; S.result = result; S.n = n_words; S.phases = phases;
    return tiles::launch_resident(barrier_stress_kernel, workgroups, 256, s, S, "apx_grid_barrier_selftest");
}
