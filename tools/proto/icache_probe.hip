// Micro-benchmark (round 4): does straight-line code larger than the instruction cache run at the full
// VALU issue rate when the launch looks like env_step_kernel's (single-wave workgroups, 40 KB of LDS each -> one wave per SIMD,
// all waves executing the same cyclic instruction stream)?
// Body = KB8 x 8 KB of independent v_fma_f32 (VOP3, 8 bytes each; VOP2 4-byte variant with -DSHORT) looped `reps` times.
// Output: ns per instruction for each code size and launch shape; the ratio to the 8 KB row is the fetch penalty.
// Build: hipcc -O3 --offload-arch=gfx950 tools/proto/icache_probe.hip -o tools/proto/icache_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <utility>
#ifdef SHORT
#define I8 asm volatile("v_add_f32 %0, %8, %0\n v_add_f32 %1, %8, %1\n v_add_f32 %2, %8, %2\n v_add_f32 %3, %8, %3\n v_add_f32 %4, %8, %4\n v_add_f32 %5, %8, %5\n v_add_f32 %6, %8, %6\n v_add_f32 %7, %8, %7\n" \
    : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(m), "v"(c));
#define IBYTES 4
#else
#define I8 asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n" \
    : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(m), "v"(c));
#define IBYTES 8
#endif
#define I64 I8 I8 I8 I8 I8 I8 I8 I8
#define I512 I64 I64 I64 I64 I64 I64 I64 I64
#define I1024 I512 I512

template <int N> __device__ __forceinline__ void body(float& r0, float& r1, float& r2, float& r3, float& r4, float& r5, float& r6, float& r7, float m, float c) {
    if constexpr (N > 0) { I1024 body<N - 1>(r0, r1, r2, r3, r4, r5, r6, r7, m, c); }
}
template <int NBLK> __global__ __launch_bounds__(64) void k(float* out, int reps, float m, float c) {
    extern __shared__ float lds[];
    float r0 = threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7;
    for (int r = 0; r < reps; ++r) {
        body<NBLK>(r0, r1, r2, r3, r4, r5, r6, r7, m, c);
    }
    if (m == 123.f) lds[threadIdx.x] = r0;
    out[blockIdx.x * 64 + threadIdx.x] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7;
}
template <int NBLK> void run(float* out, int grid, int ldsb, const char* shape) {
    const long per_rep = 1024L * NBLK;
    int reps = (int)(4000000L / per_rep); if (reps < 2) reps = 2;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipFuncSetAttribute((const void*)k<NBLK>, hipFuncAttributeMaxDynamicSharedMemorySize, ldsb);
    hipLaunchKernelGGL(k<NBLK>, dim3(grid), dim3(64), ldsb, 0, out, reps, 1.0001f, 0.5f);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int t = 0; t < 3; ++t) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<NBLK>, dim3(grid), dim3(64), ldsb, 0, out, reps, 1.0001f, 0.5f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    printf("%-28s code %4ld KB  %8ld instr/wave  %7.3f ms  %6.3f ns/instr\n", shape, per_rep * IBYTES / 1024, per_rep * reps, best, best * 1e6 / (per_rep * reps));
}
template <int NBLK> void shapes(float* out) {
    run<NBLK>(out, 128, 40848, "128 wg (<=1 wave / CU)");
    run<NBLK>(out, 256, 40848, "256 wg (1 wave / CU)");
    run<NBLK>(out, 1024, 40848, "1024 wg (1 wave / SIMD)");
    run<NBLK>(out, 2048, 20000, "2048 wg (2 waves / SIMD)");
}
int main() {
    float* out; hipMalloc(&out, 2048 * 64 * 4);
    printf("instruction bytes: %d\n", IBYTES);
    shapes<1>(out); shapes<2>(out); shapes<4>(out); shapes<6>(out); shapes<8>(out); shapes<10>(out); shapes<12>(out); shapes<16>(out); shapes<24>(out); shapes<32>(out);
    return 0;
}
