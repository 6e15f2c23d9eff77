// Does a second (idle) wave in the workgroup change global-load latency seen by wave 0?  Design input for the
// dual-wave env kernel.  Wave 0 runs a dependent load chain over an L2-resident buffer.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>   // 0: 64-thread block; 1: 128 threads, wave 1 waits at a barrier; 2: 128 threads, wave 1 exits at once
__global__ void k(const int* __restrict__ next, int* out, long long* cyc, int iters) {
    extern __shared__ float4 lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (MODE == 2 && wave == 1) return;
    if (wave == 0) {
        int p = (blockIdx.x * 64 + lane) * 16;
        const long long t0 = clock64();
        for (int i = 0; i < iters; ++i) p = next[p];
        const long long t1 = clock64();
        out[blockIdx.x * 64 + lane] = p;
        if (lane == 0) cyc[blockIdx.x] = t1 - t0;
    }
    if (MODE == 1) __syncthreads();
    if (threadIdx.x == 9999) lds[0] = make_float4(0, 0, 0, 0);
}
int main() {
    const int n = 1 << 20;   // 4 MB of ints: L2 / MALL resident
    int* h = new int[n];
    for (int i = 0; i < n; ++i) h[i] = (int)(((long long)i * 7919 + 12345) % n);
    int *d, *out; long long* cyc;
    hipMalloc(&d, n * 4); hipMalloc(&out, 64 * 64 * 4); hipMalloc(&cyc, 64 * 8);
    hipMemcpy(d, h, n * 4, hipMemcpyHostToDevice);
    const int iters = 2000;
    long long hc[64];
    auto report = [&](const char* name) {
        hipDeviceSynchronize(); hipMemcpy(hc, cyc, sizeof(hc), hipMemcpyDeviceToHost);
        double s = 0; for (int i = 0; i < 64; ++i) s += hc[i];
        printf("%-44s %.0f cycles per dependent load\n", name, s / 64 / iters);
    };
    for (size_t lds : {(size_t)0, (size_t)161792}) {
        hipFuncSetAttribute((const void*)k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 161792);
        hipFuncSetAttribute((const void*)k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 161792);
        hipFuncSetAttribute((const void*)k<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 161792);
        printf("dynamic LDS %zu B\n", lds);
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(k<0>, dim3(64), dim3(64), lds, 0, d, out, cyc, iters); report("  64-thread workgroup");
            hipLaunchKernelGGL(k<1>, dim3(64), dim3(128), lds, 0, d, out, cyc, iters); report("  128 threads, wave 1 parked at s_barrier");
            hipLaunchKernelGGL(k<2>, dim3(64), dim3(128), lds, 0, d, out, cyc, iters); report("  128 threads, wave 1 exits immediately");
        }
    }
    return 0;
}
