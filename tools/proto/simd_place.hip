// where do the two waves of a 128-thread / 160 KB-LDS workgroup land?  (same SIMD or different SIMDs)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(128) void k(unsigned* out) {
    extern __shared__ float4 lds[];
    lds[threadIdx.x] = make_float4(0, 0, 0, 0);
    const unsigned hwid = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));   // HW_REG_HW_ID, full 32 bits
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 2 + (threadIdx.x >> 6)] = hwid;
}
int main() {
    unsigned* d; hipMalloc(&d, 64 * 2 * 4);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 161792);
    hipLaunchKernelGGL(k, dim3(64), dim3(128), 161792, 0, d);
    unsigned h[128]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int same = 0;
    for (int b = 0; b < 64; ++b) {
        const unsigned a = h[2 * b], c = h[2 * b + 1];
        const int simd_a = (a >> 4) & 3, simd_c = (c >> 4) & 3, cu_a = (a >> 8) & 15, cu_c = (c >> 8) & 15;
        if (b < 8) printf("block %d: wave0 hwid %08x simd %d cu %d | wave1 hwid %08x simd %d cu %d\n", b, a, simd_a, cu_a, c, simd_c, cu_c);
        same += simd_a == simd_c;
    }
    printf("blocks with both waves on the same SIMD: %d / 64\n", same);
    return 0;
}
