// micro-benchmark: does straight-line code larger than the instruction cache stream from L2 at full issue rate
// when only ONE wave runs per CU?  (design input for the env kernel)
#include <hip/hip_runtime.h>
#include <utility>
#include <cstdio>
struct T { int parent[32]; int madr[33]; int depth[32]; int anc[32][16]; };
constexpr T make() {
    T t{};
    constexpr int par[32] = {-1,0,1,2,3,4, 5,6,7,8,9,10,8,12,13,14,14,16,14, 5,19,20,21,22,23,21,25,26,27,27,29,27};
    int tot = 0;
    for (int i = 0; i < 32; ++i) { t.parent[i] = par[i]; t.madr[i] = tot; int d = 0; for (int j = i; j >= 0; j = par[j]) { t.anc[i][d] = j; ++d; ++tot; } t.depth[i] = d; }
    t.madr[32] = tot;
    return t;
}
constexpr T TB = make();
template <int B, int E, class F> __device__ __forceinline__ void static_for(F&& f) { if constexpr (B < E) { f(std::integral_constant<int, B>{}); static_for<B + 1, E>(f); } }
template <int B, int E, class F> __device__ __forceinline__ void static_rfor(F&& f) { if constexpr (B < E) { f(std::integral_constant<int, E - 1>{}); static_rfor<B, E - 1>(f); } }

template <int COPY> __device__ __forceinline__ void factor(float (&LD)[307], float salt) {
    static_rfor<0, 32>([&](auto K) {
        constexpr int k = K, kk = TB.madr[k];
        const float dinv = __frcp_rn(LD[kk] + salt * COPY);
        static_for<1, TB.depth[k]>([&](auto A) {
            constexpr int a = A, i = TB.anc[k][a], ki = kk + a;
            const float tmp = LD[ki] * dinv;
            static_for<0, TB.depth[i]>([&](auto J) { constexpr int jj = J; LD[TB.madr[i] + jj] -= tmp * LD[ki + jj]; });
            LD[ki] = tmp;
        });
    });
}
template <int NCOPY> __global__ __launch_bounds__(64) void k(const float* M, float* out, int n, int reps, float salt) {
    const int env = blockIdx.x * 64 + threadIdx.x;
    float LD[307];
    static_for<0, 307>([&](auto I) { LD[I] = M[(size_t)I * n + env]; });
    for (int r = 0; r < reps; ++r) {
        static_for<0, NCOPY>([&](auto C) { factor<C>(LD, salt); static_for<0, 32>([&](auto D) { LD[TB.madr[D]] += 3.0f; }); });
    }
    static_for<0, 307>([&](auto I) { out[(size_t)I * n + env] = LD[I]; });
}
int main() {
    const int n = 4096;
    float *M, *out;
    hipMalloc(&M, 307 * n * 4); hipMalloc(&out, 307 * n * 4);
    hipMemset(M, 0, 307 * n * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](auto kern, int reps, const char* name, int ncopy) {
        hipLaunchKernelGGL(kern, dim3(n / 64), dim3(64), 0, 0, M, out, n, reps, 0.f);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(n / 64), dim3(64), 0, 0, M, out, n, reps, 0.f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%s: %d factorisations in %.3f ms -> %.2f us each (%.0f cycles @2.4GHz)\n", name, reps * ncopy, ms, ms * 1e3 / (reps * ncopy), ms * 1e-3 / (reps * ncopy) * 2.4e9);
    };
    run(k<1>, 256, "1 copy  (~17 KB code, I$ resident)", 1);
    run(k<4>, 64, "4 copies (~70 KB)", 4);
    run(k<16>, 16, "16 copies (~280 KB, streams from L2)", 16);
    return 0;
}
