// Building blocks of the persistent small-batch training kernels (ppo_small.hip: an epoch of PPO optimiser steps in one launch; td3_small.hip: a block of TD3 updates in
// one launch): a grid-wide barrier with the agent-scope cache maintenance that crosses XCDs, and 16 x 16 output tiles on v_mfma_f32_16x16x4_f32 whose operands go from L2
// straight into the MFMA registers.  gfx950 only.
#pragma once
#include "apx_common.h"
#include <cstdlib>

typedef float floatx4 __attribute__((ext_vector_type(4)));

namespace tiles {

__device__ __forceinline__ double wsum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
template <class T> __device__ __forceinline__ T ld_agent(const T* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// every workgroup of the grid has arrived `target` times in total.  Release: each wave writes its L2 lines back before the workgroup's arrival is counted; acquire:
// each wave drops its stale lines afterwards.  Watchdog: a workgroup that has spun ~2 s (a grid that is not resident as a whole cannot finish) raises ctr[1] and
// every barrier from then on falls through; the kernel ends with NaN scalars instead of hanging the device.
constexpr unsigned SPIN_LIMIT = 1u << 21;
__device__ __forceinline__ void grid_barrier(unsigned* ctr, unsigned target) {
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if ((++spins & 1023u) == 0 && (spins >= SPIN_LIMIT || __hip_atomic_load(ctr + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
                __hip_atomic_store(ctr + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
    }
    __syncthreads();
    __threadfence();
}

__device__ __forceinline__ void ld4(const float* p, float (&o)[4]) {      // 16-byte aligned
    const floatx4 v = *(const floatx4*)p;
    o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = v[3];
}
__device__ __forceinline__ void zero4(float (&o)[4]) { o[0] = o[1] = o[2] = o[3] = 0.f; }

// One 16 x 16 tile of C = A B on one wave.  Contraction index of (chunk kc, lane group g = lane >> 4, j) is kk = 16 kc + 4 g + j for BOTH operands, so an operand
// whose contraction axis is contiguous in memory is one 16-byte load per chunk.  fa(kc, a): a[j] = A(i = lane & 15, kk); fb(kc, b): b[j] = B(kk, n = lane & 15);
// entries with kk beyond the contraction length must come back as zero.  The NB chunks of a batch are all fetched before the first MFMA of the batch.
// Result: acc[v] = C(i = 4 g + v, n = lane & 15).  SUMA: *asum = sum of the lane's A entries (column sums of A^T for the bias gradients).
template <int NB, bool SUMA, class FA, class FB>
__device__ __forceinline__ floatx4 wave_tile(int kc0, int kc1, FA fa, FB fb, float* asum) {
    floatx4 acc = {0.f, 0.f, 0.f, 0.f};
    float as = 0.f;
    for (int k0 = kc0; k0 < kc1; k0 += NB) {
        float a[NB][4], b[NB][4];
#pragma unroll
        for (int q = 0; q < NB; ++q) { fa(k0 + q, a[q]); fb(k0 + q, b[q]); }
        __builtin_amdgcn_sched_barrier(0);      // (left alone, the scheduler sinks each load to just before its MFMAs: one L2 round trip per chunk)
#pragma unroll
        for (int q = 0; q < NB; ++q)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q][j], b[q][j], acc, 0, 0, 0);
                if constexpr (SUMA) as += a[q][j];
            }
    }
    if constexpr (SUMA) *asum = as;
    return acc;
}

// C = A^T B over the ROWS of two row-major matrices (weight / bias gradients: the contraction index is the sample row): Ac = &A[4 g][i], Bc = &B[4 g][n] of the lane.
// No guards: every row buffer of the workspace is padded to a multiple of 64 rows that stay zero (the launch clears the workspace), chunks = padded rows / 16.
template <int NB>
__device__ __forceinline__ floatx4 rows_tile_nb(const float* Ac, int lda, const float* Bc, int ldb, int chunks, float* asum) {
    return wave_tile<NB, true>(0, chunks,
        [&](int kc, float (&a)[4]) {
#pragma unroll
            for (int j = 0; j < 4; ++j) a[j] = Ac[(size_t)(16 * kc + j) * lda];
        },
        [&](int kc, float (&b)[4]) {
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = Bc[(size_t)(16 * kc + j) * ldb];
        }, asum);
}
__device__ __forceinline__ floatx4 rows_tile(const float* Ac, int lda, const float* Bc, int ldb, int chunks, float* asum) {
    return (chunks & 7) == 0 ? rows_tile_nb<8>(Ac, lda, Bc, ldb, chunks, asum) : rows_tile_nb<4>(Ac, lda, Bc, ldb, chunks, asum);
}

// Host side: launch of a persistent kernel whose workgroups wait for each other at grid_barrier.  The grid must be resident as a whole: checked against the occupancy
// the runtime reports for THIS kernel (registers, LDS) before anything is enqueued, and launched as a cooperative kernel (the runtime refuses a grid it cannot hold
// resident and serialises it against other cooperative launches of the device).  APX_COOP_LAUNCH=0 keeps the occupancy check and uses a plain launch (A/B knob).
// The barrier's watchdog stays as the second line behind both.
template <class Args>
inline int launch_resident(void (*kernel)(Args), int G, int block, hipStream_t s, Args& S, const char* what) {
    int dev = 0, cus = 0, coop = 0, per_cu = 0;
    APX_HIP(hipGetDevice(&dev));
    APX_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    APX_HIP(hipDeviceGetAttribute(&coop, hipDeviceAttributeCooperativeLaunch, dev));
    APX_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, block, 0));
    if ((long)per_cu * cus < G) {
        apx_set_error("%s: a grid of %d workgroups cannot be resident at once on this device (%d CUs x %d workgroups)", what, G, cus, per_cu);
        return APX_E_ARG;
    }
    static const bool plain = getenv("APX_COOP_LAUNCH") && atoi(getenv("APX_COOP_LAUNCH")) == 0;
    if (coop && !plain) {
        void* params[1] = {(void*)&S};
        APX_HIP(hipLaunchCooperativeKernel(kernel, dim3(G), dim3(block), params, 0u, s));
    } else {
        hipLaunchKernelGGL(kernel, dim3(G), dim3(block), 0, s, S);
        APX_LAUNCH_CHECK();
    }
    return APX_OK;
}

}  // namespace tiles
