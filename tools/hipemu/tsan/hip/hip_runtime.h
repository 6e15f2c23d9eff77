// Thread shim of HIP for ONE purpose: the grid barrier of apex_amd/csrc/mlp_tiles.h (through its stress kernel barrier_selftest.hip) under ThreadSanitizer.
// A workgroup is a std::thread of ONE "thread" (threadIdx.x == 0; __syncthreads is the identity), so the workgroups of a launch share one address space and TSan sees
// every plain load / store of the kernel and every atomic of the barrier with the memory ORDER the kernel source asks for.  What is modelled differently, because TSan has
// no model of stand-alone fences: __threadfence() is the identity and a RELAXED atomic load is promoted to ACQUIRE - it stands for "relaxed load in the spin loop + the
// agent-scope fence behind the loop" of grid_barrier.  The release side is the kernel's own fetch_add(RELEASE).  Test infrastructure only (tools/hipemu/tsan_barrier.sh).
#pragma once
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sched.h>
#include <thread>
#include <vector>
using std::min;
using std::max;

struct dim3 { unsigned x, y, z; dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {} };
typedef void* hipStream_t;
typedef int hipError_t;
constexpr hipError_t hipSuccess = 0;
inline const char* hipGetErrorString(hipError_t) { return "tsan shim"; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }
enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount, hipDeviceAttributeCooperativeLaunch };
inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
inline hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t a, int) { *v = a == hipDeviceAttributeMultiprocessorCount ? 256 : 1; return hipSuccess; }
template <class F> inline hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int* n, F, int, size_t) { *n = 1; return hipSuccess; }

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __restrict__

namespace shim {
inline thread_local dim3 t_bidx;
inline dim3 g_grid;
inline const dim3 g_tid(0, 0, 0);
}
#define threadIdx (shim::g_tid)
#define blockIdx (shim::t_bidx)
#define gridDim (shim::g_grid)

template <class A> inline hipError_t hipLaunchCooperativeKernel(void (*f)(A), dim3 grid, dim3, void** params, unsigned, hipStream_t) {
    const A a = *(const A*)params[0];
    shim::g_grid = grid;
    std::vector<std::thread> th;
    for (unsigned b = 0; b < grid.x; ++b) th.emplace_back([=]() { shim::t_bidx = dim3(b); f(a); });
    for (auto& t : th) t.join();
    return hipSuccess;
}
#define hipLaunchKernelGGL(kern, grid, block, shmem, stream, arg) do { void* p__[1] = {(void*)&(arg)}; hipLaunchCooperativeKernel(kern, grid, block, p__, 0u, stream); } while (0)

inline void __syncthreads() {}
inline void __threadfence() {}
#define __HIP_MEMORY_SCOPE_AGENT 0
template <class T> inline T shim_load(const T* p, int order) { return __atomic_load_n(p, order == __ATOMIC_RELAXED ? __ATOMIC_ACQUIRE : order); }
#define __hip_atomic_load(p, order, scope) shim_load(p, order)
#define __hip_atomic_store(p, v, order, scope) __atomic_store_n(p, v, order)
#define __hip_atomic_fetch_add(p, v, order, scope) __atomic_fetch_add(p, v, order)
inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
inline void __builtin_amdgcn_s_sleep(int) { sched_yield(); }
inline void __builtin_amdgcn_sched_barrier(int) {}
// the MFMA tile helpers of mlp_tiles.h are compiled, never run, here
typedef float shim_f4 __attribute__((ext_vector_type(4)));
inline shim_f4 __builtin_amdgcn_mfma_f32_16x16x4f32(float, float, shim_f4 c, int, int, int) { abort(); return c; }
template <class T> inline T __shfl_down(T v, int, int = 64) { abort(); return v; }
