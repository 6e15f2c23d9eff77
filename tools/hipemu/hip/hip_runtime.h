// Host emulation of the small part of HIP that apex_amd/csrc/ppo_small.hip uses, so that the kernel SOURCE can be compiled with the host clang++ and run on the CPU
// (tools/hipemu/emul_ppo_small.cpp, tests/test_kernel_emulation.py).  Test infrastructure only: nothing in the product includes this file.
//
// Execution model: a workgroup is a PROCESS (grid.x > 1: the launch forks grid.x - 1 children; every buffer the kernel touches must then live in MAP_SHARED memory,
// atomics and fences on it work across processes, and `static` = __shared__ is per process = per workgroup).  Each wave is an OS thread
// that runs its 64 lanes as ucontext fibers in round-robin order.  A wave collective (v_mfma_f32_16x16x4_f32, __shfl_*) deposits the lane's operands in one of two
// per-wave buffers and yields; when the lane is resumed every lane of the wave has deposited (collectives sit in wave-uniform control flow, as the hardware requires),
// and the lane computes its own part of the result - lane-exact operand / accumulator layout of the 16 x 16 x 4 MFMA.  __syncthreads is a pthread barrier between the
// wave threads, entered by the wave's scheduler once all of its lanes have asked for it.  The grid barrier of the kernel is its own code (atomics on shared memory).
#pragma once
#include <ucontext.h>
#include <pthread.h>
#include <sys/wait.h>
#include <unistd.h>
#include <sched.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <functional>
#include <map>
#include <string>
#include <sys/mman.h>
#include <thread>
#include <type_traits>
#include <vector>
using std::min;
using std::max;

struct dim3 { unsigned x, y, z; dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {} };
typedef void* hipStream_t;
typedef int hipError_t;
constexpr hipError_t hipSuccess = 0;
inline const char* hipGetErrorString(hipError_t) { return "emulated"; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memmove(d, s, n); return hipSuccess; }
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize };
inline hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return hipSuccess; }
#define HIP_KERNEL_NAME(...) __VA_ARGS__

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static
#define __launch_bounds__(...)
#define amdgpu_waves_per_eu(...) unused      /* __attribute__((amdgpu_waves_per_eu(a, b))): an occupancy hint, meaningless on the host */
#define __restrict__

namespace hipemu {
struct Wave;
struct Lane { ucontext_t ctx; dim3 tid; int lane; unsigned ncoll; bool done; Wave* wave; char* stack; };
constexpr size_t STACK_BYTES = 256 * 1024;
constexpr int MAX_WAVES = 16;
struct Wave {
    Lane lanes[64]; ucontext_t sched; int index;
    float A[2][64], B[2][64]; double Dv[2][64];
    int nlanes;
    bool want_barrier;
    pthread_barrier_t* block_barrier;
    const std::function<void()>* body;
};
inline thread_local Lane* g_cur = nullptr;
inline dim3 g_grid, g_block, g_bidx;
inline char g_dynsmem[160 * 1024] __attribute__((aligned(64)));      // the dynamic LDS segment of the running workgroup (`extern __shared__`, see build.sh)
inline char* g_stacks = nullptr;                                      // MAX_WAVES x 64 fiber stacks, mapped once (untouched pages cost nothing)
inline std::map<std::string, long> g_launches;      // launches per kernel expression as written at the launch site (tests ask which kernels a path really took)
inline int g_force_grid = 0;      // > 0: every launch runs with this many workgroups whatever the host code asked for
inline void yield() { Lane* l = g_cur; swapcontext(&l->ctx, &l->wave->sched); }
inline void fiber_entry(unsigned lo, unsigned hi) {
    Lane* l = (Lane*)(((uintptr_t)hi << 32) | (uintptr_t)lo);
    (*l->wave->body)();
    l->done = true;
    swapcontext(&l->ctx, &l->wave->sched);
}
inline void run_wave(Wave* W) {
    for (int i = 0; i < W->nlanes; ++i) {
        Lane& l = W->lanes[i];
        getcontext(&l.ctx);
        l.ctx.uc_stack.ss_sp = l.stack; l.ctx.uc_stack.ss_size = STACK_BYTES; l.ctx.uc_link = nullptr;
        const uintptr_t p = (uintptr_t)&l;
        makecontext(&l.ctx, (void (*)())fiber_entry, 2, (unsigned)(p & 0xffffffffu), (unsigned)(p >> 32));
    }
    for (;;) {
        int live = 0;
        for (int i = 0; i < W->nlanes; ++i) {
            Lane& l = W->lanes[i];
            if (l.done) continue;
            g_cur = &l;
            swapcontext(&W->sched, &l.ctx);
            live += !l.done;
        }
        if (W->want_barrier) { W->want_barrier = false; pthread_barrier_wait(W->block_barrier); }
        if (!live) break;
    }
}
inline void run_block(dim3 block, const std::function<void()>& body) {
    const int nthreads = (int)(block.x * block.y * block.z), nw = (nthreads + 63) / 64;
    if (nw > MAX_WAVES) { fprintf(stderr, "hipemu: workgroup of %d threads\n", nthreads); abort(); }
    if (!g_stacks) g_stacks = (char*)mmap(nullptr, (size_t)MAX_WAVES * 64 * STACK_BYTES, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    pthread_barrier_t bar; pthread_barrier_init(&bar, nullptr, nw);
    static Wave* waves[MAX_WAVES];
    for (int w = 0; w < nw; ++w) {
        if (!waves[w]) waves[w] = new Wave();
        Wave* W = waves[w]; W->index = w; W->want_barrier = false; W->block_barrier = &bar; W->body = &body;
        W->nlanes = std::min(64, nthreads - 64 * w);
        for (int i = 0; i < W->nlanes; ++i) {
            Lane& l = W->lanes[i]; const unsigned t = w * 64 + i;
            l.tid = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y)); l.lane = i; l.ncoll = 0; l.done = false; l.wave = W;
            l.stack = g_stacks + (size_t)(w * 64 + i) * STACK_BYTES;
        }
    }
    if (nw == 1) run_wave(waves[0]);
    else {
        std::vector<std::thread> th;
        for (int w = 0; w < nw; ++w) th.emplace_back(run_wave, waves[w]);
        for (auto& t : th) t.join();
    }
    pthread_barrier_destroy(&bar);
}
// Default: the workgroups of a launch run one after the other in this process (kernels whose workgroups do not wait for each other).  With g_force_grid > 0 a launch
// becomes that many CONCURRENT workgroups, one forked process each (a kernel with a grid barrier; its buffers must be MAP_SHARED).
inline void launch(dim3 grid, dim3 block, const std::function<void()>& body) {
    g_block = block;
    if (g_force_grid <= 0) {
        g_grid = grid;
        for (unsigned z = 0; z < grid.z; ++z)
            for (unsigned y = 0; y < grid.y; ++y)
                for (unsigned x = 0; x < grid.x; ++x) { g_bidx = dim3(x, y, z); run_block(block, body); }
        return;
    }
    grid = dim3(g_force_grid);
    if (grid.x > 16) { fprintf(stderr, "hipemu: at most 16 concurrent workgroups (%u)\n", grid.x); abort(); }
    g_grid = grid; g_bidx = dim3(0);
    std::vector<pid_t> kids;
    for (unsigned b = 1; b < grid.x; ++b) {
        const pid_t pid = fork();
        if (pid < 0) { perror("hipemu: fork"); abort(); }
        if (pid == 0) { g_bidx = dim3(b); run_block(block, body); _exit(0); }
        kids.push_back(pid);
    }
    run_block(block, body);
    for (pid_t pid : kids) {
        int st = 0; waitpid(pid, &st, 0);
        if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) { fprintf(stderr, "hipemu: a workgroup process died (status %d)\n", st); abort(); }
    }
}
}  // namespace hipemu

#define threadIdx (hipemu::g_cur->tid)
#define blockIdx (hipemu::g_bidx)
#define gridDim (hipemu::g_grid)
#define blockDim (hipemu::g_block)
#define hipLaunchKernelGGL(kern, grid, block, shmem, stream, ...) (hipemu::g_launches[#kern] += 1, hipemu::launch(grid, block, [=]() { kern(__VA_ARGS__); }))

// the residency queries and the cooperative launch of tiles::launch_resident (mlp_tiles.h): the emulated device holds what the test asks for (g_force_grid workgroups)
enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount, hipDeviceAttributeCooperativeLaunch };
inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
inline hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t a, int) { *v = a == hipDeviceAttributeMultiprocessorCount ? 256 : 1; return hipSuccess; }
template <class F> inline hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int* n, F, int, size_t) { *n = 1; return hipSuccess; }
template <class A> inline hipError_t hipLaunchCooperativeKernel(void (*f)(A), dim3 grid, dim3 block, void** params, unsigned, hipStream_t) {
    const A a = *(const A*)params[0];
    hipemu::g_launches["cooperative"] += 1;
    hipemu::launch(grid, block, [=]() { f(a); });
    return hipSuccess;
}

template <class T> inline T hipemu_atomic_add(T* p, T v) {
    T old, nw;
    __atomic_load(p, &old, __ATOMIC_SEQ_CST);
    do { nw = old + v; } while (!__atomic_compare_exchange(p, &old, &nw, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST));
    return old;
}
inline float atomicAdd(float* p, float v) { return hipemu_atomic_add(p, v); }
inline double atomicAdd(double* p, double v) { return hipemu_atomic_add(p, v); }
inline int atomicAdd(int* p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
inline unsigned atomicAdd(unsigned* p, unsigned v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }

inline void __syncthreads() { hipemu::g_cur->wave->want_barrier = true; hipemu::yield(); }
inline void __threadfence() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
#define __HIP_MEMORY_SCOPE_AGENT 0
template <class T> inline T hipemu_atomic_load(const T* p) { T v; __atomic_load(p, &v, __ATOMIC_SEQ_CST); return v; }
template <class T> inline void hipemu_atomic_store(T* p, T v) { __atomic_store(p, &v, __ATOMIC_SEQ_CST); }
#define __hip_atomic_load(p, order, scope) hipemu_atomic_load(p)
#define __hip_atomic_store(p, v, order, scope) hipemu_atomic_store(p, v)
#define __hip_atomic_fetch_add(p, v, order, scope) __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST)
inline void __builtin_amdgcn_s_sleep(int) { sched_yield(); }
inline void __builtin_amdgcn_sched_barrier(int) {}

typedef float hipemu_f4 __attribute__((ext_vector_type(4)));
// D = A B + C with A 16 x 4 (lane 16 k + i holds A[i][k]), B 4 x 16 (lane 16 k + n holds B[k][n]), C / D 16 x 16 (lane 16 g + n, component v: row 4 g + v, column n)
inline hipemu_f4 __builtin_amdgcn_mfma_f32_16x16x4f32(float a, float b, hipemu_f4 c, int, int, int) {
    hipemu::Lane* l = hipemu::g_cur; hipemu::Wave* W = l->wave;
    const int buf = l->ncoll++ & 1, ln = l->lane;
    W->A[buf][ln] = a; W->B[buf][ln] = b;
    hipemu::yield();
    const int n = ln & 15, g = ln >> 4;
    for (int v = 0; v < 4; ++v) {
        const int i = 4 * g + v;
        float s = c[v];
        for (int k = 0; k < 4; ++k) s = fmaf(W->A[buf][16 * k + i], W->B[buf][16 * k + n], s);
        c[v] = s;
    }
    return c;
}
typedef float hipemu_f16v __attribute__((ext_vector_type(16)));
// D = A B + C with A 32 x 2 (lane 32 k + i holds A[i][k]), B 2 x 32 (lane 32 k + n holds B[k][n]), C / D 32 x 32 (lane 32 h + n, component r: row (r & 3) + 8 (r >> 2) + 4 h)
inline hipemu_f16v __builtin_amdgcn_mfma_f32_32x32x2f32(float a, float b, hipemu_f16v c, int, int, int) {
    hipemu::Lane* l = hipemu::g_cur; hipemu::Wave* W = l->wave;
    const int buf = l->ncoll++ & 1, ln = l->lane;
    W->A[buf][ln] = a; W->B[buf][ln] = b;
    hipemu::yield();
    const int n = ln & 31, h = ln >> 5;
    for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * h;
        float s = c[r];
        for (int k = 0; k < 2; ++k) s = fmaf(W->A[buf][32 * k + i], W->B[buf][32 * k + n], s);
        c[r] = s;
    }
    return c;
}
template <class T> inline T hipemu_exchange(T v, int src_of_lane_fn(int, int), int o) {
    hipemu::Lane* l = hipemu::g_cur; hipemu::Wave* W = l->wave;
    const int buf = l->ncoll++ & 1, ln = l->lane;
    W->Dv[buf][ln] = (double)v;      // (float -> double -> float is exact)
    hipemu::yield();
    const int src = src_of_lane_fn(ln, o);
    return src >= 0 && src < 64 ? (T)W->Dv[buf][src] : v;
}
inline int hipemu_src_down(int ln, int o) { return ln + o; }
inline int hipemu_src_xor(int ln, int o) { return ln ^ o; }
template <class T> inline T __shfl_down(T v, int o, int width = 64) { return hipemu_exchange(v, hipemu_src_down, o); }
template <class T> inline T __shfl_xor(T v, int o, int width = 64) { return hipemu_exchange(v, hipemu_src_xor, o); }
