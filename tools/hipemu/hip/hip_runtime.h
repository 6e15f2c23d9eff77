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
#include <functional>
#include <thread>
#include <vector>

struct dim3 { unsigned x, y, z; dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {} };
typedef void* hipStream_t;
typedef int hipError_t;
constexpr hipError_t hipSuccess = 0;
inline const char* hipGetErrorString(hipError_t) { return "emulated"; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static
#define __launch_bounds__(...)
#define __restrict__

namespace hipemu {
struct Wave;
struct Lane { ucontext_t ctx; dim3 tid; int lane; unsigned ncoll; bool done; Wave* wave; std::vector<char> stack; };
struct Wave {
    Lane lanes[64]; ucontext_t sched; int index;
    float A[2][64], B[2][64]; double Dv[2][64];
    bool want_barrier;
    pthread_barrier_t* block_barrier;
    const std::function<void()>* body;
};
inline thread_local Lane* g_cur = nullptr;
inline dim3 g_grid, g_block, g_bidx;
inline int g_force_grid = 0;      // > 0: every launch runs with this many workgroups whatever the host code asked for
inline void yield() { Lane* l = g_cur; swapcontext(&l->ctx, &l->wave->sched); }
inline void fiber_entry(unsigned lo, unsigned hi) {
    Lane* l = (Lane*)(((uintptr_t)hi << 32) | (uintptr_t)lo);
    (*l->wave->body)();
    l->done = true;
    swapcontext(&l->ctx, &l->wave->sched);
}
inline void run_wave(Wave* W) {
    for (int i = 0; i < 64; ++i) {
        Lane& l = W->lanes[i];
        l.stack.resize(256 * 1024);
        getcontext(&l.ctx);
        l.ctx.uc_stack.ss_sp = l.stack.data(); l.ctx.uc_stack.ss_size = l.stack.size(); l.ctx.uc_link = nullptr;
        const uintptr_t p = (uintptr_t)&l;
        makecontext(&l.ctx, (void (*)())fiber_entry, 2, (unsigned)(p & 0xffffffffu), (unsigned)(p >> 32));
    }
    for (;;) {
        int live = 0;
        for (int i = 0; i < 64; ++i) {
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
    const int nw = block.x / 64;
    pthread_barrier_t bar; pthread_barrier_init(&bar, nullptr, nw);
    std::vector<Wave*> waves;
    for (int w = 0; w < nw; ++w) {
        Wave* W = new Wave(); W->index = w; W->want_barrier = false; W->block_barrier = &bar; W->body = &body;
        for (int i = 0; i < 64; ++i) { Lane& l = W->lanes[i]; l.tid = dim3(w * 64 + i); l.lane = i; l.ncoll = 0; l.done = false; l.wave = W; }
        waves.push_back(W);
    }
    std::vector<std::thread> th;
    for (Wave* W : waves) th.emplace_back(run_wave, W);
    for (auto& t : th) t.join();
    for (Wave* W : waves) delete W;
    pthread_barrier_destroy(&bar);
}
inline void launch(dim3 grid, dim3 block, const std::function<void()>& body) {
    if (g_force_grid > 0) grid = dim3(g_force_grid);
    if (grid.x < 1 || grid.x > 16 || block.x % 64 != 0) { fprintf(stderr, "hipemu: 1..16 workgroups of whole waves (grid %u, block %u)\n", grid.x, block.x); abort(); }
    if (g_force_grid > 0) grid = dim3(g_force_grid);
    g_grid = grid; g_block = block; g_bidx = dim3(0);
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
#define hipLaunchKernelGGL(kern, grid, block, shmem, stream, ...) hipemu::launch(grid, block, [=]() { kern(__VA_ARGS__); })

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
