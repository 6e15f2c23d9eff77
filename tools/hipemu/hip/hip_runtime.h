// Host emulation of the part of HIP that the kernels of apex_amd/csrc use, so that the kernel SOURCES can be compiled with the host clang++ and run on the CPU: the learner side
// (learner.hip, ppo_small.hip, td3_small.hip: emul_learner.cpp, emul_ppo_small.cpp, emul_td3_small.cpp) and the env side (env.hip with cassie_lane.h, cassie_complete.h,
// estimator_lane.h: emul_env.cpp; its inline-assembly dialect is restated in gfx950/lane_ops.h next to this directory).  Test infrastructure only: nothing in the product
// includes this file.
//
// Execution model.  A wave = 64 fibers (a six-register switch; ucontext for the sanitizer builds) that run round-robin from collective to collective.
//  * Workgroups of a PLAIN launch run one after the other in the calling process; single-wave workgroups with dynamic LDS (the env kernels) are dealt over host threads.  A
//    COOPERATIVE launch (a kernel with a grid barrier) runs apx_emul_set_workgroups concurrent workgroups, one forked process each (buffers in MAP_SHARED memory; `static` =
//    __shared__ is per process = per workgroup).  The waves of a multi-wave workgroup are OS threads; __syncthreads is a pthread barrier between them.
//  * MFMA and __shfl deposit the lane's operands in one of two per-wave buffers and yield; resumed, every lane has deposited (these kernels keep their waves in step) and the
//    lane computes its part: lane-exact operand / accumulator layouts.
//  * The 32-bit lane exchanges of the env kernels (DPP operands, ds_bpermute, v_readlane, ballots, wsync) carry a sequence number and their call site: a lane reads its source
//    lane's deposit of the SAME sequence number, waits for a lane that is late, reads zero from a lane that sits the branch out (parked at a reconvergence point) or has left,
//    and aborts when the call sites differ.  hipemu::converge() = APX_CONVERGE(): reconvergence of rows that took different paths.
//  * HIPEMU_LOCKSTEP_CHECK (build.sh lockstep): every load / store of the translation unit is traced; pairs of accesses of two lanes whose order the hardware fixes by
//    executing in lockstep, with no rendezvous between them here, are reported (APX_LOCKSTEP() in the kernel source makes the rendezvous).
//  * Knobs: HIPEMU_THREADS, HIPEMU_LDS_FILL / HIPEMU_STACK_FILL (what a workgroup finds in LDS / a lane in its "registers": results must not depend on it).
#pragma once
#include <dlfcn.h>
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
#include <atomic>
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
template <class F> inline hipError_t hipFuncSetAttribute(F, hipFuncAttribute, int) { return hipSuccess; }
#define HIP_KERNEL_NAME(...) __VA_ARGS__
// the host runtime as env.hip uses it: "device" memory is host memory (filled with a byte pattern: hipMalloc does not clear either), every launch completes inside the call,
// so streams and events have nothing to order
typedef void* hipEvent_t;
enum hipStreamCaptureStatus { hipStreamCaptureStatusNone, hipStreamCaptureStatusActive };
constexpr unsigned hipStreamNonBlocking = 1, hipEventDisableTiming = 2;
inline hipError_t hipSetDevice(int) { return hipSuccess; }
template <class T> inline hipError_t hipMalloc(T** p, size_t n) { *p = (T*)malloc(n ? n : 1); if (*p) memset((void*)*p, 0xA5, n); return *p ? hipSuccess : 2; }
inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
inline hipError_t hipMemset(void* p, int v, size_t n) { memset(p, v, n); return hipSuccess; }
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memmove(d, s, n); return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = nullptr; return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamIsCapturing(hipStream_t, hipStreamCaptureStatus* st) { *st = hipStreamCaptureStatusNone; return hipSuccess; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
inline hipError_t hipEventCreate(hipEvent_t* e) { *e = nullptr; return hipSuccess; }
inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = nullptr; return hipSuccess; }
inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 1.f; return hipSuccess; }      // (a fixed millisecond: callers divide by it)
#define HIP_SYMBOL(x) (&(x))
template <class S> inline hipError_t hipMemcpyToSymbol(S* sym, const void* src, size_t n) { memcpy((void*)sym, src, n); return hipSuccess; }
template <class S> inline hipError_t hipMemcpyFromSymbol(void* dst, S* sym, size_t n) { memcpy(dst, (const void*)sym, n); return hipSuccess; }

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
// Fiber switch: six callee-saved registers and the stack pointer (a glibc swapcontext is a sigprocmask system call per switch - 0.3 us against ~5 ns - and the env
// kernels switch 64 lanes at every DPP operand).  HIPEMU_UCONTEXT (the sanitizer builds of asan.sh, which know ucontext) keeps the portable form.
#ifndef HIPEMU_UCONTEXT
struct Ctx { void* sp; };
__attribute__((naked, noinline)) inline void ctx_switch(void** /*save_sp: rdi*/, void* /*load_sp: rsi*/) {
    asm volatile("pushq %rbp\n\tpushq %rbx\n\tpushq %r12\n\tpushq %r13\n\tpushq %r14\n\tpushq %r15\n\t"
                 "movq %rsp, (%rdi)\n\tmovq %rsi, %rsp\n\t"
                 "popq %r15\n\tpopq %r14\n\tpopq %r13\n\tpopq %r12\n\tpopq %rbx\n\tpopq %rbp\n\tret");
}
#else
struct Ctx { ucontext_t uc; };
#endif
struct Lane { Ctx ctx; dim3 tid; int lane; unsigned ncoll; bool done; Wave* wave; char* stack; const void* conv_site; unsigned conv_seq; };
#if defined(HIPEMU_STACK_MB)
constexpr size_t STACK_BYTES = (size_t)HIPEMU_STACK_MB * 1024 * 1024;
#elif defined(HIPEMU_UCONTEXT)
constexpr size_t STACK_BYTES = 16 * 1024 * 1024;      // the sanitizer builds (-O1, redzones around every local of the env kernels' inlined stages)
#else
constexpr size_t STACK_BYTES = 1024 * 1024;
#endif
constexpr int MAX_WAVES = 16;
struct Wave {
    Lane lanes[64]; Ctx sched; int index;
    float A[2][64], B[2][64]; double Dv[2][64];
    unsigned conv_counter;      // arrival order at reconvergence points (hipemu::converge)
    uint32_t X[8][64], Y[8][64]; const void* site[8][64]; unsigned seq[8][64];      // 8 deposit slots per lane: a lane may run up to 7 exchanges ahead of a reader of its deposit      // 32-bit lane exchanges (DPP, bpermute, readlane, ballot) and the call site each lane made its from
    int nlanes;
    bool want_barrier;
    pthread_barrier_t* block_barrier;
    const std::function<void()>* body;
};
inline thread_local Lane* g_cur = nullptr;
inline dim3 g_grid, g_block;
inline thread_local dim3 g_bidx;
// the dynamic LDS segment of the running workgroup (`extern __shared__`, see build.sh) and its fiber stacks: per host thread, because the single-wave workgroups of a launch
// that do not wait for each other (the env kernels) are dealt over host threads (HIPEMU_THREADS, default: the cores)
constexpr size_t DYNSMEM_BYTES = 160 * 1024;
inline thread_local char* g_dynsmem = nullptr;      // (allocated by the thread that runs the workgroup; the wave threads of a multi-wave workgroup are handed the same block)
// fiber stacks of the workgroups a host thread runs: mapped on first use for as many waves as that workgroup has, unmapped when the thread ends (the threads of a
// threaded launch live for one launch: left mapped, a test session piles up terabytes of address space and every later fork() of a workgroup process crawls)
struct StackArena {
    char* base = nullptr; size_t bytes = 0;
    ~StackArena() { if (base) munmap(base, bytes); }
};
inline thread_local StackArena g_stack_arena;
struct LdsArena { char* p = nullptr; ~LdsArena() { free(p); } };
inline thread_local LdsArena g_lds_arena;
inline std::map<std::string, long> g_launches;      // launches per kernel expression as written at the launch site (tests ask which kernels a path really took)
inline uint64_t g_block_gen = 0;      // workgroups run so far (the lockstep checker's epochs)
inline int g_force_grid = 0;      // > 0: a cooperative launch runs with this many concurrent workgroups (default 1) whatever the host code asked for
#ifndef HIPEMU_UCONTEXT
inline void lockstep_flush();
inline void yield() { lockstep_flush(); Lane* l = g_cur; ctx_switch(&l->ctx.sp, l->wave->sched.sp); }
inline void fiber_main() {
    Lane* l = g_cur;
    (*l->wave->body)();
    lockstep_flush();
    l->done = true;
    ctx_switch(&l->ctx.sp, l->wave->sched.sp);
    abort();
}
inline void fiber_init(Lane& l) {      // a stack on which ctx_switch's six pops and its ret land in fiber_main with the ABI's alignment (rsp = 8 mod 16 at entry)
    uintptr_t top = ((uintptr_t)l.stack + STACK_BYTES) & ~(uintptr_t)15;
    void** sp = (void**)(top - 8);
    *--sp = (void*)&fiber_main;
    for (int i = 0; i < 6; ++i) *--sp = nullptr;
    l.ctx.sp = sp;
}
inline void resume(Wave* W, Lane& l) { ctx_switch(&W->sched.sp, l.ctx.sp); }
#else
inline void lockstep_flush();
inline void yield() { lockstep_flush(); Lane* l = g_cur; swapcontext(&l->ctx.uc, &l->wave->sched.uc); }
inline void fiber_entry(unsigned lo, unsigned hi) {
    Lane* l = (Lane*)(((uintptr_t)hi << 32) | (uintptr_t)lo);
    (*l->wave->body)();
    l->done = true;
    swapcontext(&l->ctx.uc, &l->wave->sched.uc);
}
inline void fiber_init(Lane& l) {
    getcontext(&l.ctx.uc);
    l.ctx.uc.uc_stack.ss_sp = l.stack; l.ctx.uc.uc_stack.ss_size = STACK_BYTES; l.ctx.uc.uc_link = nullptr;
    const uintptr_t p = (uintptr_t)&l;
    makecontext(&l.ctx.uc, (void (*)())fiber_entry, 2, (unsigned)(p & 0xffffffffu), (unsigned)(p >> 32));
}
inline void resume(Wave* W, Lane& l) { swapcontext(&W->sched.uc, &l.ctx.uc); }
#endif
inline void run_wave(Wave* W) {
    for (int i = 0; i < W->nlanes; ++i) fiber_init(W->lanes[i]);
    for (;;) {
        int live = 0;
        for (int i = 0; i < W->nlanes; ++i) {
            Lane& l = W->lanes[i];
            if (l.done) continue;
            g_cur = &l;
            resume(W, l);
            live += !l.done;
        }
        if (W->want_barrier) { W->want_barrier = false; pthread_barrier_wait(W->block_barrier); }
        if (!live) break;
    }
    g_cur = nullptr;      // host code runs outside any lane
}
inline void run_block(dim3 block, const std::function<void()>& body) {
    const int nthreads = (int)(block.x * block.y * block.z), nw = (nthreads + 63) / 64;
    if (nw > MAX_WAVES) { fprintf(stderr, "hipemu: workgroup of %d threads\n", nthreads); abort(); }
    if (!g_lds_arena.p) g_lds_arena.p = (char*)aligned_alloc(64, DYNSMEM_BYTES);
    g_dynsmem = g_lds_arena.p;
    StackArena& sa = g_stack_arena;
    const size_t want = (size_t)nw * 64 * STACK_BYTES;
    if (sa.bytes < want) {
        if (sa.base) munmap(sa.base, sa.bytes);
        sa.base = (char*)mmap(nullptr, want, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0); sa.bytes = want;
    }
    char* const g_stacks = sa.base;
    pthread_barrier_t bar; pthread_barrier_init(&bar, nullptr, nw);
    static thread_local Wave* waves[MAX_WAVES];
    for (int w = 0; w < nw; ++w) {
        if (!waves[w]) waves[w] = new Wave();
        Wave* W = waves[w]; W->index = w; W->want_barrier = false; W->conv_counter = 0; memset(W->seq, 0xff, sizeof W->seq); W->block_barrier = &bar; W->body = &body;
        W->nlanes = std::min(64, nthreads - 64 * w);
        for (int i = 0; i < W->nlanes; ++i) {
            Lane& l = W->lanes[i]; const unsigned t = w * 64 + i;
            l.tid = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y)); l.lane = i; l.ncoll = 0; l.done = false; l.wave = W; l.conv_site = nullptr; l.conv_seq = 0;
            l.stack = g_stacks + (size_t)(w * 64 + i) * STACK_BYTES;
        }
    }
#ifdef HIPEMU_LOCKSTEP_CHECK
    ++g_block_gen;
#endif
    static const int lds_fill = getenv("HIPEMU_LDS_FILL") ? (int)strtol(getenv("HIPEMU_LDS_FILL"), nullptr, 0) : -1;      // what a workgroup finds in its dynamic LDS: by default what the one before it left (as on the hardware: anything); 0x00 / 0xff ...: that byte
    if (lds_fill >= 0) memset(g_dynsmem, lds_fill, DYNSMEM_BYTES);
    static const int stack_fill = getenv("HIPEMU_STACK_FILL") ? (int)strtol(getenv("HIPEMU_STACK_FILL"), nullptr, 0) : -1;      // the lanes' "registers" before the kernel body runs (top 96 KB of each fiber stack): a result that depends on this byte reads a variable it never set
    if (stack_fill >= 0) for (int w = 0; w < nw; ++w) for (int i = 0; i < waves[w]->nlanes; ++i) memset(waves[w]->lanes[i].stack + STACK_BYTES - 96 * 1024, stack_fill, 96 * 1024);
    if (nw == 1) run_wave(waves[0]);
    else {
        std::vector<std::thread> th;
        char* const lds = g_dynsmem; const dim3 bidx = g_bidx;      // (thread-local in the thread that runs the workgroup: handed to its wave threads)
        for (int w = 0; w < nw; ++w) th.emplace_back([lds, bidx](Wave* W) { g_dynsmem = lds; g_bidx = bidx; run_wave(W); }, waves[w]);
        for (auto& t : th) t.join();
    }
    pthread_barrier_destroy(&bar);
}
// A plain launch: its workgroups run one after the other in this process (or, single-wave workgroups with dynamic LDS, on host threads).  A COOPERATIVE launch (a kernel with
// a grid barrier: hipLaunchCooperativeKernel) becomes g_force_grid CONCURRENT workgroups (default 1), one forked process each; with more than one its buffers must be MAP_SHARED.
inline void launch(dim3 grid, dim3 block, const std::function<void()>& body, size_t dyn_lds = 0, bool cooperative = false) {
    g_block = block;
    if (!cooperative) {      // (a plain launch: its workgroups do not wait for each other)
        g_grid = grid;
#ifndef HIPEMU_LOCKSTEP_CHECK
        static const int host_threads = getenv("HIPEMU_THREADS") ? atoi(getenv("HIPEMU_THREADS")) : (int)std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
        const unsigned total = grid.x * grid.y * grid.z;
        // single-wave workgroups that ask for DYNAMIC LDS (the env kernels; `__shared__` arrays are process-wide statics here: a kernel that declares one keeps to one thread)
        if (host_threads > 1 && total >= 4 && block.x * block.y * block.z <= 64 && dyn_lds > 0) {
            std::atomic<unsigned> next{0};
            std::vector<std::thread> th;
            for (int t = 0; t < std::min<int>(host_threads, (int)total); ++t)
                th.emplace_back([&]() {
                    for (unsigned b = next++; b < total; b = next++) { g_bidx = dim3(b % grid.x, (b / grid.x) % grid.y, b / (grid.x * grid.y)); run_block(block, body); }
                });
            for (auto& t : th) t.join();
            return;
        }
#endif
        for (unsigned z = 0; z < grid.z; ++z)
            for (unsigned y = 0; y < grid.y; ++y)
                for (unsigned x = 0; x < grid.x; ++x) { g_bidx = dim3(x, y, z); run_block(block, body); }
        return;
    }
    grid = dim3(g_force_grid > 0 ? g_force_grid : 1);      // a cooperative launch (grid barrier inside): apx_emul_set_workgroups concurrent workgroups, one by default
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
#define hipLaunchKernelGGL(kern, grid, block, shmem, stream, ...) (hipemu::g_launches[#kern] += 1, hipemu::launch(grid, block, [=]() { kern(__VA_ARGS__); }, (size_t)(shmem)))

// the residency queries and the cooperative launch of tiles::launch_resident (mlp_tiles.h): the emulated device holds what the test asks for (g_force_grid workgroups)
enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount, hipDeviceAttributeCooperativeLaunch };
inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
inline hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t a, int) { *v = a == hipDeviceAttributeMultiprocessorCount ? 256 : 1; return hipSuccess; }
template <class F> inline hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int* n, F, int, size_t) { *n = 1; return hipSuccess; }
template <class A> inline hipError_t hipLaunchCooperativeKernel(void (*f)(A), dim3 grid, dim3 block, void** params, unsigned, hipStream_t) {
    const A a = *(const A*)params[0];
    hipemu::g_launches["cooperative"] += 1;
    hipemu::launch(grid, block, [=]() { f(a); }, 0, true);
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
inline int atomicCAS(int* p, int expect, int desired) { __atomic_compare_exchange_n(p, &expect, desired, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST); return expect; }

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


// ------------------------------------------------------------------------------------------------ 32-bit lane exchanges (the env kernels: DPP, ds_bpermute, readlane, ballot)
// Every lane deposits its word (and the address of the call site) and yields; resumed, it reads the word of its source lane.  A collective sits in wave-uniform control
// flow: a source lane that deposited from ANOTHER call site means the lanes of the wave have diverged around a collective, and the run aborts saying so.  A lane that has
// left the kernel is an inactive lane: its word reads as zero (DPP with bound_ctrl, the way the kernels use it) and it is absent from ballots.
namespace hipemu {
inline uintptr_t rel(const void* p) { Dl_info i; return dladdr(p, &i) && i.dli_fbase ? (uintptr_t)p - (uintptr_t)i.dli_fbase : (uintptr_t)p; }      // offset inside the library: llvm-symbolizer -e lib --inlines +0x..
struct Xchg { Wave* W; int buf, lane; const void* site; unsigned seq; };
#define HIPEMU_NOCOV __attribute__((no_sanitize("coverage")))
HIPEMU_NOCOV __attribute__((noinline)) inline Xchg exchange2(uint32_t x, uint32_t y) {
    Lane* l = g_cur; Wave* W = l->wave;
    const unsigned seq = l->ncoll++;
    const int buf = seq & 7, ln = l->lane;
    const void* site = __builtin_return_address(0);
    W->X[buf][ln] = x; W->Y[buf][ln] = y; W->site[buf][ln] = site; W->seq[buf][ln] = seq;
    yield();
    return Xchg{W, buf, ln, site, seq};
}
HIPEMU_NOCOV inline uint32_t peek(const Xchg& e, int src, bool second = false) {
    if (src < 0 || src >= e.W->nlanes) return 0u;      // no such lane
    for (long spins = 0;; ++spins) {
        if (e.W->seq[e.buf][src] == e.seq) {            // the source lane's deposit for THIS exchange (wherever that lane is by now)
            if (e.W->site[e.buf][src] != e.site) {
                fprintf(stderr, "hipemu: lane %d reads lane %d across DIVERGED control flow (collective call sites +0x%lx vs +0x%lx)\n", e.lane, src, (unsigned long)rel(e.site), (unsigned long)rel(e.W->site[e.buf][src]));
                abort();
            }
            return second ? e.W->Y[e.buf][src] : e.W->X[e.buf][src];
        }
        const Lane& o = e.W->lanes[src];
        if (o.done || o.conv_site) return 0u;           // it has left the kernel / sits this branch out at a reconvergence point: an inactive lane (zero operand, absent from ballots)
        if (spins > 4000000) {
            fprintf(stderr, "hipemu: lane %d waits for lane %d at site +0x%lx (its collective %u, the other lane's %u): rows that took different paths meet here without a reconvergence point\n",
                    e.lane, src, (unsigned long)rel(e.site), e.seq, o.ncoll);
            abort();
        }
        yield();                                        // the source lane has not reached this exchange yet (lanes fall out of step behind reconvergence points)
    }
}
// Reconvergence point (APX_CONVERGE() in the kernel source: where the lanes that entered a branch taken by only some ROWS of the wave - a branch that holds
// collectives - are together again; the exec mask's business on the hardware).  A lane parks here.  When every live lane of the wave is parked somewhere, the group that
// parked LAST is released: rows that sat a branch out park behind it at once, the rows inside arrive later - at an inner point first, whose group is then the
// latest, and in the end at the outer point.  A parked lane reads as an inactive lane (zero operand, absent from ballots).
HIPEMU_NOCOV __attribute__((noinline)) inline void converge() {
    Lane* l = g_cur; Wave* W = l->wave;
    l->conv_site = __builtin_return_address(0); l->conv_seq = ++W->conv_counter;
    while (l->conv_site) {                      // (cleared by whichever lane releases this lane's group)
        int parked = 0, live = 0; unsigned mx = 0, last_seq = 0; const void* last_site = nullptr;
        for (int i = 0; i < W->nlanes; ++i) {
            const Lane& o = W->lanes[i];
            if (o.done) continue;
            ++live; mx = std::max(mx, o.ncoll);
            if (o.conv_site) { ++parked; if (o.conv_seq > last_seq) { last_seq = o.conv_seq; last_site = o.conv_site; } }
        }
        if (parked == live) {
            const unsigned nc = (mx + 2) & ~1u;      // the released lanes count their collectives from a common (even) number again
            for (int i = 0; i < W->nlanes; ++i) { Lane& o = W->lanes[i]; if (!o.done && o.conv_site == last_site) { o.conv_site = nullptr; o.ncoll = nc; } }
            continue;
        }
        yield();
    }
}
inline float asf(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
inline uint32_t asu(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
// source lane of a DPP control word (gfx9 encoding), -1 = no source (a shift beyond the row end)
inline int dpp_src(int ctrl, int ln) {
    const int row = ln & ~15, r = ln & 15;
    if (ctrl < 0x100) return (ln & ~3) | ((ctrl >> (2 * (ln & 3))) & 3);                       // quad_perm
    if (ctrl >= 0x101 && ctrl <= 0x10F) { const int n = ctrl - 0x100; return r + n <= 15 ? ln + n : -1; }      // row_shl: lane i takes lane i + n
    if (ctrl >= 0x111 && ctrl <= 0x11F) { const int n = ctrl - 0x110; return r >= n ? ln - n : -1; }           // row_shr: lane i takes lane i - n
    if (ctrl >= 0x121 && ctrl <= 0x12F) { const int n = ctrl - 0x120; return row | ((r - n) & 15); }           // row_ror
    if (ctrl == 0x140) return row | (15 - r);                                                   // row_mirror
    if (ctrl == 0x141) return (ln & ~7) | (7 - (ln & 7));                                       // row_half_mirror
    if (ctrl >= 0x150 && ctrl <= 0x15F) return row | (ctrl - 0x150);                            // row_newbcast
    fprintf(stderr, "hipemu: DPP control 0x%x not emulated\n", ctrl); abort();
}
// the DPP operand of lane ln: value of its source lane, 0 without one (bound_ctrl:1 / old = 0)
inline float dpp_get(const Xchg& e, int ctrl, bool second = false) { return asf(peek(e, dpp_src(ctrl, e.lane), second)); }
}  // namespace hipemu

__attribute__((always_inline)) inline int __builtin_amdgcn_update_dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl) {
    if (old != 0 || row_mask != 0xF || bank_mask != 0xF || !bound_ctrl) { fprintf(stderr, "hipemu: update_dpp form not emulated\n"); abort(); }
    const hipemu::Xchg e = hipemu::exchange2((uint32_t)src, 0u);
    return (int)hipemu::peek(e, hipemu::dpp_src(ctrl, e.lane));
}
__attribute__((always_inline)) inline int __builtin_amdgcn_ds_bpermute(int addr, int v) {
    const hipemu::Xchg e = hipemu::exchange2((uint32_t)v, 0u);
    return (int)hipemu::peek(e, (addr >> 2) & 63);
}
__attribute__((always_inline)) inline int __builtin_amdgcn_readlane(int v, int k) {
    const hipemu::Xchg e = hipemu::exchange2((uint32_t)v, 0u);
    return (int)hipemu::peek(e, k & 63);
}
__attribute__((always_inline)) HIPEMU_NOCOV inline unsigned long long __builtin_amdgcn_ballot_w64(bool p) {
    const hipemu::Xchg e = hipemu::exchange2(p ? 1u : 0u, 0u);
    unsigned long long m = 0;
    for (int i = 0; i < e.W->nlanes; ++i) m |= (unsigned long long)(hipemu::peek(e, i) & 1u) << i;
    return m;
}
// a wave is one instruction stream: LDS / memory written by a lane in front of this point is visible to every lane behind it.  Under the emulation the lanes are fibers
// that run from collective to collective, so the hand-off needs a rendezvous
__attribute__((always_inline)) inline void __builtin_amdgcn_wave_barrier() { hipemu::exchange2(0u, 0u); }
inline void __builtin_amdgcn_fence(int, const char*) {}
inline float __builtin_amdgcn_rcpf(float x) { return 1.0f / x; }
inline float __frcp_rn(float x) { return 1.0f / x; }
inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
inline void __sincosf(float x, float* s, float* c) { *s = sinf(x); *c = cosf(x); }      // (the device intrinsic is the fast hardware form; the host's is libm's)
inline int __float_as_int(float f) { return (int)hipemu::asu(f); }
inline unsigned __float_as_uint(float f) { return hipemu::asu(f); }
inline float __int_as_float(int i) { return hipemu::asf((uint32_t)i); }
inline float __uint_as_float(unsigned u) { return hipemu::asf(u); }
struct alignas(16) float4 { float x, y, z, w; };
inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
#define __constant__ static
#define __noinline__ __attribute__((noinline))

// ------------------------------------------------------------------------------------------------ lockstep checker (build.sh lockstep -> libapx_emul_lockstep.so)
// A wave is ONE instruction stream: a lane's load in front of another lane's store in program order sees the old value whether or not anything separates the two.  The
// emulation's lanes run from rendezvous to rendezvous (collectives, wsync) one after the other, so two accesses of DIFFERENT lanes to the same word - at least one a
// store - with no rendezvous between them come out in lane order instead of program order.  This build has every load and store of the translation unit call in here
// (-fsanitize-coverage=trace-loads,trace-stores) and reports each such pair once per store / load site; the kernel source then states the dependence with
// APX_LOCKSTEP() (gfx950/lane_ops.h: nothing on the hardware, a rendezvous here).  Single-wave workgroups only (the env kernels).
#ifndef HIPEMU_LOCKSTEP_CHECK
namespace hipemu { inline void lockstep_flush() {} }
#else
#include <dlfcn.h>
#include <set>
#include <unordered_map>
namespace hipemu {
struct Shadow { uint64_t wep = ~0ull, rep = ~0ull, rmask = 0; int wlane = -1; uint32_t wold = 0; const void* wsite = nullptr; const void* rsite = nullptr; };
inline std::unordered_map<uintptr_t, Shadow> g_shadow;
inline std::set<std::pair<const void*, const void*>> g_reported;
inline long g_conflicts = 0;
inline bool g_in_check = false;
// a store is reported when it has happened (the callback runs in front of it): a store of the value the word already held orders nothing
struct Pending { uintptr_t word; uint32_t old; int lane, other; const void* site; const void* osite; bool waw; };
inline std::vector<Pending> g_pending;
inline void conflict(const char* kind, uintptr_t word, int lane, int other, const void* site, const void* osite) {
    ++g_conflicts;
    if (!g_reported.insert({site, osite}).second) return;
    const bool lds = word * 4 >= (uintptr_t)g_dynsmem && word * 4 < (uintptr_t)g_dynsmem + DYNSMEM_BYTES;
    fprintf(stderr, "hipemu lockstep: %s on %s word %ld: lane %d at +0x%lx vs lane %d at +0x%lx, no rendezvous between them\n", kind, lds ? "LDS" : "memory",
            lds ? (long)(word - (uintptr_t)g_dynsmem / 4) : (long)word, lane, (unsigned long)rel(site), other, (unsigned long)rel(osite));
}
inline void flush_pending() {
    for (const Pending& q : g_pending)
        if (*(const volatile uint32_t*)(q.word * 4) != q.old)
            conflict(q.waw ? "two lanes store DIFFERENT values to one word (the later lane's stays here; on the hardware: program order, or unspecified within one instruction)"
                           : "store behind another lane's load (program order: the load comes first or the store - the emulation ran the load first)", q.word, q.lane, q.other, q.site, q.osite);
    g_pending.clear();
}
inline void access(const void* addr, int bytes, bool store, const void* site) {
    Lane* l = g_cur;
    if (!l || g_in_check || g_block.x * g_block.y * g_block.z > 64) return;
    const uintptr_t a = (uintptr_t)addr;
    if (a >= (uintptr_t)l->stack && a < (uintptr_t)l->stack + STACK_BYTES) return;      // the lane's own stack = its registers
    if (a >= (uintptr_t)l->wave && a < (uintptr_t)l->wave + sizeof(Wave)) return;           // the emulation's own exchange buffers
    g_in_check = true;
    flush_pending();
    const uint64_t ep = (g_block_gen << 40) | l->ncoll;
    for (uintptr_t w = a >> 2; w <= (a + bytes - 1) >> 2; ++w) {
        Shadow& sh = g_shadow[w];
        const uint32_t now = *(const volatile uint32_t*)(w * 4);
        if (store) {
            if (sh.rep == ep && (sh.rmask & ~(1ull << l->lane))) g_pending.push_back(Pending{w, now, l->lane, __builtin_ctzll(sh.rmask & ~(1ull << l->lane)), site, sh.rsite, false});
            if (sh.wep == ep && sh.wlane != l->lane) g_pending.push_back(Pending{w, now, l->lane, sh.wlane, site, sh.wsite, true});
            if (sh.wep != ep || sh.wlane != l->lane) sh.wold = now;      // (stores of different lanes to one word in one interval: the sinks of predicated-off stores - not reported)
            sh.wep = ep; sh.wlane = l->lane; sh.wsite = site;
        } else {
            if (sh.wep == ep && sh.wlane != l->lane && now != sh.wold)
                conflict("load behind another lane's store of a new value", w, l->lane, sh.wlane, site, sh.wsite);
            if (sh.rep != ep) { sh.rep = ep; sh.rmask = 0; }
            sh.rmask |= 1ull << l->lane; sh.rsite = site;
        }
    }
    g_in_check = false;
}
inline void lockstep_flush() { if (!g_in_check) { g_in_check = true; flush_pending(); g_in_check = false; } }
}  // namespace hipemu
#define HIPEMU_COV(n) \
    extern "C" __attribute__((weak, noinline)) void __sanitizer_cov_load##n(void* p) { hipemu::access(p, n, false, __builtin_return_address(0)); } \
    extern "C" __attribute__((weak, noinline)) void __sanitizer_cov_store##n(void* p) { hipemu::access(p, n, true, __builtin_return_address(0)); }
HIPEMU_COV(1) HIPEMU_COV(2) HIPEMU_COV(4) HIPEMU_COV(8) HIPEMU_COV(16)
extern "C" __attribute__((weak)) void __sanitizer_cov_trace_pc_guard(uint32_t*) {}
extern "C" __attribute__((weak)) void __sanitizer_cov_trace_pc_guard_init(uint32_t*, uint32_t*) {}
extern "C" __attribute__((weak)) long apx_emul_lockstep_conflicts() { return hipemu::g_conflicts; }
#endif

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
