// The gfx950 dialect of the env kernels in ONE place: every instruction the compiler cannot be asked for in C++ (DPP-operand VOP2 forms, the op_sel packed FMA,
// v_rcp_f32) as a named inline function, and the two compiler fences the hand-spaced sequences need as macros.  The kernel sources (env.hip, cassie_lane.h,
// cassie_complete.h, estimator_lane.h, cassie_common.h) contain no inline assembly of their own; they include this file as <gfx950/lane_ops.h>.
// (tools/hipemu/gfx950/lane_ops.h is the lane-exact host restatement of these operations, under which the same kernel sources run on the CPU in the test suite.)
#pragma once
#include <hip/hip_runtime.h>
#include <gfx950/dynamic_lds.h>

// APX_PIN("+v"(a), "+v"(b), ..): an opaque definition point of the listed registers - the compiler may not move, fold or re-associate their values across it.
// APX_HAZARD_FENCE(..): the same and two wait states: no compiler-generated definition of a listed register sits right in front of a DPP read of it by inline
// assembly (which the hazard recogniser cannot see).
#define APX_PIN(...) asm volatile("" : __VA_ARGS__)
#define APX_HAZARD_FENCE(...) asm volatile("s_nop 1" : __VA_ARGS__)
// APX_LOCKSTEP(): marks a place where the code relies on the wave being ONE instruction stream: every lane's loads above this line are performed before any lane's
// stores below it (no instruction is needed for that, so this expands to nothing).  The host emulation of the kernels runs the lanes one after the other between
// rendezvous points and makes this one; its lockstep checker (tools/hipemu) finds the places that need the mark.
#define APX_LOCKSTEP() ((void)0)
// APX_CONVERGE(): behind a branch that only some of the wave's env ROWS take and that contains cross-lane operations: the point where the rows are together again (the
// exec mask's business on the hardware: nothing to emit; the host emulation parks the rows that sat the branch out here until the others arrive).
#define APX_CONVERGE() ((void)0)
namespace c4 {

// v_rcp_f32 (1 ulp): __frcp_rn and '/' expand to the ~10-instruction correctly rounded division sequence
__device__ __forceinline__ float rcpf(float x) { return __builtin_amdgcn_rcpf(x); }
// The workgroup is ONE wave: LDS operations of a wave are processed in issue order, so a write -> read hand-off between lanes needs
// no s_barrier / s_waitcnt drain, only a fence that keeps the compiler from reordering across it.
__device__ __forceinline__ void wsync() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); }
template <int CTRL> __device__ __forceinline__ float dpp(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, true));
}
// acc -= bcast_K(acc) * m with the row broadcast as the DPP source of the fmac (see factor_lane for the hazard discipline these need)
template <int K> __device__ __forceinline__ void fnmac_bcast(float& acc, float m) {
    asm volatile("v_fmac_f32_dpp %0, %0, -%1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(m), "n"(K));
}
// one step of a triangular solve on the two leg slots at once: a_s -= bcast_K(a_s) * m_s.  Each slot's chain reads through DPP what it wrote one step
// earlier: the other slot's instruction and the s_nop are the two wait states that needs (the FIRST step of a chain is preceded by solve_fence)
template <int K> __device__ __forceinline__ void solve_step2(float& a0, float& a1, float m0, float m1) {
    asm volatile("s_nop 0\n\tv_fmac_f32_dpp %0, %0, -%2 row_newbcast:%4 row_mask:0xf bank_mask:0xf\n\tv_fmac_f32_dpp %1, %1, -%3 row_newbcast:%4 row_mask:0xf bank_mask:0xf"
                 : "+v"(a0), "+v"(a1) : "v"(m0), "v"(m1), "n"(K));
}
// acc += bcast_K(src) * m / bcast_K(src) * m (src is not written by these: the only hazard is a compiler-generated definition of src right in front of
// its first DPP read, which the caller's fence excludes)
template <int K> __device__ __forceinline__ void fmac_bcast(float& acc, float src, float m) {
    asm("v_fmac_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(m), "n"(K));
}
template <int K> __device__ __forceinline__ float mul_bcast(float src, float m) {
    float r;
    asm("v_mul_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(src), "v"(m), "n"(K));
    return r;
}
template <int K> __device__ __forceinline__ void fnmac_bcast3(float& acc, float src, float m) {
    asm("v_fmac_f32_dpp %0, %1, -%2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(m), "n"(K));
}
// acc += row_shr:N(x) * x (lanes without a source N places down contribute 0)
template <int N> __device__ __forceinline__ void fmac_shr(float& acc, float x) {
    asm("v_fmac_f32_dpp %0, %1, %1 row_shr:%2 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(acc) : "v"(x), "n"(N));
}
__device__ __forceinline__ void solve_fence(float& a0, float& a1) { APX_HAZARD_FENCE("+v"(a0), "+v"(a1)); }
template <int K> __device__ __forceinline__ float rcp_bcast(float x) {
    float r;
    asm volatile("v_rcp_f32_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf\n\ts_nop 0" : "=v"(r) : "v"(x), "n"(K));
    return r;
}
// d = (a.y, a.y) * b + c as ONE v_pk_fma_f32: op_sel takes the high half of the pair `a` for both result lanes
typedef float f2pk __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2pk pk_fma_hi(f2pk a, f2pk b, f2pk c) {
    f2pk d;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}

}  // namespace c4
