// Host restatement of apex_amd/csrc/gfx950/lane_ops.h: the same names, each instruction of the gfx950 dialect spelled out per lane on the emulation's lane exchanges
// (tools/hipemu/hip/hip_runtime.h).  The emulation build finds this file where the product build finds the real one (<gfx950/lane_ops.h>, include path order), so the env
// kernel sources compile unchanged.  What is restated is the DATA MOVEMENT and the operation; rounding is the host's (v_rcp_f32 is a division here, fused multiply-adds
// are fmaf).  Test infrastructure only.
#pragma once
#include <hip/hip_runtime.h>
#include <gfx950/dynamic_lds.h>

#define APX_PIN(...) ((void)0)
#define APX_HAZARD_FENCE(...) ((void)0)
// a rendezvous of the wave's lanes (see the product header)
#define APX_LOCKSTEP() ((void)hipemu::exchange2(0u, 0u))
#define APX_CONVERGE() hipemu::converge()
namespace c4 {

__device__ __forceinline__ float rcpf(float x) { return 1.0f / x; }
__device__ __forceinline__ void wsync() { __builtin_amdgcn_wave_barrier(); }
template <int CTRL> __device__ __forceinline__ float dpp(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, true));
}
constexpr int NEWBCAST = 0x150;
// v_fmac_f32_dpp acc, acc, -m row_newbcast:K   ->   acc = fma(bcast_K(acc), -m, acc)
template <int K> __device__ __forceinline__ void fnmac_bcast(float& acc, float m) {
    const hipemu::Xchg e = hipemu::exchange2(hipemu::asu(acc), 0u);
    acc = fmaf(hipemu::dpp_get(e, NEWBCAST + K), -m, acc);
}
// the two-slot step: both instructions read their own slot's broadcast
template <int K> __device__ __forceinline__ void solve_step2(float& a0, float& a1, float m0, float m1) {
    const hipemu::Xchg e = hipemu::exchange2(hipemu::asu(a0), hipemu::asu(a1));
    a0 = fmaf(hipemu::dpp_get(e, NEWBCAST + K), -m0, a0);
    a1 = fmaf(hipemu::dpp_get(e, NEWBCAST + K, true), -m1, a1);
}
template <int K> __device__ __forceinline__ void fmac_bcast(float& acc, float src, float m) {
    const hipemu::Xchg e = hipemu::exchange2(hipemu::asu(src), 0u);
    acc = fmaf(hipemu::dpp_get(e, NEWBCAST + K), m, acc);
}
template <int K> __device__ __forceinline__ float mul_bcast(float src, float m) {
    const hipemu::Xchg e = hipemu::exchange2(hipemu::asu(src), 0u);
    return hipemu::dpp_get(e, NEWBCAST + K) * m;
}
template <int K> __device__ __forceinline__ void fnmac_bcast3(float& acc, float src, float m) {
    const hipemu::Xchg e = hipemu::exchange2(hipemu::asu(src), 0u);
    acc = fmaf(hipemu::dpp_get(e, NEWBCAST + K), -m, acc);
}
// v_fmac_f32_dpp acc, x, x row_shr:N bound_ctrl:1   ->   acc = fma(x of the lane N places down (0 without one), x, acc)
template <int N> __device__ __forceinline__ void fmac_shr(float& acc, float x) {
    const hipemu::Xchg e = hipemu::exchange2(hipemu::asu(x), 0u);
    acc = fmaf(hipemu::dpp_get(e, 0x110 + N), x, acc);
}
__device__ __forceinline__ void solve_fence(float&, float&) {}
template <int K> __device__ __forceinline__ float rcp_bcast(float x) {
    const hipemu::Xchg e = hipemu::exchange2(hipemu::asu(x), 0u);
    return 1.0f / hipemu::dpp_get(e, NEWBCAST + K);
}
// v_pk_fma_f32 d, a, b, c op_sel:[1,0,0] op_sel_hi:[1,1,1]: the HIGH half of a multiplies both halves of b
typedef float f2pk __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2pk pk_fma_hi(f2pk a, f2pk b, f2pk c) { return f2pk{fmaf(a.y, b.x, c.x), fmaf(a.y, b.y, c.y)}; }

}  // namespace c4
