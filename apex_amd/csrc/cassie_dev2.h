// Device-side Cassie rigid-body step for gfx950, second generation: one environment per lane, EVERY index static.
//
// What it replaces: cassie_sim_step_pd -> mj_step inside libcassiemujoco.so / MuJoCo 2.00
// (cassie/cassiemujoco/cassiemujoco.py:46-49, include/cassiemujoco.h:80).
//
// The robot's topology is a compile-time constant (cassie_tables.h, generated from cassie.xml), so the whole
// substep is unrolled over bodies / dofs / ancestor chains with template recursion: every array below is indexed
// by constants only and lives in VGPRs/AGPRs (no scratch addressing, no table lookups at run time).
//   * FK + velocity + RNE in one pass over the 25 bodies, all spatial quantities about the pelvis origin
//   * mass matrix in MuJoCo's sparse ancestor-chain layout (307 entries), L^T D L in registers
//   * constraint rows are sparse over [6 pelvis dofs | 13 dofs of one leg]; each is whitened on the spot,
//     y~ = D^-1/2 L^-T J^T, and parked in LDS as float4 chunks (ds_read_b128).  Projected Gauss-Seidel runs on
//     z~ = sum y~ f in registers, leg-major: 2 connects, <=1 limit, <=3 contacts per leg.  A pyramidal contact
//     stores only its 3 whitened basis vectors (normal, 2 tangents) + their Gram matrix; its 4 rows are swept
//     sequentially through the Gram matrix, which is arithmetic-for-arithmetic the same Gauss-Seidel.
#pragma once
#include <hip/hip_runtime.h>
#include <utility>
#include "cassie_tables.h"

namespace c2 {
using namespace cmt;

constexpr int NB = CM_NBODY, NV = CM_NV, NQ = CM_NQ, NJ = CM_NJNT, NU = CM_NU, NM = CM_NM;
constexpr float DT = 0.0005f, GRAV = 9.81f, MINVAL = 1e-15f;

template <int B, int E, class F> __device__ __forceinline__ void sfor(F&& f) {
    if constexpr (B < E) { f(std::integral_constant<int, B>{}); sfor<B + 1, E>(f); }
}
template <int B, int E, class F> __device__ __forceinline__ void srfor(F&& f) {   // E-1 down to B
    if constexpr (B < E) { f(std::integral_constant<int, E - 1>{}); srfor<B, E - 1>(f); }
}

struct V3 { float x, y, z; };
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
struct Q4 { float w, x, y, z; };
__device__ __forceinline__ Q4 qmul(Q4 a, Q4 b) {
    return {a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z, a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
            a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x, a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w};
}
__device__ __forceinline__ Q4 qnormalize(Q4 q) {
    const float n2 = q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z;
    if (!(n2 > 1e-30f)) return {1.f, 0.f, 0.f, 0.f};
    const float s = rsqrtf(n2);
    return {q.w * s, q.x * s, q.y * s, q.z * s};
}
struct M3 { float m[9]; };
__device__ __forceinline__ M3 q2m(Q4 q) {
    const float w = q.w, x = q.x, y = q.y, z = q.z;
    return {{1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y), 2 * (x * y + w * z),
             1 - 2 * (x * x + z * z), 2 * (y * z - w * x), 2 * (x * z - w * y), 2 * (y * z + w * x),
             1 - 2 * (x * x + y * y)}};
}
__device__ __forceinline__ V3 mul(const M3& R, V3 v) {
    return {R.m[0] * v.x + R.m[1] * v.y + R.m[2] * v.z, R.m[3] * v.x + R.m[4] * v.y + R.m[5] * v.z,
            R.m[6] * v.x + R.m[7] * v.y + R.m[8] * v.z};
}
__device__ __forceinline__ V3 col(const M3& R, int k) { return {R.m[k], R.m[3 + k], R.m[6 + k]}; }
struct SV { V3 a, l; };
__device__ __forceinline__ SV operator+(SV p, SV q) { return {p.a + q.a, p.l + q.l}; }
__device__ __forceinline__ SV operator*(SV p, float s) { return {p.a * s, p.l * s}; }
__device__ __forceinline__ SV crossMotion(SV v, SV s) { return {cross(v.a, s.a), cross(v.a, s.l) + cross(v.l, s.a)}; }
__device__ __forceinline__ SV crossForce(SV v, SV f) { return {cross(v.a, f.a) + cross(v.l, f.l), cross(v.a, f.l)}; }
__device__ __forceinline__ float sdot(SV m, SV f) { return dot(m.a, f.a) + dot(m.l, f.l); }
struct SI { float m; V3 h; float I[6]; };
__device__ __forceinline__ V3 symmul(const float* I, V3 v) {
    return {I[0] * v.x + I[3] * v.y + I[4] * v.z, I[3] * v.x + I[1] * v.y + I[5] * v.z, I[4] * v.x + I[5] * v.y + I[2] * v.z};
}
__device__ __forceinline__ SV imul(const SI& s, SV v) { return {symmul(s.I, v.a) + cross(s.h, v.l), v.l * s.m - cross(s.h, v.a)}; }
template <int K> constexpr V3 cv3(const float* p) { return V3{p[3 * K], p[3 * K + 1], p[3 * K + 2]}; }

// ------------------------------------------------------------------------------------------------ LDS row store
// float4 chunk c of this lane lives at ((c * EPW + lane) * 16) bytes: every access is one ds_read/write_b128.
// EPW = environments per workgroup = active lanes per wave (launch geometry knob).  Measured on MI355X at 4096 envs:
// EPW 64 (64 workgroups, one per CU) 12.8 ms per env step, EPW 8 (512 workgroups, 2 per CU) 22.8 ms: the substep is a serial
// per-env program, so spreading the same lanes over more waves buys nothing and the co-resident workgroups contend for the
// CU's instruction cache / LDS pipe.  Kept as a knob for large env counts only.
#ifndef APX_EPW
#define APX_EPW 64
#endif
#if defined(APX_GEN) && APX_GEN == 4
constexpr int EPW = 1;      // generation 4: the row store of an env is contiguous in its own LDS region
#else
constexpr int EPW = APX_EPW;
#endif
constexpr int CH_EQ = 0;          // 12 equality rows x 5 chunks  [16 cols | b R invA f]
constexpr int CH_LIM = 60;        // 2 limit slots x 6 chunks     [19 cols + pad | b R invA f]
constexpr int CH_CON = 72;        // 6 contact slots x 14 chunks  [n,t1,t2: 3 x 13 cols + pad | G6 R mu | b4 | f4 | invA4]
constexpr int CH_TOTAL = 158;     // 161,792 B of the CU's 163,840 (156 row chunks + 2 hand-off chunks; one workgroup per CU)
struct Lds {
    float4* base;   // already offset by lane
    __device__ __forceinline__ float4 rd(int c) const { return base[c * EPW]; }
    __device__ __forceinline__ void wr(int c, float4 v) const { base[c * EPW] = v; }
};

// local column (0..18) of a row of leg LEG -> global dof
template <int LEG> constexpr int c2d(int c) { return c < 6 ? c : c + 13 * LEG; }
constexpr int d2c(int d) { return d < 6 ? d : (d < 19 ? d : d - 13); }
// whitened supports (closed under ancestors), as local columns
struct SetPL { static constexpr int N = 16; static constexpr int c[16] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 12, 13, 14, 16, 17, 18, -1}; };   // plantar-rod <-> foot connect
struct SetAC { static constexpr int N = 16; static constexpr int c[16] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}; };    // achilles-rod <-> heel-spring connect
struct SetFT { static constexpr int N = 13; static constexpr int c[13] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 12, 13, 14, 18}; };               // chain pelvis -> foot (all contact geoms)
struct SetALL { static constexpr int N = 19; static constexpr int c[19] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18}; };
template <class SET> constexpr bool has(int col) { for (int i = 0; i < SET::N; ++i) if (SET::c[i] == col) return true; return false; }

struct Dyn2 {
    float friction; V3 fn, ft1, ft2;
};

// All the per-substep state that lives in registers between phases
struct Fwd {
    float LD[NM];
    float dsqrt[NV], disqrt[NV];
    float ut[NV];                 // u~ = D^-1/2 L^-T qfrc_smooth
    float vt[NV], wt[NV];         // D^1/2 L qvel, D^1/2 L qacc_warmstart
    float zt[NV];                 // z~
    float qacc[NV];
    SV cdof[NV];
    V3 o;
    // kinematic products the constraints / sensors need
    V3 xpos_c[2][8]; M3 xmat_c[2][8];   // per leg: achilles, heel-spring, plantar, foot, tarsus, shin (0..5)
    Q4 footq[2]; V3 footp[2];
    SV pel_cacc, pel_cvel; M3 pel_mat;
    int ncon[2], nlim[2];
    unsigned footmask;            // bit (3*leg + slot): that contact slot belongs to a foot capsule
    float foot_fz[2];
    float acc[3];
};

// x <- L^-T x (leaves -> root), x <- L^-1 x, y <- L x, y <- L^T x: all static
__device__ __forceinline__ void solve_LT(const float (&LD)[NM], float (&x)[NV]) {
    srfor<0, NV>([&](auto I) {
        constexpr int i = I;
        sfor<1, ct_dof_depth[i]>([&](auto A) { constexpr int a = A; x[ct_dof_anc[16 * i + a]] -= LD[ct_dof_madr[i] + a] * x[i]; });
    });
}
__device__ __forceinline__ void solve_L(const float (&LD)[NM], float (&x)[NV]) {
    sfor<0, NV>([&](auto I) {
        constexpr int i = I;
        sfor<1, ct_dof_depth[i]>([&](auto A) { constexpr int a = A; x[i] -= LD[ct_dof_madr[i] + a] * x[ct_dof_anc[16 * i + a]]; });
    });
}
__device__ __forceinline__ void mul_L(const float (&LD)[NM], const float (&x)[NV], float (&y)[NV]) {
    sfor<0, NV>([&](auto I) {
        constexpr int i = I;
        float s = x[i];
        sfor<1, ct_dof_depth[i]>([&](auto A) { constexpr int a = A; s += LD[ct_dof_madr[i] + a] * x[ct_dof_anc[16 * i + a]]; });
        y[i] = s;
    });
}
__device__ __forceinline__ void mul_LT(const float (&LD)[NM], const float (&x)[NV], float (&y)[NV]) {
    sfor<0, NV>([&](auto I) { y[I] = x[I]; });
    sfor<0, NV>([&](auto I) {
        constexpr int i = I;
        sfor<1, ct_dof_depth[i]>([&](auto A) { constexpr int a = A; y[ct_dof_anc[16 * i + a]] += LD[ct_dof_madr[i] + a] * x[i]; });
    });
}
template <bool WITH_SQRT>
__device__ __forceinline__ void factor(float (&LD)[NM], float (&dsqrt)[NV], float (&disqrt)[NV]) {
    srfor<0, NV>([&](auto K) {
        constexpr int k = K, kk = ct_dof_madr[k];
        const float dinv = __frcp_rn(LD[kk]);
        sfor<1, ct_dof_depth[k]>([&](auto A) {
            constexpr int a = A, i = ct_dof_anc[16 * k + a], ki = kk + a;
            const float tmp = LD[ki] * dinv;
            sfor<0, ct_dof_depth[i]>([&](auto J) { constexpr int jj = J; LD[ct_dof_madr[i] + jj] -= tmp * LD[ki + jj]; });
            LD[ki] = tmp;
        });
        if constexpr (WITH_SQRT) { disqrt[k] = rsqrtf(LD[kk]); dsqrt[k] = LD[kk] * disqrt[k]; }
    });
}

// whiten a row given over local columns of leg LEG, restricted to a static support set (closed under ancestors)
template <int LEG, class SET>
__device__ __forceinline__ void whiten(const Fwd& w, float (&J)[19]) {
    srfor<0, 19>([&](auto C) {
        constexpr int c = C;
        if constexpr (has<SET>(c)) {
            constexpr int i = c2d<LEG>(c);
            sfor<1, ct_dof_depth[i]>([&](auto A) { constexpr int a = A; J[d2c(ct_dof_anc[16 * i + a])] -= w.LD[ct_dof_madr[i] + a] * J[c]; });
        }
    });
    sfor<0, 19>([&](auto C) { constexpr int c = C; if constexpr (has<SET>(c)) J[c] *= w.disqrt[c2d<LEG>(c)]; });
}

// translational Jacobian of point p of body B accumulated (with sign) into local-column rows
template <int B>
__device__ __forceinline__ void jac_point(const Fwd& w, V3 p, float sign, float (&Jx)[19], float (&Jy)[19], float (&Jz)[19]) {
    const V3 r = p - w.o;
    constexpr int last = ct_body_lastdof[B];
    sfor<0, ct_dof_depth[last]>([&](auto A) {
        constexpr int d = ct_dof_anc[16 * last + A], c = d2c(d);
        V3 v;
        if constexpr (d < 3) v = w.cdof[d].l;                                   // pelvis slides: pure translation
        else v = w.cdof[d].l + cross(w.cdof[d].a, r);
        Jx[c] += sign * v.x; Jy[c] += sign * v.y; Jz[c] += sign * v.z;
    });
}

__device__ __forceinline__ float impedance(float pos) {   // MuJoCo solimp defaults 0.9 0.95 0.001 0.5 2
    const float x = fabsf(pos) * 1000.f;
    if (x >= 1.f) return 0.95f;
    if (x <= 0.f) return 0.9f;
    const float y = x <= 0.5f ? 2.f * x * x : 1.f - 2.f * (1.f - x) * (1.f - x);
    return 0.9f + y * 0.05f;
}
struct RowK { float K, B; };
__device__ __forceinline__ RowK solref(float timeconst) {   // dampratio 1, dmax 0.95
    return {1.f / (0.95f * 0.95f * timeconst * timeconst), 2.f / (0.95f * timeconst)};
}

template <int LEG, class SET>
__device__ __forceinline__ void dots(const Fwd& w, const float (&y)[19], float& vel, float& ju, float& jw, float& nn) {
    vel = ju = jw = nn = 0.f;
    sfor<0, 19>([&](auto C) {
        constexpr int c = C;
        if constexpr (has<SET>(c)) {
            constexpr int d = c2d<LEG>(c);
            vel += y[c] * w.vt[d]; ju += y[c] * w.ut[d]; jw += y[c] * w.wt[d]; nn += y[c] * y[c];
        }
    });
}

}  // namespace c2
