// Shared pieces of the Cassie substep kernels (gfx950): model constants, compile-time loops, 3-vector / quaternion / spatial algebra,
// MuJoCo's soft-constraint scalars (solref / solimp defaults), the per-env LDS hand-off layout and the optional phase profiler.
//
// What the substep replaces: cassie_sim_step_pd -> mj_step inside libcassiemujoco.so / MuJoCo 2.00
// (cassie/cassiemujoco/cassiemujoco.py:46-49, include/cassiemujoco.h:80).  The robot's topology is a
// compile-time constant (cassie_tables.h, generated from cassie.xml by tools/gen_model.py).
#pragma once
#include <hip/hip_runtime.h>
#include <utility>
#include "cassie_tables.h"
#include "env_state.h"

namespace c4 {
using namespace cmt;

constexpr int NB = CM_NBODY, NV = CM_NV, NQ = CM_NQ, NJ = CM_NJNT, NU = CM_NU, NM = CM_NM;
constexpr float DT = 0.0005f, GRAV = 9.81f, MINVAL = 1e-15f;

template <int B, int E, class F> __device__ __forceinline__ void sfor(F&& f) {
    if constexpr (B < E) { f(std::integral_constant<int, B>{}); sfor<B + 1, E>(f); }
}
template <int B, int E, class F> __device__ __forceinline__ void srfor(F&& f) {   // E-1 down to B
    if constexpr (B < E) { f(std::integral_constant<int, E - 1>{}); srfor<B, E - 1>(f); }
}

// Bit pattern of a float, opaque to the optimiser.  The env kernels are built with -ffast-math; there LLVM recognises `(bits & 0x7f800000) == 0x7f800000` on
// a plain bitcast as "is NaN or inf" and folds it to false (round 3: the disassembly held no trace of the round-2 divergence guards).  The empty asm
// hides where the integer came from.
__device__ __forceinline__ unsigned fbits(float v) { unsigned u = __float_as_uint(v); APX_PIN("+v"(u)); return u; }
__device__ __forceinline__ bool nonfinite(float v) { return (fbits(v) & 0x7f800000u) == 0x7f800000u; }      // NaN or +-inf

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

// ---- the same algebra on PAIRS (round 5): component .x = leg slot 0 (left), .y = slot 1 (right).  A lane of the tree stage carries body b of BOTH legs; written on
// 2-vectors, the arithmetic of the two slots is one v_pk_{mul,add,fma}_f32 per pair (VOP3P has no DPP / no select: cross-lane moves and v_cndmask stay per component, and
// a uniform operand is a splat the compiler folds into op_sel_hi).  Same expressions, same order of operations as the scalar forms above.
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 splat(float s) { return f2{s, s}; }
__device__ __forceinline__ f2 sel2(bool c, f2 a, f2 b) { return f2{c ? a.x : b.x, c ? a.y : b.y}; }
struct V3p { f2 x, y, z; };
__device__ __forceinline__ V3p lift(V3 v) { return {splat(v.x), splat(v.y), splat(v.z)}; }
__device__ __forceinline__ V3p operator+(V3p a, V3p b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3p operator-(V3p a, V3p b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3p operator*(V3p a, f2 s) { return {a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ V3p operator*(V3p a, float s) { return {a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ f2 dot(V3p a, V3p b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ f2 dot(V3 a, V3p b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3p cross(V3p a, V3p b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
__device__ __forceinline__ V3p sel2(bool c, V3p a, V3p b) { return {sel2(c, a.x, b.x), sel2(c, a.y, b.y), sel2(c, a.z, b.z)}; }
struct Q4p { f2 w, x, y, z; };
__device__ __forceinline__ Q4p lift(Q4 q) { return {splat(q.w), splat(q.x), splat(q.y), splat(q.z)}; }
__device__ __forceinline__ Q4p qmul(Q4p a, Q4p b) {
    return {a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z, a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
            a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x, a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w};
}
__device__ __forceinline__ Q4p sel2(bool c, Q4p a, Q4p b) { return {sel2(c, a.w, b.w), sel2(c, a.x, b.x), sel2(c, a.y, b.y), sel2(c, a.z, b.z)}; }
__device__ __forceinline__ Q4p qnormalize(Q4p q) {
    const f2 n2 = q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z;
    const bool ok0 = n2.x > 1e-30f, ok1 = n2.y > 1e-30f;
    const f2 s = {rsqrtf(n2.x), rsqrtf(n2.y)};
    const Q4p r = {q.w * s, q.x * s, q.y * s, q.z * s};
    return {f2{ok0 ? r.w.x : 1.f, ok1 ? r.w.y : 1.f}, f2{ok0 ? r.x.x : 0.f, ok1 ? r.x.y : 0.f}, f2{ok0 ? r.y.x : 0.f, ok1 ? r.y.y : 0.f}, f2{ok0 ? r.z.x : 0.f, ok1 ? r.z.y : 0.f}};
}
struct M3p { f2 m[9]; };
__device__ __forceinline__ M3p q2m(Q4p q) {
    const f2 w = q.w, x = q.x, y = q.y, z = q.z;
    return {{1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y), 2 * (x * y + w * z),
             1 - 2 * (x * x + z * z), 2 * (y * z - w * x), 2 * (x * z - w * y), 2 * (y * z + w * x),
             1 - 2 * (x * x + y * y)}};
}
__device__ __forceinline__ V3p mul(const M3p& R, V3p v) {
    return {R.m[0] * v.x + R.m[1] * v.y + R.m[2] * v.z, R.m[3] * v.x + R.m[4] * v.y + R.m[5] * v.z,
            R.m[6] * v.x + R.m[7] * v.y + R.m[8] * v.z};
}
__device__ __forceinline__ V3p mul(const M3& R, V3p v) {      // uniform rotation, pair vector
    return {R.m[0] * v.x + R.m[1] * v.y + R.m[2] * v.z, R.m[3] * v.x + R.m[4] * v.y + R.m[5] * v.z,
            R.m[6] * v.x + R.m[7] * v.y + R.m[8] * v.z};
}
__device__ __forceinline__ V3p col(const M3p& R, int k) { return {R.m[k], R.m[3 + k], R.m[6 + k]}; }
struct SVp { V3p a, l; };
__device__ __forceinline__ SVp lift(SV v) { return {lift(v.a), lift(v.l)}; }
__device__ __forceinline__ SVp operator+(SVp p, SVp q) { return {p.a + q.a, p.l + q.l}; }
__device__ __forceinline__ SVp operator*(SVp p, f2 s) { return {p.a * s, p.l * s}; }
__device__ __forceinline__ SVp crossMotion(SVp v, SVp s) { return {cross(v.a, s.a), cross(v.a, s.l) + cross(v.l, s.a)}; }
__device__ __forceinline__ SVp crossForce(SVp v, SVp f) { return {cross(v.a, f.a) + cross(v.l, f.l), cross(v.a, f.l)}; }
__device__ __forceinline__ f2 sdot(SVp m, SVp f) { return dot(m.a, f.a) + dot(m.l, f.l); }
struct SIp { f2 m; V3p h; f2 I[6]; };
__device__ __forceinline__ V3p symmul(const f2* I, V3p v) {
    return {I[0] * v.x + I[3] * v.y + I[4] * v.z, I[3] * v.x + I[1] * v.y + I[5] * v.z, I[4] * v.x + I[5] * v.y + I[2] * v.z};
}
__device__ __forceinline__ SVp imul(const SIp& s, SVp v) { return {symmul(s.I, v.a) + cross(s.h, v.l), v.l * s.m - cross(s.h, v.a)}; }

// local column (0..18) of a row of leg LEG -> global dof
template <int LEG> constexpr int c2d(int c) { return c < 6 ? c : c + 13 * LEG; }
constexpr int d2c(int d) { return d < 6 ? d : (d < 19 ? d : d - 13); }
// bodies whose poses the constraints need: per leg slot 0 achilles, 1 heel-spring, 2 plantar-rod, 3 foot, 4 tarsus, 5 shin
template <int LEG> constexpr int cbody(int s) { constexpr int t[6] = {5, 10, 12, 13, 9, 8}; return t[s] + 12 * LEG; }
static_assert(ct_jnt_type[0] == 0 && ct_jnt_type[1] == 0 && ct_jnt_type[2] == 0 && ct_jnt_type[3] == 2, "pelvis = 3 slides + ball");
static_assert(ct_body_parent[1] == 0, "pelvis hangs off the world");

__device__ __forceinline__ float impedance(float pos) {   // MuJoCo solimp defaults 0.9 0.95 0.001 0.5 2
    const float x = fminf(fabsf(pos) * 1000.f, 1.f);           // branch-free: y(0) = 0, y(1) = 1 are the two saturated values
    const float lo = 2.f * x * x, hi = 1.f - 2.f * (1.f - x) * (1.f - x);
    const float y = x <= 0.5f ? lo : hi;
    return 0.9f + y * 0.05f;
}
struct RowK { float K, B; };
__device__ __forceinline__ RowK solref(float timeconst) {   // dampratio 1, dmax 0.95
    return {1.f / (0.95f * 0.95f * timeconst * timeconst), 2.f / (0.95f * timeconst)};
}

// optional phase profiling (lane 0 of workgroup 0): cumulative shader cycles per phase, read by tools/t_prof.py
#ifdef APX_PROF
__device__ unsigned long long g_prof_acc[48];
__device__ unsigned long long g_prof_last;
#define PROF(i) do { if (threadIdx.x == 0 && blockIdx.x == 0) { const unsigned long long t__ = clock64(); c4::g_prof_acc[i] += t__ - c4::g_prof_last; c4::g_prof_last = t__; } } while (0)
#define PROF_START() do { if (threadIdx.x == 0 && blockIdx.x == 0) c4::g_prof_last = clock64(); } while (0)
#if APX_PROF >= 2      // fine probes inside the stages (slots 12..47); they add s_memtime + a full LDS drain each, so the coarse totals shift a little
#define PROF2(i) PROF(i)
#else
#define PROF2(i) do {} while (0)
#endif
#else
#define PROF(i) do {} while (0)
#define PROF_START() do {} while (0)
#define PROF2(i) do {} while (0)
#endif

#ifdef APX_WAVETIME
__device__ unsigned g_wavefeat[4096 * 4];      // experiment build (tools/t_wavetime.py): optional row groups each wave ran
#endif
// stage hand-off layout inside the per-env LDS region (floats, offsets from L4_WK)
constexpr int WK_M = 0, WK_CDOF = WK_M + NM, WK_SMOOTH = WK_CDOF + 6 * NV, WK_QS = WK_SMOOTH + NV, WK_PTS = WK_QS + NV,
              WK_PEL = WK_PTS + 60, WK_LD = WK_PEL + 24, WK_DISQ = WK_LD + NM, WK_ZT = WK_DISQ + NV, WK_QACC = WK_ZT + NV,
              WK_MISC = WK_QACC + NV /* ncon0 ncon1 nlim0 nlim1 footmaskL footmaskR costL costR */, WK_ZP2 = WK_MISC + 8 /* z~ pelvis warm-start parts L, R */,
              WK_TOTAL = WK_ZP2 + 12;
// WK_PTS per leg (30): eq0 p1,p2 | eq1 p1,p2 | capsule ends: foot e0,e1, tarsus e0,e1, shin e0,e1

constexpr int MAXC = 2;      // contact slots per leg per substep (oracle: MAXCON_LEG)
template <int K> __device__ __forceinline__ V3 ldv3(const St& S) { return {S.W(K), S.W(K + 1), S.W(K + 2)}; }

}  // namespace c4
