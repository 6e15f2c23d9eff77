// The COMPLETE constraint set of a forward pass, for the envs whose pass needs more rows than the lane map of cassie_lane.h instantiates (round 5).
//
// The lane-mapped rows / Gram-space sweeps of stage_rows_pgs_lane carry, per leg, 6 connect rows, ONE limit row, TWO floor contacts, and three left-right capsule
// pairs; everything else cassie.xml can produce - further capsule ends on the floor (foot, tarsus, shin: cassie.xml:119-144), the hip-pitch capsules (:101, :163-164),
// the pelvis sphere (:87), a second, third ... joint limit of a leg, the other six capsule pairs - was DETECTED and counted in I_SAT for four rounds and not solved.
// Here it is solved: when a pass saturates (SAT_* of that pass), the env's rows are rebuilt WITHOUT caps in the order of the fp64 oracle (oracle/cassie_phys.cpp:
// per leg the connects, every active limit in joint order, every penetrating capsule end foot / tarsus / shin / hip-pitch; then the pelvis sphere; then the capsule
// pairs) and swept by projected Gauss-Seidel in the whitened space z~ itself - row vector y~ = D^-1/2 L^-T J^T distributed over the env's 16 lanes (lane l: leg dof l of
// the row's leg; the 6 pelvis entries on every lane), residual = y~ . z~ + b + R f by one DPP reduction, z~ += y~ df.  No Gram matrix: the row count is open.  Measured
// (tools/t_complete_ms.py, every env of the launch in the same pose): a pass costs 0.25 - 0.9 ms against 0.08 ms of the fast path (4 - 12 x: ~ 50 us to build and whiten the
// rows, 4 - 18 us per sweep) - nothing on a rollout in which a wave steps its envs at its own pace (env_rollout_kernel), and never on a walking policy (no saturation).
// The result replaces what the capped fast path left in WK_ZT and in the contact-slot records; tree, factor and finish stages are the same code for both.
#pragma once
#include "cassie_lane.h"

namespace c4 {

// ---- pool of whitened basis vectors in LDS: record i = 24 words: [0..12] the 13 leg columns (lane l writes its own), [13] leg of the row (or one extra word of the
// record's kind), [14..15] / [16..19] the pelvis columns 4, 5 / 0..3, [20..23] row scalars (single rows: b, R, 1 / (A + R), f; the normal basis of a contact keeps more,
// below): words [12..23] are three aligned 128-bit loads.  Records 0 .. CP_ROWS_CAP - 1 live in the row store below the contact-slot records, the rest in the factor
// hand-off WK_LD (free in a substep: the factor is in registers; factor_lane clears its zero words before it loads).
// Records beyond those 38 (a robot lying on the floor: 17 contacts = 51 basis vectors) go to a per-env overflow area in HBM (apx_env::cp_pool, CP_HBM_CAP records): their
// vectors are read-only after the construction (one __threadfence), their scalars - rewritten every sweep - are accessed with agent-scope atomics (no stale L1 line).
// The solve is compiled twice: cp_solve<true> when every record of the wave's envs is in LDS (ds_* instructions only, no tier branch), cp_solve<false> otherwise.
constexpr int CP_STRIDE = 24, CP_ROWS_CAP = R4_CON / CP_STRIDE, CP_LD_CAP = NM / CP_STRIDE, CP_LDS_CAP = CP_ROWS_CAP + CP_LD_CAP, CP_HBM_CAP = 96, CP_CAP = CP_LDS_CAP + CP_HBM_CAP;
static_assert(CP_ROWS_CAP == 26 && CP_LD_CAP == 12, "basis pool");
static_assert(CP_CAP >= 12 + 16 + 3 * 17 + 2 * 9 && CP_CAP < 256, "every row cassie.xml can produce fits the pool");
// written by the tree stage for this path: the four hip-pitch capsule ends (left e0, e1, right e0, e1) and the pelvis sphere centre, 15 words in the gap between the parked
// row vectors of the fast path (38 x 16 words) and the contact-slot records
constexpr int XB_EXTRA = 38 * 16;
static_assert(XB_EXTRA + 15 <= R4_CON && 13 * XB_SZ <= XB_EXTRA, "extra collision points");

struct GeomTab { float pos[27], axis[27], half[9], radius[9]; int body[9]; };
constexpr GeomTab make_geomtab() {
    GeomTab t{};
    for (int i = 0; i < 27; ++i) { t.pos[i] = ct_geom_pos[i]; t.axis[i] = ct_geom_axis[i]; }
    for (int i = 0; i < 9; ++i) { t.half[i] = ct_geom_half[i]; t.radius[i] = ct_geom_radius[i]; t.body[i] = ct_geom_body[i]; }
    return t;
}
__device__ const GeomTab kGeom = make_geomtab();
// limited joints of a leg in joint order: leg-local dof, qpos offset inside the leg block, range
struct LimTab { int dof[8], qoff[8]; float lo[8], hi[8]; };
constexpr LimTab make_limtab() {
    LimTab t{};
    int n = 0;
    for (int j = 0; j < NJ; ++j)
        if (ct_jnt_limited[j] && ct_jnt_body[j] >= 2 && ct_jnt_body[j] < 14) {
            t.dof[n] = ct_jnt_dofadr[j] - 6; t.qoff[n] = ct_jnt_qposadr[j] - 7; t.lo[n] = ct_jnt_range[2 * j]; t.hi[n] = ct_jnt_range[2 * j + 1]; ++n;
        }
    return t;
}
__device__ const LimTab kLim = make_limtab();
static_assert(make_limtab().dof[7] == 12 && make_limtab().dof[3] == 6, "eight limited joints per leg");
// leg-local dof bitmask (bit k = leg dof k) of the dofs that move leg-local body lb
struct ChainTab { unsigned m[12]; };
constexpr ChainTab make_chaintab() { ChainTab t{}; for (int lb = 0; lb < 12; ++lb) t.m[lb] = chain_mask<0>(2 + lb) >> 6; return t; }
__device__ const ChainTab kChain = make_chaintab();

// record words: [0..12] leg columns, [13] leg of the row / one extra word per record kind, [14..15] pelvis columns 4, 5, [16..19] pelvis columns 0..3, [20..23] scalars:
// words [12..23] of a record are three aligned 128-bit LDS loads
constexpr int CPW_MISC = 13;
constexpr int cpw_p(int P) { return P < 4 ? 16 + P : 10 + P; }
constexpr int CP_LD0 = WK_LD + 1;      // first word of the WK_LD tier: 16-byte aligned inside the env's LDS region
static_assert((L4_WK + CP_LD0) % 4 == 0 && L4_ROWS % 4 == 0 && L4_ES % 4 == 0 && CP_STRIDE % 4 == 0 && CP_LD0 + CP_LD_CAP * CP_STRIDE <= WK_LD + NM, "aligned records");
// Gram blocks of the pyramidal contacts: 8-word entries (6 used) in the motion-axis table, first entry 16-byte aligned
constexpr int CP_GRAM0 = WK_CDOF + 1, CP_GRAM_CAP = (6 * NV - 1) / 8;
static_assert((L4_WK + CP_GRAM0) % 4 == 0 && CP_GRAM_CAP >= 17, "contact Gram table");
typedef float f4r __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) f4r lf4r;

struct CompleteCtx {
    const St& S; lfloat* rows; lfloat* ld; float* hbm; int l, lc; bool act; V3 o;
    float cdl[2][6];      // motion axis of this lane's leg dof, both legs
    float cdp[6][6];      // the six pelvis axes (uniform)
    LaneFac F; float disq[2], disqp[6];
    LaneVec qs, qv, qw;
};
// LDS tier: records 0 .. 25 in the row store, 26 .. 37 in WK_LD (a select between two LDS addresses, no branch); HBM tier behind
__device__ __forceinline__ lfloat* cp_lds(const CompleteCtx& C, int i) { return (i < CP_ROWS_CAP ? C.rows : C.ld - CP_STRIDE * CP_ROWS_CAP) + CP_STRIDE * i; }
__device__ __forceinline__ float* cp_hbm(const CompleteCtx& C, int i) { return C.hbm + (size_t)CP_STRIDE * (i - CP_LDS_CAP); }
// scalar words of a record (rewritten during the sweeps): plain LDS words, or agent-scope atomics on the HBM tier.  LDS = true: the caller knows that every record of
// the wave's envs is in LDS (the case outside a robot lying flat on the floor) - no tier branch, LDS instructions only
template <bool LDS>
__device__ __forceinline__ float sc_get(const CompleteCtx& C, int i, int k) {
    if (LDS || i < CP_LDS_CAP) return cp_lds(C, i)[k];
    return __int_as_float(__hip_atomic_load((int*)(cp_hbm(C, i) + k), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
template <bool LDS>
__device__ __forceinline__ void sc_set(const CompleteCtx& C, int i, int k, float v) {
    if (LDS || i < CP_LDS_CAP) cp_lds(C, i)[k] = v;
    else __hip_atomic_store((int*)(cp_hbm(C, i) + k), __float_as_int(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float lv_dot(const CompleteCtx& C, const LaneVec& x, const LaneVec& y) {
    float r = red16(C.l < 13 ? x.a[0] * y.a[0] + x.a[1] * y.a[1] : 0.f);
    sfor<0, 6>([&](auto P) { r += x.p[P] * y.p[P]; });
    return r;
}
// J += sign * d/dq [ dir . (point p of leg-local body lb of leg `leg`; lb < 0: the pelvis) ]
__device__ __forceinline__ void jac_add(const CompleteCtx& C, LaneVec& J, int leg, int lb, V3 p, V3 dir, float sign) {
    const V3 q = cross(p - C.o, dir);      // dir . (a x r) = a . (r x dir)
    sfor<0, 6>([&](auto P) { J.p[P] += sign * (C.cdp[P][3] * dir.x + C.cdp[P][4] * dir.y + C.cdp[P][5] * dir.z + C.cdp[P][0] * q.x + C.cdp[P][1] * q.y + C.cdp[P][2] * q.z); });
    if (lb < 0) return;
    const unsigned m = kChain.m[lb];
    const bool on = C.l < 13 && ((m >> C.l) & 1u);
    sfor<0, 2>([&](auto Lg) {
        const float v = C.cdl[Lg][3] * dir.x + C.cdl[Lg][4] * dir.y + C.cdl[Lg][5] * dir.z + C.cdl[Lg][0] * q.x + C.cdl[Lg][1] * q.y + C.cdl[Lg][2] * q.z;
        J.a[Lg] += (on && leg == Lg) ? sign * v : 0.f;
    });
}
// whiten in place (y~ = D^-1/2 L^-T J^T) and return the raw dots J . qvel, J . qacc_smooth, J . qacc_warmstart
__device__ __forceinline__ void whiten(const CompleteCtx& C, LaneVec& J, float& vel, float& ju, float& jw) {
    if (C.l >= 13) J.a[0] = J.a[1] = 0.f;
    vel = lv_dot(C, J, C.qv); ju = lv_dot(C, J, C.qs); jw = lv_dot(C, J, C.qw);
    solve_LT_lane(C.F, J);
    sfor<0, 2>([&](auto Sd) { J.a[Sd] *= C.disq[Sd]; });
    sfor<0, 6>([&](auto P) { J.p[P] *= C.disqp[P]; });
    if (C.l >= 13) J.a[0] = J.a[1] = 0.f;
}
// store a whitened vector as pool record(s): a row of ONE leg takes one record, a left-right pair row two (left columns + pelvis, right columns)
__device__ __forceinline__ void cp_store(const CompleteCtx& C, int i, const LaneVec& y, int leg) {
    const float col = leg ? y.a[1] : y.a[0];
    if (i < CP_LDS_CAP) {
        lfloat* r = cp_lds(C, i);
        if (C.act) r[C.l] = col;
        if (C.l == 0) { sfor<0, 6>([&](auto P) { r[cpw_p(P)] = y.p[P]; }); r[CPW_MISC] = (float)leg; }
    } else {
        float* r = cp_hbm(C, i);
        if (C.act) r[C.l] = col;
        if (C.l == 0) { sfor<0, 6>([&](auto P) { r[cpw_p(P)] = y.p[P]; }); sc_set<false>(C, i, CPW_MISC, (float)leg); }
    }
}
// a record's vector part in registers: this lane's leg column (0 on lanes 13..15), the pelvis columns, the extra word
struct RecVec { float ya, p[6], misc; };
template <bool LDS>
__device__ __forceinline__ RecVec rec_vec(const CompleteCtx& C, int i) {
    RecVec v;
    if (LDS || i < CP_LDS_CAP) {
        lfloat* r = cp_lds(C, i);
        const float col = r[C.lc];
        const f4r q0 = *(lf4r*)(r + 12), q1 = *(lf4r*)(r + 16);
        v.ya = C.act ? col : 0.f; v.misc = q0.y; v.p[0] = q1.x; v.p[1] = q1.y; v.p[2] = q1.z; v.p[3] = q1.w; v.p[4] = q0.z; v.p[5] = q0.w;
    } else {
        const float* r = cp_hbm(C, i);
        v.ya = C.act ? r[C.lc] : 0.f;
        sfor<0, 6>([&](auto P) { v.p[P] = r[cpw_p(P)]; });
        v.misc = sc_get<false>(C, i, CPW_MISC);
    }
    return v;
}
template <bool LDS>
__device__ __forceinline__ f4r rec_sc(const CompleteCtx& C, int i) {
    if (LDS || i < CP_LDS_CAP) return *(lf4r*)(cp_lds(C, i) + 20);
    return f4r{sc_get<false>(C, i, 20), sc_get<false>(C, i, 21), sc_get<false>(C, i, 22), sc_get<false>(C, i, 23)};
}
// rho = y~ . z~ and z~ += y~ df for a record of leg `leg` (lane-uniform over the env)
__device__ __forceinline__ float rv_dot(const RecVec& v, bool leg, const LaneVec& z) {
    float s = red16(v.ya * (leg ? z.a[1] : z.a[0]));
    sfor<0, 6>([&](auto P) { s += v.p[P] * z.p[P]; });
    return s;
}
__device__ __forceinline__ void rv_axpy(const RecVec& v, bool leg, float df, LaneVec& z) {
    const float t = v.ya * df;
    z.a[0] += leg ? 0.f : t; z.a[1] += leg ? t : 0.f;
    sfor<0, 6>([&](auto P) { z.p[P] += v.p[P] * df; });
}

// row kinds of the sweep list (one word per row in LDS, oracle order): kind | first record << 4
enum { CK_EQ = 0, CK_LIMIT = 1, CK_CONTACT = 2, CK_PAIR = 3 };

// Warm start, projected Gauss-Seidel over the sweep list in oracle order, hand-off to the finish stage.
template <bool LDS>
__device__ __forceinline__ void cp_solve(const CompleteCtx& C, int nlist, int pgs_iters, float mu, int over, int nfoot0, int nfoot1) {
    const St& S = C.S;
    const int l = C.l;
    auto list_at = [&](int i) -> lfloat& { return *(i < 32 ? &S.W(WK_ZT) + i : &S.W(WK_DISQ) + (i - 32)); };
    // ---------------------------------------------------------------- warm start: z~0 = sum y~ f0; kept only if its dual cost beats f = 0 (mj_fwdConstraint)
    LaneVec z{{0.f, 0.f}, {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}};
    float cost = 0.f;
    for (int i = 0; i < nlist; ++i) {
        const int w = __float_as_int(list_at(i)), kind = w & 15, rec = (w >> 4) & 255;
        const RecVec v0 = rec_vec<LDS>(C, rec);
        const f4r s0 = rec_sc<LDS>(C, rec);
        if (kind == CK_CONTACT) {
            const RecVec v1 = rec_vec<LDS>(C, rec + 1), v2 = rec_vec<LDS>(C, rec + 2);
            const f4r f = rec_sc<LDS>(C, rec + 1);
            const float Rpy = sc_get<LDS>(C, rec + 2, 20);
            const bool leg = v0.misc != 0.f;
            rv_axpy(v0, leg, f.x + f.y + f.z + f.w, z); rv_axpy(v1, leg, mu * (f.x - f.y), z); rv_axpy(v2, leg, mu * (f.z - f.w), z);
            cost += f.x * (0.5f * Rpy * f.x + s0.x) + f.y * (0.5f * Rpy * f.y + s0.y) + f.z * (0.5f * Rpy * f.z + s0.z) + f.w * (0.5f * Rpy * f.w + s0.w);
        } else {
            const float f = s0.w;
            rv_axpy(v0, kind == CK_PAIR ? false : v0.misc != 0.f, f, z);
            if (kind == CK_PAIR) rv_axpy(rec_vec<LDS>(C, rec + 1), true, f, z);
            cost += f * (0.5f * s0.y * f + s0.x);
        }
    }
    cost += 0.5f * lv_dot(C, z, z);
    const bool cold = cost > 0.f;
    if (cold) { z.a[0] = z.a[1] = 0.f; sfor<0, 6>([&](auto P) { z.p[P] = 0.f; }); }
    wsync();
    if (cold && l == 0)
        for (int i = 0; i < nlist; ++i) {
            const int w = __float_as_int(list_at(i)), kind = w & 15, rec = (w >> 4) & 255;
            if (kind == CK_CONTACT) { sfor<0, 4>([&](auto K) { sc_set<LDS>(C, rec + 1, 20 + K, 0.f); }); }
            else sc_set<LDS>(C, rec, 23, 0.f);
        }
    wsync();
    // ---------------------------------------------------------------- projected Gauss-Seidel over the list, oracle order.  The list word of entry i + 1 is fetched while
    // entry i is worked on; the loads of an entry (vectors, scalars, Gram block) issue together in front of its arithmetic.
    for (int it = 0; it < pgs_iters; ++it) {
        int w = __float_as_int(list_at(0));
        for (int i = 0; i < nlist; ++i) {
            const int kind = w & 15, rec = (w >> 4) & 255, ci = w >> 12;
            const RecVec v0 = rec_vec<LDS>(C, rec);
            const f4r s0 = rec_sc<LDS>(C, rec);
            const int wn = __float_as_int(list_at(i + 1 < nlist ? i + 1 : i));
            if (kind == CK_CONTACT) {
                const RecVec v1 = rec_vec<LDS>(C, rec + 1), v2 = rec_vec<LDS>(C, rec + 2);
                const f4r fv = rec_sc<LDS>(C, rec + 1);
                const float Rpy = sc_get<LDS>(C, rec + 2, 20);
                const lfloat* g = &S.W(CP_GRAM0 + 8 * ci);
                const f4r g0 = *(const lf4r*)g;
                const float gnn = g0.x, gn1 = g0.y, gn2 = g0.z, g11 = g0.w, g12 = g[4], g22 = g[5];
                const bool leg = v0.misc != 0.f;
                float rn = rv_dot(v0, leg, z), ra = rv_dot(v1, leg, z), rb = rv_dot(v2, leg, z);
                float f[4] = {fv.x, fv.y, fv.z, fv.w};
                const float bk[4] = {s0.x, s0.y, s0.z, s0.w};
                float dn = 0.f, d1 = 0.f, d2 = 0.f;
                sfor<0, 4>([&](auto K) {
                    constexpr int k = K;
                    const float sg = (k & 1) ? -mu : mu;
                    const float gj = k < 2 ? gn1 : gn2, gjj = k < 2 ? g11 : g22;
                    const float res = rn + sg * (k < 2 ? ra : rb) + bk[k] + Rpy * f[k];
                    const float A = gnn + 2.f * sg * gj + mu * mu * gjj + Rpy;
                    const float fnew = fmaxf(f[k] - res * rcpf(A), 0.f), df = fnew - f[k];
                    f[k] = fnew;
                    // the row n + sg t_j moves the three basis residuals
                    rn += df * (gnn + sg * gj); ra += df * (gn1 + sg * (k < 2 ? g11 : g12)); rb += df * (gn2 + sg * (k < 2 ? g12 : g22));
                    dn += df; if constexpr (k < 2) d1 += sg * df; else d2 += sg * df;
                });
                rv_axpy(v0, leg, dn, z); rv_axpy(v1, leg, d1, z); rv_axpy(v2, leg, d2, z);
                if (l == 0) {
                    if constexpr (LDS) *(lf4r*)(cp_lds(C, rec + 1) + 20) = f4r{f[0], f[1], f[2], f[3]};
                    else sfor<0, 4>([&](auto K) { sc_set<false>(C, rec + 1, 20 + K, f[K]); });
                }
            } else {
                const bool pair = kind == CK_PAIR;
                const bool leg = pair ? false : v0.misc != 0.f;
                RecVec v1 = rec_vec<LDS>(C, rec + (pair ? 1 : 0));      // the right-leg half of a pair row; a copy of the record otherwise (not used)
                float rho = rv_dot(v0, leg, z);
                const float rho1 = rv_dot(v1, true, z);
                rho += pair ? rho1 : 0.f;
                const float f = s0.w, res = rho + s0.x + s0.y * f;
                float fnew = f - res * s0.z;
                if (kind != CK_EQ) fnew = fmaxf(fnew, 0.f);
                const float df = fnew - f;
                rv_axpy(v0, leg, df, z);
                rv_axpy(v1, true, pair ? df : 0.f, z);
                if (l == 0) sc_set<LDS>(C, rec, 23, fnew);
            }
            w = wn;
            wsync();
        }
    }
    // ---------------------------------------------------------------- hand-off to the finish stage: contact-slot records of the FOOT capsules (foot-force readout), z~
    if (l == 0) {
        lfloat* const rows = C.rows;
        sfor<0, 2 * MAXC>([&](auto Sl) { lfloat* cr = rows + R4_CON + R4_CONSZ * Sl; cr[7] = 0.f; sfor<0, 4>([&](auto K) { cr[12 + K] = 0.f; }); });
        for (int i = 0; i < nlist; ++i) {
            const int w = __float_as_int(list_at(i)), kind = w & 15, rec = (w >> 4) & 255;
            if (kind != CK_CONTACT) continue;
            const int slot = (int)sc_get<LDS>(C, rec + 2, CPW_MISC);
            if (slot <= 0 || slot > MAXC) continue;
            const int leg = (int)sc_get<LDS>(C, rec, CPW_MISC);
            lfloat* cr = rows + R4_CON + R4_CONSZ * (MAXC * leg + slot - 1);
            cr[7] = 1.f; cr[8] = sc_get<LDS>(C, rec + 2, 21); cr[9] = sc_get<LDS>(C, rec + 2, 22); cr[10] = sc_get<LDS>(C, rec + 1, CPW_MISC);      // world z of the contact frame (n, t1, t2)
            sfor<0, 4>([&](auto K) { cr[12 + K] = sc_get<LDS>(C, rec + 1, 20 + K); });
        }
        S.W(WK_MISC + 0) = (float)(nfoot0 < MAXC ? nfoot0 : MAXC); S.W(WK_MISC + 1) = (float)(nfoot1 < MAXC ? nfoot1 : MAXC);
        if (over) S.I(I_SAT) |= 16;      // more rows than the pool holds (never seen): reported
    }
    wsync();
    if (l < 13) { S.W(WK_ZT + 6 + l) = z.a[0]; S.W(WK_ZT + 19 + l) = z.a[1]; }
    if (l == 0) sfor<0, 6>([&](auto P) { S.W(WK_ZT + P) = z.p[P]; });
    wsync();
}

template <bool HF>
__device__ __forceinline__ void rows_pgs_complete(const St& S, const FacRegs& FR, const FacTail& FT, float* rows_generic, int pgs_iters, const Hf& hf, float* pool) {
    lfloat* const rows = S.p + L4_ROWS;      // the env's row store (= rows_generic, as an LDS pointer: everything below compiles to ds_* instructions)
    const int l = (int)(threadIdx.x & 15);
    CompleteCtx C{S, rows, &S.W(CP_LD0), pool + (size_t)S.env * CP_HBM_CAP * CP_STRIDE, l, l < 13 ? l : 12, l < 13, {S(F_QPOS), S(F_QPOS + 1), S(F_QPOS + 2)}};
    {
        const int ll = l < 13 ? l : 12;
        sfor<0, 2>([&](auto Lg) { sfor<0, 6>([&](auto I) { C.cdl[Lg][I] = S.W(WK_CDOF + 6 * (6 + 13 * Lg + ll) + I); }); });
        sfor<0, 6>([&](auto P) { sfor<0, 6>([&](auto I) { C.cdp[P][I] = S.W(WK_CDOF + 6 * P + I); }); });
        sfor<0, 2>([&](auto Sd) {
            sfor<0, 13>([&](auto J) { C.F.Lr[Sd][J] = FR.Lr[Sd][J]; C.F.Lc[Sd][J] = FT.Lc[Sd][J]; });
            sfor<0, 6>([&](auto P) { C.F.w[Sd][P] = FR.w[Sd][P]; });
            C.F.D[Sd] = FT.D[Sd]; C.disq[Sd] = FR.disq[Sd];
        });
        sfor<0, 6>([&](auto Pi) { C.F.Dp[Pi] = FT.Dp[Pi]; C.disqp[Pi] = FR.disqp[Pi]; sfor<0, Pi>([&](auto Qi) { C.F.Lp[Pi][Qi] = FR.Lp[Pi][Qi]; }); });
        C.qs = FR.qs; C.qv = FR.qv; C.qw = FR.qw;
        if (l >= 13) { C.qs.a[0] = C.qs.a[1] = C.qv.a[0] = C.qv.a[1] = C.qw.a[0] = C.qw.a[1] = 0.f; }
    }
    const float mu = S(F_FRIC);
    const V3 fn = {S(F_FLOOR), S(F_FLOOR + 1), S(F_FLOOR + 2)}, ft1 = {S(F_FLOOR + 3), S(F_FLOOR + 4), S(F_FLOOR + 5)}, ft2 = {S(F_FLOOR + 6), S(F_FLOOR + 7), S(F_FLOOR + 8)};
    // extra collision points of the tree stage, read before the pool overwrites nothing of them (they sit outside the pool)
    float xpt[15];
    sfor<0, 15>([&](auto I) { xpt[I] = rows[XB_EXTRA + I]; });
    // the sweep list: WK_ZT is rewritten at the end, its 32 words hold up to 32 row entries meanwhile; further entries spill into WK_DISQ (32 words)
    auto list_at = [&](int i) -> lfloat& { return i < 32 ? S.W(WK_ZT + i) : S.W(WK_DISQ + (i - 32)); };
    int nrec = 0, nlist = 0, over = 0;
    int nfoot[2] = {0, 0};      // foot-capsule contacts recorded per leg for the foot-force readout of the finish stage (contact-slot records)
    auto push = [&](int kind, int rec) { if (nlist < 64) { if (l == 0) list_at(nlist) = __int_as_float(kind | rec << 4); ++nlist; } else over = 1; };

    // ---------------------------------------------------------------- single rows: connects, limits (scalars in the record: b, R, 1 / (A + R), f0)
    auto finish_single = [&](LaneVec& J, int leg, int kind, float pos, float imp_pos, float diag, float tc) {
        if (nrec + 1 > CP_CAP) { over = 1; return; }
        float vel, ju, jw;
        whiten(C, J, vel, ju, jw);
        const float nn = lv_dot(C, J, J);
        const RowK kb = solref(tc);
        const float imp = impedance(imp_pos);
        const float R = fmaxf(MINVAL, (1.f - imp) * rcpf(imp) * diag);
        const float aref = -kb.B * vel - kb.K * imp * pos;
        float f = -(jw - aref) * rcpf(R);
        if (kind != CK_EQ && f < 0.f) f = 0.f;
        cp_store(C, nrec, J, leg);
        if (l == 0) { sc_set<false>(C, nrec, 20, ju - aref); sc_set<false>(C, nrec, 21, R); sc_set<false>(C, nrec, 22, rcpf(nn + R)); sc_set<false>(C, nrec, 23, f); }
        push(kind, nrec); ++nrec;
    };
    // ---------------------------------------------------------------- pyramidal floor contact: three records (n, t1, t2).  Scalars: n-record [20..23] = b of the four pyramid
    // rows; t1-record [19] = world z of t2, [20..23] = f of the four rows; t2-record [19] = foot slot (0: not a foot capsule), [20] = R of the pyramid, [21], [22] = world z
    // of n, t1.  The 3 x 3 Gram block of the basis (6 words) goes to the contact's entry of a table in WK_CDOF (the motion axes are in registers by now).
    int ncont = 0;
    auto add_contact = [&](int leg, int lb, V3 ctr, float rad, float dist, V3 nrm, float tran, bool isfoot) {
        if (nrec + 3 > CP_CAP || ncont >= CP_GRAM_CAP) { over = 1; return; }
        V3 t1 = ft1, t2 = ft2;
        if constexpr (HF) {
            const bool uy = fabsf(nrm.y) < 0.5f;
            V3 t = {0.f, uy ? 1.f : 0.f, uy ? 0.f : 1.f};
            t = t - nrm * dot(nrm, t); t1 = t * rsqrtf(dot(t, t)); t2 = cross(nrm, t1);
        }
        const V3 cp = ctr - nrm * (rad + 0.5f * dist);
        const V3 dirs[3] = {nrm, t1, t2};
        float vel[3], ju[3], jw[3];
        LaneVec Y[3];
        sfor<0, 3>([&](auto K) {
            Y[K] = LaneVec{{0.f, 0.f}, {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}};
            jac_add(C, Y[K], leg, lb, cp, dirs[K], 1.f);
            whiten(C, Y[K], vel[K], ju[K], jw[K]);
            cp_store(C, nrec + K, Y[K], leg);
        });
        const float gnn = lv_dot(C, Y[0], Y[0]), gn1 = lv_dot(C, Y[0], Y[1]), gn2 = lv_dot(C, Y[0], Y[2]), g11 = lv_dot(C, Y[1], Y[1]), g12 = lv_dot(C, Y[1], Y[2]), g22 = lv_dot(C, Y[2], Y[2]);
        const RowK kb = solref(0.005f);
        const float imp = impedance(dist);
        const float R1 = fmaxf(MINVAL, (1.f - imp) * rcpf(imp) * (tran + mu * mu * tran));
        const float Rpy = fmaxf(MINVAL, 2.f * mu * mu * R1), iRpy = rcpf(Rpy);
        wsync();
        if (l == 0) {
            sfor<0, 4>([&](auto K) {
                constexpr int k = K;
                const float sg = (k & 1) ? -mu : mu;
                const float vk = vel[0] + sg * (k < 2 ? vel[1] : vel[2]), uk = ju[0] + sg * (k < 2 ? ju[1] : ju[2]), wk = jw[0] + sg * (k < 2 ? jw[1] : jw[2]);
                const float aref = -kb.B * vk - kb.K * imp * dist;
                sc_set<false>(C, nrec, 20 + k, uk - aref);
                sc_set<false>(C, nrec + 1, 20 + k, fmaxf(-(wk - aref) * iRpy, 0.f));
            });
            // (cp_store wrote the leg into the extra word of all three records; the t1 and t2 records keep something else there)
            sc_set<false>(C, nrec + 1, CPW_MISC, t2.z); sc_set<false>(C, nrec + 2, CPW_MISC, isfoot ? (float)(1 + nfoot[leg]) : 0.f); sc_set<false>(C, nrec + 2, 20, Rpy); sc_set<false>(C, nrec + 2, 21, nrm.z); sc_set<false>(C, nrec + 2, 22, t1.z);
            lfloat* g = &S.W(CP_GRAM0 + 8 * ncont);
            g[0] = gnn; g[1] = gn1; g[2] = gn2; g[3] = g11; g[4] = g12; g[5] = g22;
        }
        if (isfoot) ++nfoot[leg];
        push(CK_CONTACT, nrec | ncont << 8); nrec += 3; ++ncont;
    };
    auto floor_hit = [&](V3 ctr, float rad, V3& nrm) { return floor_dist_dev<HF>(hf, fn, ctr, rad, nrm); };

    for (int leg = 0; leg < 2; ++leg) {
        const lfloat* pts = &S.W(WK_PTS + 30 * leg);
        // connects (cassie.xml:225-230): plantar rod <-> foot, achilles rod <-> heel spring
        for (int e = 0; e < 2; ++e) {
            const V3 p1 = {pts[6 * e], pts[6 * e + 1], pts[6 * e + 2]}, p2 = {pts[6 * e + 3], pts[6 * e + 4], pts[6 * e + 5]};
            const V3 c = p1 - p2;
            const int b1 = ct_eq_body1[0] - 2, b2 = ct_eq_body2[0] - 2, b3 = ct_eq_body1[1] - 2, b4 = ct_eq_body2[1] - 2;      // leg-local bodies: plantar rod, foot, achilles rod, heel spring
            const float tran = S(F_BIW + 2 + 12 * leg + (e ? b3 : b1)) + S(F_BIW + 2 + 12 * leg + (e ? b4 : b2));
            const float nc = sqrtf(dot(c, c));
            for (int k = 0; k < 3; ++k) {
                const V3 dir = {k == 0 ? 1.f : 0.f, k == 1 ? 1.f : 0.f, k == 2 ? 1.f : 0.f};
                LaneVec J{{0.f, 0.f}, {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}};
                jac_add(C, J, leg, e ? b3 : b1, p1, dir, 1.f);
                jac_add(C, J, leg, e ? b4 : b2, p2, dir, -1.f);
                finish_single(J, leg, CK_EQ, k == 0 ? c.x : k == 1 ? c.y : c.z, nc, tran, 0.005f);
            }
        }
        // every active joint limit, joint order (mj_instantiateLimit)
        for (int j = 0; j < 8; ++j) {
            const float q = S(F_QPOS + 7 + 14 * leg + kLim.qoff[j]);
            const float dlo = q - kLim.lo[j], dhi = kLim.hi[j] - q;
            for (int side = 0; side < 2; ++side) {
                const float dist = side ? dhi : dlo;
                if (!(dist < 0.f)) continue;
                LaneVec J{{0.f, 0.f}, {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}};
                const float sg = side ? -1.f : 1.f;
                if (leg == 0) J.a[0] = l == kLim.dof[j] ? sg : 0.f; else J.a[1] = l == kLim.dof[j] ? sg : 0.f;
                finish_single(J, leg, CK_LIMIT, dist, dist, S(F_DIW + 6 + 13 * leg + kLim.dof[j]), 0.02f);
            }
        }
        // floor contacts of this leg: foot, tarsus, shin (ends from the tree stage), hip-pitch capsule (extra points)
        for (int gi = 0; gi < 4; ++gi) {
            const int g = 2 * gi + leg;
            const int lb = kGeom.body[g] - 2 - 12 * leg;
            const float rad = kGeom.radius[g];
            for (int e = 0; e < 2; ++e) {
                V3 ctr;
                if (gi < 3) ctr = {pts[12 + 6 * gi + 3 * e], pts[12 + 6 * gi + 3 * e + 1], pts[12 + 6 * gi + 3 * e + 2]};
                else { const int o3 = 6 * leg + 3 * e; ctr = {xpt[o3], xpt[o3 + 1], xpt[o3 + 2]}; }
                V3 nrm;
                const float dist = floor_hit(ctr, rad, nrm);
                if (!(dist < 0.f)) continue;
                add_contact(leg, lb, ctr, rad, dist, nrm, S(F_BIW + kGeom.body[g]), gi == 0);
            }
        }
    }
    {   // pelvis sphere (cassie.xml:87)
        const V3 ctr = {xpt[12], xpt[13], xpt[14]};
        V3 nrm;
        const float dist = floor_hit(ctr, kGeom.radius[8], nrm);
        if (dist < 0.f) add_contact(0, -1, ctr, kGeom.radius[8], dist, nrm, S(F_BIW + 1), false);
    }
    // left-right capsule pairs (condim 1): pair order of the oracle (left geom outer, right geom inner)
    for (int gi = 0; gi < 3; ++gi)
        for (int gj = 0; gj < 3; ++gj) {
            const lfloat* pl = &S.W(WK_PTS + 12 + 6 * gi); const lfloat* pr = &S.W(WK_PTS + 30 + 12 + 6 * gj);
            const V3 p1 = {pl[0], pl[1], pl[2]}, q1 = {pl[3], pl[4], pl[5]}, p2 = {pr[0], pr[1], pr[2]}, q2 = {pr[3], pr[4], pr[5]};
            const V3 d1 = q1 - p1, d2 = q2 - p2, r = p1 - p2;
            const float a = dot(d1, d1), e = dot(d2, d2), f = dot(d2, r), c = dot(d1, r), b = dot(d1, d2), den = a * e - b * b;
            float sp = den > 1e-12f ? fminf(fmaxf((b * f - c * e) * rcpf(den), 0.f), 1.f) : 0.f;
            float tp = (b * sp + f) * rcpf(e);
            {
                const float ia = rcpf(a), sp_lo = fminf(fmaxf(-c * ia, 0.f), 1.f), sp_hi = fminf(fmaxf((b - c) * ia, 0.f), 1.f);
                sp = tp > 1.f ? sp_hi : sp; sp = tp < 0.f ? sp_lo : sp;
                tp = fminf(fmaxf(tp, 0.f), 1.f);
            }
            const V3 c1 = p1 + d1 * sp, dv = (p2 + d2 * tp) - c1;
            const float len = sqrtf(dot(dv, dv));
            const float rl = kGeom.radius[2 * gi], rr = kGeom.radius[2 * gj + 1];
            const float dist = len - rl - rr;
            if (!(dist < 0.f && len > 1e-9f)) continue;
            if (nrec + 2 > CP_CAP) { over = 1; continue; }
            const V3 nn = dv * rcpf(fmaxf(len, 1e-12f));
            const V3 cp = c1 + nn * (rl + 0.5f * dist);
            LaneVec J{{0.f, 0.f}, {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}};
            jac_add(C, J, 1, kGeom.body[2 * gj + 1] - 14, cp, nn, 1.f);
            jac_add(C, J, 0, kGeom.body[2 * gi] - 2, cp, nn, -1.f);
            float vel, ju, jw;
            whiten(C, J, vel, ju, jw);
            const float n2 = lv_dot(C, J, J);
            const float tran = S(F_BIW + kGeom.body[2 * gi]) + S(F_BIW + kGeom.body[2 * gj + 1]);
            const RowK kb = solref(0.005f);
            const float imp = impedance(dist);
            const float R = fmaxf(MINVAL, (1.f - imp) * rcpf(imp) * tran);
            const float aref = -kb.B * vel - kb.K * imp * dist;
            cp_store(C, nrec, J, 0);
            { LaneVec Jr = J; sfor<0, 6>([&](auto P) { Jr.p[P] = 0.f; }); cp_store(C, nrec + 1, Jr, 1); }
            if (l == 0) { sc_set<false>(C, nrec, 20, ju - aref); sc_set<false>(C, nrec, 21, R); sc_set<false>(C, nrec, 22, rcpf(n2 + R)); sc_set<false>(C, nrec, 23, fmaxf(-(jw - aref) * rcpf(R), 0.f)); }
            push(CK_PAIR, nrec); nrec += 2;
        }
    if (nrec > CP_LDS_CAP) __threadfence();      // vectors on the HBM tier: visible to every lane's plain loads from here on
    wsync();
    // every record of the wave's envs in LDS (always, short of a robot lying flat on the floor): the solve compiles without the tier branches and without a global access
    APX_CONVERGE();      // (the rows of the wave that are in here got their records in loops of their own length)
    if (__builtin_amdgcn_ballot_w64(nrec > CP_LDS_CAP) == 0ull) cp_solve<true>(C, nlist, pgs_iters, mu, over, nfoot[0], nfoot[1]);
    else cp_solve<false>(C, nlist, pgs_iters, mu, over, nfoot[0], nfoot[1]);
}

// The rest of a substep for the envs of a wave whose pass saturated, OUT OF LINE and called where nothing of the fast path is live any more (after the finish stage of the
// other envs): the factor again from WK_M (the fast path's copy lives in registers that a call could only keep in callee-saved ones - carried through the sweeps, they
// pushed reloads into the inline-asm DPP sequences, which the hazard recogniser cannot see), the complete rows and their solve, then the env's own finish stage.
template <bool HF>
__device__ __noinline__ void substep_complete(St S, float* rows_arg, int pgs_iters, Hf hf, float* pool, int do_euler) {
    float* const rows = (float*)(S.p + L4_ROWS);      // = rows_arg, spelled as a cast of the LDS pointer: behind the call boundary the compiler would otherwise have to treat it as a flat address
    FacRegs FR; FacTail FT;
    stage_factor_lane<false>(S, FR, FT);
    wsync();
    rows_pgs_complete<HF>(S, FR, FT, rows, pgs_iters, hf, pool);
    wsync();
    stage_finish_lane(S, rows, do_euler != 0, FT, FR);
    wsync();
}

}  // namespace c4
