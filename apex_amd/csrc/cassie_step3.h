// Cassie substep, generation 3: same arithmetic as cassie_step2.h, re-staged so that no phase needs more live
// values than the 512-entry VGPR+AGPR file (generation 2 spilled ~16k scratch accesses per substep).
//
// Phase A  depth-first walk of the kinematic tree (template recursion): pose, motion axes, velocity, acceleration and
//          subtree force / composite inertia are returned up the recursion, so only the root-to-body PATH is live.
//          Mass-matrix rows, bias forces, motion axes and the constraint anchor points go straight to a per-env
//          workspace column in HBM/L2 (coalesced, written once, read once).
// Phase B  M -> registers, sparse L^T D L, qacc_smooth.
// Phase C  constraint rows per leg: Jacobian from the stored motion axes, dots against qvel / qacc_warm /
//          qacc_smooth on the RAW row, whitening, commit to LDS (float4 chunks).
// Phase D  projected Gauss-Seidel in the whitened space (cassie_step2.h pgs_leg).
// Phase E  qacc, sensors, implicit-damping Euler (second factorisation).
#pragma once
#include "cassie_step2.h"

namespace c3 {
using namespace c2;

// optional phase profiling (lane 0 of workgroup 0): cumulative shader cycles per phase, read by tools/t_prof.py
#ifdef APX_PROF
__device__ unsigned long long g_prof_acc[12];
__device__ unsigned long long g_prof_last;
#define PROF(i) do { if (threadIdx.x == 0 && blockIdx.x == 0) { const unsigned long long t__ = clock64(); c3::g_prof_acc[i] += t__ - c3::g_prof_last; c3::g_prof_last = t__; } } while (0)
#define PROF_START() do { if (threadIdx.x == 0 && blockIdx.x == 0) c3::g_prof_last = clock64(); } while (0)
#else
#define PROF(i) do {} while (0)
#define PROF_START() do {} while (0)
#endif

// workspace column layout (floats per env)
constexpr int WK_M = 0, WK_CDOF = WK_M + NM, WK_SMOOTH = WK_CDOF + 6 * NV, WK_QS = WK_SMOOTH + NV, WK_PTS = WK_QS + NV,
              WK_PEL = WK_PTS + 60, WK_LD = WK_PEL + 24, WK_DISQ = WK_LD + NM, WK_ZT = WK_DISQ + NV, WK_QACC = WK_ZT + NV,
              WK_MISC = WK_QACC + NV /* ncon0 ncon1 nlim0 nlim1 footmaskL footmaskR costL costR */, WK_ZP2 = WK_MISC + 8 /* z~ pelvis warm-start parts L, R */,
              WK_TOTAL = WK_ZP2 + 12;
// WK_PTS per leg (30): eq0 p1,p2 | eq1 p1,p2 | capsule ends: foot e0,e1, tarsus e0,e1, shin e0,e1

constexpr int MAXC = 2;      // contacts kept per leg per step (oracle: MAXCON_LEG); LDS keeps 3 slots per leg for layout stability
struct Node { V3 pos; Q4 quat; M3 mat; SV vel, acc; };
struct Acc { SI crb; SV frc; };

struct Fw3 {
    float zt[NV];
    int ncon[2], nlim[2];
    unsigned footmask;
    float foot_fz[2];
    float acc[3];
};

__device__ __forceinline__ SV ldcdof(const St& S, int d) {
    return {{S.W(WK_CDOF + 6 * d), S.W(WK_CDOF + 6 * d + 1), S.W(WK_CDOF + 6 * d + 2)},
            {S.W(WK_CDOF + 6 * d + 3), S.W(WK_CDOF + 6 * d + 4), S.W(WK_CDOF + 6 * d + 5)}};
}
template <int K> __device__ __forceinline__ void stv3(const St& S, V3 v) { S.W(K) = v.x; S.W(K + 1) = v.y; S.W(K + 2) = v.z; }
template <int K> __device__ __forceinline__ V3 ldv3(const St& S) { return {S.W(K), S.W(K + 1), S.W(K + 2)}; }

// ---------------------------------------------------------------------------------------------- phase A
template <bool QPOS0, int B, class QP>
__device__ __forceinline__ Acc visit(const St& S, const QP& qp, const Node& par, V3 o, SV (&pc)[14]) {
    Node me;
    constexpr int nd = ct_body_dofnum[B], d0 = ct_body_dofadr[B];
    if constexpr (B == 1) {
        me.pos = {qp(0), qp(1), qp(2)};
        me.quat = qnormalize(Q4{qp(3), qp(4), qp(5), qp(6)});
        me.mat = q2m(me.quat);
        pc[0] = {{0, 0, 0}, {1, 0, 0}}; pc[1] = {{0, 0, 0}, {0, 1, 0}}; pc[2] = {{0, 0, 0}, {0, 0, 1}};
        sfor<0, 3>([&](auto K) { pc[3 + K] = {col(me.mat, K), {0, 0, 0}}; });
        if constexpr (!QPOS0) {
            SV v = {{0, 0, 0}, {S(F_QVEL), S(F_QVEL + 1), S(F_QVEL + 2)}};
            SV a = {{0, 0, 0}, {0, 0, GRAV}};
            const SV vp = v;
            sfor<3, 6>([&](auto D) { const float qd = S(F_QVEL + D); a = a + crossMotion(vp, pc[D]) * qd; v = v + pc[D] * qd; });
            me.vel = v; me.acc = a;
        }
    } else {
        me.pos = par.pos + mul(par.mat, cv3<B>(ct_body_pos));
        Q4 quat = qmul(par.quat, Q4{ct_body_quat[4 * B], ct_body_quat[4 * B + 1], ct_body_quat[4 * B + 2], ct_body_quat[4 * B + 3]});
        if constexpr (ct_body_jntnum[B] == 1) {
            constexpr int j = ct_body_jntadr[B], adr = ct_jnt_qposadr[j];
            if constexpr (ct_jnt_type[j] == 1) {
                float sn, cs;
                __sincosf(0.5f * (qp(adr) - ct_jnt_ref[j]), &sn, &cs);
                quat = qmul(quat, Q4{cs, 0.f, 0.f, sn});
            } else {
                quat = qmul(quat, qnormalize(Q4{qp(adr), qp(adr + 1), qp(adr + 2), qp(adr + 3)}));
            }
        }
        me.quat = qnormalize(quat); me.mat = q2m(me.quat);
        if constexpr (nd > 0) {
            constexpr int j = ct_body_jntadr[B];
            const V3 r = o - me.pos;
            if constexpr (ct_jnt_type[j] == 1) { const V3 ax = col(me.mat, 2); pc[ct_dof_depth[d0] - 1] = {ax, cross(ax, r)}; }
            else sfor<0, 3>([&](auto K) { const V3 ax = col(me.mat, K); pc[ct_dof_depth[d0 + K] - 1] = {ax, cross(ax, r)}; });
        }
        if constexpr (!QPOS0) {
            SV v = par.vel, a = par.acc;
            const SV vp = v;
            sfor<0, nd>([&](auto K) {
                constexpr int d = d0 + K;
                const float qd = S(F_QVEL + d);
                a = a + crossMotion(vp, pc[ct_dof_depth[d] - 1]) * qd; v = v + pc[ct_dof_depth[d] - 1] * qd;
            });
            me.vel = v; me.acc = a;
        }
    }
    // motion axes -> workspace (the constraint Jacobians read them back in phase C)
    sfor<0, nd>([&](auto K) {
        constexpr int d = d0 + K;
        const SV c = pc[ct_dof_depth[d] - 1];
        S.W(WK_CDOF + 6 * d) = c.a.x; S.W(WK_CDOF + 6 * d + 1) = c.a.y; S.W(WK_CDOF + 6 * d + 2) = c.a.z;
        S.W(WK_CDOF + 6 * d + 3) = c.l.x; S.W(WK_CDOF + 6 * d + 4) = c.l.y; S.W(WK_CDOF + 6 * d + 5) = c.l.z;
    });
    // spatial inertia about o in world axes (mass randomisation changes the mass only, cassie.py:640)
    Acc acc;
    {
        const M3& R = me.mat;
        constexpr const float* Ib = ct_body_inertia + 9 * B;
        float RI[9], Iw[9];
        sfor<0, 3>([&](auto I) { sfor<0, 3>([&](auto K) { RI[3 * I + K] = R.m[3 * I] * Ib[K] + R.m[3 * I + 1] * Ib[3 + K] + R.m[3 * I + 2] * Ib[6 + K]; }); });
        sfor<0, 3>([&](auto I) { sfor<0, 3>([&](auto K) { if constexpr (K >= I) Iw[3 * I + K] = RI[3 * I] * R.m[3 * K] + RI[3 * I + 1] * R.m[3 * K + 1] + RI[3 * I + 2] * R.m[3 * K + 2]; }); });
        const float m = S(F_MASS + B);
        const V3 r = me.pos + mul(R, cv3<B>(ct_body_ipos)) - o;
        const float rr = dot(r, r);
        SI c;
        c.m = m; c.h = r * m;
        c.I[0] = Iw[0] + m * (rr - r.x * r.x); c.I[1] = Iw[4] + m * (rr - r.y * r.y); c.I[2] = Iw[8] + m * (rr - r.z * r.z);
        c.I[3] = Iw[1] - m * r.x * r.y; c.I[4] = Iw[2] - m * r.x * r.z; c.I[5] = Iw[5] - m * r.y * r.z;
        acc.crb = c;
        if constexpr (!QPOS0) acc.frc = imul(c, me.acc) + crossForce(me.vel, imul(c, me.vel));
    }
    // mj_setConst pass: world COM of the bodies that carry constraints (slot layout = cslot: achilles, heel-spring,
    // plantar-rod, foot, tarsus, shin), 3 floats each, 18 per leg
    if constexpr (QPOS0 && cslot(B) >= 0) stv3<WK_PTS + 30 * (B >= 14 ? 1 : 0) + 3 * cslot(B)>(S, me.pos + mul(me.mat, cv3<B>(ct_body_ipos)));
    // points the constraints need, sensors
    if constexpr (!QPOS0) {
        constexpr int leg = B >= 14 ? 1 : 0, lb = B - 12 * leg, base = WK_PTS + 30 * leg;
        sfor<0, 2>([&](auto E) {      // connect anchors of this leg's two equalities
            constexpr int e = 2 * leg + E;
            if constexpr (ct_eq_body1[e] == B) stv3<base + 6 * E>(S, me.pos + mul(me.mat, cv3<e>(ct_eq_anchor1)));
            if constexpr (ct_eq_body2[e] == B) stv3<base + 6 * E + 3>(S, me.pos + mul(me.mat, cv3<e>(ct_eq_anchor2)));
        });
        sfor<0, 3>([&](auto G) {      // capsule ends: geoms are ordered foot L,R, tarsus L,R, shin L,R
            constexpr int g = 2 * G + leg;
            if constexpr (ct_geom_body[g] == B) {
                const V3 c = me.pos + mul(me.mat, cv3<g>(ct_geom_pos));
                const V3 ax = mul(me.mat, cv3<g>(ct_geom_axis)) * ct_geom_half[g];
                stv3<base + 12 + 6 * G>(S, c + ax); stv3<base + 12 + 6 * G + 3>(S, c - ax);
            }
        });
        if constexpr (lb == 13) {     // foot pose for the reward / foot velocity (cassie.py:328-331,426-427)
            S(F_FWD + 2 + 4 * leg) = me.quat.w; S(F_FWD + 3 + 4 * leg) = me.quat.x; S(F_FWD + 4 + 4 * leg) = me.quat.y; S(F_FWD + 5 + 4 * leg) = me.quat.z;
            S(F_FWD + 10 + 3 * leg) = me.pos.x; S(F_FWD + 11 + 3 * leg) = me.pos.y; S(F_FWD + 12 + 3 * leg) = me.pos.z - 0.0550841220316708f;
        }
        if constexpr (B == 1) {
            S.W(WK_PEL + 0) = me.acc.a.x; S.W(WK_PEL + 1) = me.acc.a.y; S.W(WK_PEL + 2) = me.acc.a.z;
            S.W(WK_PEL + 3) = me.acc.l.x; S.W(WK_PEL + 4) = me.acc.l.y; S.W(WK_PEL + 5) = me.acc.l.z;
            S.W(WK_PEL + 6) = me.vel.a.x; S.W(WK_PEL + 7) = me.vel.a.y; S.W(WK_PEL + 8) = me.vel.a.z;
            S.W(WK_PEL + 9) = me.vel.l.x; S.W(WK_PEL + 10) = me.vel.l.y; S.W(WK_PEL + 11) = me.vel.l.z;
            sfor<0, 9>([&](auto K) { S.W(WK_PEL + 12 + K) = me.mat.m[K]; });
        }
    }
    // subtree
    sfor<0, ct_body_nchild[B]>([&](auto Cn) {
        constexpr int ch = ct_body_child[4 * B + Cn];
        const Acc sub = visit<QPOS0, ch>(S, qp, me, o, pc);
        acc.crb.m += sub.crb.m; acc.crb.h = acc.crb.h + sub.crb.h;
        sfor<0, 6>([&](auto I) { acc.crb.I[I] += sub.crb.I[I]; });
        if constexpr (!QPOS0) acc.frc = acc.frc + sub.frc;
        __builtin_amdgcn_sched_barrier(0);     // keep sibling subtrees from being interleaved (register pressure)
    });
    // subtree complete: mass-matrix rows (CRBA) and bias of this body's dofs
    sfor<0, nd>([&](auto K) {
        constexpr int i = d0 + K, dep = ct_dof_depth[i];
        const SV f = imul(acc.crb, pc[dep - 1]);
        sfor<0, dep>([&](auto A) {
            constexpr int a = A;
            float v = sdot(pc[dep - 1 - a], f);
            if constexpr (a == 0) v += ct_dof_armature[i];
            S.W(WK_M + ct_dof_madr[i] + a) = v;
        });
        if constexpr (!QPOS0) {
            constexpr int j = ct_dof_jnt[i];
            float fs = -S(F_DAMP + i) * S(F_QVEL + i) - sdot(pc[dep - 1], acc.frc);
            if constexpr (ct_jnt_type[j] != 2 && ct_jnt_stiffness[j] != 0.f) fs -= ct_jnt_stiffness[j] * qp(ct_jnt_qposadr[j]);
            S.W(WK_SMOOTH + i) = fs;
        }
    });
    return acc;
}

// ---------------------------------------------------------------------------------------------- phase C helpers
// translational Jacobian of point p of body B (sign-weighted) into local-column rows, motion axes from the workspace
template <int B>
__device__ __forceinline__ void jac_point3(const St& S, V3 r, float sign, float (&Jx)[19], float (&Jy)[19], float (&Jz)[19]) {
    constexpr int last = ct_body_lastdof[B];
    sfor<0, ct_dof_depth[last]>([&](auto A) {
        constexpr int d = ct_dof_anc[16 * last + A], c = d2c(d);
        if constexpr (d < 3) { (d == 0 ? Jx[c] : d == 1 ? Jy[c] : Jz[c]) += sign; }            // pelvis slides: world axes
        else {
            const SV cd = ldcdof(S, d);
            const V3 v = cd.l + cross(cd.a, r);
            Jx[c] += sign * v.x; Jy[c] += sign * v.y; Jz[c] += sign * v.z;
        }
    });
}

template <int LEG> struct LegVec { float qv[19], qw[19], qs[19]; };

template <int LEG, class SET>
__device__ __forceinline__ void rawdots(const LegVec<LEG>& lv, const float (&J)[19], float& vel, float& ju, float& jw) {
    vel = ju = jw = 0.f;
    sfor<0, 19>([&](auto C) { constexpr int c = C; if constexpr (has<SET>(c)) { vel += J[c] * lv.qv[c]; ju += J[c] * lv.qs[c]; jw += J[c] * lv.qw[c]; } });
}
// per-leg view of the factor: only the entries of the pelvis + leg-LEG dofs are ever loaded / referenced (164 of 307)
template <int LEG> struct LegLD { float LD[NM]; float disqrt[NV]; };
template <int LEG>
__device__ __forceinline__ void load_leg_factor(const St& S, LegLD<LEG>& f) {
    sfor<0, 19>([&](auto C) {
        constexpr int i = c2d<LEG>(C);
        sfor<0, ct_dof_depth[i]>([&](auto A) { constexpr int k = ct_dof_madr[i] + A; f.LD[k] = S.W(WK_LD + k); });
        f.disqrt[i] = S.W(WK_DISQ + i);
    });
}
template <int LEG, class SET>
__device__ __forceinline__ void whiten3(const LegLD<LEG>& w, float (&J)[19]) {
    srfor<0, 19>([&](auto C) {
        constexpr int c = C;
        if constexpr (has<SET>(c)) {
            constexpr int i = c2d<LEG>(c);
            sfor<1, ct_dof_depth[i]>([&](auto A) { constexpr int a = A; J[d2c(ct_dof_anc[16 * i + a])] -= w.LD[ct_dof_madr[i] + a] * J[c]; });
        }
    });
    sfor<0, 19>([&](auto C) { constexpr int c = C; if constexpr (has<SET>(c)) J[c] *= w.disqrt[c2d<LEG>(c)]; });
}

template <int LEG, class SET, int NCH>
__device__ __forceinline__ void commit3(Fw3& w, const LegLD<LEG>& lf, const Lds& L, int chunk0, float (&y)[19], const LegVec<LEG>& lv, bool unilateral,
                                        float pos, float imp_pos, float diag, float timeconst, float& cost) {
    float vel, ju, jw;
    rawdots<LEG, SET>(lv, y, vel, ju, jw);
    whiten3<LEG, SET>(lf, y);
    float nn = 0.f;
    sfor<0, 19>([&](auto C) { constexpr int c = C; if constexpr (has<SET>(c)) nn += y[c] * y[c]; });
    const RowK kb = solref(timeconst);
    const float imp = impedance(imp_pos);
    const float R = fmaxf(MINVAL, (1.f - imp) / imp * diag);
    const float aref = -kb.B * vel - kb.K * imp * pos;
    const float b = ju - aref;
    float f = -(jw - aref) / R;                      // warm start from the previous qacc (mj_constraintUpdate)
    if (unilateral && f < 0.f) f = 0.f;
    const float invA = 1.f / (nn + R);
    float vals[20];
    sfor<0, 20>([&](auto I) {
        constexpr int i = I;
        if constexpr (i < SET::N) { if constexpr (SET::c[i] >= 0) vals[i] = y[SET::c[i]]; else vals[i] = 0.f; } else vals[i] = 0.f;
    });
    sfor<0, NCH>([&](auto C) { constexpr int c = C; L.wr(chunk0 + c, make_float4(vals[4 * c], vals[4 * c + 1], vals[4 * c + 2], vals[4 * c + 3])); });
    L.wr(chunk0 + NCH, make_float4(b, R, invA, f));
    sfor<0, 19>([&](auto C) { constexpr int c = C; if constexpr (has<SET>(c)) w.zt[c2d<LEG>(c)] += y[c] * f; });
    cost += f * (0.5f * R * f + b);
    __builtin_amdgcn_sched_barrier(0);
}

template <int LEG>
__device__ __forceinline__ void build_rows3(const St& S, Fw3& w, const Lds& L, const Dyn2& dy, V3 o, float& cost) {
    LegVec<LEG> lv;
    sfor<0, 19>([&](auto C) { constexpr int d = c2d<LEG>(C); lv.qv[C] = S(F_QVEL + d); lv.qw[C] = S(F_QACCW + d); lv.qs[C] = S.W(WK_QS + d); });
    LegLD<LEG> lf;
    load_leg_factor<LEG>(S, lf);
    __builtin_amdgcn_sched_barrier(0);
    constexpr int base = WK_PTS + 30 * LEG;
    // ---- 2 connect equalities (cassie.xml:225-230)
    sfor<0, 2>([&](auto E) {
        constexpr int e = 2 * LEG + E, b1 = ct_eq_body1[e], b2 = ct_eq_body2[e];
        const V3 p1 = ldv3<base + 6 * E>(S), p2 = ldv3<base + 6 * E + 3>(S);
        const V3 c = p1 - p2;
        float J[3][19];
        sfor<0, 19>([&](auto K) { J[0][K] = 0.f; J[1][K] = 0.f; J[2][K] = 0.f; });
        jac_point3<b1>(S, p1 - o, 1.f, J[0], J[1], J[2]);
        jac_point3<b2>(S, p2 - o, -1.f, J[0], J[1], J[2]);
        const float cp[3] = {c.x, c.y, c.z};
        const float tran = S(F_BIW + b1) + S(F_BIW + b2), cn = sqrtf(dot(c, c));
        sfor<0, 3>([&](auto K) {
            constexpr int k = K, row = LEG * 6 + E * 3 + k;
            if constexpr (E == 0) commit3<LEG, SetPL, 4>(w, lf, L, CH_EQ + 5 * row, J[k], lv, false, cp[k], cn, tran, 0.005f, cost);
            else commit3<LEG, SetAC, 4>(w, lf, L, CH_EQ + 5 * row, J[k], lv, false, cp[k], cn, tran, 0.005f, cost);
        });
    });
    // ---- first active joint limit of this leg
    w.nlim[LEG] = 0;
    sfor<0, NJ>([&](auto Jn) {
        constexpr int j = Jn;
        if constexpr (ct_jnt_limited[j] && ((ct_jnt_body[j] >= 14) == (LEG == 1)) && ct_jnt_body[j] >= 2) {
            const float q = S(F_QPOS + ct_jnt_qposadr[j]);
            const float dlo = q - ct_jnt_range[2 * j], dhi = ct_jnt_range[2 * j + 1] - q;
            if ((dlo < 0.f || dhi < 0.f) && w.nlim[LEG] == 0) {
                constexpr int d = ct_jnt_dofadr[j];
                const float dist = dlo < 0.f ? dlo : dhi;
                float Jl[19];
                sfor<0, 19>([&](auto K) { Jl[K] = 0.f; });
                Jl[d2c(d)] = dlo < 0.f ? 1.f : -1.f;
                commit3<LEG, SetALL, 5>(w, lf, L, CH_LIM + 6 * LEG, Jl, lv, true, dist, dist, S(F_DIW + d), 0.02f, cost);
                w.nlim[LEG] = 1;
            }
        }
    });
    // ---- contacts: first 3 penetrating capsule ends in the order foot e0,e1, tarsus e0,e1, shin e0,e1.
    // One code body per SLOT: the Jacobian is taken along the pelvis->foot chain with the dofs below the touching
    // body masked out (tarsus contact: no foot dof; shin contact: no tarsus / foot dof).
    const V3 p0 = {ct_floor_pos[0], ct_floor_pos[1], ct_floor_pos[2]};
    int nc = 0;
    V3 cpt[MAXC]; float cdist[MAXC]; int cgeo[MAXC];
    sfor<0, 6>([&](auto I) {
        constexpr int G = I / 2;
        const V3 ctr = ldv3<base + 12 + 3 * I>(S);
        const float dist = dot(ctr - p0, dy.fn) - ct_geom_radius[2 * G + LEG];
        const bool hit = dist < 0.f && nc < MAXC;
        const V3 cp = ctr - dy.fn * (ct_geom_radius[2 * G + LEG] + 0.5f * dist);
        sfor<0, MAXC>([&](auto Sl) { if (hit && nc == Sl) { cpt[Sl] = cp; cdist[Sl] = dist; cgeo[Sl] = G; } });
        nc += hit ? 1 : 0;
    });
    w.ncon[LEG] = nc;
    sfor<0, MAXC>([&](auto Sl) {
        constexpr int s = Sl, slot = 3 * LEG + s;
        if (s < nc) {
            const int G = cgeo[s];
            const V3 r = cpt[s] - o;
            const float dist = cdist[s];
            float Jx[19], Jy[19], Jz[19];
            sfor<0, 19>([&](auto K) { Jx[K] = 0.f; Jy[K] = 0.f; Jz[K] = 0.f; });
            jac_point3<13 + 12 * LEG>(S, r, 1.f, Jx, Jy, Jz);
            if (G >= 1) { Jx[18] = 0.f; Jy[18] = 0.f; Jz[18] = 0.f; }                     // tarsus / shin: foot dof does not move the point
            if (G >= 2) { Jx[14] = 0.f; Jy[14] = 0.f; Jz[14] = 0.f; }                     // shin: nor does the tarsus dof
            float yn[19], y1[19], y2[19];
            sfor<0, 13>([&](auto I) {
                constexpr int k = SetFT::c[I];
                yn[k] = dy.fn.x * Jx[k] + dy.fn.y * Jy[k] + dy.fn.z * Jz[k];
                y1[k] = dy.ft1.x * Jx[k] + dy.ft1.y * Jy[k] + dy.ft1.z * Jz[k];
                y2[k] = dy.ft2.x * Jx[k] + dy.ft2.y * Jy[k] + dy.ft2.z * Jz[k];
            });
            float vn, un, wn, v1, u1, w1, v2, u2, w2;
            rawdots<LEG, SetFT>(lv, yn, vn, un, wn); rawdots<LEG, SetFT>(lv, y1, v1, u1, w1); rawdots<LEG, SetFT>(lv, y2, v2, u2, w2);
            whiten3<LEG, SetFT>(lf, yn); whiten3<LEG, SetFT>(lf, y1); whiten3<LEG, SetFT>(lf, y2);
            float gnn = 0.f, g11 = 0.f, g22 = 0.f, gn1 = 0.f, gn2 = 0.f, g12 = 0.f;
            sfor<0, 13>([&](auto I) {
                constexpr int k = SetFT::c[I];
                gnn += yn[k] * yn[k]; g11 += y1[k] * y1[k]; g22 += y2[k] * y2[k]; gn1 += yn[k] * y1[k]; gn2 += yn[k] * y2[k]; g12 += y1[k] * y2[k];
            });
            const float mu = dy.friction;
            const float tran = G == 0 ? S(F_BIW + 13 + 12 * LEG) : G == 1 ? S(F_BIW + 9 + 12 * LEG) : S(F_BIW + 8 + 12 * LEG);
            const RowK kb = solref(0.005f);
            const float imp = impedance(dist);
            const float R1 = fmaxf(MINVAL, (1.f - imp) / imp * (tran + mu * mu * tran));
            const float Rpy = fmaxf(MINVAL, 2.f * mu * mu * R1);       // pyramidal regulariser, impratio 1
            const float sv[4] = {mu * v1, -mu * v1, mu * v2, -mu * v2}, su[4] = {mu * u1, -mu * u1, mu * u2, -mu * u2};
            const float sw[4] = {mu * w1, -mu * w1, mu * w2, -mu * w2};
            float bk[4], fk[4];
            sfor<0, 4>([&](auto K) {
                constexpr int k = K;
                const float aref = -kb.B * (vn + sv[k]) - kb.K * imp * dist;
                bk[k] = un + su[k] - aref;
                const float f = -((wn + sw[k]) - aref) / Rpy;
                fk[k] = f < 0.f ? 0.f : f;
                cost += fk[k] * (0.5f * Rpy * fk[k] + bk[k]);
            });
            const float dn = fk[0] + fk[1] + fk[2] + fk[3], d1 = mu * (fk[0] - fk[1]), d2 = mu * (fk[2] - fk[3]);
            sfor<0, 13>([&](auto I) { constexpr int k = SetFT::c[I]; w.zt[c2d<LEG>(k)] += yn[k] * dn + y1[k] * d1 + y2[k] * d2; });
            float vals[40];
            sfor<0, 13>([&](auto I) { constexpr int k = SetFT::c[I]; vals[I] = yn[k]; vals[13 + I] = y1[k]; vals[26 + I] = y2[k]; });
            vals[39] = 0.f;
            constexpr int ch = CH_CON + 14 * slot;
            sfor<0, 10>([&](auto C) { constexpr int cc = C; L.wr(ch + cc, make_float4(vals[4 * cc], vals[4 * cc + 1], vals[4 * cc + 2], vals[4 * cc + 3])); });
            L.wr(ch + 10, make_float4(gnn, gn1, gn2, g11));
            L.wr(ch + 11, make_float4(g12, g22, Rpy, 0.f));
            L.wr(ch + 12, make_float4(bk[0], bk[1], bk[2], bk[3]));
            L.wr(ch + 13, make_float4(fk[0], fk[1], fk[2], fk[3]));
            if (G == 0) w.footmask |= 1u << slot;
            __builtin_amdgcn_sched_barrier(0);
        }
    });
}

// PGS sweep of one leg on the Fw3 register set (same arithmetic as c2::pgs_leg)
struct LdsV {       // same store, reads the compiler must re-issue every sweep (streamed, 1 instruction per 4 floats)
    float4* base;
    typedef float f4v __attribute__((ext_vector_type(4)));
    __device__ __forceinline__ float4 rd(int c) const { const f4v v = *(volatile f4v*)(base + c * EPW); return make_float4(v.x, v.y, v.z, v.w); }
    __device__ __forceinline__ void wr(int c, float4 v) const { base[c * EPW] = v; }
};
template <int LEG>
__device__ __forceinline__ void pgs_leg3(Fw3& w, const Lds& L, float mu) {
    sfor<0, 6>([&](auto Rw) {
        constexpr int row = LEG * 6 + Rw, ch = CH_EQ + 5 * row;
        float y[16];
        sfor<0, 4>([&](auto C) { const float4 v = L.rd(ch + C); y[4 * C] = v.x; y[4 * C + 1] = v.y; y[4 * C + 2] = v.z; y[4 * C + 3] = v.w; });
        const float4 m = L.rd(ch + 4);          // b R invA f
        float r4[4] = {m.x + m.y * m.w, 0.f, 0.f, 0.f};       // 4 partial sums: break the 16-deep dependent FMA chain
        sfor<0, 16>([&](auto I) { constexpr int i = I, c = Rw < 3 ? SetPL::c[i] : SetAC::c[i]; if constexpr (c >= 0) r4[i & 3] += y[i] * w.zt[c2d<LEG>(c)]; });
        const float res = (r4[0] + r4[1]) + (r4[2] + r4[3]);
        const float df = -res * m.z;
        sfor<0, 16>([&](auto I) { constexpr int i = I, c = Rw < 3 ? SetPL::c[i] : SetAC::c[i]; if constexpr (c >= 0) w.zt[c2d<LEG>(c)] += y[i] * df; });
        L.wr(ch + 4, make_float4(m.x, m.y, m.z, m.w + df));
    });
    if (w.nlim[LEG]) {
        constexpr int ch = CH_LIM + 6 * LEG;
        float y[20];
        sfor<0, 5>([&](auto C) { const float4 v = L.rd(ch + C); y[4 * C] = v.x; y[4 * C + 1] = v.y; y[4 * C + 2] = v.z; y[4 * C + 3] = v.w; });
        const float4 m = L.rd(ch + 5);
        float res = m.x + m.y * m.w;
        sfor<0, 19>([&](auto I) { res += y[I] * w.zt[c2d<LEG>(I)]; });
        float fn = m.w - res * m.z;
        fn = fn < 0.f ? 0.f : fn;
        const float df = fn - m.w;
        sfor<0, 19>([&](auto I) { w.zt[c2d<LEG>(I)] += y[I] * df; });
        L.wr(ch + 5, make_float4(m.x, m.y, m.z, fn));
    }
    sfor<0, MAXC>([&](auto Sl) {
        constexpr int s = Sl;
        if (s >= w.ncon[LEG]) return;
        constexpr int ch = CH_CON + 14 * (3 * LEG + s);
        float v[40];
        sfor<0, 10>([&](auto C) { const float4 q = L.rd(ch + C); v[4 * C] = q.x; v[4 * C + 1] = q.y; v[4 * C + 2] = q.z; v[4 * C + 3] = q.w; });
        const float4 g0 = L.rd(ch + 10), g1 = L.rd(ch + 11), bb = L.rd(ch + 12), ff = L.rd(ch + 13);
        const float gnn = g0.x, gn1 = g0.y, gn2 = g0.z, g11 = g0.w, g12 = g1.x, g22 = g1.y, R = g1.z;
        float dn2[2] = {0.f, 0.f}, d12[2] = {0.f, 0.f}, d22[2] = {0.f, 0.f};
        sfor<0, 13>([&](auto I) { constexpr int d = c2d<LEG>(SetFT::c[I]); dn2[I & 1] += v[I] * w.zt[d]; d12[I & 1] += v[13 + I] * w.zt[d]; d22[I & 1] += v[26 + I] * w.zt[d]; });
        const float dn = dn2[0] + dn2[1], d1 = d12[0] + d12[1], d2 = d22[0] + d22[1];
        float f[4] = {ff.x, ff.y, ff.z, ff.w};
        const float b[4] = {bb.x, bb.y, bb.z, bb.w};
        float sdn = 0.f, sd1 = 0.f, sd2 = 0.f;
        sfor<0, 4>([&](auto K) {
            constexpr int k = K;
            constexpr float sg = (k & 1) ? -1.f : 1.f;
            const float sm = sg * mu;
            const float gnj = k < 2 ? gn1 : gn2, gjj = k < 2 ? g11 : g22;
            const float ykz = (dn + sdn * gnn + sd1 * gn1 + sd2 * gn2) +
                              sm * (k < 2 ? (d1 + sdn * gn1 + sd1 * g11 + sd2 * g12) : (d2 + sdn * gn2 + sd1 * g12 + sd2 * g22));
            const float A = gnn + 2.f * sm * gnj + mu * mu * gjj + R;
            const float res = b[k] + R * f[k] + ykz;
            float fn = f[k] - res * __frcp_rn(A);
            fn = fn < 0.f ? 0.f : fn;
            const float df = fn - f[k];
            f[k] = fn;
            sdn += df; if constexpr (k < 2) sd1 += sm * df; else sd2 += sm * df;
        });
        sfor<0, 13>([&](auto I) { constexpr int d = c2d<LEG>(SetFT::c[I]); w.zt[d] += v[I] * sdn + v[13 + I] * sd1 + v[26 + I] * sd2; });
        L.wr(ch + 13, make_float4(f[0], f[1], f[2], f[3]));
    });
}
template <int LEG>
__device__ __forceinline__ void zero_forces3(const Fw3& w, const Lds& L) {
    sfor<0, 6>([&](auto Rw) { constexpr int ch = CH_EQ + 5 * (LEG * 6 + Rw) + 4; float4 m = L.rd(ch); m.w = 0.f; L.wr(ch, m); });
    if (w.nlim[LEG]) { constexpr int ch = CH_LIM + 6 * LEG + 5; float4 m = L.rd(ch); m.w = 0.f; L.wr(ch, m); }
    sfor<0, MAXC>([&](auto Sl) { if (Sl < w.ncon[LEG]) L.wr(CH_CON + 14 * (3 * LEG + Sl) + 13, make_float4(0.f, 0.f, 0.f, 0.f)); });
}

// ---------------------------------------------------------------------------------------------- the substep, staged
// Each stage is a separate (non-inlined) function so that the register allocator sees one phase at a time: the
// 307-entry factor lives in VGPRs+AGPRs inside a stage and crosses stage boundaries through the workspace column.

// stage A: tree walk (phase A) + actuation.  ctrl = actuator-side torques.
__device__ __forceinline__ void stage_tree(const St& S, const float (&ctrl)[10]) {
    SV pc[14];
    Node world{};
    const V3 o = {S(F_QPOS), S(F_QPOS + 1), S(F_QPOS + 2)};
    (void)visit<false, 1>(S, [&](int i) { return S(F_QPOS + i); }, world, o, pc);
    sfor<0, NU>([&](auto U) {
        constexpr int u = U;
        const float c = fminf(fmaxf(ctrl[u], -ct_act_ctrlmax[u]), ct_act_ctrlmax[u]);
        S.W(WK_SMOOTH + ct_act_dof[u]) += ct_act_gear[u] * c;
    });
}

// stage B: factorisation + qacc_smooth; the factor goes to the workspace (each leg reloads only its 164 entries)
__device__ __forceinline__ void stage_factor(const St& S) {
    float LD[NM], dsq[NV], disq[NV];
    sfor<0, NM>([&](auto I) { LD[I] = S.W(WK_M + I); });
    factor<true>(LD, dsq, disq);
    float x[NV];
    sfor<0, NV>([&](auto D) { x[D] = S.W(WK_SMOOTH + D); });
    solve_LT(LD, x);
    sfor<0, NV>([&](auto D) { x[D] *= disq[D] * disq[D]; });
    solve_L(LD, x);
    sfor<0, NV>([&](auto D) { S.W(WK_QS + D) = x[D]; S.W(WK_DISQ + D) = disq[D]; });
    sfor<0, NM>([&](auto I) { S.W(WK_LD + I) = LD[I]; });
}

// stage C: constraint rows of one leg (whitened into LDS) + warm start contributions; z~ / cost accumulate in the workspace
template <int LEG>
__device__ __forceinline__ void stage_rows_leg(const St& S, const Lds& L) {
    Fw3 w;
    const V3 o = {S(F_QPOS), S(F_QPOS + 1), S(F_QPOS + 2)};
    Dyn2 dy;
    dy.friction = S(F_FRIC);
    dy.fn = {S(F_FLOOR), S(F_FLOOR + 1), S(F_FLOOR + 2)};
    dy.ft1 = {S(F_FLOOR + 3), S(F_FLOOR + 4), S(F_FLOOR + 5)};
    dy.ft2 = {S(F_FLOOR + 6), S(F_FLOOR + 7), S(F_FLOOR + 8)};
    sfor<0, NV>([&](auto D) { w.zt[D] = 0.f; });
    w.footmask = 0u;
    float cost = 0.f;
    build_rows3<LEG>(S, w, L, dy, o, cost);
    // the two legs may run concurrently on two waves: each writes its own slots, stage_warm_check sums them
    sfor<0, 6>([&](auto C) { S.W(WK_ZP2 + 6 * LEG + C) = w.zt[C]; });
    sfor<6, 19>([&](auto C) { constexpr int d = c2d<LEG>(C); S.W(WK_ZT + d) = w.zt[d]; });
    S.W(WK_MISC + LEG) = (float)w.ncon[LEG]; S.W(WK_MISC + 2 + LEG) = (float)w.nlim[LEG];
    S.W(WK_MISC + 4 + LEG) = (float)w.footmask; S.W(WK_MISC + 6 + LEG) = cost;
}

// warm start loses to f = 0 (mj_fwdConstraint) when its dual cost is positive
__device__ __forceinline__ void stage_warm_check(const St& S, const Lds& L) {
    float cost = S.W(WK_MISC + 6) + S.W(WK_MISC + 7);
    sfor<0, 6>([&](auto C) { S.W(WK_ZT + C) = S.W(WK_ZP2 + C) + S.W(WK_ZP2 + 6 + C); });
    S.W(WK_MISC + 4) = (float)((unsigned)S.W(WK_MISC + 4) | (unsigned)S.W(WK_MISC + 5));     // merged foot mask
    sfor<0, NV>([&](auto D) { const float z = S.W(WK_ZT + D); cost += 0.5f * z * z; });
    if (cost > 0.f) {
        Fw3 w;
        w.ncon[0] = (int)S.W(WK_MISC + 0); w.ncon[1] = (int)S.W(WK_MISC + 1);
        w.nlim[0] = (int)S.W(WK_MISC + 2); w.nlim[1] = (int)S.W(WK_MISC + 3);
        zero_forces3<0>(w, L); zero_forces3<1>(w, L);
        sfor<0, NV>([&](auto D) { S.W(WK_ZT + D) = 0.f; });
    }
}

// stage D: projected Gauss-Seidel; only z~ (32 registers) + one row at a time are live
__device__ __forceinline__ void stage_pgs(const St& S, const Lds& L, int pgs_iters) {
    stage_warm_check(S, L);
    Fw3 w;     // only zt / ncon / nlim are touched
    sfor<0, NV>([&](auto D) { w.zt[D] = S.W(WK_ZT + D); });
    w.ncon[0] = (int)S.W(WK_MISC + 0); w.ncon[1] = (int)S.W(WK_MISC + 1);
    w.nlim[0] = (int)S.W(WK_MISC + 2); w.nlim[1] = (int)S.W(WK_MISC + 3);
    const float mu = S(F_FRIC);
    for (int it = 0; it < pgs_iters; ++it) {
        // the rows are loop-invariant; without this the compiler hoists ~400 row values out of the loop and then
        // spills them.  An opaque (always zero) chunk offset keeps them streaming from LDS every sweep instead; it is an
        // integer so that the pointer keeps its LDS address space (ds_read_b128, not flat_load).
        int zero_off = 0;
        asm volatile("" : "+v"(zero_off));
        const Lds Ls{L.base + zero_off};
        pgs_leg3<0>(w, Ls, mu);
        pgs_leg3<1>(w, Ls, mu);
    }
    sfor<0, NV>([&](auto D) { S.W(WK_ZT + D) = w.zt[D]; });
}

// stage D, dual-wave form: the workgroup has two waves working on the SAME 64 envs; wave LEG owns the rows of leg LEG and
// keeps them in registers for all sweeps (no LDS re-reads).  Gauss-Seidel order is unchanged (left rows, then right rows):
// the waves alternate, handing the 6 pelvis components of z~ over through LDS at a workgroup barrier.
constexpr int CH_X = 156;        // 2 chunks: z~ pelvis hand-off
template <int LEG>
__device__ __forceinline__ void stage_pgs_wave(const St& S, const Lds& L, int pgs_iters) {
    float z[19];                 // z~ in local columns: pelvis 6 | own leg 13
    sfor<0, 19>([&](auto C) { z[C] = S.W(WK_ZT + c2d<LEG>(C)); });
    const int ncon = (int)S.W(WK_MISC + LEG), nlim = (int)S.W(WK_MISC + 2 + LEG);
    const float mu = S(F_FRIC);
    // ---- rows -> registers
    float y[6][16], eb[6], eR[6], eiA[6], ef[6];
    sfor<0, 6>([&](auto Rw) {
        constexpr int ch = CH_EQ + 5 * (LEG * 6 + Rw);
        sfor<0, 4>([&](auto C) { const float4 v = L.rd(ch + C); y[Rw][4 * C] = v.x; y[Rw][4 * C + 1] = v.y; y[Rw][4 * C + 2] = v.z; y[Rw][4 * C + 3] = v.w; });
        const float4 m = L.rd(ch + 4);
        eb[Rw] = m.x; eR[Rw] = m.y; eiA[Rw] = m.z; ef[Rw] = m.w;
    });
    float cv[MAXC][40], cG[MAXC][6], cR[MAXC], cb[MAXC][4], cf[MAXC][4], ciA[MAXC][4];
    sfor<0, MAXC>([&](auto Sl) {
        constexpr int ch = CH_CON + 14 * (3 * LEG + Sl);
        if (Sl < ncon) {
            sfor<0, 10>([&](auto C) { const float4 q = L.rd(ch + C); cv[Sl][4 * C] = q.x; cv[Sl][4 * C + 1] = q.y; cv[Sl][4 * C + 2] = q.z; cv[Sl][4 * C + 3] = q.w; });
            const float4 g0 = L.rd(ch + 10), g1 = L.rd(ch + 11), bb = L.rd(ch + 12), ff = L.rd(ch + 13);
            cG[Sl][0] = g0.x; cG[Sl][1] = g0.y; cG[Sl][2] = g0.z; cG[Sl][3] = g0.w; cG[Sl][4] = g1.x; cG[Sl][5] = g1.y; cR[Sl] = g1.z;
            cb[Sl][0] = bb.x; cb[Sl][1] = bb.y; cb[Sl][2] = bb.z; cb[Sl][3] = bb.w;
            cf[Sl][0] = ff.x; cf[Sl][1] = ff.y; cf[Sl][2] = ff.z; cf[Sl][3] = ff.w;
            sfor<0, 4>([&](auto K) {      // A_kk + R of the pyramid rows n + s mu t_j is sweep-invariant
                constexpr int k = K;
                const float sm = ((k & 1) ? -mu : mu), gnj = k < 2 ? cG[Sl][1] : cG[Sl][2], gjj = k < 2 ? cG[Sl][3] : cG[Sl][5];
                ciA[Sl][k] = __frcp_rn(cG[Sl][0] + 2.f * sm * gnj + mu * mu * gjj + cR[Sl]);
            });
        }
    });
    for (int it = 0; it < pgs_iters; ++it) {
        if (LEG == 1) {          // wait for the left sweep, take over z~ pelvis
            __syncthreads();
            const float4 a = L.rd(CH_X), b = L.rd(CH_X + 1);
            z[0] = a.x; z[1] = a.y; z[2] = a.z; z[3] = a.w; z[4] = b.x; z[5] = b.y;
        }
        // equalities: 6 bilateral rows
        sfor<0, 6>([&](auto Rw) {
            float r4[4] = {eb[Rw] + eR[Rw] * ef[Rw], 0.f, 0.f, 0.f};
            sfor<0, 16>([&](auto I) { constexpr int i = I, c = Rw < 3 ? SetPL::c[i] : SetAC::c[i]; if constexpr (c >= 0) r4[i & 3] += y[Rw][i] * z[c]; });
            const float df = -((r4[0] + r4[1]) + (r4[2] + r4[3])) * eiA[Rw];
            sfor<0, 16>([&](auto I) { constexpr int i = I, c = Rw < 3 ? SetPL::c[i] : SetAC::c[i]; if constexpr (c >= 0) z[c] += y[Rw][i] * df; });
            ef[Rw] += df;
        });
        if (nlim) {              // rare: the limit row streams from LDS
            constexpr int ch = CH_LIM + 6 * LEG;
            float yl[20];
            sfor<0, 5>([&](auto C) { const float4 v = L.rd(ch + C); yl[4 * C] = v.x; yl[4 * C + 1] = v.y; yl[4 * C + 2] = v.z; yl[4 * C + 3] = v.w; });
            const float4 m = L.rd(ch + 5);
            float res = m.x + m.y * m.w;
            sfor<0, 19>([&](auto I) { res += yl[I] * z[I]; });
            float fn = m.w - res * m.z;
            fn = fn < 0.f ? 0.f : fn;
            const float df = fn - m.w;
            sfor<0, 19>([&](auto I) { z[I] += yl[I] * df; });
            L.wr(ch + 5, make_float4(m.x, m.y, m.z, fn));
        }
        // contacts: 4 pyramid rows each, swept through running basis residuals (rn, r1, r2) = (n, t1, t2) . z~
        sfor<0, MAXC>([&](auto Sl) {
            if (Sl < ncon) {
                float rn2[2] = {0.f, 0.f}, r12[2] = {0.f, 0.f}, r22[2] = {0.f, 0.f};
                sfor<0, 13>([&](auto I) { constexpr int c = SetFT::c[I]; rn2[I & 1] += cv[Sl][I] * z[c]; r12[I & 1] += cv[Sl][13 + I] * z[c]; r22[I & 1] += cv[Sl][26 + I] * z[c]; });
                float rn = rn2[0] + rn2[1], r1 = r12[0] + r12[1], r2 = r22[0] + r22[1];
                const float gnn = cG[Sl][0], gn1 = cG[Sl][1], gn2 = cG[Sl][2], g11 = cG[Sl][3], g12 = cG[Sl][4], g22 = cG[Sl][5];
                float sdn = 0.f, sd1 = 0.f, sd2 = 0.f;
                sfor<0, 4>([&](auto K) {
                    constexpr int k = K;
                    const float sm = (k & 1) ? -mu : mu;
                    const float res = cb[Sl][k] + cR[Sl] * cf[Sl][k] + rn + sm * (k < 2 ? r1 : r2);
                    float fn = cf[Sl][k] - res * ciA[Sl][k];
                    fn = fn < 0.f ? 0.f : fn;
                    const float df = fn - cf[Sl][k];
                    cf[Sl][k] = fn;
                    // y_k = n + sm t_j moves the basis residuals by df * (G n-col + sm G j-col)
                    if constexpr (k < 2) { rn += df * (gnn + sm * gn1); r1 += df * (gn1 + sm * g11); r2 += df * (gn2 + sm * g12); sd1 += sm * df; }
                    else { rn += df * (gnn + sm * gn2); r1 += df * (gn1 + sm * g12); r2 += df * (gn2 + sm * g22); sd2 += sm * df; }
                    sdn += df;
                });
                sfor<0, 13>([&](auto I) { constexpr int c = SetFT::c[I]; z[c] += cv[Sl][I] * sdn + cv[Sl][13 + I] * sd1 + cv[Sl][26 + I] * sd2; });
            }
        });
        // hand z~ pelvis to the other wave
        L.wr(CH_X, make_float4(z[0], z[1], z[2], z[3])); L.wr(CH_X + 1, make_float4(z[4], z[5], 0.f, 0.f));
        __syncthreads();
        if (LEG == 0) {          // wait for the right sweep
            __syncthreads();
            const float4 a = L.rd(CH_X), b = L.rd(CH_X + 1);
            z[0] = a.x; z[1] = a.y; z[2] = a.z; z[3] = a.w; z[4] = b.x; z[5] = b.y;
        }
    }
    // results: forces back to LDS (foot-force readout), z~ to the workspace (pelvis part: both waves hold the final value)
    sfor<0, 6>([&](auto Rw) { L.wr(CH_EQ + 5 * (LEG * 6 + Rw) + 4, make_float4(eb[Rw], eR[Rw], eiA[Rw], ef[Rw])); });
    sfor<0, MAXC>([&](auto Sl) { if (Sl < ncon) L.wr(CH_CON + 14 * (3 * LEG + Sl) + 13, make_float4(cf[Sl][0], cf[Sl][1], cf[Sl][2], cf[Sl][3])); });
    sfor<0, 19>([&](auto C) { if (LEG == 1 || C >= 6) S.W(WK_ZT + c2d<LEG>(C)) = z[C]; });
}

// stage E: qacc, foot force, IMU; then (do_euler) mj_Euler with implicit joint damping:
// (M + h D) a = qfrc_smooth + J^T f = qfrc_smooth + L^T D^1/2 z~
__device__ __forceinline__ void stage_finish(const St& S, const Lds& L, bool do_euler, float (&acc_out)[3], float (&foot_fz)[2], const float* rows4 = nullptr) {
    float LD[NM], zt[NV], disq[NV], qacc[NV];
    sfor<0, NM>([&](auto I) { LD[I] = S.W(WK_LD + I); });
    sfor<0, NV>([&](auto D) { zt[D] = S.W(WK_ZT + D); disq[D] = S.W(WK_DISQ + D); });
    sfor<0, NV>([&](auto D) { qacc[D] = zt[D] * disq[D]; });
    solve_L(LD, qacc);
    sfor<0, NV>([&](auto D) { qacc[D] += S.W(WK_QS + D); });
    {   // world z of the contact force on the foot bodies (cassie_sim_foot_forces -> get_foot_forces()[2], [8])
        const int nc[2] = {(int)S.W(WK_MISC + 0), (int)S.W(WK_MISC + 1)};
        const float mu = S(F_FRIC), nz = S(F_FLOOR + 2), t1z = S(F_FLOOR + 5), t2z = S(F_FLOOR + 8);
        foot_fz[0] = foot_fz[1] = 0.f;
#if defined(APX_GEN) && APX_GEN == 4
        sfor<0, 2 * MAXC>([&](auto Sl) {      // generation-4 row store (cassie_lane.h): slot s at rows4[624 + 20 s], [7] foot flag, [12..15] f
            constexpr int sl = Sl, lg = sl / MAXC;
            const float* cr = rows4 + 624 + 20 * sl;
            if ((sl % MAXC) < nc[lg] && cr[7] != 0.f)
                foot_fz[lg] += nz * (cr[12] + cr[13] + cr[14] + cr[15]) + mu * (t1z * (cr[12] - cr[13]) + t2z * (cr[14] - cr[15]));
        });
#else
        const unsigned footmask = (unsigned)S.W(WK_MISC + 4);
        sfor<0, 6>([&](auto Sl) {
            constexpr int sl = Sl, lg = sl / 3;
            if ((sl % 3) < nc[lg] && ((footmask >> sl) & 1u)) {
                const float4 ff = L.rd(CH_CON + 14 * sl + 13);
                foot_fz[lg] += nz * (ff.x + ff.y + ff.z + ff.w) + mu * (t1z * (ff.x - ff.y) + t2z * (ff.z - ff.w));
            }
        });
#endif
    }
    {   // accelerometer at the imu site (cassie.xml:267): classical acceleration of the site point, site frame
        SV A = {{S.W(WK_PEL), S.W(WK_PEL + 1), S.W(WK_PEL + 2)}, {S.W(WK_PEL + 3), S.W(WK_PEL + 4), S.W(WK_PEL + 5)}};
        const SV V = {{S.W(WK_PEL + 6), S.W(WK_PEL + 7), S.W(WK_PEL + 8)}, {S.W(WK_PEL + 9), S.W(WK_PEL + 10), S.W(WK_PEL + 11)}};
        M3 R;
        sfor<0, 9>([&](auto K) { R.m[K] = S.W(WK_PEL + 12 + K); });
        A.l = A.l + V3{qacc[0], qacc[1], qacc[2]};
        sfor<0, 3>([&](auto K) { A.a = A.a + col(R, K) * qacc[3 + K]; });
        const V3 r = mul(R, V3{ct_imu_pos[0], ct_imu_pos[1], ct_imu_pos[2]});
        const V3 vp = V.l + cross(V.a, r);
        const V3 a = A.l + cross(A.a, r) + cross(V.a, vp);
        acc_out[0] = dot(col(R, 0), a); acc_out[1] = dot(col(R, 1), a); acc_out[2] = dot(col(R, 2), a);
    }
    if (!do_euler) return;
    float x[NV], rhs[NV];
    sfor<0, NV>([&](auto D) { x[D] = zt[D] * LD[ct_dof_madr[D]] * disq[D]; });     // D^1/2 = D * D^-1/2
    mul_LT(LD, x, rhs);
    sfor<0, NV>([&](auto D) { rhs[D] += S.W(WK_SMOOTH + D); });
    __builtin_amdgcn_sched_barrier(0);
    sfor<0, NM>([&](auto I) { LD[I] = S.W(WK_M + I); });
    sfor<0, NV>([&](auto D) { LD[ct_dof_madr[D]] += DT * S(F_DAMP + D); });
    float d1[NV], d2[NV];
    factor<false>(LD, d1, d2);
    solve_LT(LD, rhs);
    sfor<0, NV>([&](auto D) { rhs[D] *= __frcp_rn(LD[ct_dof_madr[D]]); });
    solve_L(LD, rhs);
    float qv[NV];
    sfor<0, NV>([&](auto D) { S(F_QACCW + D) = qacc[D]; qv[D] = S(F_QVEL + D) + DT * rhs[D]; S(F_QVEL + D) = qv[D]; });
    sfor<0, NJ>([&](auto Jn) {
        constexpr int j = Jn, qa = ct_jnt_qposadr[j], da = ct_jnt_dofadr[j];
        if constexpr (ct_jnt_type[j] != 2) S(F_QPOS + qa) += DT * qv[da];
        else {
            const V3 wv = {qv[da], qv[da + 1], qv[da + 2]};
            const float nw = sqrtf(dot(wv, wv));
            Q4 q = {S(F_QPOS + qa), S(F_QPOS + qa + 1), S(F_QPOS + qa + 2), S(F_QPOS + qa + 3)};
            if (nw > 0.f) {
                float sn, cs;
                __sincosf(0.5f * nw * DT, &sn, &cs);
                const float s = sn / nw;
                q = qmul(q, Q4{cs, wv.x * s, wv.y * s, wv.z * s});
            }
            q = qnormalize(q);
            S(F_QPOS + qa) = q.w; S(F_QPOS + qa + 1) = q.x; S(F_QPOS + qa + 2) = q.y; S(F_QPOS + qa + 3) = q.z;
        }
    });
}

// ---------------------------------------------------------------------------------------------- mj_setConst subset
// body_invweight0 (translational part) of the constraint bodies and dof_invweight0 of the limited joints, at qpos0, from
// the whitened rows: |y~|^2 = J M^-1 J^T.  Runs at every reset with dynamics randomisation (cassie.py:634-660).
__device__ __forceinline__ void setconst_tree(const St& S) {
    SV pc[14];
    Node world{};
    const V3 o = {ct_qpos0[0], ct_qpos0[1], ct_qpos0[2]};
    (void)visit<true, 1>(S, [&](int i) { return ct_qpos0[i]; }, world, o, pc);
}
__device__ __forceinline__ void setconst_factor(const St& S) {
    float LD[NM], dsq[NV], disq[NV];
    sfor<0, NM>([&](auto I) { LD[I] = S.W(WK_M + I); });
    factor<true>(LD, dsq, disq);
    sfor<0, NV>([&](auto D) { S.W(WK_DISQ + D) = disq[D]; });
    sfor<0, NM>([&](auto I) { S.W(WK_LD + I) = LD[I]; });
    S(F_BIW) = 0.f;
}
template <int LEG>
__device__ __forceinline__ void setconst_leg(const St& S) {
    LegLD<LEG> lf;
    load_leg_factor<LEG>(S, lf);
    const V3 o = {ct_qpos0[0], ct_qpos0[1], ct_qpos0[2]};
    sfor<0, 6>([&](auto Sl) {
        constexpr int b = cbody<LEG>(Sl);
        const V3 c = ldv3<WK_PTS + 30 * LEG + 3 * Sl>(S);
        float J[3][19];
        sfor<0, 19>([&](auto K) { J[0][K] = 0.f; J[1][K] = 0.f; J[2][K] = 0.f; });
        jac_point3<b>(S, c - o, 1.f, J[0], J[1], J[2]);
        float tr = 0.f;
        sfor<0, 3>([&](auto A) { whiten3<LEG, SetALL>(lf, J[A]); sfor<0, 19>([&](auto K) { tr += J[A][K] * J[A][K]; }); });
        S(F_BIW + b) = tr * (1.f / 3.f);
        __builtin_amdgcn_sched_barrier(0);
    });
    sfor<0, NJ>([&](auto Jn) {
        constexpr int j = Jn;
        if constexpr (ct_jnt_limited[j] && ((ct_jnt_body[j] >= 14) == (LEG == 1)) && ct_jnt_body[j] >= 2) {
            constexpr int d = ct_jnt_dofadr[j];
            float Jl[19];
            sfor<0, 19>([&](auto K) { Jl[K] = 0.f; });
            Jl[d2c(d)] = 1.f;
            whiten3<LEG, SetALL>(lf, Jl);
            float s2 = 0.f;
            sfor<0, 19>([&](auto K) { s2 += Jl[K] * Jl[K]; });
            S(F_DIW + d) = s2;
            __builtin_amdgcn_sched_barrier(0);
        }
    });
}

}  // namespace c3
