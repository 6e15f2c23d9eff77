// Cassie substep, generation 4: ONE ENV PER 16-LANE DPP ROW (4 envs per wave64, 1024 waves at 4096 envs = one per SIMD).
//
// Why: the env-per-lane program (generations 2-3) is bounded by the serial instruction stream of one env: a wave64
// VALU instruction holds its SIMD16 for 4 cycles whatever the number of live lanes, and 4096 envs are only 64 waves.
// Giving an env 16 lanes shortens that stream: a Gauss-Seidel row update is 2 multiplies, 4 DPP adds, the scalar
// update and 2 fma instead of two 16-term loops.
//
// Lane map of an env row (l = lane & 15):
//   dof view   l = 0..12 : leg dof k = l of BOTH legs (A = left dof 6+l, B = right dof 19+l)
//              l = 13..15: pelvis dofs (A = dof l-13 (slide x,y,z), B = dof l-10 (ball x,y,z))
//   A constraint row of leg LEG is stored as two floats per lane (ya, yp): value that multiplies zA, value that
//   multiplies zB.  Left rows: (col, 0) on leg lanes; right rows: (0, col); pelvis lanes: (col l-13, col l-10).
#pragma once
#include "cassie_step3.h"

namespace c4 {
using namespace c3;
static_assert(L4_WK + WK_TOTAL <= L4_ROWS && L4_ROWS % 4 == 0 && L4_ROWS + 704 <= L4_ES && L4_ES % 64 == 16, "per-env LDS layout");

template <int CTRL> __device__ __forceinline__ float dpp(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, true));
}
// all-lanes sum over the 16-lane row (butterfly: every lane ends with the bit-identical total)
__device__ __forceinline__ float red16(float t) {
    t += dpp<0xB1>(t);      // quad_perm [1,0,3,2]
    t += dpp<0x4E>(t);      // quad_perm [2,3,0,1]
    t += dpp<0x141>(t);     // row_half_mirror
    t += dpp<0x140>(t);     // row_mirror
    return t;
}

// ------------------------------------------------------------------------------------------------ row store (LDS)
// row vector r = 13 * leg + lane (lane 0-2: plantar-rod<->foot connect x,y,z; 3-5: achilles<->heel-spring connect; 6: first
// active joint limit; 7-9 / 10-12: contact slot 0 / 1 basis n, t1, t2) at rows[24 r]: [0..18] whitened columns,
// [20..23] b, R, 1/(A+R), f (equality and limit rows).  Contact slot s = 2 * leg + slot at rows[R4_CON + 20 s]:
// [0..5] Gram (nn, n1, n2, 11, 12, 22), [6] R of the pyramid rows, [7] 1 if a foot capsule, [8..11] b, [12..15] f.
constexpr int R4_ROW = 24, R4_CON = 26 * R4_ROW, R4_CONSZ = 20, R4_TOTAL = R4_CON + 4 * R4_CONSZ;
static_assert(MAXC == 2, "lane map has two contact slots per leg");

// local-column bitmask of the dofs that move `body` (its ancestor chain)
template <int LEG> constexpr unsigned chain_mask(int body) {
    unsigned m = 0;
    const int last = ct_body_lastdof[body];
    for (int a = 0; a < ct_dof_depth[last]; ++a) m |= 1u << d2c(ct_dof_anc[16 * last + a]);
    return m;
}

// Constraint rows of leg LEG, one row vector per lane: Jacobian from the stored motion axes, dots against
// qvel / qacc_smooth / qacc_warmstart on the raw row, whitening y~ = D^-1/2 L^-T J^T (L streamed from LDS, uniform
// addresses), row scalars.  Same arithmetic as c3::build_rows3, except that a connect row takes the common ancestors of
// its two bodies as axis x (p1 - p2) instead of the difference of two point Jacobians.
struct LegRows {
    float J[19];                 // this lane's whitened row vector (local columns)
    float b, R, invA, f;         // this lane's row scalars (equality / limit lanes)
    int nc, nlim;                // uniform over the env's lanes from here on
    float cG[MAXC][6], cR[MAXC], cb[MAXC][4], cf[MAXC][4], isfoot[MAXC];
};
template <int LEG>
__device__ __forceinline__ void rows_lane(const St& S, LegRows& out) {
    const int l = threadIdx.x & 15;
    const V3 o = {S(F_QPOS), S(F_QPOS + 1), S(F_QPOS + 2)};
    const V3 fn = {S(F_FLOOR), S(F_FLOOR + 1), S(F_FLOOR + 2)}, ft1 = {S(F_FLOOR + 3), S(F_FLOOR + 4), S(F_FLOOR + 5)},
             ft2 = {S(F_FLOOR + 6), S(F_FLOOR + 7), S(F_FLOOR + 8)};
    const float mu = S(F_FRIC);
    constexpr int base = WK_PTS + 30 * LEG;
    // ---- uniform over the env's lanes: first active joint limit of this leg
    int nlim = 0, clim = -1;
    float lsign = 0.f, ldist = 0.f, ldiw = 0.f;
    sfor<0, NJ>([&](auto Jn) {
        constexpr int j = Jn;
        if constexpr (ct_jnt_limited[j] && ((ct_jnt_body[j] >= 14) == (LEG == 1)) && ct_jnt_body[j] >= 2) {
            const float q = S(F_QPOS + ct_jnt_qposadr[j]);
            const float dlo = q - ct_jnt_range[2 * j], dhi = ct_jnt_range[2 * j + 1] - q;
            if ((dlo < 0.f || dhi < 0.f) && nlim == 0) {
                constexpr int d = ct_jnt_dofadr[j];
                clim = d2c(d); lsign = dlo < 0.f ? 1.f : -1.f; ldist = dlo < 0.f ? dlo : dhi; ldiw = S(F_DIW + d); nlim = 1;
            }
        }
    });
    // ---- uniform: first MAXC penetrating capsule ends in the order foot e0,e1, tarsus e0,e1, shin e0,e1
    const V3 p0 = {ct_floor_pos[0], ct_floor_pos[1], ct_floor_pos[2]};
    int nc = 0;
    V3 cpt[MAXC]; float cdist[MAXC]; int cgeo[MAXC];
    sfor<0, MAXC>([&](auto Sl) { cpt[Sl] = {0.f, 0.f, 0.f}; cdist[Sl] = 0.f; cgeo[Sl] = 0; });
    sfor<0, 6>([&](auto I) {
        constexpr int G = I / 2;
        const V3 ctr = ldv3<base + 12 + 3 * I>(S);
        const float dist = dot(ctr - p0, fn) - ct_geom_radius[2 * G + LEG];
        const bool hit = dist < 0.f && nc < MAXC;
        const V3 cp = ctr - fn * (ct_geom_radius[2 * G + LEG] + 0.5f * dist);
        sfor<0, MAXC>([&](auto Sl) { if (hit && nc == Sl) { cpt[Sl] = cp; cdist[Sl] = dist; cgeo[Sl] = G; } });
        nc += hit ? 1 : 0;
    });
    // ---- this lane's row
    const bool isEq = l < 6, isLim = l == 6;
    const int E = l >= 3 ? 1 : 0, cs = l >= 10 ? 1 : 0;
    const bool isCon = l >= 7 && l < 13 && cs < nc;
    const int ax = isEq ? l - 3 * E : (l - 7) - 3 * cs;              // component / basis index 0..2
    constexpr unsigned mPL1 = chain_mask<LEG>(ct_eq_body1[2 * LEG]), mPL2 = chain_mask<LEG>(ct_eq_body2[2 * LEG]);
    constexpr unsigned mAC1 = chain_mask<LEG>(ct_eq_body1[2 * LEG + 1]), mAC2 = chain_mask<LEG>(ct_eq_body2[2 * LEG + 1]);
    constexpr unsigned mFT = chain_mask<LEG>(13 + 12 * LEG);
    const int G = cs ? cgeo[1] : cgeo[0];
    const unsigned mcon = mFT & ~(G >= 1 ? (1u << 18) : 0u) & ~(G >= 2 ? (1u << 14) : 0u);   // tarsus / shin contact: dofs below do not move the point
    const unsigned m1 = isEq ? (E ? mAC1 : mPL1) : isCon ? mcon : 0u, m2 = isEq ? (E ? mAC2 : mPL2) : 0u;
    V3 p1, p2;
    {
        const V3 e1 = {S.W(base + 6 * E), S.W(base + 6 * E + 1), S.W(base + 6 * E + 2)};
        const V3 e2 = {S.W(base + 6 * E + 3), S.W(base + 6 * E + 4), S.W(base + 6 * E + 5)};
        const V3 cp = cs ? cpt[1] : cpt[0];
        p1 = isEq ? e1 : cp; p2 = e2;
    }
    V3 dir;
    {
        const V3 dc = ax == 0 ? fn : ax == 1 ? ft1 : ft2;
        const V3 de = {ax == 0 ? 1.f : 0.f, ax == 1 ? 1.f : 0.f, ax == 2 ? 1.f : 0.f};
        dir = isEq ? de : dc;
    }
    const V3 q1 = cross(p1 - o, dir), q2 = cross(p2 - o, dir);       // dir . (a x r) = a . (r x dir)
    float (&J)[19] = out.J;
    sfor<0, 19>([&](auto C) {
        constexpr int c = C, d = c2d<LEG>(c);
        const V3 ca = {S.W(WK_CDOF + 6 * d), S.W(WK_CDOF + 6 * d + 1), S.W(WK_CDOF + 6 * d + 2)};
        const V3 cl = {S.W(WK_CDOF + 6 * d + 3), S.W(WK_CDOF + 6 * d + 4), S.W(WK_CDOF + 6 * d + 5)};
        const float dl = dot(dir, cl);
        const float g1 = dl + dot(q1, ca), g2 = dl + dot(q2, ca);
        float v = ((m1 >> c) & 1u) ? g1 : 0.f;
        v -= ((m2 >> c) & 1u) ? g2 : 0.f;
        if (isLim && nlim && c == clim) v = lsign;
        J[c] = v;
    });
    float vel = 0.f, ju = 0.f, jw = 0.f;
    sfor<0, 19>([&](auto C) {
        constexpr int c = C, d = c2d<LEG>(c);
        vel += J[c] * S(F_QVEL + d); ju += J[c] * S.W(WK_QS + d); jw += J[c] * S(F_QACCW + d);
    });
    srfor<0, 19>([&](auto C) {
        constexpr int c = C, i = c2d<LEG>(c);
        sfor<1, ct_dof_depth[i]>([&](auto A) { constexpr int a = A; J[d2c(ct_dof_anc[16 * i + a])] -= S.W(WK_LD + ct_dof_madr[i] + a) * J[c]; });
    });
    float nn = 0.f;
    sfor<0, 19>([&](auto C) { constexpr int c = C; J[c] *= S.W(WK_DISQ + c2d<LEG>(c)); nn += J[c] * J[c]; });
    // ---- equality / limit scalars (mj_makeImpedance, mj_referenceConstraint, warm start from qacc_warmstart)
    {
        const V3 cv = p1 - p2;
        const float cpos = ax == 0 ? cv.x : ax == 1 ? cv.y : cv.z;
        const float tranPL = S(F_BIW + ct_eq_body1[2 * LEG]) + S(F_BIW + ct_eq_body2[2 * LEG]);
        const float tranAC = S(F_BIW + ct_eq_body1[2 * LEG + 1]) + S(F_BIW + ct_eq_body2[2 * LEG + 1]);
        const float pos = isEq ? cpos : ldist, imp_pos = isEq ? sqrtf(dot(cv, cv)) : ldist;
        const float diag = isEq ? (E ? tranAC : tranPL) : ldiw;
        const RowK kb = solref(isEq ? 0.005f : 0.02f);
        const float imp = impedance(imp_pos);
        const float R = fmaxf(MINVAL, (1.f - imp) / imp * diag);
        const float aref = -kb.B * vel - kb.K * imp * pos;
        float b = ju - aref;
        float f = -(jw - aref) / R;
        if (isLim && f < 0.f) f = 0.f;
        float invA = 1.f / (nn + R), Rw = R;
        if (isLim && !nlim) { b = 0.f; f = 0.f; invA = 0.f; Rw = 1.f; }
        out.b = b; out.R = Rw; out.invA = invA; out.f = f;
    }
    // ---- contact scalars: Gram matrix of (n, t1, t2) across the three basis lanes, pyramid rows n +- mu t_j
    float a1 = 0.f, a2 = 0.f;
    sfor<0, 19>([&](auto C) { a1 += J[C] * dpp<0x111>(J[C]); a2 += J[C] * dpp<0x112>(J[C]); });     // row_shr:1, row_shr:2
    sfor<0, MAXC>([&](auto Sl) {
        constexpr int s = Sl, ln = 7 + 3 * s;
        const float gnn = dpp<0x150 + ln>(nn), g11 = dpp<0x150 + ln + 1>(nn), g22 = dpp<0x150 + ln + 2>(nn);
        const float gn1 = dpp<0x150 + ln + 1>(a1), g12 = dpp<0x150 + ln + 2>(a1), gn2 = dpp<0x150 + ln + 2>(a2);
        const float vn = dpp<0x150 + ln>(vel), v1 = dpp<0x150 + ln + 1>(vel), v2 = dpp<0x150 + ln + 2>(vel);
        const float un = dpp<0x150 + ln>(ju), u1 = dpp<0x150 + ln + 1>(ju), u2 = dpp<0x150 + ln + 2>(ju);
        const float wn = dpp<0x150 + ln>(jw), w1 = dpp<0x150 + ln + 1>(jw), w2 = dpp<0x150 + ln + 2>(jw);
        const int Gs = cgeo[s];
        const float dist = cdist[s];
        const float tran = Gs == 0 ? S(F_BIW + 13 + 12 * LEG) : Gs == 1 ? S(F_BIW + 9 + 12 * LEG) : S(F_BIW + 8 + 12 * LEG);
        const RowK kb = solref(0.005f);
        const float imp = impedance(dist);
        const float R1 = fmaxf(MINVAL, (1.f - imp) / imp * (tran + mu * mu * tran));
        const float Rpy = fmaxf(MINVAL, 2.f * mu * mu * R1);       // pyramidal regulariser, impratio 1
        const float sv[4] = {mu * v1, -mu * v1, mu * v2, -mu * v2}, su[4] = {mu * u1, -mu * u1, mu * u2, -mu * u2};
        const float sw[4] = {mu * w1, -mu * w1, mu * w2, -mu * w2};
        out.cG[s][0] = gnn; out.cG[s][1] = gn1; out.cG[s][2] = gn2; out.cG[s][3] = g11; out.cG[s][4] = g12; out.cG[s][5] = g22;
        out.cR[s] = Rpy; out.isfoot[s] = Gs == 0 ? 1.f : 0.f;
        sfor<0, 4>([&](auto K) {
            constexpr int k = K;
            const float aref = -kb.B * (vn + sv[k]) - kb.K * imp * dist;
            out.cb[s][k] = un + su[k] - aref;
            const float f = -((wn + sw[k]) - aref) / Rpy;
            out.cf[s][k] = (s < nc && f > 0.f) ? f : 0.f;
        });
    });
    out.nc = nc; out.nlim = nlim;
}

// Projected Gauss-Seidel in GRAM SPACE.  Lane r (0..12) owns basis vector r of both legs (A = left, B = right): 6 connect
// rows, the limit row, and n, t1, t2 of two contact slots.  Instead of z~ the sweep carries rho_r = y~_r . z~ for every
// basis vector; a change d of the coefficient of basis s moves it by G[r][s] d, with G = Y~ Y~^T formed once per substep
// (19-term fma with a DPP row broadcast operand; the two legs only meet in the 6 pelvis columns).  A row update is then
// broadcast + scalar update + 2 fma: no cross-lane reduction inside the 50 sweeps.  Order is leg-major as before (left: 6
// equality rows, limit, contacts; then right), a pyramidal contact sweeps its 4 rows through the 3x3 Gram block.
__device__ __forceinline__ void stage_rows_pgs_lane(const St& S, float* rows, int pgs_iters) {
    const int l = threadIdx.x & 15;
    const float mu = S(F_FRIC);
    LegRows A, B;
    rows_lane<0>(S, A);
    __builtin_amdgcn_sched_barrier(0);
    rows_lane<1>(S, B);
    __builtin_amdgcn_sched_barrier(0);
    // ---- Gram columns of this lane's two rows
    float GAA[13], GBB[13], GAB[13], GBA[13];      // GXY[s] = y~_(r,X) . y~_(s,Y)
    sfor<0, 13>([&](auto Sx) {
        constexpr int s = Sx;
        float aa = 0.f, bb = 0.f, ab = 0.f, ba = 0.f;
        sfor<0, 19>([&](auto C) {
            constexpr int c = C;
            const float sa = dpp<0x150 + s>(A.J[c]), sb = dpp<0x150 + s>(B.J[c]);
            aa += A.J[c] * sa; bb += B.J[c] * sb;
            if constexpr (c < 6) { ab += A.J[c] * sb; ba += B.J[c] * sa; }
        });
        // pin the sums here: LLVM otherwise sinks the fma chains into the conditional contact blocks that consume them while the
        // (convergent) broadcasts stay put, and ~500 broadcast values get spilled to scratch in between
        asm volatile("" : "+v"(aa), "+v"(bb), "+v"(ab), "+v"(ba));
        GAA[s] = aa; GBB[s] = bb; GAB[s] = ab; GBA[s] = ba;
    });
    // ---- row scalars to every lane: c = b + R f, R, 1/(A+R), f
    float ec[2][7], eR[2][7], eiA[2][7], ef[2][7];
    sfor<0, 7>([&](auto Sx) {
        constexpr int s = Sx;
        eR[0][s] = dpp<0x150 + s>(A.R); eiA[0][s] = dpp<0x150 + s>(A.invA); ef[0][s] = dpp<0x150 + s>(A.f); ec[0][s] = dpp<0x150 + s>(A.b);
        eR[1][s] = dpp<0x150 + s>(B.R); eiA[1][s] = dpp<0x150 + s>(B.invA); ef[1][s] = dpp<0x150 + s>(B.f); ec[1][s] = dpp<0x150 + s>(B.b);
    });
    constexpr int NCS = 2 * MAXC;
    float cG[NCS][6], cR[NCS], cb[NCS][4], cf[NCS][4], ciA[NCS][4];
    bool con[NCS];
    sfor<0, NCS>([&](auto Sl) {
        constexpr int s = Sl, leg = s / MAXC, j = s % MAXC;
        const LegRows& Lr = leg ? B : A;
        con[s] = j < Lr.nc;
        sfor<0, 6>([&](auto K) { cG[s][K] = Lr.cG[j][K]; });
        cR[s] = Lr.cR[j];
        sfor<0, 4>([&](auto K) {
            constexpr int k = K;
            cb[s][k] = Lr.cb[j][k]; cf[s][k] = Lr.cf[j][k];
            const float sm = ((k & 1) ? -mu : mu), gnj = k < 2 ? cG[s][1] : cG[s][2], gjj = k < 2 ? cG[s][3] : cG[s][5];
            ciA[s][k] = __frcp_rn(cG[s][0] + 2.f * sm * gnj + mu * mu * gjj + cR[s]);      // A_kk + R of row n + s mu t_j
        });
    });
    const int nlim[2] = {A.nlim, B.nlim};
    // ---- warm start (mj_fwdConstraint): coefficient F_s of every basis vector, rho = G F, dual cost 1/2 F.rho + sum f (R f / 2 + b)
    float FA[13], FB[13];
    sfor<0, 7>([&](auto Sx) { FA[Sx] = ef[0][Sx]; FB[Sx] = ef[1][Sx]; });
    sfor<0, NCS>([&](auto Sl) {
        constexpr int s = Sl, leg = s / MAXC, j = s % MAXC;
        const float dn = cf[s][0] + cf[s][1] + cf[s][2] + cf[s][3], d1 = mu * (cf[s][0] - cf[s][1]), d2 = mu * (cf[s][2] - cf[s][3]);
        float (&F)[13] = leg ? FB : FA;
        F[7 + 3 * j] = dn; F[8 + 3 * j] = d1; F[9 + 3 * j] = d2;
    });
    float rA = 0.f, rB = 0.f, cost = 0.f;
    sfor<0, 13>([&](auto Sx) { rA += GAA[Sx] * FA[Sx] + GAB[Sx] * FB[Sx]; rB += GBA[Sx] * FA[Sx] + GBB[Sx] * FB[Sx]; });
    {
        float own = 0.f;       // F of this lane's own rows times rho
        sfor<0, 13>([&](auto Sx) { if (l == Sx) own = FA[Sx] * rA + FB[Sx] * rB; });
        cost = 0.5f * red16(own);
    }
    sfor<0, 2>([&](auto Lg) { sfor<0, 7>([&](auto Sx) { cost += ef[Lg][Sx] * (0.5f * eR[Lg][Sx] * ef[Lg][Sx] + ec[Lg][Sx]); }); });
    sfor<0, NCS>([&](auto Sl) { sfor<0, 4>([&](auto K) { cost += cf[Sl][K] * (0.5f * cR[Sl] * cf[Sl][K] + cb[Sl][K]); }); });
    if (cost > 0.f) {
        rA = rB = 0.f;
        sfor<0, 2>([&](auto Lg) { sfor<0, 7>([&](auto Sx) { ef[Lg][Sx] = 0.f; }); });
        sfor<0, NCS>([&](auto Sl) { sfor<0, 4>([&](auto K) { cf[Sl][K] = 0.f; }); });
    }
    sfor<0, 2>([&](auto Lg) { sfor<0, 7>([&](auto Sx) { ec[Lg][Sx] += eR[Lg][Sx] * ef[Lg][Sx]; }); });      // c = b + R f
    // ---- sweeps
    for (int it = 0; it < pgs_iters; ++it) {
        sfor<0, 2>([&](auto Lg) {
            constexpr int leg = Lg;
            const float (&Ga)[13] = leg ? GAB : GAA;      // how a coefficient of this leg's basis s moves rho_A, rho_B
            const float (&Gb)[13] = leg ? GBB : GBA;
            sfor<0, 6>([&](auto Sx) {
                constexpr int s = Sx;
                const float t = (leg ? dpp<0x150 + s>(rB) : dpp<0x150 + s>(rA)) + ec[leg][s];
                const float df = -t * eiA[leg][s];
                ec[leg][s] += eR[leg][s] * df; ef[leg][s] += df;
                rA += Ga[s] * df; rB += Gb[s] * df;
            });
            if (nlim[leg]) {
                const float t = (leg ? dpp<0x150 + 6>(rB) : dpp<0x150 + 6>(rA)) + ec[leg][6];
                float fn = ef[leg][6] - t * eiA[leg][6];
                fn = fn < 0.f ? 0.f : fn;
                const float df = fn - ef[leg][6];
                ef[leg][6] = fn; ec[leg][6] += eR[leg][6] * df;
                rA += Ga[6] * df; rB += Gb[6] * df;
            }
            sfor<0, MAXC>([&](auto Sl) {
                constexpr int j = Sl, s = leg * MAXC + j, ln = 7 + 3 * j;
                if (con[s]) {
                    float rn = leg ? dpp<0x150 + ln>(rB) : dpp<0x150 + ln>(rA), r1 = leg ? dpp<0x150 + ln + 1>(rB) : dpp<0x150 + ln + 1>(rA),
                          r2 = leg ? dpp<0x150 + ln + 2>(rB) : dpp<0x150 + ln + 2>(rA);
                    const float gnn = cG[s][0], gn1 = cG[s][1], gn2 = cG[s][2], g11 = cG[s][3], g12 = cG[s][4], g22 = cG[s][5];
                    float sdn = 0.f, sd1 = 0.f, sd2 = 0.f;
                    sfor<0, 4>([&](auto K) {
                        constexpr int k = K;
                        const float sm = (k & 1) ? -mu : mu;
                        const float res = cb[s][k] + cR[s] * cf[s][k] + rn + sm * (k < 2 ? r1 : r2);
                        float fn = cf[s][k] - res * ciA[s][k];
                        fn = fn < 0.f ? 0.f : fn;
                        const float df = fn - cf[s][k];
                        cf[s][k] = fn;
                        // y_k = n + sm t_j moves the basis residuals by df * (G n-col + sm G j-col)
                        if constexpr (k < 2) { rn += df * (gnn + sm * gn1); r1 += df * (gn1 + sm * g11); r2 += df * (gn2 + sm * g12); sd1 += sm * df; }
                        else { rn += df * (gnn + sm * gn2); r1 += df * (gn1 + sm * g12); r2 += df * (gn2 + sm * g22); sd2 += sm * df; }
                        sdn += df;
                    });
                    rA += Ga[ln] * sdn + Ga[ln + 1] * sd1 + Ga[ln + 2] * sd2;
                    rB += Gb[ln] * sdn + Gb[ln + 1] * sd1 + Gb[ln + 2] * sd2;
                }
            });
        });
    }
    // ---- z~ = sum_r y~_r F_r back to the dof layout of the finish stage
    float ownA = 0.f, ownB = 0.f;
    sfor<0, 7>([&](auto Sx) { if (l == Sx) { ownA = ef[0][Sx]; ownB = ef[1][Sx]; } });
    sfor<0, NCS>([&](auto Sl) {
        constexpr int s = Sl, leg = s / MAXC, ln = 7 + 3 * (s % MAXC);
        const float dn = cf[s][0] + cf[s][1] + cf[s][2] + cf[s][3], d1 = mu * (cf[s][0] - cf[s][1]), d2 = mu * (cf[s][2] - cf[s][3]);
        float& own = leg ? ownB : ownA;
        if (l == ln) own = dn;
        if (l == ln + 1) own = d1;
        if (l == ln + 2) own = d2;
    });
    if (l >= 13) ownA = ownB = 0.f;
    sfor<0, 19>([&](auto C) {
        constexpr int c = C;
        if constexpr (c < 6) { const float z = red16(A.J[c] * ownA + B.J[c] * ownB); if (l == 0) S.W(WK_ZT + c) = z; }
        else { const float za = red16(A.J[c] * ownA), zb = red16(B.J[c] * ownB); if (l == 0) { S.W(WK_ZT + c) = za; S.W(WK_ZT + c + 13) = zb; } }
    });
    if (l == 0) {
        S.W(WK_MISC + 0) = (float)A.nc; S.W(WK_MISC + 1) = (float)B.nc; S.W(WK_MISC + 2) = (float)A.nlim; S.W(WK_MISC + 3) = (float)B.nlim;
        // contact forces to the row store (foot-force readout in the finish stage)
        sfor<0, NCS>([&](auto Sl) {
            constexpr int s = Sl, leg = s / MAXC, j = s % MAXC;
            float* cr = rows + R4_CON + R4_CONSZ * s;
            cr[7] = (leg ? B : A).isfoot[j];
            sfor<0, 4>([&](auto K) { cr[12 + K] = cf[s][K]; });
        });
    }
}

}  // namespace c4
