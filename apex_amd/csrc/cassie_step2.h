// forward dynamics + Euler + mj_setConst subset, fully static (see cassie_dev2.h for the formulation)
#pragma once
#include "cassie_dev2.h"
#include "env_state.h"

namespace c2 {

static_assert(ct_jnt_type[0] == 0 && ct_jnt_type[1] == 0 && ct_jnt_type[2] == 0 && ct_jnt_type[3] == 2, "pelvis = 3 slides + ball");
static_assert(ct_body_parent[1] == 0, "pelvis hangs off the world");

// which bodies' poses the constraints need: per leg slot 0 achilles, 1 heel-spring, 2 plantar-rod, 3 foot, 4 tarsus, 5 shin
template <int LEG> constexpr int cbody(int s) { constexpr int t[6] = {5, 10, 12, 13, 9, 8}; return t[s] + 12 * LEG; }
constexpr int cslot(int b) {
    const int l = b >= 14 ? b - 12 : b;
    return l == 5 ? 0 : l == 10 ? 1 : l == 12 ? 2 : l == 13 ? 3 : l == 9 ? 4 : l == 8 ? 5 : -1;
}

// FK + cdof + (velocity, RNE forward) + inertias + backward accumulation + CRBA + bias, for qpos given by `qp`.
// QPOS0 = true: configuration-only pass for mj_setConst (no velocities, no forces).
template <bool QPOS0, class QP>
__device__ __forceinline__ void kinematics_dynamics(const St& S, Fwd& w, const QP& qp, float (&smooth)[NV]) {
    V3 xpos[NB]; Q4 xquat[NB]; M3 xmat[NB]; SV cvel[NB], cacc[NB], cfrc[NB]; SI cin[NB];
    float qvel[NV];
    if constexpr (!QPOS0) sfor<0, NV>([&](auto D) { qvel[D] = S(F_QVEL + D); });
    // ---- pelvis: 3 world-aligned slides + ball
    xpos[1] = {qp(0), qp(1), qp(2)};
    xquat[1] = qnormalize(Q4{qp(3), qp(4), qp(5), qp(6)});
    xmat[1] = q2m(xquat[1]);
    w.o = xpos[1];
    w.cdof[0] = {{0, 0, 0}, {1, 0, 0}}; w.cdof[1] = {{0, 0, 0}, {0, 1, 0}}; w.cdof[2] = {{0, 0, 0}, {0, 0, 1}};
    sfor<0, 3>([&](auto K) { w.cdof[3 + K] = {col(xmat[1], K), {0, 0, 0}}; });
    if constexpr (!QPOS0) {
        SV v = {{0, 0, 0}, {qvel[0], qvel[1], qvel[2]}};
        SV a = {{0, 0, 0}, {0, 0, GRAV}};
        const SV vp = v;
        sfor<3, 6>([&](auto D) { a = a + crossMotion(vp, w.cdof[D]) * qvel[D]; v = v + w.cdof[D] * qvel[D]; });
        cvel[1] = v; cacc[1] = a;
    }
    // ---- all other bodies (parents precede children)
    sfor<1, NB>([&](auto Bi) {
        constexpr int b = Bi, p = ct_body_parent[b];
        if constexpr (b >= 2) {
            const V3 pos = xpos[p] + mul(xmat[p], cv3<b>(ct_body_pos));
            Q4 quat = qmul(xquat[p], Q4{ct_body_quat[4 * b], ct_body_quat[4 * b + 1], ct_body_quat[4 * b + 2], ct_body_quat[4 * b + 3]});
            if constexpr (ct_body_jntnum[b] == 1) {
                constexpr int j = ct_body_jntadr[b], adr = ct_jnt_qposadr[j];
                if constexpr (ct_jnt_type[j] == 1) {
                    static_assert(ct_jnt_axis[3 * j] == 0.f && ct_jnt_axis[3 * j + 1] == 0.f && ct_jnt_axis[3 * j + 2] == 1.f, "hinges turn about local z");
                    float sn, cs;
                    __sincosf(0.5f * (qp(adr) - ct_jnt_ref[j]), &sn, &cs);
                    quat = qmul(quat, Q4{cs, 0.f, 0.f, sn});
                } else {
                    quat = qmul(quat, qnormalize(Q4{qp(adr), qp(adr + 1), qp(adr + 2), qp(adr + 3)}));
                }
            }
            xquat[b] = qnormalize(quat); xpos[b] = pos; xmat[b] = q2m(xquat[b]);
            if constexpr (ct_body_jntnum[b] == 1) {
                constexpr int j = ct_body_jntadr[b], d0 = ct_jnt_dofadr[j];
                const V3 r = w.o - pos;
                if constexpr (ct_jnt_type[j] == 1) { const V3 ax = col(xmat[b], 2); w.cdof[d0] = {ax, cross(ax, r)}; }
                else sfor<0, 3>([&](auto K) { const V3 ax = col(xmat[b], K); w.cdof[d0 + K] = {ax, cross(ax, r)}; });
            }
            if constexpr (!QPOS0) {
                SV v = cvel[p], a = cacc[p];
                const SV vp = v;
                sfor<0, ct_body_dofnum[b]>([&](auto K) {
                    constexpr int d = ct_body_dofadr[b] + K;
                    a = a + crossMotion(vp, w.cdof[d]) * qvel[d]; v = v + w.cdof[d] * qvel[d];
                });
                cvel[b] = v; cacc[b] = a;
            }
        }
        // spatial inertia about o, world axes (mass randomisation changes mass only, cassie.py:640)
        {
            const M3& R = xmat[b];
            constexpr const float* Ib = ct_body_inertia + 9 * b;
            float RI[9], Iw[9];
            sfor<0, 3>([&](auto I) { sfor<0, 3>([&](auto K) { RI[3 * I + K] = R.m[3 * I] * Ib[K] + R.m[3 * I + 1] * Ib[3 + K] + R.m[3 * I + 2] * Ib[6 + K]; }); });
            sfor<0, 3>([&](auto I) { sfor<0, 3>([&](auto K) { if constexpr (K >= I) Iw[3 * I + K] = RI[3 * I] * R.m[3 * K] + RI[3 * I + 1] * R.m[3 * K + 1] + RI[3 * I + 2] * R.m[3 * K + 2]; }); });
            const float m = S(F_MASS + b);
            const V3 r = xpos[b] + mul(R, cv3<b>(ct_body_ipos)) - w.o;
            const float rr = dot(r, r);
            SI c;
            c.m = m; c.h = r * m;
            c.I[0] = Iw[0] + m * (rr - r.x * r.x); c.I[1] = Iw[4] + m * (rr - r.y * r.y); c.I[2] = Iw[8] + m * (rr - r.z * r.z);
            c.I[3] = Iw[1] - m * r.x * r.y; c.I[4] = Iw[2] - m * r.x * r.z; c.I[5] = Iw[5] - m * r.y * r.z;
            cin[b] = c;
            if constexpr (!QPOS0) cfrc[b] = imul(c, cacc[b]) + crossForce(cvel[b], imul(c, cvel[b]));
        }
        // keep what the constraints / sensors need
        if constexpr (cslot(b) >= 0) { w.xpos_c[b >= 14][cslot(b)] = xpos[b]; w.xmat_c[b >= 14][cslot(b)] = xmat[b]; }
        if constexpr (b == 13 || b == 25) { w.footq[b == 25] = xquat[b]; w.footp[b == 25] = xpos[b]; }
    });
    w.pel_mat = xmat[1];
    if constexpr (!QPOS0) { w.pel_cacc = cacc[1]; w.pel_cvel = cvel[1]; }
    // ---- backward: subtree forces and composite inertias
    srfor<2, NB>([&](auto Bi) {
        constexpr int b = Bi, p = ct_body_parent[b];
        if constexpr (!QPOS0) cfrc[p] = cfrc[p] + cfrc[b];
        cin[p].m += cin[b].m; cin[p].h = cin[p].h + cin[b].h;
        sfor<0, 6>([&](auto I) { cin[p].I[I] += cin[b].I[I]; });
    });
    // ---- qfrc_smooth = passive - bias (+ actuation added by the caller)
    if constexpr (!QPOS0) {
        sfor<0, NV>([&](auto D) {
            constexpr int d = D, j = ct_dof_jnt[d];
            float f = -S(F_DAMP + d) * qvel[d] - sdot(w.cdof[d], cfrc[ct_dof_body[d]]);
            if constexpr (ct_jnt_type[j] != 2 && ct_jnt_stiffness[j] != 0.f) f -= ct_jnt_stiffness[j] * qp(ct_jnt_qposadr[j]);
            smooth[d] = f;
        });
    }
    // ---- CRBA straight into the factorisation buffer
    sfor<0, NV>([&](auto I) {
        constexpr int i = I;
        const SV f = imul(cin[ct_dof_body[i]], w.cdof[i]);
        sfor<0, ct_dof_depth[i]>([&](auto A) {
            constexpr int a = A;
            float v = sdot(w.cdof[ct_dof_anc[16 * i + a]], f);
            if constexpr (a == 0) v += ct_dof_armature[i];
            w.LD[ct_dof_madr[i] + a] = v;
        });
    });
}

// ---------------------------------------------------------------------------------------------- constraint rows
// equality row -> LDS chunks [16 cols | b R invA f]; returns warm-start contributions through z~ / cost
template <int LEG, class SET, int NCH>
__device__ __forceinline__ void commit_simple(Fwd& w, const Lds& L, int chunk0, const float (&y)[19],
                                              bool unilateral, float pos, float imp_pos, float diag, float timeconst, float& cost) {
    float vel, ju, jw, nn;
    dots<LEG, SET>(w, y, vel, ju, jw, nn);
    const RowK kb = solref(timeconst);
    const float imp = impedance(imp_pos);
    const float R = fmaxf(MINVAL, (1.f - imp) / imp * diag);
    const float aref = -kb.B * vel - kb.K * imp * pos;
    const float b = ju - aref;
    float f = -(jw - aref) / R;                      // warm start from the previous qacc (mj_constraintUpdate)
    if (unilateral && f < 0.f) f = 0.f;
    const float invA = 1.f / (nn + R);
    // columns, in the order of `sup`, padded with zeros
    float vals[20];
    sfor<0, 20>([&](auto I) {
        constexpr int i = I;
        if constexpr (i < SET::N) { if constexpr (SET::c[i] >= 0) vals[i] = y[SET::c[i]]; else vals[i] = 0.f; } else vals[i] = 0.f;
    });
    sfor<0, NCH>([&](auto C) { constexpr int c = C; L.wr(chunk0 + c, make_float4(vals[4 * c], vals[4 * c + 1], vals[4 * c + 2], vals[4 * c + 3])); });
    L.wr(chunk0 + NCH, make_float4(b, R, invA, f));
    sfor<0, 19>([&](auto C) { constexpr int c = C; if constexpr (has<SET>(c)) w.zt[c2d<LEG>(c)] += y[c] * f; });
    cost += f * (0.5f * R * f + b);
}

template <int LEG>
__device__ __forceinline__ void build_rows(const St& S, Fwd& w, const Lds& L, const Dyn2& dy, float& cost) {
    // ---- 2 connect equalities (cassie.xml:225-230): plantar-rod <-> foot, achilles-rod <-> heel-spring
    sfor<0, 2>([&](auto E) {
        constexpr int e = 2 * LEG + E, b1 = ct_eq_body1[e], b2 = ct_eq_body2[e];
        const V3 p1 = w.xpos_c[LEG][cslot(b1)] + mul(w.xmat_c[LEG][cslot(b1)], cv3<e>(ct_eq_anchor1));
        const V3 p2 = w.xpos_c[LEG][cslot(b2)] + mul(w.xmat_c[LEG][cslot(b2)], cv3<e>(ct_eq_anchor2));
        const V3 c = p1 - p2;
        float J[3][19];
        sfor<0, 19>([&](auto K) { J[0][K] = 0.f; J[1][K] = 0.f; J[2][K] = 0.f; });
        jac_point<b1>(w, p1, 1.f, J[0], J[1], J[2]);
        jac_point<b2>(w, p2, -1.f, J[0], J[1], J[2]);
        const float cp[3] = {c.x, c.y, c.z};
        const float tran = S(F_BIW + b1) + S(F_BIW + b2), cn = sqrtf(dot(c, c));
        sfor<0, 3>([&](auto K) {
            constexpr int k = K, row = LEG * 6 + E * 3 + k;
            if constexpr (E == 0) { whiten<LEG, SetPL>(w, J[k]); commit_simple<LEG, SetPL, 4>(w, L, CH_EQ + 5 * row, J[k], false, cp[k], cn, tran, 0.005f, cost); }
            else { whiten<LEG, SetAC>(w, J[k]); commit_simple<LEG, SetAC, 4>(w, L, CH_EQ + 5 * row, J[k], false, cp[k], cn, tran, 0.005f, cost); }
        });
    });
    // ---- first active joint limit of this leg (solreflimit default 0.02 1)
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
                whiten<LEG, SetALL>(w, Jl);
                commit_simple<LEG, SetALL, 5>(w, L, CH_LIM + 6 * LEG, Jl, true, dist, dist, S(F_DIW + d), 0.02f, cost);
                w.nlim[LEG] = 1;
            }
        }
    });
    // ---- contacts: foot / tarsus / shin capsule ends vs the floor plane, first 3 penetrating ones in that order
    w.ncon[LEG] = 0;
    const V3 p0 = {ct_floor_pos[0], ct_floor_pos[1], ct_floor_pos[2]};
    sfor<0, 3>([&](auto G) {
        constexpr int g = 2 * G + LEG, b = ct_geom_body[g], sl = cslot(b);
        const V3 c = w.xpos_c[LEG][sl] + mul(w.xmat_c[LEG][sl], cv3<g>(ct_geom_pos));
        const V3 ax = mul(w.xmat_c[LEG][sl], cv3<g>(ct_geom_axis));
        sfor<0, 2>([&](auto E) {
            const V3 ctr = E == 0 ? c + ax * ct_geom_half[g] : c - ax * ct_geom_half[g];
            const float dist = dot(ctr - p0, dy.fn) - ct_geom_radius[g];
            if (dist < 0.f && w.ncon[LEG] < 3) {
                const int slot = 3 * LEG + w.ncon[LEG];
                const V3 cp = ctr - dy.fn * (ct_geom_radius[g] + 0.5f * dist);
                float Jx[19], Jy[19], Jz[19];
                sfor<0, 19>([&](auto K) { Jx[K] = 0.f; Jy[K] = 0.f; Jz[K] = 0.f; });
                jac_point<b>(w, cp, 1.f, Jx, Jy, Jz);
                float yn[19], y1[19], y2[19];
                sfor<0, 13>([&](auto I) {
                    constexpr int k = SetFT::c[I];
                    yn[k] = dy.fn.x * Jx[k] + dy.fn.y * Jy[k] + dy.fn.z * Jz[k];
                    y1[k] = dy.ft1.x * Jx[k] + dy.ft1.y * Jy[k] + dy.ft1.z * Jz[k];
                    y2[k] = dy.ft2.x * Jx[k] + dy.ft2.y * Jy[k] + dy.ft2.z * Jz[k];
                });
                whiten<LEG, SetFT>(w, yn); whiten<LEG, SetFT>(w, y1); whiten<LEG, SetFT>(w, y2);
                float vn, un, wn, gnn, v1, u1, w1, g11, v2, u2, w2, g22, gn1 = 0.f, gn2 = 0.f, g12 = 0.f;
                dots<LEG, SetFT>(w, yn, vn, un, wn, gnn);
                dots<LEG, SetFT>(w, y1, v1, u1, w1, g11);
                dots<LEG, SetFT>(w, y2, v2, u2, w2, g22);
                sfor<0, 13>([&](auto I) { constexpr int k = SetFT::c[I]; gn1 += yn[k] * y1[k]; gn2 += yn[k] * y2[k]; g12 += y1[k] * y2[k]; });
                const float mu = dy.friction, tran = S(F_BIW + b);
                const RowK kb = solref(0.005f);
                const float imp = impedance(dist);
                const float R1 = fmaxf(MINVAL, (1.f - imp) / imp * (tran + mu * mu * tran));
                const float Rpy = fmaxf(MINVAL, 2.f * mu * mu * R1);       // pyramidal regulariser, impratio 1
                // rows k = 0..3: n + mu t1, n - mu t1, n + mu t2, n - mu t2
                const float sv[4] = {mu * v1, -mu * v1, mu * v2, -mu * v2}, su[4] = {mu * u1, -mu * u1, mu * u2, -mu * u2};
                const float sw[4] = {mu * w1, -mu * w1, mu * w2, -mu * w2};
                float bk[4], fk[4];
                sfor<0, 4>([&](auto K) {
                    constexpr int k = K;
                    const float aref = -kb.B * (vn + sv[k]) - kb.K * imp * dist;
                    bk[k] = un + su[k] - aref;
                    float f = -((wn + sw[k]) - aref) / Rpy;
                    fk[k] = f < 0.f ? 0.f : f;
                    cost += fk[k] * (0.5f * Rpy * fk[k] + bk[k]);
                });
                const float dn = fk[0] + fk[1] + fk[2] + fk[3], d1 = mu * (fk[0] - fk[1]), d2 = mu * (fk[2] - fk[3]);
                sfor<0, 13>([&](auto I) { constexpr int k = SetFT::c[I]; w.zt[c2d<LEG>(k)] += yn[k] * dn + y1[k] * d1 + y2[k] * d2; });
                float vals[40];
                sfor<0, 13>([&](auto I) { constexpr int k = SetFT::c[I]; vals[I] = yn[k]; vals[13 + I] = y1[k]; vals[26 + I] = y2[k]; });
                vals[39] = 0.f;
                const int ch = CH_CON + 14 * slot;
                sfor<0, 10>([&](auto C) { constexpr int cc = C; L.wr(ch + cc, make_float4(vals[4 * cc], vals[4 * cc + 1], vals[4 * cc + 2], vals[4 * cc + 3])); });
                L.wr(ch + 10, make_float4(gnn, gn1, gn2, g11));
                L.wr(ch + 11, make_float4(g12, g22, Rpy, 0.f));
                L.wr(ch + 12, make_float4(bk[0], bk[1], bk[2], bk[3]));
                L.wr(ch + 13, make_float4(fk[0], fk[1], fk[2], fk[3]));
                if (g < 2) w.footmask |= 1u << slot;
                w.ncon[LEG] += 1;
            }
        });
    });
}

// one Gauss-Seidel sweep over the rows of leg LEG (z~ in registers, rows from LDS)
template <int LEG>
__device__ __forceinline__ void pgs_leg(Fwd& w, const Lds& L, float mu) {
    // equalities: 6 rows, bilateral
    sfor<0, 6>([&](auto Rw) {
        constexpr int row = LEG * 6 + Rw, ch = CH_EQ + 5 * row;
        float y[16];
        sfor<0, 4>([&](auto C) { const float4 v = L.rd(ch + C); y[4 * C] = v.x; y[4 * C + 1] = v.y; y[4 * C + 2] = v.z; y[4 * C + 3] = v.w; });
        const float4 m = L.rd(ch + 4);          // b R invA f
        float res = m.x + m.y * m.w;
        sfor<0, 16>([&](auto I) {
            constexpr int i = I, c = Rw < 3 ? SetPL::c[i] : SetAC::c[i];
            if constexpr (c >= 0) res += y[i] * w.zt[c2d<LEG>(c)];
        });
        const float df = -res * m.z;
        sfor<0, 16>([&](auto I) {
            constexpr int i = I, c = Rw < 3 ? SetPL::c[i] : SetAC::c[i];
            if constexpr (c >= 0) w.zt[c2d<LEG>(c)] += y[i] * df;
        });
        L.wr(ch + 4, make_float4(m.x, m.y, m.z, m.w + df));
    });
    // limit slot
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
    // contacts: 4 pyramid rows each, swept through the Gram matrix of (n, t1, t2)
    sfor<0, 3>([&](auto Sl) {
        constexpr int s = Sl;
        if (s >= w.ncon[LEG]) return;
        constexpr int ch = CH_CON + 14 * (3 * LEG + s);
        float v[40];
        sfor<0, 10>([&](auto C) { const float4 q = L.rd(ch + C); v[4 * C] = q.x; v[4 * C + 1] = q.y; v[4 * C + 2] = q.z; v[4 * C + 3] = q.w; });
        const float4 g0 = L.rd(ch + 10), g1 = L.rd(ch + 11), bb = L.rd(ch + 12), ff = L.rd(ch + 13);
        const float gnn = g0.x, gn1 = g0.y, gn2 = g0.z, g11 = g0.w, g12 = g1.x, g22 = g1.y, R = g1.z;
        float dn = 0.f, d1 = 0.f, d2 = 0.f;
        sfor<0, 13>([&](auto I) { constexpr int d = c2d<LEG>(SetFT::c[I]); dn += v[I] * w.zt[d]; d1 += v[13 + I] * w.zt[d]; d2 += v[26 + I] * w.zt[d]; });
        float f[4] = {ff.x, ff.y, ff.z, ff.w};
        const float b[4] = {bb.x, bb.y, bb.z, bb.w};
        float sdn = 0.f, sd1 = 0.f, sd2 = 0.f;      // accumulated force changes along n, t1, t2
        sfor<0, 4>([&](auto K) {
            constexpr int k = K;
            constexpr float sg = (k & 1) ? -1.f : 1.f;
            const float sm = sg * mu;
            // row vector y_k = n + sm * t_j  (j = 1 for k < 2, else 2)
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
__device__ __forceinline__ void zero_forces(const Fwd& w, const Lds& L) {
    sfor<0, 6>([&](auto Rw) { constexpr int ch = CH_EQ + 5 * (LEG * 6 + Rw) + 4; float4 m = L.rd(ch); m.w = 0.f; L.wr(ch, m); });
    if (w.nlim[LEG]) { constexpr int ch = CH_LIM + 6 * LEG + 5; float4 m = L.rd(ch); m.w = 0.f; L.wr(ch, m); }
    for (int s = 0; s < 3; ++s) if (s < w.ncon[LEG]) L.wr(CH_CON + 14 * (3 * LEG + s) + 13, make_float4(0.f, 0.f, 0.f, 0.f));
}

// mj_forward: leaves qacc, u~, z~, LD (factor of M) in `w`, M itself in the global workspace column of this env
__device__ __forceinline__ void forward2(const St& S, Fwd& w, const Lds& L, const float (&ctrl)[10], int pgs_iters) {
    float smooth[NV];
    kinematics_dynamics<false>(S, w, [&](int i) { return S(F_QPOS + i); }, smooth);
    sfor<0, NU>([&](auto U) {
        constexpr int u = U;
        const float c = fminf(fmaxf(ctrl[u], -ct_act_ctrlmax[u]), ct_act_ctrlmax[u]);
        smooth[ct_act_dof[u]] += ct_act_gear[u] * c;
    });
    sfor<0, NM>([&](auto I) { S.W(I) = w.LD[I]; });                       // M, re-read by the Euler step
    factor<true>(w.LD, w.dsqrt, w.disqrt);
    sfor<0, NV>([&](auto D) { w.ut[D] = smooth[D]; });
    solve_LT(w.LD, w.ut);
    sfor<0, NV>([&](auto D) { w.ut[D] *= w.disqrt[D]; });
    {
        float tmp[NV];
        sfor<0, NV>([&](auto D) { tmp[D] = S(F_QVEL + D); });
        mul_L(w.LD, tmp, w.vt);
        sfor<0, NV>([&](auto D) { w.vt[D] *= w.dsqrt[D]; tmp[D] = S(F_QACCW + D); });
        mul_L(w.LD, tmp, w.wt);
        sfor<0, NV>([&](auto D) { w.wt[D] *= w.dsqrt[D]; });
    }
    Dyn2 dy;
    dy.friction = S(F_FRIC);
    dy.fn = {S(F_FLOOR), S(F_FLOOR + 1), S(F_FLOOR + 2)};
    dy.ft1 = {S(F_FLOOR + 3), S(F_FLOOR + 4), S(F_FLOOR + 5)};
    dy.ft2 = {S(F_FLOOR + 6), S(F_FLOOR + 7), S(F_FLOOR + 8)};
    sfor<0, NV>([&](auto D) { w.zt[D] = 0.f; });
    w.footmask = 0u;
    float cost = 0.f;
    build_rows<0>(S, w, L, dy, cost);
    build_rows<1>(S, w, L, dy, cost);
    sfor<0, NV>([&](auto D) { cost += 0.5f * w.zt[D] * w.zt[D]; });
    if (cost > 0.f) {                                                      // warm start loses to f = 0 (mj_fwdConstraint)
        zero_forces<0>(w, L); zero_forces<1>(w, L);
        sfor<0, NV>([&](auto D) { w.zt[D] = 0.f; });
    }
    for (int it = 0; it < pgs_iters; ++it) {
        pgs_leg<0>(w, L, dy.friction);
        pgs_leg<1>(w, L, dy.friction);
    }
    sfor<0, NV>([&](auto D) { w.qacc[D] = (w.ut[D] + w.zt[D]) * w.disqrt[D]; });
    solve_L(w.LD, w.qacc);
    // world z of the contact force on the foot bodies (cassie_sim_foot_forces -> get_foot_forces()[2], [8])
    w.foot_fz[0] = w.foot_fz[1] = 0.f;
    sfor<0, 6>([&](auto Sl) {
        constexpr int sl = Sl, lg = sl / 3;
        if ((sl % 3) < w.ncon[lg] && ((w.footmask >> sl) & 1u)) {
            const float4 ff = L.rd(CH_CON + 14 * sl + 13);
            w.foot_fz[lg] += dy.fn.z * (ff.x + ff.y + ff.z + ff.w) + dy.friction * (dy.ft1.z * (ff.x - ff.y) + dy.ft2.z * (ff.z - ff.w));
        }
    });
    // accelerometer at the imu site (cassie.xml:267): classical acceleration of the site point, site frame
    {
        SV A = w.pel_cacc;
        sfor<0, 6>([&](auto D) { A = A + w.cdof[D] * w.qacc[D]; });
        const V3 r = mul(w.pel_mat, V3{ct_imu_pos[0], ct_imu_pos[1], ct_imu_pos[2]});
        const V3 om = w.pel_cvel.a;
        const V3 vp = w.pel_cvel.l + cross(om, r);
        const V3 a = A.l + cross(A.a, r) + cross(om, vp);
        w.acc[0] = dot(col(w.pel_mat, 0), a); w.acc[1] = dot(col(w.pel_mat, 1), a); w.acc[2] = dot(col(w.pel_mat, 2), a);
    }
}

// mj_Euler with implicit joint damping: (M + h D) a = qfrc_smooth + J^T f = L^T D^1/2 (u~ + z~)
__device__ __forceinline__ void euler2(const St& S, Fwd& w) {
    float x[NV], rhs[NV];
    sfor<0, NV>([&](auto D) { x[D] = (w.ut[D] + w.zt[D]) * w.dsqrt[D]; });
    mul_LT(w.LD, x, rhs);
    sfor<0, NM>([&](auto I) { w.LD[I] = S.W(I); });
    sfor<0, NV>([&](auto D) { w.LD[ct_dof_madr[D]] += DT * S(F_DAMP + D); });
    factor<false>(w.LD, w.dsqrt, w.disqrt);
    solve_LT(w.LD, rhs);
    sfor<0, NV>([&](auto D) { rhs[D] *= __frcp_rn(w.LD[ct_dof_madr[D]]); });
    solve_L(w.LD, rhs);
    float qv[NV];
    sfor<0, NV>([&](auto D) { S(F_QACCW + D) = w.qacc[D]; qv[D] = S(F_QVEL + D) + DT * rhs[D]; S(F_QVEL + D) = qv[D]; });
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

}  // namespace c2
