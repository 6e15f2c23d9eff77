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
#include "cassie_common.h"

namespace c4 {
static_assert(L4_WK + WK_TOTAL <= L4_ROWS && L4_ROWS % 4 == 0 && L4_ROWS + 704 <= L4_ES && L4_ES % 64 == 16, "per-env LDS layout");

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

// level of leg dof j in the ancestor chain of leg dof lane l (1 = parent dof, ...; 0 = j is not a proper ancestor of l), one nibble per lane
constexpr unsigned long long crba_level_table(int j) {
    unsigned long long t = 0;
    for (int k = 0; k < 13; ++k)
        for (int a = 1; a < ct_dof_depth[6 + k]; ++a)
            if (ct_dof_anc[16 * (6 + k) + a] == 6 + j) t |= (unsigned long long)a << (4 * k);
    return t;
}
// ------------------------------------------------------------------------------------------------ tree stage
// Kinematics, velocities, RNE bias forces, composite inertias and the mass matrix, lane-parallel over the 12 bodies of a
// leg (lane b: left body 2+b in slot 0, right body 14+b in slot 1); the pelvis is computed by every lane.  Bodies are
// numbered depth-first, so a subtree is the contiguous range [b, b + ndesc].  Same formulas as c3::visit (all spatial
// quantities about the pelvis origin o, world axes); results go to the same workspace slots (WK_M in MuJoCo's
// ancestor-chain layout, WK_CDOF, WK_SMOOTH, WK_PTS, WK_PEL, F_FWD foot pose).
constexpr int WK_CTRL = WK_QACC;                        // actuator-side torques from the io stage (10)
constexpr int WK_DUMMY = WK_QACC + 16;                  // sink for predicated-off stores (keeps them branch-free)
// a pair store `S.W(off) .. S.W(off + 32)` (leg slot 1 of a WK_PTS point sits 30 words behind slot 0) aimed at the sink lands in the unused tail of WK_ZP2
static_assert(WK_DUMMY + 30 >= WK_ZP2 + 4 && WK_DUMMY + 33 <= WK_TOTAL, "sink of the pair stores");
constexpr unsigned long long nib(int a0, int a1, int a2, int a3, int a4, int a5, int a6, int a7, int a8, int a9, int a10, int a11) {
    return (unsigned long long)a0 | (unsigned long long)a1 << 4 | (unsigned long long)a2 << 8 | (unsigned long long)a3 << 12 |
           (unsigned long long)a4 << 16 | (unsigned long long)a5 << 20 | (unsigned long long)a6 << 24 | (unsigned long long)a7 << 28 |
           (unsigned long long)a8 << 32 | (unsigned long long)a9 << 36 | (unsigned long long)a10 << 40 | (unsigned long long)a11 << 44;
}
// leg-local body b = 0..11: hip-roll, hip-yaw, hip-pitch, achilles-rod, knee, knee-spring, shin, tarsus, heel-spring, foot-crank, plantar-rod, foot
constexpr unsigned long long TB_DEPTH = nib(1, 2, 3, 4, 4, 5, 5, 6, 7, 7, 8, 7);         // pelvis = 0
constexpr unsigned long long TB_PARENT = nib(15, 0, 1, 2, 2, 4, 4, 6, 7, 7, 9, 7);       // leg-local, 15 = pelvis
constexpr unsigned long long TB_PAR2 = nib(15, 15, 0, 1, 1, 2, 2, 4, 6, 6, 7, 6);         // parent of parent
constexpr unsigned long long TB_PAR4 = nib(15, 15, 15, 15, 15, 0, 0, 1, 2, 2, 4, 2);     // 4th ancestor
constexpr unsigned long long TB_NDESC = nib(11, 10, 9, 0, 7, 0, 5, 4, 0, 1, 0, 0);
constexpr unsigned long long TB_QOFF = nib(0, 1, 2, 3, 7, 15, 8, 9, 10, 11, 12, 13);     // qpos offset inside the leg block (15 = no joint)
constexpr unsigned long long TB_DOFF = nib(0, 1, 2, 3, 6, 15, 7, 8, 9, 10, 11, 12);      // dof offset inside the leg block
// leg-local dof k = 0..12 -> leg-local body
constexpr unsigned long long TD_BODY = nib(0, 1, 2, 3, 3, 3, 4, 6, 7, 8, 9, 10) | (11ull << 48);
constexpr unsigned long long TD_PDOF = nib(15, 0, 1, 2, 3, 4, 2, 6, 7, 8, 8, 10) | (8ull << 48);        // parent dof inside the leg (15 = pelvis)
constexpr unsigned long long TD_DEPTH = nib(7, 8, 9, 10, 11, 12, 10, 11, 12, 13, 13, 14) | (13ull << 48);   // = ct_dof_depth (6 pelvis ancestors included)
static_assert(ct_dof_depth[6 + 11] == 14 && ct_dof_depth[6 + 12] == 13 && ct_dof_depth[6 + 6] == 10 && ct_dof_anc[16 * 17 + 1] == 16 && ct_dof_anc[16 * 18 + 1] == 14, "dof chains");
__device__ __forceinline__ int nibble(unsigned long long t, int i) { return (int)((t >> (4 * i)) & 15ull); }
static_assert(ct_body_parent[8] == 6 && ct_body_parent[13] == 9 && ct_body_parent[12] == 11 && ct_body_dofadr[8] == 13 && ct_jnt_qposadr[9] == 15 &&
              ct_body_dofadr[13] == 18 && ct_jnt_qposadr[14] == 20 && ct_body_dofnum[5] == 3 && ct_body_dofnum[7] == 0, "leg topology tables");

// spatial inertia of a body about o in world axes + its RNE force (c3::visit)
__device__ __forceinline__ void body_inertia_force(const M3& R, V3 pos, V3 o, const float (&Ib)[9], V3 ipos, float m, const SV& vel, const SV& acc, SI& c, SV& frc) {
    float RI[9], Iw[9];
    sfor<0, 3>([&](auto I) { sfor<0, 3>([&](auto K) { RI[3 * I + K] = R.m[3 * I] * Ib[K] + R.m[3 * I + 1] * Ib[3 + K] + R.m[3 * I + 2] * Ib[6 + K]; }); });
    sfor<0, 3>([&](auto I) { sfor<0, 3>([&](auto K) { if constexpr (K >= I) Iw[3 * I + K] = RI[3 * I] * R.m[3 * K] + RI[3 * I + 1] * R.m[3 * K + 1] + RI[3 * I + 2] * R.m[3 * K + 2]; }); });
    const V3 r = pos + mul(R, ipos) - o;
    const float rr = dot(r, r);
    c.m = m; c.h = r * m;
    c.I[0] = Iw[0] + m * (rr - r.x * r.x); c.I[1] = Iw[4] + m * (rr - r.y * r.y); c.I[2] = Iw[8] + m * (rr - r.z * r.z);
    c.I[3] = Iw[1] - m * r.x * r.y; c.I[4] = Iw[2] - m * r.x * r.z; c.I[5] = Iw[5] - m * r.y * r.z;
    frc = imul(c, acc) + crossForce(vel, imul(c, vel));
}
// ... of the lane's body in BOTH leg slots (pair form, cassie_common.h)
__device__ __forceinline__ void body_inertia_force(const M3p& R, V3p pos, V3 o, const f2 (&Ib)[9], V3p ipos, f2 m, const SVp& vel, const SVp& acc, SIp& c, SVp& frc) {
    f2 RI[9], Iw[9];
    sfor<0, 3>([&](auto I) { sfor<0, 3>([&](auto K) { RI[3 * I + K] = R.m[3 * I] * Ib[K] + R.m[3 * I + 1] * Ib[3 + K] + R.m[3 * I + 2] * Ib[6 + K]; }); });
    sfor<0, 3>([&](auto I) { sfor<0, 3>([&](auto K) { if constexpr (K >= I) Iw[3 * I + K] = RI[3 * I] * R.m[3 * K] + RI[3 * I + 1] * R.m[3 * K + 1] + RI[3 * I + 2] * R.m[3 * K + 2]; }); });
    const V3p r = pos + mul(R, ipos) - lift(o);
    const f2 rr = dot(r, r);
    c.m = m; c.h = r * m;
    c.I[0] = Iw[0] + m * (rr - r.x * r.x); c.I[1] = Iw[4] + m * (rr - r.y * r.y); c.I[2] = Iw[8] + m * (rr - r.z * r.z);
    c.I[3] = Iw[1] - m * r.x * r.y; c.I[4] = Iw[2] - m * r.x * r.z; c.I[5] = Iw[5] - m * r.y * r.z;
    frc = imul(c, acc) + crossForce(vel, imul(c, vel));
}

struct Qpos0 { float v[CM_NQ]; };
constexpr Qpos0 make_qpos0() { Qpos0 q{}; for (int i = 0; i < CM_NQ; ++i) q.v[i] = ct_qpos0[i]; return q; }
__device__ const Qpos0 kQpos0 = make_qpos0();

template <int CTRL> __device__ __forceinline__ f2 dpp2(f2 v) { return f2{dpp<CTRL>(v.x), dpp<CTRL>(v.y)}; }
// Exchange records of the tree stage (row-store region): one per leg-local body, PAIR layout like the constant table: word k of leg slot sd at XB_SZ lb + 2 k + sd.
// [0..9] composite inertia (m, h3, I6), [10..15] subtree force, [16..18] the body's COM relative to the pelvis origin; record 12 = sink of the shadow lanes.
constexpr int XB_SZ = 40, XB_SPARE = 12;
static_assert(13 * XB_SZ <= 704, "exchange records fit the row-store region");

// QPOS0 = true: the configuration-only pass of mj_setConst (qpos0, zero velocities; emits the COM of the bodies that carry
// constraints instead of anchors / capsule ends)
//
// Round 5: the stage is written on (left, right) PAIRS.  A lane carries leg-local body b of both legs; every quantity of the two slots is an f2 and the arithmetic is
// v_pk_{mul,add,fma}_f32 - one instruction where rounds 1-4 issued two (the stage had 0 packed of 1 816 fp32 arithmetic instructions).  What cannot pack stays per
// component: DPP row shifts / broadcasts and ds_bpermute (VOP3P has no DPP form), selects, the transcendental pairs, LDS traffic at per-slot addresses.
template <bool QPOS0>
__device__ __forceinline__ void stage_tree_lane(const St& S, float* xb) {
    auto qp = [&](int i) -> float { if constexpr (QPOS0) return kQpos0.v[i]; else return S(F_QPOS + i); };
    auto qp2 = [&](int i) -> f2 { return f2{qp(i), qp(i + 14)}; };                 // the same joint of the two legs (qpos blocks 7.., 21..)
    auto qv2 = [&](int i) -> f2 { return f2{S(F_QVEL + i), S(F_QVEL + i + 13)}; };   // dof blocks 6.., 19..
    const int l = threadIdx.x & 15;
    const int lb = l < 12 ? l : 11;                                     // lanes 12..15 shadow the foot lane (their body results go to the spare record)
    const bool bl = l < 12;
    // ---- per-lane model constants of the two bodies (left, right): one 64-bit LDS operand per pair
    const int par = nibble(TB_PARENT, lb), qoff = nibble(TB_QOFF, lb), doff = nibble(TB_DOFF, lb);
    const bool hasj = qoff != 15, ball = lb == 3;
    const int qadr = 7 + qoff, dadr = 6 + doff;                         // left leg; the right leg's are + 14 / + 13
    const int cb = CT_BODY + CT_BODYSZ * lb;
    auto ct2 = [&](int k) -> f2 { return f2{ctf(cb + 2 * k), ctf(cb + 2 * k + 1)}; };
    const V3p bpos = {ct2(0), ct2(1), ct2(2)}, ipos = {ct2(3), ct2(4), ct2(5)};
    const Q4p bquat = {ct2(6), ct2(7), ct2(8), ct2(9)};
    f2 Ib[9];
    sfor<0, 9>([&](auto K) { Ib[K] = ct2(10 + K); });
    // the shadow lanes carry a massless copy of the foot: their composite inertia and force are exactly 0, so the suffix sums below need no masking
    const float blf = bl ? 1.f : 0.f;
    const f2 mass = f2{S(F_MASS + 2 + lb), S(F_MASS + 14 + lb)} * blf;
    sfor<0, 9>([&](auto K) { Ib[K] = Ib[K] * blf; });
    const int xrec = XB_SZ * (bl ? lb : XB_SPARE);                      // where this lane's exchange record goes
    float jref = 0.f; jref = lb == 7 ? ct_jnt_ref[10] : jref; jref = lb == 4 ? ct_jnt_ref[8] : jref;      // knee, tarsus (plain selects: a nested ?: becomes branches)
    static_assert(ct_jnt_ref[19] == ct_jnt_ref[8] && ct_jnt_ref[21] == ct_jnt_ref[10] && ct_jnt_ref[9] == 0.f, "joint refs");
    // ---- pelvis (every lane, uniform)
    const V3 o = {qp(0), qp(1), qp(2)};
    const Q4 pquat = qnormalize(Q4{qp(3), qp(4), qp(5), qp(6)});
    const M3 pmat = q2m(pquat);
    SV pc[6] = {{{0, 0, 0}, {1, 0, 0}}, {{0, 0, 0}, {0, 1, 0}}, {{0, 0, 0}, {0, 0, 1}}, {col(pmat, 0), {0, 0, 0}}, {col(pmat, 1), {0, 0, 0}}, {col(pmat, 2), {0, 0, 0}}};
    SV pvel, pacc;
    {
        SV v = {{0, 0, 0}, {S(F_QVEL), S(F_QVEL + 1), S(F_QVEL + 2)}};
        SV a = {{0, 0, 0}, {0, 0, GRAV}};
        const SV vp = v;
        sfor<3, 6>([&](auto D) { const float qd = S(F_QVEL + D); a = a + crossMotion(vp, pc[D]) * qd; v = v + pc[D] * qd; });
        pvel = v; pacc = a;
    }
    if (l == 0) {
        sfor<0, 6>([&](auto D) {
            S.W(WK_CDOF + 6 * D) = pc[D].a.x; S.W(WK_CDOF + 6 * D + 1) = pc[D].a.y; S.W(WK_CDOF + 6 * D + 2) = pc[D].a.z;
            S.W(WK_CDOF + 6 * D + 3) = pc[D].l.x; S.W(WK_CDOF + 6 * D + 4) = pc[D].l.y; S.W(WK_CDOF + 6 * D + 5) = pc[D].l.z;
        });
        S.W(WK_PEL + 0) = pacc.a.x; S.W(WK_PEL + 1) = pacc.a.y; S.W(WK_PEL + 2) = pacc.a.z;
        S.W(WK_PEL + 3) = pacc.l.x; S.W(WK_PEL + 4) = pacc.l.y; S.W(WK_PEL + 5) = pacc.l.z;
        S.W(WK_PEL + 6) = pvel.a.x; S.W(WK_PEL + 7) = pvel.a.y; S.W(WK_PEL + 8) = pvel.a.z;
        S.W(WK_PEL + 9) = pvel.l.x; S.W(WK_PEL + 10) = pvel.l.y; S.W(WK_PEL + 11) = pvel.l.z;
        sfor<0, 9>([&](auto K) { S.W(WK_PEL + 12 + K) = pmat.m[K]; });
    }
    // ---- local joint rotation and joint velocities of this lane's bodies (hinge axis = local z, ball = x, y, z)
    Q4p lq; f2 qd[3];
    {
        Q4p jq = {splat(1.f), splat(0.f), splat(0.f), splat(0.f)};
        qd[0] = qd[1] = qd[2] = splat(0.f);
        if (hasj) {
            if (ball) {
                jq = qnormalize(Q4p{qp2(qadr), qp2(qadr + 1), qp2(qadr + 2), qp2(qadr + 3)});
                sfor<0, 3>([&](auto K) { qd[K] = qv2(dadr + K); });
            } else {
                const f2 h = (qp2(qadr) - jref) * 0.5f;
                float s0, c0, s1, c1;
                __sincosf(h.x, &s0, &c0); __sincosf(h.y, &s1, &c1);
                jq = {f2{c0, c1}, splat(0.f), splat(0.f), f2{s0, s1}};
                qd[2] = qv2(dadr);
            }
        }
        lq = qmul(bquat, jq);
    }
    PROF2(12);
    // ---- pointer jumping over the ancestor chain (depth <= 8: 3 rounds).  Round r composes a body's transform with that of its 2^r-th ancestor, which by then spans
    // 2^r levels itself.  The ancestor's transform is fetched straight from its lane with ds_bpermute (the LDS crossbar, no memory): seven words per leg and round, one
    // wait per round.  A row_shr + select form is 7 VALU per word and is the trap of this kernel: clang predicates `c ? dpp(v) : r` as a DPP move under an exec mask,
    // and a DPP read of a DISABLED source lane returns 0.
    V3p tp = bpos; Q4p tq = lq;
    {
        const int jumpl[3] = {par, nibble(TB_PAR2, lb), nibble(TB_PAR4, lb)};
        const int rowbase = (int)(threadIdx.x & 48u);
        sfor<0, 3>([&](auto Rn) {
            constexpr int r = Rn;
            const bool on = jumpl[r] != 15;
            const int src = 4 * (rowbase + (on ? jumpl[r] : l));
            auto fetch = [&](f2 v) { return f2{__int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(v.x))), __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(v.y)))}; };
            const V3p ap = {fetch(tp.x), fetch(tp.y), fetch(tp.z)};
            const Q4p aq = {fetch(tq.w), fetch(tq.x), fetch(tq.y), fetch(tq.z)};
            const V3p np = ap + mul(q2m(aq), tp);
            const Q4p nq = qmul(aq, tq);
            tp = sel2(on, np, tp); tq = sel2(on, nq, tq);
        });
    }
    PROF2(13);
    const V3p pos = lift(o) + mul(pmat, tp);
    const Q4p quat = qnormalize(qmul(lift(pquat), tq));
    const M3p mat = q2m(quat);
    SVp cdof[3], own;                                                   // own = this body's joint velocity contribution sum_K cdof_K qd_K
    {
        const V3p r = lift(o) - pos;
        own = {{splat(0.f), splat(0.f), splat(0.f)}, {splat(0.f), splat(0.f), splat(0.f)}};
        sfor<0, 3>([&](auto K) { const V3p ax = col(mat, K); cdof[K] = {ax, cross(ax, r)}; own = own + cdof[K] * qd[K]; });
    }
    // chain sum of a spatial vector: val_b <- sum of val over b and its ancestors inside the leg.  Bodies are numbered depth-first, so "a is an
    // ancestor of b" is the interval test a <= b <= a + ndesc_a: add val_a at lane a, take it away again at lane a + ndesc_a + 1, and the inclusive
    // prefix sum over the lanes is the chain sum.  The take-away lanes are fixed offsets (achilles rod -> knee, knee spring -> shin, heel spring ->
    // foot crank: one lane up; foot crank and plantar rod -> foot), so the whole thing is 6 DPP row shifts and 6 adds per float: no LDS round trip.
    static_assert(TB_NDESC == nib(11, 10, 9, 0, 7, 0, 5, 4, 0, 1, 0, 0), "take-away lanes of the chain sum");
    const float cm1 = (lb == 4 || lb == 6 || lb == 9 || lb == 11) ? 1.f : 0.f, cm2 = lb == 11 ? 1.f : 0.f;
    // (the twelve scans of a chain sum advance STAGE BY STAGE: one scan after the other is a chain of dependent DPP adds with a hazard nop between each pair)
    auto chain_sum = [&](SVp& val) {
        f2 d[6] = {val.a.x, val.a.y, val.a.z, val.l.x, val.l.y, val.l.z};
        f2 s1[6], s2[6];
        sfor<0, 6>([&](auto I) { s1[I] = dpp2<0x111>(d[I]); });
        sfor<0, 6>([&](auto I) { s2[I] = dpp2<0x112>(d[I]); });
        sfor<0, 6>([&](auto I) { d[I] = d[I] - s1[I] * cm1 - s2[I] * cm2; });
        sfor<0, 6>([&](auto I) { d[I].x += dpp<0x111>(d[I].x); d[I].y += dpp<0x111>(d[I].y); });
        sfor<0, 6>([&](auto I) { d[I].x += dpp<0x112>(d[I].x); d[I].y += dpp<0x112>(d[I].y); });
        sfor<0, 6>([&](auto I) { d[I].x += dpp<0x114>(d[I].x); d[I].y += dpp<0x114>(d[I].y); });
        sfor<0, 6>([&](auto I) { d[I].x += dpp<0x118>(d[I].x); d[I].y += dpp<0x118>(d[I].y); });
        val.a = {d[0], d[1], d[2]}; val.l = {d[3], d[4], d[5]};
    };
    SVp vel = own, acc;
    chain_sum(vel);
    {
        vel = vel + lift(pvel);
        const SVp vp = {vel.a - own.a, vel.l - own.l};                  // parent body's velocity
        SVp t = {{splat(0.f), splat(0.f), splat(0.f)}, {splat(0.f), splat(0.f), splat(0.f)}};
        sfor<0, 3>([&](auto K) { t = t + crossMotion(vp, cdof[K]) * qd[K]; });
        acc = t;
    }
    chain_sum(acc);
    PROF2(14);
    acc = acc + lift(pacc);
    if (bl && hasj) sfor<0, 2>([&](auto Sd) {
        constexpr int sd = Sd;
        if (ball) sfor<0, 3>([&](auto K) {
            float* c = (float*)&S.W(WK_CDOF + 6 * (dadr + 13 * sd + K));
            c[0] = cdof[K].a.x[sd]; c[1] = cdof[K].a.y[sd]; c[2] = cdof[K].a.z[sd]; c[3] = cdof[K].l.x[sd]; c[4] = cdof[K].l.y[sd]; c[5] = cdof[K].l.z[sd];
        });
        else {
            float* c = (float*)&S.W(WK_CDOF + 6 * (dadr + 13 * sd));
            c[0] = cdof[2].a.x[sd]; c[1] = cdof[2].a.y[sd]; c[2] = cdof[2].a.z[sd]; c[3] = cdof[2].l.x[sd]; c[4] = cdof[2].l.y[sd]; c[5] = cdof[2].l.z[sd];
        }
    });
    PROF2(15);
    // ---- inertia + RNE force of the own bodies and of the pelvis
    SIp crb; SVp frc;
    body_inertia_force(mat, pos, o, Ib, ipos, mass, vel, acc, crb, frc);
    SI pcrb; SV pfrc;
    {
        float Ipel[9];
        sfor<0, 9>([&](auto K) { Ipel[K] = ct_body_inertia[9 + K]; });
        body_inertia_force(pmat, o, o, Ipel, cv3<1>(ct_body_ipos), S(F_MASS + 1), pvel, pacc, pcrb, pfrc);
    }
    PROF2(16);
    // ---- subtree sums (composite inertia, subtree force).  Depth-first numbering: every subtree interval [b, b + ndesc] ends at the foot (lane 11) except the leaves
    // (own value) and the foot crank (itself + the plantar rod).  So the subtree sum of a body on the path to the foot is the inclusive SUFFIX sum over lanes b..11 -
    // four row_shl adds, sums only (a prefix-DIFFERENCE form cancels in fp32: round 3) - and the others are one select each.  The shadow lanes 12..15 are massless
    // (above) and add exact zeros.  No LDS reads.
    {
        static_assert(TB_NDESC == nib(11, 10, 9, 0, 7, 0, 5, 4, 0, 1, 0, 0), "subtree intervals end at the foot, except leaves and the foot crank");
        const bool leafb = lb == 3 || lb == 5 || lb == 8 || lb == 10, crank = lb == 9;
        // sixteen pairs of suffix sums in two batches of eight, each stage by stage (one sum after the other is a chain of dependent DPP adds with hazard nops; all
        // sixteen at once keep 96 registers live and push the non-inlined stage past the callee-saved VGPRs it can park in AGPRs)
        f2 x0[16];
        x0[0] = crb.m; x0[1] = crb.h.x; x0[2] = crb.h.y; x0[3] = crb.h.z;
        sfor<0, 6>([&](auto K) { x0[4 + K] = crb.I[K]; });
        x0[10] = frc.a.x; x0[11] = frc.a.y; x0[12] = frc.a.z; x0[13] = frc.l.x; x0[14] = frc.l.y; x0[15] = frc.l.z;
        sfor<0, 2>([&](auto Bt) {
            constexpr int b0 = 8 * Bt;
            f2 sfx[8], two[8];
            sfor<0, 8>([&](auto I) { sfx[I].x = x0[b0 + I].x + dpp<0x101>(x0[b0 + I].x); sfx[I].y = x0[b0 + I].y + dpp<0x101>(x0[b0 + I].y); two[I] = sfx[I]; });
            sfor<0, 8>([&](auto I) { sfx[I].x += dpp<0x102>(sfx[I].x); sfx[I].y += dpp<0x102>(sfx[I].y); });
            sfor<0, 8>([&](auto I) { sfx[I].x += dpp<0x104>(sfx[I].x); sfx[I].y += dpp<0x104>(sfx[I].y); });
            sfor<0, 8>([&](auto I) { sfx[I].x += dpp<0x108>(sfx[I].x); sfx[I].y += dpp<0x108>(sfx[I].y); });
            sfor<0, 8>([&](auto I) { x0[b0 + I] = sel2(crank, two[I], sel2(leafb, x0[b0 + I], sfx[I])); });
            if constexpr (Bt == 0) __builtin_amdgcn_sched_barrier(0);
        });
        // the composite record: pairs are adjacent words, 128-bit stores
        float* p = xb + xrec;
        sfor<0, 16>([&](auto I) { p[2 * I] = x0[I].x; p[2 * I + 1] = x0[I].y; });
        if constexpr (!QPOS0) {      // the body's COM relative to o: where an external wrench on this body acts (mjData.xfrc_applied, below)
            const V3p cr = (pos - lift(o)) + mul(mat, ipos);
            p[32] = cr.x.x; p[33] = cr.x.y; p[34] = cr.y.x; p[35] = cr.y.y; p[36] = cr.z.x; p[37] = cr.z.y;
        }
    }
    wsync();
    PROF2(17);
    {   // pelvis composite = own + the two hip-roll subtrees (record 0, both slots)
        sfor<0, 2>([&](auto Sd) {
            const float* p = xb + Sd;
            pcrb.m += p[0]; pcrb.h = pcrb.h + V3{p[2], p[4], p[6]};
            sfor<0, 6>([&](auto K) { pcrb.I[K] += p[8 + 2 * K]; });
            pfrc.a = pfrc.a + V3{p[20], p[22], p[24]}; pfrc.l = pfrc.l + V3{p[26], p[28], p[30]};
        });
    }
    // ---- collision geoms the constraint stage does not instantiate (pelvis sphere cassie.xml:87, hip-pitch capsules :101,164): only
    // tested against the floor plane here; a hit is reported through I_SAT (SAT_BODY_FLOOR) by the constraint stage
    if constexpr (!QPOS0) {
        const V3 fn = {S(F_FLOOR), S(F_FLOOR + 1), S(F_FLOOR + 2)}, p0 = {ct_floor_pos[0], ct_floor_pos[1], ct_floor_pos[2]};
        static_assert(ct_geom_body[8] == 1 && ct_geom_body[6] == 4 && ct_geom_body[7] == 16, "pelvis sphere, hip-pitch capsules");
        // (branch-free: every lane evaluates both tests on its own bodies, lane 0 keeps the sphere result and lane 2 = hip pitch the capsule result)
        const float hitp = dot(o + mul(pmat, cv3<8>(ct_geom_pos)) - p0, fn) - ct_geom_radius[8] < 0.f ? 1.f : 0.f;
        constexpr V3 gp0 = cv3<6>(ct_geom_pos), gp1 = cv3<7>(ct_geom_pos), ga0 = cv3<6>(ct_geom_axis), ga1 = cv3<7>(ct_geom_axis);
        const V3p c = pos + mul(mat, V3p{f2{gp0.x, gp1.x}, f2{gp0.y, gp1.y}, f2{gp0.z, gp1.z}});
        const V3p ax = mul(mat, V3p{f2{ga0.x, ga1.x}, f2{ga0.y, ga1.y}, f2{ga0.z, ga1.z}}) * f2{ct_geom_half[6], ct_geom_half[7]};
        const f2 da = dot(fn, c + ax - lift(p0)), db = dot(fn, c - ax - lift(p0));
        const float d0 = fminf(da.x, db.x) - ct_geom_radius[6], d1 = fminf(da.y, db.y) - ct_geom_radius[7];
        const float hitc = (d0 < 0.f || d1 < 0.f) ? 1.f : 0.f;
        int ho = WK_DUMMY; ho = l == 0 ? WK_MISC + 4 : ho; ho = l == 2 ? WK_MISC + 5 : ho;
        S.W(ho) = l == 0 ? hitp : hitc;
        // the points themselves for the complete-row path (cassie_complete.h): hip-pitch capsule ends (left e0, e1, right e0, e1) from the hip-pitch lane, the pelvis
        // sphere centre from lane 0, 15 words of the row store that neither the exchange records nor the row stage's parked vectors touch
        constexpr int XE = 38 * 16;
#ifndef APX_NO_TREE_EXTRA      /* A/B build */
        if (l == 2) {
            const V3p e0 = c + ax, e1 = c - ax;
            float* q = xb + XE;
            q[0] = e0.x.x; q[1] = e0.y.x; q[2] = e0.z.x; q[3] = e1.x.x; q[4] = e1.y.x; q[5] = e1.z.x;
            q[6] = e0.x.y; q[7] = e0.y.y; q[8] = e0.z.y; q[9] = e1.x.y; q[10] = e1.y.y; q[11] = e1.z.y;
        }
        if (l == 0) { const V3 pc = o + mul(pmat, cv3<8>(ct_geom_pos)); float* q = xb + XE + 12; q[0] = pc.x; q[1] = pc.y; q[2] = pc.z; }
#endif
    }
    // ---- anchor points, capsule ends, foot pose (body lanes that own them)
    {
        auto put2 = [&](int off, V3p p) {      // leg slot sd's point at off + 30 sd
            S.W(off) = p.x.x; S.W(off + 1) = p.y.x; S.W(off + 2) = p.z.x; S.W(off + 30) = p.x.y; S.W(off + 31) = p.y.y; S.W(off + 32) = p.z.y;
        };
        constexpr int base = WK_PTS;
        if constexpr (QPOS0) {      // world COM of the constraint bodies, slot order of c2::cslot: achilles, heel-spring, plantar-rod, foot, tarsus, shin; then hip pitch
            const int cs = l == 3 ? 0 : l == 8 ? 1 : l == 10 ? 2 : l == 11 ? 3 : l == 7 ? 4 : l == 6 ? 5 : l == 2 ? 6 : -1;      // (6: hip pitch, for its invweight)
            if (cs >= 0) put2(base + 3 * cs, pos + mul(mat, ipos));
        } else {
        static_assert(ct_eq_body1[0] == 12 && ct_eq_body2[0] == 13 && ct_eq_body1[1] == 5 && ct_eq_body2[1] == 10, "connect bodies");
        static_assert(ct_geom_body[0] == 13 && ct_geom_body[2] == 9 && ct_geom_body[4] == 8, "capsule bodies");
        auto pairc = [](V3 a, V3 b) { return V3p{f2{a.x, b.x}, f2{a.y, b.y}, f2{a.z, b.z}}; };
        // Branch-free (round 3; the six `if (l == ..)` blocks per leg were six exec-mask regions each): every lane transforms ONE anchor and ONE capsule with
        // constants picked by select chains and stores them to its slots, or to the dummy words when it owns none.
        {   // connect anchors: plantar rod (10) eq 0 anchor 1, foot (11) eq 0 anchor 2, achilles rod (3) eq 1 anchor 1, heel spring (8) eq 1 anchor 2
            const V3p a10 = pairc(cv3<0>(ct_eq_anchor1), cv3<2>(ct_eq_anchor1)), a11 = pairc(cv3<0>(ct_eq_anchor2), cv3<2>(ct_eq_anchor2)),
                      a3 = pairc(cv3<1>(ct_eq_anchor1), cv3<3>(ct_eq_anchor1)), a8 = pairc(cv3<1>(ct_eq_anchor2), cv3<3>(ct_eq_anchor2));
            V3p ap = a10; int ao = base + 0;
            ap = sel2(l == 11, a11, ap); ao = l == 11 ? base + 3 : ao;
            ap = sel2(l == 3, a3, ap); ao = l == 3 ? base + 6 : ao;
            ap = sel2(l == 8, a8, ap); ao = l == 8 ? base + 9 : ao;
            const bool has = l == 10 || l == 11 || l == 3 || l == 8;
            put2(has ? ao : WK_DUMMY, pos + mul(mat, ap));
        }
        {   // capsules: foot (11), tarsus (7), shin (6)
            const V3p p11 = pairc(cv3<0>(ct_geom_pos), cv3<1>(ct_geom_pos)), x11 = pairc(cv3<0>(ct_geom_axis), cv3<1>(ct_geom_axis)),
                      p7 = pairc(cv3<2>(ct_geom_pos), cv3<3>(ct_geom_pos)), x7 = pairc(cv3<2>(ct_geom_axis), cv3<3>(ct_geom_axis)),
                      p6 = pairc(cv3<4>(ct_geom_pos), cv3<5>(ct_geom_pos)), x6 = pairc(cv3<4>(ct_geom_axis), cv3<5>(ct_geom_axis));
            V3p cp = p11, cx = x11; f2 ch = {ct_geom_half[0], ct_geom_half[1]}; int co = base + 12;
            cp = sel2(l == 7, p7, cp); cx = sel2(l == 7, x7, cx); ch = sel2(l == 7, f2{ct_geom_half[2], ct_geom_half[3]}, ch); co = l == 7 ? base + 18 : co;
            cp = sel2(l == 6, p6, cp); cx = sel2(l == 6, x6, cx); ch = sel2(l == 6, f2{ct_geom_half[4], ct_geom_half[5]}, ch); co = l == 6 ? base + 24 : co;
            const bool has = l == 11 || l == 7 || l == 6;
            const V3p c = pos + mul(mat, cp), ax = mul(mat, cx) * ch;
            put2(has ? co : WK_DUMMY, c + ax); put2(has ? co + 3 : WK_DUMMY, c - ax);
        }
        }
    }
    PROF2(18);
    // ---- mass-matrix rows and bias forces, dof lanes: k = 0..12 -> leg dof k of both legs; 13..15 -> pelvis dofs (l-13, l-10)
    SVp cdv, fv; int madrv[2];
    {
        const bool leg = l < 13;
        const int d0 = leg ? 6 + l : l - 13, d1 = leg ? 19 + l : l - 10;
        const int dbl = l == 12 ? 11 : nibble(TD_BODY, leg ? l : 0);      // leg-local body of the lane's dof
        SIp c; SVp fsub;
        {   // (selects, not a branch around a struct copy: the copy form kept the pelvis composite in a 40-byte stack object = scratch.  The loads are
            // pinned by an empty asm: clang otherwise predicates each of the loads on `leg`, one exec-mask region per word)
            const float* p = xb + XB_SZ * dbl;
            f2 w[16];
            sfor<0, 16>([&](auto K) { w[K] = f2{p[2 * K], p[2 * K + 1]}; });
            APX_PIN("+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]), "+v"(w[4]), "+v"(w[5]), "+v"(w[6]), "+v"(w[7]), "+v"(w[8]), "+v"(w[9]), "+v"(w[10]),
                         "+v"(w[11]), "+v"(w[12]), "+v"(w[13]), "+v"(w[14]), "+v"(w[15]));
            c.m = sel2(leg, w[0], splat(pcrb.m)); c.h = {sel2(leg, w[1], splat(pcrb.h.x)), sel2(leg, w[2], splat(pcrb.h.y)), sel2(leg, w[3], splat(pcrb.h.z))};
            sfor<0, 6>([&](auto K) { c.I[K] = sel2(leg, w[4 + K], splat(pcrb.I[K])); });
            fsub = {{sel2(leg, w[10], splat(pfrc.a.x)), sel2(leg, w[11], splat(pfrc.a.y)), sel2(leg, w[12], splat(pfrc.a.z))},
                    {sel2(leg, w[13], splat(pfrc.l.x)), sel2(leg, w[14], splat(pfrc.l.y)), sel2(leg, w[15], splat(pfrc.l.z))}};
        }
        const float* cp0 = (const float*)&S.W(WK_CDOF + 6 * d0);
        const float* cp1 = (const float*)&S.W(WK_CDOF + 6 * d1);
        const SVp cd = {{f2{cp0[0], cp1[0]}, f2{cp0[1], cp1[1]}, f2{cp0[2], cp1[2]}}, {f2{cp0[3], cp1[3]}, f2{cp0[4], cp1[4]}, f2{cp0[5], cp1[5]}}};
        const SVp f = imul(c, cd);
        const int dep0 = leg ? (l == 12 ? 13 : nibble(TD_DEPTH, l)) : d0 + 1, dep1 = leg ? dep0 : d1 + 1;
        const int madr0 = cti(CT_MADR + d0), madr1 = cti(CT_MADR + d1);
        {
            const f2 dg = sdot(cd, f) + f2{ctf(CT_ARM + d0), ctf(CT_ARM + d1)};
            S.W(WK_M + madr0) = dg.x; S.W(WK_M + madr1) = dg.y;
        }
        // off-diagonal entries M[d][anc] = cdof_anc . f_d.  The pelvis ancestors' axes are known to every lane (unit translations, the columns of the
        // pelvis rotation about o itself), so their six entries need no operand fetch at all; the leg ancestors follow below, both legs together.
        sfor<0, 6>([&](auto Pp) {
            constexpr int p = Pp;
            f2 v;
            if constexpr (p < 3) v = p == 0 ? f.l.x : p == 1 ? f.l.y : f.l.z; else v = dot(col(pmat, p - 3), f.a);
            S.W(p < dep0 - 1 ? WK_M + madr0 + dep0 - 1 - p : WK_DUMMY) = v.x;
            S.W(p < dep1 - 1 ? WK_M + madr1 + dep1 - 1 - p : WK_DUMMY) = v.y;
        });
        cdv = cd; fv = f; madrv[0] = madr0; madrv[1] = madr1;
        // qfrc_smooth = passive - bias + actuation
        const int k = l;       // leg-local dof
        f2 fs = -(f2{S(F_DAMP + d0), S(F_DAMP + d1)} * f2{S(F_QVEL + d0), S(F_QVEL + d1)}) - sdot(cd, fsub);
        // (branch-free: every lane reads, the coefficient selects)
        fs -= qp2(ct_jnt_qposadr[9]) * (l == 7 ? ct_jnt_stiffness[9] : 0.f);            // shin spring
        fs -= qp2(ct_jnt_qposadr[11]) * (l == 9 ? ct_jnt_stiffness[11] : 0.f);          // heel spring
        static_assert(ct_jnt_stiffness[20] == ct_jnt_stiffness[9] && ct_jnt_stiffness[22] == ct_jnt_stiffness[11] && ct_jnt_qposadr[20] == ct_jnt_qposadr[9] + 14, "springs");
        {   // actuated dofs: hip roll, yaw, pitch (k = 0, 1, 2), knee (6), foot (12); the other lanes read drive 0 with gear 0
            int u = 0; u = k == 1 ? 1 : u; u = k == 2 ? 2 : u; u = k == 6 ? 3 : u; u = k == 12 ? 4 : u;
            const bool act = ((0x1047u >> k) & 1u) != 0u;
            const f2 cmax = {ctf(CT_CMAX + u), ctf(CT_CMAX + u + 5)}, g = f2{ctf(CT_GEAR + u), ctf(CT_GEAR + u + 5)} * (act ? 1.f : 0.f);
            const f2 ct = {S.W(WK_CTRL + u), S.W(WK_CTRL + u + 5)};
            fs += g * f2{fminf(fmaxf(ct.x, -cmax.x), cmax.x), fminf(fmaxf(ct.y, -cmax.y), cmax.y)};
        }
        if constexpr (!QPOS0) {
            // external wrench (one row of mjData.xfrc_applied: body I_XBODY, applied at that body's COM): J^T (f, tau) on the dofs of the body's ancestor
            // chain.  The six free-joint dofs always see it; a leg dof sees it when its body is an ancestor-or-self of the pushed body, which with the
            // depth-first numbering is the interval test  body(dof) <= pushed <= body(dof) + ndesc.  Branch-free: a 0 / 1 weight.
            const int xbd = S.I(I_XBODY);                       // 0 / 1: pelvis (the harnesses' default), 2..25: a leg body
            const bool xleg = xbd >= 2;
            const int xsd = xbd >= 14 ? 1 : 0, xlb = xbd - 2 - 12 * xsd;
            const V3 xf = {S(F_XFRC), S(F_XFRC + 1), S(F_XFRC + 2)}, xt = {S(F_XFRC + 3), S(F_XFRC + 4), S(F_XFRC + 5)};
            const float* xr = xb + XB_SZ * (xleg ? xlb : 0) + 32 + xsd;
            const V3 rpp = mul(pmat, V3{cm_body_ipos[3], cm_body_ipos[4], cm_body_ipos[5]});      // pelvis: xipos - o
            const V3 rp = {xleg ? xr[0] : rpp.x, xleg ? xr[2] : rpp.y, xleg ? xr[4] : rpp.z};
            const int dnd = nibble(TB_NDESC, dbl);
            const bool inchain = xleg && dbl <= xlb && xlb <= dbl + dnd;
            const f2 hit = {(l >= 13 || (inchain && xsd == 0)) ? 1.f : 0.f, (l >= 13 || (inchain && xsd == 1)) ? 1.f : 0.f};
            fs += hit * (dot(xt + cross(rp, xf), cd.a) + dot(xf, cd.l));
        }
        S.W(WK_SMOOTH + d0) = fs.x; S.W(WK_SMOOTH + d1) = fs.y;
    }
    // leg ancestors: a UNIFORM loop over the nine leg dofs that have descendants (hip roll / yaw / pitch, the first two achilles-rod axes, knee, shin,
    // tarsus, foot crank).  Lane j's axis reaches the row as the DPP row-broadcast SOURCE of the multiply-adds of cdof_j . f_own (v_mul_f32_dpp + 5 v_fmac_f32_dpp per
    // slot: rounds 3-4 issued six v_mov_b32_dpp and six v_fma per slot), and a per-lane level table says where (or whether: level 0 = not an ancestor, the store goes
    // to the dummy word) the entry belongs in the ancestor-chain layout.
    {
        constexpr int ANC[9] = {0, 1, 2, 3, 4, 6, 7, 8, 10};
        const int l4 = 4 * l;
        float ca[2][6], fa[2][6];
        sfor<0, 2>([&](auto Sd) {
            constexpr int sd = Sd;
            ca[sd][0] = cdv.a.x[sd]; ca[sd][1] = cdv.a.y[sd]; ca[sd][2] = cdv.a.z[sd]; ca[sd][3] = cdv.l.x[sd]; ca[sd][4] = cdv.l.y[sd]; ca[sd][5] = cdv.l.z[sd];
            fa[sd][0] = fv.a.x[sd]; fa[sd][1] = fv.a.y[sd]; fa[sd][2] = fv.a.z[sd]; fa[sd][3] = fv.l.x[sd]; fa[sd][4] = fv.l.y[sd]; fa[sd][5] = fv.l.z[sd];
        });
        // the axes are read through DPP by inline asm the hazard recogniser cannot see: no compiler-generated definition right in front of the first read
        APX_HAZARD_FENCE("+v"(ca[0][0]), "+v"(ca[0][1]), "+v"(ca[0][2]), "+v"(ca[0][3]), "+v"(ca[0][4]), "+v"(ca[0][5]),
                                 "+v"(ca[1][0]), "+v"(ca[1][1]), "+v"(ca[1][2]), "+v"(ca[1][3]), "+v"(ca[1][4]), "+v"(ca[1][5]));
        sfor<0, 9>([&](auto Jn) {
            constexpr int j = ANC[Jn];
            constexpr unsigned long long T = crba_level_table(j);
            const int a = (int)((T >> l4) & 15ull);
            sfor<0, 2>([&](auto Sd) {
                constexpr int sd = Sd;
                float v = mul_bcast<j>(ca[sd][0], fa[sd][0]);
                sfor<1, 6>([&](auto K) { fmac_bcast<j>(v, ca[sd][K], fa[sd][K]); });
                S.W(a ? WK_M + madrv[sd] + a : WK_DUMMY) = v;
            });
        });
    }
    PROF2(19);
    // ---- foot pose for the reward / foot velocity (cassie.py:328-331,426-427)
    if (!QPOS0 && l == 11) sfor<0, 2>([&](auto Sd) {
        constexpr int sd = Sd;
        S(F_FWD + 2 + 4 * sd) = quat.w[sd]; S(F_FWD + 3 + 4 * sd) = quat.x[sd]; S(F_FWD + 4 + 4 * sd) = quat.y[sd]; S(F_FWD + 5 + 4 * sd) = quat.z[sd];
        S(F_FWD + 10 + 3 * sd) = pos.x[sd]; S(F_FWD + 11 + 3 * sd) = pos.y[sd]; S(F_FWD + 12 + 3 * sd) = pos.z[sd] - 0.0550841220316708f;
    });
}

// ------------------------------------------------------------------------------------------------ factor / solve, dof lanes
// M = L^T D L (MuJoCo's reverse Cholesky on the kinematic tree), lane-parallel over the 13 dofs of a leg (slot 0 = left,
// 1 = right); the 6 pelvis dofs are carried uniformly by every lane.  A lane keeps its row of the (symmetric) leg block by
// IDENTITY of the other dof (R[j], j = 0..12, zero when j is neither ancestor nor descendant) plus the 6 pelvis couplings, so
// eliminating dof k is: broadcast row k (lane k), every lane i < k does R_i[j] -= (R_i[k] / D_k) * R_k[j].  The elimination
// leaves D_i L[i][j] below the diagonal and L[j][i] above it, i.e. both the row and the column view the solves need.
constexpr bool leg_anc(int k, int j) {       // is leg dof j a proper ancestor of leg dof k
    for (int a = 1; a < ct_dof_depth[6 + k]; ++a) if (ct_dof_anc[16 * (6 + k) + a] == 6 + j) return true;
    return false;
}
struct MIdx { unsigned short v[16 * 13]; };
// A structural zero of the mass matrix is the offset of a word that HOLDS zero (round 5; it was the marker 0xFFFF, a clamped address and a select per entry and slot):
// WK_FZERO for the left leg's block and WK_FZERO + M_LEGSZ for the right one's, i.e. the same block-relative offset MI_ZERO for both.  The two words sit in the factor
// hand-off WK_LD (only mj_setConst stores a factor there: fac_store) and in the unused tail of WK_ZP2; factor_lane clears them before it loads.
constexpr int WK_FZERO = WK_ZP2 + 10 - (ct_dof_madr[19] - ct_dof_madr[6]), MI_ZERO = WK_FZERO - (WK_M + ct_dof_madr[6]);
static_assert(WK_FZERO >= WK_LD && WK_FZERO < WK_LD + NM && WK_ZP2 + 10 < WK_TOTAL && MI_ZERO > 0 && MI_ZERO < 0xFFFF, "zero words of the factor loads");
constexpr MIdx make_midx() {            // offset of M[k][j] inside a leg's block of the ancestor-chain layout, MI_ZERO = structural zero
    MIdx t{};
    for (int k = 0; k < 16; ++k)
        for (int j = 0; j < 13; ++j) {
            int off = MI_ZERO;
            if (k < 13) {
                const int lo = k > j ? k : j, hi = k > j ? j : k;      // the entry is stored in the row of the deeper dof
                for (int a = 0; a < ct_dof_depth[6 + lo]; ++a)
                    if (ct_dof_anc[16 * (6 + lo) + a] == 6 + hi) off = ct_dof_madr[6 + lo] + a - ct_dof_madr[6];
            }
            t.v[13 * k + j] = (unsigned short)off;
        }
    return t;
}
// The wave-constant table (env_state.h CT_*), built at compile time and copied HBM -> LDS once per launch.
struct CTab { unsigned v[CT_TOTAL]; };
constexpr CTab make_ct() {
    CTab t{};
    auto fb = [](float x) { return __builtin_bit_cast(unsigned, x); };
    for (int b = 2; b < 26; ++b) {      // (left, right) pair layout: env_state.h ct_body_word
        const int sd = b >= 14 ? 1 : 0, lb = b - 2 - 12 * sd;
        for (int k = 0; k < 3; ++k) { t.v[ct_body_word(lb, sd, k)] = fb(ct_body_pos[3 * b + k]); t.v[ct_body_word(lb, sd, 3 + k)] = fb(ct_body_ipos[3 * b + k]); }
        for (int k = 0; k < 4; ++k) t.v[ct_body_word(lb, sd, 6 + k)] = fb(ct_body_quat[4 * b + k]);
        for (int k = 0; k < 9; ++k) t.v[ct_body_word(lb, sd, 10 + k)] = fb(ct_body_inertia[9 * b + k]);
    }
    for (int d = 0; d < 32; ++d) { t.v[CT_MADR + d] = (unsigned)ct_dof_madr[d]; t.v[CT_ARM + d] = fb(ct_dof_armature[d]); }
    for (int u = 0; u < 10; ++u) { t.v[CT_GEAR + u] = fb(ct_act_gear[u]); t.v[CT_CMAX + u] = fb(ct_act_ctrlmax[u]); }
    const MIdx m = make_midx();
    for (int l = 0; l < 16; ++l)
        for (int w = 0; w < 7; ++w)
            t.v[CT_MIDX + 7 * l + w] = (unsigned)m.v[13 * l + 2 * w] | (2 * w + 1 < 13 ? (unsigned)m.v[13 * l + 2 * w + 1] << 16 : 0u);
    return t;
}
__device__ const CTab kCT = make_ct();
__device__ __forceinline__ void ct_fill() {                 // all 64 lanes of the workgroup's single wave
    lint* dst = (lint*)apx_lds4 + L4_EPW * L4_ES;
    for (int i = threadIdx.x; i < CT_TOTAL; i += 64) dst[i] = (int)kCT.v[i];
    wsync();
}
constexpr int M_LEG0 = ct_dof_madr[6], M_LEGSZ = ct_dof_madr[19] - ct_dof_madr[6];
static_assert(M_LEG0 == 21 && M_LEGSZ == 143 && CM_NM == M_LEG0 + 2 * M_LEGSZ, "mass-matrix layout");

struct LaneFac {
    float Lr[2][13], Lc[2][13], w[2][6], D[2], invD[2];      // row view L[me][j], column view L[i][me], pelvis couplings L[me][p]
    float Lp[6][6], Dp[6], invDp[6];                          // pelvis block (uniform): L[i][j] for j < i
};
struct LaneVec { float a[2]; float p[6]; };                   // a[slot] = this lane's leg dof, p = pelvis dofs (uniform)
// (see whiten_regs)
struct FacRegs {
    float Lr[2][13], w[2][6], disq[2];       // L[me][leg dof j], L[me][pelvis dof p], 1/sqrt(D_me)
    float Lp[6][6], disqp[6];                // pelvis block L[i][j] (j < i), 1/sqrt(D_p): uniform
    LaneVec qs, qv, qw;                      // qacc_smooth, qvel, qacc_warmstart: a[slot] = this lane's leg dof, p = pelvis dofs (uniform)
};
struct LaneIdx { int l, own, dep; unsigned short mi[13]; };   // own = offset of the lane's diagonal entry inside the leg block
__device__ __forceinline__ LaneIdx lane_idx() {
    LaneIdx x;
    x.l = threadIdx.x & 15;
    const int lc = x.l < 13 ? x.l : 12;
    x.own = cti(CT_MADR + 6 + lc) - M_LEG0;
    x.dep = lc == 12 ? 13 : nibble(TD_DEPTH, lc);
    sfor<0, 7>([&](auto Wd) {
        const unsigned w = (unsigned)cti(CT_MIDX + 7 * x.l + Wd);
        x.mi[2 * Wd] = (unsigned short)(w & 0xFFFFu);
        if constexpr (2 * Wd + 1 < 13) x.mi[2 * Wd + 1] = (unsigned short)(w >> 16);
    });
    return x;
}

// factorise WK_M (DAMP: + hdamp * joint damping on the diagonal: mj_Euler's implicit damping; the lane adds it to ITS diagonal word of WK_M in place - the tree stage
// rewrites the whole matrix every substep - so that the row loads below bring it along: a `l == J ? hd : 0` per entry was 26 selects + 26 adds per factorisation)
template <bool DAMP>
__device__ __forceinline__ void factor_lane(const St& S, const LaneIdx& X, float hdamp, LaneFac& F) {
    const int l = X.l;
    float R[2][13], P[2][6], dg[2];
    S.W(WK_FZERO) = 0.f; S.W(WK_FZERO + M_LEGSZ) = 0.f;      // (every lane, same words: no exec-mask region)
    if constexpr (DAMP) {
        float d[2];
        sfor<0, 2>([&](auto Sd) { d[Sd] = S.W(WK_M + M_LEG0 + M_LEGSZ * Sd + X.own) + hdamp * S(F_DAMP + 6 + 13 * Sd + (l < 13 ? l : 12)); });
        APX_LOCKSTEP();      // (pelvis lanes 13..15 read lane 12's diagonal word, for a result that goes to the sink)
        sfor<0, 2>([&](auto Sd) { S.W(l < 13 ? WK_M + M_LEG0 + M_LEGSZ * Sd + X.own : WK_DUMMY) = d[Sd]; });
        wsync();
    }
    sfor<0, 2>([&](auto Sd) {
        constexpr int sd = Sd, blk = WK_M + M_LEG0 + M_LEGSZ * sd;
        sfor<0, 13>([&](auto J) { R[sd][J] = S.W(blk + X.mi[J]); });      // a structural zero loads the zero word
        sfor<0, 6>([&](auto Pp) { const float v = S.W(blk + X.own + X.dep - 1 - Pp); P[sd][Pp] = l < 13 ? v : 0.f; });
        // the lane's own diagonal entry is carried separately (dg -= (R_l[k] / D_k) R_l[k] at every step): reading R[sd][l] back at the end
        // is a 13-way select on the lane index, which the compiler turns into a tree of divergent branches
        dg[sd] = S.W(blk + X.own);
    });
    // ---- legs: eliminate dof 12 .. 0 of both legs at once.  An eliminated entry is x -= t * bcast_k(x): written as v_fmac_f32 with the broadcast as its
    // DPP source operand and t negated by the source modifier (ONE instruction; the compiler's own selection is v_mov_b32_dpp + v_fma_f32 with a neg modifier, its DPP combiner does not take
    // tied-accumulator instructions).  Inline asm is invisible to the hazard recogniser, so the distances are kept by construction: the statements are
    // volatile (program order), a register written by one of them is read through DPP again no sooner than 6 statements later (order per step: next pivot's
    // column, the other columns, pelvis couplings of leg 0, next pivot's reciprocal, pelvis couplings of leg 1), the fence below claims every register so
    // that no compiler-generated definition can sit right in front of its first DPP read, and the reciprocal carries its own wait state for the
    // transcendental-forwarding rule of gfx940+.
#define APX_FENCE19(sd) APX_HAZARD_FENCE("+v"(R[sd][0]), "+v"(R[sd][1]), "+v"(R[sd][2]), "+v"(R[sd][3]), "+v"(R[sd][4]), "+v"(R[sd][5]), "+v"(R[sd][6]), \
        "+v"(R[sd][7]), "+v"(R[sd][8]), "+v"(R[sd][9]), "+v"(R[sd][10]), "+v"(R[sd][11]), "+v"(R[sd][12]),                                               \
        "+v"(P[sd][0]), "+v"(P[sd][1]), "+v"(P[sd][2]), "+v"(P[sd][3]), "+v"(P[sd][4]), "+v"(P[sd][5]))
    APX_FENCE19(0); APX_FENCE19(1);
    float inv[2];
    sfor<0, 2>([&](auto Sd) { inv[Sd] = rcp_bcast<12>(R[Sd][12]); });
    srfor<0, 13>([&](auto K) {
        constexpr int k = K;
        float tmp[2];
        sfor<0, 2>([&](auto Sd) {
            constexpr int sd = Sd;
            tmp[sd] = l < k ? R[sd][k] * inv[sd] : 0.f;
            dg[sd] -= tmp[sd] * R[sd][k];
        });
        if constexpr (k >= 1) sfor<0, 2>([&](auto Sd) { if constexpr (leg_anc(k, k - 1)) fnmac_bcast<k>(R[Sd][k - 1], tmp[Sd]); });
        srfor<0, (k >= 1 ? k - 1 : 0)>([&](auto J) { constexpr int j = J; sfor<0, 2>([&](auto Sd) { if constexpr (leg_anc(k, j)) fnmac_bcast<k>(R[Sd][j], tmp[Sd]); }); });
        sfor<0, 6>([&](auto Pp) { fnmac_bcast<k>(P[0][Pp], tmp[0]); });
        if constexpr (k >= 1) sfor<0, 2>([&](auto Sd) { inv[Sd] = rcp_bcast<k - 1>(R[Sd][k - 1]); });
        sfor<0, 6>([&](auto Pp) { fnmac_bcast<k>(P[1][Pp], tmp[1]); });
        sfor<0, 2>([&](auto Sd) { R[Sd][k] = l < k ? tmp[Sd] : R[Sd][k]; });
    });
    APX_FENCE19(0); APX_FENCE19(1);      // ... and no compiler-generated DPP read right behind the last write
#undef APX_FENCE19
    sfor<0, 2>([&](auto Sd) {
        constexpr int sd = Sd;
        const float D = l < 13 ? dg[sd] : 1.f;
        F.D[sd] = D; F.invD[sd] = rcpf(D);
        sfor<0, 13>([&](auto J) { F.Lr[sd][J] = (J < l && l < 13) ? R[sd][J] * F.invD[sd] : 0.f; F.Lc[sd][J] = J > l ? R[sd][J] : 0.f; });
        sfor<0, 6>([&](auto Pp) { F.w[sd][Pp] = P[sd][Pp] * F.invD[sd]; });
    });
    // ---- pelvis block: Schur complement of the two legs, then a 6x6 factorisation carried by every lane
    float Pm[6][6];
    sfor<0, 6>([&](auto Pi) {
        sfor<0, Pi + 1>([&](auto Qi) {
            constexpr int p = Pi, q = Qi;
            float m = S.W(WK_M + ct_dof_madr[p] + (p - q));
            if constexpr (DAMP && p == q) m += hdamp * S(F_DAMP + p);
            Pm[p][q] = m - red16(F.w[0][q] * P[0][p] + F.w[1][q] * P[1][p]);
        });
    });
    srfor<0, 6>([&](auto K) {
        constexpr int k = K;
        F.Dp[k] = Pm[k][k]; F.invDp[k] = rcpf(Pm[k][k]);
        sfor<0, k>([&](auto I) {
            constexpr int i = I;
            const float tmp = Pm[k][i] * F.invDp[k];
            sfor<0, i + 1>([&](auto J) { Pm[i][J] -= tmp * Pm[k][J]; });
            F.Lp[k][i] = tmp;
        });
    });
}
// factor -> WK_LD (ancestor-chain layout, read by the row stage) and WK_DISQ
__device__ __forceinline__ void fac_store(const St& S, const LaneIdx& X, const LaneFac& F) {
    const int l = X.l;
    sfor<0, 2>([&](auto Sd) {      // branch-free: stores that do not apply go to the env's dummy word
        constexpr int sd = Sd, blk = WK_LD + M_LEG0 + M_LEGSZ * sd;
        const bool leg = l < 13;
        S.W(leg ? blk + X.own : WK_DUMMY) = F.D[sd];
        sfor<0, 12>([&](auto J) { S.W((J < l && leg && X.mi[J] != MI_ZERO) ? blk + X.mi[J] : WK_DUMMY) = F.Lr[sd][J]; });
        sfor<0, 6>([&](auto Pp) { S.W(leg ? blk + X.own + X.dep - 1 - Pp : WK_DUMMY) = F.w[sd][Pp]; });
        S.W(leg ? WK_DISQ + 6 + 13 * sd + l : WK_DUMMY) = rsqrtf(F.D[sd]);
    });
    if (l == 0) sfor<0, 6>([&](auto Pi) {
        constexpr int p = Pi;
        S.W(WK_LD + ct_dof_madr[p]) = F.Dp[p]; S.W(WK_DISQ + p) = rsqrtf(F.Dp[p]);
        sfor<0, p>([&](auto Qi) { S.W(WK_LD + ct_dof_madr[p] + (p - Qi)) = F.Lp[p][Qi]; });
    });
}
__device__ __forceinline__ LaneVec vec_load(const St& S, const LaneIdx& X, int off_state /* -1 = workspace */, int off) {
    LaneVec x;
    const int lc = X.l < 13 ? X.l : 12;
    sfor<0, 2>([&](auto Sd) { x.a[Sd] = off_state >= 0 ? S(off_state + 6 + 13 * Sd + lc) : S.W(off + 6 + 13 * Sd + lc); });
    sfor<0, 6>([&](auto Pp) { x.p[Pp] = off_state >= 0 ? S(off_state + Pp) : S.W(off + Pp); });
    if (X.l >= 13) x.a[0] = x.a[1] = 0.f;
    return x;
}
// x <- L^-T x (leaves -> root)
__device__ __forceinline__ void solve_LT_lane(const LaneFac& F, LaneVec& x) {
    solve_fence(x.a[0], x.a[1]);
    srfor<0, 13>([&](auto I) { constexpr int i = I; solve_step2<i>(x.a[0], x.a[1], F.Lc[0][i], F.Lc[1][i]); });
    solve_fence(x.a[0], x.a[1]);
    sfor<0, 6>([&](auto Pp) { x.p[Pp] -= red16(F.w[0][Pp] * x.a[0] + F.w[1][Pp] * x.a[1]); });
    srfor<1, 6>([&](auto I) { constexpr int i = I; sfor<0, i>([&](auto J) { x.p[J] -= F.Lp[i][J] * x.p[i]; }); });
}
// x <- L^-1 x (root -> leaves)
__device__ __forceinline__ void solve_L_lane(const LaneFac& F, LaneVec& x) {
    sfor<1, 6>([&](auto I) { constexpr int i = I; sfor<0, i>([&](auto J) { x.p[i] -= F.Lp[i][J] * x.p[J]; }); });
    sfor<0, 2>([&](auto Sd) { sfor<0, 6>([&](auto Pp) { x.a[Sd] -= F.w[Sd][Pp] * x.p[Pp]; }); });
    solve_fence(x.a[0], x.a[1]);
    sfor<0, 13>([&](auto Jj) { constexpr int j = Jj; solve_step2<j>(x.a[0], x.a[1], F.Lr[0][j], F.Lr[1][j]); });
    solve_fence(x.a[0], x.a[1]);
}
// y = L^T x
__device__ __forceinline__ LaneVec mul_LT_lane(const LaneFac& F, const LaneVec& x) {
    LaneVec y = x;
    { float x0 = x.a[0], x1 = x.a[1]; solve_fence(x0, x1);
      sfor<0, 13>([&](auto I) { constexpr int i = I; fmac_bcast<i>(y.a[0], x0, F.Lc[0][i]); fmac_bcast<i>(y.a[1], x1, F.Lc[1][i]); }); }
    sfor<0, 6>([&](auto Pp) { y.p[Pp] += red16(F.w[0][Pp] * x.a[0] + F.w[1][Pp] * x.a[1]); });
    sfor<1, 6>([&](auto I) { constexpr int i = I; sfor<0, i>([&](auto J) { y.p[J] += F.Lp[i][J] * x.p[i]; }); });
    return y;
}

// stage B: factorisation + qacc_smooth = M^-1 qfrc_smooth.  STORE = false (the substep): nothing goes back to LDS - the share of the factor that the row stage needs (FR) and
// the whole factor for the finish stage (F) stay in registers; the compiler parks what it cannot keep across the sweeps in AGPRs (one v_accvgpr_write / _read per word,
// where rounds 1-4 stored the factor to the ancestor-chain layout with a select + address per word and the finish stage loaded it back: 2.9 k + 1.4 k cycles per substep).
// STORE = true (mj_setConst, setconst_rows_lane reads WK_LD / WK_DISQ through whiten_lane).
struct FacTail { float Lc[2][13], D[2], Dp[6]; };      // what the finish stage needs of the factor beyond FacRegs: the column view, the diagonal
template <bool STORE>
__device__ __forceinline__ void stage_factor_lane(const St& S, FacRegs& FR, FacTail& FT) {
    const LaneIdx X = lane_idx();
    LaneFac F;
    factor_lane<false>(S, X, 0.f, F);
    PROF2(20);
    if constexpr (STORE) fac_store(S, X, F);
    PROF2(21);
    LaneVec x = vec_load(S, X, -1, WK_SMOOTH);
    FR.qv = vec_load(S, X, F_QVEL, 0); FR.qw = vec_load(S, X, F_QACCW, 0);
    solve_LT_lane(F, x);
    sfor<0, 2>([&](auto Sd) { x.a[Sd] *= F.invD[Sd]; });
    sfor<0, 6>([&](auto Pp) { x.p[Pp] *= F.invDp[Pp]; });
    solve_L_lane(F, x);
    if constexpr (STORE) {
        if (X.l < 13) sfor<0, 2>([&](auto Sd) { S.W(WK_QS + 6 + 13 * Sd + X.l) = x.a[Sd]; });
        if (X.l == 0) sfor<0, 6>([&](auto Pp) { S.W(WK_QS + Pp) = x.p[Pp]; });
    }
    FR.qs = x;
    sfor<0, 2>([&](auto Sd) {
        sfor<0, 13>([&](auto J) { FR.Lr[Sd][J] = F.Lr[Sd][J]; });
        sfor<0, 6>([&](auto Pp) { FR.w[Sd][Pp] = F.w[Sd][Pp]; });
        FR.disq[Sd] = rsqrtf(F.D[Sd]);
    });
    sfor<0, 6>([&](auto Pi) { FR.disqp[Pi] = rsqrtf(F.Dp[Pi]); FT.Dp[Pi] = F.Dp[Pi]; sfor<0, Pi>([&](auto Qi) { FR.Lp[Pi][Qi] = F.Lp[Pi][Qi]; }); });
    sfor<0, 2>([&](auto Sd) { FT.D[Sd] = F.D[Sd]; sfor<0, 13>([&](auto J) { FT.Lc[Sd][J] = F.Lc[Sd][J]; }); });
}
__device__ __forceinline__ void stage_factor_lane(const St& S) { FacRegs FR; FacTail FT; stage_factor_lane<true>(S, FR, FT); }

// stage E: qacc, foot force, IMU, then (do_euler) mj_Euler with implicit joint damping:
// (M + h D) a = qfrc_smooth + J^T f = qfrc_smooth + L^T D^1/2 z~
__device__ __forceinline__ void stage_finish_lane(const St& S, const float* rows, bool do_euler, const FacTail& FT, const FacRegs& FR) {
    const LaneIdx X = lane_idx();
    const int l = X.l, lc = l < 13 ? l : 12;
    LaneFac F;      // the first factorisation, reassembled from the registers of the factor stage (the row stage fenced FR: its copies are the live ones)
    sfor<0, 2>([&](auto Sd) {
        sfor<0, 13>([&](auto J) { F.Lr[Sd][J] = FR.Lr[Sd][J]; F.Lc[Sd][J] = FT.Lc[Sd][J]; });
        sfor<0, 6>([&](auto Pp) { F.w[Sd][Pp] = FR.w[Sd][Pp]; });
        F.D[Sd] = FT.D[Sd];
    });
    sfor<0, 6>([&](auto Pi) { F.Dp[Pi] = FT.Dp[Pi]; sfor<0, Pi>([&](auto Qi) { F.Lp[Pi][Qi] = FR.Lp[Pi][Qi]; }); });
    const LaneVec z = vec_load(S, X, -1, WK_ZT);
    const LaneVec& qs = FR.qs;
    const float (&disq)[2] = FR.disq; const float (&disqp)[6] = FR.disqp;
    LaneVec qacc;
    sfor<0, 2>([&](auto Sd) { qacc.a[Sd] = z.a[Sd] * disq[Sd]; });
    sfor<0, 6>([&](auto Pp) { qacc.p[Pp] = z.p[Pp] * disqp[Pp]; });
    PROF2(28);
    solve_L_lane(F, qacc);
    sfor<0, 2>([&](auto Sd) { qacc.a[Sd] += qs.a[Sd]; });
    sfor<0, 6>([&](auto Pp) { qacc.p[Pp] += qs.p[Pp]; });
    if (l == 0) {
        // sensor snapshot of the PRE-integration state (sensordata is one mj_step1 old when step_ethercat reads it).  Every load before the first store: the compiler orders
        // an LDS load behind any earlier LDS store it cannot prove disjoint, and the copy-by-copy form was a chain of load -> wait -> store round trips on lane 0
        {
            float mp[10], jp[6], qq[4], gy[3];
            sfor<0, 10>([&](auto U) { mp[U] = S(F_QPOS + ct_act_qposadr[U]); });
            sfor<0, 6>([&](auto K) { jp[K] = S(F_QPOS + ct_jsens_qposadr[K]); });
            sfor<0, 4>([&](auto K) { qq[K] = S(F_QPOS + 3 + K); });
            sfor<0, 3>([&](auto K) { gy[K] = S(F_QVEL + 3 + K); });
            // world z of the contact force on the foot bodies (cassie_sim_foot_forces -> get_foot_forces()[2], [8]); branch-free: every slot record is read, the
            // slots that do not count (beyond the leg's contact count, or not a foot capsule) get weight 0
            const float nc0 = S.W(WK_MISC + 0), nc1 = S.W(WK_MISC + 1);
            const float mu = S(F_FRIC);
            float crv[2 * MAXC][8];
            sfor<0, 2 * MAXC>([&](auto Sl) {
                const float* cr = rows + R4_CON + R4_CONSZ * Sl;
                crv[Sl][0] = cr[7]; crv[Sl][1] = cr[8]; crv[Sl][2] = cr[9]; crv[Sl][3] = cr[10];
                crv[Sl][4] = cr[12]; crv[Sl][5] = cr[13]; crv[Sl][6] = cr[14]; crv[Sl][7] = cr[15];
            });
            float fz[2] = {0.f, 0.f};
            sfor<0, 2 * MAXC>([&](auto Sl) {
                constexpr int sl = Sl, lg = sl / MAXC;
                const float* c = crv[sl];      // c[1..3] = world z of the slot's contact frame (n, t1, t2), c[4..7] = the four pyramid forces
                const bool on = (float)(sl % MAXC) < (lg ? nc1 : nc0) && c[0] != 0.f;
                const float v = c[1] * (c[4] + c[5] + c[6] + c[7]) + mu * (c[2] * (c[4] - c[5]) + c[3] * (c[6] - c[7]));
                fz[lg] += on ? v : 0.f;
            });
            sfor<0, 10>([&](auto U) { S(F_SNAP + SN_MPOS + U) = mp[U]; });
            sfor<0, 6>([&](auto K) { S(F_SNAP + SN_JPOS + K) = jp[K]; });
            sfor<0, 4>([&](auto K) { S(F_SNAP + SN_QUAT + K) = qq[K]; });
            sfor<0, 3>([&](auto K) { S(F_SNAP + SN_GYRO + K) = gy[K]; });
            S(F_FWD + 0) = fz[0]; S(F_FWD + 1) = fz[1];
        }
        {   // accelerometer at the imu site (cassie.xml:267): classical acceleration of the site point, site frame
            SV A = {{S.W(WK_PEL), S.W(WK_PEL + 1), S.W(WK_PEL + 2)}, {S.W(WK_PEL + 3), S.W(WK_PEL + 4), S.W(WK_PEL + 5)}};
            const SV V = {{S.W(WK_PEL + 6), S.W(WK_PEL + 7), S.W(WK_PEL + 8)}, {S.W(WK_PEL + 9), S.W(WK_PEL + 10), S.W(WK_PEL + 11)}};
            M3 R;
            sfor<0, 9>([&](auto K) { R.m[K] = S.W(WK_PEL + 12 + K); });
            A.l = A.l + V3{qacc.p[0], qacc.p[1], qacc.p[2]};
            sfor<0, 3>([&](auto K) { A.a = A.a + col(R, K) * qacc.p[3 + K]; });
            const V3 r = mul(R, V3{ct_imu_pos[0], ct_imu_pos[1], ct_imu_pos[2]});
            const V3 vp = V.l + cross(V.a, r);
            const V3 a = A.l + cross(A.a, r) + cross(V.a, vp);
            S(F_SNAP + SN_ACC) = dot(col(R, 0), a); S(F_SNAP + SN_ACC + 1) = dot(col(R, 1), a); S(F_SNAP + SN_ACC + 2) = dot(col(R, 2), a);
        }
    }
    PROF2(29);
    if (!do_euler) return;
    // rhs = qfrc_smooth + L^T D^1/2 z~
    LaneVec x;
    sfor<0, 2>([&](auto Sd) { x.a[Sd] = z.a[Sd] * F.D[Sd] * disq[Sd]; });
    sfor<0, 6>([&](auto Pp) { x.p[Pp] = z.p[Pp] * F.Dp[Pp] * disqp[Pp]; });
    LaneVec rhs = mul_LT_lane(F, x);
    const LaneVec sm = vec_load(S, X, -1, WK_SMOOTH);
    sfor<0, 2>([&](auto Sd) { rhs.a[Sd] += sm.a[Sd]; });
    sfor<0, 6>([&](auto Pp) { rhs.p[Pp] += sm.p[Pp]; });
    __builtin_amdgcn_sched_barrier(0);
    PROF2(30);
    factor_lane<true>(S, X, DT, F);
    PROF2(31);
    solve_LT_lane(F, rhs);
    sfor<0, 2>([&](auto Sd) { rhs.a[Sd] *= F.invD[Sd]; });
    sfor<0, 6>([&](auto Pp) { rhs.p[Pp] *= F.invDp[Pp]; });
    solve_L_lane(F, rhs);
    PROF2(32);
    // ---- integrate: qvel += h a; hinges / slides qpos += h qvel; ball joints rotate by h w.
    // Three phases - every load, the arithmetic, every store: written as read-modify-write per item (`S(F_QPOS + qa) += ..`, rotate() reading and writing its
    // quaternion) the compiler had to keep each load behind the previous store (it cannot prove two LDS words disjoint): a chain of ~ 20 LDS round trips.
    static_assert(ct_jnt_qposadr[7] == 10 && ct_jnt_qposadr[8] == 14 && ct_jnt_qposadr[14] == 20 && ct_jnt_qposadr[18] == 24, "qpos layout");
    const int hoff = lc < 3 ? lc : lc + 1;      // leg-local hinge dof k -> qpos offset inside the leg block: k 0,1,2 -> 0,1,2; k >= 6 -> k + 1 (the ball's quaternion takes 4)
    float qv[2], qvp[6], hq[2], ppos[3];
    Q4 bq[2], pq;
    sfor<0, 2>([&](auto Sd) {
        constexpr int sd = Sd;
        qv[sd] = S(F_QVEL + 6 + 13 * sd + lc);
        hq[sd] = S(F_QPOS + 7 + 14 * sd + hoff);
        bq[sd] = {S(F_QPOS + 10 + 14 * sd), S(F_QPOS + 11 + 14 * sd), S(F_QPOS + 12 + 14 * sd), S(F_QPOS + 13 + 14 * sd)};
    });
    sfor<0, 6>([&](auto Pp) { qvp[Pp] = S(F_QVEL + Pp); });
    sfor<0, 3>([&](auto K) { ppos[K] = S(F_QPOS + K); });
    pq = {S(F_QPOS + 3), S(F_QPOS + 4), S(F_QPOS + 5), S(F_QPOS + 6)};
    auto rotate = [&](Q4 q, V3 wv) {
        const float nw = sqrtf(dot(wv, wv));
        float sn, cs;
        __sincosf(0.5f * nw * DT, &sn, &cs);
        const float sc = sn * rcpf(fmaxf(nw, 1e-30f));
        const Q4 qr = qmul(q, Q4{cs, wv.x * sc, wv.y * sc, wv.z * sc});
        const bool mv = nw > 0.f;
        return qnormalize(Q4{mv ? qr.w : q.w, mv ? qr.x : q.x, mv ? qr.y : q.y, mv ? qr.z : q.z});
    };
    sfor<0, 2>([&](auto Sd) {
        constexpr int sd = Sd;
        qv[sd] += DT * rhs.a[sd];
        const float w1 = dpp<0x150 + 4>(qv[sd]), w2 = dpp<0x150 + 5>(qv[sd]);      // achilles ball joint: dofs k = 3, 4, 5 on lanes 3..5
        bq[sd] = rotate(bq[sd], V3{qv[sd], w1, w2});                                 // (meaningful on lane 3 only)
        hq[sd] += DT * qv[sd];
    });
    sfor<0, 6>([&](auto Pp) { qvp[Pp] += DT * rhs.p[Pp]; });
    sfor<0, 3>([&](auto K) { ppos[K] += DT * qvp[K]; });
    pq = rotate(pq, V3{qvp[3], qvp[4], qvp[5]});
    sfor<0, 2>([&](auto Sd) {
        constexpr int sd = Sd;
        if (l < 13) { S(F_QVEL + 6 + 13 * sd + l) = qv[sd]; S(F_QACCW + 6 + 13 * sd + l) = qacc.a[sd]; }
        if (l == 3) { S(F_QPOS + 10 + 14 * sd) = bq[sd].w; S(F_QPOS + 11 + 14 * sd) = bq[sd].x; S(F_QPOS + 12 + 14 * sd) = bq[sd].y; S(F_QPOS + 13 + 14 * sd) = bq[sd].z; }
        else if (l < 13 && l != 4 && l != 5) S(F_QPOS + 7 + 14 * sd + hoff) = hq[sd];
    });
    if (l == 0) {
        sfor<0, 6>([&](auto Pp) { S(F_QVEL + Pp) = qvp[Pp]; S(F_QACCW + Pp) = qacc.p[Pp]; });
        sfor<0, 3>([&](auto K) { S(F_QPOS + K) = ppos[K]; });
        S(F_QPOS + 3) = pq.w; S(F_QPOS + 4) = pq.x; S(F_QPOS + 5) = pq.y; S(F_QPOS + 6) = pq.z;
    }
}

// y~ = D^-1/2 L^-T J^T for this lane's row over the 19 local columns of leg LEG (L streamed from LDS, uniform addresses); returns |y~|^2
template <int LEG>
__device__ __forceinline__ float whiten_lane(const St& S, float (&J)[19]) {
    srfor<0, 19>([&](auto C) {
        constexpr int c = C, i = c2d<LEG>(c);
        sfor<1, ct_dof_depth[i]>([&](auto A) { constexpr int a = A; J[d2c(ct_dof_anc[16 * i + a])] -= S.W(WK_LD + ct_dof_madr[i] + a) * J[c]; });
    });
    float nn = 0.f;
    sfor<0, 19>([&](auto C) { constexpr int c = C; J[c] *= S.W(WK_DISQ + c2d<LEG>(c)); nn += J[c] * J[c]; });
    return nn;
}

// What the row stage needs of the factor stage, kept in REGISTERS across the stage boundary (lane l = leg dof l of both legs): the rows of L, 1/sqrt(D), the (uniform)
// pelvis block, and the lane-distributed vectors of the three raw dots.  The whitening then takes every L entry as the DPP row-broadcast source of its
// multiply-add (v_fmac_f32_dpp) instead of streaming the factor back from LDS: ~130 LDS reads + ~75 AGPR round trips per leg less.
template <int LEG>
__device__ __forceinline__ float whiten_regs(const FacRegs& F, float (&J)[19]) {
    srfor<0, 19>([&](auto C) {
        constexpr int c = C, i = c2d<LEG>(c);
        sfor<1, ct_dof_depth[i]>([&](auto A) {
            constexpr int a = A, g = ct_dof_anc[16 * i + a], t = d2c(g);
            if constexpr (i >= 6) {
                constexpr int li = i - 6 - 13 * LEG;
                if constexpr (g >= 6) fnmac_bcast3<li>(J[t], F.Lr[LEG][g - 6 - 13 * LEG], J[c]);
                else fnmac_bcast3<li>(J[t], F.w[LEG][g], J[c]);
            } else J[t] -= F.Lp[i][g] * J[c];
        });
    });
    float nn = 0.f;
    sfor<0, 19>([&](auto C) {
        constexpr int c = C, i = c2d<LEG>(c);
        if constexpr (i >= 6) J[c] = mul_bcast<i - 6 - 13 * LEG>(F.disq[LEG], J[c]); else J[c] *= F.disqp[i];
        nn += J[c] * J[c];
    });
    return nn;
}
// J . x for a lane-distributed vector x
template <int LEG>
__device__ __forceinline__ float dot_regs(const LaneVec& x, const float (&J)[19]) {
    float r = 0.f;
    sfor<0, 6>([&](auto C) { r += J[C] * x.p[C]; });
    sfor<6, 19>([&](auto C) { constexpr int c = C; fmac_bcast<c - 6>(r, x.a[LEG], J[c]); });
    return r;
}

// mj_setConst subset, lane-parallel (after stage_tree_lane<true> and stage_factor_lane): body_invweight0 (translational) of the
// six constraint bodies of leg LEG and dof_invweight0 of its eight limited joints, |y~|^2 = J M^-1 J^T of unit rows.
// Two passes of 13 row vectors: pass 0 = bodies 0..3 (x, y, z) + limit 0, pass 1 = bodies 4, 5 + limits 1..7.
template <int LEG>
__device__ __forceinline__ void setconst_rows_lane(const St& S) {
    const int l = threadIdx.x & 15;
    const V3 o = {ct_qpos0[0], ct_qpos0[1], ct_qpos0[2]};
    sfor<0, 2>([&](auto Ps) {
        constexpr int pass = Ps;
        const bool isBody = pass == 0 ? l < 12 : l < 6;
        const int sl = (pass == 0 ? 0 : 4) + l / 3, ax = l % 3;                       // body slot, world axis
        const int jl = pass == 0 ? 0 : l - 5;                                      // limited-joint index 0..7 (pass 0: lane 12, pass 1: lanes 6..12)
        const int kl = jl == 0 ? 0 : jl == 1 ? 1 : jl == 2 ? 2 : jl == 3 ? 6 : jl == 4 ? 7 : jl == 5 ? 8 : jl == 6 ? 10 : 12;      // its leg dof
        static_assert(ct_jnt_limited[4] == 1 && ct_jnt_limited[8] == 1 && ct_jnt_limited[9] == 1 && ct_jnt_limited[10] == 1 && ct_jnt_limited[12] == 1 && ct_jnt_limited[14] == 1 &&
                      ct_jnt_limited[11] == 0 && ct_jnt_limited[13] == 0 && ct_jnt_dofadr[12] == 16 && ct_jnt_dofadr[14] == 18, "limited joints of a leg");
        const bool isLim = !isBody && l < 13;
        unsigned m = 0u;
        sfor<0, 6>([&](auto Sl) { if (sl == Sl) m = chain_mask<LEG>(cbody<LEG>(Sl)); });
        if (!isBody) m = 0u;
        const int slc = sl < 6 ? sl : 5;
        const V3 com = {S.W(WK_PTS + 30 * LEG + 3 * slc), S.W(WK_PTS + 30 * LEG + 3 * slc + 1), S.W(WK_PTS + 30 * LEG + 3 * slc + 2)};
        const V3 dir = {ax == 0 ? 1.f : 0.f, ax == 1 ? 1.f : 0.f, ax == 2 ? 1.f : 0.f};
        const V3 q = cross(com - o, dir);
        float J[19];
        sfor<0, 19>([&](auto C) {
            constexpr int c = C, d = c2d<LEG>(c);
            const V3 ca = {S.W(WK_CDOF + 6 * d), S.W(WK_CDOF + 6 * d + 1), S.W(WK_CDOF + 6 * d + 2)};
            const V3 cl = {S.W(WK_CDOF + 6 * d + 3), S.W(WK_CDOF + 6 * d + 4), S.W(WK_CDOF + 6 * d + 5)};
            float v = ((m >> c) & 1u) ? dot(dir, cl) + dot(q, ca) : 0.f;
            if (isLim && c == 6 + kl) v = 1.f;
            J[c] = v;
        });
        const float nn = whiten_lane<LEG>(S, J);
        const float tr = nn + dpp<0x111>(nn) + dpp<0x112>(nn);                     // x + y + z on the z lane (row_shr 1, 2)
        int bsel = 0;
        sfor<0, 6>([&](auto Sl) { if (sl == Sl) bsel = cbody<LEG>(Sl); });
        if (isBody && ax == 2) S(F_BIW + bsel) = tr * (1.f / 3.f);
        if (isLim) S(F_DIW + 6 + 13 * LEG + kl) = nn;
    });
    {   // third pass (round 5, for the complete-row path: cassie_complete.h): the bodies of the collision geoms the lane map does not instantiate - the hip-pitch body of
        // this leg (capsule cassie.xml:101 / :163) on lanes 0..2 and, once (LEG 0), the pelvis (sphere :87) on lanes 3..5
        static_assert(ct_qpos0[3] == 1.f && ct_qpos0[4] == 0.f && ct_qpos0[5] == 0.f && ct_qpos0[6] == 0.f && ct_geom_body[6] == 4 && ct_geom_body[7] == 16 && ct_geom_body[8] == 1, "pelvis upright at qpos0; hip-pitch / pelvis geoms");
        const bool hip = l < 3, pel = LEG == 0 && l >= 3 && l < 6;
        const int ax = l % 3;
        const unsigned m = hip ? chain_mask<LEG>(4 + 12 * LEG) : (pel ? 0x3Fu : 0u);
        V3 com = {S.W(WK_PTS + 30 * LEG + 18), S.W(WK_PTS + 30 * LEG + 19), S.W(WK_PTS + 30 * LEG + 20)};
        const V3 pc = o + cv3<1>(ct_body_ipos);
        com = {pel ? pc.x : com.x, pel ? pc.y : com.y, pel ? pc.z : com.z};
        const V3 dir = {ax == 0 ? 1.f : 0.f, ax == 1 ? 1.f : 0.f, ax == 2 ? 1.f : 0.f};
        const V3 q = cross(com - o, dir);
        float J[19];
        sfor<0, 19>([&](auto C) {
            constexpr int c = C, d = c2d<LEG>(c);
            const V3 ca = {S.W(WK_CDOF + 6 * d), S.W(WK_CDOF + 6 * d + 1), S.W(WK_CDOF + 6 * d + 2)};
            const V3 cl = {S.W(WK_CDOF + 6 * d + 3), S.W(WK_CDOF + 6 * d + 4), S.W(WK_CDOF + 6 * d + 5)};
            J[c] = ((m >> c) & 1u) ? dot(dir, cl) + dot(q, ca) : 0.f;
        });
        const float nn = whiten_lane<LEG>(S, J);
        const float tr = nn + dpp<0x111>(nn) + dpp<0x112>(nn);
        if (l == 2) S(F_BIW + 4 + 12 * LEG) = tr * (1.f / 3.f);
        if (LEG == 0 && l == 5) S(F_BIW + 1) = tr * (1.f / 3.f);
    }
    if (l == 0) S(F_BIW) = 0.f;
}

// left-leg vs right-leg capsules (foot, tarsus, shin against foot, tarsus, shin: cassie.xml:23-35, condim 1, frictionless): lane 3 i + j
// tests the pair (left geom i, right geom j) like mjc_CapsuleCapsule's general case (closest points of the two axes), records go to the
// row scratch; every lane then counts the hits, and lane 13 + k takes the k-th penetrating pair as ITS constraint row (MAXX = 2 rows; more
// is reported through SAT_LEG_LEG).  Pair order = the oracle's (left geom outer, right geom inner).
// surface under a point: signed distance of a sphere (centre c, radius rad) to the floor plane (n = fn) or, with a height field, to the
// plane of the grid triangle under it ((00, 10, 01) / (11, 01, 10) split of a cell, as in the oracle's floor_query); n = that surface normal
template <bool HF>
__device__ __forceinline__ float floor_dist_dev(const Hf& hf, V3 fn, V3 c, float rad, V3& n) {
    if constexpr (!HF) { n = fn; return dot(c - V3{ct_floor_pos[0], ct_floor_pos[1], ct_floor_pos[2]}, fn) - rad; }
    else {
    const float dx = 2.f * hf.sx / (float)(hf.ncol - 1), dy = 2.f * hf.sy / (float)(hf.nrow - 1);
    const float u = fminf(fmaxf((c.x + hf.sx) / dx, 0.f), (float)(hf.ncol - 1)), v = fminf(fmaxf((c.y + hf.sy) / dy, 0.f), (float)(hf.nrow - 1));
    const int ci = min((int)u, hf.ncol - 2), ri = min((int)v, hf.nrow - 2);      // the cell index is clamped as an INTEGER: a float margin is below fp32 resolution on large fields
    const float fu = u - (float)ci, fv = v - (float)ri;
    const float* q = hf.data + (size_t)ri * hf.ncol + ci;
    const float h00 = hf.sz * q[0], h10 = hf.sz * q[1], h01 = hf.sz * q[hf.ncol], h11 = hf.sz * q[hf.ncol + 1];
    const bool lo = fu + fv <= 1.f;
    const float gx = (lo ? h10 - h00 : h11 - h01) / dx, gy = (lo ? h01 - h00 : h11 - h10) / dy;
    const float hh = lo ? h00 + fu * (h10 - h00) + fv * (h01 - h00) : h11 + (1.f - fu) * (h01 - h11) + (1.f - fv) * (h10 - h11);
    const float inv = rsqrtf(gx * gx + gy * gy + 1.f);
    n = {-gx * inv, -gy * inv, inv};
    return (c.z - hh) * inv - rad;
    }
}
#ifndef APX_MAXX
#define APX_MAXX 3
#endif
constexpr int MAXX = APX_MAXX;      // leg-leg rows on lanes 13, 14, 15 (round 3: a policy recovering from a push keeps three capsule pairs in contact)
struct XPair { int gi, gj; V3 n, cp; float dist; };
constexpr int XSEL = 80, XSEL_SZ = 12;                    // compacted records of the MAXX selected pairs behind the 9 pair records (floats in the row scratch)
// the selected pair of lane 13 + k (k < nx), re-read where it is needed instead of being carried in registers through the row stage
__device__ __forceinline__ XPair xpair_load(const float* rec, int l) {
    const float* q = rec + XSEL + XSEL_SZ * (l >= 13 && l < 13 + MAXX ? l - 13 : 0);
    return XPair{(int)q[8], (int)q[9], {q[2], q[3], q[4]}, {q[5], q[6], q[7]}, q[1]};
}
__device__ __forceinline__ int legleg_pairs_lane(const St& S, float* rec, int& xmask) {
    xmask = 0;
    const int l = threadIdx.x & 15, li = l < 9 ? l / 3 : 0, rj = l < 9 ? l - 3 * (l / 3) : 0;
    {
        const lfloat* pl = &S.W(WK_PTS + 12 + 6 * li); const lfloat* pr = &S.W(WK_PTS + 30 + 12 + 6 * rj);
        const V3 p1 = {pl[0], pl[1], pl[2]}, q1 = {pl[3], pl[4], pl[5]}, p2 = {pr[0], pr[1], pr[2]}, q2 = {pr[3], pr[4], pr[5]};
        const V3 d1 = q1 - p1, d2 = q2 - p2, r = p1 - p2;
        const float a = dot(d1, d1), e = dot(d2, d2), f = dot(d2, r), c = dot(d1, r), b = dot(d1, d2), den = a * e - b * b;
        float sp = den > 1e-12f ? fminf(fmaxf((b * f - c * e) * rcpf(den), 0.f), 1.f) : 0.f;
        float tp = (b * sp + f) * rcpf(e);
        {   // the two clamped cases by plain selects (the if / else-if form is three exec-mask regions in every substep, ahead of the early out)
            const float ia = rcpf(a), sp_lo = fminf(fmaxf(-c * ia, 0.f), 1.f), sp_hi = fminf(fmaxf((b - c) * ia, 0.f), 1.f);
            sp = tp > 1.f ? sp_hi : sp; sp = tp < 0.f ? sp_lo : sp;
            tp = fminf(fmaxf(tp, 0.f), 1.f);
        }
        const V3 c1 = p1 + d1 * sp, dv = (p2 + d2 * tp) - c1;
        const float len = sqrtf(dot(dv, dv));
        const float rl = li == 0 ? ct_geom_radius[0] : ct_geom_radius[2], rr = rj == 0 ? ct_geom_radius[1] : ct_geom_radius[3];
        static_assert(ct_geom_radius[2] == ct_geom_radius[4] && ct_geom_radius[3] == ct_geom_radius[5], "tarsus / shin capsule radii");
        const float dist = len - rl - rr;
        const V3 nn = dv * rcpf(fmaxf(len, 1e-12f));                 // from the left geom to the right geom
        const V3 cp = c1 + nn * (rl + 0.5f * dist);
        const bool hit = l < 9 && dist < 0.f && len > 1e-9f;
        // wave-uniform early out: almost every substep has no penetrating pair in any of the wave's 4 envs
        if (__builtin_amdgcn_ballot_w64(hit) == 0ull) return 0;
        if (l < 9) {
            float* p = rec + 8 * l;
            p[0] = hit ? 1.f : 0.f; p[1] = dist; p[2] = nn.x; p[3] = nn.y; p[4] = nn.z; p[5] = cp.x; p[6] = cp.y; p[7] = cp.z;
        }
    }
    wsync();
    XPair x; x.gi = 0; x.gj = 0; x.n = {0.f, 0.f, 0.f}; x.cp = {0.f, 0.f, 0.f}; x.dist = 0.f;
    int nx = 0;
    const int k = l - 13;
    {   // the nine hit flags first, then ONE record read at the selected index (a load inside `if (h && nx == k)` per pair was nine serialised LDS round trips)
        float hit[9];
        sfor<0, 9>([&](auto P) { hit[P] = rec[8 * P]; });
        int sel = -1;
        sfor<0, 9>([&](auto P) {
            constexpr int p = P;
            const bool h = hit[p] != 0.f;
            sel = (h && nx == k) ? p : sel;
            nx += h ? 1 : 0; xmask |= h ? (1 << p) : 0;
        });
        const float* q = rec + 8 * (sel >= 0 ? sel : 0);
        const float q1 = q[1], q2 = q[2], q3 = q[3], q4 = q[4], q5 = q[5], q6 = q[6], q7 = q[7];
        const bool on = sel >= 0;
        const int sp = on ? sel : 0;
        x.gi = sp / 3; x.gj = sp - 3 * (sp / 3);
        x.dist = on ? q1 : 0.f; x.n = {on ? q2 : 0.f, on ? q3 : 0.f, on ? q4 : 0.f}; x.cp = {on ? q5 : 0.f, on ? q6 : 0.f, on ? q7 : 0.f};
    }
    if (k >= 0 && k < MAXX) {
        float* q = rec + XSEL + XSEL_SZ * k;
        q[1] = x.dist; q[2] = x.n.x; q[3] = x.n.y; q[4] = x.n.z; q[5] = x.cp.x; q[6] = x.cp.y; q[7] = x.cp.z; q[8] = (float)x.gi; q[9] = (float)x.gj;
    }
    wsync();
    return nx;
}

// Constraint rows of leg LEG, one row vector per lane: Jacobian from the stored motion axes, dots against
// qvel / qacc_smooth / qacc_warmstart on the raw row, whitening y~ = D^-1/2 L^-T J^T (L streamed from LDS, uniform
// addresses), row scalars.  Same arithmetic as c3::build_rows3, except that a connect row takes the common ancestors of
// its two bodies as axis x (p1 - p2) instead of the difference of two point Jacobians.
struct LegRows {
    float J[19];                 // this lane's whitened row vector (local columns)
    float b, R, invA, f;         // this lane's row scalars (equality / limit lanes)
    float vel, ju, jw, nn;       // this lane's raw dots J . qvel, J . qacc_smooth, J . qacc_warmstart and |y~|^2 (the leg-leg lanes combine the two legs)
    int nc, nlim;                // uniform over the env's lanes from here on
    int over;                    // SAT_LIMITS / SAT_CONTACTS: more active limits / penetrating capsule ends than the lane map has slots for
    int lmask, cmask;            // which limited joints of the leg are outside their range / which capsule ends penetrate (row-set signature, I_ROWSET)
    float cG[MAXC][6], cR[MAXC], cb[MAXC][4], cf[MAXC][4], isfoot[MAXC];
    float cfz[MAXC][3];          // world z of the contact frame (n, t1, t2) of each slot: the foot-force readout (cassie_sim_foot_forces)
};
template <int LEG, bool HF>
__device__ __forceinline__ void rows_lane(const St& S, const FacRegs& FR, LegRows& out, int nxp, bool anyx, const float* rec, const Hf& hf) {
    const int l = threadIdx.x & 15;
    const V3 o = {S(F_QPOS), S(F_QPOS + 1), S(F_QPOS + 2)};
    const V3 fn = {S(F_FLOOR), S(F_FLOOR + 1), S(F_FLOOR + 2)}, ft1 = {S(F_FLOOR + 3), S(F_FLOOR + 4), S(F_FLOOR + 5)},
             ft2 = {S(F_FLOOR + 6), S(F_FLOOR + 7), S(F_FLOOR + 8)};
    const float mu = S(F_FRIC);
    constexpr int base = WK_PTS + 30 * LEG;
    // ---- uniform over the env's lanes: first active joint limit of this leg
    int nlim = 0, clim = -1, over = 0, lmask = 0, lbit = 0;
    float lsign = 0.f, ldist = 0.f, ldiw = 0.f;
    // wave-uniform early out: in almost every substep no limited joint of any of the wave's four envs is outside its range
    float lmin = 1.f;
    sfor<0, NJ>([&](auto Jn) {
        constexpr int j = Jn;
        if constexpr (ct_jnt_limited[j] && ((ct_jnt_body[j] >= 14) == (LEG == 1)) && ct_jnt_body[j] >= 2) {
            const float q = S(F_QPOS + ct_jnt_qposadr[j]);
            lmin = fminf(lmin, fminf(q - ct_jnt_range[2 * j], ct_jnt_range[2 * j + 1] - q));
        }
    });
    if (__builtin_amdgcn_ballot_w64(lmin < 0.f) != 0ull)
    sfor<0, NJ>([&](auto Jn) {
        constexpr int j = Jn;
        if constexpr (ct_jnt_limited[j] && ((ct_jnt_body[j] >= 14) == (LEG == 1)) && ct_jnt_body[j] >= 2) {
            const float q = S(F_QPOS + ct_jnt_qposadr[j]);
            const float dlo = q - ct_jnt_range[2 * j], dhi = ct_jnt_range[2 * j + 1] - q;
            lmask |= (dlo < 0.f || dhi < 0.f) ? (1 << lbit) : 0; ++lbit;
            if (dlo < 0.f || dhi < 0.f) {
                if (nlim == 0) {
                    constexpr int d = ct_jnt_dofadr[j];
                    clim = d2c(d); lsign = dlo < 0.f ? 1.f : -1.f; ldist = dlo < 0.f ? dlo : dhi; ldiw = S(F_DIW + d); nlim = 1;
                } else over |= SAT_LIMITS;
            }
        }
    });
    // ---- first MAXC penetrating capsule ends in the order foot e0,e1, tarsus e0,e1, shin e0,e1.  Lane-parallel (round 5): lane i < 6 tests end i, the row's six hit bits
    // come back through a ballot, and the two selected ends are fetched from their lanes with ds_bpermute - ~30 instructions per leg where the uniform loop over the six
    // ends (every lane computing all of them, eight selects per end to keep "the first two") took ~170.
    const V3 p0 = {ct_floor_pos[0], ct_floor_pos[1], ct_floor_pos[2]};
    int nc = 0, cmask = 0;
    static_assert(MAXC == 2, "two contact slots per leg");
    V3 cpt0, cpt1, cn0 = fn, cn1 = fn; float cdist[MAXC]; int cgeo[MAXC];
    {
        const int ce = l < 6 ? l : 5;
        const V3 ctr = {S.W(base + 12 + 3 * ce), S.W(base + 12 + 3 * ce + 1), S.W(base + 12 + 3 * ce + 2)};
        float rad = ct_geom_radius[4 + LEG]; rad = ce < 4 ? ct_geom_radius[2 + LEG] : rad; rad = ce < 2 ? ct_geom_radius[0 + LEG] : rad;
        V3 sn;
        const float dist = floor_dist_dev<HF>(hf, fn, ctr, rad, sn);
        const V3 cp = ctr - sn * (rad + 0.5f * dist);
        const unsigned rowsh = threadIdx.x & 48u;
        cmask = (int)((__builtin_amdgcn_ballot_w64(l < 6 && dist < 0.f) >> rowsh) & 0x3Full);
        const int rest = cmask & (cmask - 1), npen = __builtin_popcount(cmask);
        const int i0 = cmask ? __builtin_ctz(cmask) : 0, i1 = rest ? __builtin_ctz(rest) : 0;
#ifdef APX_NEG_MAXC1      /* NEGATIVE CONTROL of the parity suite (make VARIANT=maxc1 EXTRA=-DAPX_NEG_MAXC1): only ONE floor contact per leg is instantiated; the teacher-forced test must fail on it */
        nc = npen < 1 ? npen : 1;
#else
        nc = npen < MAXC ? npen : MAXC;
        over |= npen > MAXC ? SAT_CONTACTS : 0;
#endif
        const int s0 = 4 * (int)(rowsh + i0), s1 = 4 * (int)(rowsh + i1);
        auto f0 = [&](float v) { return __int_as_float(__builtin_amdgcn_ds_bpermute(s0, __float_as_int(v))); };
        auto f1 = [&](float v) { return __int_as_float(__builtin_amdgcn_ds_bpermute(s1, __float_as_int(v))); };
        const V3 a0 = {f0(cp.x), f0(cp.y), f0(cp.z)}, a1 = {f1(cp.x), f1(cp.y), f1(cp.z)};
        const float d0 = f0(dist), d1 = f1(dist);
        const bool h0 = nc > 0, h1 = nc > 1;
        cpt0 = {h0 ? a0.x : 0.f, h0 ? a0.y : 0.f, h0 ? a0.z : 0.f}; cpt1 = {h1 ? a1.x : 0.f, h1 ? a1.y : 0.f, h1 ? a1.z : 0.f};
        cdist[0] = h0 ? d0 : 0.f; cdist[1] = h1 ? d1 : 0.f; cgeo[0] = h0 ? i0 >> 1 : 0; cgeo[1] = h1 ? i1 >> 1 : 0;
        if constexpr (HF) {
            const V3 n0 = {f0(sn.x), f0(sn.y), f0(sn.z)}, n1 = {f1(sn.x), f1(sn.y), f1(sn.z)};
            cn0 = {h0 ? n0.x : fn.x, h0 ? n0.y : fn.y, h0 ? n0.z : fn.z}; cn1 = {h1 ? n1.x : fn.x, h1 ? n1.y : fn.y, h1 ? n1.z : fn.z};
        }
    }
    // ---- this lane's row
    const bool isEq = l < 6, isLim = l == 6;
    const int E = l >= 3 ? 1 : 0, cs = l >= 10 ? 1 : 0;
    const bool isCon = l >= 7 && l < 13 && cs < nc;
    const int ax = isEq ? l - 3 * E : (l - 7) - 3 * cs;              // component / basis index 0..2
    constexpr unsigned mPL1 = chain_mask<LEG>(ct_eq_body1[2 * LEG]), mPL2 = chain_mask<LEG>(ct_eq_body2[2 * LEG]);
    constexpr unsigned mAC1 = chain_mask<LEG>(ct_eq_body1[2 * LEG + 1]), mAC2 = chain_mask<LEG>(ct_eq_body2[2 * LEG + 1]);
    constexpr unsigned mFT = chain_mask<LEG>(13 + 12 * LEG);
    // leg-leg row of lane 13 + k: this leg's half of n . (J_right(cp) - J_left(cp)); the pelvis columns cancel between the halves exactly
    // (both bodies hang off the pelvis) and are left out of both
    XPair xp = {0, 0, {0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}, 0.f};
    if (anyx) xp = xpair_load(rec, l);                   // wave-uniform: some env of the wave has a leg-leg contact
    const bool isX = l >= 13 && (l - 13) < nxp && (l - 13) < MAXX;
    const int G = isX ? (LEG == 0 ? xp.gi : xp.gj) : (cs ? cgeo[1] : cgeo[0]);
    const unsigned mcon = mFT & ~(G >= 1 ? (1u << 18) : 0u) & ~(G >= 2 ? (1u << 14) : 0u);   // tarsus / shin contact: dofs below do not move the point
    unsigned m1 = 0u; m1 = isX ? (mcon & ~0x3Fu) : m1; m1 = isCon ? mcon : m1; m1 = isEq ? (E ? mAC1 : mPL1) : m1;
    const unsigned m2 = isEq ? (E ? mAC2 : mPL2) : 0u;
    V3 p1, p2;
    {
        const V3 e1 = {S.W(base + 6 * E), S.W(base + 6 * E + 1), S.W(base + 6 * E + 2)};
        const V3 e2 = {S.W(base + 6 * E + 3), S.W(base + 6 * E + 4), S.W(base + 6 * E + 5)};
        V3 cp = {cs ? cpt1.x : cpt0.x, cs ? cpt1.y : cpt0.y, cs ? cpt1.z : cpt0.z};
        cp = {isX ? xp.cp.x : cp.x, isX ? xp.cp.y : cp.y, isX ? xp.cp.z : cp.z};
        p1 = isEq ? e1 : cp; p2 = e2;
    }
    V3 dir;
    {
        // scalar selects only: `isEq ? de : dc` on two V3 temporaries becomes a load through a selected POINTER, which keeps both
        // temporaries in scratch (6 stores + 3 loads per leg per substep = 8x the algorithmic HBM traffic of the whole kernel)
        // contact frame of this lane's slot: the floor frame, or (height field) mju_makeFrame of the triangle normal
        V3 cn = fn, c1 = ft1, c2 = ft2;
        if constexpr (HF) {
            cn = {cs ? cn1.x : cn0.x, cs ? cn1.y : cn0.y, cs ? cn1.z : cn0.z};
            const bool uy = fabsf(cn.y) < 0.5f;
            V3 t = {0.f, uy ? 1.f : 0.f, uy ? 0.f : 1.f};
            t = t - cn * dot(cn, t); c1 = t * rsqrtf(dot(t, t)); c2 = cross(cn, c1);
        }
        float dcx = c2.x, dcy = c2.y, dcz = c2.z;
        dcx = ax == 1 ? c1.x : dcx; dcy = ax == 1 ? c1.y : dcy; dcz = ax == 1 ? c1.z : dcz;
        dcx = ax == 0 ? cn.x : dcx; dcy = ax == 0 ? cn.y : dcy; dcz = ax == 0 ? cn.z : dcz;
        dir = {isEq ? (ax == 0 ? 1.f : 0.f) : dcx, isEq ? (ax == 1 ? 1.f : 0.f) : dcy, isEq ? (ax == 2 ? 1.f : 0.f) : dcz};
        constexpr float sx = LEG == 0 ? -1.f : 1.f;
        if (isX) dir = {sx * xp.n.x, sx * xp.n.y, sx * xp.n.z};
    }
    const V3 q1 = cross(p1 - o, dir), q2 = cross(p2 - o, dir);       // dir . (a x r) = a . (r x dir)
    float (&J)[19] = out.J;
    {   // The 19 motion axes come from REGISTERS: lane l holds the axis of leg dof l (and lanes 0..5 the pelvis axes), loaded once per leg, and a column's six words are the
        // DPP row-broadcast sources of its nine multiply-adds (v_mul_f32_dpp / v_fmac_f32_dpp) - 18 LDS reads per leg where streaming the columns took 114 (and, before the
        // loads were chunked, 19 exposed round trips)
        float cdl[6], cdp[6];
        {
            const int ll = l < 13 ? l : 12, lp = l < 6 ? l : 5;
            sfor<0, 6>([&](auto I) { cdl[I] = S.W(WK_CDOF + 6 * (6 + 13 * LEG) + 6 * ll + I); cdp[I] = S.W(WK_CDOF + 6 * lp + I); });
        }
        APX_HAZARD_FENCE("+v"(cdl[0]), "+v"(cdl[1]), "+v"(cdl[2]), "+v"(cdl[3]), "+v"(cdl[4]), "+v"(cdl[5]),
                                 "+v"(cdp[0]), "+v"(cdp[1]), "+v"(cdp[2]), "+v"(cdp[3]), "+v"(cdp[4]), "+v"(cdp[5]));
        sfor<0, 19>([&](auto C) {
            constexpr int c = C, src = c < 6 ? c : c - 6;
            const float (&cd)[6] = c < 6 ? cdp : cdl;
            float dl = mul_bcast<src>(cd[3], dir.x), g1 = mul_bcast<src>(cd[0], q1.x), g2 = mul_bcast<src>(cd[0], q2.x);
            fmac_bcast<src>(dl, cd[4], dir.y); fmac_bcast<src>(g1, cd[1], q1.y); fmac_bcast<src>(g2, cd[1], q2.y);
            fmac_bcast<src>(dl, cd[5], dir.z); fmac_bcast<src>(g1, cd[2], q1.z); fmac_bcast<src>(g2, cd[2], q2.z);
            float v = ((m1 >> c) & 1u) ? g1 + dl : 0.f;
            v -= ((m2 >> c) & 1u) ? g2 + dl : 0.f;
            if (isLim && nlim && c == clim) v = lsign;
            J[c] = v;
        });
    }
    PROF2(25);
    const float vel = dot_regs<LEG>(FR.qv, J), ju = dot_regs<LEG>(FR.qs, J), jw = dot_regs<LEG>(FR.qw, J);
    PROF2(26);
    const float nn = whiten_regs<LEG>(FR, J);
    // the next reader takes J through DPP (row_shr) and cannot see the asm writes
    APX_HAZARD_FENCE("+v"(J[0]), "+v"(J[1]), "+v"(J[2]), "+v"(J[3]), "+v"(J[4]), "+v"(J[5]), "+v"(J[6]), "+v"(J[7]), "+v"(J[8]), "+v"(J[9]), "+v"(J[10]), "+v"(J[11]),
                 "+v"(J[12]), "+v"(J[13]), "+v"(J[14]), "+v"(J[15]), "+v"(J[16]), "+v"(J[17]), "+v"(J[18]));
    out.vel = vel; out.ju = ju; out.jw = jw; out.nn = nn;
    PROF2(27);
    // ---- equality / limit scalars (mj_makeImpedance, mj_referenceConstraint, warm start from qacc_warmstart)
    {
        const V3 cv = p1 - p2;
        float cpos = cv.z; cpos = ax == 1 ? cv.y : cpos; cpos = ax == 0 ? cv.x : cpos;
        const float tranPL = S(F_BIW + ct_eq_body1[2 * LEG]) + S(F_BIW + ct_eq_body2[2 * LEG]);
        const float tranAC = S(F_BIW + ct_eq_body1[2 * LEG + 1]) + S(F_BIW + ct_eq_body2[2 * LEG + 1]);
        const float pos = isEq ? cpos : ldist, imp_pos = isEq ? sqrtf(dot(cv, cv)) : ldist;
        const float diag = isEq ? (E ? tranAC : tranPL) : ldiw;
        const RowK kb = solref(isEq ? 0.005f : 0.02f);
        const float imp = impedance(imp_pos);
        const float R = fmaxf(MINVAL, (1.f - imp) * rcpf(imp) * diag);
        const float aref = -kb.B * vel - kb.K * imp * pos;
        float b = ju - aref;
        float f = -(jw - aref) * rcpf(R);
        if (isLim && f < 0.f) f = 0.f;
        float invA = rcpf(nn + R), Rw = R;
        if (isLim && !nlim) { b = 0.f; f = 0.f; invA = 0.f; Rw = 1.f; }
        out.b = b; out.R = Rw; out.invA = invA; out.f = f;
    }
    // ---- contact scalars: Gram matrix of (n, t1, t2) across the three basis lanes, pyramid rows n +- mu t_j
    float a1 = 0.f, a2 = 0.f;
    // a1 += J[C] * row_shr:1(J[C]), a2 likewise with row_shr:2: the shifted operand is the DPP source of the fmac (J is fenced right above)
    sfor<0, 19>([&](auto C) { fmac_shr<1>(a1, J[C]); fmac_shr<2>(a2, J[C]); });
    sfor<0, MAXC>([&](auto Sl) {
        constexpr int s = Sl, ln = 7 + 3 * s;
        const float gnn = dpp<0x150 + ln>(nn), g11 = dpp<0x150 + ln + 1>(nn), g22 = dpp<0x150 + ln + 2>(nn);
        const float gn1 = dpp<0x150 + ln + 1>(a1), g12 = dpp<0x150 + ln + 2>(a1), gn2 = dpp<0x150 + ln + 2>(a2);
        const float vn = dpp<0x150 + ln>(vel), v1 = dpp<0x150 + ln + 1>(vel), v2 = dpp<0x150 + ln + 2>(vel);
        const float un = dpp<0x150 + ln>(ju), u1 = dpp<0x150 + ln + 1>(ju), u2 = dpp<0x150 + ln + 2>(ju);
        const float wn = dpp<0x150 + ln>(jw), w1 = dpp<0x150 + ln + 1>(jw), w2 = dpp<0x150 + ln + 2>(jw);
        const int Gs = cgeo[s];
        const float dist = cdist[s];
        float tran = S(F_BIW + 8 + 12 * LEG);                 // all three are read, plain selects (loads in the arms of a nested ?: become branches)
        { const float t9 = S(F_BIW + 9 + 12 * LEG), t13 = S(F_BIW + 13 + 12 * LEG); tran = Gs == 1 ? t9 : tran; tran = Gs == 0 ? t13 : tran; }
        const RowK kb = solref(0.005f);
        const float imp = impedance(dist);
        const float R1 = fmaxf(MINVAL, (1.f - imp) * rcpf(imp) * (tran + mu * mu * tran));
        const float Rpy = fmaxf(MINVAL, 2.f * mu * mu * R1);       // pyramidal regulariser, impratio 1
        const float iRpy = rcpf(Rpy);
        const float sv[4] = {mu * v1, -mu * v1, mu * v2, -mu * v2}, su[4] = {mu * u1, -mu * u1, mu * u2, -mu * u2};
        const float sw[4] = {mu * w1, -mu * w1, mu * w2, -mu * w2};
        out.cG[s][0] = gnn; out.cG[s][1] = gn1; out.cG[s][2] = gn2; out.cG[s][3] = g11; out.cG[s][4] = g12; out.cG[s][5] = g22;
        out.cR[s] = Rpy; out.isfoot[s] = Gs == 0 ? 1.f : 0.f;
        sfor<0, 4>([&](auto K) {
            constexpr int k = K;
            const float aref = -kb.B * (vn + sv[k]) - kb.K * imp * dist;
            out.cb[s][k] = un + su[k] - aref;
            const float f = -((wn + sw[k]) - aref) * iRpy;
            out.cf[s][k] = (s < nc && f > 0.f) ? f : 0.f;
        });
    });
    out.nc = nc; out.nlim = nlim; out.over = over; out.lmask = lmask; out.cmask = cmask;
    sfor<0, MAXC>([&](auto Sl) {
        V3 cn = fn, c1 = ft1, c2 = ft2;
        if constexpr (HF) {
            cn = Sl ? cn1 : cn0;
            const bool uy = fabsf(cn.y) < 0.5f;
            V3 t = {0.f, uy ? 1.f : 0.f, uy ? 0.f : 1.f};
            t = t - cn * dot(cn, t); c1 = t * rsqrtf(dot(t, t)); c2 = cross(cn, c1);
        }
        out.cfz[Sl][0] = cn.z; out.cfz[Sl][1] = c1.z; out.cfz[Sl][2] = c2.z;
    });
}

// Projected Gauss-Seidel in GRAM SPACE.  Lane r (0..12) owns basis vector r of both legs (A = left, B = right): 6 connect
// rows, the limit row, and n, t1, t2 of two contact slots.  Instead of z~ the sweep carries rho_r = y~_r . z~ for every
// basis vector; a change d of the coefficient of basis s moves it by G[r][s] d, with G = Y~ Y~^T formed once per substep
// (19-term fma with a DPP row broadcast operand; the two legs only meet in the 6 pelvis columns).  A row update is then
// broadcast + scalar update + 2 fma: no cross-lane reduction inside the 50 sweeps.  Order is leg-major as before (left: 6
// equality rows, limit, contacts; then right), a pyramidal contact sweeps its 4 rows through the 3x3 Gram block.
template <bool HF>
__device__ __forceinline__ void stage_rows_pgs_lane(const St& S, FacRegs& FR, float* rows, int pgs_iters, const Hf& hf) {
    const int l = threadIdx.x & 15;
    const float mu = S(F_FRIC);
    LegRows A, B;
    int xmask;
    const int nxp = legleg_pairs_lane(S, rows, xmask);
    const bool anyx = __builtin_amdgcn_ballot_w64(nxp > 0) != 0ull;          // wave-uniform: some env of the wave has a leg-leg contact
    {   // every register the row stage reads through DPP: fenced once against compiler-generated definitions right in front of the first read
#define APX_F13(a) "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12])
#define APX_F6(a) "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5])
        APX_HAZARD_FENCE(APX_F13(FR.Lr[0]), APX_F6(FR.w[0]), "+v"(FR.disq[0]), "+v"(FR.qs.a[0]), "+v"(FR.qv.a[0]), "+v"(FR.qw.a[0]));
        APX_HAZARD_FENCE(APX_F13(FR.Lr[1]), APX_F6(FR.w[1]), "+v"(FR.disq[1]), "+v"(FR.qs.a[1]), "+v"(FR.qv.a[1]), "+v"(FR.qw.a[1]));
#undef APX_F13
#undef APX_F6
    }
    rows_lane<0, HF>(S, FR, A, nxp, anyx, rows, hf);
    PROF2(23);
    rows_lane<1, HF>(S, FR, B, nxp, anyx, rows, hf);
    PROF2(24);
    // bookkeeping that must not stay live across the sweeps: saturation report, contact / limit counts, and the contact slot records for
    // the foot-force readout of the finish stage (foot flag, world z of the slot's frame)
    if (l == 0) {
        const int sat = A.over | B.over | ((S.W(WK_MISC + 4) + S.W(WK_MISC + 5) > 0.f) ? SAT_BODY_FLOOR : 0) | (nxp > MAXX ? SAT_LEG_LEG : 0);
        if (sat) S.I(I_SAT) = (S.I(I_SAT) | sat) + 256;
        S.W(WK_MISC + 7) = sat ? 1.f : 0.f;      // this pass needs rows beyond the lane map: sim_step_pd hands the env to the complete-row path (cassie_complete.h) instead of the finish stage
        {   // row-set signature of this forward pass, folded into the env step's hash (multiplicative hash over two words; the oracle folds the same words): limited joints out of
            // range (8 bits per leg), penetrating capsule ends (6 per leg), pelvis sphere / hip-pitch capsules on the floor, the 9 left x right capsule pairs
            const unsigned bf = (S.W(WK_MISC + 4) > 0.f ? 1u : 0u) | (S.W(WK_MISC + 5) > 0.f ? 2u : 0u);
            const unsigned s1 = (unsigned)A.lmask | (unsigned)B.lmask << 8 | (unsigned)A.cmask << 16 | (unsigned)B.cmask << 22 | bf << 28;
            unsigned h = (unsigned)S.I(I_ROWSET);
            const unsigned s2 = (unsigned)xmask | (unsigned)(int)S.W(WK_MISC + 6) << 16;      // leg-leg pairs | the state estimator's load switches of this substep (estimator_lane.h)
            h = (h ^ s1) * 0x9E3779B1u; h ^= h >> 15; h = (h ^ s2) * 0x9E3779B1u; h ^= h >> 15;
            S.I(I_ROWSET) = (int)h;
        }
        S.W(WK_MISC + 0) = (float)A.nc; S.W(WK_MISC + 1) = (float)B.nc; S.W(WK_MISC + 2) = (float)A.nlim; S.W(WK_MISC + 3) = (float)B.nlim;
    }
    if (l == 0) sfor<0, 2 * MAXC>([&](auto Sl) {
        constexpr int s = Sl, leg = s / MAXC, j = s % MAXC;
        float* cr = rows + R4_CON + R4_CONSZ * s;
        cr[7] = (leg ? B : A).isfoot[j];
        sfor<0, 3>([&](auto K) { cr[8 + K] = (leg ? B : A).cfz[j][K]; });
    });
    // ---- leg-leg rows (lanes 13 + k): scalars from the two halves.  Row = (XL | XR) with XL = A.J, XR = B.J on that lane; its pelvis
    // part is the SUM of the two whitened pelvis parts, so |y~|^2 = |XL|^2 + |XR|^2 + 2 XL_pel . XR_pel
    const int nx = nxp < MAXX ? nxp : MAXX;
    float xb_l = 0.f, xR_l = 1.f, xiA_l = 0.f, xf_l = 0.f;
    if (anyx) {
        const XPair xp = xpair_load(rows, l);
        float pp = 0.f;
        sfor<0, 6>([&](auto C) { pp += A.J[C] * B.J[C]; });
        const float nnx = A.nn + B.nn + 2.f * pp;
        const int bi = xp.gi == 0 ? 13 : xp.gi == 1 ? 9 : 8, bj = xp.gj == 0 ? 25 : xp.gj == 1 ? 21 : 20;
        static_assert(ct_geom_body[0] == 13 && ct_geom_body[2] == 9 && ct_geom_body[4] == 8 && ct_geom_body[1] == 25 && ct_geom_body[3] == 21 && ct_geom_body[5] == 20, "capsule bodies");
        const float tran = S(F_BIW + bi) + S(F_BIW + bj);
        const RowK kb = solref(0.005f);
        const float imp = impedance(xp.dist);
        const float R = fmaxf(MINVAL, (1.f - imp) * rcpf(imp) * tran);
        const float aref = -kb.B * (A.vel + B.vel) - kb.K * imp * xp.dist;
        const bool isX = l >= 13 && (l - 13) < nx;
        xb_l = isX ? (A.ju + B.ju) - aref : 0.f;
        xR_l = isX ? R : 1.f;
        xiA_l = isX ? rcpf(nnx + R) : 0.f;
        xf_l = isX ? fmaxf(-((A.jw + B.jw) - aref) * rcpf(R), 0.f) : 0.f;
    }
    __builtin_amdgcn_sched_barrier(0);
    PROF(5);
    // ---- Gram columns of this lane's two rows
    float GAA[13], GBB[13], GAB[13], GBA[13];      // GXY[s] = y~_(r,X) . y~_(s,Y)
    typedef float f2g __attribute__((ext_vector_type(2)));
    f2g GX[MAXX];                                  // leg-leg row k = (XL | XR) on lane 13 + k: how its coefficient moves (rho_A, rho_B) of this lane
    {
        // every product takes its broadcast operand through DPP inside the multiply-add (v_fmac_f32_dpp): 2 instructions per leg column and source lane,
        // 4 per pelvis column, where a separate v_mov_b32_dpp per broadcast + packed fmas took 3 and 4
#define APX_FENCE19J(L) APX_HAZARD_FENCE("+v"(L.J[0]), "+v"(L.J[1]), "+v"(L.J[2]), "+v"(L.J[3]), "+v"(L.J[4]), "+v"(L.J[5]), "+v"(L.J[6]), "+v"(L.J[7]), \
        "+v"(L.J[8]), "+v"(L.J[9]), "+v"(L.J[10]), "+v"(L.J[11]), "+v"(L.J[12]), "+v"(L.J[13]), "+v"(L.J[14]), "+v"(L.J[15]), "+v"(L.J[16]), "+v"(L.J[17]), "+v"(L.J[18]))
        APX_FENCE19J(A); APX_FENCE19J(B);
#undef APX_FENCE19J
        sfor<0, 13>([&](auto Sx) {
            constexpr int s = Sx;
            float aa = mul_bcast<s>(A.J[0], A.J[0]), bb = mul_bcast<s>(B.J[0], B.J[0]);
            float ab = mul_bcast<s>(B.J[0], A.J[0]), ba = mul_bcast<s>(A.J[0], B.J[0]);      // the legs only meet in the pelvis columns
            sfor<1, 19>([&](auto C) {
                constexpr int c = C;
                fmac_bcast<s>(aa, A.J[c], A.J[c]); fmac_bcast<s>(bb, B.J[c], B.J[c]);
                if constexpr (c < 6) { fmac_bcast<s>(ab, B.J[c], A.J[c]); fmac_bcast<s>(ba, A.J[c], B.J[c]); }
            });
            // pin the sums here: LLVM otherwise sinks the fma chains into the conditional contact blocks that consume them
            APX_PIN("+v"(aa), "+v"(bb), "+v"(ab), "+v"(ba));
            GAA[s] = aa; GBB[s] = bb; GAB[s] = ab; GBA[s] = ba;
        });
        sfor<0, MAXX>([&](auto K) { GX[K] = f2g{0.f, 0.f}; });
        if (anyx) sfor<0, MAXX>([&](auto K) {
            constexpr int s = 13 + K;
            // (A_r . XL + A_r,pel . XR_pel, B_r . XR + B_r,pel . XL_pel)
            float aa = mul_bcast<s>(A.J[0], A.J[0]), bb = mul_bcast<s>(B.J[0], B.J[0]);
            sfor<1, 19>([&](auto C) { constexpr int c = C; fmac_bcast<s>(aa, A.J[c], A.J[c]); fmac_bcast<s>(bb, B.J[c], B.J[c]); });
            sfor<0, 6>([&](auto C) { constexpr int c = C; fmac_bcast<s>(aa, B.J[c], A.J[c]); fmac_bcast<s>(bb, A.J[c], B.J[c]); });
            APX_PIN("+v"(aa), "+v"(bb));
            GX[K] = f2g{aa, bb};
        });
    }
    // The row vectors are needed again only for the z~ read-back after the sweeps: park them in the (now free) row store instead of
    // keeping 38 registers live across the loop.  Column c of leg g of lane l at rows[(2 c + g) 16 + l]: conflict-free, own lane only.
    sfor<0, 19>([&](auto C) { rows[(2 * C + 0) * 16 + l] = A.J[C]; rows[(2 * C + 1) * 16 + l] = B.J[C]; });
    static_assert(38 * 16 <= R4_CON, "parked row vectors fit below the contact slot records");
    // leg-leg row scalars to every lane
    float xb[MAXX], xR[MAXX], xiA[MAXX], xf[MAXX]; bool xact[MAXX];
    sfor<0, MAXX>([&](auto K) {
        constexpr int k = K;
        xb[k] = dpp<0x150 + 13 + k>(xb_l); xR[k] = dpp<0x150 + 13 + k>(xR_l); xiA[k] = dpp<0x150 + 13 + k>(xiA_l); xf[k] = dpp<0x150 + 13 + k>(xf_l);
        xact[k] = k < nx;
    });
    // ---- row scalars to every lane: c = b + R f, R, 1/(A+R), f
    float ec[2][7], eR[2][7], eiA[2][7], ef[2][7];
    sfor<0, 7>([&](auto Sx) {
        constexpr int s = Sx;
        eR[0][s] = dpp<0x150 + s>(A.R); eiA[0][s] = dpp<0x150 + s>(A.invA); ef[0][s] = dpp<0x150 + s>(A.f); ec[0][s] = dpp<0x150 + s>(A.b);
        eR[1][s] = dpp<0x150 + s>(B.R); eiA[1][s] = dpp<0x150 + s>(B.invA); ef[1][s] = dpp<0x150 + s>(B.f); ec[1][s] = dpp<0x150 + s>(B.b);
    });
    constexpr int NCS = 2 * MAXC;
    float cG[NCS][6], cR[NCS], cb[NCS][4], cf[NCS][4], ciA[NCS][4], kn[NCS][4], k1[NCS][4], k2[NCS][4];
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
            ciA[s][k] = con[s] ? rcpf(cG[s][0] + 2.f * sm * gnj + mu * mu * gjj + cR[s]) : 0.f;      // 1 / (A_kk + R) of row n + s mu t_j; 0 = the slot is a no-op in the sweeps
            // row k = n + sm t_j moves the basis residuals (rn, r1, r2) by df * (G n-col + sm G j-col)
            kn[s][k] = cG[s][0] + sm * gnj; k1[s][k] = cG[s][1] + sm * (k < 2 ? cG[s][3] : cG[s][4]); k2[s][k] = cG[s][2] + sm * (k < 2 ? cG[s][4] : cG[s][5]);
        });
    });
    const int nlim[2] = {A.nlim, B.nlim};
    // Wave-uniform activity of the conditional row groups: inside the sweeps only SCALAR branches are taken (skip a group when no env of
    // the wave has it); an env without the row runs it with zero coefficients (iA = 0, f = 0: the update is exactly 0), so there is no
    // exec-mask region in the loop at all.
    bool anyc[NCS], anyl[2];
    sfor<0, NCS>([&](auto Sl) { anyc[Sl] = __builtin_amdgcn_ballot_w64(con[Sl]) != 0ull; });
    sfor<0, 2>([&](auto Lg) { anyl[Lg] = __builtin_amdgcn_ballot_w64(nlim[Lg] != 0) != 0ull; });
#ifdef APX_WAVETIME      /* experiment build: which optional row groups did this wave run (per launch: substeps with leg-leg rows, with a limit row, sum of active contact slots) */
    if (threadIdx.x == 0) {
        g_wavefeat[blockIdx.x * 4 + 0] += anyx ? 1u : 0u; g_wavefeat[blockIdx.x * 4 + 1] += (anyl[0] || anyl[1]) ? 1u : 0u;
        g_wavefeat[blockIdx.x * 4 + 2] += (unsigned)anyc[0] + (unsigned)anyc[1] + (unsigned)anyc[2] + (unsigned)anyc[3];
    }
#endif
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
    // a coefficient of leg-leg row k moves z~ by (XL | XR): rho_A by XL . A_r + XR_pel . A_r,pel, rho_B likewise
    sfor<0, MAXX>([&](auto K) { rA += GX[K].x * xf[K]; rB += GX[K].y * xf[K]; });
    {
        // F of this lane's own rows times rho: the own coefficients by plain select chains (a 13-way `if (l == s)` over computed values
        // becomes a tree of divergent branches); lanes 0..6 hold their own row's f in A.f / B.f
        float fa = l < 7 ? A.f : 0.f, fb = l < 7 ? B.f : 0.f;
        sfor<7, 13>([&](auto Sx) { fa = l == Sx ? FA[Sx] : fa; fb = l == Sx ? FB[Sx] : fb; });
        float own = fa * rA + fb * rB;
        sfor<0, MAXX>([&](auto K) { own = l == 13 + K ? xf[K] * (rA + rB) : own; });
        cost = 0.5f * red16(own);
    }
    sfor<0, MAXX>([&](auto K) { cost += xf[K] * (0.5f * xR[K] * xf[K] + xb[K]); });
    sfor<0, 2>([&](auto Lg) { sfor<0, 7>([&](auto Sx) { cost += ef[Lg][Sx] * (0.5f * eR[Lg][Sx] * ef[Lg][Sx] + ec[Lg][Sx]); }); });
    sfor<0, NCS>([&](auto Sl) { sfor<0, 4>([&](auto K) { cost += cf[Sl][K] * (0.5f * cR[Sl] * cf[Sl][K] + cb[Sl][K]); }); });
    if (cost > 0.f) {
        rA = rB = 0.f;
        sfor<0, 2>([&](auto Lg) { sfor<0, 7>([&](auto Sx) { ef[Lg][Sx] = 0.f; }); });
        sfor<0, NCS>([&](auto Sl) { sfor<0, 4>([&](auto K) { cf[Sl][K] = 0.f; }); });
        sfor<0, MAXX>([&](auto K) { xf[K] = 0.f; });
    }
    PROF(6);
    // ---- sweeps.  Packed fp32 (v_pk_fma_f32) wherever two independent updates share the multiplier: (rho_A, rho_B), the
    // running (n, active tangent) residuals of a pyramid, and its (sum df, sum +-mu df) accumulators.
    // The regulariser of the equality / limit rows rides on the Gram diagonal: the sweep carries rho' = rho + R f on the row's own
    // lane (G'[s][s] = G[s][s] + R_s), so a row's residual is rho'_s + b_s with b constant and no per-row bookkeeping of c = b + R f
    // is left in the loop; the coefficient f_s itself is accumulated on its own lane only (one fma that fills the DPP hazard slot).
    typedef float f2 __attribute__((ext_vector_type(2)));
    const float f0A = l < 7 ? (cost > 0.f ? 0.f : A.f) : 0.f, f0B = l < 7 ? (cost > 0.f ? 0.f : B.f) : 0.f;      // this lane's own coefficients
    f2 r = {rA + (l < 7 ? A.R * f0A : 0.f), rB + (l < 7 ? B.R * f0B : 0.f)};
    f2 GpX[MAXX];                                           // how the coefficient of leg-leg row k moves (rho'_A, rho'_B); its regulariser rides on rho'_A of lane 13 + k
    sfor<0, MAXX>([&](auto K) {
        constexpr int s = 13 + K;
        GpX[K] = f2{GX[K].x + (l == s ? xR[K] : 0.f), GX[K].y};
        r.x += l == s ? xR[K] * xf[K] : 0.f;
    });
    f2 Gp[2][13];                                           // how a coefficient of leg L's basis s moves (rho'_A, rho'_B)
    sfor<0, 13>([&](auto Sx) {
        constexpr int s = Sx;
        Gp[0][s] = f2{GAA[s] + ((s < 7 && l == s) ? A.R : 0.f), GBA[s]};
        Gp[1][s] = f2{GAB[s], GBB[s] + ((s < 7 && l == s) ? B.R : 0.f)};
    });
    // Equality rows (round 2).  (1) The constant b_s is folded into the carried residual of the row's own lane (r~ = rho' + b) and the
    // scale -1/(A_ss + R_s) into the Gram columns g_s, so row s is t_s = r~[s], r~ += g_s t_s.  (2) The six rows of a leg are linear, so
    // their Gauss-Seidel pass is composed ONCE per substep: t = T x with x = the six own residuals at the start of the pass and
    // T = (I - N)^-1, N[s][j] = g_j[s] (j < s), and r~ += sum_j W_j x_j with W_j = sum_{s >= j} g_s T[s][j].  Same arithmetic as the
    // row-by-row pass up to rounding, but the six broadcasts of a pass no longer wait for each other.  (3) f_s = f0_s - sum_it t_s /
    // (A_ss + R_s) needs sum_it t = T (sum_it x): the own residual is accumulated once per pass and T applied after the loop.
    r.x += l < 7 ? A.b : 0.f; r.y += l < 7 ? B.b : 0.f;
    f2 W[2][6];
    float Trow[2][6];                                       // row l of T on lanes 0..5
    sfor<0, 2>([&](auto Lg) {
        constexpr int leg = Lg;
        f2 g[6];
        sfor<0, 6>([&](auto Sx) { g[Sx] = Gp[leg][Sx] * (-eiA[leg][Sx]); });
        float N[6][6], T[6][6];                             // N[s][k] = component `leg` of g_k on lane s, to every lane (k < s)
        sfor<1, 6>([&](auto Sr) { sfor<0, Sr>([&](auto Kk) { N[Sr][Kk] = leg ? dpp<0x150 + Sr>(g[Kk].y) : dpp<0x150 + Sr>(g[Kk].x); }); });
        sfor<1, 6>([&](auto Sr) {
            constexpr int sr = Sr;
            sfor<0, sr>([&](auto Jc) {
                constexpr int j = Jc;
                float acc = N[sr][j];
                sfor<j + 1, sr>([&](auto Kk) { acc += N[sr][Kk] * T[Kk][j]; });
                T[sr][j] = acc;
            });
        });
        sfor<0, 6>([&](auto Jc) {
            constexpr int j = Jc;
            f2 w = g[j];
            sfor<j + 1, 6>([&](auto Sr) { w += g[Sr] * T[Sr][j]; });
            float tr = l == j ? 1.f : 0.f;
            sfor<j + 1, 6>([&](auto Sr) { tr = l == Sr ? T[Sr][j] : tr; });
            APX_PIN("+v"(w.x), "+v"(w.y), "+v"(tr));          // materialise here: fast-math would otherwise re-expand the composition inside the loop
            W[leg][j] = w; Trow[leg][j] = tr;
        });
    });
    f2 racc = {0.f, 0.f};                                   // (left, right) sum over the sweeps of the own residual at the start of the leg's pass
    // Pyramid (round 2): the four SCALED row residuals X_k = wp_k - iA_k u_k (u_k = r_n +- mu r_t) are formed at once.  Row k:
    // df = max(X_k, -f_k) with wp = (alpha - 1) f_k + beta off the chain (alpha = 1 - iA R, beta = -iA b), f_k += df, and the LATER rows'
    // X_j move by -iA_j K[j][k] df with K[j][k] = d_j' G3 d_k (d_k = n + s_k mu t_a(k)); rho' of every lane moves by GpRow[k] df.
    float cam1[NCS][4], cbeta[NCS][4], K10[NCS], K32[NCS];
    f2 K23a[NCS], nK23b[NCS], GpRow[NCS][4];
    sfor<0, NCS>([&](auto Sl) {
        constexpr int s = Sl, leg = s / MAXC, ln = 7 + 3 * (s % MAXC);
        sfor<0, 4>([&](auto K) {
            constexpr int k = K;
            cam1[s][k] = -ciA[s][k] * cR[s]; cbeta[s][k] = -ciA[s][k] * cb[s][k];
            GpRow[s][k] = Gp[leg][ln] + Gp[leg][ln + (k < 2 ? 1 : 2)] * ((k & 1) ? -mu : mu);
        });
        // K[j][k] = kn[k] + s_j (a_j == 1 ? k1[k] : k2[k]); stored pre-multiplied by the LATER row's 1 / (A_jj + R): the scaled residual
        // X_j = wp_j - iA_j u_j of a later row moves by -iA_j K[j][k] df_k, so a row's place on the dependency chain is one fma + one max
        K10[s] = ciA[s][1] * (kn[s][0] - mu * k1[s][0]);
        K23a[s] = f2{ciA[s][2] * (kn[s][0] + mu * k2[s][0]), ciA[s][3] * (kn[s][0] - mu * k2[s][0])};
        nK23b[s] = f2{-ciA[s][2] * (kn[s][1] + mu * k2[s][1]), -ciA[s][3] * (kn[s][1] - mu * k2[s][1])};      // (negated: the update is one fma)
        K32[s] = ciA[s][3] * (kn[s][2] - mu * k2[s][2]);
    });
    f2 ciAp[NCS][2];
    sfor<0, NCS>([&](auto Sl) { ciAp[Sl][0] = f2{ciA[Sl][0], ciA[Sl][1]}; ciAp[Sl][1] = f2{ciA[Sl][2], ciA[Sl][3]}; });
    const f2 cpm = {mu, -mu};
    f2 cfp[NCS][2], cam1p[NCS][2], cbetap[NCS][2];          // pyramid rows (0, 1) / (2, 3) of a slot packed for v_pk_fma_f32 / v_pk_add_f32
    sfor<0, NCS>([&](auto Sl) { sfor<0, 2>([&](auto H) {
        cfp[Sl][H] = f2{cf[Sl][2 * H], cf[Sl][2 * H + 1]}; cam1p[Sl][H] = f2{cam1[Sl][2 * H], cam1[Sl][2 * H + 1]}; cbetap[Sl][H] = f2{cbeta[Sl][2 * H], cbeta[Sl][2 * H + 1]};
    }); });
    for (int it = 0; it < pgs_iters; ++it) {
        sfor<0, 2>([&](auto Lg) {
            constexpr int leg = Lg;
            {
                float x[6];
                sfor<0, 6>([&](auto Sx) { x[Sx] = leg ? dpp<0x150 + Sx>(r.y) : dpp<0x150 + Sx>(r.x); });
                if constexpr (leg) racc.y += r.y; else racc.x += r.x;
                const f2 u = W[leg][0] * x[0] + W[leg][1] * x[1] + W[leg][2] * x[2], v = W[leg][3] * x[3] + W[leg][4] * x[4] + W[leg][5] * x[5];
                r += u + v;
            }
            if (anyl[leg]) {
                const float t = leg ? dpp<0x150 + 6>(r.y) : dpp<0x150 + 6>(r.x);
                const float fn = fmaxf(ef[leg][6] - t * eiA[leg][6], 0.f);
                const float df = fn - ef[leg][6];
                ef[leg][6] = fn;
                r += Gp[leg][6] * df;
            }
            sfor<0, MAXC>([&](auto Sl) {
                constexpr int j = Sl, s = leg * MAXC + j, ln = 7 + 3 * j;
                if (anyc[s]) {
                    const float rn = leg ? dpp<0x150 + ln>(r.y) : dpp<0x150 + ln>(r.x), r1 = leg ? dpp<0x150 + ln + 1>(r.y) : dpp<0x150 + ln + 1>(r.x),
                                r2 = leg ? dpp<0x150 + ln + 2>(r.y) : dpp<0x150 + ln + 2>(r.x);
                    f2 U01 = f2{rn, rn} + cpm * r1, U23 = f2{rn, rn} + cpm * r2;
                    f2 wa = cam1p[s][0] * cfp[s][0] + cbetap[s][0], wb = cam1p[s][1] * cfp[s][1] + cbetap[s][1];      // packed, off the chain
#define APX_PIN2(v) APX_PIN("+v"(v))      /* fix the association: fast-math would gather the corrections into one late sum */
                    APX_PIN2(wa); APX_PIN2(wb);      // fast-math would re-associate them into the chain (pinned as PAIRS: pinning the halves one by one costs a v_mov pair + s_nop per slot and sweep)
                    f2 X01 = wa - ciAp[s][0] * U01, X23 = wb - ciAp[s][1] * U23;        // scaled residuals of the four rows before any of them moved
                    APX_PIN2(X01); APX_PIN2(X23);
                    f2 da, db;
                    da.x = fmaxf(X01.x, -cfp[s][0].x);
                    X01.y -= K10[s] * da.x; X23 -= K23a[s] * da.x;
                    APX_PIN2(X23);
                    da.y = fmaxf(X01.y, -cfp[s][0].y);
                    X23 = pk_fma_hi(da, nK23b[s], X23);                                  // X23 -= K23b da.y
                    APX_PIN2(X23);
                    db.x = fmaxf(X23.x, -cfp[s][1].x);
                    X23.y -= K32[s] * db.x;
                    db.y = fmaxf(X23.y, -cfp[s][1].y);
                    cfp[s][0] += da; cfp[s][1] += db;
                    // the multipliers da.y / db.y are the HIGH halves of the (da.x, da.y) / (db.x, db.y) pairs: taken in place by op_sel (pk_fma_hi) instead of a v_mov into the low
                    // half of a fresh pair (3 moves per slot and sweep, 50 sweeps per substep)
                    r = pk_fma_hi(db, GpRow[s][3], GpRow[s][2] * db.x + pk_fma_hi(da, GpRow[s][1], GpRow[s][0] * da.x + r));
                }
            });
        });
        if (anyx) sfor<0, MAXX>([&](auto K) {               // leg-leg rows last, in pair order (frictionless: one unilateral row each)
            constexpr int k = K, s = 13 + k;
            {       // an env without pair k has xiA = xf = 0: df = 0
                const float t = dpp<0x150 + s>(r.x) + dpp<0x150 + s>(r.y) + xb[k];
                const float fn = fmaxf(xf[k] - t * xiA[k], 0.f);
                const float df = fn - xf[k];
                xf[k] = fn;
                r += GpX[k] * df;
            }
        });
    }
    sfor<0, NCS>([&](auto Sl) { cf[Sl][0] = cfp[Sl][0].x; cf[Sl][1] = cfp[Sl][0].y; cf[Sl][2] = cfp[Sl][1].x; cf[Sl][3] = cfp[Sl][1].y; });
    rA = r.x; rB = r.y;
    PROF(7);
    // ---- z~ = sum_r y~_r F_r back to the dof layout of the finish stage
    float ownA = 0.f, ownB = 0.f;                           // lanes 0..5: f0 - iA sum t; the limit lane's f is uniform (ef[.][6])
    {   // sum_it t_s = (T sum_it x)_s on the row's own lane
        float ta = 0.f, tb = 0.f;
        sfor<0, 6>([&](auto Jc) { ta += Trow[0][Jc] * dpp<0x150 + Jc>(racc.x); tb += Trow[1][Jc] * dpp<0x150 + Jc>(racc.y); });
        if (l < 6) { ownA = f0A - A.invA * ta; ownB = f0B - B.invA * tb; }
    }
    if (l == 6) { ownA = ef[0][6]; ownB = ef[1][6]; }
    sfor<0, NCS>([&](auto Sl) {
        constexpr int s = Sl, leg = s / MAXC, ln = 7 + 3 * (s % MAXC);
        const float dn = cf[s][0] + cf[s][1] + cf[s][2] + cf[s][3], d1 = mu * (cf[s][0] - cf[s][1]), d2 = mu * (cf[s][2] - cf[s][3]);
        float& own = leg ? ownB : ownA;
        own = l == ln ? dn : own; own = l == ln + 1 ? d1 : own; own = l == ln + 2 ? d2 : own;
    });
    if (l >= 13) ownA = ownB = 0.f;
    sfor<0, MAXX>([&](auto K) { if (l == 13 + K && xact[K]) ownA = ownB = xf[K]; });
    {   // All 38 parked values are loaded BEFORE the first store (the compiler cannot prove that a store to the hand-off words does not alias the row store, and orders
        // every later load behind it: the column-by-column form compiled to load -> full LDS wait -> four dependent DPP adds with hazard nops -> store, 19 times = 4 k
        // cycles), and the 32 butterflies advance stage by stage so that independent chains fill each other's DPP hazard slots.
        float t[32];
        sfor<0, 19>([&](auto C) {
            constexpr int c = C;
            const float ja = rows[(2 * c + 0) * 16 + l], jb = rows[(2 * c + 1) * 16 + l];
            if constexpr (c < 6) t[c] = ja * ownA + jb * ownB;
            else { t[c] = ja * ownA; t[c + 13] = jb * ownB; }
        });
        sfor<0, 32>([&](auto I) { t[I] += dpp<0xB1>(t[I]); });       // quad_perm [1,0,3,2]
        sfor<0, 32>([&](auto I) { t[I] += dpp<0x4E>(t[I]); });       // quad_perm [2,3,0,1]
        sfor<0, 32>([&](auto I) { t[I] += dpp<0x141>(t[I]); });      // row_half_mirror
        sfor<0, 32>([&](auto I) { t[I] += dpp<0x140>(t[I]); });      // row_mirror
        // (the butterfly leaves the bit-identical total on every lane: all 16 lanes store it to the same word, no exec-mask region per column)
        sfor<0, 32>([&](auto I) { S.W(WK_ZT + I) = t[I]; });
    }
    if (l == 0) {
        // contact forces to the row store (foot-force readout in the finish stage)
        sfor<0, NCS>([&](auto Sl) {
            constexpr int s = Sl, leg = s / MAXC, j = s % MAXC;
            float* cr = rows + R4_CON + R4_CONSZ * s;
            sfor<0, 4>([&](auto K) { cr[12 + K] = cf[s][K]; });
        });
    }
}

}  // namespace c4
