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
static_assert(L4_WK + WK_TOTAL <= L4_ROWS && L4_ROWS % 4 == 0 && L4_ROWS + 4 * CH_TOTAL <= L4_ES && L4_ES % 64 == 16, "per-env LDS layout");

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

// position of local column c (0..18) inside the packed value list of a whitened support, -1 = structurally zero
__device__ __forceinline__ int pos_PL(int c) { return c <= 8 ? c : (c >= 12 && c <= 14) ? c - 3 : c >= 16 ? c - 4 : -1; }
__device__ __forceinline__ int pos_AC(int c) { return c <= 15 ? c : -1; }
__device__ __forceinline__ int pos_FT(int c) { return c <= 8 ? c : (c >= 12 && c <= 14) ? c - 3 : c == 18 ? 12 : -1; }

struct Pair { float a, p; };
// gather this lane's (ya, yp) of a row of leg LEG from the packed list at float offset `off` of the env's row store
template <int LEG, class POS>
__device__ __forceinline__ Pair lane_cols(const float* rows, int off, int l, POS pos) {
    const int cA = l < 13 ? 6 + l : l - 13, cP = l - 10;
    const int iA = pos(cA), iP = l < 13 ? -1 : pos(cP);
    const float vA = iA >= 0 ? rows[off + iA] : 0.f, vP = iP >= 0 ? rows[off + iP] : 0.f;
    if (l < 13) return LEG == 0 ? Pair{vA, 0.f} : Pair{0.f, vA};
    return Pair{vA, vP};
}

// Projected Gauss-Seidel, leg-major (left: 6 equality rows, limit, contacts; then right), in the whitened space.
// rows = the env's row store (gen-3 chunk format, written by the row stage); S.W(WK_ZT) = warm-started z~.
__device__ __forceinline__ void stage_pgs_lane(const St& S, const float* rows, int pgs_iters) {
    const int l = threadIdx.x & 15;
    float zA = S.W(WK_ZT + (l < 13 ? 6 + l : l - 13)), zB = S.W(WK_ZT + (l < 13 ? 19 + l : l - 10));
    const int ncon[2] = {(int)S.W(WK_MISC + 0), (int)S.W(WK_MISC + 1)}, nlim[2] = {(int)S.W(WK_MISC + 2), (int)S.W(WK_MISC + 3)};
    const float mu = S(F_FRIC);
    float ea[12], ep[12], eb[12], eR[12], eiA[12], ef[12];
    sfor<0, 12>([&](auto Rw) {
        constexpr int row = Rw, ch = CH_EQ + 5 * row;
        Pair v;
        if constexpr ((row % 6) < 3) v = lane_cols<row / 6>(rows, 4 * ch, l, pos_PL); else v = lane_cols<row / 6>(rows, 4 * ch, l, pos_AC);
        ea[row] = v.a; ep[row] = v.p;
        eb[row] = rows[4 * ch + 16]; eR[row] = rows[4 * ch + 17]; eiA[row] = rows[4 * ch + 18]; ef[row] = rows[4 * ch + 19];
    });
    float la[2] = {0.f, 0.f}, lp[2] = {0.f, 0.f}, lb[2] = {0.f, 0.f}, lR[2] = {0.f, 0.f}, liA[2] = {0.f, 0.f}, lf[2] = {0.f, 0.f};
    sfor<0, 2>([&](auto Lg) {
        constexpr int ch = CH_LIM + 6 * Lg;
        if (nlim[Lg]) {
            const Pair v = lane_cols<Lg>(rows, 4 * ch, l, [](int c) { return c; });
            la[Lg] = v.a; lp[Lg] = v.p;
            lb[Lg] = rows[4 * ch + 20]; lR[Lg] = rows[4 * ch + 21]; liA[Lg] = rows[4 * ch + 22]; lf[Lg] = rows[4 * ch + 23];
        }
    });
    constexpr int NCS = 2 * MAXC;
    float na[NCS], np[NCS], t1a[NCS], t1p[NCS], t2a[NCS], t2p[NCS], cG[NCS][6], cR[NCS], cb[NCS][4], cf[NCS][4], ciA[NCS][4];
    sfor<0, NCS>([&](auto Sl) {
        constexpr int s = Sl, leg = s / MAXC, ch = CH_CON + 14 * (3 * leg + s % MAXC);
        na[s] = np[s] = t1a[s] = t1p[s] = t2a[s] = t2p[s] = 0.f;
        if ((s % MAXC) < ncon[leg]) {
            const Pair vn = lane_cols<leg>(rows, 4 * ch, l, pos_FT), v1 = lane_cols<leg>(rows, 4 * ch + 13, l, pos_FT), v2 = lane_cols<leg>(rows, 4 * ch + 26, l, pos_FT);
            na[s] = vn.a; np[s] = vn.p; t1a[s] = v1.a; t1p[s] = v1.p; t2a[s] = v2.a; t2p[s] = v2.p;
            sfor<0, 6>([&](auto K) { cG[s][K] = rows[4 * ch + 40 + K]; });
            cR[s] = rows[4 * ch + 46];
            sfor<0, 4>([&](auto K) {
                constexpr int k = K;
                cb[s][k] = rows[4 * ch + 48 + k]; cf[s][k] = rows[4 * ch + 52 + k];
                const float sm = ((k & 1) ? -mu : mu), gnj = k < 2 ? cG[s][1] : cG[s][2], gjj = k < 2 ? cG[s][3] : cG[s][5];
                ciA[s][k] = __frcp_rn(cG[s][0] + 2.f * sm * gnj + mu * mu * gjj + cR[s]);      // A_kk + R of row n + s mu t_j
            });
        }
    });
    for (int it = 0; it < pgs_iters; ++it) {
        sfor<0, 2>([&](auto Lg) {
            constexpr int leg = Lg;
            sfor<0, 6>([&](auto Rw) {
                constexpr int r = 6 * leg + Rw;
                const float t = red16(ea[r] * zA + ep[r] * zB);
                const float df = -(t + eb[r] + eR[r] * ef[r]) * eiA[r];
                ef[r] += df;
                zA += ea[r] * df; zB += ep[r] * df;
            });
            if (nlim[leg]) {
                const float t = red16(la[leg] * zA + lp[leg] * zB);
                float fn = lf[leg] - (t + lb[leg] + lR[leg] * lf[leg]) * liA[leg];
                fn = fn < 0.f ? 0.f : fn;
                const float df = fn - lf[leg];
                lf[leg] = fn;
                zA += la[leg] * df; zB += lp[leg] * df;
            }
            sfor<0, MAXC>([&](auto Sl) {
                constexpr int s = leg * MAXC + Sl;
                if (Sl < ncon[leg]) {
                    float rn = red16(na[s] * zA + np[s] * zB), r1 = red16(t1a[s] * zA + t1p[s] * zB), r2 = red16(t2a[s] * zA + t2p[s] * zB);
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
                        if constexpr (k < 2) { rn += df * (gnn + sm * gn1); r1 += df * (gn1 + sm * g11); r2 += df * (gn2 + sm * g12); sd1 += sm * df; }
                        else { rn += df * (gnn + sm * gn2); r1 += df * (gn1 + sm * g12); r2 += df * (gn2 + sm * g22); sd2 += sm * df; }
                        sdn += df;
                    });
                    zA += na[s] * sdn + t1a[s] * sd1 + t2a[s] * sd2;
                    zB += np[s] * sdn + t1p[s] * sd1 + t2p[s] * sd2;
                }
            });
        });
    }
    S.W(WK_ZT + (l < 13 ? 6 + l : l - 13)) = zA; S.W(WK_ZT + (l < 13 ? 19 + l : l - 10)) = zB;
    // contact forces back to the row store (foot-force readout in the finish stage)
    if (l == 0) {
        float* w = const_cast<float*>(rows);
        sfor<0, NCS>([&](auto Sl) {
            constexpr int s = Sl, leg = s / MAXC, ch = CH_CON + 14 * (3 * leg + s % MAXC);
            if ((s % MAXC) < ncon[leg]) sfor<0, 4>([&](auto K) { w[4 * ch + 52 + K] = cf[s][K]; });
        });
    }
}

}  // namespace c4
