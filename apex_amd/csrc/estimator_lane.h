// The reference's state estimator (state_output_step of libcassiemujoco.so; interface cassie/cassiemujoco/include/StateOutput.h:33-36,
// consumed by CassieEnv.get_full_state, cassie/cassie.py:793,817-850) for one env per 16-lane DPP row, fp32.  The algorithm was decoded
// from the binary and is pinned by golden G11 through its fp64 restatement oracle/cassie_estimator.cpp; this file is the same filter bank
// written for the lane layout of the substep kernel:
//
//   leg kinematics      lanes 0..6 = the left chain hip roll, yaw, pitch, knee, shin, tarsus, foot (SENSOR angles, foot = motor angle),
//                       lanes 8..14 = the right chain: local transforms, then a 3-round DPP prefix product (row_shr 1, 2, 4)
//   heel springs        the achilles-rod closure is a 13-term cosine series: one term per lane (0..12), both legs in two register slots,
//                       two Newton steps from the previous solution (the reference's Levenberg-Marquardt converges to the same root from
//                       the same warm start: the kernel is compared with it at 1e-5 rad)
//   foot force          spring Jacobian with the tarsus on the closure, 2 x 3 basic solution (pivot columns by selects), lanes 5 / 13
//   Kalman filters      covariance ROW per lane (lanes 0..5), the x, y and z filters in three register slots; the diagonal measurement noise
//                       makes the batch update a sequence of scalar updates, each: 8 row broadcasts, rsqrt, 7 fma per lane, and symmetric
//                       to the bit (P[l][b] -= u_l u_b with a commutative product)
//
// Persistent state (heel solution, three filters, terrain) has no room in the env's LDS region (the CU's 160 KB hold exactly four single-wave
// workgroups): it lives in an env-major HBM record, 168 floats per env, that the row lanes load with dwordx4 at the top of the io stage and store at
// its end: 1.3 KB per env and substep through L2, no LDS staging.
#pragma once
#include "cassie_lane.h"

namespace est {
using c4::V3; using c4::Q4; using c4::dpp; using c4::sfor; using c4::rcpf; using c4::nonfinite;

constexpr int REC = 168;                       // floats per env: lane r < 7 owns [24 r, 24 r + 24)
// lanes 0..5: Px[6] Py[6] Pz[6] xX xY xZ pad3;  lane 6: heelL heelR terrain inited | L foot pos3 pad | R foot pos3 pad | L foot quat4 | R foot quat4 | pad4
constexpr float E_DT = 0.0005f, E_G = 9.806f, E_M = 31.f, E_W2 = 9.806f /* g / pendulum height 1.0 */;
constexpr float K_SHIN = 1500.f, K_HEEL = 1250.f;
constexpr float HEEL_LIM = 0.78539816f - 1e-6f;
// closure series (oracle/cassie_estimator.cpp kHeelTerm; the 2 X term of 7.5e-19 is dropped): lane -> coefficient, phase, multiples of knee / shin / tarsus / heel
constexpr float HT_C[13] = {-0.00015856770032083081f, -0.0079482557784711465f, -0.0085406453671835261f, -4.3720724282115655e-05f, 0.018481200740211659f,
                            -0.02882698448043499f, -0.02856193646166626f, -0.018328433377681426f, 0.067028201125701736f, 0.1044527473921103f,
                            -0.0051004621727941879f, -2.8055954234903667e-05f, -0.10358933105069197f};
constexpr float HT_P[13] = {0.60103412848472038f, 1.1775614272403667f, 1.4203984658003772f, 0.64700411135490599f, 0.66324136802307054f, 1.1315914443701811f,
                            -0.50615334252706334f, -1.1693947105501337f, -0.61727138515288482f, 0.045969982870185652f, 0.51432005921729607f,
                            -0.016237256668164488f, -0.55212332539724895f};
constexpr float HT_C0 = -0.0245857384f;
//                               T-X  TSK  X  TSK-X  K  T  TSKX XTS  S  KS  TS  TS-X  TX
constexpr unsigned bits13(int a0, int a1, int a2, int a3, int a4, int a5, int a6, int a7, int a8, int a9, int a10, int a11, int a12) {
    return (unsigned)a0 | a1 << 1 | a2 << 2 | a3 << 3 | a4 << 4 | a5 << 5 | a6 << 6 | a7 << 7 | a8 << 8 | a9 << 9 | a10 << 10 | a11 << 11 | a12 << 12;
}
constexpr unsigned MK = bits13(0, 1, 0, 1, 1, 0, 1, 0, 0, 1, 0, 0, 0), MS = bits13(0, 1, 0, 1, 0, 0, 1, 1, 1, 1, 1, 1, 0), MT = bits13(1, 1, 0, 1, 0, 1, 1, 1, 0, 0, 1, 1, 1),
                   MXP = bits13(0, 0, 1, 0, 0, 0, 1, 1, 0, 0, 0, 0, 1), MXN = bits13(1, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 1, 0);      // heel enters with +1 / -1
constexpr float FOOT_OFF_X = 0.017620176f, FOOT_OFF_Y = 0.052189981f;      // origin of the routine's foot frame inside the foot body
constexpr float IMU_RX = 0.03155f, IMU_RZ = -0.079996f;                    // the routine's IMU offset (cassie.xml:265 has -0.07996)

template <int N> __device__ __forceinline__ float bc(float x) { return dpp<0x150 + N>(x); }      // row_newbcast: lane N of the env's row to all 16
template <int N, class F> __device__ __forceinline__ float selN(int i, F f) {      // f(i) by a chain of plain selects (a lane-indexed table would be a VMEM load)
    float r = f(std::integral_constant<int, N - 1>{});
    sfor<0, N - 1>([&](auto K) { constexpr int k = N - 2 - K; r = i == k ? f(std::integral_constant<int, k>{}) : r; });
    return r;
}
__device__ __forceinline__ V3 qrot(Q4 q, V3 v) {       // rotate v by the unit quaternion q
    const V3 u = {q.x, q.y, q.z};
    const V3 t = c4::cross(u, v) * 2.f;
    return v + t * q.w + c4::cross(u, t);
}
struct F6 { float P[6]; float x; };             // a filter's covariance row and state entry of this lane (zero on lanes >= 6)

// scalar measurement z = x[I] - x[J] (J < 0: z = x[I]) with variance r
template <int I, int J> __device__ __forceinline__ void kf_update(F6& f, float z, float r) {
    float Ph = f.P[I];
    if constexpr (J >= 0) Ph -= f.P[J];
    float hP[6];
    sfor<0, 6>([&](auto B) { hP[B] = bc<B>(Ph); });                     // (h^T P)[b] = (P h)[b]: P is symmetric
    float s = hP[I] + r, hx = bc<I>(f.x);
    if constexpr (J >= 0) { s -= hP[J]; hx -= bc<J>(f.x); }
    const float rs = rsqrtf(s), u = Ph * rs;
    f.x += u * rs * (z - hx);
    sfor<0, 6>([&](auto B) { f.P[B] -= u * (hP[B] * rs); });            // u_l u_b: commutative, so P stays symmetric to the bit
}
// P <- A P A^T + diag(q), A = I + dt e0 e1^T + e1 a^T (a[1] == 0)
__device__ __forceinline__ void kf_predict_cov(F6& f, const float (&a)[6], const float (&q)[6], int l) {
    float w = 0.f;
    sfor<0, 6>([&](auto K) { if constexpr (K != 1) w += f.P[K] * a[K]; });
    const float d = f.P[1], c0 = l == 0 ? E_DT : 0.f, c1 = l == 1 ? 1.f : 0.f;
    float Bm[6];
    sfor<0, 6>([&](auto B) { Bm[B] = f.P[B] + c0 * bc<B>(d) + c1 * bc<B>(w); });
    float t = 0.f;
    sfor<0, 6>([&](auto K) { if constexpr (K != 1) t += a[K] * Bm[K]; });
    f.P[0] = Bm[0] + E_DT * Bm[1]; f.P[1] = Bm[1] + t;
    sfor<2, 6>([&](auto B) { f.P[B] = Bm[B]; });
    sfor<0, 6>([&](auto B) { f.P[B] += l == B ? q[B] : 0.f; });
}
// horizontal filter: EKF on the linear inverted pendulum; state [p, v, foot L, foot R, load share, disturbance force]
__device__ __forceinline__ void hfilter_step(F6& f, int l, float zL, float zR, float fl, float fr, float acc) {
    const float p = bc<0>(f.x), v = bc<1>(f.x), pL = bc<2>(f.x), pR = bc<3>(f.x), al = bc<4>(f.x), fd = bc<5>(f.x);
    const float tot = fl + fr;
    const bool contact = !(1.f > tot);
    const float alpha_m = contact ? fl * rcpf(contact ? tot : 1.f) : 0.5f, cw = contact ? E_DT * E_W2 : 0.f;
    const float a[6] = {cw, 0.f, -cw * al, -cw * (1.f - al), -cw * (pL - pR), contact ? E_DT / E_M : 0.f};
    const float v1 = v + cw * (p - al * pL - (1.f - al) * pR) + a[5] * fd;
    f.x = l == 0 ? p + E_DT * v : (l == 1 ? v1 : f.x);
    const float q[6] = {1e-8f, 1e-8f, 50.f > fl ? 1e-6f : 1e-10f, 50.f > fr ? 1e-6f : 1e-10f, 1e-5f, 1e-2f};
    kf_predict_cov(f, a, q, l);
    kf_update<0, 2>(f, zL, 1e-6f);
    kf_update<0, 3>(f, zR, 1e-6f);
    kf_update<4, -1>(f, alpha_m, 1e-6f);
    kf_update<1, -1>(f, v + E_DT * acc, 1.f);
}
// vertical filter: [z, vz, foot L z, foot R z, disturbance force] (row 5 / column 5 stay zero)
__device__ __forceinline__ void zfilter_step(F6& f, int l, float zL, float zR, float fl, float fr) {
    const float p = bc<0>(f.x), v = bc<1>(f.x), fd = bc<4>(f.x);
    const float a[6] = {0.f, 0.f, 0.f, 0.f, E_DT / E_M, 0.f};
    const float v1 = v + a[4] * fd + E_DT * ((fl + fr) * (1.f / E_M) - E_G);
    f.x = l == 0 ? p + E_DT * v : (l == 1 ? v1 : f.x);
    const float q[6] = {1e-8f, 1e-8f, 50.f > fl ? 1e-6f : 1e-10f, 50.f > fr ? 1e-6f : 1e-10f, 0.01f, 0.f};
    kf_predict_cov(f, a, q, l);
    kf_update<0, 2>(f, zL, 1e-6f);
    kf_update<0, 3>(f, zR, 1e-6f);
}

typedef float f4 __attribute__((ext_vector_type(4)));
struct Rec { f4 v[6]; };                         // this lane's 24 floats of the env record
__device__ __forceinline__ Rec rec_load(const float* wk, int env, int l) {
    Rec r;
    const f4* g = (const f4*)(wk + (size_t)env * REC) + 6 * (l < 7 ? l : 6);
    sfor<0, 6>([&](auto K) { r.v[K] = g[K]; });
    return r;
}
__device__ __forceinline__ void rec_store(float* wk, int env, int l, const Rec& r) {
    if (l < 7) { f4* g = (f4*)(wk + (size_t)env * REC) + 6 * l; sfor<0, 6>([&](auto K) { g[K] = r.v[K]; }); }
}

// One estimator update.  Reads the encoder outputs (F_SO motor / joint positions, written earlier in the io stage by other lanes: the caller
// fences) and the IMU snapshot from the env's LDS region, writes translationalVelocity, translationalAcceleration and the height entry.
__device__ __forceinline__ void est_step_lane(const St& S, Rec& rec) {
    const int l = threadIdx.x & 15;
    // ---------------------------------------------------------------- unpack the record
    F6 fx, fy, fz;
    sfor<0, 6>([&](auto B) { constexpr int b = B; fx.P[b] = rec.v[b / 4][b % 4]; fy.P[b] = rec.v[(6 + b) / 4][(6 + b) % 4]; fz.P[b] = rec.v[(12 + b) / 4][(12 + b) % 4]; });
    fx.x = rec.v[4][2]; fy.x = rec.v[4][3]; fz.x = rec.v[5][0];
    const bool row = l < 6;
    sfor<0, 6>([&](auto B) { fx.P[B] = row ? fx.P[B] : 0.f; fy.P[B] = row ? fy.P[B] : 0.f; fz.P[B] = row ? fz.P[B] : 0.f; });
    fx.x = row ? fx.x : 0.f; fy.x = row ? fy.x : 0.f; fz.x = row ? fz.x : 0.f;
    // a non-finite word anywhere in the stored record (fast-math min / max clamps would turn some of them into finite garbage instead of carrying them to
    // the test after the update): this update starts from state_output_setup.  One hardware add chain per lane carries a NaN / inf to a single bit test.
    float rsum = 0.f;
    sfor<0, 6>([&](auto K) { rsum += (rec.v[K][0] + rec.v[K][1]) + (rec.v[K][2] + rec.v[K][3]); });
    const bool stale = c4::red16(nonfinite(rsum) ? 1.f : 0.f) > 0.f;
    if (stale) {
        sfor<0, 6>([&](auto B) { fx.P[B] = fy.P[B] = fz.P[B] = 0.f; });
        fx.x = fy.x = fz.x = 0.f;
    }
    // (the broadcasts are taken unconditionally and pinned: `c ? dpp(v) : 0` compiles to a DPP move under an exec mask, and a DPP read of a lane that the
    // mask disabled returns 0.  Harmless here, `stale` is uniform over the env's row, but tools/dpp_audit.py keeps the library free of the shape.)
    float b0 = bc<6>(rec.v[0][0]), b1 = bc<6>(rec.v[0][1]), b2 = bc<6>(rec.v[0][2]), b3 = bc<6>(rec.v[0][3]);
    APX_PIN("+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3));
    float heel[2] = {stale ? 0.f : b0, stale ? 0.f : b1};
    float terr = stale ? 0.f : b2;
    const float inited = stale ? 0.f : b3;
    // ---------------------------------------------------------------- heel springs: two Newton steps on the closure series, one term per lane
    const int lt = l < 13 ? l : 12;
    const float hc = l < 13 ? selN<13>(lt, [](auto K) { return HT_C[K]; }) : 0.f, hp = selN<13>(lt, [](auto K) { return HT_P[K]; });
    const float nk = (float)((MK >> lt) & 1u), ns = (float)((MS >> lt) & 1u), nt = (float)((MT >> lt) & 1u);
    const float nx = (float)((MXP >> lt) & 1u) - (float)((MXN >> lt) & 1u);
    float u0[2], gS[2], gT[2], gX[2];
    sfor<0, 2>([&](auto Lg) {
        constexpr int lg = Lg;
        u0[lg] = nk * S(F_SO + SO_MPOS + 5 * lg + 3) + ns * S(F_SO + SO_JPOS + 3 * lg) + nt * S(F_SO + SO_JPOS + 3 * lg + 1) + hp;
    });
    sfor<0, 2>([&](auto It) {
        sfor<0, 2>([&](auto Lg) {
            constexpr int lg = Lg;
            float sn, cs;
            __sincosf(u0[lg] + nx * heel[lg], &sn, &cs);
            const float r = c4::red16(hc * cs) + HT_C0, ds = -hc * sn;
            gX[lg] = c4::red16(ds * nx);
            if constexpr (It == 1) { gS[lg] = c4::red16(ds * ns); gT[lg] = c4::red16(ds * nt); }
            heel[lg] = fminf(fmaxf(heel[lg] - r * rcpf(gX[lg]), -HEEL_LIM), HEEL_LIM);      // (the Jacobian below takes the gradient of the second evaluation point: O(step^2) off the root)
        });
    });
    PROF2(33);
    // ---------------------------------------------------------------- leg kinematics: local transforms, prefix product over the chain
    const int k = l & 7, lg = l >> 3;
    constexpr unsigned long long KB = c4::nib(2, 3, 4, 6, 8, 9, 13, 2, 0, 0, 0, 0);
    const int cb = ct_body_word(c4::nibble(KB, k) - 2, lg, 0);      // body id nibble + 12 lg; consecutive words of a slot are 2 apart (pair layout)
    const int aidx = k < 4 ? SO_MPOS + 5 * lg + k : (k == 4 ? SO_JPOS + 3 * lg : (k == 5 ? SO_JPOS + 3 * lg + 1 : SO_MPOS + 5 * lg + 4));
    float jref = 0.f; jref = k == 3 ? cmt::ct_jnt_ref[8] : jref; jref = k == 5 ? cmt::ct_jnt_ref[10] : jref;      // knee, tarsus
    static_assert(cmt::ct_jnt_ref[19] == cmt::ct_jnt_ref[8] && cmt::ct_jnt_ref[21] == cmt::ct_jnt_ref[10], "joint refs mirror");
    Q4 q; V3 o;
    {
        float sn, cs;
        __sincosf(0.5f * (S(F_SO + aidx) - jref), &sn, &cs);
        q = c4::qmul(Q4{ctf(cb + 12), ctf(cb + 14), ctf(cb + 16), ctf(cb + 18)}, Q4{cs, 0.f, 0.f, sn});
        o = {ctf(cb), ctf(cb + 2), ctf(cb + 4)};
        if (k == 7) { q = {1.f, 0.f, 0.f, 0.f}; o = {0.f, 0.f, 0.f}; }
    }
    sfor<0, 3>([&](auto Rn) {
        constexpr int d = 1 << Rn, ctl = 0x110 + d;                          // row_shr:d
        const Q4 qa = {dpp<ctl>(q.w), dpp<ctl>(q.x), dpp<ctl>(q.y), dpp<ctl>(q.z)};
        const V3 oa = {dpp<ctl>(o.x), dpp<ctl>(o.y), dpp<ctl>(o.z)};
        const bool ok = k >= d;                                              // the source lane is in the same 8-lane chain
        const V3 on = oa + qrot(qa, o);
        const Q4 qn = c4::qmul(qa, q);
        o = {ok ? on.x : o.x, ok ? on.y : o.y, ok ? on.z : o.z};
        q = {ok ? qn.w : q.w, ok ? qn.x : q.x, ok ? qn.y : q.y, ok ? qn.z : q.z};
    });
    PROF2(34);
    // foot-frame origin on the foot lanes (k == 6), handed to the shin / tarsus lanes (row_shl 2 / 1) for d foot / d angle = z x (foot - joint origin)
    const V3 pf = o + qrot(q, V3{FOOT_OFF_X, FOOT_OFF_Y, 0.f});
    const V3 p1 = {dpp<0x101>(pf.x), dpp<0x101>(pf.y), dpp<0x101>(pf.z)}, p2 = {dpp<0x102>(pf.x), dpp<0x102>(pf.y), dpp<0x102>(pf.z)};
    const V3 pfoot = k == 4 ? p2 : p1;
    const V3 dj = c4::cross(qrot(q, V3{0.f, 0.f, 1.f}), pfoot - o);        // valid on k = 4 (shin) and k = 5 (tarsus)
    // tarsus lanes (5, 13): spring Jacobian with the tarsus on the closure, basic solution of the 2 x 3 system, world force
    const V3 dS = {dpp<0x111>(dj.x), dpp<0x111>(dj.y), dpp<0x111>(dj.z)};  // from the shin lane (row_shr:1)
    const Q4 pq = {S(F_SNAP + SN_QUAT), S(F_SNAP + SN_QUAT + 1), S(F_SNAP + SN_QUAT + 2), S(F_SNAP + SN_QUAT + 3)};
    float fzw;
    {
        const float gs = lg ? gS[1] : gS[0], gt = lg ? gT[1] : gT[0], gx = lg ? gX[1] : gX[0], hl = lg ? heel[1] : heel[0];
        const float igt = rcpf(gt);
        const V3 av = dS - dj * (gs * igt), bv = dj * (-gx * igt);
        const float M0[3] = {-av.x, -av.y, -av.z}, M1[3] = {-bv.x, -bv.y, -bv.z};
        const float t0 = K_SHIN * S(F_SO + SO_JPOS + 3 * lg), t1 = K_HEEL * hl;
        // pivot 1: the column of largest norm; pivot 2: the largest remainder after projecting pivot 1 out
        const float n0 = M0[0] * M0[0] + M1[0] * M1[0], n1 = M0[1] * M0[1] + M1[1] * M1[1], n2 = M0[2] * M0[2] + M1[2] * M1[2];
        const int j1 = (n1 > n0 && n1 >= n2) ? 1 : (n2 > n0 ? 2 : 0);
        const float a0 = j1 == 0 ? M0[0] : (j1 == 1 ? M0[1] : M0[2]), a1 = j1 == 0 ? M1[0] : (j1 == 1 ? M1[1] : M1[2]);
        const float in1 = rcpf(fmaxf(a0 * a0 + a1 * a1, 1e-30f));
        float rem[3];
        sfor<0, 3>([&](auto Jc) { const float pr = (a0 * M0[Jc] + a1 * M1[Jc]) * in1, ra = M0[Jc] - a0 * pr, rb = M1[Jc] - a1 * pr; rem[Jc] = j1 == Jc ? -1.f : ra * ra + rb * rb; });
        const int j2 = (rem[1] > rem[0] && rem[1] >= rem[2]) ? 1 : (rem[2] > rem[0] ? 2 : 0);
        const float b0 = j2 == 0 ? M0[0] : (j2 == 1 ? M0[1] : M0[2]), b1 = j2 == 0 ? M1[0] : (j2 == 1 ? M1[1] : M1[2]);
        const float idet = rcpf(a0 * b1 - b0 * a1);
        const float s1 = (t0 * b1 - b0 * t1) * idet, s2 = (a0 * t1 - t0 * a1) * idet;
        const V3 f = {(j1 == 0 ? s1 : 0.f) + (j2 == 0 ? s2 : 0.f), (j1 == 1 ? s1 : 0.f) + (j2 == 1 ? s2 : 0.f), (j1 == 2 ? s1 : 0.f) + (j2 == 2 ? s2 : 0.f)};
        fzw = qrot(pq, f).z;
    }
    const float fl = fmaxf(0.f, -bc<5>(fzw)), fr = fmaxf(0.f, -bc<13>(fzw));
    // world-aligned foot offsets (foot lanes 6 / 14) to every lane
    const V3 fw = qrot(pq, pf);
    const float lfx = bc<6>(fw.x), lfy = bc<6>(fw.y), lfz = bc<6>(fw.z), rfx = bc<14>(fw.x), rfy = bc<14>(fw.y), rfz = bc<14>(fw.z);
    // ---------------------------------------------------------------- translationalAcceleration (pelvis frame) and its world-aligned copy
    const V3 wg = {S(F_SNAP + SN_GYRO), S(F_SNAP + SN_GYRO + 1), S(F_SNAP + SN_GYRO + 2)};
    const V3 cen = c4::cross(wg, c4::cross(wg, V3{IMU_RX, 0.f, IMU_RZ}));
    const c4::M3 Rp = c4::q2m(pq);
    const V3 ab = {S(F_SNAP + SN_ACC) - Rp.m[6] * E_G - cen.x, S(F_SNAP + SN_ACC + 1) - Rp.m[7] * E_G - cen.y, S(F_SNAP + SN_ACC + 2) - Rp.m[8] * E_G - cen.z};
    const V3 aw = c4::mul(Rp, ab);
    PROF2(35);
    // ---------------------------------------------------------------- first call after state_output_setup: zero pelvis state, feet at MINUS the offset
    if (inited == 0.f) {
        sfor<0, 6>([&](auto B) { const float dg = (row && l == B) ? 1e-6f : 0.f; fx.P[B] = dg; fy.P[B] = dg; fz.P[B] = (B < 5) ? dg : 0.f; });
        fx.x = l == 2 ? -lfx : (l == 3 ? -rfx : (l == 4 ? 0.5f : 0.f));
        fy.x = l == 2 ? -lfy : (l == 3 ? -rfy : (l == 4 ? 0.5f : 0.f));
        fz.x = l == 2 ? -lfz : (l == 3 ? -rfz : (l == 4 ? E_M * E_G : 0.f));
    }
    // the estimator's own discrete switches (process noise of a foot state below / above 50 N of estimated load, terrain update above 1 N): part of the row-set signature
    // of this substep (I_ROWSET, folded by the constraint stage) - a load within fp32 round-off of 50 N flips a filter gain, which is a set difference, not round-off
    if (l == 0) S.W(c4::WK_MISC + 6) = (float)((50.f > fl ? 1 : 0) | (50.f > fr ? 2 : 0) | (fl + fr > 1.f ? 4 : 0));
    hfilter_step(fx, l, -lfx, -rfx, fl, fr, aw.x);
    hfilter_step(fy, l, -lfy, -rfy, fl, fr, aw.y);
    zfilter_step(fz, l, -lfz, -rfz, fl, fr);
    PROF2(36);
    // ---------------------------------------------------------------- terrain height: low-pass of the load-weighted kinematic foot height while loaded
    const float pz = bc<0>(fz.x);
    if (fl + fr > 1.f) {
        const float a = fl * rcpf(fl + fr);
        terr = 0.0004997501249375313f * (a * (pz + lfz) + (1.f - a) * (pz + rfz)) + 0.9995002498750625f * terr;
    }
    // ---------------------------------------------------------------- outputs (state_out_t) and the record
    const float vx = bc<1>(fx.x), vy = bc<1>(fy.x), vz = bc<1>(fz.x);      // (DPP reads outside the lane-0 region: a disabled source lane reads as 0)
    if (l == 0) {
        S(F_SO + SO_TVEL) = vx; S(F_SO + SO_TVEL + 1) = vy; S(F_SO + SO_TVEL + 2) = vz;
        S(F_SO + SO_TACC) = ab.x; S(F_SO + SO_TACC + 1) = ab.y; S(F_SO + SO_TACC + 2) = ab.z;
        S(F_SO + SO_HEIGHT) = pz - terr;
    }
    // a diverged env (non-finite sensors) must not poison the persistent record: restart the estimator (state_output_setup) on the next substep
    // (c4::nonfinite: a bit-pattern test that survives -ffast-math).  One NaN anywhere in a filter reaches every row's state within the update.
    const bool badl = nonfinite(fx.x) || nonfinite(fy.x) || nonfinite(fz.x) || nonfinite(heel[0]) || nonfinite(heel[1]) || nonfinite(terr) || nonfinite(fx.P[0]) || nonfinite(fy.P[0]) || nonfinite(fz.P[0]);
    const bool bad = c4::red16(badl ? 1.f : 0.f) > 0.f;
    float keep = bad ? 0.f : 1.f;
    if (bad) {
        sfor<0, 6>([&](auto B) { fx.P[B] = fy.P[B] = fz.P[B] = 0.f; });
        fx.x = fy.x = fz.x = 0.f; heel[0] = heel[1] = 0.f; terr = 0.f;
    }
    sfor<0, 6>([&](auto B) { constexpr int b = B; rec.v[b / 4][b % 4] = fx.P[b]; rec.v[(6 + b) / 4][(6 + b) % 4] = fy.P[b]; rec.v[(12 + b) / 4][(12 + b) % 4] = fz.P[b]; });
    rec.v[4][2] = fx.x; rec.v[4][3] = fy.x; rec.v[5][0] = fz.x;
    // leftFoot / rightFoot .position and .orientation (input_profile "min", cassie.py:829-837): foot body frame times the routine's constant frame offset
    Q4 fq = c4::qmul(q, Q4{0.24184476264797528f, -0.24184476264797528f, -0.66446302438867138f, 0.66446302438867138f});
    if (fq.w < 0.f) fq = {-fq.w, -fq.x, -fq.y, -fq.z};
    const float rpx = bc<14>(pf.x), rpy = bc<14>(pf.y), rpz = bc<14>(pf.z), rqw = bc<14>(fq.w), rqx = bc<14>(fq.x), rqy = bc<14>(fq.y), rqz = bc<14>(fq.z);
    if (l == 6) {
        rec.v[0][0] = heel[0]; rec.v[0][1] = heel[1]; rec.v[0][2] = terr; rec.v[0][3] = keep;
        rec.v[1] = bad ? f4{0.f, 0.f, 0.f, 0.f} : f4{pf.x, pf.y, pf.z, 0.f}; rec.v[2] = bad ? f4{0.f, 0.f, 0.f, 0.f} : f4{rpx, rpy, rpz, 0.f};
        rec.v[3] = bad ? f4{1.f, 0.f, 0.f, 0.f} : f4{fq.w, fq.x, fq.y, fq.z}; rec.v[4] = bad ? f4{1.f, 0.f, 0.f, 0.f} : f4{rqw, rqx, rqy, rqz};
    }
}

}  // namespace est
