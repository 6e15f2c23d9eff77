// Batched Cassie-v0 environment for gfx950: one HIP launch = one env step (simrate physics substeps at 2 kHz,
// reward, termination, command resampling, observation, optional auto-reset) for every env on the GPU.
//
// Replaces cassie/cassie.py:293-496,523-680,787-859 + cassie/rewards/clock_rewards.py:6-110 +
// cassie/phase_function.py:5-136 and the per-env ctypes FFI below them (cassie/cassiemujoco/cassiemujoco.py).
//
// Layout: one env per 16-lane DPP row (4 envs per wave64), the env's whole state resident in LDS for the launch.  Persistent
// state is SoA in HBM, field-major: st[field * n_envs + env].  Algorithmic HBM traffic per env step is the state read + write
// (2 * 590 words) + action/obs/reward I/O (246 B) = 4 965 B: the kernel is VALU-issue / LDS-latency bound.
#include "apx_common.h"
#include "cassie_model_gen.h"
#include "cassie_lane.h"
#include "cassie_complete.h"
#include "estimator_lane.h"

#include <new>
#include <cstring>

using namespace cmt;
using c4::V3; using c4::Q4; using c4::M3; using c4::qmul; using c4::q2m; using c4::col; using c4::cross; using c4::dot; using c4::mul;
constexpr int NQ = CM_NQ, NV = CM_NV, NB = CM_NBODY;
constexpr float GRAV = 9.81f;

#include "env_state.h"
#include "cassie_traj_gen.h"

// ------------------------------------------------------------------------------------------------ native substep model
__constant__ float kP[5] = {100.f, 100.f, 88.f, 96.f, 50.f};
__constant__ float kD[5] = {10.f, 10.f, 8.f, 9.6f, 5.f};
__constant__ float kOffset[10] = {0.0045f, 0.0f, 0.4973f, -1.1997f, -1.5968f, 0.0045f, 0.0f, 0.4973f, -1.1997f, -1.5968f};
__constant__ float kNeutralFoot[4] = {-0.24790886454547323f, -0.24679713195445646f, -0.6609396704367185f, 0.663921021343526f};
__constant__ float kTorqueLimit[5] = {140.63f, 140.63f, 216.16f, 216.16f, 45.14f};
constexpr float cFir[9] = {2727.f, 534.f, -2658.f, -795.f, 72.f, 110.f, 19.f, -6.f, -3.f};
#define PI_F 3.14159265358979323846f

__device__ __forceinline__ float* rows4() { return (float*)(apx_lds4 + ((threadIdx.x >> 4) * L4_ES + L4_ROWS) / 4); }

#ifndef APX_STAGE
#define APX_STAGE __noinline__
#endif
// ---- the substep in five stages (cassie_lane.h); the env's state and every stage hand-off live in the env's LDS region
// stage 1: encoders + estimator -> PD -> safeties -> motor model / delay: lanes 0..9 = the ten drives,
// lanes 10..15 = the six joint encoders, lane 0 = estimator.  mode 0: forward pass only with zero ctrl (cassie_sim_set_const ends
// in mj_forward)
// per-drive model constants as selects on the drive index inside a leg (u5 = 0 roll, 1 yaw, 2 pitch, 3 knee, 4 foot) instead of table
// loads: a lane-indexed table needs a 64-bit address per lane that the compiler keeps alive across all 50 substeps (2 VGPRs each)
// (one plain select per step: a nested ?: chain of constants is turned into divergent branches, i.e. exec-mask regions, by the compiler)
template <class T> __device__ __forceinline__ T sel5(int u5, T a0, T a1, T a2, T a3, T a4) {
    T r = a4;
    r = u5 == 3 ? a3 : r; r = u5 == 2 ? a2 : r; r = u5 == 1 ? a1 : r; r = u5 == 0 ? a0 : r;
    return r;
}
static_assert(cmt::ct_act_gear[5] == cmt::ct_act_gear[0] && cmt::ct_act_gear[9] == cmt::ct_act_gear[4] && cmt::ct_act_bits[7] == cmt::ct_act_bits[2] &&
              cmt::ct_act_rpm[8] == cmt::ct_act_rpm[3] && cmt::ct_act_ctrlmax[6] == cmt::ct_act_ctrlmax[1] && cmt::ct_act_dof[5] == cmt::ct_act_dof[0] + 13 &&
              cmt::ct_act_dof[9] == cmt::ct_act_dof[4] + 13, "the two legs carry the same drives");
template <bool CARRY>
__device__ __forceinline__ void stage1_io_lane(St S, int mode, est::Rec& carried, float* estrec) {   // (as a called function the reset kernel faults: kept inline)
    PROF_START();
    int l = threadIdx.x & 15;
    APX_PIN("+v"(l));      // opaque: otherwise every per-lane constant below is hoisted out of the 50-substep loop and has to be
                                     // kept (= spilled to scratch) across the constraint stage, which needs the whole register file
    if (mode == 0) { if (l < 10) S.W(c4::WK_CTRL + l) = 0.f; if (l == 0) S.W(c4::WK_MISC + 6) = 0.f; PROF(0); return; }
    est::Rec local;
    if constexpr (!CARRY) local = est::rec_load(estrec, S.env, l);      // in flight while the encoder model runs
    est::Rec& rec = CARRY ? carried : local;
    const int flags = S.I(I_FLAGS);
    const bool mot = l < 10;
    const int u = mot ? l : 0, u5 = u >= 5 ? u - 5 : u, k = mot ? 0 : l - 10;
    float sdepth = 0.f, ssign = 1.f, tau_cmd = 0.f, mvel = 0.f, mpos_l = 0.f;
    const float gear = sel5(u5, cmt::ct_act_gear[0], cmt::ct_act_gear[1], cmt::ct_act_gear[2], cmt::ct_act_gear[3], cmt::ct_act_gear[4]);
    if (mot) {
        // drive encoder: truncating quantiser + 9-tap FIR velocity
        const float scale = 2.f * PI_F / (float)(1 << sel5(u5, cmt::ct_act_bits[0], cmt::ct_act_bits[1], cmt::ct_act_bits[2], cmt::ct_act_bits[3], cmt::ct_act_bits[4]));
        const float nq = truncf(S(F_SNAP + SN_MPOS + u) * gear / scale);
        float h[9];
        if (!(flags & 1)) { _Pragma("unroll") for (int i = 0; i < 9; ++i) h[i] = nq; }
        else { _Pragma("unroll") for (int i = 8; i > 0; --i) h[i] = S(F_MENC + u * 9 + i - 1); h[0] = nq; }
        float acc = 0.f;
        _Pragma("unroll") for (int i = 0; i < 9; ++i) { S(F_MENC + u * 9 + i) = h[i]; acc += cFir[i] * h[i]; }
        const float mpos = nq * scale / gear;
        mpos_l = mpos;
        mvel = acc * scale / gear / PI_F;
        S(F_SO + SO_MPOS + u) = mpos; S(F_SO + SO_MVEL + u) = mvel;
        // pd_input_step: tau = P (pTarget - q) + D (0 - qd), no clamp (G9); pd_in_t is zero until the first env.step
        const float kp = sel5(u5, 100.f, 100.f, 88.f, 96.f, 50.f), kd = sel5(u5, 10.f, 10.f, 8.f, 9.6f, 5.f);      // cassie.py:57-58
        tau_cmd = (flags & 16) ? kp * (S(F_PDT + u) - mpos) + kd * (0.f - mvel) : 0.f;
        // cassie_core_sim_step (G10): soft joint-limit zones 0.15 rad inside the drive limits
        constexpr float DEG = PI_F / 180.f;
        const float lo_deg = sel5(u5, -15.f, -22.f, -50.f, -156.f, -140.f);
        const float hi_deg = sel5(u5, 20.f, 22.f, 80.f, -42.f, -35.f);
        float lo = lo_deg * DEG + 0.15f, hi = hi_deg * DEG - 0.15f;
        if (u >= 5 && u5 < 2) { const float t = lo; lo = -hi; hi = -t; }      // roll / yaw mirror on the right leg
        sdepth = fmaxf(0.f, fmaxf(mpos - hi, lo - mpos));
        ssign = mpos > hi ? -1.f : 1.f;
    }
    // coupled zone (golden G10b): hip pitch + knee below -135 deg is one more zone acting on both drives of the leg
    float cdepth = 0.f;
    {
        const float pk = mpos_l + c4::dpp<0x101>(mpos_l);                          // row_shl:1 brings lane u+1 (knee) to the pitch lane
        const float dcp = fmaxf(0.f, -0.75f * PI_F - pk);                          // valid on the pitch lanes u = 2, 7
        const float dck = c4::dpp<0x111>(dcp);                                     // row_shr:1: the knee lanes u = 3, 8 take their leg's depth
        cdepth = (mot && u5 == 2) ? dcp : (mot && u5 == 3) ? dck : 0.f;
    }
    // global torque scale = product over the drives (the coupled zone counts once per leg: on the pitch lane)
    float sscale = fmaxf(0.f, 1.f - sdepth * (1.f / 0.15f)) * ((mot && u5 == 2) ? fmaxf(0.f, 1.f - cdepth * (1.f / 0.15f)) : 1.f);
    sscale *= c4::dpp<0xB1>(sscale); sscale *= c4::dpp<0x4E>(sscale); sscale *= c4::dpp<0x141>(sscale); sscale *= c4::dpp<0x140>(sscale);
    if (mot) {
        const float sKp = sel5(u5, 1000.f, 800.f, 1200.f, 1200.f, 100.f), sKd = sel5(u5, 12.f, 12.f, 36.f, 36.f, 7.f);
        const float d = sdepth;
        float tau = sscale * tau_cmd + ssign * sKp * d * (1.f + d * (1.f / 0.15f)) - fminf(1.f, d * (1.f / 0.15f)) * sKd * mvel;
        tau += sKp * cdepth * (1.f + cdepth * (1.f / 0.15f)) - fminf(1.f, cdepth * (1.f / 0.15f)) * sKd * mvel;      // coupled zone (0 off the pitch / knee lanes)
        const float tql = sel5(u5, 140.63f, 140.63f, 216.16f, 216.16f, 45.14f);
        tau = fminf(fmaxf(tau, -tql), tql);
        // cassie_sim_step_ethercat: torque-speed curve, 6-deep delay line
        const float wmax = sel5(u5, cmt::ct_act_rpm[0], cmt::ct_act_rpm[1], cmt::ct_act_rpm[2], cmt::ct_act_rpm[3], cmt::ct_act_rpm[4]) * 2.f * PI_F / 60.f;
        const float tmax = sel5(u5, cmt::ct_act_ctrlmax[0], cmt::ct_act_ctrlmax[1], cmt::ct_act_ctrlmax[2], cmt::ct_act_ctrlmax[3], cmt::ct_act_ctrlmax[4]);
        const int adof = sel5(u5, cmt::ct_act_dof[0], cmt::ct_act_dof[1], cmt::ct_act_dof[2], cmt::ct_act_dof[3], cmt::ct_act_dof[4]) + (u >= 5 ? 13 : 0);
        const float om = fabsf(S(F_QVEL + adof) * gear);
        const float tlim = fminf(fmaxf(2.f * tmax * (1.f - om / wmax), 0.f), tmax);
        const float cmd = tau / gear;
        const float un = (cmd < 0.f ? -1.f : 1.f) * fminf(fabsf(cmd), tlim);
        float fifo[6];
        _Pragma("unroll") for (int i = 5; i > 0; --i) fifo[i] = S(F_FIFO + u * 6 + i - 1);
        fifo[0] = un;
        _Pragma("unroll") for (int i = 0; i < 6; ++i) S(F_FIFO + u * 6 + i) = fifo[i];
        S.W(c4::WK_CTRL + u) = fifo[5];
        S(F_SO + SO_TORQUE + u) = gear * fifo[5];
    } else {   // joint encoders: quantiser + biquad velocity
        static_assert(cmt::ct_jsens_bits[0] == 18 && cmt::ct_jsens_bits[1] == 18 && cmt::ct_jsens_bits[2] == 13 && cmt::ct_jsens_bits[3] == 18 && cmt::ct_jsens_bits[4] == 18 && cmt::ct_jsens_bits[5] == 13, "joint encoder bits");
        const float scale = 2.f * PI_F / (float)(1 << ((k == 2 || k == 5) ? 13 : 18));
        const float x = truncf(S(F_SNAP + SN_JPOS + k) / scale) * scale;
        float xs[4], y0, y1;
        if (!(flags & 2)) { xs[0] = xs[1] = xs[2] = xs[3] = x; y0 = y1 = 0.f; }
        else { xs[0] = x; _Pragma("unroll") for (int i = 1; i < 4; ++i) xs[i] = S(F_JENCX + k * 4 + i - 1); y0 = S(F_JENCY + k * 2); y1 = S(F_JENCY + k * 2 + 1); }
        const float y = 12.348f * (xs[0] + xs[1] - xs[2] - xs[3]) + 1.7658f * y0 - 0.79045f * y1;
        _Pragma("unroll") for (int i = 0; i < 4; ++i) S(F_JENCX + k * 4 + i) = xs[i];
        S(F_JENCY + k * 2) = y; S(F_JENCY + k * 2 + 1) = y0;
        S(F_SO + SO_JPOS + k) = x; S(F_SO + SO_JVEL + k) = y;
    }
    if (l == 0) {
        S.I(I_FLAGS) = flags | 3;
        // state estimator, pass-through fields (pelvis.orientation / rotationalVelocity = vectorNav orientation / gyro)
        _Pragma("unroll") for (int i = 0; i < 4; ++i) S(F_SO + SO_QUAT + i) = S(F_SNAP + SN_QUAT + i);
        _Pragma("unroll") for (int i = 0; i < 3; ++i) S(F_SO + SO_ROTVEL + i) = S(F_SNAP + SN_GYRO + i);
    }
    c4::wsync();                                          // the encoder lanes' outputs are the estimator's inputs
    PROF2(37);
    est::est_step_lane(S, rec);                           // the 7 filtered fields: translationalVelocity, translationalAcceleration, height
    if constexpr (!CARRY) est::rec_store(estrec, S.env, l, rec);      // (CARRY: the record stays in registers, loaded / stored once per env step by the caller)
    PROF2(38);
    PROF(0);
}
__device__ APX_STAGE void stage1b_tree_lane(St S) {
    PROF_START();
    c4::stage_tree_lane<false>(S, rows4());
    PROF(1);
}
// the factor stage is inlined: as a function it needs two callee-saved VGPRs more than it can park in AGPRs (8 B of scratch per call, the
// only scratch user of the step kernel); inlined, every kernel of this file is scratch-free at the same speed
#ifndef APX_STAGE_FACTOR
#define APX_STAGE_FACTOR __forceinline__
#endif
__device__ APX_STAGE_FACTOR void stage2a_factor(St S, c4::FacRegs& FR, c4::FacTail& F) {
    PROF_START();
    c4::stage_factor_lane<false>(S, FR, F);
    PROF(2);
}
template <bool HF>
__device__ __forceinline__ void stage3_rows_pgs_lane(St S, c4::FacRegs& FR, const Cfg& cfg) {
    PROF_START();
    c4::stage_rows_pgs_lane<HF>(S, FR, rows4(), cfg.pgs_iters, cfg.hf);
    PROF(3);
}
// inlined like the rows / PGS stage: as a function it needs 36 callee-saved VGPRs, i.e. 36 scratch stores + 36 loads per lane per
// substep (9 KB per wave-substep), several times the algorithmic HBM traffic of the whole kernel
__device__ __forceinline__ void stage4_finish(St S, int mode, const c4::FacTail& F, const c4::FacRegs& FR) {
    PROF_START();
    c4::stage_finish_lane(S, rows4(), mode != 0, F, FR);
    PROF(4);
}
// mj_setConst (sim.set_const after dynamics randomisation), lane-parallel: tree at qpos0 -> factor -> |y~|^2 of unit rows
__device__ __forceinline__ void setconst_lane(const St& S) {
    c4::stage_tree_lane<true>(S, rows4());
    c4::wsync();
    c4::stage_factor_lane(S);
    c4::wsync();
    c4::setconst_rows_lane<0>(S);
    c4::setconst_rows_lane<1>(S);
    c4::wsync();
}
// one 2 kHz substep (cassie_sim_step_pd): the wave holds 4 envs, one per 16-lane row; every call site is reached by all lanes
// `rec` = this lane's share of the env's state-estimator record (estimator_lane.h), carried in registers over the substeps of an env step (round 5: it went through L2 twice per
// substep; in the one-launch rollout the hot lines were written back to HBM ~190 times per launch - 515 MB against 68 MB of algorithmic traffic)
template <bool HF, bool CARRY = true>
__device__ __forceinline__ void sim_step_pd(const St& S, const Cfg& cfg, int mode, est::Rec& rec) {
#ifdef APX_PROF
    const unsigned long long t0__ = clock64();
#endif
    stage1_io_lane<CARRY>(S, mode, rec, cfg.est);
    c4::wsync();
    stage1b_tree_lane(S);
    c4::wsync();
    c4::FacRegs FR;                    // the share of the factor that the row stage reads, in registers across the stage boundary
    c4::FacTail F;                     // ... and the rest of the factor for the finish stage (parked in AGPRs across the sweeps by the register allocator)
    stage2a_factor(S, FR, F);
    c4::wsync();
    stage3_rows_pgs_lane<HF>(S, FR, cfg);
    c4::wsync();
    // a pass that needs more rows than the lane map carries (third capsule end of a leg on the floor, second limit of a leg, hip-pitch capsule / pelvis sphere on the floor,
    // more than three capsule pairs) leaves the fast path here: its env skips the finish stage and gets its COMPLETE row set, solve and finish out of line (cassie_complete.h)
#ifdef APX_NO_COMPLETE      /* A/B build: the capped fast path alone (rounds 1-4) */
    stage4_finish(S, mode, F, FR);
    c4::wsync();
#else
    const bool satp = cfg.complete_pool != nullptr && S.W(c4::WK_MISC + 7) != 0.f;
    if (!satp) stage4_finish(S, mode, F, FR);
    APX_CONVERGE();
    c4::wsync();
    if (__builtin_amdgcn_ballot_w64(satp) != 0ull) {
        if (satp) c4::substep_complete<HF>(S, rows4(), cfg.pgs_iters, cfg.hf, cfg.complete_pool, mode);
        APX_CONVERGE();
        c4::wsync();
    }
#endif
#ifdef APX_PROF
    if (threadIdx.x == 0 && blockIdx.x == 0) c4::g_prof_acc[8] += clock64() - t0__;
#endif
}

// the same with the record loaded from / stored to its HBM home inside the io stage of the substep (per-step launch kernel, reset paths; mode 0 does not touch it)
template <bool HF>
__device__ __forceinline__ void sim_step_pd(const St& S, const Cfg& cfg, int mode) {
    est::Rec none;      // (not touched: the io stage loads and stores the record itself)
    sim_step_pd<HF, false>(S, cfg, mode, none);
}

// ------------------------------------------------------------------------------------------------ env logic
struct ClockK { float x[8]; float phaselen; };
__device__ __forceinline__ void clock_knots(float swing, float stance, int freq, ClockK& c) {
    const float total = 2.f * swing + 2.f * stance;
    c.phaselen = total * freq;
    const float seg[5] = {0.f, swing, swing + stance, 2.f * swing + stance, total};
    for (int s = 0; s < 4; ++s) {
        const float a = seg[s] * freq, b = seg[s + 1] * freq, off = (b - a) * 0.1f;
        c.x[2 * s] = a + off; c.x[2 * s + 1] = b - off;
    }
}
// value of clock `which` (0 left_frc, 1 left_vel, 2 right_frc, 3 right_vel) on segment s (0 right swing, 1 dbl, 2 left swing, 3 dbl)
__device__ __forceinline__ float clock_val(int which, int s, int mode, int inc) {
    // reference arrays r_frc, r_vel, l_frc, l_vel (phase_function.py:16-96); "left" clock = the r_* pair (cassie.py:559)
    float r_frc, r_vel, l_frc, l_vel;
    const float pos = inc ? 1.f : 0.f;
    if (s == 0) { l_vel = r_frc = -1.f; l_frc = r_vel = pos; }
    else if (s == 2) { l_vel = r_frc = pos; l_frc = r_vel = -1.f; }
    else if (mode == 2) { l_frc = r_frc = -1.f; l_vel = r_vel = pos; }
    else if (mode == 0) { l_frc = r_frc = l_vel = r_vel = 0.f; }
    else if (inc) { l_frc = r_frc = 1.f; l_vel = r_vel = -1.f; }
    else if (s == 1) { r_frc = 0.f; l_frc = -1.f; r_vel = -1.f; l_vel = 0.f; }     // phase_function.py:54-55 quirk
    else { l_frc = r_frc = 0.f; l_vel = r_vel = -1.f; }
    return which == 0 ? r_frc : which == 1 ? r_vel : which == 2 ? l_frc : l_vel;
}
__device__ float clock_eval(const ClockK& c, int which, float ph, int mode, int inc) {
    if (ph > c.x[0] + c.phaselen) ph -= c.phaselen;
    // 10 knots: last of previous cycle, 8 of this cycle, first of the next
    float xa = c.x[7] - c.phaselen, ya = clock_val(which, 3, mode, inc);
    for (int i = 0; i < 9; ++i) {
        const float xb = i < 8 ? c.x[i] : c.x[0] + c.phaselen;
        const float yb = clock_val(which, i < 8 ? i / 2 : 0, mode, inc);
        if (ph <= xb) {
            if (ph < xa) return ya;
            const float t = (ph - xa) / (xb - xa);
            return ya + (yb - ya) * (3.f * t * t - 2.f * t * t * t);
        }
        xa = xb; ya = yb;
    }
    return ya;
}

__device__ __forceinline__ void yaw_inv_rotate3(float yaw, const float* v, float* out) {
    float sz, cz;
    sincosf(0.5f * yaw, &sz, &cz);
    Q4 q = {cz, 0.f, 0.f, sz};
    if (q.w < 0.f) q = {-q.w, 0.f, 0.f, -q.z};
    const Q4 iq = {q.w, 0.f, 0.f, -q.z};
    const Q4 r = qmul(iq, qmul(Q4{0.f, v[0], v[1], v[2]}, q));
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}

// self.phase of the reference as a float: I_PHASE + 0.5 x the half bit of I_FLAGS (bit 5); bit 6 = self.phase_add is 1.5 (tools/test_commands.py:86)
__device__ __forceinline__ float fphase(const St& S) { return (float)S.I(I_PHASE) + 0.5f * (float)((S.I(I_FLAGS) >> 5) & 1); }
// self.phase += self.phase_add; wrap (cassie.py:447-453, 511-515)
__device__ __forceinline__ void advance_phase(const St& S) {
    int flags = S.I(I_FLAGS), phase = S.I(I_PHASE);
    const int half = (flags >> 5) & 1;
    if (flags & 64) { phase += 1 + half; flags ^= 32; } else phase += 1;
    if ((float)phase + 0.5f * (float)((flags >> 5) & 1) > S(F_CMD + 5)) { phase = 0; flags &= ~32; S.I(I_COUNTER) += 1; }
    S.I(I_PHASE) = phase; S.I(I_FLAGS) = flags;
}
// get_full_state (cassie/cassie.py:787-859), written straight to obs[env*50 ..]
__device__ void write_obs(const St& S, const Cfg& cfg, float* o) {
    const float yaw = S(F_CMD + 2);
    Q4 nq;
    {
        float sz, cz;
        sincosf(0.5f * yaw, &sz, &cz);
        Q4 q = {cz, 0.f, 0.f, sz};
        if (q.w < 0.f) q = {-q.w, 0.f, 0.f, -q.z};
        nq = qmul(Q4{q.w, 0.f, 0.f, -q.z}, Q4{S(F_SO + SO_QUAT), S(F_SO + SO_QUAT + 1), S(F_SO + SO_QUAT + 2), S(F_SO + SO_QUAT + 3)});
        if (nq.w < 0.f) nq = {-nq.w, -nq.x, -nq.y, -nq.z};
    }
    int n;
    if (cfg.input_profile == 1) {      // input_profile "min" (cassie.py:829-837): the estimator's foot positions / orientations from its record (estimator_lane.h, lane 6)
        const float* r = cfg.est + (size_t)S.env * est::REC + 24 * 6 + 4;
        for (int k = 0; k < 3; ++k) { o[k] = r[k]; o[3 + k] = r[4 + k]; o[10 + k] = S(F_SO + SO_ROTVEL + k); }
        o[6] = nq.w; o[7] = nq.x; o[8] = nq.y; o[9] = nq.z;
        for (int k = 0; k < 4; ++k) { o[13 + k] = r[8 + k]; o[17 + k] = r[12 + k]; }
        n = APX_OBS_MIN;
    } else {
        o[0] = S(F_SO + SO_HEIGHT);
        o[1] = nq.w; o[2] = nq.x; o[3] = nq.y; o[4] = nq.z;
        for (int u = 0; u < 10; ++u) o[5 + u] = S(F_SO + SO_MPOS + u) + S(F_MNOISE + u);
        float v[3];
        for (int k = 0; k < 3; ++k) v[k] = S(F_SO + SO_TVEL + k);
        yaw_inv_rotate3(yaw, v, o + 15);
        for (int k = 0; k < 3; ++k) o[18 + k] = S(F_SO + SO_ROTVEL + k);
        for (int u = 0; u < 10; ++u) o[21 + u] = S(F_SO + SO_MVEL + u);
        for (int k = 0; k < 3; ++k) v[k] = S(F_SO + SO_TACC + k);
        yaw_inv_rotate3(yaw, v, o + 31);
        for (int k = 0; k < 6; ++k) o[34 + k] = S(F_SO + SO_JPOS + k) + S(F_JNOISE + k);
        for (int k = 0; k < 6; ++k) o[40 + k] = S(F_SO + SO_JVEL + k);
        n = 46;
    }
    const float ang = 2.f * PI_F * fphase(S) / S(F_CMD + 5);
    o[n] = sinf(ang); o[n + 1] = cosf(ang);
    if (cfg.command_profile == 0) { o[n + 2] = S(F_CMD + 0); o[n + 3] = S(F_CMD + 1); return; }
    // command_profile "phase" (cassie.py:805-808): clock, swing / stance duration, one-hot stance mode (grounded, aerial, zero), speed, side speed
    const int sm = (int)S(F_CMD + 6);
    o[n + 2] = S(F_CMD + 3); o[n + 3] = S(F_CMD + 4);
    o[n + 4] = sm == 1 ? 1.f : 0.f; o[n + 5] = sm == 2 ? 1.f : 0.f; o[n + 6] = sm == 0 ? 1.f : 0.f;
    o[n + 7] = S(F_CMD + 0); o[n + 8] = S(F_CMD + 1);
}

__device__ __forceinline__ void clock_from_speed(const St& S, float speed, int freq) {   // cassie.py:556-559
    const float total = (0.9f - 0.25f / 3.0f * fabsf(speed)) * 0.5f;
    const float swing = (0.30f + ((0.70f - 0.30f) / 3.f) * fabsf(speed)) * total;
    const float stance = (0.70f - ((0.70f - 0.30f) / 3.f) * fabsf(speed)) * total;
    S(F_CMD + 3) = swing; S(F_CMD + 4) = stance; S(F_CMD + 5) = (2.f * swing + 2.f * stance) * freq;
}

// CassieTrajEnv.get_ref_state (cassie/cassie_traj.py:926-972), walking trajectory at simrate 50, written into qpos / qvel
__device__ void traj_pose(const St& S, float phase, float phaselen, float speed, int counter) {
    if (phase > phaselen) phase = 0.f;
    if (phase > floorf((float)TRAJ_LEN / 50.f) - 1.f) phase = floorf((phase / phaselen) * (float)TRAJ_LEN / 50.f);
    const int row = (int)phase;
    for (int i = 0; i < NQ; ++i) S(F_QPOS + i) = traj_table[row][i];
    for (int i = 0; i < NV; ++i) S(F_QVEL + i) = traj_table[row][35 + i];
    S(F_QPOS) = S(F_QPOS) * speed + TRAJ_DX * (float)counter * speed;
    S(F_QPOS + 1) = 0.f;
    S(F_QVEL) *= speed;
}

// CassieEnv.reset (cassie/cassie.py:523-680)
// reset, part 1 (wave 0): command / clock / dynamics-randomisation draws (cassie.py:525-657) and the init pose
__device__ void env_reset_draws(const St& S, const Cfg& cfg, int episode) {
    Rng r{cfg.seed_lo, cfg.seed_hi, cfg.env_base + (unsigned)S.env, (unsigned)episode * RNG_RESET_BLOCK, RNG_RESET};
    const float speed0 = cfg.env_kind == 1 ? (float)r.randint(41u) / 10.f : r.uniform(-0.3f, 4.0f);      // cassie_traj.py:608: random.randint(0, 40) / 10
    (void)r.uniform(-0.3f, 0.3f);
    S(F_CMD) = speed0;                                   // kept for the trajectory-pose reset; replaced by the command redraw after the settle step
    if (cfg.command_profile == 0) { clock_from_speed(S, speed0, 2000 / cfg.simrate); S(F_CMD + 6) = (float)cfg.stance_mode; }
    else {      // command_profile "phase" (cassie.py:529-545): swing / stance duration and stance mode drawn per episode
        // durations and cycle length in fp64 from the INTEGER draws, like the reference's Python floats: (2 x 0.07 + 2 x 0.13) x 40 is exactly 16 there and 15.999999 in
        // fp32, and `phase > phaselen` (cassie.py:451) would wrap one step early
        double swing, stance;
        if (cfg.command_profile == 2) {                  // "library" reward variant (:531-539)
            S(F_CMD) = (float)r.randint(31u) / 10.f;
            const double total = (double)(3u + r.randint(4u)) / 10.0, ratio = (double)(2u + r.randint(7u)) / 10.0;
            swing = total * ratio; stance = total - swing;
        } else { swing = (double)(1u + r.randint(50u)) / 100.0; stance = (double)(1u + r.randint(30u)) / 100.0; }
        const unsigned pick = r.randint(3u);             // np.random.choice(["grounded", "aerial", "zero"])
        S(F_CMD + 3) = (float)swing; S(F_CMD + 4) = (float)stance; S(F_CMD + 5) = (float)((2.0 * swing + 2.0 * stance) * (double)(2000 / cfg.simrate));
        S(F_CMD + 6) = pick == 0u ? 1.f : pick == 1u ? 2.f : 0.f;
    }
    S.I(I_PHASE) = (int)r.randint((unsigned)floorf(S(F_CMD + 5)) + 1u); S.I(I_FLAGS) &= ~32;
    S.I(I_TIME) = 0; S.I(I_COUNTER) = 0;
    if (cfg.dyn_rand) {
        for (int d = 0; d < 6; ++d) { (void)r.u01(); S(F_DAMP + d) = cm_dof_damping[d]; }
        for (int leg = 0; leg < 2; ++leg)
            for (int k = 0; k < 13; ++k) {
                const int d = 6 + 13 * leg + k;
                const float u = r.u01();
                const bool vary = !(k == 9 || k == 11);                  // heel-spring, plantar-rod keep their damping
                const float lo = vary ? 0.3f : 1.f, hi = vary ? 5.f : 1.f;
                S(F_DAMP + d) = fmaxf(0.f, cm_dof_damping[d] * (lo + (hi - lo) * u));
            }
        (void)r.u01(); S(F_MASS) = 0.f;
        for (int b = 1; b < NB; ++b) S(F_MASS + b) = fmaxf(0.f, cm_body_mass[b] * r.uniform(0.5f, 1.5f));
        S(F_FRIC) = r.uniform(0.4f, 1.1f); (void)r.u01(); (void)r.u01();
        const float roll = r.uniform(-0.03f, 0.03f), pitch = r.uniform(-0.03f, 0.03f);
        float sy, cy, sx, cx;
        sincosf(0.5f * pitch, &sy, &cy); sincosf(0.5f * roll, &sx, &cx);
        Q4 fq = {cx * cy, cy * sx, cx * sy, sx * sy};
        if (fq.w < 0.f) fq = {-fq.w, -fq.x, -fq.y, -fq.z};
        const M3 Rf = q2m(fq);
        const V3 nrm = col(Rf, 2);
        V3 t1 = fabsf(nrm.y) < 0.5f ? V3{0.f, 1.f, 0.f} : V3{0.f, 0.f, 1.f};
        t1 = t1 - nrm * dot(nrm, t1); t1 = t1 * rsqrtf(dot(t1, t1));
        const V3 t2 = cross(nrm, t1);
        const float fl[9] = {nrm.x, nrm.y, nrm.z, t1.x, t1.y, t1.z, t2.x, t2.y, t2.z};
        for (int k = 0; k < 9; ++k) S(F_FLOOR + k) = fl[k];
        for (int u = 0; u < 10; ++u) S(F_MNOISE + u) = r.uniform(-0.01f, 0.01f);
        for (int k = 0; k < 6; ++k) S(F_JNOISE + k) = r.uniform(-0.01f, 0.01f);
    }
    for (int i = 0; i < NQ; ++i) S(F_QPOS + i) = cm_init_qpos[i];
    for (int i = 0; i < NV; ++i) { S(F_QVEL + i) = 0.f; S(F_QACCW + i) = 0.f; }
}
// reset, part 3 (wave 0): command redraw after the settle step (cassie.py:667-670)
__device__ void env_reset_finish(const St& S, const Cfg& cfg) {
    Rng r{cfg.seed_lo, cfg.seed_hi, cfg.env_base + (unsigned)S.env, (unsigned)S.I(I_EPISODE) * RNG_RESET_BLOCK + RNG_RESET_TAIL, RNG_RESET};      // the redraws close the episode's block
    for (int k = 0; k < 6; ++k) S(F_FOOTPREV + k) = S(F_FWD + 10 + k);
    S(F_CMD + 2) = 0.f;
    S(F_CMD + 0) = r.uniform(-0.3f, 4.0f);
    S(F_CMD + 1) = r.uniform(-0.3f, 0.3f);
}
__device__ int g_reset_miss = 0;
// CassieEnv.reset (cassie/cassie.py:523-680) up to its settle step; called by all 16 lanes of the env's row
__device__ __forceinline__ void env_restart_head(const St& S, const Cfg& cfg, int n) {
    const bool lead = (threadIdx.x & 15) == 0;
    if (cfg.est_lifetime > 0 && S.I(I_AGE) >= cfg.est_lifetime) {      // this env instance has served a PPO.sample call's worth of steps: the next one starts with a new estimator
        const int l = threadIdx.x & 15;
        if (l < 7) { est::Rec z; c4::sfor<0, 6>([&](auto K) { z.v[K] = est::f4{0.f, 0.f, 0.f, 0.f}; }); est::rec_store(cfg.est, S.env, l, z); }
        c4::wsync();
        if (lead) S.I(I_AGE) = 0;
    }
    // Everything of the reset up to its settle step is a function of (seed, env, episode index) alone and sits in the env's ring (env_reset_kernel, part 0: ahead of time
    // while the learner runs, or on demand right before this).  Here: copy the image; the caller runs the settle step.
    const int ep = S.I(I_EPISODE) + 1, slot = ep % RST_K;
    {
        const int l = threadIdx.x & 15;
        if (lead && cfg.rst_int[(size_t)(2 * slot) * n + S.env] != ep) atomicAdd(&g_reset_miss, 1);      // (cannot happen: the first part of env_reset_kernel fills the slot; reported by apx_env_get_field("reset_miss"))
        const float* img = cfg.rst + (size_t)slot * F_TOTAL * n + S.env;
        {   // qpos, qvel, qacc_warmstart, mass, damping, friction, floor, invweights, encoder offsets: every load of the image in flight before the first LDS store (as a plain
            // copy loop the compiler waited for each HBM load in turn: ~30 dependent round trips, a third of the kernel)
            constexpr int NCP = (F_PDT + 15) / 16;
            float v[NCP];
#pragma unroll
            for (int k = 0; k < NCP; ++k) { const int f = l + 16 * k; v[k] = img[(size_t)(f < F_PDT ? f : 0) * n]; }
#pragma unroll
            for (int k = 0; k < NCP; ++k) { const int f = l + 16 * k; if (f < F_PDT) S(f) = v[k]; }
        }
        for (int f = F_SNAP + l; f < F_SNAP + 26; f += 16) S(f) = img[(size_t)f * n];        // sensor snapshot of the forward pass at the init pose
        if (l < 7) S(F_CMD + l) = img[(size_t)(F_CMD + l) * n];
        S(F_FWD + l) = img[(size_t)(F_FWD + l) * n];
        if (lead) { S.I(I_PHASE) = cfg.rst_int[(size_t)(2 * slot + 1) * n + S.env]; S.I(I_FLAGS) &= ~32; S.I(I_TIME) = 0; S.I(I_COUNTER) = 0; }
    }
    APX_LOCKSTEP();                                   // every lane has read I_EPISODE (its ring slot) before the lead moves it on
    if (lead) S.I(I_EPISODE) = ep;
    c4::wsync();
    if (cfg.env_kind == 1) {                          // CassieTrajEnv.reset: set_qpos / set_qvel with the reference state of the start phase
        if (lead) traj_pose(S, (float)S.I(I_PHASE), S(F_CMD + 5), S(F_CMD), 0);      // (cassie_traj.py:752-758); no mj_forward follows, so the
        c4::wsync();                                  // settle step below still reads the init-pose sensor snapshot, like the reference
    }
    // the caller runs the settle step (cassie.py:665, stale pd_in_t: F_PDT is not part of the image) and env_reset_finish
}

// clock_reward (cassie/rewards/clock_rewards.py:6-110)
__device__ float clock_reward(const St& S, const Cfg& cfg, const float* action, float lfrc, float rfrc, float lor, float ror) {
    const float fmax = 250.f, vmax = 2.0f;
    const float nlf = fminf(lfrc, fmax) / fmax, nrf = fminf(rfrc, fmax) / fmax;
    float lv = 0.f, rv = 0.f;
    for (int k = 0; k < 3; ++k) { lv += S(F_FOOTVEL + k) * S(F_FOOTVEL + k); rv += S(F_FOOTVEL + 3 + k) * S(F_FOOTVEL + 3 + k); }
    const float nlv = fminf(sqrtf(lv), vmax) / vmax, nrv = fminf(sqrtf(rv), vmax) / vmax;
    const float qw = S(F_QPOS + 3), speed = S(F_CMD);
    const float com_orient = 10.f * (1.f - qw * qw), foot_orient = 10.f * (lor + ror);
    const float com_vel_err = fabsf(S(F_QVEL) - speed);
    float straight = fabsf(S(F_QPOS + 1));
    if (straight < 0.05f) straight = 0.f;
    float hdiff = fabsf(S(F_QPOS + 2) - 0.9f);
    if (hdiff < 0.05f + 0.05f * speed) hdiff = 0.f;
    float pacc = 0.f;
    for (int k = 0; k < 3; ++k) pacc += fabsf(S(F_SO + SO_ROTVEL + k)) + fabsf(S(F_SO + SO_TACC + k));
    const float pelvis_motion = straight + hdiff + 0.25f * pacc;
    ClockK ck;
    clock_knots(S(F_CMD + 3), S(F_CMD + 4), 2000 / cfg.simrate, ck);
    const float ph = fphase(S);
    const int smode = (int)S(F_CMD + 6);                  // per env: the phase command profile draws it at every reset
    const float lfc = clock_eval(ck, 0, ph, smode, cfg.incentive), lvc = clock_eval(ck, 1, ph, smode, cfg.incentive);
    const float rfc = clock_eval(ck, 2, ph, smode, cfg.incentive), rvc = clock_eval(ck, 3, ph, smode, cfg.incentive);
    if (cfg.reward_kind == 2) {   // max_vel_clock_reward (clock_rewards.py:416-480): caps 400 N / 3 m/s, tanh terms, forward-velocity bonus
        const float mlf = fminf(lfrc, 400.f) / 400.f, mrf = fminf(rfrc, 400.f) / 400.f;
        const float mlv = fminf(sqrtf(lv), 3.f) / 3.f, mrv = fminf(sqrtf(rv), 3.f) / 3.f;
        float hd = fabsf(S(F_QPOS + 2) - 1.0f);
        if (hd < 0.2f) hd = 0.f;
        return 0.1f * expf(-15.f * (1.f - qw * qw)) + 0.1f * expf(-foot_orient) + 0.1f * expf(-(straight + hd)) +
               0.2f * (tanhf(lfc * mlf) + tanhf(rfc * mrf)) + 0.2f * (tanhf(lvc * mlv) + tanhf(rvc * mrv)) + 0.3f * (S(F_QVEL) / 3.0f);
    }
    if (cfg.reward_kind == 1) {   // early_clock_reward (clock_rewards.py:119-223): caps 350 N / 3 m/s, tanh scores, 5 terms
        const float elf = fminf(lfrc, 350.f) / 350.f, erf = fminf(rfrc, 350.f) / 350.f;
        const float elv = fminf(sqrtf(lv), 3.f) / 3.f, erv = fminf(sqrtf(rv), 3.f) / 3.f;
        return 0.250f * (tanhf(lfc * elf) + tanhf(rfc * erf)) + 0.350f * (tanhf(lvc * elv) + tanhf(rvc * erv)) +
               0.200f * expf(-com_vel_err) + 0.100f * expf(-((1.f - qw * qw) + (lor + ror))) + 0.100f * expf(-(straight + hdiff));
    }
    const float frc_score = tanf(PI_F / 4.f * lfc * nlf) + tanf(PI_F / 4.f * rfc * nrf);
    const float vel_score = tanf(PI_F / 4.f * lvc * nlv) + tanf(PI_F / 4.f * rvc * nrv);
    const float hip_roll = fabsf(S(F_QVEL + 6)) + fabsf(S(F_QVEL + 13));          // clock_rewards.py:74 (sic)
    float tq = 0.f, ac = 0.f;
    for (int u = 0; u < 10; ++u) { tq += fabsf(S(F_PREVTQ + u) - S(F_SO + SO_TORQUE + u)); ac += fabsf(S(F_PREVACT + u) - action[u]); }
    return 0.200f * frc_score + 0.200f * vel_score + 0.200f * expf(-(com_orient + foot_orient)) +
           0.150f * expf(-pelvis_motion) + 0.150f * expf(-com_vel_err) + 0.050f * expf(-hip_roll) +
           0.025f * expf(-0.25f * tq / 10.f) + 0.025f * expf(-5.f * ac / 10.f);
}

// ------------------------------------------------------------------------------------------------ kernels
#ifndef APX_WAVES_PER_EU
#define APX_WAVES_PER_EU 1
#endif
// one env per 16-lane row: the env's whole state is staged HBM -> LDS at the top of a launch and back at the end
#define ENV_SETUP                                                                                   \
    const int l = threadIdx.x & 15, row = threadIdx.x >> 4;                                          \
    /* XCD-aware: workgroup i runs on XCD i % 8 (own L2 each); give every XCD a CONTIGUOUS eighth of the envs so that the  \
       8 workgroups sharing a 128-B line of a state row (4 envs x 4 B each) hit the same L2 instead of fetching it 8 times */ \
    const int blk = (gridDim.x & 7) == 0 ? (int)(blockIdx.x & 7) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x; \
    const int env = blk * L4_EPW + row;                                                              \
    if (row >= L4_EPW || env >= n) return;                                                                            \
    const bool lead = l == 0;                                                                        \
    const St S{(lfloat*)apx_lds4 + row * L4_ES, env};                                              \
    c4::ct_fill();
__device__ __forceinline__ void load_state(const St& S, const float* st, const int* ist, int n) {
    const int l = threadIdx.x & 15;
    constexpr int NIT = (F_TOTAL + 15) / 16;
    float v[NIT];
#pragma unroll
    for (int i = 0; i < NIT; ++i) { const int f = 16 * i + l; v[i] = f < F_TOTAL ? st[(size_t)f * n + S.env] : 0.f; }      // all loads in flight
#pragma unroll
    for (int i = 0; i < NIT; ++i) { const int f = 16 * i + l; if (f < F_TOTAL) S(f) = v[i]; }
    if (l < I_TOTAL) S.I(l) = ist[(size_t)l * n + S.env];
    c4::wsync();
}
__device__ __forceinline__ void store_state(const St& S, float* st, int* ist, int n) {
    c4::wsync();
    const int l = threadIdx.x & 15;
    for (int f = l; f < F_TOTAL; f += 16) st[(size_t)f * n + S.env] = S(f);
    if (l < I_TOTAL) ist[(size_t)l * n + S.env] = S.I(l);
}

__global__ __launch_bounds__(64) void env_init_kernel(float* st, int* ist, int n, Cfg cfg) {
    const int env = blockIdx.x * 64 + threadIdx.x;
    if (env >= n) return;
    auto G = [&](int f) -> float& { return st[(size_t)f * n + env]; };
    for (int f = 0; f < F_TOTAL; ++f) G(f) = 0.f;
    for (int f = 0; f < I_TOTAL; ++f) ist[(size_t)f * n + env] = 0;
    for (int i = 0; i < NQ; ++i) G(F_QPOS + i) = cm_init_qpos[i];
    for (int b = 0; b < NB; ++b) G(F_MASS + b) = cm_body_mass[b];
    for (int d = 0; d < NV; ++d) G(F_DAMP + d) = cm_dof_damping[d];
    G(F_FRIC) = 1.f; G(F_CMD + 6) = (float)cfg.stance_mode;
    const float fl[9] = {0, 0, 1, 0, 1, 0, -1, 0, 0};     // n = z, t1 = y, t2 = n x t1 = -x
    for (int k = 0; k < 9; ++k) G(F_FLOOR + k) = fl[k];
}

__global__ __launch_bounds__(64, APX_WAVES_PER_EU) void env_setconst_kernel(float* st, int* ist, float* wk, int n, Cfg cfg) {
    ENV_SETUP
    load_state(S, st, ist, n);
    setconst_lane(S);
    store_state(S, st, ist, n);
}


// CassieEnv.reset in two parts of ONE kernel.  Part 0, the IMAGE: the part of episode ep's reset that does not depend on how the episode before it ends -
// draws from the episode-keyed stream, mj_setConst on the randomised model, the forward pass at the init pose - written to the env's ring slot (ep % RST_K); the working copy
// starts from the env's current state and nothing of the env itself is modified.  Part 1, the RESTART: copy the image's fields into the env, settle step, tail draws, observation.
//   ahead > 0 (apx_env_prepare_resets, while the learner runs): part 0 only, for episode I_EPISODE + ahead of the envs of `mask` whose slot does not hold it yet;
//   ahead = 0 (apx_env_reset, the auto-reset of apx_env_step with mask = the done flags): part 0 for the envs of `mask` whose slot does not hold episode I_EPISODE + 1 (normally
//   none: one launch per env step instead of two), then part 1 for all envs of `mask`.
// One kernel: an image is always computed by the same machine code (the forward-pass instance of the substep), a settle step by the other instance, so a restart is the
// same bits whether its image was prepared ahead or computed on demand (two KERNELS with their own inlined copies differed by an ulp under -ffast-math).
template <bool HF>
__global__ __launch_bounds__(64, APX_WAVES_PER_EU) void env_reset_kernel(float* st, int* ist, float* wk, int n, Cfg cfg, float* rst, int* rst_int, int ahead, const uint8_t* mask, float* obs) {
    ENV_SETUP
    const bool in_mask = !mask || mask[env] != 0;
    const int ep = ist[(size_t)I_EPISODE * n + env] + (ahead > 0 ? ahead : 1), slot = ep % RST_K;
    const bool need = in_mask && rst_int[(size_t)(2 * slot) * n + env] != ep, restart = ahead == 0 && in_mask;
    if (__builtin_amdgcn_ballot_w64(need || restart) == 0ull) return;      // wave-uniform: nothing to do for the four envs
    load_state(S, st, ist, n);
    if (__builtin_amdgcn_ballot_w64(need) != 0ull) {
        if (need) {
            if (lead) env_reset_draws(S, cfg, ep);
            c4::wsync();
            if (cfg.dyn_rand) setconst_lane(S);
            sim_step_pd<HF>(S, cfg, 0);                    // forward pass only
            c4::wsync();
            float* img = rst + (size_t)slot * F_TOTAL * n + env;
            for (int f = l; f < F_TOTAL; f += 16) img[(size_t)f * n] = S(f);
            if (lead) { rst_int[(size_t)(2 * slot + 1) * n + env] = S.I(I_PHASE); rst_int[(size_t)(2 * slot) * n + env] = ep; }
            if (restart) { __threadfence(); APX_LOCKSTEP(); load_state(S, st, ist, n); }      // back to the env's own state: the restart copies the image's fields over it (the lead's read of I_PHASE above comes first)
        }
        APX_CONVERGE();
    }
    if (__builtin_amdgcn_ballot_w64(restart) != 0ull) {
        if (restart) {
            env_restart_head(S, cfg, n);
            sim_step_pd<HF>(S, cfg, 1);                    // cassie.py:665 (stale pd_in_t)
            if (lead) env_reset_finish(S, cfg);
            c4::wsync();
            if (obs && lead) write_obs(S, cfg, obs + (size_t)env * cfg.obs_dim);
        }
        APX_CONVERGE();
    }
    if (restart) store_state(S, st, ist, n);
}

#ifdef APX_WAVETIME      /* experiment build: shader-clock duration of every wave of the step kernel (the launch lasts as long as its slowest wave) */
__device__ unsigned long long g_wavetime[4096];
#endif
// CassieEnv.step (cassie/cassie.py:389-496) of one env on its 16-lane row, state resident in LDS: PD targets from the action, simrate substeps with the per-substep
// accumulators, phase / time, termination, reward, command resampling.  act10 = the env's ten action entries (HBM row of the action batch for env_step_kernel, LDS words
// for env_rollout_kernel); returns reward and done flag on the lead lane.  Outputs (observation, reward, done) are the caller's.
// CARRY: the estimator record stays in registers over the substeps of the step (the one-launch rollout: one load / store per step); the per-step launch kernel keeps the
// per-substep memory form - carried, its 24 more live registers cost it 1.5 % (2.19 vs 2.15 ms, same box), where in the rollout kernel the launch time does not move
template <bool HF, bool CARRY = false>
__device__ __forceinline__ void env_step_core(const St& S, const Cfg& cfg, int env, bool lead, const float* act10, float& rew_out, int& dn_out) {
    // nothing of the env-step bookkeeping stays in registers across the substeps (the constraint stage needs every one of the 512):
    // the action is re-read at the end, the four per-substep accumulators live in spare hand-off words of the env's LDS region
    constexpr int ACC = c4::WK_ZP2;                     // lfrc, rfrc, lor, ror
    if (lead) {
        for (int u = 0; u < 10; ++u)
            S(F_PDT + u) = act10[u] + kOffset[u] - (cfg.dyn_rand ? S(F_MNOISE + u) : 0.f);     // cassie.py:295-298
        S.I(I_FLAGS) |= 16; S.I(I_ROWSET) = 0;
        for (int k = 0; k < 4; ++k) S.W(ACC + k) = 0.f;
    }
    c4::wsync();
    const int el = threadIdx.x & 15;
    est::Rec erec;
    if constexpr (CARRY) erec = est::rec_load(cfg.est, S.env, el);
    for (int i = 0; i < cfg.simrate; ++i) {
        if constexpr (CARRY) sim_step_pd<HF>(S, cfg, 1, erec);             // all lanes (barriers inside)
        else sim_step_pd<HF>(S, cfg, 1);
        if (!lead) continue;
        {   // every load before the first store (a load behind a store to a word the compiler cannot prove different waits for it: these were ten LDS round trips)
            float fw[16], pv[6], ac[4];
#pragma unroll
            for (int k = 0; k < 16; ++k) fw[k] = S(F_FWD + k);
#pragma unroll
            for (int k = 0; k < 6; ++k) pv[k] = S(F_FOOTPREV + k);
#pragma unroll
            for (int k = 0; k < 4; ++k) ac[k] = S.W(ACC + k);
#pragma unroll
            for (int k = 0; k < 6; ++k) { S(F_FOOTVEL + k) = (fw[10 + k] - pv[k]) / 0.0005f; S(F_FOOTPREV + k) = fw[10 + k]; }      // cassie.py:328-331
            float il = 0.f, ir = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) { il += kNeutralFoot[k] * fw[2 + k]; ir += kNeutralFoot[k] * fw[6 + k]; }
            S.W(ACC + 0) = ac[0] + fw[0]; S.W(ACC + 1) = ac[1] + fw[1];                                                              // cassie.py:418-420
            S.W(ACC + 2) = ac[2] + 1.f - il * il; S.W(ACC + 3) = ac[3] + 1.f - ir * ir;                                              // cassie.py:426-427
        }
    }
    if constexpr (CARRY) { est::rec_store(cfg.est, S.env, el, erec); c4::wsync(); }      // (before write_obs: the min input profile reads the estimator's foot poses from the record)
    if (lead) {
        float act[10];
        for (int u = 0; u < 10; ++u) act[u] = act10[u];
        const float inv = 1.f / (float)cfg.simrate;
        const float lfrc = S.W(ACC + 0) * inv, rfrc = S.W(ACC + 1) * inv, lor = S.W(ACC + 2) * inv, ror = S.W(ACC + 3) * inv;
        const float height = S(F_QPOS + 2);
        const int time = S.I(I_TIME) + 1;
        advance_phase(S);
        S.I(I_TIME) = time; S.I(I_AGE) += 1;
        // non-finite test on the bit pattern, through c4::fbits: it must survive -ffast-math (finite-math-only folds `h != h` AND the plain bit test away)
        const bool h_nan = c4::nonfinite(height);
        int dn = (height < 0.4f || height > 3.0f || h_nan) ? 1 : 0;
        int flags = S.I(I_FLAGS);
        if (!(flags & 4)) for (int u = 0; u < 10; ++u) S(F_PREVACT + u) = act[u];
        if (!(flags & 8)) for (int u = 0; u < 10; ++u) S(F_PREVTQ + u) = S(F_SO + SO_TORQUE + u);
        S.I(I_FLAGS) = flags | 12;
        float rew = clock_reward(S, cfg, act, lfrc, rfrc, lor, ror);
        // a diverged env (non-finite height or reward) ends its episode with reward 0: one NaN in the rollout grid would poison the
        // return scan, the advantage moments and from there every parameter (bit-pattern tests: they must survive -ffast-math)
        if (c4::nonfinite(rew) || h_nan) {
            rew = 0.f; dn = 1;
            S.I(I_FLAGS) &= ~3;      // the encoder filters restart from the first sample after the reset: the joint-velocity biquad is recursive and would carry a NaN for ever
        }
        for (int u = 0; u < 10; ++u) { S(F_PREVACT + u) = act[u]; S(F_PREVTQ + u) = S(F_SO + SO_TORQUE + u); }
        {   // command resampling, cassie.py:483-491; fixed 6 draws per step
            Rng r{cfg.seed_lo, cfg.seed_hi, cfg.env_base + (unsigned)env, (unsigned)S.I(I_RNG), RNG_STEP};
            if (cfg.env_kind == 1 && cfg.dyn_rand) (void)r.u01();                 // cassie_traj.py:463-464 draws a simrate the loop never uses
            { const unsigned k = r.randint(300); const float u = r.uniform(-0.2f, 0.2f); if (k == 0) S(F_CMD + 2) += u; }
            { const unsigned k = r.randint(100); const float u = r.uniform(-0.3f, 4.0f); if (k == 0) S(F_CMD + 0) = fminf(fmaxf(u, -0.3f), 4.0f); }
            { const unsigned k = r.randint(300); const float u = r.uniform(-0.3f, 0.3f); if (k == 0) S(F_CMD + 1) = u; }
            S.I(I_RNG) = (int)r.ctr;
        }
        if (!dn && time >= cfg.max_traj_len) dn = 2;
        rew_out = rew; dn_out = dn;
    }
}
template <bool HF>
__global__ __launch_bounds__(64, APX_WAVES_PER_EU) void env_step_kernel(float* st, int* ist, float* wk, int n, Cfg cfg, const float* action, float* obs,
                                                      float* reward, uint8_t* done, float* final_obs) {
#ifdef APX_WAVETIME
    const unsigned long long wt0__ = clock64();
#endif
    ENV_SETUP
    load_state(S, st, ist, n);
    float rew = 0.f; int dn = 0;
    env_step_core<HF>(S, cfg, env, lead, action + (size_t)env * APX_ACT_DIM, rew, dn);
    if (lead) {
        reward[env] = rew;
        done[env] = (uint8_t)dn;
        write_obs(S, cfg, obs + (size_t)env * cfg.obs_dim);
        if (dn && final_obs) for (int k = 0; k < cfg.obs_dim; ++k) final_obs[(size_t)env * cfg.obs_dim + k] = obs[(size_t)env * cfg.obs_dim + k];
    }
    store_state(S, st, ist, n);
#ifdef APX_WAVETIME
    if (threadIdx.x == 0 && blockIdx.x < 4096) g_wavetime[blockIdx.x] = clock64() - wt0__;
#endif
}

// ------------------------------------------------------------------------------------------------ the T-step rollout as ONE launch (round 5)
// PPO.sample's inner loop (rl/algos/ppo.py:160-184) - actor forward, action = mean + sigma * noise, env step, auto-reset - for T steps inside one kernel.  Envs never
// interact and the policy is fixed during a rollout, so nothing orders one wave against another: a wave steps ITS four envs through all T steps at its own pace.  What the
// per-step launches of rounds 1-4 paid between two env steps - the masked reset launch (73 us, of which 45 are the latency of the settle substep of ~14 envs), the actor
// forward launch (27 us) and the launch gaps, 113 us of every 2 180 - is gone: a restart costs its settle substep to the ONE wave that holds the finished env, the forward
// is 4 rows of fp32 VALU work per wave (~ 15 us), and there is no grid-wide join until the rollout ends.
//
// Policy forward of the wave's four envs (Gaussian_FF_Actor, rl/policies/actor.py:142-215): lane n carries unit 64 g + n of ALL FOUR envs, an input h[e][k] reaches the
// lanes as a scalar operand (v_readlane), the weights stream from L2 in k-major order (Wt[k][n]: one coalesced 256-B row segment per load; re-laid once per rollout by
// transpose_kernel).  Plain sequential-k fp32 fma per unit: not the MFMA forward's summation order - the rollout's means differ from mlp_fused_fwd_kernel's by round-off
// (1e-7), which is why PPO recomputes the old-policy means of the batch with the learner's own forward after the rollout (apex_amd/ppo.py).
constexpr int WK_ACT = c4::WK_QACC + 20;      // the env's current action (10 words): env_step_core reads it at both ends of the step
static_assert(WK_ACT >= c4::WK_DUMMY + 3 && WK_ACT + 10 <= c4::WK_QACC + NV, "action words inside the spare tail of WK_QACC");
struct RolloutArgs {
    const float *Wt0, *b0, *Wt1, *b1, *Wt2, *b2;      // k-major weights [K][N], biases
    const float *mean, *stdv;                          // observation normaliser (may be NULL)
    float sigma; const float* noise;                   // [T, n, A] or NULL
    int H, T;
    float *obs_grid, *act_grid, *mu_grid, *rew_grid; uint8_t* done_grid; float *fin_grid, *obs_next;
    float* rst; int* rst_int;                          // the reset ring, writable (an image missing from it is computed in place)
    Cfg cfg;                                           // a copy of the launch's Cfg for the out-of-line restart (rollout_restart)
    // recurrent actor (apx_rollout_lstm; Gaussian_LSTM_Actor, rl/policies/actor.py:218-311): two LSTMCell(128) + the head (Wt2 / b2 above, H = 128).  Gate matrices k-major
    // [in][4 H] (gate order i, f, g, o), the two bias vectors of a cell as they are in the parameter block, the carried (h, c) of every env [n][2 cells][h | c][128]
    int lstm = 0;
    const float *Gx0 = nullptr, *Gh0 = nullptr, *bi0 = nullptr, *bh0 = nullptr, *Gx1 = nullptr, *Gh1 = nullptr, *bi1 = nullptr, *bh1 = nullptr;
    float* hc = nullptr;
    // deterministic actor (apx_rollout_td3)
    float max_action = 1.f; int noise_scalar = 0;
};
template <int KG, int NG>
__device__ __forceinline__ void ff_layer4(const float* __restrict__ Wt, const float* __restrict__ bias, int K, int N, const float (&hin)[KG][4], float (&hout)[NG][4], bool relu) {
    const int lane = threadIdx.x;
    float acc[NG][4];
    c4::sfor<0, NG>([&](auto G) {
        const int nn = 64 * G + lane;
        const float b = nn < N ? bias[nn] : 0.f;
        c4::sfor<0, 4>([&](auto E) { acc[G][E] = b; });
    });
    c4::sfor<0, KG>([&](auto Kg) {
        constexpr int kg = Kg;
        const int kend = K - 64 * kg < 64 ? K - 64 * kg : 64;
#pragma unroll 4
        for (int kk = 0; kk < kend; ++kk) {
            const float* wr = Wt + (size_t)(64 * kg + kk) * N;
            float w[NG];
            c4::sfor<0, NG>([&](auto G) { const int nn = 64 * G + lane; w[G] = nn < N ? wr[nn] : 0.f; });
            c4::sfor<0, 4>([&](auto E) {
                const float hk = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(hin[kg][E]), kk));
                c4::sfor<0, NG>([&](auto G) { acc[G][E] = fmaf(hk, w[G], acc[G][E]); });
            });
        }
    });
    c4::sfor<0, NG>([&](auto G) { c4::sfor<0, 4>([&](auto E) { hout[G][E] = relu ? fmaxf(acc[G][E], 0.f) : acc[G][E]; }); });
}
// acc[G][E] += sum_k hin_E[k] Wt[k][64 G + lane]  (all N = 64 NG columns exist)
template <int KG, int NG>
__device__ __forceinline__ void ff_accum4(const float* __restrict__ Wt, int K, const float (&hin)[KG][4], float (&acc)[NG][4]) {
    const int lane = threadIdx.x;
    constexpr int N = 64 * NG;
    c4::sfor<0, KG>([&](auto Kg) {
        constexpr int kg = Kg;
        const int kend = K - 64 * kg < 64 ? K - 64 * kg : 64;
#pragma unroll 4
        for (int kk = 0; kk < kend; ++kk) {
            const float* wr = Wt + (size_t)(64 * kg + kk) * N + lane;
            float w[NG];
            c4::sfor<0, NG>([&](auto G) { w[G] = wr[64 * G]; });
            c4::sfor<0, 4>([&](auto E) {
                const float hk = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(hin[kg][E]), kk));
                c4::sfor<0, NG>([&](auto G) { acc[G][E] = fmaf(hk, w[G], acc[G][E]); });
            });
        }
    });
}
__device__ __forceinline__ float sigm_dev(float x) { return 1.f / (1.f + expf(-x)); }
// One LSTMCell(128) for the wave's four envs (torch.nn.LSTMCell: gates = W_ih x + b_ih + W_hh h + b_hh, c' = f c + i g, h' = o tanh c').  Lane j owns the hidden units
// j and 64 + j: gate column 64 G + lane = gate G >> 1 of unit 64 (G & 1) + lane.  hcw = the wave's carried states, cell `cell`.
template <int KG>
__device__ __forceinline__ void lstm_cell4(const float* Gx, int Kx, const float* Gh, const float* bi, const float* bh, float* hcw, int cell, const float (&xin)[KG][4], float (&hout)[2][4]) {
    const int lane = threadIdx.x;
    float g[8][4], h[2][4], c[2][4];
    c4::sfor<0, 8>([&](auto G) { const float b = bi[64 * G + lane] + bh[64 * G + lane]; c4::sfor<0, 4>([&](auto E) { g[G][E] = b; }); });
    c4::sfor<0, 4>([&](auto E) { c4::sfor<0, 2>([&](auto U) {
        const float* q = hcw + 512 * E + 256 * cell + 64 * U + lane;
        h[U][E] = q[0]; c[U][E] = q[128];
    }); });
    ff_accum4<KG, 8>(Gx, Kx, xin, g);
    ff_accum4<2, 8>(Gh, 128, h, g);
    c4::sfor<0, 4>([&](auto E) { c4::sfor<0, 2>([&](auto U) {
        const float i = sigm_dev(g[U][E]), f = sigm_dev(g[2 + U][E]), gg = tanhf(g[4 + U][E]), o = sigm_dev(g[6 + U][E]);
        const float cn = f * c[U][E] + i * gg, hn = o * tanhf(cn);
        float* q = hcw + 512 * E + 256 * cell + 64 * U + lane;
        q[0] = hn; q[128] = cn; hout[U][E] = hn;
    }); });
}
__device__ __forceinline__ const RolloutArgs* ra_ptr(const RolloutArgs* p) { APX_PIN("+s"(p)); return p; }
__device__ __forceinline__ lfloat* env_region(int e) { return (lfloat*)apx_lds4 + e * L4_ES; }
// the recurrent actor's step for the wave's four envs: x = this lane's normalised observation entry of env 0..3; returns the action means (lane < 10).  The carried (h, c)
// wait in HBM / L2 between two steps (the 2 kHz loop needs every register).
struct Mu4 { float v[4]; };
__device__ __noinline__ Mu4 rollout_lstm_actor(const RolloutArgs* rap, int D, int blk, float x0, float x1, float x2, float x3) {
    float* hcw = rap->hc + (size_t)(blk * L4_EPW) * 512;
    const float xin[1][4] = {{x0, x1, x2, x3}};
    float ha[2][4], hb[2][4], mu[1][4];
    lstm_cell4<1>(rap->Gx0, D, rap->Gh0, rap->bi0, rap->bh0, hcw, 0, xin, ha);
    lstm_cell4<2>(rap->Gx1, 128, rap->Gh1, rap->bi1, rap->bh1, hcw, 1, ha, hb);
    ff_layer4<2, 1>(rap->Wt2, rap->b2, 128, APX_ACT_DIM, hb, mu, false);
    return Mu4{{mu[0][0], mu[0][1], mu[0][2], mu[0][3]}};
}
// the restart of the finished envs of a wave inside env_rollout_kernel (all 64 lanes call; `fin` = this lane's env restarts)
template <bool HF>
__device__ __noinline__ void rollout_restart(St S, const RolloutArgs* rap, float* st, int* ist, int n, bool fin) {
    const Cfg cfg = rap->cfg;
    const int l = threadIdx.x & 15, env = S.env;
    const bool lead = l == 0;
    lfloat* stage = S.p + L4_ROWS;
    const int ep = S.I(I_EPISODE) + 1, slot = ep % RST_K;
    const bool miss = fin && rap->rst_int[(size_t)(2 * slot) * n + env] != ep;
    if (__builtin_amdgcn_ballot_w64(miss) != 0ull) {      // the ring does not hold the episode (more resets of one env than prepared images): compute the image in place, like part 0 of env_reset_kernel
        if (miss) {
            store_state(S, st, ist, n);                    // the env's own state waits in HBM while its LDS region is the working copy
            APX_LOCKSTEP();                                // (every lane has copied its words out before the lead's draws overwrite them)
            if (lead) env_reset_draws(S, cfg, ep);
            c4::wsync();
            if (cfg.dyn_rand) setconst_lane(S);
            sim_step_pd<HF>(S, cfg, 0);                    // forward pass only
            c4::wsync();
            float* img = rap->rst + (size_t)slot * F_TOTAL * n + env;
            for (int f = l; f < F_TOTAL; f += 16) img[(size_t)f * n] = S(f);
            if (lead) { rap->rst_int[(size_t)(2 * slot + 1) * n + env] = S.I(I_PHASE); rap->rst_int[(size_t)(2 * slot) * n + env] = ep; }
            __threadfence();
            APX_LOCKSTEP();
            load_state(S, st, ist, n);
        }
        APX_CONVERGE();
    }
    if (fin) {
        env_restart_head(S, cfg, n);
        sim_step_pd<HF>(S, cfg, 1);                        // cassie.py:665 (stale pd_in_t)
        if (lead) env_reset_finish(S, cfg);
        c4::wsync();
        if (lead) write_obs(S, cfg, (float*)stage);
    }
    APX_CONVERGE();
    c4::wsync();
}
// MODE 0: feed-forward Gaussian actor (PPO), 1: recurrent actor (apx_rollout_lstm), 2: feed-forward deterministic actor with tanh head and clipped exploration noise (TD3)
template <bool HF, int MODE = 0>
__global__ __launch_bounds__(64, APX_WAVES_PER_EU) void env_rollout_kernel(float* st, int* ist, float* wk, int n, Cfg cfg, const RolloutArgs* rap) {
    // The rollout's pointers are read from the argument block where they are used (through a pointer the optimiser cannot see through: not hoisted), not carried through the substeps: the constraint stage needs every
    // register, and two dozen loop-invariant pointers in SGPRs spilled into it (v_writelane / scratch inside the 2 kHz loop)
#define RA(f) (ra_ptr(rap)->f)
    ENV_SETUP
    load_state(S, st, ist, n);
    const int D = cfg.obs_dim, A = APX_ACT_DIM, lane = threadIdx.x;
    lfloat* stage = S.p + L4_ROWS;      // the env's observation (<= 64 words) between two steps: the row store is free at a step boundary
    for (int k = l; k < D; k += 16) stage[k] = RA(obs_grid)[(size_t)env * D + k];      // t = 0: the caller's current observation
    if constexpr (MODE == 1) { float* q = RA(hc) + (size_t)(blk * L4_EPW) * 512 + lane; for (int k = 0; k < 4 * 512; k += 64) q[k] = 0.f; }      // every rollout starts at an episode start: init_hidden_state (actor.py:291-293)
    c4::wsync();
    for (int t = 0; t < RA(T); ++t) {
        {   // ---- actor forward of the wave's four envs
            float hin[1][4], mu[1][4];
            const float mn = (RA(mean) && lane < D) ? RA(mean)[lane] : 0.f, sd = (RA(stdv) && lane < D) ? RA(stdv)[lane] : 1.f;
            c4::sfor<0, 4>([&](auto E) { const float v = lane < D ? (env_region(E) + L4_ROWS)[lane] : 0.f; hin[0][E] = lane < D ? (v - mn) / sd : 0.f; });
            if constexpr (MODE == 1) {      // recurrent actor (its own instantiation of the kernel: the feed-forward rollout carries none of this), out of line (cold for the feed-forward rollout; 24 KB of unrolled gate products that the kernel body does not have to carry)
                const Mu4 m = rollout_lstm_actor(rap, D, blk, hin[0][0], hin[0][1], hin[0][2], hin[0][3]);
                c4::sfor<0, 4>([&](auto E) { mu[0][E] = m.v[E]; });
            } else {
                float h1[4][4], h2[4][4];
                ff_layer4<1, 4>(RA(Wt0), RA(b0), D, RA(H), hin, h1, true);
                ff_layer4<4, 4>(RA(Wt1), RA(b1), RA(H), RA(H), h1, h2, true);
                ff_layer4<4, 1>(RA(Wt2), RA(b2), RA(H), A, h2, mu, false);
            }
            if (lane < A) c4::sfor<0, 4>([&](auto E) {
                const int ee = blk * L4_EPW + E;
                const size_t o = ((size_t)t * n + ee) * A + lane;
                float m = mu[0][E], a;
                if constexpr (MODE == 2) {      // FF_Actor.forward (rl/policies/actor.py: max_action * tanh) + the collector's exploration noise, clipped (sync_td3.py:77-78: one scalar per env step; async_td3.py:253-256: per dimension)
                    m = RA(max_action) * tanhf(m);
                    const float nz = RA(noise) ? RA(noise)[RA(noise_scalar) ? (size_t)t * n + ee : o] : 0.f;
                    a = fminf(fmaxf(m + RA(sigma) * nz, -1.f), 1.f);
                } else a = m + (RA(noise) ? RA(sigma) * RA(noise)[o] : 0.f);
                RA(mu_grid)[o] = m; RA(act_grid)[o] = a;
                (env_region(E) + L4_WK + WK_ACT)[lane] = a;
            });
            c4::wsync();
        }
        // ---- env step
        float rew = 0.f; int dn = 0;
        env_step_core<HF, true>(S, cfg, env, lead, (const float*)&S.W(WK_ACT), rew, dn);
        if (lead) {
            RA(rew_grid)[(size_t)t * n + env] = rew; RA(done_grid)[(size_t)t * n + env] = (uint8_t)dn;
            write_obs(S, cfg, (float*)stage);
            S.W(c4::WK_MISC + 7) = (float)dn;
        }
        c4::wsync();
        const bool fin = S.W(c4::WK_MISC + 7) != 0.f;
        if (fin) for (int k = l; k < D; k += 16) RA(fin_grid)[((size_t)t * n + env) * D + k] = stage[k];
        // ---- auto-reset of the finished envs (CassieEnv.reset, cassie.py:523-680): the restart part of env_reset_kernel, on this wave alone, OUT OF LINE - inlined, the
        // scalars it needs (ring pointers, seeds, the estimator's lifetime ...) stayed live through the 2 kHz loop above and spilled into it
        if (__builtin_amdgcn_ballot_w64(fin) != 0ull) {
            rollout_restart<HF>(S, rap, st, ist, n, fin);
            if (MODE == 1 && fin) { float* q = RA(hc) + (size_t)env * 512 + l; for (int k = 0; k < 512; k += 16) q[k] = 0.f; }      // a new episode starts from the zero state (ppo.py:164-168)
            if (MODE == 1) APX_LOCKSTEP();      // (the next step's cell loads every env's (h, c) on all 64 lanes: behind these stores in the wave's memory order)
        }
        float* on = (t + 1 < RA(T) ? RA(obs_grid) + (size_t)(t + 1) * n * D : RA(obs_next)) + (size_t)env * D;
        for (int k = l; k < D; k += 16) on[k] = stage[k];
    }
    store_state(S, st, ist, n);
#undef RA
}
__global__ void transpose_kernel(const float* __restrict__ W /* [N][K] */, float* __restrict__ Wt /* [K][N] */, int N, int K) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * K) return;
    const int k = i / N, nn = i - k * N;
    Wt[i] = W[(size_t)nn * K + k];
}

// action == NULL: raw substeps with the current pd targets (tests): n_sub x cassie_sim_step_pd.
// action != NULL: CassieEnv.step_basic (cassie/cassie.py:498-521, 355-387): new pd targets, n_sub substeps, time / phase bookkeeping,
// observation; no reward, termination, trackers or command resampling
template <bool HF>
__global__ __launch_bounds__(64, APX_WAVES_PER_EU) void env_substep_kernel(float* st, int* ist, float* wk, int n, Cfg cfg, int n_sub,
                                                                            const float* action, float* obs) {
    ENV_SETUP
    load_state(S, st, ist, n);
    if (action && lead) {
        for (int u = 0; u < 10; ++u) S(F_PDT + u) = action[(size_t)env * APX_ACT_DIM + u] + kOffset[u] - (cfg.dyn_rand ? S(F_MNOISE + u) : 0.f);
        S.I(I_FLAGS) |= 16; S.I(I_ROWSET) = 0;      // (the row-set hash covers the substeps of this step_basic, like env_step's)
    }
    c4::wsync();
    for (int i = 0; i < n_sub; ++i) sim_step_pd<HF>(S, cfg, 1);
    if (action && lead) {
        S.I(I_TIME) += 1;
        advance_phase(S);
        if (obs) write_obs(S, cfg, obs + (size_t)env * cfg.obs_dim);
    }
    store_state(S, st, ist, n);
}
// CassieEnv.reset_for_test(full_reset=False) (cassie/cassie.py:682-742)
template <bool HF>
__global__ __launch_bounds__(64, APX_WAVES_PER_EU) void env_reset_for_test_kernel(float* st, int* ist, float* wk, int n, Cfg cfg, float* obs, int full) {
    ENV_SETUP
    load_state(S, st, ist, n);
    if (lead) {
        S.I(I_PHASE) = 0; S.I(I_TIME) = 0; S.I(I_COUNTER) = 0; S.I(I_FLAGS) &= ~(32 | 64);      // phase_add = 1 (cassie.py:687)
        S(F_CMD + 0) = 0.f; S(F_CMD + 2) = 0.f;                                  // speed, orient_add (side speed is NOT reset)
        S(F_CMD + 3) = 0.15f; S(F_CMD + 4) = 0.25f; S(F_CMD + 5) = (2.f * 0.15f + 2.f * 0.25f) * (float)(2000 / cfg.simrate); S(F_CMD + 6) = 1.f;      // grounded (cassie.py:699-702)
    }
    c4::wsync();
    if (full) {
        // cassie_sim_full_reset (binary: qpos <- init pose, qvel / ctrl / qfrc_applied / xfrc_applied / qacc <- 0, the 6 x 10 torque delay
        // line <- 0, state_output_setup; qacc_warmstart and the encoder filters are NOT touched) + reset_cassie_state (cassie.py:733-746)
        if (lead) {
            for (int i = 0; i < NQ; ++i) S(F_QPOS + i) = cm_init_qpos[i];
            for (int i = 0; i < NV; ++i) S(F_QVEL + i) = 0.f;
            for (int i = 0; i < 60; ++i) S(F_FIFO + i) = 0.f;
            for (int i = 0; i < 6; ++i) S(F_XFRC + i) = 0.f;
            const float mp[5] = {0.0045f, 0.f, 0.4973f, -1.1997f, -1.5968f}, jp[3] = {0.f, 1.4267f, -1.5968f};
            for (int u = 0; u < 10; ++u) { S(F_SO + SO_MPOS + u) = mp[u % 5]; S(F_SO + SO_MVEL + u) = 0.f; }
            for (int k = 0; k < 6; ++k) { S(F_SO + SO_JPOS + k) = jp[k % 3]; S(F_SO + SO_JVEL + k) = 0.f; }
            S(F_SO + SO_QUAT) = 1.f;
            for (int k = 0; k < 3; ++k) { S(F_SO + SO_QUAT + 1 + k) = 0.f; S(F_SO + SO_ROTVEL + k) = 0.f; S(F_SO + SO_TVEL + k) = 0.f; S(F_SO + SO_TACC + k) = 0.f; }
            S(F_SO + SO_HEIGHT) = 1.01f;                                         // pelvis.position[2] = 1.01, terrain.height = 0
        }
        if (l < 7) { est::Rec z; c4::sfor<0, 6>([&](auto K) { z.v[K] = est::f4{0.f, 0.f, 0.f, 0.f}; }); est::rec_store(cfg.est, S.env, l, z); }      // state_output_setup: the estimator restarts
        if (lead) S.I(I_AGE) = 0;
        c4::wsync();
    } else {
        sim_step_pd<HF>(S, cfg, 1);                                        // self.cassie_state = self.sim.step_pd(self.u), stale targets
    }
    if (cfg.dyn_rand) {                                                          // default dynamics, set_const (back to the init pose), flat floor, no encoder offsets
        if (lead) {
            for (int b = 0; b < NB; ++b) S(F_MASS + b) = cm_body_mass[b];
            for (int d = 0; d < NV; ++d) S(F_DAMP + d) = cm_dof_damping[d];
            S(F_FRIC) = 1.f;
            const float fl[9] = {0, 0, 1, 0, 1, 0, -1, 0, 0};
            for (int k = 0; k < 9; ++k) S(F_FLOOR + k) = fl[k];
            for (int u = 0; u < 10; ++u) S(F_MNOISE + u) = 0.f;
            for (int k = 0; k < 6; ++k) S(F_JNOISE + k) = 0.f;
            for (int i = 0; i < NQ; ++i) S(F_QPOS + i) = cm_init_qpos[i];
            for (int i = 0; i < NV; ++i) { S(F_QVEL + i) = 0.f; S(F_QACCW + i) = 0.f; }
        }
        c4::wsync();
        setconst_lane(S);
        sim_step_pd<HF>(S, cfg, 0);                                        // cassie_sim_set_const ends in mj_forward
    }
    if (lead) write_obs(S, cfg, obs + (size_t)env * cfg.obs_dim);
    store_state(S, st, ist, n);
}
static constexpr size_t LDS_BYTES = (size_t)(L4_EPW * L4_ES + CT_TOTAL) * sizeof(float);      // env regions + wave-constant table   // 37,120 B per wave, 4 waves per CU
#define ENV_GRID(n) dim3((n) / L4_EPW)
#define ENV_BLOCK dim3(64)
#define SETCONST_GRID(n) dim3((n) / L4_EPW)
#define SETCONST_LDS LDS_BYTES

// CassieEnv.update_speed (cassie/cassie.py:757-775), clock command profile; the phase rescale is done in fp64 like the reference
// (the old cycle length comes from the stored fp32 swing / stance, so the truncation can differ from an all-fp64 evaluation only
// when the quotient is within fp32 rounding of an integer)
__global__ void env_update_speed_kernel(float* st, int* ist, int n, Cfg cfg, const float* speed, const float* side) {
    const int env = blockIdx.x * blockDim.x + threadIdx.x;
    if (env >= n) return;
    auto F = [&](int f) -> float& { return st[(size_t)f * n + env]; };
    const float sp = fminf(fmaxf(speed[env], -0.3f), 4.0f), sd = side ? fminf(fmaxf(side[env], -0.3f), 0.3f) : 0.f;
    if (cfg.command_profile != 0) { F(F_CMD + 0) = sp; F(F_CMD + 1) = sd; return; }      // cassie.py:756-761: durations kept (>= 0.01 by construction), phase unchanged
    const double s = sp, total = (0.9 - 0.25 / 3.0 * s) / 2;
    const double swing = (0.30 + ((0.70 - 0.30) / 3) * s) * total, stance = (0.70 - ((0.70 - 0.30) / 3) * s) * total;
    const double freq = (double)(2000 / cfg.simrate);
    const double old_len = (2.0 * (double)F(F_CMD + 3) + 2.0 * (double)F(F_CMD + 4)) * freq, new_len = (2.0 * swing + 2.0 * stance) * freq;
    int& phase = ist[(size_t)I_PHASE * n + env];
    int& flg = ist[(size_t)I_FLAGS * n + env];
    phase = (int)(new_len * ((double)phase + 0.5 * (double)((flg >> 5) & 1)) / old_len); flg &= ~32;
    F(F_CMD + 0) = sp; F(F_CMD + 1) = sd; F(F_CMD + 3) = (float)swing; F(F_CMD + 4) = (float)stance; F(F_CMD + 5) = (float)new_len;
}

// ------------------------------------------------------------------------------------------------ C ABI

static Cfg make_cfg(const apx_env& env) {
    const apx_env_cfg& c = env.cfg;
    return Cfg{Hf{env.hf, env.hf_nrow, env.hf_ncol, env.hf_size[0], env.hf_size[1], env.hf_size[2]}, c.simrate, c.dynamics_randomization, c.stance_mode, c.have_incentive, c.max_traj_len, c.pgs_iters,
               (unsigned)c.seed, (unsigned)(c.seed >> 32), (unsigned)c.env_id_base, c.reward_kind, c.env_kind, c.command_profile,
               (c.input_profile ? APX_OBS_MIN : 46) + (c.command_profile == 0 ? 4 : 9), c.est_lifetime, c.input_profile, env.wk, env.rst, env.rst_int, env.complete_rows ? env.cp_pool : nullptr};
}

extern "C" void apx_env_default_cfg(apx_env_cfg* c) {
    if (!c) return;
    *c = apx_env_cfg{};
    c->n_envs = 4096; c->simrate = 50; c->dynamics_randomization = 1; c->reward_kind = 0; c->stance_mode = 0;
    c->have_incentive = 1; c->max_traj_len = 400; c->seed = 0; c->device = 0; c->pgs_iters = 50; c->est_lifetime = 169;      /* 5096 // 30, apex.py:244-246 */
}

extern "C" int apx_env_create(const apx_env_cfg* cfg, apx_env_t** out) {
    APX_REQUIRE(cfg && out, "null");
    APX_REQUIRE(cfg->n_envs > 0 && cfg->n_envs % 64 == 0, "n_envs must be a positive multiple of 64");
    APX_REQUIRE(cfg->simrate > 0 && cfg->simrate <= 2000, "simrate: 1..2000 substeps per env step (cycle lengths follow 2000 // simrate like cassie.py:545)");      // (the shipped policies of the reference ran at 60)
    APX_REQUIRE(cfg->env_kind == 0 || (cfg->env_kind == 1 && cfg->simrate == 50), "env_kind: 0 Cassie-v0, 1 CassieTraj-v0 (walking trajectory table is for simrate 50)");
    APX_REQUIRE(cfg->reward_kind >= 0 && cfg->reward_kind <= 2, "reward_kind: 0 clock_reward, 1 early_clock_reward, 2 max_vel_clock_reward");
    APX_REQUIRE(cfg->pgs_iters > 0 && cfg->max_traj_len > 0, "pgs_iters / max_traj_len");
    APX_REQUIRE(cfg->input_profile == 0 || cfg->input_profile == 1, "input_profile: 0 full, 1 min");
    APX_REQUIRE(cfg->command_profile >= 0 && cfg->command_profile <= 2 && (cfg->command_profile == 0 || cfg->env_kind == 0), "command_profile: 0 clock, 1 phase, 2 phase (library draws); phase needs Cassie-v0");
    APX_HIP(hipSetDevice(cfg->device));
    apx_env* e = new (std::nothrow) apx_env;
    APX_REQUIRE(e, "alloc");
    e->cfg = *cfg; e->n = cfg->n_envs; e->st = nullptr; e->ist = nullptr;
    APX_HIP(hipMalloc(&e->st, sizeof(float) * (size_t)F_TOTAL * e->n));
    APX_HIP(hipMalloc(&e->ist, sizeof(int) * (size_t)I_TOTAL * e->n));
    e->wk = nullptr;
    APX_HIP(hipMalloc(&e->wk, sizeof(float) * (size_t)est::REC * e->n));      // state-estimator records (estimator_lane.h); the stage hand-off itself lives in LDS
    APX_HIP(hipMemset(e->wk, 0, sizeof(float) * (size_t)est::REC * e->n));  // state_output_setup
    e->rst = nullptr; e->rst_int = nullptr;
    APX_HIP(hipMalloc(&e->rst, sizeof(float) * (size_t)RST_K * F_TOTAL * e->n));
    APX_HIP(hipMalloc(&e->rst_int, sizeof(int) * (size_t)RST_K * 2 * e->n));
    APX_HIP(hipMemset(e->rst_int, 0xFF, sizeof(int) * (size_t)RST_K * 2 * e->n));      // -1: no slot holds an episode
    e->hf = nullptr; e->hf_nrow = e->hf_ncol = 0; e->hf_size[0] = e->hf_size[1] = e->hf_size[2] = 0.f;
    e->timing = 0; e->ev = nullptr; e->ev_cap = e->ev_n = 0; e->t_ms = 0.0; e->t_launches = 0;
    // up to 2048 envs leave half of the SIMDs idle during an env step: the images of the restarted envs' next episodes are computed there (apx_env_set_refill overrides)
    e->refill = getenv("APX_REFILL") ? atoi(getenv("APX_REFILL")) != 0 : e->n <= 2048;
    e->refill_pending = 0; e->refill_due = 0; e->side = nullptr; e->ev_reset = nullptr; e->ev_refill = nullptr;
    e->pol_wt = nullptr; e->pol_wt_n = 0; e->roll_launches = 0; e->roll_ms = 0.0;
    e->complete_rows = getenv("APX_COMPLETE_ROWS") ? atoi(getenv("APX_COMPLETE_ROWS")) != 0 : 1;
    e->cp_pool = nullptr;
    APX_HIP(hipMalloc(&e->cp_pool, sizeof(float) * (size_t)e->n * c4::CP_HBM_CAP * c4::CP_STRIDE));
    const Cfg c = make_cfg(*e);
    hipLaunchKernelGGL(env_init_kernel, dim3(e->n / 64), dim3(64), 0, 0, e->st, e->ist, e->n, c);
    APX_LAUNCH_CHECK();
    for (const void* fn : {(const void*)env_rollout_kernel<false>, (const void*)env_rollout_kernel<true>, (const void*)env_step_kernel<false>, (const void*)env_step_kernel<true>, (const void*)env_reset_kernel<false>, (const void*)env_reset_kernel<true>, 
                           (const void*)env_substep_kernel<false>, (const void*)env_substep_kernel<true>, (const void*)env_reset_for_test_kernel<false>,
                           (const void*)env_reset_for_test_kernel<true>})
        APX_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES));
    hipLaunchKernelGGL(env_setconst_kernel, SETCONST_GRID(e->n), dim3(64), SETCONST_LDS, 0, e->st, e->ist, e->wk, e->n, c);
    APX_LAUNCH_CHECK();
    APX_HIP(hipDeviceSynchronize());
    *out = e;
    return APX_OK;
}

extern "C" int apx_env_destroy(apx_env_t* e) {
    if (!e) return APX_OK;
    if (e->side) (void)hipStreamSynchronize((hipStream_t)e->side);      // a ring refill in flight on the env's own stream still reads the state and writes the ring
    (void)hipFree(e->pol_wt); (void)hipFree(e->cp_pool); (void)hipFree(e->st); (void)hipFree(e->ist); (void)hipFree(e->wk); (void)hipFree(e->hf); (void)hipFree(e->rst); (void)hipFree(e->rst_int);
    for (int i = 0; i < e->ev_cap; ++i) (void)hipEventDestroy((hipEvent_t)e->ev[i]);
    free(e->ev);
    if (e->side) (void)hipStreamDestroy((hipStream_t)e->side);
    if (e->ev_reset) (void)hipEventDestroy((hipEvent_t)e->ev_reset);
    if (e->ev_refill) (void)hipEventDestroy((hipEvent_t)e->ev_refill);
    delete e;
    return APX_OK;
}

// prepared resets depend on the model inputs of the forward pass (terrain, external wrench, fields written through the setters): drop them when one of those changes
// a refill launched by the previous step writes the ring on the side stream: whatever touches the ring next on `stream` waits for it
static int refill_join(apx_env* e, void* stream) {
    if (e->refill_pending) {
        // a stream under graph capture cannot wait for an event recorded outside the capture: drain the refill on the host instead (it was launched by an earlier,
        // uncaptured step; the capture then starts from a ring nobody writes)
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing((hipStream_t)stream, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone) APX_HIP(hipEventSynchronize((hipEvent_t)e->ev_refill));
        else APX_HIP(hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)e->ev_refill, 0));
        e->refill_pending = 0;
    }
    return APX_OK;
}
static int invalidate_prepared(apx_env* e, void* stream) {
    { const int rc = refill_join(e, stream); if (rc != APX_OK) return rc; }
    APX_HIP(hipMemsetAsync(e->rst_int, 0xFF, sizeof(int) * (size_t)RST_K * 2 * e->n, (hipStream_t)stream));
    return APX_OK;
}
static int launch_reset_raw(apx_env* e, int ahead, const uint8_t* mask, float* obs, void* stream);
static int launch_reset(apx_env* e, int ahead, const uint8_t* mask, float* obs, void* stream) {
    { const int rc = refill_join(e, stream); if (rc != APX_OK) return rc; }
    return launch_reset_raw(e, ahead, mask, obs, stream);
}
// In-rollout refill: the envs that restarted in the reset just launched on `stream` have consumed a ring slot; part 0 of env_reset_kernel for their next RST_K episodes
// runs on the side stream (no mask: a wave whose four envs hold their next episodes leaves at once), next to whatever `stream` does until its next reset launch.  Part 0
// reads I_EPISODE (stable between resets) and writes only the ring; its result is a function of (seed, env, episode) alone, not of the env state it starts from.
static int launch_refill(apx_env* e, void* stream) {
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing((hipStream_t)stream, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone) return APX_OK;      // a captured rollout keeps to its one stream
    if (!e->side) {
        APX_HIP(hipStreamCreateWithFlags((hipStream_t*)&e->side, hipStreamNonBlocking));
        APX_HIP(hipEventCreateWithFlags((hipEvent_t*)&e->ev_reset, hipEventDisableTiming));
        APX_HIP(hipEventCreateWithFlags((hipEvent_t*)&e->ev_refill, hipEventDisableTiming));
    }
    APX_HIP(hipEventRecord((hipEvent_t)e->ev_reset, (hipStream_t)stream));
    APX_HIP(hipStreamWaitEvent((hipStream_t)e->side, (hipEvent_t)e->ev_reset, 0));
    for (int ahead = 1; ahead <= RST_K; ++ahead) { const int rc = launch_reset_raw(e, ahead, nullptr, nullptr, e->side); if (rc != APX_OK) return rc; }
    APX_HIP(hipEventRecord((hipEvent_t)e->ev_refill, (hipStream_t)e->side));
    e->refill_pending = 1;
    return APX_OK;
}
extern "C" int apx_env_set_complete_rows(apx_env_t* e, int on) {
    APX_REQUIRE(e, "env");
    e->complete_rows = on != 0;
    return invalidate_prepared(e, nullptr);      // a prepared image holds a forward pass of the other kind
}
extern "C" int apx_env_set_refill(apx_env_t* e, int on) {
    APX_REQUIRE(e, "env");
    e->refill = on != 0;
    if (!on) e->refill_due = 0;
    if (!on && e->refill_pending) { APX_HIP(hipEventSynchronize((hipEvent_t)e->ev_refill)); e->refill_pending = 0; }      // switching off drains the refill in flight
    return APX_OK;
}
static int launch_reset_raw(apx_env* e, int ahead, const uint8_t* mask, float* obs, void* stream) {
    if (e->hf) hipLaunchKernelGGL(HIP_KERNEL_NAME(env_reset_kernel<true>), ENV_GRID(e->n), ENV_BLOCK, LDS_BYTES, (hipStream_t)stream, e->st, e->ist, e->wk, e->n, make_cfg(*e), e->rst, e->rst_int, ahead, mask, obs);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(env_reset_kernel<false>), ENV_GRID(e->n), ENV_BLOCK, LDS_BYTES, (hipStream_t)stream, e->st, e->ist, e->wk, e->n, make_cfg(*e), e->rst, e->rst_int, ahead, mask, obs);
    APX_LAUNCH_CHECK();
    return APX_OK;
}
extern "C" int apx_env_prepare_resets(apx_env_t* e, void* stream) {
    APX_REQUIRE(e, "env");
    for (int ahead = 1; ahead <= RST_K; ++ahead) { const int rc = launch_reset(e, ahead, nullptr, nullptr, stream); if (rc != APX_OK) return rc; }
    return APX_OK;
}

extern "C" int apx_env_set_hfield(apx_env_t* e, const float* data, int nrow, int ncol, const float* size3, void* stream) {
    APX_REQUIRE(e, "env");
    APX_HIP(hipStreamSynchronize((hipStream_t)stream));           // kernels in flight still read the old field
    if (e->refill_pending) { APX_HIP(hipEventSynchronize((hipEvent_t)e->ev_refill)); e->refill_pending = 0; }      // ... and so does a ring refill on the env's own stream
    { const int rc = invalidate_prepared(e, stream); if (rc != APX_OK) return rc; }
    (void)hipFree(e->hf); e->hf = nullptr; e->hf_nrow = e->hf_ncol = 0;
    if (!data) return APX_OK;                                     // back to the plane
    APX_REQUIRE(nrow >= 2 && ncol >= 2 && size3 && size3[0] > 0.f && size3[1] > 0.f, "height field: nrow, ncol >= 2 and positive half extents");
    APX_HIP(hipMalloc(&e->hf, sizeof(float) * (size_t)nrow * ncol));
    APX_HIP(hipMemcpy(e->hf, data, sizeof(float) * (size_t)nrow * ncol, hipMemcpyDefault));      // host or device source
    e->hf_nrow = nrow; e->hf_ncol = ncol; e->hf_size[0] = size3[0]; e->hf_size[1] = size3[1]; e->hf_size[2] = size3[2];
    return APX_OK;
}

extern "C" int apx_env_reset(apx_env_t* e, const uint8_t* mask, float* obs_out, void* stream) {
    APX_REQUIRE(e, "env");
    return launch_reset(e, 0, mask, obs_out, stream);
}

extern "C" int apx_env_update_speed(apx_env_t* e, const float* speed, const float* side_speed, void* stream) {
    APX_REQUIRE(e && speed, "null pointer");
    hipLaunchKernelGGL(env_update_speed_kernel, dim3(apx_cdiv((long)e->n, 256)), dim3(256), 0, (hipStream_t)stream, e->st, e->ist, e->n,
                       make_cfg(*e), speed, side_speed);
    APX_LAUNCH_CHECK();
    return APX_OK;
}

extern "C" int apx_env_step_basic(apx_env_t* e, const float* action, float* obs, void* stream) {
    APX_REQUIRE(e && action && obs, "null pointer");
    if (e->hf) hipLaunchKernelGGL(HIP_KERNEL_NAME(env_substep_kernel<true>), ENV_GRID(e->n), ENV_BLOCK, LDS_BYTES, (hipStream_t)stream, e->st, e->ist, e->wk, e->n, make_cfg(*e),
                       e->cfg.simrate, action, obs);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(env_substep_kernel<false>), ENV_GRID(e->n), ENV_BLOCK, LDS_BYTES, (hipStream_t)stream, e->st, e->ist, e->wk, e->n, make_cfg(*e),
                       e->cfg.simrate, action, obs);
    APX_LAUNCH_CHECK();
    return APX_OK;
}

extern "C" int apx_env_reset_for_test(apx_env_t* e, float* obs_out, int full_reset, void* stream) {
    APX_REQUIRE(e && obs_out, "null pointer");
    e->cfg.stance_mode = 1;                               // reset_for_test switches to the grounded clock (cassie.py:702)
    if (e->hf) hipLaunchKernelGGL(HIP_KERNEL_NAME(env_reset_for_test_kernel<true>), ENV_GRID(e->n), ENV_BLOCK, LDS_BYTES, (hipStream_t)stream, e->st, e->ist, e->wk, e->n,
                       make_cfg(*e), obs_out, full_reset);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(env_reset_for_test_kernel<false>), ENV_GRID(e->n), ENV_BLOCK, LDS_BYTES, (hipStream_t)stream, e->st, e->ist, e->wk, e->n,
                       make_cfg(*e), obs_out, full_reset);
    APX_LAUNCH_CHECK();
    return APX_OK;
}

__global__ void scatter_kernel(float* st, int n, int f0, int cnt, const float* in);
__global__ void fill_int_kernel(int* p, int n, int v) { const int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = v; }
extern "C" int apx_env_apply_force_body(apx_env_t* e, const float* xfrc, int body, void* stream) {
    APX_REQUIRE(e && xfrc, "null pointer");
    APX_REQUIRE(body >= 1 && body < ES_NB, "body id out of range (1 = cassie-pelvis ... 25 = right-foot)");
    { const int rc = invalidate_prepared(e, stream); if (rc != APX_OK) return rc; }
    hipLaunchKernelGGL(scatter_kernel, dim3(apx_cdiv((long)e->n * 6, 256)), dim3(256), 0, (hipStream_t)stream, e->st, e->n, (int)F_XFRC, 6, xfrc);
    hipLaunchKernelGGL(fill_int_kernel, dim3(apx_cdiv((long)e->n, 256)), dim3(256), 0, (hipStream_t)stream, e->ist + (size_t)I_XBODY * e->n, e->n, body);
    APX_LAUNCH_CHECK();
    return APX_OK;
}
extern "C" int apx_env_apply_force(apx_env_t* e, const float* xfrc, void* stream) { return apx_env_apply_force_body(e, xfrc, 1, stream); }

extern "C" int apx_env_step(apx_env_t* e, const float* action, float* obs, float* reward, uint8_t* done, float* final_obs,
                            int auto_reset, void* stream) {
    APX_REQUIRE(e && action && obs && reward && done, "null pointer");
    if (e->refill && e->refill_due) { e->refill_due = 0; const int rc = launch_refill(e, stream); if (rc != APX_OK) return rc; }      // runs next to the env step launched below
    const bool timed = e->timing && e->ev_n + 2 <= e->ev_cap;
    if (timed) APX_HIP(hipEventRecord((hipEvent_t)e->ev[e->ev_n], (hipStream_t)stream));
    if (e->hf) hipLaunchKernelGGL(HIP_KERNEL_NAME(env_step_kernel<true>), ENV_GRID(e->n), ENV_BLOCK, LDS_BYTES, (hipStream_t)stream, e->st, e->ist, e->wk, e->n,
                       make_cfg(*e), action, obs, reward, done, final_obs);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(env_step_kernel<false>), ENV_GRID(e->n), ENV_BLOCK, LDS_BYTES, (hipStream_t)stream, e->st, e->ist, e->wk, e->n,
                       make_cfg(*e), action, obs, reward, done, final_obs);
    APX_LAUNCH_CHECK();
    if (timed) { APX_HIP(hipEventRecord((hipEvent_t)e->ev[e->ev_n + 1], (hipStream_t)stream)); e->ev_n += 2; }
    if (auto_reset) {      // finished envs restart in a second launch on the same stream (mask = done flags)
        { const int rc = launch_reset(e, 0, done, obs, stream); if (rc != APX_OK) return rc; }
        e->refill_due = e->refill;      // launched in front of the NEXT env step (below): right behind the reset it would share the chip with the latency-bound policy step
    }
    return APX_OK;
}

// Kernel timing for the roofline line of bench.py: while enabled, every env_step_kernel launch (apx_env_step and the steps inside apx_rollout) is
// bracketed by a hipEvent pair on the launch stream; apx_env_timing_read drains the pairs into (total ms, launches).  Up to 4096 launches
// between two reads are recorded, later ones run untimed.
#define APX_TRY(x) do { int rc__ = (x); if (rc__ != APX_OK) return rc__; } while (0)
static int timing_drain(apx_env* e) {
    for (int i = 0; i + 1 < e->ev_n; i += 2) {
        APX_HIP(hipEventSynchronize((hipEvent_t)e->ev[i + 1]));
        float ms = 0.f;
        APX_HIP(hipEventElapsedTime(&ms, (hipEvent_t)e->ev[i], (hipEvent_t)e->ev[i + 1]));
        e->t_ms += ms; e->t_launches += 1;
    }
    e->ev_n = 0;
    return APX_OK;
}
extern "C" int apx_env_timing(apx_env_t* e, int enable) {
    APX_REQUIRE(e, "env");
    if (enable && !e->ev) {
        const int cap = 8192;
        e->ev = (void**)calloc(cap, sizeof(void*));
        APX_REQUIRE(e->ev, "alloc");
        for (int i = 0; i < cap; ++i) { hipEvent_t ev; APX_HIP(hipEventCreate(&ev)); e->ev[i] = ev; e->ev_cap = i + 1; }
    }
    if (!enable) APX_TRY(timing_drain(e));
    e->timing = enable ? 1 : 0;
    return APX_OK;
}
extern "C" int apx_env_timing_read(apx_env_t* e, double* total_ms, int64_t* launches, int reset) {
    APX_REQUIRE(e && total_ms && launches, "null pointer");
    APX_TRY(timing_drain(e));
    *total_ms = e->t_ms; *launches = e->t_launches;
    if (reset) { e->t_ms = 0.0; e->t_launches = 0; }
    return APX_OK;
}

// ---- apx_rollout: PPO.sample's inner loop (rl/algos/ppo.py:160-181) for the whole batch as ONE C-ABI call: T x (actor forward -> action =
// mean + sigma * noise -> env step with auto-reset), every result written straight into the caller's [T, N, .] rollout grids
__global__ void act_noise_kernel(const float* __restrict__ mu, const float* __restrict__ noise, float sigma, long n, float* __restrict__ act) {
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i < n) act[i] = mu[i] + (noise ? sigma * noise[i] : 0.f);
}
static int rollout_ff(apx_env_t* e, int td3, float max_action, int noise_scalar, const float* actor, int H, const float* obs_mean, const float* obs_std, float sigma, const float* noise, int T,
                      float* obs_grid, float* act_grid, float* mu_grid, float* rew_grid, uint8_t* done_grid, float* fin_grid, float* obs_next, void* stream) {
    APX_REQUIRE(e && actor && obs_grid && act_grid && mu_grid && rew_grid && done_grid && fin_grid && obs_next && T > 0, "rollout arguments");
    const int D = make_cfg(*e).obs_dim, A = APX_ACT_DIM;
    const long N = e->n;
    {   // ONE launch for the whole rollout (env_rollout_kernel) for the reference's 2 x 256 actor; APX_ROLLOUT_STEPWISE=1 or a stream under graph capture keep the per-step launches
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        const bool capturing = hipStreamIsCapturing((hipStream_t)stream, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone;
        static const bool stepwise = getenv("APX_ROLLOUT_STEPWISE") && atoi(getenv("APX_ROLLOUT_STEPWISE")) != 0;
        if (H == 256 && D <= 64 && !stepwise && !capturing) {
            const long nw = (long)D * H + (long)H * H + (long)H * A;
            if (e->pol_wt_n != nw) { (void)hipFree(e->pol_wt); e->pol_wt = nullptr; APX_HIP(hipMalloc(&e->pol_wt, sizeof(float) * nw + sizeof(RolloutArgs) + 16)); e->pol_wt_n = nw; }
            { const int rc = refill_join(e, stream); if (rc != APX_OK) return rc; }      // a ring refill of an earlier stepwise call
            e->refill_due = 0;
            const float* W0 = actor; const float* b0 = W0 + (long)H * D; const float* W1 = b0 + H; const float* b1 = W1 + (long)H * H; const float* W2 = b1 + H; const float* b2 = W2 + (long)A * H;
            float* Wt0 = e->pol_wt; float* Wt1 = Wt0 + (long)D * H; float* Wt2 = Wt1 + (long)H * H;
            hipLaunchKernelGGL(transpose_kernel, dim3(apx_cdiv((long)H * D, 256)), dim3(256), 0, (hipStream_t)stream, W0, Wt0, H, D);
            hipLaunchKernelGGL(transpose_kernel, dim3(apx_cdiv((long)H * H, 256)), dim3(256), 0, (hipStream_t)stream, W1, Wt1, H, H);
            hipLaunchKernelGGL(transpose_kernel, dim3(apx_cdiv((long)A * H, 256)), dim3(256), 0, (hipStream_t)stream, W2, Wt2, A, H);
            APX_LAUNCH_CHECK();
            RolloutArgs ra{Wt0, b0, Wt1, b1, Wt2, b2, obs_mean, obs_std, sigma, noise, H, T, obs_grid, act_grid, mu_grid, rew_grid, done_grid, fin_grid, obs_next, e->rst, e->rst_int, make_cfg(*e)};
            ra.max_action = max_action; ra.noise_scalar = noise_scalar;
            RolloutArgs* rap = (RolloutArgs*)(e->pol_wt + nw);      // the argument block lives behind the weights (the kernel reads it field by field where it needs one)
            APX_HIP(hipMemcpyAsync(rap, &ra, sizeof(ra), hipMemcpyHostToDevice, (hipStream_t)stream));
            const bool timed = e->timing && e->ev_n + 2 <= e->ev_cap;
            if (timed) APX_HIP(hipEventRecord((hipEvent_t)e->ev[e->ev_n], (hipStream_t)stream));
            if (td3) {
                if (e->hf) hipLaunchKernelGGL(HIP_KERNEL_NAME(env_rollout_kernel<true, 2>), ENV_GRID(e->n), ENV_BLOCK, LDS_BYTES, (hipStream_t)stream, e->st, e->ist, e->wk, e->n, make_cfg(*e), (const RolloutArgs*)rap);
                else hipLaunchKernelGGL(HIP_KERNEL_NAME(env_rollout_kernel<false, 2>), ENV_GRID(e->n), ENV_BLOCK, LDS_BYTES, (hipStream_t)stream, e->st, e->ist, e->wk, e->n, make_cfg(*e), (const RolloutArgs*)rap);
            } else if (e->hf) hipLaunchKernelGGL(HIP_KERNEL_NAME(env_rollout_kernel<true>), ENV_GRID(e->n), ENV_BLOCK, LDS_BYTES, (hipStream_t)stream, e->st, e->ist, e->wk, e->n, make_cfg(*e), (const RolloutArgs*)rap);
            else hipLaunchKernelGGL(HIP_KERNEL_NAME(env_rollout_kernel<false>), ENV_GRID(e->n), ENV_BLOCK, LDS_BYTES, (hipStream_t)stream, e->st, e->ist, e->wk, e->n, make_cfg(*e), (const RolloutArgs*)rap);
            APX_LAUNCH_CHECK();
            if (timed) { APX_HIP(hipEventRecord((hipEvent_t)e->ev[e->ev_n + 1], (hipStream_t)stream)); e->ev_n += 2; }
            e->roll_launches += 1;
            return APX_OK;
        }
    }
    APX_REQUIRE(!td3, "apx_rollout_td3: the one-launch path only (hidden width 256, observation width <= 64, no graph capture, APX_ROLLOUT_STEPWISE unset)");
    for (int t = 0; t < T; ++t) {
        float* obs = obs_grid + (size_t)t * N * D; float* mu = mu_grid + (size_t)t * N * A; float* act = act_grid + (size_t)t * N * A;
        const float* nz = noise ? noise + (size_t)t * N * A : nullptr;
        int rc = apx_mlp_forward_act(actor, D, H, A, obs, N, obs_mean, obs_std, mu, act, nz, sigma, stream);      // forward + noise in one launch (the reference's 2 x 256 shape)
        if (rc < 0) return APX_E_HIP;
        if (rc == 0) {
            rc = apx_mlp_forward(actor, D, H, A, obs, N, nullptr, nullptr, 0, obs_mean, obs_std, nullptr, nullptr, nullptr, mu, stream);
            if (rc != APX_OK) return rc;
            hipLaunchKernelGGL(act_noise_kernel, dim3(apx_cdiv(N * A, 256)), dim3(256), 0, (hipStream_t)stream, mu, nz, sigma, N * A, act);
            APX_LAUNCH_CHECK();
        }
        rc = apx_env_step(e, act, t + 1 < T ? obs_grid + (size_t)(t + 1) * N * D : obs_next, rew_grid + (size_t)t * N, done_grid + (size_t)t * N,
                          fin_grid + (size_t)t * N * D, 1, stream);
        if (rc != APX_OK) return rc;
    }
    return APX_OK;
}

extern "C" int apx_rollout(apx_env_t* e, const float* actor, int H, const float* obs_mean, const float* obs_std, float sigma, const float* noise, int T,
                           float* obs_grid, float* act_grid, float* mu_grid, float* rew_grid, uint8_t* done_grid, float* fin_grid, float* obs_next, void* stream) {
    return rollout_ff(e, 0, 1.f, 0, actor, H, obs_mean, obs_std, sigma, noise, T, obs_grid, act_grid, mu_grid, rew_grid, done_grid, fin_grid, obs_next, stream);
}
extern "C" int apx_rollout_td3(apx_env_t* e, const float* actor, int H, float max_action, float act_noise, const float* noise, int noise_per_dim, int T,
                               float* obs_grid, float* act_grid, float* mu_grid, float* rew_grid, uint8_t* done_grid, float* fin_grid, float* obs_next, void* stream) {
    return rollout_ff(e, 1, max_action, noise_per_dim ? 0 : 1, actor, H, nullptr, nullptr, act_noise, noise, T, obs_grid, act_grid, mu_grid, rew_grid, done_grid, fin_grid, obs_next, stream);
}
// apx_rollout with the recurrent actor: the whole T-step rollout as ONE env_rollout_kernel launch, the two LSTM cells and the head evaluated per wave inside it
extern "C" int apx_rollout_lstm(apx_env_t* e, const float* actor, int H, int L, const float* obs_mean, const float* obs_std, float sigma, const float* noise, int T,
                                float* obs_grid, float* act_grid, float* mu_grid, float* rew_grid, uint8_t* done_grid, float* fin_grid, float* obs_next, void* stream) {
    APX_REQUIRE(e && actor && obs_grid && act_grid && mu_grid && rew_grid && done_grid && fin_grid && obs_next && T > 0, "rollout arguments");
    const int D = make_cfg(*e).obs_dim, A = APX_ACT_DIM;
    APX_REQUIRE(H == 128 && L == 2 && D <= 64, "apx_rollout_lstm: 2 x LSTMCell(128), observation width <= 64");
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    APX_REQUIRE(!(hipStreamIsCapturing((hipStream_t)stream, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone), "apx_rollout_lstm under graph capture");
    const long G = 4 * H, n4 = (e->n + 3) / 4 * 4;
    const long nw = (long)D * G + 3 * (long)H * G + (long)H * A + n4 * 512 + 1;      // (+ 1: never the size of the feed-forward actor's block)
    if (e->pol_wt_n != nw) { (void)hipFree(e->pol_wt); e->pol_wt = nullptr; APX_HIP(hipMalloc(&e->pol_wt, sizeof(float) * nw + sizeof(RolloutArgs) + 16)); e->pol_wt_n = nw; }
    { const int rc = refill_join(e, stream); if (rc != APX_OK) return rc; }
    e->refill_due = 0;
    // parameter block (include/apx.h apx_lstm_forward): per cell weight_ih [4H, in], weight_hh [4H, H], bias_ih, bias_hh; then the head
    const float* Wi0 = actor; const float* Wh0 = Wi0 + G * D; const float* bi0 = Wh0 + G * H; const float* bh0 = bi0 + G;
    const float* Wi1 = bh0 + G; const float* Wh1 = Wi1 + G * H; const float* bi1 = Wh1 + G * H; const float* bh1 = bi1 + G;
    const float* W2 = bh1 + G; const float* b2 = W2 + (long)A * H;
    float* Gx0 = e->pol_wt; float* Gh0 = Gx0 + (long)D * G; float* Gx1 = Gh0 + (long)H * G; float* Gh1 = Gx1 + (long)H * G; float* Wt2 = Gh1 + (long)H * G; float* hc = Wt2 + (long)H * A;
    hipLaunchKernelGGL(transpose_kernel, dim3(apx_cdiv(G * D, 256)), dim3(256), 0, (hipStream_t)stream, Wi0, Gx0, (int)G, D);
    hipLaunchKernelGGL(transpose_kernel, dim3(apx_cdiv(G * H, 256)), dim3(256), 0, (hipStream_t)stream, Wh0, Gh0, (int)G, H);
    hipLaunchKernelGGL(transpose_kernel, dim3(apx_cdiv(G * H, 256)), dim3(256), 0, (hipStream_t)stream, Wi1, Gx1, (int)G, H);
    hipLaunchKernelGGL(transpose_kernel, dim3(apx_cdiv(G * H, 256)), dim3(256), 0, (hipStream_t)stream, Wh1, Gh1, (int)G, H);
    hipLaunchKernelGGL(transpose_kernel, dim3(apx_cdiv((long)A * H, 256)), dim3(256), 0, (hipStream_t)stream, W2, Wt2, A, H);
    APX_LAUNCH_CHECK();
    RolloutArgs ra{nullptr, nullptr, nullptr, nullptr, Wt2, b2, obs_mean, obs_std, sigma, noise, H, T, obs_grid, act_grid, mu_grid, rew_grid, done_grid, fin_grid, obs_next, e->rst, e->rst_int, make_cfg(*e)};
    ra.lstm = 1; ra.Gx0 = Gx0; ra.Gh0 = Gh0; ra.bi0 = bi0; ra.bh0 = bh0; ra.Gx1 = Gx1; ra.Gh1 = Gh1; ra.bi1 = bi1; ra.bh1 = bh1; ra.hc = hc;
    RolloutArgs* rap = (RolloutArgs*)(((uintptr_t)(e->pol_wt + nw) + 15) & ~(uintptr_t)15);
    APX_HIP(hipMemcpyAsync(rap, &ra, sizeof(ra), hipMemcpyHostToDevice, (hipStream_t)stream));
    const bool timed = e->timing && e->ev_n + 2 <= e->ev_cap;
    if (timed) APX_HIP(hipEventRecord((hipEvent_t)e->ev[e->ev_n], (hipStream_t)stream));
    if (e->hf) hipLaunchKernelGGL(HIP_KERNEL_NAME(env_rollout_kernel<true, 1>), ENV_GRID(e->n), ENV_BLOCK, LDS_BYTES, (hipStream_t)stream, e->st, e->ist, e->wk, e->n, make_cfg(*e), (const RolloutArgs*)rap);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(env_rollout_kernel<false, 1>), ENV_GRID(e->n), ENV_BLOCK, LDS_BYTES, (hipStream_t)stream, e->st, e->ist, e->wk, e->n, make_cfg(*e), (const RolloutArgs*)rap);
    APX_LAUNCH_CHECK();
    if (timed) { APX_HIP(hipEventRecord((hipEvent_t)e->ev[e->ev_n + 1], (hipStream_t)stream)); e->ev_n += 2; }
    e->roll_launches += 1;
    return APX_OK;
}

// [F, n] SoA  <->  [n, cnt] row-major
__global__ void gather_kernel(const float* st, int n, int f0, int cnt, float* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * cnt) return;
    const int env = i / cnt, k = i - env * cnt;
    out[i] = st[(size_t)(f0 + k) * n + env];
}
__global__ void scatter_kernel(float* st, int n, int f0, int cnt, const float* in) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * cnt) return;
    const int env = i / cnt, k = i - env * cnt;
    st[(size_t)(f0 + k) * n + env] = in[i];
}
__global__ void gather_int_kernel(const int* ist, int n, int cnt, float* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * cnt) return;
    const int env = i / cnt, k = i - env * cnt;
    out[i] = (float)ist[(size_t)k * n + env];
}
__global__ void scatter_int_kernel(int* ist, int n, int cnt, const float* in) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * cnt) return;
    const int env = i / cnt, k = i - env * cnt;
    ist[(size_t)k * n + env] = (int)in[i];
}

struct FieldDesc { const char* name; int f0, cnt; };
static const FieldDesc kFields[] = {
    {"qpos", F_QPOS, NQ}, {"qvel", F_QVEL, NV}, {"qacc_warm", F_QACCW, NV}, {"mass", F_MASS, NB}, {"damping", F_DAMP, NV},
    {"friction", F_FRIC, 1}, {"floor", F_FLOOR, 9}, {"body_invweight0", F_BIW, NB}, {"dof_invweight0", F_DIW, NV},
    {"motor_noise", F_MNOISE, 10}, {"joint_noise", F_JNOISE, 6}, {"pd_target", F_PDT, 10}, {"tq_fifo", F_FIFO, 60},
    {"so_mpos", F_SO + SO_MPOS, 10}, {"so_mvel", F_SO + SO_MVEL, 10}, {"so_torque", F_SO + SO_TORQUE, 10},
    {"so_jpos", F_SO + SO_JPOS, 6}, {"so_jvel", F_SO + SO_JVEL, 6}, {"so_quat", F_SO + SO_QUAT, 4},
    {"so_rotvel", F_SO + SO_ROTVEL, 3}, {"so_tvel", F_SO + SO_TVEL, 3}, {"so_tacc", F_SO + SO_TACC, 3}, {"so_height", F_SO + SO_HEIGHT, 1},
    {"foot_vel", F_FOOTVEL, 6}, {"prev_action", F_PREVACT, 10}, {"prev_torque", F_PREVTQ, 10}, {"cmd", F_CMD, 7}, {"fwd", F_FWD, 16}, {"xfrc", F_XFRC, 6},
    {"menc", F_MENC, 90}, {"jenc_x", F_JENCX, 24}, {"jenc_y", F_JENCY, 12}, {"snap", F_SNAP, 26}, {"foot_prev", F_FOOTPREV, 6},      // (teacher-forced parity tests copy the oracle's whole state in)
};

static const FieldDesc* find_field(const char* name) {
    for (const auto& f : kFields)
        if (!strcmp(f.name, name)) return &f;
    return nullptr;
}

extern "C" int apx_env_get_field(apx_env_t* e, const char* name, float* out, void* stream) {
    APX_REQUIRE(e && name && out, "null pointer");
    if (!strcmp(name, "ints")) {
        hipLaunchKernelGGL(gather_int_kernel, dim3(apx_cdiv((long)e->n * I_TOTAL, 256)), dim3(256), 0, (hipStream_t)stream, e->ist, e->n, (int)I_TOTAL, out);
        APX_LAUNCH_CHECK();
        return I_TOTAL;
    }
    if (!strcmp(name, "ints_bits")) {      // the same words bit for bit (an int32 view of `out`): counters above 2^24 do not survive the float conversion
        hipLaunchKernelGGL(gather_kernel, dim3(apx_cdiv((long)e->n * I_TOTAL, 256)), dim3(256), 0, (hipStream_t)stream, (const float*)e->ist, e->n, 0, (int)I_TOTAL, out);
        APX_LAUNCH_CHECK();
        return I_TOTAL;
    }
#ifdef APX_PROF
    if (!strcmp(name, "prof")) {   // 48 cumulative phase cycle counters, then reset
        unsigned long long h[48];
        APX_HIP(hipMemcpyFromSymbol(h, HIP_SYMBOL(c4::g_prof_acc), sizeof(h)));
        float hf[48];
        for (int i = 0; i < 48; ++i) hf[i] = (float)h[i];
        APX_HIP(hipMemcpy(out, hf, sizeof(hf), hipMemcpyHostToDevice));
        unsigned long long z[48] = {0};
        APX_HIP(hipMemcpyToSymbol(HIP_SYMBOL(c4::g_prof_acc), z, sizeof(z)));
        return 48;
    }
#endif
#ifdef APX_WAVETIME
    if (!strcmp(name, "wavetime")) {      // shader-clock cycles of each wave (= workgroup, 4 consecutive envs) of the most recent env_step_kernel launch, one value per env row
        static unsigned long long h[4096];
        APX_HIP(hipDeviceSynchronize());
        APX_HIP(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_wavetime), sizeof(h)));
        static unsigned f[4096 * 4];
        APX_HIP(hipMemcpyFromSymbol(f, HIP_SYMBOL(c4::g_wavefeat), sizeof(f)));
        static float hf[4096 * 4 * 4];
        const int nw = e->n / 4 < 4096 ? e->n / 4 : 4096;      // out rows = waves: [cycles, substeps with leg-leg rows, with a limit row, sum of active contact slots]
        for (int i = 0; i < nw; ++i) { hf[4 * i] = (float)h[i]; hf[4 * i + 1] = (float)f[4 * i]; hf[4 * i + 2] = (float)f[4 * i + 1]; hf[4 * i + 3] = (float)f[4 * i + 2]; }
        APX_HIP(hipMemcpy(out, hf, sizeof(float) * (size_t)nw * 4, hipMemcpyHostToDevice));
        static unsigned z[4096 * 4];
        APX_HIP(hipMemcpyToSymbol(HIP_SYMBOL(c4::g_wavefeat), z, sizeof(z)));
        return 4;
    }
#endif
    if (!strcmp(name, "reset_miss")) {      // resets that found no prepared image (must stay 0); then cleared
        int h = 0;
        APX_HIP(hipMemcpyFromSymbol(&h, HIP_SYMBOL(g_reset_miss), sizeof(h)));
        const float hf = (float)h;
        APX_HIP(hipMemcpy(out, &hf, sizeof(hf), hipMemcpyHostToDevice));
        const int z = 0;
        APX_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_reset_miss), &z, sizeof(z)));
        return 1;
    }
#ifdef APX_CHECK
    if (!strcmp(name, "oob")) {       // first out-of-range S / S.W / S.I index of the checked build: kind (1 state, 2 workspace, 3 int), index, env, lane; then cleared
        int h[4];
        APX_HIP(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_oob), sizeof(h)));
        float hf[4] = {(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
        APX_HIP(hipMemcpy(out, hf, sizeof(hf), hipMemcpyHostToDevice));
        const int z[4] = {0, 0, 0, 0};
        APX_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_oob), z, sizeof(z)));
        return 4;
    }
#endif
    if (!strcmp(name, "substep")) {   // debugging hook: out[0] (host-readable count is not needed) - run one raw substep
        if (e->hf) hipLaunchKernelGGL(HIP_KERNEL_NAME(env_substep_kernel<true>), ENV_GRID(e->n), ENV_BLOCK, LDS_BYTES, (hipStream_t)stream, e->st, e->ist, e->wk, e->n, make_cfg(*e), 1,
                           (const float*)nullptr, (float*)nullptr);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(env_substep_kernel<false>), ENV_GRID(e->n), ENV_BLOCK, LDS_BYTES, (hipStream_t)stream, e->st, e->ist, e->wk, e->n, make_cfg(*e), 1,
                           (const float*)nullptr, (float*)nullptr);
        APX_LAUNCH_CHECK();
        return 0;
    }
    if (!strcmp(name, "est")) {       // the state-estimator records, [n][est::REC] as stored (estimator_lane.h)
        APX_HIP(hipMemcpyAsync(out, e->wk, sizeof(float) * (size_t)est::REC * e->n, hipMemcpyDeviceToDevice, (hipStream_t)stream));
        return est::REC;
    }
    const FieldDesc* f = find_field(name);
    APX_REQUIRE(f, "unknown field");
    hipLaunchKernelGGL(gather_kernel, dim3(apx_cdiv((long)e->n * f->cnt, 256)), dim3(256), 0, (hipStream_t)stream, e->st, e->n, f->f0, f->cnt, out);
    APX_LAUNCH_CHECK();
    return f->cnt;
}

extern "C" int apx_env_set_field(apx_env_t* e, const char* name, const float* in, void* stream) {
    APX_REQUIRE(e && name, "null pointer");
    { const int rc = invalidate_prepared(e, stream); if (rc != APX_OK) return rc; }
    if (!strcmp(name, "set_const")) {   // recompute invweight0 after mass edits (sim.set_const)
        hipLaunchKernelGGL(env_setconst_kernel, SETCONST_GRID(e->n), dim3(64), SETCONST_LDS, (hipStream_t)stream, e->st, e->ist, e->wk, e->n, make_cfg(*e));
        APX_LAUNCH_CHECK();
        return 0;
    }
    APX_REQUIRE(in, "null pointer");
    if (!strcmp(name, "ints")) {
        hipLaunchKernelGGL(scatter_int_kernel, dim3(apx_cdiv((long)e->n * I_TOTAL, 256)), dim3(256), 0, (hipStream_t)stream, e->ist, e->n, (int)I_TOTAL, in);
        APX_LAUNCH_CHECK();
        return I_TOTAL;
    }
    if (!strcmp(name, "est")) {
        APX_HIP(hipMemcpyAsync(e->wk, in, sizeof(float) * (size_t)est::REC * e->n, hipMemcpyDeviceToDevice, (hipStream_t)stream));
        return est::REC;
    }
    const FieldDesc* f = find_field(name);
    APX_REQUIRE(f, "unknown field");
    hipLaunchKernelGGL(scatter_kernel, dim3(apx_cdiv((long)e->n * f->cnt, 256)), dim3(256), 0, (hipStream_t)stream, e->st, e->n, f->f0, f->cnt, in);
    APX_LAUNCH_CHECK();
    return f->cnt;
}

extern "C" int apx_env_get_state(apx_env_t* e, float* qpos, float* qvel, void* stream) {
    APX_REQUIRE(e && qpos && qvel, "null pointer");
    int rc = apx_env_get_field(e, "qpos", qpos, stream);
    if (rc < 0) return rc;
    rc = apx_env_get_field(e, "qvel", qvel, stream);
    return rc < 0 ? rc : APX_OK;
}
extern "C" int apx_env_set_state(apx_env_t* e, const float* qpos, const float* qvel, void* stream) {
    APX_REQUIRE(e && qpos && qvel, "null pointer");
    int rc = apx_env_set_field(e, "qpos", qpos, stream);
    if (rc < 0) return rc;
    rc = apx_env_set_field(e, "qvel", qvel, stream);
    return rc < 0 ? rc : APX_OK;
}
