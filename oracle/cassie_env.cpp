// CPU ORACLE (test infrastructure only).  See cassie_env.h.
#include "cassie_env.h"
#include "cassie_traj_gen.h"
#include <cstdio>

namespace orc {

static const double PI = 3.14159265358979323846;
static const double kTorqueLimit[5] = {140.63, 140.63, 216.16, 216.16, 45.14};   // cassie_sim_init presets, SURVEY §2.2
// ---------------------------------------------------------------------------------------------- Philox4x32-10
static inline void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}
uint32_t Philox::next_u32() {
    uint32_t c[4] = {ctr, env, 0x41505845u, dom};   // (draw index, env id, tag, stream domain)
    uint32_t k0 = key0, k1 = key1;
    for (int r = 0; r < 10; ++r) {
        philox_round(c, k0, k1);
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    ++ctr;
    return c[0];
}

// ---------------------------------------------------------------------------------------------- clock splines
// cassie/phase_function.py:5-136.  Every knot of the PCHIP interpolants is adjacent to a flat segment, so every
// PCHIP derivative is 0 (scipy: d_k = 0 when a neighbouring secant is 0) and the interpolant is: constant on the
// flat pieces, cubic smoothstep y1 + (y2-y1)(3t^2 - 2t^3) between them, periodic with period phaselen.
void make_clock(Clock& c, double swing, double stance, double relax, int mode, int inc, int freq) {
    const double total = 2 * swing + 2 * stance;
    c.phaselen = total * freq;
    const double seg[4][2] = {{0.0, swing}, {swing, swing + stance}, {swing + stance, 2 * swing + stance},
                              {2 * swing + stance, total}};
    for (int s = 0; s < 4; ++s) {
        const double a = seg[s][0] * freq, b = seg[s][1] * freq, off = (b - a) * relax;
        c.x[2 * s] = a + off; c.x[2 * s + 1] = b - off;
    }
    // value tables; names follow the reference's arrays: r_frc, r_vel, l_frc, l_vel (phase_function.py:16-19)
    double r_frc[4], r_vel[4], l_frc[4], l_vel[4];
    // right swing (:27-32)
    l_vel[0] = r_frc[0] = -1; l_frc[0] = r_vel[0] = inc ? 1 : 0;
    // double stances (:39-57, :78-96)
    for (int s = 1; s <= 3; s += 2) {
        if (mode == 2) {            // aerial
            l_frc[s] = r_frc[s] = -1; l_vel[s] = r_vel[s] = inc ? 1 : 0;
        } else if (mode == 0) {     // zero
            l_frc[s] = r_frc[s] = 0; l_vel[s] = r_vel[s] = 0;
        } else {                    // grounded
            if (!inc) {
                if (s == 1) {       // quirk :54-55: second assignment overwrites l_frc, l_vel keeps its initial 0
                    r_frc[s] = 0; l_frc[s] = -1; r_vel[s] = -1; l_vel[s] = 0;
                } else {            // :93-94 assign l_vel correctly
                    l_frc[s] = r_frc[s] = 0; l_vel[s] = r_vel[s] = -1;
                }
            } else {
                l_frc[s] = r_frc[s] = 1; l_vel[s] = r_vel[s] = -1;
            }
        }
    }
    // left swing (:64-70)
    l_vel[2] = r_frc[2] = inc ? 1 : 0; l_frc[2] = r_vel[2] = -1;
    for (int s = 0; s < 4; ++s)
        for (int k = 0; k < 2; ++k) {
            // create_phase_reward returns ([r_frc, r_vel], [l_frc, l_vel], ...) and cassie.py:559 binds them to
            // (left_clock, right_clock): the "left" clock is the r_* pair.
            c.y[0][2 * s + k] = r_frc[s]; c.y[1][2 * s + k] = r_vel[s];
            c.y[2][2 * s + k] = l_frc[s]; c.y[3][2 * s + k] = l_vel[s];
        }
}

double Clock::eval(int which, double ph) const {
    // knots of the previous / next cycle are the same knots shifted by -/+ phaselen
    const double* yy = y[which];
    double xk[10], yk[10];
    xk[0] = x[7] - phaselen; yk[0] = yy[7];
    for (int i = 0; i < 8; ++i) { xk[i + 1] = x[i]; yk[i + 1] = yy[i]; }
    xk[9] = x[0] + phaselen; yk[9] = yy[0];
    // the reference evaluates at integer phase in [0, phaselen] (+1 before the wrap), inside [xk[0], xk[9]] for
    // relax >= 0; beyond the last knot of the 3-cycle table it would extrapolate, which never happens here.
    for (int i = 0; i < 9; ++i)
        if (ph <= xk[i + 1]) {
            if (ph < xk[i]) return yk[i];
            const double t = (ph - xk[i]) / (xk[i + 1] - xk[i]);
            return yk[i] + (yk[i + 1] - yk[i]) * (3 * t * t - 2 * t * t * t);
        }
    // ph beyond x[0] + phaselen: next-cycle pieces
    return eval(which, ph - phaselen);
}

// ---------------------------------------------------------------------------------------------- helpers

static const double kP[5] = {100, 100, 88, 96, 50}, kD[5] = {10.0, 10.0, 8.0, 9.6, 5.0};   // cassie.py:57-58
static const double kOffset[10] = {0.0045, 0.0, 0.4973, -1.1997, -1.5968, 0.0045, 0.0, 0.4973, -1.1997, -1.5968};  // :107
static const double kNeutralFoot[4] = {-0.24790886454547323, -0.24679713195445646, -0.6609396704367185, 0.663921021343526};  // :121
static const int kFir[9] = {2727, 534, -2658, -795, 72, 110, 19, -6, -3};

// cassie_core_sim_step (include/CassieCoreSim.h), characterised through the reference's callable binary
// (tools/refprobe/probe_safety*.py) and pinned by golden G10 to 1e-13:
//   * each drive has a soft zone that starts 0.15 rad inside its limits [-15,20] [-22,22] [-50,80] [-156,-42] [-140,-35]
//     deg (roll / yaw mirrored on the right leg); intrusion depth d_i
//   * every commanded torque on both legs is scaled by  s = prod_i max(0, 1 - d_i / 0.15)
//   * the intruding drive additionally gets  -+ Kp d (1 + d/0.15) - min(1, d/0.15) Kd qd   (Kp 1000 800 1200 1200 100,
//     Kd 12 12 36 36 7), then everything is clamped to the drive torqueLimit; radio channel 8 <= 0 zeroes all torques.
// Not modelled: the coupled hip-pitch + knee zone (pitch + knee < ~-2.36 rad, a deep squat).
void core_safety(const double* q, const double* qd, const double* cmd, double radio, double* out) {
    static const double lo_deg[5] = {-15, -22, -50, -156, -140}, hi_deg[5] = {20, 22, 80, -42, -35};
    static const double Kp[5] = {1000, 800, 1200, 1200, 100}, Kd[5] = {12, 12, 36, 36, 7};
    double d[10], sg[10], s = 1.0;
    for (int u = 0; u < 10; ++u) {
        const int j = u % 5;
        double lo = lo_deg[j] * PI / 180 + 0.15, hi = hi_deg[j] * PI / 180 - 0.15;
        if (u >= 5 && j < 2) { const double t = lo; lo = -hi; hi = -t; }
        d[u] = std::max(0.0, std::max(q[u] - hi, lo - q[u]));
        sg[u] = q[u] > hi ? -1.0 : 1.0;
        s *= std::max(0.0, 1.0 - d[u] / 0.15);
    }
    // coupled zone (golden G10b): hip pitch + knee below -135 deg behaves like one more zone whose restoring spring-damper acts
    // on BOTH drives (each with its own velocity) and whose depth enters the global scale
    double dc[2];
    for (int leg = 0; leg < 2; ++leg) {
        dc[leg] = std::max(0.0, -0.75 * PI - (q[5 * leg + 2] + q[5 * leg + 3]));
        s *= std::max(0.0, 1.0 - dc[leg] / 0.15);
    }
    for (int u = 0; u < 10; ++u) {
        const int j = u % 5;
        double t = s * cmd[u] + sg[u] * Kp[j] * d[u] * (1.0 + d[u] / 0.15) - std::min(1.0, d[u] / 0.15) * Kd[j] * qd[u];
        if (j == 2 || j == 3) { const double c = dc[u / 5]; t += Kp[j] * c * (1.0 + c / 0.15) - std::min(1.0, c / 0.15) * Kd[j] * qd[u]; }
        t = std::min(std::max(t, -kTorqueLimit[j]), kTorqueLimit[j]);
        out[u] = radio > 0 ? t : 0.0;
    }
}

static void forward_snapshot(Env& e, Work& w, const double* ctrl) {
    forward(e.par, e.st, w, ctrl);
    e.iter_sum += e.st.solver_iter; ++e.iter_passes; ++e.iter_hist[std::min(std::max(e.st.solver_iter, 0), 50)];
    e.sat_acc |= e.st.sat;                        // constraint sets beyond the HIP kernel's per-leg caps, accumulated over the env's life
    const State& s = e.st;
    for (int u = 0; u < 10; ++u) { e.snap_mpos[u] = s.qpos[cm_act_qposadr[u]]; e.snap_mvel[u] = s.qvel[cm_act_dof[u]]; }
    for (int k = 0; k < 6; ++k) { e.snap_jpos[k] = s.qpos[cm_jsens_qposadr[k]]; e.snap_jvel[k] = s.qvel[cm_jsens_dofadr[k]]; }
    for (int k = 0; k < 4; ++k) e.snap_quat[k] = s.qpos[3 + k];
    for (int k = 0; k < 3; ++k) { e.snap_gyro[k] = s.sens_gyro[k]; e.snap_acc[k] = s.sens_acc[k]; }
}

static void foot_positions(const State& s, double* fp) {   // cassie_sim_foot_positions
    fp[0] = s.xpos[13].x; fp[1] = s.xpos[13].y; fp[2] = s.xpos[13].z - 0.0550841220316708;
    fp[3] = s.xpos[25].x; fp[4] = s.xpos[25].y; fp[5] = s.xpos[25].z - 0.0550841220316708;
}

// One 2 kHz substep: encoders + estimator -> PD -> safeties -> motor model/delay -> mj_step
void sim_step_pd(Env& e) {
    static thread_local Work w;
    // --- drive encoders (10): truncating quantiser + 9-tap FIR velocity
    for (int u = 0; u < 10; ++u) {
        const double scale = 2 * PI / (double)(1 << cm_act_bits[u]);
        const double n = std::trunc(e.snap_mpos[u] * cm_act_gear[u] / scale);
        if (!e.menc_primed) for (int k = 0; k < 9; ++k) e.menc_hist[u][k] = n;
        for (int k = 8; k > 0; --k) e.menc_hist[u][k] = e.menc_hist[u][k - 1];
        e.menc_hist[u][0] = n;
        double acc = 0;
        for (int k = 0; k < 9; ++k) acc += kFir[k] * e.menc_hist[u][k];
        e.so_mpos[u] = n * scale / cm_act_gear[u];
        e.so_mvel[u] = acc * scale / cm_act_gear[u] / PI;
    }
    e.menc_primed = 1;
    // --- joint encoders (6): quantiser + biquad velocity
    for (int k = 0; k < 6; ++k) {
        const double scale = 2 * PI / (double)(1 << cm_jsens_bits[k]);
        const double x = std::trunc(e.snap_jpos[k] / scale) * scale;
        if (!e.jenc_primed) { for (int i = 0; i < 4; ++i) e.jenc_x[k][i] = x; for (int i = 0; i < 3; ++i) e.jenc_y[k][i] = 0; }
        for (int i = 3; i > 0; --i) e.jenc_x[k][i] = e.jenc_x[k][i - 1];
        e.jenc_x[k][0] = x;
        const double y = 12.348 * (e.jenc_x[k][0] + e.jenc_x[k][1] - e.jenc_x[k][2] - e.jenc_x[k][3]) + 1.7658 * e.jenc_y[k][0] - 0.79045 * e.jenc_y[k][1];
        e.jenc_y[k][1] = e.jenc_y[k][0]; e.jenc_y[k][0] = y;
        e.so_jpos[k] = x; e.so_jvel[k] = y;
    }
    e.jenc_primed = 1;
    // --- state estimator (state_output_step): 39 pass-through fields + the 7 filtered ones from the restated filter bank (cassie_estimator.h)
    for (int k = 0; k < 4; ++k) e.so_quat[k] = e.snap_quat[k];
    for (int k = 0; k < 3; ++k) e.so_rotvel[k] = e.snap_gyro[k];
    {
        EstSensors in;
        for (int u = 0; u < 10; ++u) in.mpos[u] = e.so_mpos[u];
        for (int k = 0; k < 6; ++k) in.jpos[k] = e.so_jpos[k];
        for (int k = 0; k < 4; ++k) in.quat[k] = e.snap_quat[k];
        for (int k = 0; k < 3; ++k) { in.gyro[k] = e.snap_gyro[k]; in.acc[k] = e.snap_acc[k]; }
        state_output_step(e.est, in);
        for (int k = 0; k < 3; ++k) { e.so_tvel[k] = e.est.vel[k]; e.so_tacc[k] = e.est.tacc[k]; }
        e.so_height = e.est.pos[2] - e.est.terrain;          // pelvis.position[2] - terrain.height, cassie.py:793
    }
    // --- pd_input_step: tau = P (pTarget - q) + D (dTarget - qd), no clamp (PdInput.h; SURVEY §2.2 bit-exact probe)
    double tau[10], ctrl[10];
    for (int u = 0; u < 10; ++u) tau[u] = e.pd_P[u] * (e.pd_target[u] - e.so_mpos[u]) + e.pd_D[u] * (0.0 - e.so_mvel[u]);
    // --- cassie_core_sim_step: joint-limit safety zones + clamp to the drive torqueLimit
    core_safety(e.so_mpos, e.so_mvel, tau, 1.0, tau);
    // --- cassie_sim_step_ethercat: torque-speed curve, 6-deep delay line
    for (int u = 0; u < 10; ++u) {
        const double wmax = cm_act_rpm[u] * 2 * PI / 60.0, tmax = cm_act_ctrlmax[u];
        const double om = std::fabs(e.st.qvel[cm_act_dof[u]] * cm_act_gear[u]);
        const double tlim = std::min(std::max(2 * tmax * (1 - om / wmax), 0.0), tmax);
        const double cmd = tau[u] / cm_act_gear[u];
        const double un = (cmd < 0 ? -1.0 : 1.0) * std::min(std::fabs(cmd), tlim);
        for (int k = 5; k > 0; --k) e.tq_fifo[u][k] = e.tq_fifo[u][k - 1];
        e.tq_fifo[u][0] = un;
        ctrl[u] = e.tq_fifo[u][5];
        e.so_torque[u] = cm_act_gear[u] * ctrl[u];
    }
    // --- mj_step1 + mj_step2
    forward_snapshot(e, w, ctrl);
    euler(e.par, e.st, w);
}

static inline double fphase(const Env& e) { return e.phase + 0.5 * e.phase_half; }
// self.phase += self.phase_add; wrap (cassie.py:447-453, 511-515) with phase_add 1 or 1.5
static void advance_phase(Env& e) {
    if (e.phase_add15) { e.phase += 1 + e.phase_half; e.phase_half ^= 1; } else e.phase += 1;
    if (fphase(e) > e.clock.phaselen) { e.phase = 0; e.phase_half = 0; e.counter += 1; }
}
static void quat_yaw_inverse_apply(double yaw, const double* v, int n, double* out) {
    // rotate_to_orient (cassie.py:280-291): q = euler2quat(z=yaw); iq = inverse(q)
    const double cz = std::cos(yaw / 2), sz = std::sin(yaw / 2);
    Q4 q = {cz, 0, 0, sz};
    if (q.w < 0) q = {-q.w, -q.x, -q.y, -q.z};
    const Q4 iq = {q.w, -q.x, -q.y, -q.z};
    if (n == 3) {   // rotate_by_quaternion(vec, iq) = iq * (0,v) * inverse(iq)
        const Q4 r = qmul(iq, qmul(Q4{0, v[0], v[1], v[2]}, q));
        out[0] = r.x; out[1] = r.y; out[2] = r.z;
    } else {
        Q4 r = qmul(iq, Q4{v[0], v[1], v[2], v[3]});
        if (r.w < 0) r = {-r.w, -r.x, -r.y, -r.z};
        out[0] = r.w; out[1] = r.x; out[2] = r.y; out[3] = r.z;
    }
}

// get_full_state, cassie.py:787-859 (input_profile full, command_profile clock)
void env_obs(const Env& e, double* o) {
    int n;
    if (e.cfg.input_profile == 1) {      // input_profile "min" (cassie.py:829-837): foot positions, pelvis orientation, rotational velocity, foot orientations
        for (int k = 0; k < 3; ++k) { o[k] = e.est.foot_rel[0][k]; o[3 + k] = e.est.foot_rel[1][k]; o[10 + k] = e.so_rotvel[k]; }
        quat_yaw_inverse_apply(e.orient_add, e.so_quat, 4, o + 6);
        for (int k = 0; k < 4; ++k) { o[13 + k] = e.est.foot_quat[0][k]; o[17 + k] = e.est.foot_quat[1][k]; }
        n = 21;
    } else {
        o[0] = e.so_height;
        quat_yaw_inverse_apply(e.orient_add, e.so_quat, 4, o + 1);
        for (int u = 0; u < 10; ++u) o[5 + u] = e.so_mpos[u] + e.motor_noise[u];
        quat_yaw_inverse_apply(e.orient_add, e.so_tvel, 3, o + 15);
        for (int k = 0; k < 3; ++k) o[18 + k] = e.so_rotvel[k];
        for (int u = 0; u < 10; ++u) o[21 + u] = e.so_mvel[u];
        quat_yaw_inverse_apply(e.orient_add, e.so_tacc, 3, o + 31);
        for (int k = 0; k < 6; ++k) o[34 + k] = e.so_jpos[k] + e.joint_noise[k];
        for (int k = 0; k < 6; ++k) o[40 + k] = e.so_jvel[k];
        n = 46;
    }
    o[n] = std::sin(2 * PI * fphase(e) / e.clock.phaselen);
    o[n + 1] = std::cos(2 * PI * fphase(e) / e.clock.phaselen);
    if (e.cfg.command_profile == 0) { o[n + 2] = e.speed; o[n + 3] = e.side_speed; return; }
    // command_profile "phase" (cassie.py:805-808): clock, swing / stance duration, encode_stance_mode (grounded, aerial, zero), speed, side speed
    o[n + 2] = e.swing_duration; o[n + 3] = e.stance_duration;
    o[n + 4] = e.cfg.stance_mode == 1; o[n + 5] = e.cfg.stance_mode == 2; o[n + 6] = e.cfg.stance_mode == 0;
    o[n + 7] = e.speed; o[n + 8] = e.side_speed;
}

void env_init(Env& e, const EnvCfg& cfg, uint32_t env_id) {
    std::memset(&e, 0, sizeof(e));
    e.cfg = cfg;
    state_output_setup(e.est);
    default_params(e.par);
    e.par.pgs_iters = cfg.pgs_iters;
    e.rng = Philox{(uint32_t)cfg.seed, (uint32_t)(cfg.seed >> 32), env_id, 0, 0};
    reset_state(e.st);
}

static void set_clock_from_speed(Env& e) {   // cassie.py:556-559
    const double total = (0.9 - 0.25 / 3.0 * std::fabs(e.speed)) / 2;
    const double swing = (0.30 + ((0.70 - 0.30) / 3) * std::fabs(e.speed)) * total;
    const double stance = (0.70 - ((0.70 - 0.30) / 3) * std::fabs(e.speed)) * total;
    e.swing_duration = swing; e.stance_duration = stance;
    make_clock(e.clock, swing, stance, 0.1, e.cfg.stance_mode, e.cfg.have_incentive, 2000 / e.cfg.simrate);
}

// CassieEnv.step_basic (cassie.py:498-521): simrate x step_sim_basic (PD targets = action + offset - encoder offsets, cassie.py:355-387),
// time / phase bookkeeping, NO reward, termination, trackers or command resampling
// multiplicative hash over the row-set signatures of one forward pass, like the kernel's I_ROWSET: constraint rows (rowsig[0]), leg-leg pairs | the state estimator's load switches
static void fold_rowset(Env& e) {
    const unsigned w[2] = {e.st.rowsig[0], e.st.rowsig[1] | (unsigned)e.est.sw << 16};
    for (int k = 0; k < 2; ++k) { e.rowset_hash = (e.rowset_hash ^ w[k]) * 0x9E3779B1u; e.rowset_hash ^= e.rowset_hash >> 15; }
}
void env_step_basic(Env& e, const double* action, double* obs) {
    static const double offset[10] = {0.0045, 0.0, 0.4973, -1.1997, -1.5968, 0.0045, 0.0, 0.4973, -1.1997, -1.5968};
    static const double P[5] = {100, 100, 88, 96, 50}, D[5] = {10.0, 10.0, 8.0, 9.6, 5.0};
    for (int u = 0; u < 10; ++u) {
        e.pd_target[u] = action[u] + offset[u] - (e.cfg.dynamics_randomization ? e.motor_noise[u] : 0.0);
        e.pd_P[u] = P[u % 5]; e.pd_D[u] = D[u % 5];
    }
    e.rowset_hash = 0;
    for (int i = 0; i < e.cfg.simrate; ++i) {
        sim_step_pd(e);
        fold_rowset(e);      // like env_step: the parity tests bin step_basic steps by row set too
    }
    e.time += 1; advance_phase(e);
    env_obs(e, obs);
}

void env_clock_from_speed(Env& e) { set_clock_from_speed(e); }

// CassieEnv.update_speed (cassie.py:757-775), clock command profile: clip the commands, rebuild the clock from the NEW speed
// (no abs() here, unlike reset) and rescale the phase to the new cycle length
void env_update_speed(Env& e, double new_speed, double new_side_speed) {
    e.speed = std::min(std::max(new_speed, -0.3), 4.0);
    e.side_speed = std::min(std::max(new_side_speed, -0.3), 0.3);
    if (e.cfg.command_profile != 0) return;      // cassie.py:756-761: the phase profile keeps its durations (set_up_phase_reward only resets flags), phase unchanged
    const double total = (0.9 - 0.25 / 3.0 * e.speed) / 2;
    const double swing = (0.30 + ((0.70 - 0.30) / 3) * e.speed) * total;
    const double stance = (0.70 - ((0.70 - 0.30) / 3) * e.speed) * total;
    const double old_phaselen = e.clock.phaselen;
    e.swing_duration = swing; e.stance_duration = stance;
    make_clock(e.clock, swing, stance, 0.1, e.cfg.stance_mode, e.cfg.have_incentive, 2000 / e.cfg.simrate);
    e.phase = (int)(e.clock.phaselen * fphase(e) / old_phaselen); e.phase_half = 0;
}

// CassieEnv.reset_for_test(full_reset=False) (cassie.py:682-742): evaluation start.  Counters and commands to zero, the fixed
// 0.15 / 0.25 grounded clock, ONE step_pd with the stale pd targets, THEN the default dynamics + set_const (which also puts the
// robot back into the init pose), flat floor, zero encoder offsets.  The returned observation is built from the state estimate
// of that one step_pd (self.cassie_state), i.e. from before set_const.
void env_reset_for_test(Env& e, double* obs, bool full_reset) {
    static thread_local Work w;
    e.phase = 0; e.time = 0; e.counter = 0; e.orient_add = 0; e.speed = 0; e.phase_half = 0; e.phase_add15 = 0;      // cassie.py:687 phase_add = 1
    e.cfg.stance_mode = 1;
    e.swing_duration = 0.15; e.stance_duration = 0.25;
    make_clock(e.clock, 0.15, 0.25, 0.1, e.cfg.stance_mode, e.cfg.have_incentive, 2000 / e.cfg.simrate);
    if (!full_reset) {
        e.l_foot_frc = e.r_foot_frc = e.l_foot_orient_cost = e.r_foot_orient_cost = 0;
        sim_step_pd(e);
    } else {
        // cassie_sim_full_reset as the shipped binary does it (disassembly): qpos <- the 35 init values, mju_zero of qvel, ctrl,
        // qfrc_applied, xfrc_applied, qacc; the 60-double torque delay line zeroed; state_output_setup.  qacc_warmstart, time, the
        // encoder filters and the PD block are left alone.  Then CassieEnv.reset_cassie_state (cassie.py:733-746).
        for (int i = 0; i < NQ; ++i) e.st.qpos[i] = cm_init_qpos[i];
        for (int i = 0; i < NV; ++i) e.st.qvel[i] = 0;
        for (int u = 0; u < 10; ++u) for (int k = 0; k < 6; ++k) e.tq_fifo[u][k] = 0;
        for (int k = 0; k < 6; ++k) e.st.xfrc[k] = 0;
        static const double mp[5] = {0.0045, 0, 0.4973, -1.1997, -1.5968}, jp[3] = {0, 1.4267, -1.5968};
        for (int u = 0; u < 10; ++u) { e.so_mpos[u] = mp[u % 5]; e.so_mvel[u] = 0; }
        for (int k = 0; k < 6; ++k) { e.so_jpos[k] = jp[k % 3]; e.so_jvel[k] = 0; }
        e.so_quat[0] = 1; e.so_quat[1] = e.so_quat[2] = e.so_quat[3] = 0;
        for (int k = 0; k < 3; ++k) e.so_rotvel[k] = e.so_tvel[k] = e.so_tacc[k] = 0;
        e.so_height = 1.01;
        state_output_setup(e.est); e.est_age = 0;
    }
    if (e.cfg.dynamics_randomization) {
        const int iters = e.par.pgs_iters;
        default_params(e.par);
        e.par.pgs_iters = iters;
        set_const(e.par);
        for (int i = 0; i < NQ; ++i) e.st.qpos[i] = cm_init_qpos[i];
        for (int i = 0; i < NV; ++i) { e.st.qvel[i] = 0; e.st.qacc_warm[i] = 0; }
        double zero[10] = {0};
        forward_snapshot(e, w, zero);
        for (int u = 0; u < 10; ++u) e.motor_noise[u] = 0;
        for (int k = 0; k < 6; ++k) e.joint_noise[k] = 0;
    }
    env_obs(e, obs);
}

// CassieTrajEnv.get_ref_state (cassie_traj.py:926-972) for the walking trajectory at simrate 50: the row is phase * simrate of the 2 kHz
// trajectory; a phase beyond the trajectory's own cycle (floor(len / simrate) - 1 = 32) is rescaled by phase / phaselen; x scales with
// the commanded speed and advances by one cycle length per completed cycle, the lateral target is 0
void traj_ref_state(double phase, double phaselen, double speed, int counter, double* qpos, double* qvel) {
    if (phase > phaselen) phase = 0;
    if (phase > std::floor(TRAJ_LEN / 50.0) - 1) phase = std::floor((phase / phaselen) * TRAJ_LEN / 50.0);
    const int row = (int)phase;                                  // int(phase * simrate) / simrate
    for (int i = 0; i < NQ; ++i) qpos[i] = traj_table[row][i];
    for (int i = 0; i < NV; ++i) qvel[i] = traj_table[row][35 + i];
    qpos[0] *= speed;
    qpos[0] += TRAJ_DX * counter * speed;
    qpos[1] = 0;
    qvel[0] *= speed;
}

// CassieEnv.reset, cassie.py:523-680 (env_kind 1: CassieTrajEnv.reset, cassie_traj.py:599-778)
void env_reset(Env& e, double* obs) {
    static thread_local Work w;
    // the reference builds a new CassieEnv (-> cassie_sim_init -> a new estimator) per PPO.sample call (rl/algos/ppo.py:152); a lock-step env outlives
    // the call, so the estimator's lifetime is carried: after est_lifetime env steps the next reset starts from state_output_setup
    if (e.cfg.est_lifetime > 0 && e.est_age >= e.cfg.est_lifetime) { state_output_setup(e.est); e.est_age = 0; }
    // reset draws come from their own counter-based stream keyed by (seed, env, episode index): order-independent of the per-step command draws, so the kernel can
    // prepare the next episode's randomised model ahead of time (apx_env_prepare_resets).  Draw ORDER inside a reset is the reference's (golden G14).
    e.episode += 1;
    Philox r{e.rng.key0, e.rng.key1, e.rng.env, (uint32_t)e.episode * 128u, 1};
    e.speed = e.cfg.env_kind == 1 ? (double)r.randint(41) / 10 : r.uniform(-0.3, 4.0);      // cassie_traj.py:608: random.randint(0, 40) / 10
    e.side_speed = r.uniform(-0.3, 0.3);
    if (e.cfg.command_profile == 0) set_clock_from_speed(e);
    else {      // command_profile "phase", cassie.py:529-545: swing / stance duration and stance mode drawn per episode
        double swing, stance;
        if (e.cfg.command_profile == 2) {                // "library" in the reward string (:531-539)
            e.speed = (double)r.randint(31) / 10;
            const double total = (double)(3 + r.randint(4)) / 10, ratio = (double)(2 + r.randint(7)) / 10;
            swing = total * ratio; stance = total - swing;
        } else { swing = (double)(1 + r.randint(50)) / 100; stance = (double)(1 + r.randint(30)) / 100; }
        const uint32_t pick = r.randint(3);              // np.random.choice(["grounded", "aerial", "zero"])
        e.cfg.stance_mode = pick == 0 ? 1 : pick == 1 ? 2 : 0;
        e.swing_duration = swing; e.stance_duration = stance;
        make_clock(e.clock, swing, stance, 0.1, e.cfg.stance_mode, e.cfg.have_incentive, 2000 / e.cfg.simrate);
    }
    e.phase_half = 0;
    e.phase = (int)r.randint((uint32_t)std::floor(e.clock.phaselen) + 1);   // random.randint(0, floor(phaselen)) inclusive
    e.time = 0; e.counter = 0;
    if (e.cfg.dynamics_randomization) {
        // damping (:569-597): hips, achilles, knee, shin, tarsus, foot-crank, foot x U[0.3,5]; pelvis, heel-spring, plantar fixed
        static const int vary[13] = {1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 1, 0, 1};
        for (int d = 0; d < 6; ++d) { (void)r.uniform01(); e.par.damping[d] = cm_dof_damping[d]; }
        for (int leg = 0; leg < 2; ++leg)
            for (int k = 0; k < 13; ++k) {
                const int d = 6 + 13 * leg + k;
                const double u = r.uniform01();
                const double lo = vary[k] ? 0.3 : 1.0, hi = vary[k] ? 5.0 : 1.0;
                e.par.damping[d] = std::max(0.0, cm_dof_damping[d] * (lo + (hi - lo) * u));
            }
        // mass (:599-622): every body x U[0.5,1.5]; world stays 0
        (void)r.uniform01(); e.par.mass[0] = 0;
        for (int b = 1; b < NB; ++b) e.par.mass[b] = std::max(0.0, cm_body_mass[b] * r.uniform(0.5, 1.5));
        // friction (:627-632): translational U[0.4,1.1]; torsional / rolling are drawn but unused with condim 3
        e.par.friction = r.uniform(0.4, 1.1); (void)r.uniform01(); (void)r.uniform01();
        // floor tilt (:644-648): euler2quat(z=0, y=pitch, x=roll)
        const double roll = r.uniform(-0.03, 0.03), pitch = r.uniform(-0.03, 0.03);
        const double cy = std::cos(pitch / 2), sy = std::sin(pitch / 2), cx = std::cos(roll / 2), sx = std::sin(roll / 2);
        Q4 fq = {cx * cy, cy * sx, cx * sy, sx * sy};     // quaternion_function.py:58-71 with z = 0
        if (fq.w < 0) fq = {-fq.w, -fq.x, -fq.y, -fq.z};
        e.par.floor_quat = fq;
        for (int u = 0; u < 10; ++u) e.motor_noise[u] = r.uniform(-0.01, 0.01);
        for (int k = 0; k < 6; ++k) e.joint_noise[k] = r.uniform(-0.01, 0.01);
        set_const(e.par);                                   // sim.set_const -> mj_setConst
    }
    // cassie_sim_set_const: init qpos, zero qvel, mj_forward; PD / delay line / encoder / estimator state is NOT reset
    for (int i = 0; i < NQ; ++i) e.st.qpos[i] = cm_init_qpos[i];
    for (int i = 0; i < NV; ++i) { e.st.qvel[i] = 0; e.st.qacc_warm[i] = 0; }
    double zero[10] = {0};
    forward_snapshot(e, w, zero);
    foot_positions(e.st, e.foot_pos_prev);
    if (e.cfg.env_kind == 1)     // cassie_traj.py:752-758: set_qpos / set_qvel with the reference state of the start phase; no mj_forward, so the
        traj_ref_state(e.phase, e.clock.phaselen, e.speed, 0, e.st.qpos, e.st.qvel);      // next step_pd still reads the init-pose sensors
    sim_step_pd(e);                                         // cassie.py:665: one step with the stale self.u
    foot_positions(e.st, e.foot_pos_prev);
    e.orient_add = 0;
    r.ctr = (uint32_t)e.episode * 128u + 126u;              // the two command redraws sit at the end of the episode's block (the number of draws before them depends on the command profile)
    e.speed = r.uniform(-0.3, 4.0);                         // cassie.py:669-670 (clock keeps the FIRST speed draw)
    e.side_speed = r.uniform(-0.3, 0.3);
    e.l_foot_frc = e.r_foot_frc = e.l_foot_orient_cost = e.r_foot_orient_cost = 0;
    if (obs) env_obs(e, obs);
}

static double clock_reward(Env& e, const double* action) {   // cassie/rewards/clock_rewards.py:6-110
    const State& s = e.st;
    const double fmax = 250, vmax = 2.0;
    const double nlf = std::min(e.l_foot_frc, fmax) / fmax, nrf = std::min(e.r_foot_frc, fmax) / fmax;
    const double lv = std::sqrt(e.l_foot_vel[0] * e.l_foot_vel[0] + e.l_foot_vel[1] * e.l_foot_vel[1] + e.l_foot_vel[2] * e.l_foot_vel[2]);
    const double rv = std::sqrt(e.r_foot_vel[0] * e.r_foot_vel[0] + e.r_foot_vel[1] * e.r_foot_vel[1] + e.r_foot_vel[2] * e.r_foot_vel[2]);
    const double nlv = std::min(lv, vmax) / vmax, nrv = std::min(rv, vmax) / vmax;
    const double com_orient = 10 * (1 - s.qpos[3] * s.qpos[3]);
    const double foot_orient = 10 * (e.l_foot_orient_cost + e.r_foot_orient_cost);
    const double com_vel_err = std::fabs(s.qvel[0] - e.speed);
    double straight = std::fabs(s.qpos[1]);
    if (straight < 0.05) straight = 0;
    double hdiff = std::fabs(s.qpos[2] - 0.9);
    if (hdiff < 0.05 + 0.05 * e.speed) hdiff = 0;
    double pacc = 0;
    for (int k = 0; k < 3; ++k) pacc += std::fabs(e.so_rotvel[k]) + std::fabs(e.so_tacc[k]);
    const double pelvis_motion = straight + hdiff + 0.25 * pacc;
    const double lfc = e.clock.eval(0, fphase(e)), lvc = e.clock.eval(1, fphase(e)), rfc = e.clock.eval(2, fphase(e)), rvc = e.clock.eval(3, fphase(e));
    const double frc_score = std::tan(PI / 4 * lfc * nlf) + std::tan(PI / 4 * rfc * nrf);
    const double vel_score = std::tan(PI / 4 * lvc * nlv) + std::tan(PI / 4 * rvc * nrv);
    const double hip_roll = std::fabs(s.qvel[6]) + std::fabs(s.qvel[13]);    // :74 indexes qvel[13] (left shin), sic
    double tq = 0, ac = 0;
    for (int u = 0; u < 10; ++u) { tq += std::fabs(e.prev_torque[u] - e.so_torque[u]); ac += std::fabs(e.prev_action[u] - action[u]); }
    const double torque_pen = 0.25 * tq / 10, action_pen = 5 * ac / 10;
    double* t = e.last_reward_terms;
    t[0] = 0.200 * frc_score; t[1] = 0.200 * vel_score; t[2] = 0.200 * std::exp(-(com_orient + foot_orient));
    t[3] = 0.150 * std::exp(-pelvis_motion); t[4] = 0.150 * std::exp(-com_vel_err); t[5] = 0.050 * std::exp(-hip_roll);
    t[6] = 0.025 * std::exp(-torque_pen); t[7] = 0.025 * std::exp(-action_pen);
    return t[0] + t[1] + t[2] + t[3] + t[4] + t[5] + t[6] + t[7];
}

// early_clock_reward (cassie/rewards/clock_rewards.py:119-223): tanh scores, caps 350 N / 3 m/s, no torque / action /
// hip-roll / pelvis-acceleration terms
static double early_clock_reward(Env& e, const double* action) {
    (void)action;
    const State& s = e.st;
    const double fmax = 350, vmax = 3.0;
    const double nlf = std::min(e.l_foot_frc, fmax) / fmax, nrf = std::min(e.r_foot_frc, fmax) / fmax;
    const double lv = std::sqrt(e.l_foot_vel[0] * e.l_foot_vel[0] + e.l_foot_vel[1] * e.l_foot_vel[1] + e.l_foot_vel[2] * e.l_foot_vel[2]);
    const double rv = std::sqrt(e.r_foot_vel[0] * e.r_foot_vel[0] + e.r_foot_vel[1] * e.r_foot_vel[1] + e.r_foot_vel[2] * e.r_foot_vel[2]);
    const double nlv = std::min(lv, vmax) / vmax, nrv = std::min(rv, vmax) / vmax;
    const double com_orient = 1 - s.qpos[3] * s.qpos[3], foot_orient = e.l_foot_orient_cost + e.r_foot_orient_cost;
    const double com_vel_err = std::fabs(e.speed - s.qvel[0]);
    double straight = std::fabs(s.qpos[1]);
    if (straight < 0.05) straight = 0;
    double hdiff = std::fabs(s.qpos[2] - 0.9);
    if (hdiff < 0.05 + 0.05 * e.speed) hdiff = 0;
    const double lfc = e.clock.eval(0, fphase(e)), lvc = e.clock.eval(1, fphase(e)), rfc = e.clock.eval(2, fphase(e)), rvc = e.clock.eval(3, fphase(e));
    const double frc_score = std::tanh(lfc * nlf) + std::tanh(rfc * nrf), vel_score = std::tanh(lvc * nlv) + std::tanh(rvc * nrv);
    return 0.250 * frc_score + 0.350 * vel_score + 0.200 * std::exp(-com_vel_err) + 0.100 * std::exp(-(com_orient + foot_orient)) +
           0.100 * std::exp(-(straight + hdiff));
}
// max_vel_clock_reward (cassie/rewards/clock_rewards.py:416-480): caps 400 N / 3 m/s, tanh clock terms, forward-velocity bonus
static double max_vel_clock_reward(Env& e, const double* action) {
    (void)action;
    const State& s = e.st;
    const double fmax = 400, vmax = 3.0;
    const double nlf = std::min(e.l_foot_frc, fmax) / fmax, nrf = std::min(e.r_foot_frc, fmax) / fmax;
    const double lv = std::sqrt(e.l_foot_vel[0] * e.l_foot_vel[0] + e.l_foot_vel[1] * e.l_foot_vel[1] + e.l_foot_vel[2] * e.l_foot_vel[2]);
    const double rv = std::sqrt(e.r_foot_vel[0] * e.r_foot_vel[0] + e.r_foot_vel[1] * e.r_foot_vel[1] + e.r_foot_vel[2] * e.r_foot_vel[2]);
    const double nlv = std::min(lv, vmax) / vmax, nrv = std::min(rv, vmax) / vmax;
    const double com_orient = 15 * (1 - s.qpos[3] * s.qpos[3]), foot_orient = 10 * (e.l_foot_orient_cost + e.r_foot_orient_cost);
    double straight = std::fabs(s.qpos[1]);
    if (straight < 0.05) straight = 0;
    double hdiff = std::fabs(s.qpos[2] - 1.0);               // +- 0.2 m dead zone around 1.0 m (:452-455)
    if (hdiff < 0.2) hdiff = 0;
    const double lfc = e.clock.eval(0, fphase(e)), lvc = e.clock.eval(1, fphase(e)), rfc = e.clock.eval(2, fphase(e)), rvc = e.clock.eval(3, fphase(e));
    const double frc = std::tanh(lfc * nlf) + std::tanh(rfc * nrf), vel = std::tanh(lvc * nlv) + std::tanh(rvc * nrv);
    return 0.1 * std::exp(-com_orient) + 0.1 * std::exp(-foot_orient) + 0.1 * std::exp(-(straight + hdiff)) + 0.2 * frc + 0.2 * vel +
           0.3 * (s.qvel[0] / 3.0);
}
double eval_clock_reward(Env& e, const double* action) {
    return e.cfg.reward_kind == 2 ? max_vel_clock_reward(e, action) : e.cfg.reward_kind == 1 ? early_clock_reward(e, action) : clock_reward(e, action);
}

// CassieEnv.step, cassie.py:389-496
int env_step(Env& e, const double* action, double* obs, double* reward) {
    if (e.cfg.env_kind == 1 && e.cfg.dynamics_randomization) (void)e.rng.uniform01();   // cassie_traj.py:463-464 draws a simrate that the loop never uses
    e.l_foot_frc = e.r_foot_frc = 0;
    e.l_foot_orient_cost = e.r_foot_orient_cost = 0;
    for (int u = 0; u < 10; ++u) {      // step_simulation :295-326
        e.pd_target[u] = action[u] + kOffset[u] - (e.cfg.dynamics_randomization ? e.motor_noise[u] : 0.0);
        e.pd_P[u] = kP[u % 5]; e.pd_D[u] = kD[u % 5];
    }
    e.rowset_hash = 0;
    for (int i = 0; i < e.cfg.simrate; ++i) {
        sim_step_pd(e);
        fold_rowset(e);
        double fp[6];
        foot_positions(e.st, fp);
        for (int k = 0; k < 3; ++k) { e.l_foot_vel[k] = (fp[k] - e.foot_pos_prev[k]) / 0.0005; e.r_foot_vel[k] = (fp[3 + k] - e.foot_pos_prev[3 + k]) / 0.0005; }
        for (int k = 0; k < 6; ++k) e.foot_pos_prev[k] = fp[k];
        e.l_foot_frc += e.st.foot_force[0][2]; e.r_foot_frc += e.st.foot_force[1][2];
        const Q4 ql = e.st.xquat[13], qr = e.st.xquat[25];
        const double il = kNeutralFoot[0] * ql.w + kNeutralFoot[1] * ql.x + kNeutralFoot[2] * ql.y + kNeutralFoot[3] * ql.z;
        const double ir = kNeutralFoot[0] * qr.w + kNeutralFoot[1] * qr.x + kNeutralFoot[2] * qr.y + kNeutralFoot[3] * qr.z;
        e.l_foot_orient_cost += 1 - il * il; e.r_foot_orient_cost += 1 - ir * ir;
    }
    const double inv = 1.0 / e.cfg.simrate;
    e.l_foot_frc *= inv; e.r_foot_frc *= inv; e.l_foot_orient_cost *= inv; e.r_foot_orient_cost *= inv;
    const double height = e.st.qpos[2];
    e.time += 1; e.est_age += 1; advance_phase(e);
    int done = (height < 0.4 || height > 3.0 || !(height == height)) ? 1 : 0;
    if (!e.has_prev_action) { for (int u = 0; u < 10; ++u) e.prev_action[u] = action[u]; e.has_prev_action = 1; }
    if (!e.has_prev_torque) { for (int u = 0; u < 10; ++u) e.prev_torque[u] = e.so_torque[u]; e.has_prev_torque = 1; }
    *reward = eval_clock_reward(e, action);
    for (int u = 0; u < 10; ++u) { e.prev_action[u] = action[u]; e.prev_torque[u] = e.so_torque[u]; }
    // early_term_cutoff is forced to -99 (cassie.py:773) => the reward never terminates
    Philox& r = e.rng;   // command resampling :483-491, fixed 6 draws per step
    { const uint32_t k = r.randint(300); const double u = r.uniform(-0.2, 0.2); if (k == 0) e.orient_add += u; }
    { const uint32_t k = r.randint(100); const double u = r.uniform(-0.3, 4.0); if (k == 0) e.speed = std::min(std::max(u, -0.3), 4.0); }
    { const uint32_t k = r.randint(300); const double u = r.uniform(-0.3, 0.3); if (k == 0) e.side_speed = u; }
    if (obs) env_obs(e, obs);
    if (!done && e.time >= e.cfg.max_traj_len) done = 2;
    return done;
}

}  // namespace orc
