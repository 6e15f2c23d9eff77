// CPU ORACLE (test infrastructure only): C entry points for tests/ (ctypes) and bench.py's cpu_baseline leg.
#include <cstdlib>
#include "cassie_env.h"
#include <cstring>
#include <thread>
#include <vector>
#include <chrono>
#include <atomic>

using namespace orc;

extern "C" {

void* orc_env_new(int simrate, int dyn_rand, int reward_kind, int stance_mode, int incentive, int max_traj_len,
                  int pgs_iters, uint64_t seed, uint32_t env_id) {
    EnvCfg c;
    c.simrate = simrate; c.dynamics_randomization = dyn_rand; c.reward_kind = reward_kind; c.stance_mode = stance_mode;
    c.have_incentive = incentive; c.max_traj_len = max_traj_len; c.pgs_iters = pgs_iters; c.seed = seed;
    Env* e = new Env;
    env_init(*e, c, env_id);
    return e;
}
void orc_env_free(void* h) { std::free(const_cast<float*>(((Env*)h)->par.hf_data)); delete (Env*)h; }
void orc_env_reset(void* h, double* obs) { env_reset(*(Env*)h, obs); }
int orc_env_step(void* h, const double* action, double* obs, double* reward) { return env_step(*(Env*)h, action, obs, reward); }
void orc_env_substep(void* h) { sim_step_pd(*(Env*)h); }
void orc_env_step_basic(void* h, const double* action, double* obs) { env_step_basic(*(Env*)h, action, obs); }
void orc_env_set_input_profile(void* h, int ip) { ((Env*)h)->cfg.input_profile = ip; }     // 1 = min (cassie.py:829-837)
void orc_env_set_command_profile(void* h, int cp) { ((Env*)h)->cfg.command_profile = cp; }     // 1 / 2 = phase (obs 55)
void orc_env_set_kind(void* h, int kind) { ((Env*)h)->cfg.env_kind = kind; }     // 1 = CassieTraj-v0 (trajectory-pose reset)
void orc_traj_ref_state(double phase, double phaselen, double speed, int counter, double* qpos, double* qvel) { traj_ref_state(phase, phaselen, speed, counter, qpos, qvel); }
void orc_env_update_speed(void* h, double speed, double side_speed) { env_update_speed(*(Env*)h, speed, side_speed); }
void orc_env_reset_for_test(void* h, double* obs, int full_reset) { env_reset_for_test(*(Env*)h, obs, full_reset != 0); }
void orc_env_apply_force(void* h, const double* xfrc) { for (int k = 0; k < 6; ++k) ((Env*)h)->st.xfrc[k] = xfrc[k]; ((Env*)h)->st.xfrc_body = 1; }   // CassieSim.apply_force on cassie-pelvis
void orc_env_apply_force_body(void* h, const double* xfrc, int body) { for (int k = 0; k < 6; ++k) ((Env*)h)->st.xfrc[k] = xfrc[k]; ((Env*)h)->st.xfrc_body = body; }   // ... on any body (mjData.xfrc_applied row `body`)
// test helper: the command / clock / phase state a training reset with first speed draw `speed0` leaves behind (cassie.py:553-563)
void orc_env_set_command(void* h, double speed0, int phase) { Env& e = *(Env*)h; e.speed = speed0; env_clock_from_speed(e); e.phase = phase; }
void orc_env_obs(void* h, double* obs) { env_obs(*(Env*)h, obs); }

// raw physics: n plain mj_step-like steps with a fixed actuator-side ctrl[10]
void orc_phys_step(void* h, const double* ctrl, int n) {
    static thread_local Work w;
    Env& e = *(Env*)h;
    for (int i = 0; i < n; ++i) { step(e.par, e.st, w, ctrl); e.sat_acc |= e.st.sat; }
}
static thread_local Work g_fwd_work;      // (kept after the call: orc_efc_rows reads the constraint rows of the pass)
void orc_phys_forward(void* h, const double* ctrl) {
    Env& e = *(Env*)h;
    forward(e.par, e.st, g_fwd_work, ctrl);
    e.sat_acc |= e.st.sat;
}
double orc_constraint_violation(void* h) { return constraint_violation(((Env*)h)->st); }
void orc_com_velocity(void* h, double* out) { static thread_local Work w; Env& e = *(Env*)h; com_velocity(e.par, e.st, w, out); }
void orc_momentum(void* h, double* out) { static thread_local Work w; Env& e = *(Env*)h; momentum(e.par, e.st, w, out); }
void orc_inverse_dynamics(void* h, const double* qacc, double* out) { static thread_local Work w; Env& e = *(Env*)h; inverse_dynamics(e.par, e.st, w, qacc, out); }
// constraint rows of the most recent orc_phys_forward of THIS thread: J [nefc, NV] and the row types (0 equality, 1 limit, 2 contact); returns nefc
int orc_efc_rows(void* h, double* J, int* type) {
    Env& e = *(Env*)h;
    for (int i = 0; i < e.st.nefc; ++i) { for (int d = 0; d < NV; ++d) J[i * NV + d] = g_fwd_work.rows[i].J[d]; type[i] = g_fwd_work.rows[i].type; }
    return e.st.nefc;
}
double orc_total_energy(void* h) { static thread_local Work w; Env& e = *(Env*)h; return total_energy(e.par, e.st, w); }

#define FIELD(nm, ptr, cnt) if (!std::strcmp(name, nm)) { if (set) std::memcpy((void*)(ptr), io, sizeof(double) * (cnt)); else std::memcpy(io, (ptr), sizeof(double) * (cnt)); return cnt; }
static int field(Env& e, const char* name, double* io, bool set) {
    if (!set && !std::strcmp(name, "foot_low")) {      // per leg: world z of the foot body origin, lowest world z of the foot capsule (end centre - radius)
        for (int leg = 0; leg < 2; ++leg) {
            const int g = leg, b = cm_geom_body[g];
            const V3 c = e.st.xpos[b] + mul(e.st.xmat[b], v3(cm_geom_pos + 3 * g)), ax = mul(e.st.xmat[b], v3(cm_geom_axis + 3 * g));
            const double z0 = (c + ax * cm_geom_half[g]).z, z1 = (c - ax * cm_geom_half[g]).z;
            io[2 * leg] = e.st.xpos[b].z; io[2 * leg + 1] = std::min(z0, z1) - cm_geom_radius[g];
        }
        return 4;
    }
    FIELD("qpos", e.st.qpos, NQ) FIELD("qvel", e.st.qvel, NV) FIELD("qacc", e.st.qacc, NV) FIELD("qacc_warm", e.st.qacc_warm, NV)
    FIELD("mass", e.par.mass, NB) FIELD("damping", e.par.damping, NV) FIELD("friction", &e.par.friction, 1)
    FIELD("floor_quat", &e.par.floor_quat, 4) FIELD("body_invweight0", e.par.body_invweight0, 2 * NB)
    FIELD("dof_invweight0", e.par.dof_invweight0, NV)
    FIELD("foot_force", e.st.foot_force, 6) FIELD("efc_force", e.st.efc_force, MAXEFC)
    FIELD("motor_noise", e.motor_noise, 10) FIELD("joint_noise", e.joint_noise, 6)
    FIELD("pd_target", e.pd_target, 10) FIELD("pd_P", e.pd_P, 10) FIELD("pd_D", e.pd_D, 10)
    FIELD("speed", &e.speed, 1) FIELD("side_speed", &e.side_speed, 1) FIELD("orient_add", &e.orient_add, 1)
    FIELD("so_mpos", e.so_mpos, 10) FIELD("so_mvel", e.so_mvel, 10) FIELD("so_torque", e.so_torque, 10)
    FIELD("so_jpos", e.so_jpos, 6) FIELD("so_jvel", e.so_jvel, 6) FIELD("so_quat", e.so_quat, 4)
    FIELD("so_rotvel", e.so_rotvel, 3) FIELD("so_tvel", e.so_tvel, 3) FIELD("so_tacc", e.so_tacc, 3) FIELD("so_height", &e.so_height, 1) FIELD("est_heel", e.est.heel, 2) FIELD("est_hx", e.est.hx, 12) FIELD("est_hP", e.est.hP, 72) FIELD("est_zx", e.est.zx, 5) FIELD("est_zP", e.est.zP, 25)
    FIELD("est_terrain", &e.est.terrain, 1) FIELD("est_pos", e.est.pos, 3) FIELD("est_vel", e.est.vel, 3) FIELD("est_foot_rel", e.est.foot_rel, 6) FIELD("est_foot_force", e.est.foot_force, 6) FIELD("est_foot_quat", e.est.foot_quat, 8)
    FIELD("snap_acc", e.snap_acc, 3) FIELD("snap_gyro", e.snap_gyro, 3) FIELD("snap_quat", e.snap_quat, 4)
    FIELD("snap_mpos", e.snap_mpos, 10) FIELD("snap_jpos", e.snap_jpos, 6)
    FIELD("l_foot_vel", e.l_foot_vel, 3) FIELD("r_foot_vel", e.r_foot_vel, 3)
    FIELD("reward_terms", e.last_reward_terms, 8) FIELD("clock_x", e.clock.x, 8) FIELD("phaselen", &e.clock.phaselen, 1)
    FIELD("swing_stance", &e.swing_duration, 2) FIELD("prev_action", e.prev_action, 10) FIELD("prev_torque", e.prev_torque, 10)
    FIELD("tq_fifo", e.tq_fifo, 60) FIELD("menc_hist", e.menc_hist, 90) FIELD("jenc_x", e.jenc_x, 24) FIELD("jenc_y", e.jenc_y, 18) FIELD("foot_pos_prev", e.foot_pos_prev, 6)
    if (!std::strcmp(name, "clock_rebuild")) {      // (set) the clock tables from the env's swing / stance durations: a state copied field by field carries the durations, not the knot tables
        if (set) make_clock(e.clock, e.swing_duration, e.stance_duration, 0.1, e.cfg.stance_mode, e.cfg.have_incentive, 2000 / e.cfg.simrate);
        return 0;
    }
    if (!std::strcmp(name, "enc_primed")) { if (set) { e.menc_primed = (int)io[0]; e.jenc_primed = (int)io[1]; } else { io[0] = e.menc_primed; io[1] = e.jenc_primed; } return 2; }
    if (!std::strcmp(name, "phase_add")) { if (set) { e.phase_add15 = io[0] > 1.25; e.phase_half = (int)io[1]; } else { io[0] = e.phase_add15 ? 1.5 : 1.0; io[1] = e.phase_half; } return 2; }
    if (!std::strcmp(name, "stance_mode")) { if (set) e.cfg.stance_mode = (int)io[0]; else io[0] = e.cfg.stance_mode; return 1; }      // per-env under the phase command profile
    if (!std::strcmp(name, "episode")) { if (set) e.episode = (int)io[0]; else io[0] = e.episode; return 1; }
    if (!std::strcmp(name, "est_age")) { if (set) { e.est_age = (int)io[0]; e.cfg.est_lifetime = (int)io[1]; } else { io[0] = e.est_age; io[1] = e.cfg.est_lifetime; } return 2; }
    if (!std::strcmp(name, "est_flags")) { if (set) e.est.inited = (int)io[0]; else { io[0] = e.est.inited; io[1] = e.est.lm_iters; } return 2; }
    if (!std::strcmp(name, "xpos")) { if (!set) std::memcpy(io, e.st.xpos, sizeof(double) * 3 * NB); return 3 * NB; }
    if (!std::strcmp(name, "xquat")) { if (!set) std::memcpy(io, e.st.xquat, sizeof(double) * 4 * NB); return 4 * NB; }
    if (!std::strcmp(name, "efc_type")) { if (!set) for (int i = 0; i < MAXEFC; ++i) io[i] = e.st.efc_type[i]; return MAXEFC; }
    if (!std::strcmp(name, "solver_hist")) { if (!set) for (int i = 0; i < 51; ++i) io[i] = e.iter_hist[i]; return 51; }
    if (!std::strcmp(name, "solver")) {     // solver_iter of the last forward pass, sum and count over the env's life, tolerance, meaninertia
        if (set) { e.par.tolerance = io[3]; return 5; }
        io[0] = e.st.solver_iter; io[1] = (double)e.iter_sum; io[2] = (double)e.iter_passes; io[3] = e.par.tolerance; io[4] = e.par.meaninertia;
        return 5;
    }
    if (!std::strcmp(name, "ints")) {   // time, phase, counter, ncon, nefc, rng ctr, has_prev_action, has_prev_torque, sat flags (accumulated), ncon1
        int* p[10] = {&e.time, &e.phase, &e.counter, &e.st.ncon, &e.st.nefc, (int*)&e.rng.ctr, &e.has_prev_action, &e.has_prev_torque, &e.sat_acc, &e.st.ncon1};
        if (set) { for (int i = 0; i < 8; ++i) *p[i] = (int)io[i]; return 8; }     // the first 8 are settable (tests); sat / ncon1 are read-only
        for (int i = 0; i < 10; ++i) io[i] = *p[i];
        io[10] = (double)(e.rowset_hash & 0xffffu); io[11] = (double)(e.rowset_hash >> 16);      // row-set hash of the most recent env step, two 16-bit halves
        return 12;
    }
    return -1;
}
int orc_env_get(void* h, const char* name, double* out) { return field(*(Env*)h, name, out, false); }
int orc_env_set(void* h, const char* name, const double* in) { return field(*(Env*)h, name, (double*)in, true); }
// the restated state estimator on its own (golden G11: the reference binary's output on a recorded sensor stream)
void* orc_est_create() { StateOutput* s = new StateOutput; state_output_setup(*s); return s; }
void orc_est_destroy(void* h) { delete (StateOutput*)h; }
void orc_est_setup(void* h) { state_output_setup(*(StateOutput*)h); }
// in: mpos10 jpos6 quat4 gyro3 acc3 (26 doubles); out: pos3 vel3 tacc3 terrain foot_rel6 foot_force6 heel2 lm_iters foot_quat8 (33 doubles)
void orc_est_step(void* h, const double* in, double* out) {
    StateOutput& s = *(StateOutput*)h;
    EstSensors x;
    std::memcpy(x.mpos, in, sizeof(double) * 10); std::memcpy(x.jpos, in + 10, sizeof(double) * 6); std::memcpy(x.quat, in + 16, sizeof(double) * 4);
    std::memcpy(x.gyro, in + 20, sizeof(double) * 3); std::memcpy(x.acc, in + 23, sizeof(double) * 3);
    state_output_step(s, x);
    std::memcpy(out, s.pos, sizeof(double) * 3); std::memcpy(out + 3, s.vel, sizeof(double) * 3); std::memcpy(out + 6, s.tacc, sizeof(double) * 3);
    out[9] = s.terrain; std::memcpy(out + 10, s.foot_rel, sizeof(double) * 6); std::memcpy(out + 16, s.foot_force, sizeof(double) * 6);
    out[22] = s.heel[0]; out[23] = s.heel[1]; out[24] = s.lm_iters; std::memcpy(out + 25, s.foot_quat, sizeof(double) * 8);
}
double orc_est_heel_residual(double knee, double shin, double tarsus, double heel, double* grad4) { return heel_residual(knee, shin, tarsus, heel, grad4); }
void orc_est_hfilter_step(double* x6, double* P36, double zL, double zR, double fl, double fr, double acc) { hfilter_step(x6, P36, zL, zR, fl, fr, acc); }
void orc_est_zfilter_step(double* x5, double* P25, double zL, double zR, double fl, double fr) { zfilter_step(x5, P25, zL, zR, fl, fr); }
void orc_est_mldivide23(const double* M6 /* row-major 2 x 3 */, const double* tau, double* x) { const double M[2][3] = {{M6[0], M6[1], M6[2]}, {M6[3], M6[4], M6[5]}}; mldivide23(M, tau, x); }
unsigned long long orc_flops(int reset) { const unsigned long long v = g_flops; if (reset) g_flops = 0; return v; }      // instrumented op count of this thread
void orc_env_set_const(void* h) { set_const(((Env*)h)->par); }
void orc_env_set_kernel_caps(void* h, int on) { ((Env*)h)->par.kernel_caps = on; }
// CassieSim("cassie_hfield.xml") + set_hfield_data (util/eval.py:73-76): the env keeps its own copy; data == nullptr goes back to the plane
void orc_env_set_hfield(void* h, const float* data, int nrow, int ncol, double sx, double sy, double sz) {
    Env& e = *(Env*)h;
    std::free(const_cast<float*>(e.par.hf_data)); e.par.hf_data = nullptr;
    if (!data) return;
    float* own = (float*)std::malloc(sizeof(float) * (size_t)nrow * ncol);
    std::memcpy(own, data, sizeof(float) * (size_t)nrow * ncol);
    e.par.hf_data = own; e.par.hf_nrow = nrow; e.par.hf_ncol = ncol; e.par.hf_size[0] = sx; e.par.hf_size[1] = sy; e.par.hf_size[2] = sz;
}
void orc_floor_query(void* h, double x, double y, double* out) { double hh; V3 n; floor_query(((Env*)h)->par, x, y, hh, n); out[0] = hh; out[1] = n.x; out[2] = n.y; out[3] = n.z; }

void orc_clock_eval(double swing, double stance, double relax, int mode, int inc, int freq, int n, const double* ph,
                    double* out, double* phaselen) {
    Clock c;
    make_clock(c, swing, stance, relax, mode, inc, freq);
    *phaselen = c.phaselen;
    for (int i = 0; i < n; ++i)
        for (int k = 0; k < 4; ++k) out[4 * i + k] = c.eval(k, ph[i]);
}

// clock_reward on explicitly given inputs (golden G7): scal = l_frc, r_frc, l_orient, r_orient, speed, phase, swing, stance
double orc_clock_reward_eval(void* h, const double* qpos, const double* qvel, const double* scal, const double* foot_vel,
                             const double* rotvel, const double* tacc, const double* torque, const double* prev_torque,
                             const double* prev_action, const double* action) {
    Env& e = *(Env*)h;
    std::memcpy(e.st.qpos, qpos, sizeof(double) * NQ); std::memcpy(e.st.qvel, qvel, sizeof(double) * NV);
    e.l_foot_frc = scal[0]; e.r_foot_frc = scal[1]; e.l_foot_orient_cost = scal[2]; e.r_foot_orient_cost = scal[3];
    e.speed = scal[4]; e.phase = (int)scal[5];
    make_clock(e.clock, scal[6], scal[7], 0.1, e.cfg.stance_mode, e.cfg.have_incentive, 2000 / e.cfg.simrate);
    for (int k = 0; k < 3; ++k) { e.l_foot_vel[k] = foot_vel[k]; e.r_foot_vel[k] = foot_vel[3 + k]; e.so_rotvel[k] = rotvel[k]; e.so_tacc[k] = tacc[k]; }
    for (int u = 0; u < 10; ++u) { e.so_torque[u] = torque[u]; e.prev_torque[u] = prev_torque[u]; e.prev_action[u] = prev_action[u]; }
    return eval_clock_reward(e, action);
}

void orc_core_safety(const double* q, const double* qd, const double* cmd, double radio, double* out) { core_safety(q, qd, cmd, radio, out); }

uint32_t orc_philox(uint64_t seed, uint32_t env, uint32_t ctr, uint32_t dom) {
    Philox p{(uint32_t)seed, (uint32_t)(seed >> 32), env, ctr, dom};
    return p.next_u32();
}

// CPU baseline (bench.py `cpu_baseline`, kind "port"): n_envs envs x n_steps env steps (auto-reset), zero-mean
// pseudo-random actions, one env per task over `threads` std::threads.  Returns env-steps per second.
double orc_rollout_bench(int n_envs, int n_steps, int threads, uint64_t seed, double act_std) {
    std::vector<Env*> envs(n_envs);
    EnvCfg c; c.seed = seed;
    for (int i = 0; i < n_envs; ++i) { envs[i] = new Env; env_init(*envs[i], c, (uint32_t)i); }
    std::atomic<int> next{0};
    auto t0 = std::chrono::steady_clock::now();
    auto work = [&]() {
        for (;;) {
            const int i = next.fetch_add(1);
            if (i >= n_envs) return;
            Env& e = *envs[i];
            Philox an{(uint32_t)(seed ^ 0x5bd1e995u), 77u, (uint32_t)i, 0, 0};
            double obs[50], rew, act[10];
            env_reset(e, obs);
            for (int t = 0; t < n_steps; ++t) {
                for (int u = 0; u < 10; ++u) act[u] = act_std * (an.uniform01() + an.uniform01() + an.uniform01() - 1.5) * 2.0;
                if (env_step(e, act, obs, &rew)) env_reset(e, obs);
            }
        }
    };
    std::vector<std::thread> th;
    for (int k = 0; k < threads; ++k) th.emplace_back(work);
    for (auto& t : th) t.join();
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    for (auto* e : envs) delete e;
    return (double)n_envs * n_steps / dt;
}

}  // extern "C"
