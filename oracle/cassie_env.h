// CPU ORACLE (test infrastructure only): Cassie-v0 env logic around the physics restatement.
// Follows cassie/cassie.py (CassieEnv.step_simulation :293-351, step :389-496, reset :523-680, get_full_state
// :787-859), cassie/rewards/clock_rewards.py:6-110, cassie/phase_function.py:5-136 and the decoded native substep
// of SURVEY.md §2.2 (pd_input_step -> cassie_core_sim_step -> cassie_sim_step_ethercat -> state_output_step).
#pragma once
#include <cstdint>
#include "cassie_phys.h"
#include "cassie_estimator.h"

namespace orc {

// ---- counter-based RNG shared (by construction, not by code) with the HIP kernels: Philox4x32-10 ----
struct Philox {
    uint32_t key0, key1;      // seed
    uint32_t env;             // stream id
    uint32_t ctr;             // draws consumed so far (persisted per env)
    uint32_t dom = 0;         // stream domain (4th counter word): 0 = the per-step command draws (sequential counter), 1 = reset draws, counter = 128 * episode + k
    uint32_t next_u32();
    double uniform01() { return ((double)(float)(next_u32() >> 8) + 0.5) * (1.0 / 16777216.0); }
    double uniform(double a, double b) { return a + (b - a) * uniform01(); }
    uint32_t randint(uint32_t n) { return (uint32_t)(((uint64_t)next_u32() * n) >> 32); }   // [0, n)
};

struct EnvCfg {
    int simrate = 50;
    int dynamics_randomization = 1;
    int reward_kind = 0;      // 0 clock_reward, 1 early_clock_reward, 2 max_vel_clock_reward
    int stance_mode = 0;      // 0 zero, 1 grounded, 2 aerial
    int have_incentive = 1;
    int max_traj_len = 400;
    int pgs_iters = 50;
    uint64_t seed = 0;
    int command_profile = 0;  // 0 clock (obs 50), 1 phase (cassie.py:266-271,529-545,805-808: obs 55), 2 phase with the "library" draws (:531-539)
    int est_lifetime = 169;   // env steps after which the next reset also restarts the estimator (one PPO.sample call of the reference builds one CassieEnv: ppo.py:152; 5096 // 30, apex.py:244-246); 0 = never
    int input_profile = 0;    // 0 full (46 estimator entries), 1 min (21: foot positions, pelvis orientation / rotational velocity, foot orientations; cassie.py:246-256,829-837)
    int env_kind = 0;         // 0 Cassie-v0 (cassie/cassie.py), 1 CassieTraj-v0 with the CLI defaults (cassie/cassie_traj.py: trajectory-pose reset)
};

struct Clock {                // the four clock splines of create_phase_reward, as knot tables for ONE cycle
    double x[8];              // knot positions inside the cycle (phase units)
    double y[4][8];           // [0] left_clock[0] (= r_frc), [1] left_clock[1] (= r_vel), [2] right_clock[0], [3] right_clock[1]
    double phaselen;
    double eval(int which, double phase) const;
};
void make_clock(Clock& c, double swing, double stance, double relax, int stance_mode, int have_incentive, int freq);

struct Env {
    EnvCfg cfg;
    Params par;
    State st;
    Philox rng;
    // episode / command state (cassie.py:71-78,117-119)
    int time, phase, counter;
    int episode;               // resets done so far: episode e draws its reset from the order-independent stream (seed, env, dom 1, 128 e + k), so a reset can be computed ahead of time
    int est_age;               // env steps since the estimator object was set up
    int phase_half, phase_add15;   // self.phase_add = 1.5 of the command harness (tools/test_commands.py:86, cassie.py:448): the phase is phase + 0.5 phase_half
    double speed, side_speed, orient_add;
    Clock clock;
    double swing_duration, stance_duration;     // of the current clock (observed by the phase command profile)
    // encoder offsets (cassie.py:652-654)
    double motor_noise[10], joint_noise[6];
    // pd_in_t persists across resets (cassie.py:665 steps with the stale self.u)
    double pd_target[10]; double pd_P[10], pd_D[10];
    // native blocks' state: 6-deep torque delay, encoder filters (SURVEY.md §2.2 decoded spec)
    double tq_fifo[10][6];
    double menc_hist[10][9]; int menc_primed;
    double jenc_x[6][4], jenc_y[6][3]; int jenc_primed;
    // sensor snapshot consumed by the NEXT substep (sensordata is one mj_step1 old when step_ethercat reads it)
    double snap_mpos[10], snap_mvel[10], snap_jpos[6], snap_jvel[6], snap_quat[4], snap_gyro[3], snap_acc[3];
    StateOutput est;           // the reference's state estimator object (cassie_estimator.h): persists across episodes, cleared by the full reset
    // state_out_t fields get_full_state / the reward read (cassie.py:817-850, clock_rewards.py:48,77)
    double so_mpos[10], so_mvel[10], so_torque[10], so_jpos[6], so_jvel[6], so_quat[4], so_rotvel[3], so_tvel[3],
        so_tacc[3], so_height;
    // trackers
    double l_foot_vel[3], r_foot_vel[3], foot_pos_prev[6];
    double l_foot_frc, r_foot_frc, l_foot_orient_cost, r_foot_orient_cost;
    double prev_action[10], prev_torque[10]; int has_prev_action, has_prev_torque;
    double last_reward_terms[8];
    long iter_sum, iter_passes; int iter_hist[51];   // solver statistics over every forward pass since env_init: sum / count / histogram of State::solver_iter
    uint32_t rowset_hash = 0;  // hash of State::rowsig over the forward passes of the most recent env_step (the kernel keeps the same in I_ROWSET)
    int sat_acc;               // OR of State::sat over every forward pass since env_init (SatFlag bits)
};

void env_init(Env& e, const EnvCfg& cfg, uint32_t env_id);
void env_reset(Env& e, double* obs);
void env_step_basic(Env& e, const double* action, double* obs);           // CassieEnv.step_basic, cassie.py:498-521
void env_clock_from_speed(Env& e);                                         // swing / stance / clock from e.speed, cassie.py:556-559
void env_update_speed(Env& e, double new_speed, double new_side_speed);   // CassieEnv.update_speed, cassie.py:757-775
void traj_ref_state(double phase, double phaselen, double speed, int counter, double* qpos, double* qvel);   // CassieTrajEnv.get_ref_state, cassie_traj.py:926-972 (walking trajectory, simrate 50)
void env_reset_for_test(Env& e, double* obs, bool full_reset = false);    // CassieEnv.reset_for_test(full_reset), cassie.py:682-742
// returns done flag: 0 running, 1 terminated (height), 2 truncated at max_traj_len (only reported, no reset here)
int env_step(Env& e, const double* action, double* obs, double* reward);
void env_obs(const Env& e, double* obs);
void sim_step_pd(Env& e);
void core_safety(const double* q, const double* qd, const double* cmd, double radio, double* out);   // cassie_core_sim_step model
double eval_clock_reward(Env& e, const double* action);   // exposed for the golden-vector tests          // one 2 kHz substep with the current pd targets (cassie_sim_step_pd)

}  // namespace orc
