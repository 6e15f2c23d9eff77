/*
 * apx.h — C ABI of libapx.so, the MI355X-native engine behind osudrl/apex's Cassie-v0 PPO hot path.
 *
 * Boundary rules: extern "C", plain pointers and sizes, no torch / C++ types.  Every pointer
 * marked [dev] is DEVICE memory owned by the caller (e.g. a PyTorch-ROCm tensor's data_ptr()); [host] is
 * host memory.  `stream` is a hipStream_t passed as void* (NULL = the null stream).  Functions return
 * APX_OK (0) or a negative error code; apx_last_error() gives the message for the calling thread.
 * Nothing here ever falls back to a CPU path: without a GPU every compute entry point returns APX_E_HIP.
 *
 * Each entry point cites the reference interface (osudrl/apex, file:line) it replaces.
 */
#ifndef APX_H
#define APX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define APX_OK 0
#define APX_E_ARG (-1)
#define APX_E_HIP (-2)
#define APX_E_STATE (-3)

#define APX_OBS_DIM 50   /* cassie/cassie.py:236-265  (46 estimator + 2 clock + 2 speed), command_profile clock */
#define APX_OBS_DIM_PHASE 55   /* + swing, stance, one-hot stance mode (cassie.py:266-271) */
#define APX_OBS_MIN 21         /* estimator entries of input_profile min (cassie.py:237,246-256): obs = 21 + 4 (clock) or 21 + 9 (phase) */
#define APX_ACT_DIM 10   /* cassie/cassie.py:68 */
#define APX_NQ 35        /* cassie/cassiemujoco/cassiemujoco.py:36-39 */
#define APX_NV 32

int apx_version(void);
const char* apx_last_error(void);
/* number of visible HIP devices (0 when there is no GPU); never fails */
int apx_device_count(void);

/* ------------------------------------------------------------------------------------------------------------
 * Learner half
 * ---------------------------------------------------------------------------------------------------------- */

/* Discounted-return backward scan over a [T, N] rollout grid (env-per-column).
 * Replaces PPOBuffer.finish_path, rl/algos/ppo.py:73-89 (+ the bootstrap rule at ppo.py:183-184).
 *   rew[T*N] f32, end[T*N] u8 (non-zero = last step of an episode), boot[T*N] f32 (bootstrap value used at an
 *   episode end: (not done)*V(s_{t+1})), last_val[N] f32 (V of the state after the grid, for running episodes),
 *   ret[T*N] f32 out.  fp64 recurrence like the reference; lambda is not a parameter because the reference never
 *   reads it (ppo.py:50,103,112).                                                         all pointers [dev] */
int apx_returns_scan(const float* rew, const uint8_t* end, const float* boot, const float* last_val, double gamma,
                     int T, int N, float* ret, void* stream);

/* Advantage normalisation, rl/algos/ppo.py:395-396, split so that N>1 ranks can all-reduce the moments.
 *   apx_adv_moments: moments[3] f64 [dev] <- (sum a, sum a^2, count) of a = ret - val over n elements.
 *   apx_adv_apply:   adv[i] = (ret[i]-val[i] - mean) / (std + eps)                         all pointers [dev] */
int apx_adv_moments(const float* ret, const float* val, int64_t n, double* moments, void* stream);
int apx_adv_apply(const float* ret, const float* val, int64_t n, double mean, double std_unbiased, double eps,
                  float* adv, void* stream);
/*   apx_adv_apply_moments: the same with (mean, unbiased std) derived on the device from moments[3] (as written by
 *   apx_adv_moments, summed over ranks by the caller's all-reduce): no host round trip inside an iteration. */
int apx_adv_apply_moments(const float* ret, const float* val, int64_t n, const double* moments, double eps, float* adv,
                          void* stream);

/* Parameter block of one 3-layer ReLU MLP (rl/policies/actor.py:142-215, critic.py:37-77), fp32, torch layout
 * [out,in], packed in state_dict order: W0[H*D] b0[H] W1[H*H] b1[H] W2[O*H] b2[O]. */
size_t apx_mlp_param_count(int D, int H, int O);

/* y[B,O] = MLP(x) with optional input normalisation (obs_mean/obs_std may be NULL = none; FF_V in train mode,
 * critic.py:66-67).  xn_out [B,D] receives the prepared input and act1/act2 the post-ReLU hidden activations [B,H]
 * for a later backward; all three may be NULL (inference: nothing but y is written) for the fused shapes H = 256,
 * D <= 64, O <= 128 (the reference's 2 x 256 nets), otherwise they are required as scratch.
 * idx (may be NULL) gathers rows: x_row = x[idx[b]].  sign_perm (may be NULL) = int32[D] signed permutation
 * applied BEFORE normalisation (SymmetricEnv.mirror_clock_observation, rl/envs/wrappers.py:59-67): entry j>=0 takes
 * +x[j], entry -(j+1) takes -x[j]; clock_mask bit c set => column c additionally gets sin(asin(.)+pi).
 * Arithmetic: fp32 MFMA (exact fp32 products) - the reference's own network precision and the only mode (round 5 removed the bf16 GEMM
 * option: an MI200-era 32x32x8_1k kernel worth +0.6 % end to end, not the parity mode).  all pointers [dev] */
int apx_mlp_forward(const float* params, int D, int H, int O, const float* x, int64_t B, const int64_t* idx,
                    const int32_t* sign_perm, uint64_t clock_mask, const float* obs_mean, const float* obs_std,
                    float* xn_out, float* act1, float* act2, float* y, void* stream);

/* TD3 primitives (next row f2; rl/algos/sync_td3.py:133-209 TD3.train with FF_Actor, rl/policies/actor.py:43-72, and Dual_Q_Critic,
 * rl/policies/critic.py:118-168, each Q a 3-layer ReLU MLP on cat(state, action)).  The networks run through apx_mlp_forward (no
 * normalisation); these entry points add what PPO did not need:
 * apx_mlp_backward: grads += d(loss)/d(params) for dy[B,O] (grads NULL = skip the parameter gradients) and optionally
 *   dx[B,D] = d(loss)/d(input); xn / a1 / a2 are the arrays apx_mlp_forward kept; scratch = 2*B*H floats.
 * apx_polyak: target <- tau*param + (1-tau)*target (sync_td3.py:196-202).
 * apx_td3_cat_action: critic input [B, D+A] = cat(state, max_action*tanh(pre_tanh) [+ clamp(noise, +-noise_clip), clamped to
 *   +-max_action]) (FF_Actor.forward; target-policy smoothing sync_td3.py:146-151 when noise != NULL).
 * apx_td3_critic_loss: target_Q = reward + notdone*discount*min(tq1, tq2) (:154-156); dq1 / dq2 = gradients of
 *   mse(q1, target) + mse(q2, target) (:168-169); acc3 = (critic loss, sum q1, sum q2) f64 [dev].
 * apx_td3_actor_grad: d(loss)/d(pre_tanh) from dx = d(-mean Q1)/d(critic input) (:183). */
int apx_mlp_backward(const float* params, float* grads, int D, int H, int O, const float* xn, const float* a1, const float* a2,
                     const float* dy, int64_t B, float* dx, float* scratch, void* stream);
int apx_polyak(float* target, const float* param, int64_t n, float tau, void* stream);
int apx_td3_cat_action(const float* state, const float* pre_tanh, const float* noise, float noise_clip, float max_action, int64_t B,
                       int D, int A, float* out, void* stream);
int apx_td3_critic_loss(const float* q1, const float* q2, const float* tq1, const float* tq2, const float* reward, const float* notdone,
                        float discount, int64_t B, float* dq1, float* dq2, double* acc3, void* stream);
int apx_td3_actor_grad(const float* dx, const float* pre_tanh, float max_action, int64_t B, int D, int A, float* dpre, void* stream);

/* Recurrent actor / critic (next row f1): Gaussian_LSTM_Actor (rl/policies/actor.py:218-311) and LSTM_V (rl/policies/critic.py:236-296) =
 * L stacked nn.LSTMCell(H) + Linear(H, O).  Parameter block in state_dict order: per cell weight_ih[4H,in] weight_hh[4H,H] bias_ih[4H]
 * bias_hh[4H] (gate order i,f,g,o), then network_out weight[O,H] bias[O].
 * apx_lstm_forward: x[T,B,D] f32 [dev] = prepared (normalised) inputs of a padded batch of trajectories (actor.py:259-269) or T = 1 for
 * a rollout step; hc = [L][2][B][H] carried (h, c), read as the start state and overwritten with the final one, or NULL = zero start
 * (init_hidden_state, actor.py:291-293); save = apx_lstm_workspace_floats(T,B,H,L) floats (gates / cell / hidden states, kept for
 * apx_lstm_backward); y[T,B,O].
 * apx_lstm_backward: grads (same layout as params) += d(loss)/d(params) for dy[T,B,O], zero start state; scratch =
 * apx_lstm_bwd_scratch_floats(T,B,D,H) floats. */
size_t apx_lstm_param_count(int D, int H, int L, int O);
size_t apx_lstm_workspace_floats(int T, int64_t B, int H, int L);
size_t apx_lstm_bwd_scratch_floats(int T, int64_t B, int D, int H);
int apx_lstm_forward(const float* params, int D, int H, int L, int O, const float* x, int T, int64_t B, float* hc, float* save,
                     float* y, void* stream);
/* The rollout's one-step recurrent pass as ONE launch (PPO.sample's policy step, rl/algos/ppo.py:160-184 over rl/policies/actor.py:253-289): input normalisation
 * (obs_mean / obs_std, or NULL for a prepared input), init_hidden_state for the rows whose `reset` byte is non-zero (NULL: none), two LSTMCell(128), the linear head and,
 * optionally, act = y + sigma * noise.  `packed` = the network's parameters re-laid by apx_lstm_step_pack ([W_ih | W_hh] per cell with W_ih padded to 64 columns, summed
 * biases, head; apx_lstm_step_pack_floats floats, 0 = shape not supported: L = 2, H = 128, D <= 64, O <= 16) - pack again whenever the parameters change.
 * hc = [2][2][B][128] carried (h, c), updated in place; x[B,D], y[B,O], act[B,O] or NULL, noise[B,O] or NULL; all f32 [dev]. */
size_t apx_lstm_step_pack_floats(int D, int H, int L, int O);
int apx_lstm_step_pack(const float* params, int D, int H, int L, int O, float* packed, void* stream);
int apx_lstm_step(const float* packed, int D, int H, int L, int O, const float* x, const float* obs_mean, const float* obs_std, const uint8_t* reset, float* hc,
                  int64_t B, float* y, float* act, const float* noise, float sigma, void* stream);
int apx_lstm_backward(const float* params, float* grads, int D, int H, int L, int O, const float* x, int T, int64_t B,
                      const float* save, const float* dy, float* scratch, void* stream);

/* The padded minibatch of the recurrent update (rl/algos/ppo.py:411-430: pad_sequence over the sampled trajectories, zero rows behind a trajectory's end) gathered out of
 * the rollout grid in one launch.  Which grid row column b holds at step t: either idx [T, B] int64 (flat grid row, -1 = padded), or - idx NULL - the trajectory list
 * traj [n_traj, 3] int64 = (grid column n, t0, t1) with sel [B] int64 (NULL: 0..B-1) naming column b's trajectory and N = the grid's column count: step t of the
 * trajectory is grid row (t0 + t) N + n, padded from t1 - t0 on (no index tensor is built at all).  obs [rows_total, D], act [rows_total, A], ret / adv [rows_total].
 * Outputs (all [T, B, .] f32): obs_raw (LSTM_V's input, critic.py:262-263), xn = (obs - obs_mean) / obs_std, act_p, ret_p, adv_p, mask (1 = real row), and - when
 * obs_sign_perm is given (SymmetricEnv.mirror_clock_observation, rl/envs/wrappers.py:59-67; clock_mask as in apx_mlp_forward) - xa [T, 2 B, D] = [xn | normalised
 * mirrored observation] along the batch axis (pi(s) and pi(M s) share the weights: one 2 B-column pass).  xa and obs_sign_perm are both NULL or both given. [dev] */
int apx_rec_gather(const int64_t* idx, const int64_t* traj, const int64_t* sel, int64_t N, int T, int64_t B, int D, int A, const float* obs, const float* act, const float* ret, const float* adv,
                   const int32_t* obs_sign_perm, uint64_t clock_mask, const float* obs_mean, const float* obs_std, float* obs_raw, float* xn, float* xa,
                   float* act_p, float* ret_p, float* adv_p, float* mask, void* stream);

/* The loss stage of PPO.update_policy alone (rl/algos/ppo.py:284-318,338-345) for callers that run their own forward / backward (the
 * recurrent path): rows = T*B entries of a padded batch, mu / mum (mirrored branch, NULL = no mirror loss) [rows,A], v [rows], act
 * [rows,A], ret / adv / mask [rows] (mask NULL = all ones; it weights the actor and critic terms only, every mean is over ALL rows like
 * the reference's plain .mean() over the padded tensor), old_mu [rows,A].  Writes d(loss)/d(mu), d(loss)/d(mum), d(loss)/d(v) and the
 * six scalars (actor loss, entropy, critic loss, ratio mean, KL mean, mirror loss) f64 [dev]; acc_ws = 8 doubles [dev]. */
int apx_ppo_loss(const float* mu, const float* mum, const float* v, const float* act, const float* ret, const float* adv,
                 const float* old_mu, const float* mask, const int32_t* act_sign_perm, int64_t rows, int A, float fixed_std,
                 float clip, float mirror_coeff, float* dmu, float* dmum, float* dv, double* scalars_out, double* acc_ws, void* stream);

/* One PPO minibatch step: rl/algos/ppo.py:276-345 PPO.update_policy (feed-forward branch).
 * All device pointers; scalars_out[6] f64 [dev] receives (actor_loss, entropy, critic_loss, ratio.mean, kl.mean,
 * mirror_loss) exactly as update_policy returns them. */
typedef struct apx_ppo_args {
    /* networks + Adam state (fp32, apx_mlp_param_count floats each) */
    float* actor; float* actor_m; float* actor_v; float* actor_grad;
    float* critic; float* critic_m; float* critic_v; float* critic_grad;
    int D, H, A;
    /* rollout buffer (whole iteration) and minibatch selection */
    const float* obs;      /* [Btot, D] raw observations                    (ppo.py:393) */
    const float* act;      /* [Btot, A]                                                   */
    const float* ret;      /* [Btot]                                                      */
    const float* adv;      /* [Btot] normalised advantages                  (ppo.py:396) */
    const float* old_mu;   /* [Btot, A] old policy means                    (ppo.py:284-285) */
    const int64_t* idx;    /* [mb] row indices of this minibatch (BatchSampler, ppo.py:416) or NULL = 0..mb-1 */
    int64_t mb;
    const float* obs_mean; const float* obs_std;   /* [D] shared normaliser (ppo.py:546-549) */
    /* mirror loss (ppo.py:302-318): NULL sign_perm disables it */
    const int32_t* obs_sign_perm; uint64_t clock_mask; const int32_t* act_sign_perm;
    /* hyper-parameters */
    float fixed_std, clip, entropy_coeff, grad_clip, lr, adam_eps, mirror_coeff;
    int adam_t;            /* 1-based optimiser step count (bias correction) */
    int grad_only;         /* 1 = stop after gradients (no clip/Adam): lets N>1 ranks all-reduce actor_grad/critic_grad.
                            * 2 then 3 = the same in two calls: 2 ends when the ACTOR's gradient is final (forwards, losses, scalars, actor backward), 3 runs the critic's
                            * backward on the activations call 2 left in the workspace - so that the actor half of the all-reduce travels while the critic's backward runs */
    /* scratch: apx_ppo_workspace_bytes(mb, D, H, A) bytes [dev] */
    void* workspace; size_t workspace_bytes;
    double* scalars_out;
} apx_ppo_args;

size_t apx_ppo_workspace_bytes(int64_t mb, int D, int H, int A);
int apx_ppo_minibatch(const apx_ppo_args* args, void* stream);

/* One EPOCH of optimiser steps in one launch, for the small minibatches of the reference's parity run (apex.py:242: minibatch_size 64): the loop
 * `for indices in sampler: self.update_policy(...)` of rl/algos/ppo.py:417-438 with sampler = BatchSampler(SubsetRandomSampler(range(B)), mb, drop_last=True)
 * (:414-415).  perm [nb * mb] int64 [dev] = the epoch's sample order; step k = apx_ppo_minibatch on idx = perm[k mb : (k + 1) mb] with adam_t = args->adam_t + k
 * (forwards, losses, backwards, clip_grad_norm_, Adam - same arithmetic, fp32 MFMA; only the summation order of the GEMMs differs).  args as for apx_ppo_minibatch
 * except: idx NULL, grad_only 0, adam_t = step count of the FIRST minibatch, workspace >= apx_ppo_epoch_workspace_bytes, scalars_out [nb, 6] f64 [dev] = the six
 * scalars of every step.  actor_grad / critic_grad hold the last step's gradients afterwards.  Single-GPU only (a gradient all-reduce per step needs the per-step
 * entry point).  apx_ppo_epoch_supported: H = 256, D <= 64, A <= 16, mb a multiple of 16 in 16..1024; the parameter blocks and the workspace 16-byte aligned. */
int apx_ppo_epoch_supported(int64_t mb, int D, int H, int A);
size_t apx_ppo_epoch_workspace_bytes(int64_t mb, int64_t nb, int D, int H, int A);
int apx_ppo_epoch(const apx_ppo_args* args, const int64_t* perm, int64_t nb, void* stream);
/* A block of TD3 updates in one launch: `iterations` passes of the loop body of TD3.train (rl/algos/sync_td3.py:133-209, the same body in async_td3.py) - sample a batch,
 * target-policy smoothing (:148-154), clipped double-Q target (:156-159), both critic regressions + Adam (:161-172), every policy_freq-th iteration the actor step on
 * -Q1(s, pi(s)) (:175-186) and the Polyak averaging of both targets (:188-202).  The batches are rows `ind` [U, B] int64 [dev] of the replay tensors (the reference
 * draws them with np.random.randint inside the loop, rl/utils/remote_replay.py:78-90), the smoothing noise `noise` [U, B, A] ~ N(0, policy_noise) [dev] is clamped to
 * +-noise_clip here.  critic / critic_t / critic_m / critic_v: Dual_Q_Critic's two networks as one flat block (q1 then q2, apx_mlp_param_count(D + A, 256, 1) floats
 * each).  it0 = the iteration counter of the first update (the actor step runs when it % policy_freq == 0), t_a / t_c = Adam steps already taken by the two
 * optimisers.  stats_out [U, 4] f64 [dev]: critic loss, sum q1, sum q2, actor loss (0 without the actor step) of every update.  Adam as torch.optim.Adam (no clipping).
 * apx_td3_updates_supported: H = 256, 48 <= D, D + A <= 64, A <= 12, B a multiple of 16 in 16..4096; parameter blocks and workspace 16-byte aligned. */
typedef struct apx_td3_args {
    float* actor; float* actor_t; float* actor_m; float* actor_v;
    float* critic; float* critic_t; float* critic_m; float* critic_v;
    int D, H, A;
    const float* state; const float* next_state; const float* action; const float* reward; const float* notdone;   /* replay buffer [cap, .] */
    const int64_t* ind; const float* noise;
    int64_t B, U;
    int it0, policy_freq;
    float max_action, noise_clip, discount, tau, a_lr, c_lr, adam_eps;
    int t_a, t_c;
    void* workspace; size_t workspace_bytes;
    double* stats_out;
} apx_td3_args;
int apx_td3_updates_supported(int64_t B, int D, int H, int A);
size_t apx_td3_updates_workspace_bytes(int64_t B, int64_t U, int D, int H, int A);
int apx_td3_updates(const apx_td3_args* args, void* stream);
/* Diagnostic of the grid-wide barrier the two persistent launches above rely on (no reference counterpart: the reference's optimiser steps are separate torch calls):
 * `workgroups` (1..256) resident workgroups run `phases` phases; in each one workgroup overwrites n_words words with the phase's pattern by plain stores, all pass the
 * barrier, every workgroup reads the words back by plain loads, second barrier.  workspace [2 + n_words] u32 [dev], result [4] u64 [dev]: words read stale (must be 0),
 * watchdog flag (must be 0), phases completed by workgroup 0, sum over workgroups of phases completed (must be workgroups x phases). */
int apx_grid_barrier_selftest(int workgroups, int phases, int n_words, unsigned* workspace, unsigned long long* result, void* stream);

/* second half when grad_only=1 was used: global-norm clip (clip_grad_norm_, ppo.py:326,335) + Adam (ppo.py:355-356)
 * on an (all-reduced) gradient; scale multiplies the gradient first (1/world_size for an averaged sum). */
int apx_clip_adam(float* param, float* m, float* v, float* grad, int64_t n, float grad_scale, float grad_clip,
                  float lr, float adam_eps, int adam_t, double* sumsq_scratch, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Environment half: batched Cassie-v0.  Replaces the per-env FFI of cassie/cassiemujoco/cassiemujoco.py:33-336
 * (cassie_sim_init / cassie_sim_step_pd / accessors, include/cassiemujoco.h:41-275) and the Python env logic of
 * cassie/cassie.py:293-496,523-680,787-859 + cassie/rewards/clock_rewards.py:6-110 with ONE handle per GPU.
 * ---------------------------------------------------------------------------------------------------------- */
typedef struct apx_env apx_env_t;

typedef struct apx_env_cfg {
    int n_envs;                 /* envs on this GPU (multiple of 64) */
    int simrate;                /* physics substeps per env step, apex.py:18 (default 50) */
    int dynamics_randomization; /* apex.py:19 */
    int reward_kind;            /* 0 clock_reward (clock_rewards.py:6-110), 1 early_clock_reward (:119-223), 2 max_vel_clock_reward (:416-480) */
    int stance_mode;            /* 0 zero, 1 grounded, 2 aerial (cassie.py:209-214) */
    int have_incentive;         /* cassie.py:91 */
    int max_traj_len;           /* apex.py:248: auto-reset horizon used by apx_env_step's truncation flag */
    uint64_t seed;              /* Philox key; stream = (seed, env id) */
    int device;                 /* HIP device ordinal */
    int pgs_iters;              /* cassie.xml:5 iterations (50) */
    int env_id_base;            /* global index of this shard's env 0 (RNG stream id = env_id_base + local env) */
    int env_kind;               /* 0 Cassie-v0 (cassie/cassie.py), 1 CassieTraj-v0 with the CLI defaults traj=walking, command_profile=clock,
                                 * input_profile=full, no_delta (cassie/cassie_traj.py): same step, reset to the reference trajectory's pose
                                 * of the random start phase (:599-778, get_ref_state :926-972); needs simrate 50 */
    int command_profile;        /* 0 clock (obs 50: ..., sin, cos, speed, side speed), 1 phase (cassie.py:266-271,529-545,805-808: swing / stance
                                 * duration and the stance mode are drawn per reset, obs 55: ..., sin, cos, swing, stance, one-hot stance mode,
                                 * speed, side speed), 2 phase with the "library" draws (:531-539) */
    int est_lifetime;           /* env steps after which the next CassieEnv.reset also restarts the state estimator (state_output_setup), 0 = never.  The reference
                                 * builds a NEW CassieEnv -> cassie_sim_init -> estimator per PPO.sample call (rl/algos/ppo.py:152), i.e. an estimator object serves
                                 * num_steps // num_procs env steps (apex.py:244-246 defaults: 5096 // 30 = 169) and then the episodes of the next call start
                                 * from a fresh one; a lock-step env lives for the whole training run, so the lifetime is carried per env instead */
    int input_profile;          /* 0 full: the 46 joint-level estimator entries (cassie.py:244,839-850); 1 min: 21 entries = leftFoot / rightFoot .position, pelvis
                                 * orientation, rotational velocity, leftFoot / rightFoot .orientation of state_out_t (cassie.py:246-256,829-837) */
    int reserved[2];
} apx_env_cfg;

void apx_env_default_cfg(apx_env_cfg* cfg);
int apx_env_create(const apx_env_cfg* cfg, apx_env_t** out);
int apx_env_destroy(apx_env_t* env);

/* CassieEnv.reset (cassie/cassie.py:523-680) for envs whose mask byte is non-zero (mask NULL = all);
 * obs_out[n_envs*50] f32 [dev] (rows of un-reset envs are left untouched). */
/* Terrain: CassieSim("cassie_hfield.xml") + set_hfield_data (cassie/cassiemujoco/cassiemujoco.py:309-315, util/eval.py:73-76,
 * cassie_hfield.xml:69,74).  data [nrow, ncol] raw elevations (host or device pointer, copied; rows along y, columns along x),
 * size3 = (x half extent, y half extent, elevation scale); elevation = data * size3[2].  Replaces the floor plane for every env of
 * the handle (foot / tarsus / shin capsule ends against the triangle under them); data == NULL restores the plane. */
int apx_env_set_hfield(apx_env_t* env, const float* data, int nrow, int ncol, const float* size3, void* stream);

int apx_env_reset(apx_env_t* env, const uint8_t* mask, float* obs_out, void* stream);

/* Scheduling hook without a reference counterpart: compute, for every env, the part of its NEXT two CassieEnv.reset calls (cassie/cassie.py:523-665) that does
 * not depend on how the running episode ends - command / clock / dynamics-randomisation draws (:525-657), sim.set_const (:660: mj_setConst, init pose, mj_forward) -
 * into a per-env ring, e.g. while the learner runs.  A later reset (apx_env_reset, the auto-reset of apx_env_step / apx_rollout) that finds its episode in the ring
 * copies it and only runs the settle step_pd (:665) and the command redraw (:669-670); otherwise it computes everything as before.  Results are bit-identical either
 * way: reset draws are keyed by (seed, env, episode index), not by the env's running draw counter.  The ring is dropped by apx_env_set_hfield,
 * apx_env_apply_force(_body) and apx_env_set_field. */
int apx_env_prepare_resets(apx_env_t* env, void* stream);
/* The same hook inside a rollout: when on, every auto-reset of apx_env_step / apx_rollout is followed by the ring refill of the envs that just restarted (their next two
 * episodes), launched on a stream the env owns in front of the following env step (behind whatever the caller put on `stream` in between: the policy step), so that it
 * runs next to that env step; the next reset launch waits for it.  Default: on for n_envs <= 2048 (an env step of up
 * to 2048 envs leaves half of the SIMDs idle), off above.  Results do not depend on it (same bits, see apx_env_prepare_resets).  Switching it off waits for a refill
 * in flight; a step issued on a stream under graph capture skips it (switch it off BEFORE capturing a rollout, so that no refill is pending inside the capture). */
int apx_env_set_refill(apx_env_t* env, int on);
/* Rows beyond the lane map of the fast constraint stage (a third penetrating capsule end of a leg, a second active joint limit of a leg, the hip-pitch capsules
 * cassie.xml:101,163 or the pelvis sphere :87 on the floor, more than three left-right capsule pairs): on (default), such a forward pass is solved with its COMPLETE
 * row set in the order of mj_makeConstraint's per-leg restatement (oracle/cassie_phys.cpp) by an out-of-line z~-space Gauss-Seidel; off, the rows are capped as in
 * rounds 1-4.  Either way I_SAT (apx_env_get_field "ints") counts the passes.  Replaces nothing of the reference: MuJoCo always instantiates every row. */
int apx_env_set_complete_rows(apx_env_t* env, int on);

/* Evaluation-side API.
 * CassieEnv.update_speed (cassie/cassie.py:757-775, clock command profile) for every env: speed[n_envs] f32 [dev],
 * side_speed[n_envs] f32 [dev] or NULL (= 0): commands are clipped to [-0.3, 4] / [-0.3, 0.3], the clock is rebuilt from the
 * new speed and the phase is rescaled to the new cycle length. */
int apx_env_update_speed(apx_env_t* env, const float* speed, const float* side_speed, void* stream);
/* CassieEnv.reset_for_test(full_reset) (cassie/cassie.py:682-742) for every env: counters / commands to zero, the fixed
 * 0.15 / 0.25 grounded clock (the handle's stance mode becomes grounded); full_reset = 0: one step_pd with the stale pd targets;
 * full_reset = 1: cassie_sim_full_reset (include/cassiemujoco.h:202: init pose, zero velocities, external wrench and torque
 * delay line) + reset_cassie_state (cassie.py:733-746); then default dynamics + set_const, flat floor, zero encoder offsets;
 * obs_out[n_envs*50] f32 [dev]. */
int apx_env_reset_for_test(apx_env_t* env, float* obs_out, int full_reset, void* stream);
/* CassieSim.apply_force(xfrc, "cassie-pelvis") (cassiemujoco.py:99-103, cassie_sim_apply_force include/cassiemujoco.h:184):
 * xfrc[n_envs*6] f32 [dev] = world-frame force xyz then torque xyz on the pelvis at its centre of mass; it stays applied
 * until overwritten (tools/eval_perturb.py:62,70) or a full reset. */
int apx_env_apply_force(apx_env_t* env, const float* xfrc, void* stream);
/* CassieSim.apply_force(xfrc, body_name) for any body (tools/eval_perturb.py's perturb_body): body = MuJoCo body id of cassie.xml,
 * 1 = cassie-pelvis, 2..13 = left hip-roll, hip-yaw, hip-pitch, achilles-rod, knee, knee-spring, shin, tarsus, heel-spring, foot-crank,
 * plantar-rod, foot, 14..25 = the same on the right.  The wrench acts at that body's centre of mass (one row of mjData.xfrc_applied).
 * ONE pushed body at a time per handle: the call replaces the previous wrench whatever body that was on (the reference keeps a row per
 * body; its harnesses only ever push one). */
int apx_env_apply_force_body(apx_env_t* env, const float* xfrc, int body, void* stream);
/* CassieEnv.step_basic (cassie/cassie.py:498-521, 355-387) for every env: the substeps and the time / phase bookkeeping of a
 * step, without reward, termination, trackers or command resampling (evaluation at a fixed command); obs[n_envs*50]. */
int apx_env_step_basic(apx_env_t* env, const float* action, float* obs, void* stream);

/* CassieEnv.step (cassie/cassie.py:389-496) for every env: action[n_envs*10] f32 -> obs[n_envs*50] f32,
 * reward[n_envs] f32, done[n_envs] u8 (1 terminated, 2 truncated at max_traj_len).  With auto_reset != 0 an env
 * that finished is reset by a second launch on the same stream; its terminal observation (needed for the bootstrap
 * value, ppo.py:183) goes to final_obs (may be NULL; rows of envs that did not finish are left untouched) and obs gets
 * the post-reset observation.                                                                all pointers [dev] */
/* PPO.sample's rollout loop (rl/algos/ppo.py:160-181) for the whole batch as ONE call: for t < T:
 * mu_grid[t] = actor(obs_grid[t]) (3-layer ReLU MLP, hidden H, input normalisation obs_mean / obs_std or NULL),
 * act_grid[t] = mu_grid[t] + sigma * noise[t] (noise [T, N, act] ~ N(0, 1) supplied by the caller; NULL = deterministic),
 * env step with auto-reset -> obs_grid[t + 1] (obs_next after the last step), rew_grid[t], done_grid[t] (1 terminated, 2 time limit),
 * fin_grid[t] (final observation of the envs that finished).  obs_grid[0] holds the current observation on entry.  All [dev]. */
int apx_rollout(apx_env_t* env, const float* actor, int H, const float* obs_mean, const float* obs_std, float sigma, const float* noise, int T,
                float* obs_grid, float* act_grid, float* mu_grid, float* rew_grid, uint8_t* done_grid, float* fin_grid, float* obs_next,
                void* stream);

/* The same rollout with the recurrent actor (PPO.sample over Gaussian_LSTM_Actor, rl/algos/ppo.py:160-184 / rl/policies/actor.py:253-293): actor = the parameter block of
 * apx_lstm_forward (L = 2 cells of H = 128, head of 10; other shapes: APX_E_ARG, the caller keeps its per-step loop of apx_lstm_step + apx_env_step).  Every env starts
 * the rollout from the zero hidden state and returns to it when its episode ends (init_hidden_state); the carried states live in the handle.  One launch: a wave steps
 * its four envs at its own pace, so a forward pass that needs the complete row set (apx_env_set_complete_rows) delays that wave alone, not the launch of every env.
 * Grids as in apx_rollout. */
int apx_rollout_lstm(apx_env_t* env, const float* actor, int H, int L, const float* obs_mean, const float* obs_std, float sigma, const float* noise, int T,
                     float* obs_grid, float* act_grid, float* mu_grid, float* rew_grid, uint8_t* done_grid, float* fin_grid, float* obs_next,
                     void* stream);

/* The collection phase of TD3 as the same one-launch rollout (rl/algos/sync_td3.py:59-101 collect_experience: the policy is fixed while transitions are collected):
 * actor = FF_Actor's parameter block (50-256-256-10, apx_mlp_forward layout), action = clip(max_action * tanh(net(obs)) + act_noise * noise, -1, 1) with noise
 * [T, n_envs] (one scalar per env step, sync_td3.py:77-78; noise_per_dim = 0) or [T, n_envs, 10] (async_td3.py:253-256; noise_per_dim = 1), NULL = none.
 * mu_grid receives the squashed means.  Other hidden widths, a stream under capture or APX_ROLLOUT_STEPWISE=1: APX_E_ARG (the caller steps the env itself). */
int apx_rollout_td3(apx_env_t* env, const float* actor, int H, float max_action, float act_noise, const float* noise, int noise_per_dim, int T,
                    float* obs_grid, float* act_grid, float* mu_grid, float* rew_grid, uint8_t* done_grid, float* fin_grid, float* obs_next,
                    void* stream);

int apx_env_step(apx_env_t* env, const float* action, float* obs, float* reward, uint8_t* done, float* final_obs,
                 int auto_reset, void* stream);

/* Measurement hook (no counterpart in the reference; bench.py's roofline.achieved): while enabled, every env_step_kernel launch - from
 * apx_env_step and from the steps inside apx_rollout - is bracketed by a hipEvent pair on the launch stream.  apx_env_timing_read waits
 * for the recorded launches and returns their summed duration and count since the last reset (at most 4096 launches between two reads
 * are recorded, later ones run untimed). */
int apx_env_timing(apx_env_t* env, int enable);
int apx_env_timing_read(apx_env_t* env, double* total_ms, int64_t* launches, int reset);

/* Raw simulator state access (cassie_sim_qpos / cassie_sim_qvel, include/cassiemujoco.h:83-87): copies, SoA ->
 * [n_envs, 35] / [n_envs, 32] row-major f32.  Used by tests and by apx_env_set_state for parity replays. */
int apx_env_get_state(apx_env_t* env, float* qpos, float* qvel, void* stream);
int apx_env_set_state(apx_env_t* env, const float* qpos, const float* qvel, void* stream);
/* named per-env scalar/vector fields for tests and tools: returns count of floats per env, or <0 */
int apx_env_get_field(apx_env_t* env, const char* name, float* out, void* stream);
int apx_env_set_field(apx_env_t* env, const char* name, const float* in, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* APX_H */
