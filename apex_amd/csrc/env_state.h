// Per-env persistent state layout (SoA in HBM, field-major) + tiny helpers shared by the env kernels.
#pragma once
#include <gfx950/lane_ops.h>      // (hip_runtime.h + the inline-assembly dialect of the env kernels)
#include "../../include/apx.h"

constexpr int ES_NQ = 35, ES_NV = 32, ES_NB = 26;
// ------------------------------------------------------------------------------------------------ state layout
enum Field : int {
    F_QPOS = 0, F_QVEL = F_QPOS + ES_NQ, F_QACCW = F_QVEL + ES_NV, F_MASS = F_QACCW + ES_NV, F_DAMP = F_MASS + ES_NB,
    F_FRIC = F_DAMP + ES_NV, F_FLOOR = F_FRIC + 1 /* n, t1, t2: 9 */, F_BIW = F_FLOOR + 9, F_DIW = F_BIW + ES_NB,
    F_MNOISE = F_DIW + ES_NV, F_JNOISE = F_MNOISE + 10, F_PDT = F_JNOISE + 6, F_FIFO = F_PDT + 10, F_MENC = F_FIFO + 60,
    F_JENCX = F_MENC + 90, F_JENCY = F_JENCX + 24, F_SNAP = F_JENCY + 12 /* mpos10 jpos6 quat4 gyro3 acc3 */,
    F_SO = F_SNAP + 26 /* mpos10 mvel10 torque10 jpos6 jvel6 quat4 rotvel3 tvel3 tacc3 height1 */, F_FOOTPREV = F_SO + 56,
    F_FOOTVEL = F_FOOTPREV + 6, F_PREVACT = F_FOOTVEL + 6, F_PREVTQ = F_PREVACT + 10,
    F_CMD = F_PREVTQ + 10 /* speed side orient swing stance phaselen stance_mode */, F_FWD = F_CMD + 7 /* foot force z L,R; foot quat L4 R4; foot pos 6 */,
    F_XFRC = F_FWD + 16 /* external wrench, world frame: force xyz, torque xyz (the mjData.xfrc_applied row of body I_XBODY) */,
    F_TOTAL = F_XFRC + 6      // (the state estimator's persistent state is not here: env-major records in apx_env::wk, estimator_lane.h)
};
enum SnapOff { SN_MPOS = 0, SN_JPOS = 10, SN_QUAT = 16, SN_GYRO = 20, SN_ACC = 23 };
enum SoOff { SO_MPOS = 0, SO_MVEL = 10, SO_TORQUE = 20, SO_JPOS = 30, SO_JVEL = 36, SO_QUAT = 42, SO_ROTVEL = 46, SO_TVEL = 49, SO_TACC = 52, SO_HEIGHT = 55 };
enum IField : int { I_TIME = 0, I_PHASE, I_COUNTER, I_RNG, I_FLAGS /* bit0 menc primed, 1 jenc primed, 2 prev_action, 3 prev_torque */,
                    I_AGE /* env steps since the state estimator of this env was set up (apx_env_cfg.est_lifetime) */,
                    I_SAT /* constraint sets beyond what the kernel instantiates: bits 0-3 = SAT_* flags seen since env creation, bits 8.. = number of such forward passes */,
                    I_XBODY /* MuJoCo body id the F_XFRC wrench acts on (0 / 1 = cassie-pelvis) */,
                    I_ROWSET /* hash of the active constraint-row sets (limits, capsule ends, body-floor, leg-leg pairs) of the 50 forward passes of the most recent env step:
                                the parity tests bin kernel-vs-oracle errors by "same row sets in every substep" (tests/test_gpu_env.py) */,
                    I_EPISODE /* resets done so far: episode e takes its reset draws from the stream (seed, env, RNG_RESET, 128 e + k) */, I_TOTAL };
// what a forward pass needed beyond the kernel's per-leg caps (same bits as oracle/cassie_phys.h SatFlag): > 2 penetrating capsule ends on a
// leg, > 1 active joint limit on a leg, pelvis sphere / hip-pitch capsule on the floor, a left-right capsule pair in contact
enum SatFlag : int { SAT_CONTACTS = 1, SAT_LIMITS = 2, SAT_BODY_FLOOR = 4, SAT_LEG_LEG = 8 };

struct apx_env {
    apx_env_cfg cfg;
    float* st;      // [F_TOTAL, n]
    int* ist;       // [I_TOTAL, n]
    // prepared resets (apx_env_prepare_resets): RST_K ring slots per env, slot = episode % RST_K; rst = [RST_K x F_TOTAL, n] images of the state fields a reset defines up to its
    // settle step (randomised model, set_const invweights, init pose, forward-pass sensor snapshot), rst_int = [RST_K x 2, n]: the episode the slot holds (-1 none), its start phase
    float* rst; int* rst_int;
    float* wk;      // state-estimator records, [n][est::REC] env-major (estimator_lane.h); the stage hand-off itself lives in LDS
    int n;
    float* hf; int hf_nrow, hf_ncol; float hf_size[3];      // device copy of the height field (apx_env_set_hfield), or nullptr
    // apx_env_timing: hipEvent pairs around every env_step_kernel launch, on the launch stream (bench.py's roofline.achieved)
    int timing; void** ev; int ev_cap, ev_n; double t_ms; long t_launches;
    // in-rollout refill of the reset ring (apx_env_set_refill): after the auto-reset of a step, part 0 of env_reset_kernel for the next two episodes of the envs that just
    // restarted runs on the env's own side stream, next to the following env step; the next reset launch waits for it
    int refill, refill_pending, refill_due; void* side; void* ev_reset; void* ev_refill;
    // apx_rollout as one launch (env_rollout_kernel): the actor's weights in k-major order, re-laid at the start of every rollout; rollouts run that way, launches timed
    float* pol_wt; long pol_wt_n; long roll_launches; double roll_ms;
    int complete_rows; float* cp_pool;      // apx_env_set_complete_rows (default 1); the HBM tier of the complete-row path's basis pool
};

// global-address-space pointers: St is passed by value into a non-inlined device function, where the compiler could
// not otherwise prove the address space and would fall back to flat_load / flat_store
typedef __attribute__((address_space(1))) float gfloat;
typedef __attribute__((address_space(1))) int gint;
// Per-env LDS region (floats): [0,585) state fields | [586,591) int fields | [L4_WK, +WK_TOTAL) stage hand-off |
// [L4_ROWS, +632) constraint-row store (158 float4 chunks).  Stride L4_ES = 16 (mod 64): the four envs of a wave sit
// on disjoint LDS bank groups, so a 16-lane access with consecutive addresses is conflict-free.
typedef __attribute__((address_space(3))) float lfloat;
typedef __attribute__((address_space(3))) int lint;
#ifndef APX_L4_EPW
#define APX_L4_EPW 4
#endif
constexpr int L4_INT = 580 /* >= F_TOTAL */, L4_WK = 592, L4_ROWS = 1664, L4_ES = 2384, L4_EPW = APX_L4_EPW;      // envs per wave (16 lanes each)
static_assert(F_TOTAL <= L4_INT && L4_INT + I_TOTAL <= L4_WK, "LDS state region");
// Wave-constant table behind the four env regions (one copy per single-wave workgroup, filled from HBM once per launch by ct_fill):
// model constants that the stages index by LANE (body records, dof / actuator tables, mass-matrix index map).  Without it every such
// read was a global load from a __device__ table inside the 2 kHz substep.  38 144 + 2 704 = 40 848 B per workgroup: four still fit a CU.
constexpr int CT_BODY = 0 /* leg-local bodies 0..11 as (left, right) PAIRS: pos3 ipos3 quat4 inertia9 pad1, word k of leg slot sd at 40 lb + 2 k + sd (the tree stage loads a pair as one 64-bit operand of its packed arithmetic) */,
              CT_BODYSZ = 40, CT_MADR = CT_BODY + 12 * CT_BODYSZ, CT_ARM = CT_MADR + 32,
              CT_GEAR = CT_ARM + 32, CT_CMAX = CT_GEAR + 10, CT_MIDX = CT_CMAX + 10 /* 16 lanes x 7 words: 13 ushort offsets */, CT_TOTAL = CT_MIDX + 16 * 7;
static_assert(CT_TOTAL == 676 && (L4_EPW * L4_ES) % 4 == 0, "wave-constant table");
APX_DYNAMIC_LDS(float4, apx_lds4, 16);   // dynamic LDS: [env regions | wave-constant table]
__device__ __forceinline__ float ctf(int i) { return ((const lfloat*)apx_lds4)[L4_EPW * L4_ES + i]; }
// word k of the record of leg-local body lb (body id 2 + 12 sd + lb), leg slot sd
constexpr int ct_body_word(int lb, int sd, int k) { return CT_BODY + CT_BODYSZ * lb + 2 * k + sd; }
__device__ __forceinline__ int cti(int i) { return ((const lint*)apx_lds4)[L4_EPW * L4_ES + i]; }
#ifdef APX_CHECK
// `make VARIANT=check EXTRA=-DAPX_CHECK`: every S(f) / S.W(i) / S.I(f) index is range-checked; the first violation is recorded (kind, index, env, lane) in
// g_oob and the access is redirected to word 0 of its region, so the run continues and apx_env_get_field("oob") reports it (tests/test_gpu_env.py)
__device__ int g_oob[4] = {0, 0, 0, 0};
__device__ __forceinline__ int apx_chk(int idx, int lo, int hi, int kind, int env) {
    if (idx < lo || idx >= hi) { if (atomicCAS(&g_oob[0], 0, kind) == 0) { g_oob[1] = idx; g_oob[2] = env; g_oob[3] = (int)threadIdx.x; } return lo; }
    return idx;
}
#define APX_CHK(idx, lo, hi, kind) apx_chk(idx, lo, hi, kind, env)
#else
#define APX_CHK(idx, lo, hi, kind) (idx)
#endif
struct St {
    lfloat* p; int env;
    __device__ __forceinline__ lfloat& operator()(int f) const { return p[APX_CHK(f, 0, L4_INT, 1)]; }
    __device__ __forceinline__ lfloat& W(int i) const { return p[L4_WK + APX_CHK(i, 0, L4_ROWS - L4_WK, 2)]; }
    __device__ __forceinline__ lint& I(int f) const { return ((lint*)p)[L4_INT + APX_CHK(f, 0, L4_WK - L4_INT, 3)]; }
};

// terrain of cassie_hfield.xml (util/eval.py:73-76): nrow x ncol raw elevations (rows along y, columns along x) over [-sx, sx] x [-sy, sy],
// elevation = data * sz; data == nullptr: the floor plane of cassie.xml:73
struct Hf { const float* data; int nrow, ncol; float sx, sy, sz; };
constexpr int RST_K = 2;
struct Cfg { Hf hf; int simrate, dyn_rand, stance_mode, incentive, max_traj_len, pgs_iters; unsigned seed_lo, seed_hi, env_base; int reward_kind, env_kind, command_profile, obs_dim, est_lifetime, input_profile; float* est /* state-estimator records, [n][est::REC] (estimator_lane.h) */; const float* rst; const int* rst_int; /* prepared resets or nullptr */ float* complete_pool; /* non-NULL: a pass beyond the lane map's row caps is solved with its complete row set (cassie_complete.h; the pointer = the HBM tier of its basis pool, [n][96 x 24]); NULL: capped, counted in I_SAT (rounds 1-4) */ };

// ------------------------------------------------------------------------------------------------ Philox4x32-10
__device__ __forceinline__ unsigned philox(unsigned k0, unsigned k1, unsigned env, unsigned ctr, unsigned dom) {
    unsigned c0 = ctr, c1 = env, c2 = 0x41505845u, c3 = dom;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
        const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1, n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return c0;
}
// stream domains (4th counter word): 0 = the per-step command draws, ctr = the env's running counter I_RNG; 1 = reset draws, ctr = 128 * episode + k: a reset is a
// function of (seed, env, episode index) alone, so the next episode's randomised model can be prepared ahead of time (env_reset_kernel, part 0)
constexpr unsigned RNG_STEP = 0u, RNG_RESET = 1u, RNG_RESET_BLOCK = 128u, RNG_RESET_TAIL = 126u;
struct Rng {
    unsigned k0, k1, env, ctr, dom;
    __device__ __forceinline__ unsigned u32() { return philox(k0, k1, env, ctr++, dom); }
    __device__ __forceinline__ float u01() { return ((float)(u32() >> 8) + 0.5f) * (1.0f / 16777216.0f); }
    __device__ __forceinline__ float uniform(float a, float b) { return a + (b - a) * u01(); }
    __device__ __forceinline__ unsigned randint(unsigned n) { return (unsigned)(((unsigned long long)u32() * n) >> 32); }
};

