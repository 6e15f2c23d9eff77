// CPU ORACLE (test infrastructure only — see oracle/__init__.py): fp64 rigid-body step for the Cassie model.
//
// Restates, for the subset of features cassie.xml uses, the pipeline behind the reference's `cassie_sim_step_pd`
// -> MuJoCo 2.00 `mj_step` (cassie/cassiemujoco/cassiemujoco.py:46-49).  MuJoCo 2.00 itself is a
// closed-source, licence-gated third-party binary that is absent from /root/reference, so this follows MuJoCo's
// PUBLISHED algorithm description (Computation chapter: kinematics, CRBA, RNE, soft-constraint model with
// solref/solimp impedance, pyramidal friction cones, PGS on the dual, semi-implicit Euler with implicit joint
// damping) and is anchored on the model file (cassie.xml) and the reference's call sites.
// PARITY: no reference test or runnable binary pins a physics STATE (no per-step MuJoCo fixture exists: "parity unpinned" at that level).  What the
// reference's own data files pin since round 5 (tests/test_oracle_env.py::test_g23_* / test_g24_*): the smooth dynamics against the recorded torques of an external 2 kHz
// simulation of the robot (G23, a few per cent), and closed-loop outcomes against a table the reference generated UNDER MUJOCO - its push sweep of its shipped policy (G24:
// mean -2.7 %, correlation 0.94 over 40 cells, 11.5 N mean difference at 10 N resolution).
//
// Deliberately written dense and simple (32x32 mass matrix, dense Jacobians, Cholesky) so that it shares no
// structure with the HIP kernel it checks.
#pragma once
#include <cmath>
#include <cstring>
#include <algorithm>
#include "cassie_model_gen.h"

namespace orc {

constexpr int NB = CM_NBODY, NV = CM_NV, NQ = CM_NQ, NJ = CM_NJNT, NG = CM_NGEOM, NEQ = CM_NEQ, NU = CM_NU;
// The oracle instantiates EVERY constraint cassie.xml can produce (cassie.xml:18-35,73,87,101,119-144 and the limited joints): all
// active joint limits, both ends of the foot / tarsus / shin / hip-pitch capsules and the pelvis sphere against the floor (condim 3,
// pyramidal), and the 3 x 3 left-right capsule pairs (condim 1, frictionless).  The HIP kernel keeps the first KERNEL_MAXCON_LEG floor
// contacts (order foot, tarsus, shin), the first KERNEL_MAXLIM_LEG limit per leg and the first KERNEL_MAXLEGLEG leg-leg pairs; `State::sat` reports when a
// substep needed more, so that the cap is a checked property of a rollout, not an assumption.
constexpr int KERNEL_MAXCON_LEG = 2, KERNEL_MAXLIM_LEG = 1, KERNEL_MAXLEGLEG = 3;
constexpr int MAXCON_LEG = 8, MAXLIM_LEG = 8;      // 4 capsules x 2 ends; 8 limited joints per leg
constexpr int MAXCON = 2 * MAXCON_LEG + 1, MAXLIM = 2 * MAXLIM_LEG, MAXCON1 = 9;      // + pelvis sphere; 9 frictionless leg-leg contacts
constexpr int MAXEFC = 3 * NEQ + MAXLIM + 4 * MAXCON + MAXCON1;
enum SatFlag { SAT_CONTACTS = 1, SAT_LIMITS = 2, SAT_BODY_FLOOR = 4, SAT_LEG_LEG = 8 };
constexpr double MINVAL = 1e-15;
constexpr double DT = 0.0005;                      // cassie.xml:5
constexpr double GRAV = 9.81;

struct V3 { double x, y, z; };
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator*(V3 a, double s) { return {a.x * s, a.y * s, a.z * s}; }
inline double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline V3 v3(const double* p) { return {p[0], p[1], p[2]}; }
inline double norm(V3 a) { return std::sqrt(dot(a, a)); }

struct Q4 { double w, x, y, z; };
inline Q4 qmul(Q4 a, Q4 b) {
    return {a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z, a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
            a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x, a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w};
}
inline Q4 qnormalize(Q4 q) {
    double n = std::sqrt(q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z);
    if (n < MINVAL) return {1, 0, 0, 0};
    return {q.w / n, q.x / n, q.y / n, q.z / n};
}
inline Q4 qaxisangle(V3 axis, double ang) {
    double s = std::sin(ang * 0.5);
    return {std::cos(ang * 0.5), axis.x * s, axis.y * s, axis.z * s};
}
struct M3 { double m[9]; };   // row-major
inline M3 q2m(Q4 q) {
    double w = q.w, x = q.x, y = q.y, z = q.z;
    return {{1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y), 2 * (x * y + w * z),
             1 - 2 * (x * x + z * z), 2 * (y * z - w * x), 2 * (x * z - w * y), 2 * (y * z + w * x),
             1 - 2 * (x * x + y * y)}};
}
inline V3 mul(const M3& R, V3 v) {
    return {R.m[0] * v.x + R.m[1] * v.y + R.m[2] * v.z, R.m[3] * v.x + R.m[4] * v.y + R.m[5] * v.z,
            R.m[6] * v.x + R.m[7] * v.y + R.m[8] * v.z};
}
inline V3 col(const M3& R, int k) { return {R.m[k], R.m[3 + k], R.m[6 + k]}; }

// spatial motion vector about the reference point o: [angular; linear velocity of the body point at o]
struct SV { V3 a, l; };
inline SV operator+(SV p, SV q) { return {p.a + q.a, p.l + q.l}; }
inline SV operator*(SV p, double s) { return {p.a * s, p.l * s}; }
inline SV crossMotion(SV v, SV s) { return {cross(v.a, s.a), cross(v.a, s.l) + cross(v.l, s.a)}; }
inline SV crossForce(SV v, SV f) { return {cross(v.a, f.a) + cross(v.l, f.l), cross(v.a, f.l)}; }
inline double sdot(SV m, SV f) { return dot(m.a, f.a) + dot(m.l, f.l); }

// spatial inertia about o (world axes): mass, first moment h = m*r, rotational inertia about o (symmetric 3x3)
struct SI { double m; V3 h; double I[6]; };   // I: xx yy zz xy xz yz
inline SI operator+(SI p, SI q) {
    SI r{p.m + q.m, p.h + q.h, {}};
    for (int i = 0; i < 6; ++i) r.I[i] = p.I[i] + q.I[i];
    return r;
}
inline V3 symmul(const double* I, V3 v) {
    return {I[0] * v.x + I[3] * v.y + I[4] * v.z, I[3] * v.x + I[1] * v.y + I[5] * v.z,
            I[4] * v.x + I[5] * v.y + I[2] * v.z};
}
// force = I * motion  ->  [moment about o; force]
inline SV imul(const SI& s, SV v) { return {symmul(s.I, v.a) + cross(s.h, v.l), v.l * s.m - cross(s.h, v.a)}; }

struct Params {                  // per-env model parameters touched by dynamics randomisation (cassie.py:568-657)
    double mass[NB];
    double damping[NV];
    double friction;             // sliding friction of every geom (floor wins by priority, cassie.xml:73)
    Q4 floor_quat;               // set_geom_quat(floor), cassie.py:644-648
    double body_invweight0[NB][2];
    double dof_invweight0[NV];
    int pgs_iters;
    // mjOption.tolerance: MuJoCo's PGS stops once the scaled cost improvement of a sweep falls below it (default 1e-8, cassie.xml:5 leaves it).
    // The HIP kernel always runs opt.iterations = 50 sweeps (four envs share a wave: a per-env exit saves nothing), so the oracle's default
    // is 0 = never stop early.  With 1e-8 the solver stops after 35 sweeps on average on a walking gait (12 % of the passes use all 50) and
    // the 150-step return of the trained policy moves by < 0.1 % (tests/test_oracle_env.py::test_solver_tolerance_knob): the extra sweeps
    // only tighten the same fixed point.
    double tolerance = 0;
    double meaninertia = 1;      // mjModel.stat.meaninertia = trace(M(qpos0)) / nv, recomputed by set_const like body_invweight0
    // terrain (cassie_hfield.xml:69,74, util/eval.py:73-76): when hf_data != nullptr the floor is a height field instead of the plane:
    // hf_nrow x hf_ncol samples (row = y, column = x) over [-hf_size[0], hf_size[0]] x [-hf_size[1], hf_size[1]], elevation = data * hf_size[2]
    const float* hf_data = nullptr; int hf_nrow = 0, hf_ncol = 0; double hf_size[3] = {0, 0, 0};
    int kernel_caps = 0;         // 1: instantiate only what the HIP kernel instantiates (first KERNEL_MAX* per leg, no pelvis / hip-pitch / leg-leg rows);
                                 // `State::sat` is reported either way, so a test can check the kernel's arithmetic on a saturated state AND its flag
};

struct Row { double J[NV]; double pos, vel, R, aref, diag; int type; };   // type 0 equality, 1 limit, 2 contact

struct State {
    double qpos[NQ], qvel[NV], qacc_warm[NV];
    // products of the most recent forward pass (evaluated at the PRE-integration state, like mjData after mj_step)
    V3 xpos[NB]; Q4 xquat[NB]; M3 xmat[NB];
    double qacc[NV];
    int ncon, nefc;
    double efc_force[MAXEFC];
    int efc_type[MAXEFC];        // 0 equality, 1 limit, 2 contact (pyramid edge or frictionless)
    double foot_force[2][3];     // world contact force on the left / right foot body
    double sens_acc[3];          // accelerometer at the imu site (cassie.xml:267), sensor frame
    double sens_gyro[3];
    double con_dist[MAXCON]; int con_geom[MAXCON];
    V3 con_frame[MAXCON][3];     // contact frames (normal, two tangents) of the floor contacts of the most recent forward pass
    int sat;                     // SatFlag bits of the most recent forward pass: the constraint set exceeded what the HIP kernel instantiates
    int solver_iter;             // sweeps the PGS solver ran in the most recent forward pass (mjData.solver_iter)
    int ncon1;                   // leg-leg (frictionless) contacts of the most recent forward pass
    unsigned rowsig[2];          // row-set signature of the most recent forward pass (what was DETECTED, before any cap): [0] = limited joints out of range (8 bits per leg) |
                                 // penetrating foot / tarsus / shin capsule ends (6 per leg) << 16 | (pelvis sphere, any hip-pitch capsule) on the floor << 28; [1] = the 9 left x right pairs
    double xfrc[6] = {0, 0, 0, 0, 0, 0};   // one row of mjData.xfrc_applied: world force xyz, torque xyz, applied at the COM of body xfrc_body
    int xfrc_body = 1;                     // the body of that row (1 = cassie-pelvis, the harnesses' default; one pushed body at a time)
};

struct Work {
    SV cdof[NV], cdofdot[NV], cvel[NB];
    SI cinert[NB], crb[NB];
    double M[NV][NV], L[NV][NV];
    double bias[NV], passive[NV], smooth[NV];
    Row rows[MAXEFC];
    V3 anchor[NJ];
    V3 o;
};

// instrumented operation count: floating-point multiplies + adds of the restatement, counted where they happen and only
// when both operands are structurally nonzero (the dense loops below run over zeros that a tree-sparse implementation never touches)
extern thread_local unsigned long long g_flops;
void default_params(Params& p);
void set_const(Params& p);                                        // mj_setConst subset: invweight0 at qpos0
void reset_state(State& s);                                       // cassie_sim_set_const: init qpos, zero qvel
void forward(const Params& p, State& s, Work& w, const double* ctrl);   // mj_forward (ctrl = actuator-side torque)
void euler(const Params& p, State& s, Work& w);                   // mj_Euler with implicit joint damping
inline void step(const Params& p, State& s, Work& w, const double* ctrl) { forward(p, s, w, ctrl); euler(p, s, w); }

// diagnostics used by the invariant tests
// surface under (x, y): elevation and unit normal of the height-field triangle there (flat plane z = floor pos when no height field is set)
void floor_query(const Params& p, double x, double y, double& h, V3& n);
double constraint_violation(const State& s);                      // max |p1-p2| over the 4 connect constraints
void com_velocity(const Params& p, const State& s, Work& w, double out[3]);   // total linear momentum / total mass
double total_energy(const Params& p, const State& s, Work& w);    // kinetic + gravity + spring potential
// total momentum of the tree from the free joint's rows of M qvel: out = linear momentum [3], angular momentum about the pelvis origin in world axes [3], pelvis
// origin [3], total mass, centre of mass [3], then per foot the capsule centre [3] and axis [3] (tests/test_oracle_env.py: the momentum balance of Agility's gait, G23)
void momentum(const Params& p, const State& s, Work& w, double out[25]);
// inverse dynamics of the unconstrained tree: out = M(q) qacc + bias(q, qvel) - passive(q, qvel) (the generalised force that actuators + constraints must supply)
void inverse_dynamics(const Params& p, const State& s, Work& w, const double* qacc, double* out);

}  // namespace orc
