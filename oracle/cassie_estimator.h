// CPU ORACLE (test infrastructure only — see oracle/__init__.py): fp64 restatement of the reference's state estimator,
// `state_output_step` of libcassiemujoco.so (interface: cassie/cassiemujoco/include/StateOutput.h:33-36; consumed by
// CassieEnv.get_full_state, cassie/cassie.py:817-850).  The routine exists only as machine code; it was decoded from the unstripped
// binary (disassembly + direct calls of its internal routines + dumps of its 0x1070-byte state object, tools/refprobe/decode_estimator/)
// and is PINNED: golden G11 is the binary's own output on a 3000-substep sensor stream, reproduced here to < 1e-7 on every filtered
// output incl. start-up (tests/test_oracle_env.py::test_g11_state_estimator_restated).
//
// What the routine is (addresses = offsets inside libcassiemujoco.so):
//   1. leg kinematics from the SENSOR angles (0x22730): foot frame relative to the pelvis through hip roll / yaw / pitch, knee, shin,
//      tarsus and the foot MOTOR angle (the chain of cassie.xml:89-140 with a fixed offset inside the foot body);
//   2. heel-spring deflection of both legs by a warm-started Levenberg-Marquardt solve (Madsen / Nielsen, tau 1e-3, 5 iterations,
//      eps 1.49e-8) of the achilles-rod closure, a 14-term cosine series in (knee, shin, tarsus, heel) (0x18e00, 0x19510, 0x22a93-0x2316d);
//   3. foot force in the pelvis frame from the two leg-spring torques (1500 shin, 1250 heel N m / rad) through the spring Jacobian with the
//      tarsus following the rod closure, solved like MATLAB's `\` for a 2 x 3 system: QR with column pivoting, basic solution (0x21000);
//   4. two horizontal extended Kalman filters (x, y; 0x1cd10) on a linear-inverted-pendulum model, state [pelvis p, v, left foot, right
//      foot, load share alpha, disturbance force], measurements [p - foot L, p - foot R, alpha = FL / (FL + FR), v + dt a_imu], and one
//      vertical Kalman filter (inline, 0x2a858-0x2c607), state [z, vz, left foot z, right foot z, disturbance force], input
//      (FL + FR) / m - g; foot process noise 1e-6 in swing (force < 50 N) and 1e-10 in stance;
//   5. terrain height = first-order low-pass (0.9995 / 0.0005 per 2 kHz sample) of the load-weighted kinematic foot height while loaded;
//   6. translationalAcceleration = accelerometer - R^T (0, 0, 9.806) - w x (w x r_imu), unfiltered (round 2, golden G11 as well).
#pragma once
#include "cassie_phys.h"

namespace orc {

struct EstSensors { double mpos[10], jpos[6], quat[4], gyro[3], acc[3]; };   // cassie_out_t fields the routine reads (drive / joint positions, vectorNav)

struct StateOutput {
    // persistent state (survives cassie_sim_set_const; cleared by state_output_setup = cassie_sim_full_reset)
    double heel[2];               // heel-spring deflection L, R: warm start of the next solve
    double hx[2][6], hP[2][36];   // horizontal filters (x, y): state and covariance (row-major)
    double zx[5], zP[25];         // vertical filter
    double terrain;               // low-passed terrain height
    int inited;
    // outputs of the most recent step (state_out_t fields)
    double pos[3], vel[3], tacc[3];          // pelvis.position, pelvis.translationalVelocity (world-aligned axes), pelvis.translationalAcceleration
    double foot_rel[2][3];                   // leftFoot / rightFoot .position (pelvis frame)
    double foot_quat[2][4];                  // leftFoot / rightFoot .orientation (pelvis frame; the foot body's frame turned by the routine's constant offset)
    double foot_force[2][3];                 // estimated foot force, world z exact, x / y in the heading frame of the binary not reproduced (unused)
    int lm_iters;                            // Levenberg-Marquardt iterations of the most recent heel solve (diagnostics)
    int sw;                                  // discrete switches of the most recent step: bit 0 / 1 = left / right foot load below 50 N (process noise of the foot states), bit 2 = terrain update (load > 1 N); part of the row-set signature
};

void state_output_setup(StateOutput& s);
void state_output_step(StateOutput& s, const EstSensors& in);
// pieces exposed for the unit tests
double heel_residual(double knee, double shin, double tarsus, double heel, double* grad4 /* d/d(knee, shin, tarsus, heel) or nullptr */);
void heel_solve(double heel[2], const double legL[3], const double legR[3], int* iters);
void mldivide23(const double M[2][3], const double tau[2], double x[3]);
// one step of a horizontal filter (0x1cd10; x[6], P[36] row-major, in place): measurements zL = p - foot L, zR = p - foot R, foot loads fl, fr >= 0, IMU acceleration
void hfilter_step(double* x, double* P, double zL, double zR, double fl, double fr, double acc);
// one step of the vertical filter (inline in the binary, 0x2a858-0x2c607; x[5], P[25])
void zfilter_step(double* x, double* P, double zL, double zR, double fl, double fr);

}  // namespace orc
