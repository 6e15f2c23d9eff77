"""Algorithmic work of the dominant kernel, used by bench.py's `roofline` object.

ENV_STEP_BYTES: HBM bytes one env step MUST move with all 50 substeps fused in one launch = read + write of the
persistent per-env state (F_TOTAL fp32 fields + 5 int fields, env.hip `enum Field`) + action in, obs/reward/done out.
ENV_STEP_FLOP: fp32 operations of one env step of the tree-sparse formulation, counted analytically per substep
 x 50.
"""
F_TOTAL = 580          # floats of persistent state per env (env_state.h: enum Field)
I_TOTAL = 10         # int fields per env (env_state.h: enum IField)
EST_REC = 168          # state-estimator record per env (estimator_lane.h): read + written once per env step at least (it moves through L2 every substep)
ENV_STEP_BYTES = 2 * 4 * (F_TOTAL + I_TOTAL + EST_REC) + 4 * 10 + 4 * 50 + 4 + 1


def rollout_bytes_per_env(T, D=50, A=10):
    """Algorithmic HBM bytes per env of ONE env_rollout_kernel launch (round 5: T env steps per launch): the persistent state and the estimator record cross HBM once per
    launch instead of once per step; per step: observation out (the next step reads it from LDS), mean + action out, noise in, reward + done out."""
    return 2 * 4 * (F_TOTAL + I_TOTAL + EST_REC) + 4 * D + T * (4 * D + 2 * 4 * A + 4 * A + 4 + 1)

# per-substep fp32 op count (multiply-add = 2), typical walking state: 12 equality rows + 2 contacts (8 rows) = 20 rows
_FK = 25 * 95 + 32 * 12
_INERTIA = 25 * 110
_VEL_RNE = 32 * 40 + 25 * 130 + 25 * 12 + 32 * 11
_CRBA = 25 * 10 + 32 * 45 + 307 * 11
_FACTOR = 2 * 2 * 1750              # two sparse LDL factorisations
_SOLVES = 2 * 5 * 307               # u~, v~, w~, qacc, Euler rhs/solve
_ROWS = 20 * (2 * 60 + 2 * 100 + 2 * 19 * 4)     # Jacobian + whitening + commit dots
_PGS = 50 * 20 * (2 * 19 * 2 + 8)
_MISC = 1500
_EST = 2 * 2 * 13 * 8 + 14 * 60 + 475 + 170      # state estimator in its lane form (estimator_lane.h): two closure evaluations per leg, leg kinematics, three Kalman filters, force solve / IMU
SUBSTEP_FLOP = _FK + _INERTIA + _VEL_RNE + _CRBA + _FACTOR + _SOLVES + _ROWS + _PGS + _MISC + _EST
ENV_STEP_FLOP = 50 * SUBSTEP_FLOP
# The figure the roofline uses: INSTRUMENTED count of the fp64 CPU restatement (oracle/cassie_phys.cpp counts every multiply / add where it
# happens, skipping structural zeros of its dense loops), mean over 20 env steps of a random-action rollout with
# resets: `python -c "from oracle import sim; print(sim.count_flops(20))"` -> 7 286 544 - the call bench.py's cpu_baseline leg makes, so the constant and the
# re-measured figure are ONE number (round 4 carried 7.17e6 here, a count from before the estimator's last additions, next to the re-measured 7.29e6 in the
# same JSON object).  Used when the cpu_baseline leg is skipped (--no_cpu_baseline, N > 1); bench.py passes whichever it used to every derived quantity.
ENV_STEP_FLOP_COUNTED = 7_286_544
