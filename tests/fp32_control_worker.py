"""Worker of the fp32-control parity tests (run as a subprocess with ORC_REAL=float): loads teacher-forcing records produced by the fp64 oracle
(state before an env step + action), replays each env step in the fp32 build of the SAME oracle sources from the identical state, and writes observation,
reward, done flag, row-set hash, drive torques, motor positions, qpos and qvel.
usage: ORC_REAL=float python tests/fp32_control_worker.py records.npz out.npz [scenario name of tests/tf_scenarios.py; default: Cassie-v0 with dynamics randomisation]"""
import os
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main(src, dst, scenario=None):
    from oracle import sim as S
    from tests.state_xfer import ORACLE_STATE_FIELDS, oracle_load_state
    assert S._F32, "run with ORC_REAL=float"
    with np.load(src) as z:
        g = {k: z[k] for k in z.files}      # materialised once: NpzFile re-reads the zip member on every access and is not thread-safe
    n_env, n_step = g["action"].shape[:2]
    sc = None
    if scenario:
        from tests.tf_scenarios import BY_NAME
        sc = BY_NAME[scenario]
    obs = np.zeros((n_env, n_step, g["obs"].shape[-1])); rew = np.zeros((n_env, n_step)); done = np.zeros((n_env, n_step), dtype=np.int64); hsh = np.zeros((n_env, n_step), dtype=np.int64)
    tq = np.zeros((n_env, n_step, 10)); mp = np.zeros((n_env, n_step, 10)); qp = np.zeros((n_env, n_step, 35)); qv = np.zeros((n_env, n_step, 32))
    envs = sc.make_oracle(S, n_env) if sc else None      # (the scenario's own preparation, in fp32: every state word is overwritten below)

    def run(i):
        if envs is None:
            e = S.OracleEnv(seed=int(g["seed"]), env_id=i); e.reset()
        else:
            e = envs[i]
        for t in range(n_step):
            d = {k: g["st_" + k][i, t] for k in ORACLE_STATE_FIELDS}; d["ints"] = g["st_ints"][i, t]
            oracle_load_state(e, d)
            if sc is None:
                o, r, dn = e.step(g["action"][i, t])
            else:
                o, r, dn = sc.step_oracle(e, g["action"][i, t])
            ii = e.get("ints")
            obs[i, t] = o; rew[i, t] = r; done[i, t] = dn; hsh[i, t] = int(ii[10]) | int(ii[11]) << 16
            tq[i, t] = e.get("so_torque"); mp[i, t] = e.get("so_mpos"); qp[i, t] = e.get("qpos"); qv[i, t] = e.get("qvel")
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        list(ex.map(run, range(n_env)))
    np.savez(dst, obs=obs, rew=rew, done=done, hash=hsh, torque=tq, mpos=mp, qpos=qp, qvel=qv)


if __name__ == "__main__":
    main(*sys.argv[1:])
