"""Worker of the fp32-control parity test (run as a subprocess with ORC_REAL=float): loads teacher-forcing records produced by the fp64 oracle
(state before an env step + action), replays each env step in the fp32 build of the SAME oracle sources from the identical state, and writes observation,
reward, done flag and row-set hash.  usage: ORC_REAL=float python tests/fp32_control_worker.py records.npz out.npz"""
import os
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main(src, dst):
    from oracle import sim as S
    from tests.state_xfer import ORACLE_STATE_FIELDS, oracle_load_state
    assert S._F32, "run with ORC_REAL=float"
    with np.load(src) as z:
        g = {k: z[k] for k in z.files}      # materialised once: NpzFile re-reads the zip member on every access and is not thread-safe
    n_env, n_step = g["action"].shape[:2]
    obs = np.zeros((n_env, n_step, g["obs"].shape[-1])); rew = np.zeros((n_env, n_step)); done = np.zeros((n_env, n_step), dtype=np.int64); hsh = np.zeros((n_env, n_step), dtype=np.int64)

    def run(i):
        e = S.OracleEnv(seed=int(g["seed"]), env_id=i)
        e.reset()
        for t in range(n_step):
            d = {k: g["st_" + k][i, t] for k in ORACLE_STATE_FIELDS}; d["ints"] = g["st_ints"][i, t]
            oracle_load_state(e, d)
            o, r, dn = e.step(g["action"][i, t])
            ii = e.get("ints")
            obs[i, t] = o; rew[i, t] = r; done[i, t] = dn; hsh[i, t] = int(ii[10]) | int(ii[11]) << 16
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        list(ex.map(run, range(n_env)))
    np.savez(dst, obs=obs, rew=rew, done=done, hash=hsh)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
