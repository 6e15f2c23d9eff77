"""The oracle's push sweep against the MuJoCo table the reference ships (golden G24) on a lattice of cells, written as a table:
    python tools/g24_cells.py [every_nth_direction=10] [out=profiles/r06_g24_cells.json] [procs=8]
10 -> directions 0, 10, .., 90 x all 28 phases = 280 cells (about 12 minutes on 8 cores).  Test infrastructure (uses oracle/)."""
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "tests")]


def summary(mine, ref):
    mine, ref = np.asarray(mine, np.float64), np.asarray(ref, np.float64)
    d = np.abs(mine - ref)
    return {"cells": int(mine.size), "oracle_mean_N": round(float(mine.mean()), 2), "mujoco_mean_N": round(float(ref.mean()), 2),
            "mean_rel": round(float(mine.mean() / ref.mean() - 1.0), 4), "corr_cells": round(float(np.corrcoef(mine.ravel(), ref.ravel())[0, 1]), 4),
            "mean_abs_diff_N": round(float(d.mean()), 2), "max_abs_diff_N": float(d.max()), "identical": int((d == 0).sum()),
            "within_10N": int((d <= 10).sum()), "within_20N": int((d <= 20).sum())}


def main():
    import ref_policy_eval as R
    nth = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    out = sys.argv[2] if len(sys.argv) > 2 else os.path.join(REPO, "profiles", "r06_g24_cells.json")
    procs = int(sys.argv[3]) if len(sys.argv) > 3 else min(8, os.cpu_count() or 1)
    tag = sys.argv[4] if len(sys.argv) > 4 else "a"
    g = R.fixture()
    cells = [(tag, a, p) for a in range(0, 100, nth) for p in range(28)]
    t0 = time.time()
    with mp.get_context("fork").Pool(procs) as pool:
        res = pool.map(R.push_cell, cells, chunksize=1)
    mine = np.array([r[3] for r in res]).reshape(-1, 28)
    ref = np.array([g[f"{tag}_eval_perturbs"][a, p] for _, a, p in cells]).reshape(-1, 28)
    doc = {"what": "largest 0.2 s pelvis push survived [N] per (direction, gait phase): the reference's shipped policy '%s' on the fp64 oracle vs the reference's own MuJoCo table (eval_perturbs.npy), protocol of tools/eval_perturb.py:97-160" % tag,
           "directions": list(range(0, 100, nth)), "phases": list(range(28)), "oracle": mine.astype(int).tolist(), "mujoco": ref.astype(int).tolist(),
           "summary": summary(mine, ref), "direction_mean_corr": round(float(np.corrcoef(mine.mean(1), ref.mean(1))[0, 1]), 4), "seconds": round(time.time() - t0, 1), "procs": procs}
    json.dump(doc, open(out, "w"), indent=1)
    print(json.dumps(doc["summary"]), doc["direction_mean_corr"])


if __name__ == "__main__":
    main()
