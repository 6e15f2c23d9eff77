"""A/B record for the env kernel's -ffast-math build flag: the same seeded 12-step rollout of 64 envs in the HIP env
and in the fp64 oracle, once per library build; reports the error against the oracle per observation group and the kernel time.
usage (GPU box, repo root):  python tools/t_fastmath_ab.py            -> runs itself once per library and prints one JSON object
builds:  make -C apex_amd/csrc                         (product: -ffast-math on env.o)
         make -C apex_amd/csrc VARIANT=nofm FASTMATH=  (IEEE build: lib/libapx_nofm.so)"""
import json, os, subprocess, sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GROUPS = {"pose_and_motor_pos": slice(0, 15), "pelvis_vel": slice(15, 21), "motor_vel": slice(21, 31), "pelvis_acc": slice(31, 34),
          "joint_pos": slice(34, 40), "joint_vel": slice(40, 46)}


def one():
    sys.path.insert(0, ROOT)
    import torch
    from apex_amd.vecenv import CassieVecEnv
    from oracle import sim as S
    N, T = 64, 12
    genv = CassieVecEnv(n_envs=N, dynamics_randomization=True, seed=3)
    oenv = [S.OracleEnv(dyn_rand=True, seed=3, env_id=i) for i in range(N)]
    genv.reset(); [e.reset() for e in oenv]
    rng = np.random.RandomState(0)
    err = {k: [] for k in GROUPS}; rew_err = []
    for t in range(T):
        act = (rng.randn(N, 10) * 0.15).astype(np.float32)
        obs, rew, done, fin = genv.step(torch.tensor(act, device=genv.device), auto_reset=False)
        obs, rew = obs.cpu().numpy().astype(np.float64), rew.cpu().numpy()
        ref = [e.step(act[i].astype(np.float64)) for i, e in enumerate(oenv)]
        o = np.stack([r[0] for r in ref]); r = np.array([r[1] for r in ref])
        for k, sl in GROUPS.items():
            err[k].append(float(np.abs(obs[:, sl] - o[:, sl]).max()))
        rew_err.append(float(np.abs(rew - r).max()))
    # kernel time on 4096 envs (the bench geometry), hipEvents around the launches
    big = CassieVecEnv(n_envs=4096, seed=1); big.reset()
    a = torch.zeros(4096, 10, device=big.device)
    for _ in range(3):
        big.step(a)
    big.kernel_timing(True); big.kernel_timing_read()
    for _ in range(20):
        big.step(torch.randn(4096, 10, device=big.device) * 0.1)
    ms, n = big.kernel_timing_read()
    print(json.dumps({"lib": os.environ.get("APX_LIB", "libapx.so"), "max_abs_err_vs_fp64_oracle_at_step_1_6_12": {k: [v[0], v[5], v[11]] for k, v in err.items()},
                      "reward_err_step_1_6_12": [rew_err[0], rew_err[5], rew_err[11]], "env_step_kernel_ms": round(ms / n, 4)}))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "one":
        one(); sys.exit(0)
    out = {}
    for name, lib in (("fast_math (shipped)", None), ("ieee (-ffast-math off)", os.path.join(ROOT, "apex_amd", "lib", "libapx_nofm.so"))):
        env = dict(os.environ)
        if lib:
            env["APX_LIB"] = lib
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "one"], env=env, capture_output=True, text=True, cwd=ROOT)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        out[name] = json.loads(line[-1]) if line else {"error": r.stderr[-500:]}
    print(json.dumps(out, indent=1))
