"""bench.py — env-steps/sec (whole job) of the Cassie-v0 PPO hot path on N MI355X GPUs.

One "step" = one PPO iteration on every rank: a T-step lock-step rollout of 4096 envs per GPU (policy/value forward,
50 x 2 kHz physics substeps per env step, reward, observation), the discounted-return scan, advantage normalisation
and `epochs` passes of clipped-ratio/value/mirror minibatch updates (Adam, global-norm clip).  Inputs are resident in
HBM when the timed region starts.  Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
# test hook (tests/test_kernel_emulation_env.py runs the three workloads' code paths, JSON assembly included, on the emulated kernels): the TD3 and recurrent workloads shrink to 64 envs and 2 - 3 steps
# (their BASELINE shapes are fixed in the code), the observation statistics pass is skipped; the line's config carries "tiny_test_shape": true.  Never set for a measurement.
TINY = os.environ.get("APX_BENCH_TINY") == "1"
VALU_PEAK_TFLOPS = 157.3       # fp32 vector peak = fp32 MFMA peak (MI355X_MICROARCH.md)


def cpu_baseline(seconds_hint=12.0, n_envs=4096, rollout_len=32, minibatch=16384, epochs=3):
    """The fp64 C++ oracle (kind "port") timed on this host's cores, bounded sample: sampling (one env per thread) and, beside it, the
    end-to-end rate of one PPO iteration of the same shape (sampling at that rate + the numpy fp64 learner of oracle/learner.py timed on
    ONE minibatch and scaled to the iteration's minibatch count).  Also returns the instrumented flop count of an env step."""
    from oracle import sim, learner as L
    from apex_amd.vecenv import MIRRORED_OBS, MIRRORED_ACTS
    cores = os.cpu_count() or 1
    probe = sim.rollout_bench(cores, 4, cores)                        # env-steps/s on a tiny probe
    n_steps = int(max(8, min(400, seconds_hint * probe / cores)))
    v = sim.rollout_bench(cores, n_steps, cores)
    flop = sim.count_flops(20)
    # learner: one minibatch of the bench shape through the fp64 numpy oracle (BLAS threads as numpy is configured)
    rng = np.random.RandomState(0)
    H, mb = 256, minibatch                 # one FULL minibatch of the bench shape is timed (about 1 s of numpy per 16 384 rows)
    shapes = [(H, 50), (H,), (H, H), (H,), (10, H), (10,)]
    Wa = [rng.randn(*s) * 0.05 for s in shapes]; Wc = [rng.randn(*s) * 0.05 for s in shapes[:4] + [(1, H), (1,)]]
    obs = rng.randn(mb, 50); ph = rng.rand(mb) * 6.28; obs[:, 46] = np.sin(ph); obs[:, 47] = np.cos(ph); act = rng.randn(mb, 10) * 0.3; ret = rng.randn(mb, 1); adv = rng.randn(mb, 1)
    class Keep:
        def step(self, p, g): return p
    t0 = time.time()
    L.ppo_update(Wa, Wa, Wc, Keep(), Keep(), obs, act, ret, adv, np.zeros(50), np.ones(50), float(np.exp(-1.5)), M_obs=L.mirror_matrix(MIRRORED_OBS), M_act=L.mirror_matrix(MIRRORED_ACTS))
    t_mb = (time.time() - t0) * (minibatch / mb)
    steps_it = n_envs * rollout_len
    t_update = t_mb * epochs * (steps_it // minibatch)
    e2e = steps_it / (steps_it / v + t_update)
    return {"value": round(v, 1), "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": f"{cores} envs x {n_steps} env steps (50 substeps each), sampling only, fp64 dense oracle, one env per thread",
            "end_to_end": {"value": round(e2e, 1), "unit": "env-steps/s",
                           "sample": f"one iteration of {steps_it} env steps at the sampling rate above + {epochs} epochs x {steps_it // minibatch} minibatches of {minibatch} "
                                     f"through the fp64 numpy learner (one full {mb}-row minibatch timed: {t_mb:.2f} s; the iteration's update = that x {epochs * (steps_it // minibatch)})"},
            "flop_per_env_step_counted": int(flop)}



def kernel_source_hash():
    """sha1 over the env kernel's sources: a PMC profile is only quoted while it was taken on THIS kernel"""
    import hashlib
    h = hashlib.sha1()
    for f in ("env.hip", "cassie_lane.h", "cassie_complete.h", "estimator_lane.h", "cassie_common.h", "env_state.h", os.path.join("gfx950", "lane_ops.h")):
        h.update(open(os.path.join(REPO, "apex_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:12]


def kernel_isa_hash():
    """sha1 over the INSTRUCTION STREAM of the env code object of the built library (the largest gfx950 object of libapx.so, disassembled, addresses stripped), or None
    without the ROCm binutils.  profiles/kernel_identity.json maps the source hash a profile was taken on to this hash: a refactor that leaves every instruction in place
    (round 6: the inline assembly moved into gfx950/lane_ops.h, emulation marks that expand to nothing) keeps the profile quotable."""
    import hashlib, shutil, subprocess, tempfile
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    lib = os.path.join(REPO, "apex_amd", "lib", "libapx.so")
    if not (os.path.exists(objdump) and os.path.exists(lib)):
        return None
    d = tempfile.mkdtemp()
    try:
        shutil.copy(lib, os.path.join(d, "lib.so"))
        subprocess.run([objdump, "--offloading", "lib.so"], cwd=d, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        cos = [os.path.join(d, f) for f in os.listdir(d) if "gfx950" in f]
        if not cos:
            return None
        asm = subprocess.run([objdump, "-d", "--mcpu=gfx950", max(cos, key=os.path.getsize)], capture_output=True, text=True).stdout
        body = "\n".join(ln.split(":", 1)[1] if ":" in ln else ln for ln in asm.split("\n")[2:])
        return hashlib.sha1(body.encode()).hexdigest()[:12]
    finally:
        shutil.rmtree(d, ignore_errors=True)


_ISA = {}


def _profile_is_of_this_kernel(profile_hash):
    """the profile was taken on this tree's env kernel: same source hash, or (profiles/kernel_identity.json) the same instruction stream"""
    if profile_hash == kernel_source_hash():
        return True
    try:
        ident = json.load(open(os.path.join(REPO, "profiles", "kernel_identity.json")))
        want = ident.get(profile_hash, {}).get("isa_sha1")
        if not want:
            return False
        if "h" not in _ISA:
            _ISA["h"] = kernel_isa_hash()
        return _ISA["h"] == want
    except Exception:
        return False


def _pmc_traffic_bytes(kernel="env_step_kernel"):
    """HBM bytes per launch of `kernel` (env_step_kernel: one env step; env_rollout_kernel: a T-step rollout) from the committed rocprofv3 PMC passes (profiles/r05_env_step_pmc_hbm.txt, 4096 envs): 2 x
    FETCH_SIZE (gfx950 correction of the microarchitecture guide) + WRITE_SIZE, both in KB.  The profile records the hash of the kernel
    sources it was taken on (tools/profile_round.sh); a profile of another kernel is NOT quoted: None + a note on stderr."""
    import re
    path = os.path.join(REPO, "profiles", "r05_env_step_pmc_hbm.txt")
    try:
        txt = open(path).read()
        m = re.search(r"kernel sources sha1: (\w+)", txt)
        if not m or not _profile_is_of_this_kernel(m.group(1)):
            print("bench.py: %s was taken on another build of the env kernel (%s vs %s): roofline.traffic = null; re-run tools/profile_round.sh" % (
                path, m.group(1) if m else "no hash", kernel_source_hash()), file=sys.stderr)
            return None
        f = float(re.search(kernel + r"[^:]*: .*?FETCH_SIZE=([0-9.e+]+)", txt).group(1))
        w = float(re.search(kernel + r"[^:]*: .*?WRITE_SIZE=([0-9.e+]+)", txt).group(1))
        return int((2.0 * f + w) * 1024)
    except Exception:
        return None


def _pmc_issue(flop_step):
    """The instruction-issue view of the env kernel, from the SQ counter pass of the same profile round (profiles/r05_env_step_pmc_sq.txt + r05_env_step_pmc_issue.txt, quoted only while
    the HBM pass next to it carries this tree's kernel-source hash).  One single-wave workgroup sits on each SIMD, and a SIMD can start at most one VALU
    instruction per quad-cycle, the unit SQ_WAVE_CYCLES counts in: valu_issue_frac = SQ_INSTS_VALU / SQ_WAVE_CYCLES is how much of that ceiling the
    instruction stream uses, wait_frac = SQ_WAIT_ANY / SQ_WAVE_CYCLES how much of the time the wave sits in s_waitcnt.  flop_per_lane_instr relates it to the
    flop roofline: a stream of packed FMAs (what the 157.3 TFLOP/s peak assumes) would be 4, plain FMAs 2."""
    import re
    from apex_amd import roofline
    if _pmc_traffic_bytes() is None:
        return None
    try:
        txt = open(os.path.join(REPO, "profiles", "r05_env_step_pmc_sq.txt")).read()
        line = re.search(r"env_step_kernel[^:]*: (.*)", txt).group(1)
        g = lambda k: float(re.search(k + r"=([0-9.e+]+)", line).group(1))
        valu, wave, wait = g("SQ_INSTS_VALU"), g("SQ_WAVE_CYCLES"), g("SQ_WAIT_ANY")
        out = {"valu_issue_frac": round(valu / wave, 4), "wait_frac": round(wait / wave, 4), "valu_instr_per_launch": valu,
               "flop_per_lane_instr": round(flop_step * 4096 / (valu * 64.0), 3),      # the SAME flop figure as roofline.frac
               "source": "profiles/r05_env_step_pmc_sq.txt + r05_env_step_pmc_issue.txt (rocprofv3 --pmc, 4096 envs, per-dispatch means)"}
        try:      # the split of the wait (round 4, tools/profile_issue.sh): instruction-issue shares, LDS issue stalls, instruction-cache misses, dynamic arithmetic share
            it = open(os.path.join(REPO, "profiles", "r05_env_step_pmc_issue.txt")).read()
            if _profile_is_of_this_kernel(re.search(r"kernel sources sha1: (\w+)", it).group(1)):
                def gi(k):
                    m = re.search(r"env_step_kernel[^\n]*?" + k + r"=([0-9.e+]+)", it)
                    return float(m.group(1)) if m else None
                wc = gi("SQ_WAVE_CYCLES")
                out["split"] = {"active_any_frac": round(gi("SQ_ACTIVE_INST_ANY") / wc, 4), "active_valu_frac": round(gi("SQ_ACTIVE_INST_VALU") / wc, 4),
                                "active_salu_frac": round(gi("SQ_ACTIVE_INST_SCA") / wc, 4), "active_lds_frac": round(gi("SQ_ACTIVE_INST_LDS") / wc, 4),
                                "wait_any_frac": round(gi("SQ_WAIT_ANY") / wc, 4), "wait_inst_lds_frac": round(gi("SQ_WAIT_INST_LDS") / wc, 4),
                                "idle_frac_not_waiting": round(1.0 - (gi("SQ_ACTIVE_INST_ANY") + gi("SQ_WAIT_ANY")) / wc, 4),
                                "icache_hit_rate": round(gi("SQC_ICACHE_HITS") / gi("SQC_ICACHE_REQ"), 5), "icache_misses_per_launch": gi("SQC_ICACHE_MISSES") + gi("SQC_ICACHE_MISSES_DUPLICATE"),
                                "valu_fp32_arith_share": round((gi("SQ_INSTS_VALU_ADD_F32") + gi("SQ_INSTS_VALU_MUL_F32") + gi("SQ_INSTS_VALU_FMA_F32") + gi("SQ_INSTS_VALU_TRANS_F32")) / gi("SQ_INSTS_VALU"), 4),
                                "valu_int32_share": round(gi("SQ_INSTS_VALU_INT32") / gi("SQ_INSTS_VALU"), 4),
                                "bound": "LDS latency with one wave per SIMD (s_waitcnt on ds_read results; LDS issue stalls and instruction fetch are each < 1 % / < 4 %)"}
        except Exception:
            pass
        return out
    except Exception:
        return None


def _init_group(world, local, share):
    """The process group of the N > 1 path: RCCL ("nccl" IS RCCL on ROCm), one rank per GPU.  APX_FORCE_DIST=1 also builds it at world_size 1 (launched through
    torch.distributed.run --nproc-per-node 1), so that init_process_group("nccl", device_id=...), the gradient / moment / scalar all-reduces on device tensors and
    their event timing can be executed on a 1-GPU box; APX_BENCH_SHARE_GPU=1 (tests) puts every rank on cuda:0 over gloo."""
    if world == 1 and os.environ.get("APX_FORCE_DIST") != "1":
        return None
    if share:
        torch.distributed.init_process_group("gloo")
    else:
        torch.distributed.init_process_group("nccl", device_id=torch.device("cuda", local))
    return torch.distributed.group.WORLD


def _gather_per_rank(sample_s, optimize_s, allreduce_ms, device, dist_on, world):
    """{"sample_s": [...], "optimize_s": [...], "allreduce_ms_per_step": [...]} over the ranks (rank order): what a scaling record needs beyond rank 0's view"""
    v = torch.tensor([sample_s, optimize_s, allreduce_ms], dtype=torch.float64, device=device)
    if dist_on and world > 1:
        out = [torch.zeros_like(v) for _ in range(world)]
        torch.distributed.all_gather(out, v)
    else:
        out = [v]
    m = torch.stack(out).cpu().numpy()
    return {"sample_s": [round(float(x), 4) for x in m[:, 0]], "optimize_s": [round(float(x), 4) for x in m[:, 1]], "allreduce_ms_per_step": [round(float(x), 3) for x in m[:, 2]]}


def main_td3(a):
    """BASELINE.json configs[4] (next row f2): Cassie-v0 TD3, 1 GPU, 10^6-transition replay in HBM; a "step" = 32 lock-step env steps of 4096
    envs, each followed by 4 twin-critic updates on 1024 samples."""
    assert a.gpus == 1 and int(os.environ.get("WORLD_SIZE", 1)) == 1, "the TD3 workload is single-GPU (BASELINE.json configs[4])"
    torch.cuda.set_device(0)
    from apex_amd.vecenv import CassieVecEnv
    from apex_amd.td3 import TD3
    n_envs, T, upd, bs = (64, 2, 1, 64) if TINY else (4096, 32, 4, 1024)
    env = CassieVecEnv(n_envs=n_envs, seed=0)
    algo = TD3(env, "/tmp/apx_bench_unused", batch_size=bs, updates_per_step=upd, replay_size=4096 if TINY else 1_000_000, seed=0, one_launch_updates=a.td3_one_launch)
    algo.init_networks(0)
    run = (lambda: algo.collect_and_train_async(T, load_freq=10)) if a.td3_async else (lambda: algo.collect_and_train(T))      # --td3_async: rl/algos/async_td3.py's decoupled form
    for _ in range(a.warmup):
        run()
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(a.steps):
        out = run()
    torch.cuda.synchronize(); dt = time.time() - t0
    print(json.dumps({"metric": "env-steps/sec Cassie-v0 TD3 @4096 envs, replay in HBM" + (" (asynchronous: updates next to the following env step)" if a.td3_async else ""),
                      "value": round(a.steps * T * n_envs / dt, 1), "unit": "env-steps/s",
                      "n_gpus": 1, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 2), "higher_is_better": True,
                      "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                      "config": {"workload": "Cassie-v0 TD3, 1M-transition replay buffer in HBM, twin-critic update in HIP (BASELINE.json configs[4])",
                                 "envs_per_gpu": n_envs, "collect_steps": T, "updates_per_env_step": upd, "batch_size": bs, "replay_capacity": 1000000,
                                 "mode": "async (behaviour copy re-loaded every 10 lock steps, rl/algos/async_td3.py)" if a.td3_async else "sync (rl/algos/sync_td3.py)",
                                 "update_block_as_one_launch": bool(a.td3_one_launch and not a.td3_async), **({"tiny_test_shape": True} if TINY else {})},
                      "updates_per_s": round(a.steps * T * upd / dt, 1), "replay_size": int(algo.replay.size)}))


def main_recurrent(a):
    """BASELINE.json configs[3] (next row f1): CassieTraj-v0 recurrent PPO, 2048 envs/GPU, whole-trajectory minibatches.  Same contract:
    W warm-up iterations, K timed ones between barriers, one JSON line on rank 0."""
    rank = int(os.environ.get("RANK", 0)); world = int(os.environ.get("WORLD_SIZE", 1)); local = int(os.environ.get("LOCAL_RANK", 0))
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}"
    share = os.environ.get("APX_BENCH_SHARE_GPU") == "1"      # test hook: all ranks on cuda:0 over gloo
    if share:
        local = 0
    torch.cuda.set_device(local)
    group = _init_group(world, local, share)
    dist_on = group is not None
    from apex_amd.vecenv import CassieVecEnv
    from apex_amd.ppo_recurrent import RecurrentPPO
    from apex_amd import dist as adist
    n_envs, T = (a.n_envs if a.n_envs != 4096 else 2048), 400          # whole episodes: T = max_traj_len (every trajectory starts at an episode start, zero hidden state)
    if TINY:
        n_envs, T = 64, 3
    env = CassieVecEnv(n_envs=n_envs, seed=0, device=local, env_id_base=adist.shard_env_base(rank, n_envs), env_name="CassieTraj-v0", **({"max_traj_len": T} if TINY else {}))
    args = dict(gamma=0.99, lam=0.95, lr=1e-4, eps=1e-5, entropy_coeff=0.0, clip=0.2, minibatch_size=min(1024, n_envs // 2), epochs=a.epochs,
                num_steps=T * n_envs * world, max_traj_len=T if TINY else 400, max_grad_norm=0.05, mirror=True, seed=0)
    algo = RecurrentPPO(args, "/tmp/apx_bench_unused", env, rank=rank, world_size=world, group=group)
    algo.init_networks(0)
    if not TINY:      # (at least 50 env steps: setup, not part of what the tiny shape is there to exercise)
        algo.normalization_params(10000)

    def barrier():
        torch.cuda.synchronize()
        if dist_on:
            torch.distributed.barrier(); torch.cuda.synchronize()
    for _ in range(a.warmup):
        algo.iteration()
    if dist_on:
        adist.timing(True)
    barrier(); t0 = time.time(); samp = opt = 0.0
    epochs_run = opt_steps = 0
    for _ in range(a.steps):
        out = algo.iteration(); samp += out["sample_time"]; opt += out["optimize_time"]
        epochs_run += out.get("epochs", a.epochs); opt_steps += out.get("optimiser_steps", 0)
    barrier(); dt = time.time() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device=env.device)
    if dist_on:
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
    dt = float(tmax)
    ar_ms, ar_calls = adist.timing_read() if dist_on else (0.0, 0)
    adist.timing(False)
    per_rank = _gather_per_rank(samp / a.steps, opt / a.steps, ar_ms / a.steps, env.device, dist_on, world)
    if rank == 0:
        steps_total = a.steps * T * n_envs * world
        print(json.dumps({"metric": "env-steps/sec (whole node) CassieTraj-v0 recurrent PPO @2048 envs/GPU", "value": round(steps_total / dt, 1),
                          "unit": "env-steps/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 2),
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                          "config": {"workload": "CassieTraj-v0 recurrent PPO (LSTM 2x128 actor/critic, whole-trajectory minibatches), 2048 envs/GPU (BASELINE.json configs[3])",
                                     "envs_per_gpu": n_envs, "rollout_len": T, "minibatch_trajectories": min(1024, n_envs // 2), "epochs": a.epochs, "mirror_loss": True, **({"tiny_test_shape": True} if TINY else {}),
                                     "parallelism": f"dp{world} (env shards; optimiser steps per epoch agreed by a MAX all-reduce, 1 gradient all-reduce per step)"},
                          "sample_s": round(samp / a.steps, 3), "optimize_s": round(opt / a.steps, 3),
                          "optimiser_step_us": round(opt / opt_steps * 1e6, 2) if opt_steps else None, "epochs_run_per_step": round(epochs_run / a.steps, 2),      # update time per optimiser step (whole-trajectory minibatches) over the epochs that ran
                          "collectives": {"rccl_ranks_seen": torch.distributed.get_world_size() if dist_on else 1, "backend": torch.distributed.get_backend() if dist_on else None,
                                          "allreduce_calls_per_step": round(ar_calls / a.steps, 1), "allreduce_ms_per_step": round(ar_ms / a.steps, 3),
                                          "gradient_floats": int(algo.learner.grad_flat.numel()), "per_rank": per_rank}}))
    if dist_on:
        torch.distributed.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--n_envs", type=int, default=4096)
    ap.add_argument("--rollout_len", type=int, default=32)
    ap.add_argument("--minibatch", type=int, default=16384)
    ap.add_argument("--epochs", type=int, default=3)
    ap.add_argument("--no_cpu_baseline", action="store_true")
    ap.add_argument("--epoch_kernel", action="store_true", help="cassie_ppo, one GPU, --minibatch <= 256: an epoch's optimiser steps as ONE launch (apx_ppo_epoch) instead of 16 launches per step")
    ap.add_argument("--td3_one_launch", action="store_true", help="cassie_td3 workload only (synchronous form): the update block behind a collection as ONE launch (apx_td3_updates) instead of ~60 launches per update")
    ap.add_argument("--td3_async", action="store_true", help="cassie_td3 workload only: the asynchronous variant (collection and updates on two streams)")
    ap.add_argument("--workload", default="cassie_ppo", choices=["cassie_ppo", "cassietraj_recurrent", "cassie_td3"],
                    help="cassie_ppo = BASELINE.json configs[1] (the headline, default); cassietraj_recurrent = configs[3]: CassieTraj-v0, LSTM 2x128, 2048 envs/GPU; cassie_td3 = configs[4]: TD3, 1M-transition replay in HBM")
    a = ap.parse_args()
    if a.workload == "cassietraj_recurrent":
        return main_recurrent(a)
    if a.workload == "cassie_td3":
        return main_td3(a)

    rank = int(os.environ.get("RANK", 0)); world = int(os.environ.get("WORLD_SIZE", 1)); local = int(os.environ.get("LOCAL_RANK", 0))
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}"
    group = None
    # test hook (tools/t_two_ranks.sh): APX_BENCH_SHARE_GPU=1 puts every rank on cuda:0 with the gloo backend, so that the
    # N > 1 control flow (env shards, gradient / moment all-reduces, max-over-ranks timing) can be exercised on a 1-GPU box.
    # The driver's multi-GPU runs use one GPU per rank over RCCL.
    share = os.environ.get("APX_BENCH_SHARE_GPU") == "1"
    if share:
        local = 0
    torch.cuda.set_device(local)
    group = _init_group(world, local, share)
    dist_on = group is not None
    from apex_amd.vecenv import CassieVecEnv
    from apex_amd.ppo import PPO
    from apex_amd import dist as adist
    env = CassieVecEnv(n_envs=a.n_envs, seed=0, device=local, env_id_base=adist.shard_env_base(rank, a.n_envs))
    args = dict(gamma=0.99, lam=0.95, lr=1e-4, eps=1e-5, entropy_coeff=0.0, clip=0.2, minibatch_size=a.minibatch,
                epochs=a.epochs, num_steps=a.rollout_len * a.n_envs * world, max_traj_len=400, max_grad_norm=0.05,
                mirror=True, std_dev=-1.5, seed=0, graph=os.environ.get("APX_ROLLOUT_GRAPH", "0") == "1", epoch_kernel=a.epoch_kernel)      # fp32 MFMA: the reference's own network precision and the library's only mode
    algo = PPO(args, "/tmp/apx_bench_unused", env, rank=rank, world_size=world, group=group)
    algo.init_networks(0)
    if not TINY:
        algo.normalization_params(10000)

    def barrier():
        torch.cuda.synchronize()
        if dist_on:
            torch.distributed.barrier()
            torch.cuda.synchronize()

    for _ in range(a.warmup):
        algo.iteration()
    # per-launch duration of the dominant kernel (env step): hipEvent pairs recorded by the library around every env_step_kernel launch of
    # the timed region, on the stream the kernel is launched on (include/apx.h apx_env_timing)
    env.kernel_timing(True); env.kernel_timing_read(reset=True)
    if dist_on:
        adist.timing(True)
    barrier()
    t0 = time.time()
    samp = opt = 0.0
    epochs_run = 0
    for _ in range(a.steps):
        out = algo.iteration()
        samp += out["sample_time"]; opt += out["optimize_time"]; epochs_run += out["epochs"]
    barrier()
    dt = time.time() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device=env.device)
    if dist_on:
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
    dt = float(tmax)
    ar_ms, ar_calls = adist.timing_read() if dist_on else (0.0, 0)
    adist.timing(False)
    per_rank = _gather_per_rank(samp / a.steps, opt / a.steps, ar_ms / a.steps, env.device, dist_on, world)
    k_total_ms, k_launches = env.kernel_timing_read(reset=True)      # the env_step_kernel launches of the timed region
    if k_launches == 0:      # APX_ROLLOUT_GRAPH=1: the rollout was captured during warm-up, before the event pairs were switched on -> a few eager timed launches
        for _ in range(8):
            env.step(algo.b_act[0] if hasattr(algo, "b_act") else torch.zeros(a.n_envs, 10, device=env.device), auto_reset=True)
        k_total_ms, k_launches = env.kernel_timing_read(reset=True)
    env.kernel_timing(False)
    k_ms = k_total_ms / max(k_launches, 1)
    # round 5: apx_rollout is ONE launch per T-step rollout (env_rollout_kernel: policy forward, env step and restart of every step inside); the per-step launches
    # (env_step_kernel) remain behind APX_ROLLOUT_STEPWISE=1, graph capture and the non-2x256 shapes
    one_launch = k_launches <= a.steps
    spl = a.rollout_len if one_launch else 1                       # env steps per launch
    k_name = "env_rollout_kernel" if one_launch else "env_step_kernel"

    # MFMA side: the actor's 3-layer fp32 forward on one minibatch (50 -> 256 -> 256 -> 10), events on the launch stream
    mb_rows = min(a.minibatch, a.rollout_len * a.n_envs)
    xb = algo.b_obs.view(-1, 50)[:mb_rows].contiguous()
    for _ in range(3):
        algo.learner.actor.forward(xb, algo.learner.obs_mean, algo.learner.obs_std)
    g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g0.record()
    for _ in range(20):
        algo.learner.actor.forward(xb, algo.learner.obs_mean, algo.learner.obs_std)
    g1.record(); torch.cuda.synchronize()
    mlp_ms = g0.elapsed_time(g1) / 20
    mlp_flop = 2.0 * mb_rows * (50 * 256 + 256 * 256 + 256 * 10)

    if rank == 0:
        steps_total = a.steps * a.rollout_len * a.n_envs * world
        from apex_amd import roofline
        bytes_per_env_launch = roofline.rollout_bytes_per_env(a.rollout_len) if one_launch else roofline.ENV_STEP_BYTES
        achieved = bytes_per_env_launch * a.n_envs / (k_ms * 1e-3) / 1e9
        cpu = None
        if not a.no_cpu_baseline and world == 1:      # the CPU baseline is timed on rank 0 at N = 1 only; it also re-measures the flop count
            cpu = cpu_baseline(n_envs=a.n_envs, rollout_len=a.rollout_len, minibatch=a.minibatch, epochs=a.epochs)
        flop_step = cpu["flop_per_env_step_counted"] if cpu else roofline.ENV_STEP_FLOP_COUNTED
        res = {
            "metric": "env-steps/sec (whole node) Cassie-v0 PPO @4096 envs/GPU", "value": round(steps_total / dt, 1),
            "unit": "env-steps/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(dt / a.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "Cassie-v0 PPO, 4096 batched envs/GPU, 2x256 MLP actor/critic (BASELINE.json configs[1])",
                       "envs_per_gpu": a.n_envs, "rollout_len": a.rollout_len, "simrate": 50, "minibatch": a.minibatch, "optimiser_steps_as_one_launch_per_epoch": algo.epoch_kernel_in_use(mb_rows), "epoch_kernel_requested": bool(a.epoch_kernel),
                       "epochs": a.epochs, "mirror_loss": True, "dynamics_randomization": True, "eval_rollouts": False, **({"tiny_test_shape": True} if TINY else {}),
                       "parallelism": f"dp{world} (env shards, 1 RCCL grad all-reduce per optimiser step)"},
            "sampling_env_steps_per_s": round(a.steps * a.rollout_len * a.n_envs * world / max(samp, 1e-9), 1),
            "sample_s": round(samp / a.steps, 3), "optimize_s": round(opt / a.steps, 3),
            "optimiser_step_us": round(opt / max(1, epochs_run * ((a.rollout_len * a.n_envs) // min(a.minibatch, a.rollout_len * a.n_envs))) * 1e6, 2), "epochs_run_per_step": round(epochs_run / a.steps, 2),      # update time per optimiser step over the epochs that ran (the KL test of ppo.py:449 may end an iteration's update early)
            # what the collective path actually was in this run (explains a scaling curve on its own): ranks the process group saw, backend, and the
            # gradient / scalar all-reduces of the timed region (hipEvents on the launch stream of rank 0)
            "collectives": {"rccl_ranks_seen": torch.distributed.get_world_size() if dist_on else 1, "backend": torch.distributed.get_backend() if dist_on else None,
                            "allreduce_calls_per_step": round(ar_calls / a.steps, 1), "allreduce_ms_per_step": round(ar_ms / a.steps, 3), "gradient_floats": 160523,
                            # the gradient travels in two halves (actor, critic): the window measured per optimiser step runs from the start of the actor half to the join of
                            # both and CONTAINS the critic's backward it overlaps; per_rank = what each rank saw (sampling, update, that window), for the scaling record
                            "per_rank": per_rank},
            # the binding bound of the dominant kernel is the fp32 vector pipe (HBM traffic is 1.1 x the algorithmic bytes and
            # < 0.1 % of the peak): achieved = instrumented flops of the CPU restatement per env step x envs / launch time.  The HBM view
            # north_star asks for is reported beside it.
            "roofline": {"kernel": k_name, "env_steps_per_launch_per_env": spl, "bound": "valu", "achieved": round(flop_step * a.n_envs * spl / (k_ms * 1e-3) / 1e12, 4),
                         "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(flop_step * a.n_envs * spl / (k_ms * 1e-3) / 1e12 / VALU_PEAK_TFLOPS, 6),
                         # the same launch time against the KERNEL's own (tree-sparse) operation count: the dense oracle executes ~13 % more operations than the kernel needs
                         "frac_sparse": round(roofline.ENV_STEP_FLOP * a.n_envs * spl / (k_ms * 1e-3) / 1e12 / VALU_PEAK_TFLOPS, 6), "flop_per_env_step_sparse": roofline.ENV_STEP_FLOP,
                         # traffic: PMC bytes per launch of THIS kernel; issue: the SQ counters of env_step_kernel (the same substep code, one env step per launch: tools/t_pmc.py)
                         "traffic": _pmc_traffic_bytes(k_name), "issue": _pmc_issue(flop_step), "ms_per_launch": round(k_ms, 4), "ms_per_env_step": round(k_ms / spl, 4), "launches_timed": k_launches,
                         "flop_per_env_step": flop_step, "flop_source": "oracle-equivalent flops: instrumented count of the dense fp64 restatement oracle/cassie_phys.cpp (oracle.sim.count_flops); frac_sparse uses the hand count of the kernel's tree-sparse formulation",
                         "hbm": {"bytes_per_env_per_launch": bytes_per_env_launch, "achieved_GBps": round(achieved, 3), "peak_GBps": HBM_PEAK_GBS, "frac": round(achieved / HBM_PEAK_GBS, 6)},
                         "mlp_forward_mfma": {"what": "actor forward, %d x (50-256-256-10), fp32 MFMA (v_mfma_f32_32x32x2_f32), one fused launch (input normalisation + 3 layers, activations in LDS)" % mb_rows,
                                              "ms": round(mlp_ms, 4), "achieved_tflops": round(mlp_flop / (mlp_ms * 1e-3) / 1e12, 2),
                                              "peak_tflops": VALU_PEAK_TFLOPS, "frac": round(mlp_flop / (mlp_ms * 1e-3) / 1e12 / VALU_PEAK_TFLOPS, 4)}},
        }
        if cpu:
            res["cpu_baseline"] = cpu
        print(json.dumps(res))
    if dist_on:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
