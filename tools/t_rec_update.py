"""Where the recurrent update spends its time (host view, with device syncs): python tools/t_rec_update.py   (GPU box)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from apex_amd.vecenv import CassieVecEnv
from apex_amd.ppo_recurrent import RecurrentPPO
from apex_amd import engine

env = CassieVecEnv(n_envs=2048, seed=0, device=0, env_name="CassieTraj-v0")
args = dict(gamma=0.99, lam=0.95, lr=1e-4, eps=1e-5, entropy_coeff=0.0, clip=0.2, minibatch_size=1024, epochs=3, num_steps=400 * 2048, max_traj_len=400, max_grad_norm=0.05, mirror=True, seed=0)
algo = RecurrentPPO(args, "/tmp/apx_unused", env)
algo.init_networks(0); algo.normalization_params(10000)
algo.iteration()
sync = torch.cuda.synchronize
sync(); t0 = time.time(); ret = algo.sample(); sync(); t1 = time.time()
print("sample %.1f ms" % ((t1 - t0) * 1e3))
L, T, N = algo.learner, algo.T, algo.N
val = algo.b_val.view(-1); retf = ret.view(-1)
t = time.time(); adv = engine.normalize_advantages(retf, val, algo.eps); sync(); print("normalize_advantages %.2f ms" % ((time.time() - t) * 1e3))
t = time.time(); trajs = algo.trajectories(); print("trajectories %.2f ms: %d trajectories, mean length %.1f, max %d" % ((time.time() - t) * 1e3, len(trajs), (trajs[:, 2] - trajs[:, 1]).mean(), (trajs[:, 2] - trajs[:, 1]).max()))
L.sync_old()
order = np.random.RandomState(0).permutation(len(trajs))
flat = lambda x, d: x.view(T * N, d)
tt = dict(index=0.0, gather=0.0, minibatch=0.0); nmb = 0; tmax = []
sync(); te = time.time()
for k in range(0, len(order), 1024):
    t = time.time(); idx = algo.padded_index(trajs[order[k:k + 1024]]); sync(); tt["index"] += time.time() - t
    t = time.time(); o_p, a_p, r_p, d_p, m_p, prep = L.gather(idx, flat(algo.b_obs, 50), flat(algo.b_act, 10), retf, adv, mirror=True); sync(); tt["gather"] += time.time() - t
    t = time.time(); L.minibatch(o_p, a_p, r_p, d_p, m_p, mirror=True, prepared=prep); sync(); tt["minibatch"] += time.time() - t
    nmb += 1; tmax.append(idx.shape[0])
print("epoch with syncs %.1f ms over %d minibatches; per minibatch: " % ((time.time() - te) * 1e3, nmb) + ", ".join("%s %.2f ms" % (k, v / nmb * 1e3) for k, v in tt.items()))
print("T_max per minibatch: mean %.1f min %d max %d" % (np.mean(tmax), min(tmax), max(tmax)))
sync(); t = time.time(); losses = algo.update(ret); sync(); print("update() %.1f ms" % ((time.time() - t) * 1e3))
