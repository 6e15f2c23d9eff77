"""Stage-level shader-clock profile of sim_step_pd (lane 0 of workgroup 0).  Build: make -C apex_amd/csrc VARIANT=prof EXTRA=-DAPX_PROF
(coarse, slots 0-8) or VARIANT=prof2 EXTRA=-DAPX_PROF=2 (fine probes, slots 12-32); run: APX_LIB=apex_amd/lib/libapx_prof.so python tools/t_prof.py"""
import sys, torch, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from apex_amd.vecenv import CassieVecEnv
env = CassieVecEnv(n_envs=4096, seed=0)
env.reset()
torch.manual_seed(0); act = torch.randn(4096,10,device="cuda")*0.2
for _ in range(2): env.step(act, auto_reset=False)
buf = torch.zeros(4096*128, device='cuda')
from apex_amd import _lib; from apex_amd.engine import _p, _stream
lib=_lib.load()
lib.apx_env_get_field(env._h, b"prof", _p(buf), _stream())   # reset counters
torch.cuda.synchronize(); t0=time.time()
K=4
for _ in range(K): env.step(act, auto_reset=False)
torch.cuda.synchronize(); dt=(time.time()-t0)/K
lib.apx_env_get_field(env._h, b"prof", _p(buf), _stream())
p = buf[:48].cpu().numpy() / (K*50)
names=["io_model","tree_walk","factor","pgs_tail(z~)","finish+euler","rows(2 legs)","gram+warm","pgs_sweeps"]
print("env step ms %.2f"%(dt*1e3))
for n,v in zip(names,p): print("%-16s %9.0f cycles/substep"%(n,v))
print("total %.0f cycles/substep"%p[:8].sum()); print("whole sim_step_pd %.0f cycles/substep" % p[8])
fine = {12: "tree: constants, pelvis, joint quats", 13: "tree: pose pointer-jumping (3 rounds)", 14: "tree: velocity + acceleration chain sums", 15: "tree: cdof store",
        16: "tree: inertia + RNE force, records", 17: "tree: subtree sums + store", 18: "tree: pelvis composite, anchors / capsules", 19: "tree: CRBA rows + bias forces",
        20: "factor: elimination + pelvis block", 21: "factor: store", 23: "rows leg 0 (tail: scalars)", 24: "rows leg 1 (tail: scalars)",
        25: "rows: limit/contact detection + Jacobian (both legs)", 26: "rows: raw dots (both legs)", 27: "rows: whitening (both legs)",
        28: "finish: factor load + vectors", 29: "finish: qacc solve + sensors", 30: "finish: rhs = smooth + L^T D^1/2 z", 31: "finish: second factorisation",
        32: "finish: solves", 37: "io: encoders, PD, safety, delay line", 33: "estimator: record unpack + heel springs (2 Newton steps)", 34: "estimator: leg kinematics (DPP prefix product)",
        35: "estimator: spring Jacobian, force solve, IMU", 36: "estimator: three Kalman filters", 38: "estimator: terrain, outputs, record store"}
if p[12:].sum() > 0:
    for k in sorted(fine): print("  [%2d] %-52s %8.0f" % (k, fine[k], p[k]))
