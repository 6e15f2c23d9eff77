import sys, os, ctypes
import numpy as np, torch
sys.path.insert(0, "/root/repo/tools/refprobe"); sys.path.insert(0, "/root/repo")
from native_blocks import cm, make_out, DRIVES, JOINTS
from oracle.sim import OracleEnv
policy = torch.load("/root/repo/trained_models/r03_cassie_v0_clock/actor.pt", weights_only=False); policy.eval()
est = cm.state_output_alloc(); cm.state_output_setup(est)
def dump(est):
    p = ctypes.cast(est, ctypes.POINTER(ctypes.c_double))
    return np.array([p[i] for i in range(526)])
e = OracleEnv(dyn_rand=False, seed=3)
obs = e.reset(); obs = e.reset_for_test(); e.update_speed(1.0)
off = np.array([0.0045, 0, 0.4973, -1.1997, -1.5968] * 2)
inp, outs, sts, truth = [], [], [], []
sts.append(dump(est))
for step in range(60):
    with torch.no_grad():
        act = policy(torch.tensor(obs, dtype=torch.float32), deterministic=True).numpy().astype(np.float64)
    e.set("pd_target", act + off); e.set("pd_P", [100, 100, 88, 96, 50] * 2); e.set("pd_D", [10, 10, 8, 9.6, 5] * 2)
    for sub in range(50):
        e.substep()
        out = make_out()
        mp, mv, tq, jp, jv = e.get("so_mpos"), e.get("so_mvel"), e.get("so_torque"), e.get("so_jpos"), e.get("so_jvel")
        for i in range(10):
            d = getattr(out.leftLeg if i < 5 else out.rightLeg, DRIVES[i % 5]); d.position, d.velocity, d.torque = mp[i], mv[i], tq[i]
        for i in range(6):
            j = getattr(out.leftLeg if i < 3 else out.rightLeg, JOINTS[i % 3]); j.position, j.velocity = jp[i], jv[i]
        q, gy, ac = e.get("so_quat"), e.get("so_rotvel"), e.get("snap_acc")
        for kk in range(4): out.pelvis.vectorNav.orientation[kk] = q[kk]
        for kk in range(3): out.pelvis.vectorNav.angularVelocity[kk] = gy[kk]; out.pelvis.vectorNav.linearAcceleration[kk] = ac[kk]
        so = cm.state_out_t(); cm.state_output_step(est, out, so)
        inp.append(np.concatenate([mp, mv, tq, jp, jv, q, gy, ac]))   # 10+10+10+6+6+4+3+3 = 52
        outs.append(np.frombuffer(bytes(so), dtype=np.float64)[:61].copy())  # up to terrain.slope
        sts.append(dump(est))
        truth.append(np.concatenate([e.get("qpos")[:7], e.get("qvel")[:6]]))
    ints = e.get("ints"); ints[0] += 1; ints[1] += 1
    if ints[1] > e.get("phaselen")[0]: ints[1] = 0; ints[2] += 1
    e.set("ints", ints); obs = e.obs()
np.savez("/tmp/est_stream.npz", inp=np.array(inp), out=np.array(outs), st=np.array(sts), truth=np.array(truth))
print(np.array(inp).shape, np.array(sts).shape)
