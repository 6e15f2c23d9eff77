"""Golden G11c: the reference state estimator's HEIGHT output (pelvis.position[2] - terrain.height, the first observation entry,
cassie.py:793) on our sensor stream while a policy trained with this build (any actor.pt of apex.py ppo; the committed fixture used a 1000-iteration one) walks for 3 s, per 2 kHz substep,
together with the true pelvis z and the lowest world z of the two foot soles (foot capsule end - radius).  Pins the height model of
the build: height = z - L, L' = (lowest sole z - L) / tau."""
import sys, os
import numpy as np, torch
sys.path.insert(0, "/root/repo")
from native_blocks import cm, make_out, DRIVES, JOINTS
from oracle.sim import OracleEnv
from common import GOLD
policy = torch.load(os.path.join(sys.argv[1], "actor.pt"), weights_only=False); policy.eval()
est = cm.state_output_alloc(); cm.state_output_setup(est)
e = OracleEnv(dyn_rand=False, seed=3)
obs = e.reset(); obs = e.reset_for_test(); e.update_speed(0.0)
off = np.array([0.0045, 0, 0.4973, -1.1997, -1.5968] * 2)
k = 0
rec = []
for step in range(120):
    with torch.no_grad():
        act = policy(torch.tensor(obs, dtype=torch.float32), deterministic=True).numpy().astype(np.float64)
    e.set("pd_target", act + off); e.set("pd_P", [100, 100, 88, 96, 50] * 2); e.set("pd_D", [10, 10, 8, 9.6, 5] * 2)
    for sub in range(50):
        e.substep(); k += 1
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
        fl = e.get("foot_low")
        rec.append((e.get("qpos")[2], min(fl[1], fl[3]), so.pelvis.position[2] - so.terrain.height, so.pelvis.position[2], so.terrain.height))
    ints = e.get("ints"); ints[0] += 1; ints[1] += 1
    if ints[1] > e.get("phaselen")[0]: ints[1] = 0; ints[2] += 1
    e.set("ints", ints); obs = e.obs()

rec = np.array(rec)
np.savez_compressed(os.path.join(GOLD, "g11c_estimator_height.npz"), z=rec[:, 0].astype(np.float32), sole_low=rec[:, 1].astype(np.float32), ref_height=rec[:, 2].astype(np.float32), ref_pelvis_z=rec[:, 3].astype(np.float32), ref_terrain=rec[:, 4].astype(np.float32))
# fit tau and L0
z, sl, rh = rec[:, 0], rec[:, 1], rec[:, 2]
best = None
for tau in np.arange(0.7, 1.5, 0.01):
    for L0 in np.arange(0.10, 0.16, 0.002):
        L = L0; err = []
        for i in range(len(z)):
            L += 0.0005 / tau * (sl[i] - L)
            err.append(z[i] - L - rh[i])
        m = np.abs(np.array(err)[10:]).max()
        if best is None or m < best[0]: best = (m, tau, L0)
print("best max |error| %.4f m at tau %.2f s, L0 %.3f" % best)
