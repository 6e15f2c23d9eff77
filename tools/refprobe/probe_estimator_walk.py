"""G11b: the reference's state_output_step fed with OUR oracle's sensor stream while a TRAINED policy walks (the on-distribution
counterpart of probe_estimator.py, which uses a falling robot).  Reports the error of the closed-form estimator-lite on the 7
filtered outputs and writes the subsampled stream + reference outputs as a golden.
usage: python probe_estimator_walk.py <run dir with actor.pt>"""
import sys, os
import numpy as np
import torch
sys.path.insert(0, "/root/repo")
from native_blocks import cm, make_out, DRIVES, JOINTS
from oracle.sim import OracleEnv
from common import GOLD
np.set_printoptions(precision=4, suppress=True, linewidth=200)

policy = torch.load(os.path.join(sys.argv[1], "actor.pt"), weights_only=False); policy.eval()
est = cm.state_output_alloc(); cm.state_output_setup(est)
e = OracleEnv(dyn_rand=False, seed=3)
obs = e.reset(); obs = e.reset_for_test(); e.update_speed(1.0)
off = np.array([0.0045, 0, 0.4973, -1.1997, -1.5968] * 2)
def q2m(q):
    w, x, y, z = q
    return np.array([[1-2*(y*y+z*z), 2*(x*y-w*z), 2*(x*z+w*y)], [2*(x*y+w*z), 1-2*(x*x+z*z), 2*(y*z-w*x)], [2*(x*z-w*y), 2*(y*z+w*x), 1-2*(x*x+y*y)]])
rows = []
for step in range(200):
    with torch.no_grad():
        act = policy(torch.tensor(obs, dtype=torch.float32), deterministic=True).numpy().astype(np.float64)
    # the same substeps env.step_basic runs, tapped one by one
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
        for k in range(4): out.pelvis.vectorNav.orientation[k] = q[k]
        for k in range(3): out.pelvis.vectorNav.angularVelocity[k] = gy[k]; out.pelvis.vectorNav.linearAcceleration[k] = ac[k]
        so = cm.state_out_t(); cm.state_output_step(est, out, so)
    qpos, qvel = e.get("qpos"), e.get("qvel"); R = q2m(q)
    rows.append(dict(z=qpos[2], quat=q.copy(), acc=ac.copy(), gyro=gy.copy(), v_world=qvel[:3].copy(),
                     ref_height=so.pelvis.position[2] - so.terrain.height, ref_tvel=np.array(so.pelvis.translationalVelocity[:]),
                     ref_tacc=np.array(so.pelvis.translationalAcceleration[:]),
                     lite_height=e.get("so_height")[0], lite_tvel=e.get("so_tvel").copy(), lite_tacc=e.get("so_tacc").copy()))
    obs = e.obs() if False else None
    # advance the env bookkeeping exactly like step_basic would (time / phase) and rebuild the observation
    ints = e.get("ints"); ints[0] += 1; ints[1] += 1
    if ints[1] > e.get("phaselen")[0]: ints[1] = 0; ints[2] += 1
    e.set("ints", ints)
    obs = e.obs()
    if qpos[2] < 0.5: break
A = lambda k: np.array([r[k] for r in rows])
print("env steps walked:", len(rows), " final pelvis z %.3f  mean forward speed %.2f m/s" % (rows[-1]["z"], A("v_world")[20:, 0].mean()))
for nm, ref, lite in (("height", A("ref_height"), A("lite_height")), ("tvel", A("ref_tvel"), A("lite_tvel")), ("tacc", A("ref_tacc"), A("lite_tacc"))):
    err = np.abs(lite - ref); sig = np.std(ref, axis=0)
    print("%-6s mean |estimator-lite - reference filter| = %s   (std of the reference signal %s)" % (nm, np.round(err.mean(0), 4), np.round(sig, 4)))
np.savez_compressed(os.path.join(GOLD, "g11b_estimator_walk.npz"), z=A("z"), quat=A("quat"), acc=A("acc"), gyro=A("gyro"), v_world=A("v_world"),
                    ref_height=A("ref_height"), ref_tvel=A("ref_tvel"), ref_tacc=A("ref_tacc"))
print("wrote g11b", len(rows))

# ---- what is the filter's height relative to?  (pelvis z minus terrain estimate)
if os.environ.get("PROBE_HEIGHT"):
    pass
