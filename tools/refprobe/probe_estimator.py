"""G11 procedure: drive the reference's own state_output_step with the sensor stream of OUR physics
restatement and see what the 7 filtered outputs are relative to ground truth (frames, offsets)."""
import sys, os
import numpy as np
sys.path.insert(0, "/root/repo")
from native_blocks import cm, make_out, DRIVES, JOINTS
from oracle.sim import OracleEnv
np.set_printoptions(precision=4, suppress=True, linewidth=200)

est = cm.state_output_alloc(); cm.state_output_setup(est)
e = OracleEnv(dyn_rand=False, seed=2)
e.reset()
rng = np.random.RandomState(0)
def quat2mat(q):
    w,x,y,z=q
    return np.array([[1-2*(y*y+z*z),2*(x*y-w*z),2*(x*z+w*y)],[2*(x*y+w*z),1-2*(x*x+z*z),2*(y*z-w*x)],[2*(x*z-w*y),2*(y*z+w*x),1-2*(x*x+y*y)]])
rows=[]
for step in range(40):
    act = rng.randn(10)*0.1
    # replicate env.step's target setting then substep manually so that we can tap every substep
    e.set("pd_target", act + np.array([0.0045,0,0.4973,-1.1997,-1.5968]*2)); e.set("pd_P", [100,100,88,96,50]*2); e.set("pd_D", [10,10,8,9.6,5]*2)
    for sub in range(50):
        e.substep()
        out = make_out()
        mp, mv, tq = e.get("so_mpos"), e.get("so_mvel"), e.get("so_torque")
        jp, jv = e.get("so_jpos"), e.get("so_jvel")
        for i in range(10):
            leg = out.leftLeg if i < 5 else out.rightLeg
            d = getattr(leg, DRIVES[i % 5]); d.position, d.velocity, d.torque = mp[i], mv[i], tq[i]
        for i in range(6):
            leg = out.leftLeg if i < 3 else out.rightLeg
            j = getattr(leg, JOINTS[i % 3]); j.position, j.velocity = jp[i], jv[i]
        q = e.get("so_quat"); gy = e.get("so_rotvel"); ac = e.get("snap_acc")
        for k in range(4): out.pelvis.vectorNav.orientation[k] = q[k]
        for k in range(3): out.pelvis.vectorNav.angularVelocity[k] = gy[k]; out.pelvis.vectorNav.linearAcceleration[k] = ac[k]
        so = cm.state_out_t()
        cm.state_output_step(est, out, so)
        qpos, qvel = e.get("qpos"), e.get("qvel")
        R = quat2mat(q)
        rows.append(dict(z=qpos[2], est_z=so.pelvis.position[2], th=so.terrain.height, est_v=np.array(so.pelvis.translationalVelocity[:]),
                         v_world=qvel[:3].copy(), v_body=R.T@qvel[:3], est_a=np.array(so.pelvis.translationalAcceleration[:]),
                         a_raw=ac.copy(), a_world=R@ac - np.array([0,0,9.81]), a_body_mg=ac - R.T@np.array([0,0,9.81]),
                         est_q=np.array(so.pelvis.orientation[:]), q=q.copy(), est_w=np.array(so.pelvis.rotationalVelocity[:]), w=gy.copy(),
                         est_pos=np.array(so.pelvis.position[:]), lf=np.array(so.leftFoot.position[:]), ncon=e.get("ints")[3],
                         lfrc=np.array(so.leftFoot.toeForce[:])))
    if e.get("qpos")[2] < 0.45: break
import pickle
R=rows
for k in list(range(0, len(R), max(1,len(R)//25))):
    r=R[k]
    print("t=%4d z %.3f est_z %.3f terr %.3f | est_v %s world %s body %s | est_a %s a_world %s a_body-g %s ncon %d" % (k, r['z'], r['est_z'], r['th'], r['est_v'], r['v_world'], r['v_body'], r['est_a'], r['a_world'], r['a_body_mg'], r['ncon']))
ev=np.array([r['est_v'] for r in R]); vw=np.array([r['v_world'] for r in R]); vb=np.array([r['v_body'] for r in R])
print("vel err vs world:", np.abs(ev-vw).mean(0), " vs body:", np.abs(ev-vb).mean(0))
ea=np.array([r['est_a'] for r in R]); aw=np.array([r['a_world'] for r in R]); ab=np.array([r['a_body_mg'] for r in R])
print("acc err vs world:", np.abs(ea-aw).mean(0), " vs body-g:", np.abs(ea-ab).mean(0))
print("height: mean(est_z - terr - (z)) =", np.mean([r['est_z']-r['th']-r['z'] for r in R]), " std", np.std([r['est_z']-r['th']-r['z'] for r in R]))
print("quat passthrough err", max(np.abs(r['est_q']-r['q']).max() for r in R), " gyro passthrough err", max(np.abs(r['est_w']-r['w']).max() for r in R))

# ---- G11 golden: subsampled sensor stream + the reference estimator's 7 filtered outputs on it
from common import GOLD
idx = np.arange(0, len(R), 8)
np.savez_compressed(os.path.join(GOLD, "g11_estimator.npz"),
    z=np.array([R[i]['z'] for i in idx]), quat=np.array([R[i]['q'] for i in idx]), acc=np.array([R[i]['a_raw'] for i in idx]),
    v_world=np.array([R[i]['v_world'] for i in idx]),
    ref_height=np.array([R[i]['est_z'] - R[i]['th'] for i in idx]), ref_tvel=np.array([R[i]['est_v'] for i in idx]),
    ref_tacc=np.array([R[i]['est_a'] for i in idx]), ref_quat=np.array([R[i]['est_q'] for i in idx]), ref_rotvel=np.array([R[i]['est_w'] for i in idx]),
    gyro=np.array([R[i]['w'] for i in idx]))
print("wrote g11", len(idx))
