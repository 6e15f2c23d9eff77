import sys, numpy as np
sys.path.insert(0, "/root/repo/tools/refprobe"); sys.path.insert(0, "/root/repo")
from native_blocks import cm, make_out, DRIVES, JOINTS
np.set_printoptions(precision=5, suppress=True, linewidth=220)
def run(n=600, mod=None):
    est = cm.state_output_alloc(); cm.state_output_setup(est)
    res = []
    for t in range(n):
        o = make_out()
        o.pelvis.vectorNav.linearAcceleration[2] = 9.806
        if mod: mod(o, t)
        so = cm.state_out_t(); cm.state_output_step(est, o, so)
        res.append(dict(v=np.array(so.pelvis.translationalVelocity[:]), p=np.array(so.pelvis.position[:]), th=so.terrain.height,
                        lf=np.array(so.leftFoot.position[:]), rf=np.array(so.rightFoot.position[:]), ltf=np.array(so.leftFoot.toeForce[:]), lhf=np.array(so.leftFoot.heelForce[:]),
                        rtf=np.array(so.rightFoot.toeForce[:]), lfv=np.array(so.leftFoot.footTranslationalVelocity[:]) if hasattr(so.leftFoot, 'footTranslationalVelocity') else None))
    return res
def show(tag, r, idx=(0, 1, 2, 5, 10, 50, 200, 599)):
    print(tag)
    for i in idx:
        x = r[i]; print("  t=%3d v %s pz %.4f terr %.4f lf %s ltoeF %s lheelF %s" % (i, x['v'], x['p'][2], x['th'], x['lf'], x['ltf'], x['lhf']))
show("nominal", run())
def m1(o, t): o.leftLeg.hipPitchDrive.velocity = 1.0 if t >= 10 else 0.0
show("left hip pitch vel 1 from t=10", run(mod=m1))
def m2(o, t): o.leftLeg.hipPitchDrive.velocity = 1.0 if t >= 10 else 0.0; o.rightLeg.hipPitchDrive.velocity = 1.0 if t >= 10 else 0.0
show("both hip pitch vel 1 from t=10", run(mod=m2))
def m3(o, t):
    # deflect left springs (shin joint) to signal ground force
    o.leftLeg.shinJoint.position = -0.02
show("left shin spring deflected", run(mod=m3))
