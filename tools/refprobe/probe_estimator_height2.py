import sys, numpy as np
sys.path.insert(0, "/root/repo/tools/refprobe"); sys.path.insert(0, "/root/repo")
from native_blocks import cm, make_out
np.set_printoptions(precision=5, suppress=True, linewidth=220)
def run(n=4000, mod=None, every=(0,1,2,5,20,100,500,1000,2000,3999)):
    est = cm.state_output_alloc(); cm.state_output_setup(est)
    out=[]
    for t in range(n):
        o = make_out(); o.pelvis.vectorNav.linearAcceleration[2] = 9.806
        if mod: mod(o, t)
        so = cm.state_out_t(); cm.state_output_step(est, o, so)
        if t in every: out.append((t, so.pelvis.position[2], so.terrain.height, so.leftFoot.position[2], so.rightFoot.position[2], so.leftFoot.toeForce[2], so.pelvis.translationalVelocity[2]))
    return out
def show(tag, r):
    print(tag)
    for x in r: print("  t=%4d pz %.5f terr %.5f h %.5f lfz %.5f rfz %.5f ltoeFz %.2f vz %.5f" % (x[0], x[1], x[2], x[1]-x[2], x[3], x[4], x[5], x[6]))
show("nominal", run())
def knee(o, t):
    for leg in (o.leftLeg, o.rightLeg):
        leg.kneeDrive.position = -1.1997 - 0.3; leg.footDrive.position = -1.5968 + 0.0
        leg.tarsusJoint.position = 1.4267 + 0.3
show("knee bent 0.3 more (both legs), tarsus +0.3", run(mod=knee))
def spring(o, t):
    for leg in (o.leftLeg, o.rightLeg): leg.shinJoint.position = -0.01
show("both shin springs deflected -0.01", run(mod=spring))
def spring_l(o, t):
    o.leftLeg.shinJoint.position = -0.01
show("left shin spring deflected -0.01", run(mod=spring_l))
