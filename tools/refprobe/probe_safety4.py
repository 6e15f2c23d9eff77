import numpy as np
from native_blocks import *
np.set_printoptions(precision=4, suppress=True, linewidth=220)
lo_deg=np.array([-15,-22,-50,-156,-140.]); hi_deg=np.array([20,22,80,-42,-35.])
LO=np.deg2rad(lo_deg)+0.15; HI=np.deg2rad(hi_deg)-0.15
def run(setup, cmd=10.0, joints=None):
    out = make_out()
    for idx, q, v in setup: set_motor(out, idx, pos=q, vel=v)
    if joints:
        for leg, name, q in joints: getattr(getattr(out, leg), name).position = q
    return core_step(new_core(), out, [cmd]*10)
print("knee lo zone d=.1:", run([(3, LO[3]-0.1, 0)]))
print("knee hi zone d=.1:", run([(3, HI[3]+0.1, 0)]))
print("pitch lo zone d=.1:", run([(2, LO[2]-0.1, 0)]))
print("pitch hi zone d=.1:", run([(2, HI[2]+0.1, 0)]))
print("pitch lo d=.22:", run([(2, LO[2]-0.22, 0)]))
print("roll hi d=.1:", run([(0, HI[0]+0.1, 0)]))
print("foot lo d=.1:", run([(4, LO[4]-0.1, 0)]))
# scan hip pitch + knee combos in free interval for extra torques
print("scan pitch/knee (both free): torque on pitch, knee with cmd 0")
for qp in np.linspace(LO[2]+0.01, HI[2]-0.01, 7):
    print(" pitch %.2f:" % qp, [tuple(np.round(run([(2, qp, 0), (3, qk, 0)], cmd=0.0)[[2,3]],1)) for qk in np.linspace(LO[3]+0.01, HI[3]-0.01, 7)])
# do joint encoders matter?
print("tarsus joint moved:", run([], joints=[("leftLeg","tarsusJoint",2.5)]), run([], joints=[("leftLeg","shinJoint",0.5)]))
