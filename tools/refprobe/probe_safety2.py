import numpy as np
from native_blocks import *
np.set_printoptions(precision=4, suppress=True, linewidth=200)
HI = [0.199066, 0.233972, 1.246263, -0.883038, -0.760865]
LO = [-0.111799, -0.233972, -0.722665, -2.572714, -2.293461]
def run(setup, cmd=10.0):
    out = make_out()
    for idx, q, v in setup: set_motor(out, idx, pos=q, vel=v)
    return core_step(new_core(), out, [cmd]*10)
print("two joints, left leg: j0 d=.05, j1 d=.10, cmd=10:", run([(0, HI[0]+0.05, 0), (1, HI[1]+0.10, 0)]))
print("   expected if s=min: s=.333 -> others 3.33; j0: s*10 + (1-s0)*(-1000*(.15+.05)) ; ", 0.3333*10 - (0.05/0.15)*1000*0.2, " j1:", 0.3333*10 - (0.10/0.15)*800*0.25)
print("   expected if s=prod: s=.222 -> others 2.22")
print("left j0 d=.05, right j0 (mirror) d=.08:", run([(0, HI[0]+0.05, 0), (5, -HI[0]-0.08, 0)]))
print("right leg free interval check j5:", [run([(5, q, 0)], cmd=0)[5] for q in (-0.21, -0.20, -0.19, 0.10, 0.112, 0.12)])
print("right yaw j6:", [run([(6, q, 0)], cmd=0)[6] for q in (-0.24, -0.23, 0.23, 0.24)])
# does the velocity come from the drive velocity field?  and is the position the drive position (not joint)?
print("vel only in zone:", run([(0, 0.0, 5.0)], cmd=0)[:5])
# hysteresis / state? call twice on same core
core = new_core(); out = make_out(); set_motor(out, 0, pos=HI[0]+0.05)
print("call1", core_step(core, out, [0]*10)[:2], "call2", core_step(core, out, [0]*10)[:2])
out2 = make_out()
print("after leaving zone", core_step(core, out2, [5]*10)[:3])
# radio gate
o = make_out(); o.pelvis.radio.channel[8] = 0.0
print("radio ch8=0:", core_step(new_core(), o, [5]*10))
o = make_out(); o.pelvis.radio.channel[8] = -1.0
print("radio ch8=-1:", core_step(new_core(), o, [5]*10))
# far beyond: d=0.2, 0.25 for joint 4 and joint 0
for idx in (0, 4):
    print("joint", idx, "deep:", [(d, round(run([(idx, HI[idx]+d, 0)], cmd=0)[idx], 3)) for d in (0.14, 0.15, 0.16, 0.18, 0.19, 0.2, 0.25, 0.3)])
