import numpy as np
from native_blocks import *
np.set_printoptions(precision=5, suppress=True, linewidth=220)
lo_deg=np.array([-15,-22,-50,-156,-140.]); hi_deg=np.array([20,22,80,-42,-35.])
LO=np.deg2rad(lo_deg)+0.15; HI=np.deg2rad(hi_deg)-0.15
def run(setup, cmd=0.0):
    out = make_out()
    for idx, q, v in setup: set_motor(out, idx, pos=q, vel=v)
    return core_step(new_core(), out, [cmd]*10)
# 1. threshold: fix pitch = -0.3, scan knee
qp = -0.3
print("knee scan at pitch -0.3 (sum, tau_pitch, tau_knee, others):")
for qk in np.linspace(-2.3, -1.9, 21):
    r = run([(2, qp, 0), (3, qk, 0)])
    print("  sum %.4f  tau %.4f %.4f   other %.3f" % (qp+qk, r[2], r[3], r[0]))
T = -0.75*np.pi
print("--- cmd = 10, d = 0.05:", run([(2, -0.3, 0), (3, T - 0.05 + 0.3, 0)], cmd=10.0))
print("--- cmd = 10, d = 0.10:", run([(2, -0.3, 0), (3, T - 0.10 + 0.3, 0)], cmd=10.0))
print("--- cmd = 10, d = 0.05, right leg:", run([(7, -0.3, 0), (8, T - 0.05 + 0.3, 0)], cmd=10.0))
for vp, vk in ((1.0, 0.0), (0.0, 1.0), (1.0, 1.0), (-2.0, 0.0)):
    r0 = run([(2, -0.3, 0), (3, T - 0.05 + 0.3, 0)]); r = run([(2, -0.3, vp), (3, T - 0.05 + 0.3, vk)])
    print("vel pitch %.1f knee %.1f at d=.05: dtau pitch %.4f knee %.4f" % (vp, vk, r[2]-r0[2], r[3]-r0[3]))
for vp, vk in ((1.0, 0.0), (0.0, 1.0)):
    r0 = run([(2, -0.3, 0), (3, T - 0.20 + 0.3, 0)]); r = run([(2, -0.3, vp), (3, T - 0.20 + 0.3, vk)])
    print("vel pitch %.1f knee %.1f at d=.20 (unclamped?): tau0 %.3f %.3f dtau %.4f %.4f" % (vp, vk, r0[2], r0[3], r[2]-r0[2], r[3]-r0[3]))
# interaction with the individual knee zone: knee LO = %.4f
print("LO knee", LO[3], "LO pitch", LO[2])
# both zones: pitch at its own low zone d=.05 and coupled d=.05
print("pitch own zone d=.05 + coupled d=.05:", run([(2, LO[2]-0.05, 0), (3, T - 0.05 - (LO[2]-0.05), 0)], cmd=10.0))
print("pitch own zone d=.05 alone:", run([(2, LO[2]-0.05, 0), (3, -1.2, 0)], cmd=10.0))
