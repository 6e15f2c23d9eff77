import numpy as np
from native_blocks import *
hi4 = np.deg2rad(-35) - 0.15
def run(idx, q, v, cmd=0.0):
    out = make_out(); set_motor(out, idx, pos=q, vel=v)
    return core_step(new_core(), out, [cmd]*10)[idx]
for d in (0.10, 0.14, 0.15, 0.16, 0.17, 0.18):
    t0 = run(4, hi4 + d, 0.0); tp = run(4, hi4 + d, 1.0); tn = run(4, hi4 + d, -1.0); tc = run(4, hi4+d, 0.0, cmd=5.0)
    print("d=%.2f  P-part %.5f  (model %.5f)   D(v=+1) %.5f  D(v=-1) %.5f   (d/.15)*7=%.5f  cmd-part %.5f" % (d, t0, -100*d*(1+d/0.15), tp - t0, tn - t0, d/0.15*7, tc - t0))
