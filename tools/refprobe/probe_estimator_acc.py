import sys, numpy as np
sys.path.insert(0, "/root/repo/tools/refprobe"); sys.path.insert(0, "/root/repo")
from native_blocks import cm, make_out
np.set_printoptions(precision=5, suppress=True, linewidth=200)
def run(quat, gyro, acc, n=400, gyro_fn=None):
    est = cm.state_output_alloc(); cm.state_output_setup(est)
    res = []
    for t in range(n):
        o = make_out()
        g = gyro if gyro_fn is None else gyro_fn(t)
        for k in range(4): o.pelvis.vectorNav.orientation[k] = quat[k]
        for k in range(3): o.pelvis.vectorNav.angularVelocity[k] = g[k]; o.pelvis.vectorNav.linearAcceleration[k] = acc[k]
        so = cm.state_out_t(); cm.state_output_step(est, o, so)
        res.append((np.array(so.pelvis.translationalAcceleration[:]), np.array(so.pelvis.translationalVelocity[:]), np.array(so.pelvis.rotationalVelocity[:]), so.pelvis.position[2], so.terrain.height))
    return res
q0 = [1, 0, 0, 0]
r = run(q0, [0, 0, 0], [0, 0, 9.806]); print("static:", r[0][0], r[-1][0], "vel", r[-1][1])
r = run(q0, [0, 0, 0], [1.0, 0, 9.806]); print("ax=1:", r[0][0], r[1][0], r[-1][0], "vel", r[10][1], r[-1][1])
for w in ([1, 0, 0], [0, 1, 0], [0, 0, 1], [2, 0, 0], [1, 1, 0]):
    r = run(q0, w, [0, 0, 9.806]); print("gyro", w, "acc out:", r[0][0], r[-1][0], " vel", r[-1][1])
# angular acceleration: gyro ramp 0 -> 1 rad/s over 200 samples (alpha = 10 rad/s^2 at 2 kHz)
r = run(q0, None, [0, 0, 9.806], gyro_fn=lambda t: [0.005 * t, 0, 0]); print("alpha_x=10:", r[1][0], r[100][0], r[-1][0])
r = run(q0, None, [0, 0, 9.806], gyro_fn=lambda t: [0, 0.005 * t, 0]); print("alpha_y=10:", r[1][0], r[100][0], r[-1][0])
# tilted
import math
th = 0.2; qt = [math.cos(th / 2), math.sin(th / 2), 0, 0]
r = run(qt, [0, 0, 0], [0, 0, 9.806]); print("roll 0.2, acc (0,0,g):", r[-1][0], " expect a - R^T g =", np.array([0, 0, 9.806]) - np.array([0, math.sin(th) * 9.806, math.cos(th) * 9.806]))
