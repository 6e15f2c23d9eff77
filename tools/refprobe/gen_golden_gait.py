"""Golden G23: the reference's own data file cassie/trajectory/stepdata.bin (one gait cycle of Agility's Cassie simulator at 2 kHz: time, qpos [35], qvel [32], ...;
cassie/trajectory/trajectory.py:4-22) subsampled to every 4th sample as float32 - 421 poses of an EXTERNALLY simulated walking robot.  tests/test_oracle_env.py
holds the oracle's kinematic chain to it (the stance foot lies on the floor, does not slide; the swing foot clears the floor), its inertial model (momentum balance
about the stance foot) and its smooth dynamics (inverse dynamics of the swing leg against the RECORDED motor torques)."""
from common import REF, GOLD
import os
import numpy as np

n = 1 + 35 + 32 + 10 + 10 + 10
d = np.fromfile(os.path.join(REF, "cassie", "trajectory", "stepdata.bin"), dtype=np.double).reshape(-1, n)
sub = d[::4]
# qacc: central difference of the recorded velocities at the FULL 2 kHz rate (the recorded joint torques are compared with the oracle's inverse dynamics on it)
qacc = np.gradient(d[:, 36:68], d[1, 0] - d[0, 0], axis=0)[::4]
np.savez_compressed(os.path.join(GOLD, "g23_agility_gait.npz"), time=sub[:, 0], qpos=sub[:, 1:36].astype(np.float32), qvel=sub[:, 36:68].astype(np.float32),
                    torque=sub[:, 68:78].astype(np.float32), qacc=qacc.astype(np.float32),
                    dt_full=np.float64(d[1, 0] - d[0, 0]), n_full=np.int64(len(d)))
print("wrote g23_agility_gait.npz", sub.shape)
