"""Harness for the reference's MuJoCo-free native blocks (pd_input_step, cassie_core_sim_step, state_output_step in
libcassiemujoco.so), callable through the reference's own ctypes module loaded by file path.
Runs only in the build container."""
import importlib.util
import os
import sys

import numpy as np

sys.dont_write_bytecode = True
_spec = importlib.util.spec_from_file_location("cm", "/root/reference/cassie/cassiemujoco/cassiemujoco_ctypes.py")
_cwd = os.getcwd()
os.chdir("/root/reference/cassie/cassiemujoco")          # the module dlopens ./libcassiemujoco.so
cm = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(cm)
os.chdir(_cwd)

DRIVES = ["hipRollDrive", "hipYawDrive", "hipPitchDrive", "kneeDrive", "footDrive"]
JOINTS = ["shinJoint", "tarsusJoint", "footJoint"]
GEAR = [25, 25, 16, 16, 50]
TLIM = [140.63, 140.63, 216.16, 216.16, 45.14]
NOMINAL = [0.0045, 0.0, 0.4973, -1.1997, -1.5968]
NOM_JOINT = [0.0, 1.4267, -1.5968]


def make_out():
    """cassie_out_t preset like cassie_sim_init's template."""
    o = cm.cassie_out_t()
    o.isCalibrated = True
    o.pelvis.radio.channel[8] = 1.0
    o.pelvis.radio.radioReceiverSignalGood = True
    o.pelvis.radio.receiverMedullaSignalGood = True
    o.pelvis.vectorNav.orientation[0] = 1.0
    o.pelvis.vectorNav.dataGood = True
    for leg, sgn in ((o.leftLeg, 1), (o.rightLeg, -1)):
        for k, name in enumerate(DRIVES):
            d = getattr(leg, name)
            d.gearRatio = GEAR[k]; d.torqueLimit = TLIM[k]; d.dcLinkVoltage = 48.0
            d.position = NOMINAL[k] * (sgn if k == 0 else 1)
        for k, name in enumerate(JOINTS):
            getattr(leg, name).position = NOM_JOINT[k]
    return o


def set_motor(o, idx, pos=None, vel=None):
    leg = o.leftLeg if idx < 5 else o.rightLeg
    d = getattr(leg, DRIVES[idx % 5])
    if pos is not None: d.position = pos
    if vel is not None: d.velocity = vel


def core_step(core, out, torques):
    u = cm.cassie_user_in_t()
    for i in range(10): u.torque[i] = float(torques[i])
    cin = cm.cassie_in_t()
    cm.cassie_core_sim_step(core, u, out, cin)
    res = []
    for leg in (cin.leftLeg, cin.rightLeg):
        for name in DRIVES:
            res.append(getattr(leg, name).torque)
    return np.array(res)


def new_core():
    c = cm.cassie_core_sim_alloc(); cm.cassie_core_sim_setup(c); return c


if __name__ == "__main__":
    core = new_core()
    out = make_out()
    print("nominal pass-through:", core_step(core, out, [1, 2, 3, 4, 5, -1, -2, -3, -4, -5]))
    print("clamp:", core_step(core, out, [500] * 10))
    # sweep left hip roll position, zero command
    for idx, lo, hi in ((0, -0.30, 0.45), (1, -0.45, 0.45), (2, -0.95, 1.45), (3, -2.9, -0.6), (4, -2.5, -0.5)):
        print("joint", idx)
        for q in np.linspace(lo, hi, 16):
            out = make_out(); set_motor(out, idx, pos=q)
            t0 = core_step(new_core(), out, [0] * 10)
            t1 = core_step(new_core(), out, [10] * 10)
            print("  q=%+.3f  tau(cmd 0)=%s  scale=%.3f" % (q, np.round(t0[[idx]], 2), (t1 - t0)[(idx + 1) % 10] / 10))
