"""Golden G11: the reference's own state estimator, `state_output_step` of libcassiemujoco.so, on a 2 kHz sensor
stream of this build's oracle while the trained policy stands up and walks (60 env steps = 3000 substeps, start-up included), plus
known-answer vectors of three of its internal routines called directly inside the loaded binary (the closure residual 0x18e00, the
2 x 3 mldivide 0x21000, one horizontal-filter step 0x1cd10).  Runs only in the build container (needs /root/reference).

The sensor stream is rounded to float32 BEFORE it is fed to the binary (except the unit quaternion, kept in fp64), so the fixture stores exactly
what the binary saw.
usage: python tools/refprobe/gen_golden_estimator.py [run dir with actor.pt]"""
import sys, os, ctypes
import numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from native_blocks import cm, make_out, DRIVES, JOINTS
from oracle.sim import OracleEnv
from common import GOLD

run_dir = sys.argv[1] if len(sys.argv) > 1 else "/root/repo/trained_models/r03_cassie_v0_clock"
policy = torch.load(os.path.join(run_dir, "actor.pt"), weights_only=False); policy.eval()
lib = cm._libraries["./libcassiemujoco.so"]
base = ctypes.cast(lib.state_output_step, ctypes.c_void_p).value - 0x296b0      # load address of the library (state_output_step sits at 0x296b0)
D = ctypes.c_double


def feed(est, s26):
    out = make_out()
    for i in range(10):
        getattr(out.leftLeg if i < 5 else out.rightLeg, DRIVES[i % 5]).position = float(s26[i])
    for i in range(6):
        getattr(out.leftLeg if i < 3 else out.rightLeg, JOINTS[i % 3]).position = float(s26[10 + i])
    for k in range(4): out.pelvis.vectorNav.orientation[k] = float(s26[16 + k])
    for k in range(3): out.pelvis.vectorNav.angularVelocity[k] = float(s26[20 + k]); out.pelvis.vectorNav.linearAcceleration[k] = float(s26[23 + k])
    so = cm.state_out_t(); cm.state_output_step(est, out, so)
    return np.concatenate([so.pelvis.position[:], so.pelvis.translationalVelocity[:], so.pelvis.translationalAcceleration[:], [so.terrain.height],
                           so.leftFoot.position[:], so.rightFoot.position[:], so.leftFoot.orientation[:], so.rightFoot.orientation[:]])


# ---- 1. the whole routine on a walking stream
est = cm.state_output_alloc(); cm.state_output_setup(est)
e = OracleEnv(dyn_rand=False, seed=3)
obs = e.reset(); obs = e.reset_for_test(); e.update_speed(1.0)
off = np.array([0.0045, 0, 0.4973, -1.1997, -1.5968] * 2)
sens, ref = [], []
for step in range(60):
    with torch.no_grad():
        act = policy(torch.tensor(obs, dtype=torch.float32), deterministic=True).numpy().astype(np.float64)
    e.set("pd_target", act + off); e.set("pd_P", [100, 100, 88, 96, 50] * 2); e.set("pd_D", [10, 10, 8, 9.6, 5] * 2)
    for sub in range(50):
        e.substep()
        s26 = np.concatenate([e.get("so_mpos"), e.get("so_jpos"), e.get("so_quat"), e.get("so_rotvel"), e.get("snap_acc")])
        q = s26[16:20].copy(); s26 = s26.astype(np.float32).astype(np.float64); s26[16:20] = q      # the quaternion stays fp64 (unit to 1e-16: a float32 one is not, and the binary does not normalise it)
        sens.append(s26); ref.append(feed(est, s26))
    ints = e.get("ints"); ints[0] += 1; ints[1] += 1
    if ints[1] > e.get("phaselen")[0]: ints[1] = 0; ints[2] += 1
    e.set("ints", ints); obs = e.obs()
sens, ref = np.array(sens), np.array(ref)
# a second pass over the first 300 samples after state_output_setup on the SAME object: the full reset restarts everything
cm.state_output_setup(est)
ref_again = np.array([feed(est, s) for s in sens[:300]])
assert np.array_equal(ref_again, ref[:300])

# ---- 2. internal routines, called inside the loaded binary
rng = np.random.default_rng(11)
f18e00 = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.c_void_p, D, ctypes.c_void_p, ctypes.c_void_p)(base + 0x18e00)
f21000 = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p)(base + 0x21000)
f1cd10 = ctypes.CFUNCTYPE(None, ctypes.c_void_p, D, D, D, D, D, D)(base + 0x1cd10)
res_in, res_out = [], []
for _ in range(40):
    mot = np.array([0.0045, 0, 0.4973, -1.1997, -1.5968]) + rng.uniform(-0.3, 0.3, 5); jn = np.array([0, 1.4267, -1.5968]) + rng.uniform(-0.15, 0.15, 3); x = rng.uniform(-0.15, 0.15)
    a = (D * 5)(*mot); b = (D * 3)(*jn); r = (D * 1)(0); gr = (D * 8)()
    f18e00(ctypes.addressof(a), ctypes.addressof(b), x, ctypes.addressof(r), ctypes.addressof(gr))
    res_in.append([mot[3], jn[0], jn[1], x]); res_out.append([r[0], gr[3], gr[5], gr[6], gr[7]])      # residual, d/d(knee, shin, tarsus, heel)
ml_in, ml_out = [], []
for _ in range(40):
    M = rng.normal(size=(2, 3)); tau = rng.normal(size=2)
    a = (D * 6)(*M.T.reshape(-1)); b = (D * 2)(*tau); o = (D * 3)()
    f21000(ctypes.addressof(a), ctypes.addressof(b), ctypes.addressof(o))
    ml_in.append(np.concatenate([M.reshape(-1), tau])); ml_out.append(o[:])
obj = (D * 99).from_address(ctypes.cast(est, ctypes.c_void_p).value + 0x6e8); cm.state_output_setup(est); template = np.array(obj[:])
hf_in, hf_out = [], []
for _ in range(40):
    x0 = rng.normal(size=6) * 0.3; x0[4] = rng.uniform(0, 1)
    L = rng.normal(size=(6, 6)) * 0.02; P0 = L @ L.T + np.eye(6) * 1e-5
    args = [rng.normal() * 0.1, rng.normal() * 0.3, rng.normal() * 0.3, float(rng.choice([0, 0.3, 30, 49, 51, 200])), float(rng.choice([0, 0.3, 30, 200])), rng.normal()]
    buf = (D * 99)(*template); buf[0:6] = list(x0); buf[58:94] = list(P0.T.reshape(-1))
    f1cd10(ctypes.addressof(buf), *args)
    o = np.array(buf[:])
    hf_in.append(np.concatenate([x0, P0.reshape(-1), args])); hf_out.append(np.concatenate([o[0:6], o[58:94].reshape(6, 6).T.reshape(-1)]))
np.savez_compressed(os.path.join(GOLD, "g11_state_estimator.npz"), sens=np.delete(sens, [16, 17, 18, 19], axis=1).astype(np.float32), quat=sens[:, 16:20], ref=ref[::5].astype(np.float64), ref_last=ref[-1],
                    res_in=np.array(res_in), res_out=np.array(res_out), ml_in=np.array(ml_in), ml_out=np.array(ml_out), hf_in=np.array(hf_in), hf_out=np.array(hf_out))
print("wrote g11_state_estimator.npz:", sens.shape, ref[::5].shape, os.path.getsize(os.path.join(GOLD, "g11_state_estimator.npz")), "bytes")
