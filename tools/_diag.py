import sys, torch, numpy as np
sys.path.insert(0, "/root/repo")
from apex_amd.vecenv import CassieVecEnv
e = CassieVecEnv(n_envs=64, seed=52); e.reset()
z = torch.zeros(64, 10, device=e.device)
for _ in range(3): e.step(z, auto_reset=False)
qp = e.get_field("qpos"); qp[5, 2] = float("nan"); qp[11, 2] = float("inf"); e.set_field("qpos", qp)
names = ["qpos","qvel","qacc_warm","mass","damping","friction","floor","body_invweight0","dof_invweight0","motor_noise","joint_noise","pd_target","tq_fifo","so_mpos","so_mvel","so_torque","so_jpos","so_jvel","so_quat","so_rotvel","so_tvel","so_tacc","so_height","foot_vel","prev_action","prev_torque","cmd","fwd","xfrc","menc","jenc_x","jenc_y","snap","foot_prev","est"]
for t in range(4):
    obs, rew, done, _ = e.step(z, auto_reset=True)
    bad = [(n, sorted(set(np.argwhere(~np.isfinite(e.get_field(n).cpu().numpy()))[:, 0].tolist()))) for n in names]
    print(t, "done", done.cpu().numpy().nonzero()[0], "obs bad", np.argwhere(~np.isfinite(obs.cpu().numpy())).tolist()[:10], [b for b in bad if b[1]])
