"""Per-launch durations of the learner kernels of ONE minibatch, from a rocprofv3 --kernel-trace csv (run on the GPU box):
  cd /tmp && rocprofv3 --kernel-trace --output-format csv -d OUT -- python tools/t_pmc_learner.py ; python tools/t_learner_trace.py OUT"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"].split("(")[0][:40] for r in rows]
# last ppo_loss_kernel -> following kernels until the next clip_adam pair
idx = [i for i, n in enumerate(names) if n.startswith("ppo_loss_kernel")]
i0 = idx[-1]
j0 = max(0, i0 - 6)
t0 = int(rows[j0]["Start_Timestamp"])
for r, n in list(zip(rows, names))[j0:i0 + 18]:
    print("%-42s start %8.1f us  dur %7.1f us  grid %s" % (n, (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Grid_Size", "")))
