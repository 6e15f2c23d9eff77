"""Timeline of one PPO iteration from a rocprofv3 --kernel-trace .db (rocpd sqlite): for the last `n_steps` env_step_kernel launches, what runs between
consecutive env steps (kernel, duration, idle gap in front of it), and the totals per kernel name.  usage: python tools/rocprof_timeline.py <results.db> [out.txt]"""
import sqlite3, sys, collections

def _stepwise(rows, steps, a, b):
    out = []
    span = rows[b][1] - rows[a][1]
    out.append("# %d env steps: span %.3f ms = %.1f us per step" % (32, span / 1e6, span / 32e3))
    tot = collections.Counter(); cnt = collections.Counter(); gap = 0
    for i in range(a, b):
        n, s0, e0 = rows[i]; tot[n] += e0 - s0; cnt[n] += 1
        gap += max(0, rows[i + 1][1] - e0)
    out.append("# kernel time %.3f ms, idle between kernels %.3f ms" % (sum(tot.values()) / 1e6, gap / 1e6))
    for n, t in tot.most_common(): out.append("%-50s %5d calls %9.1f us total %8.2f us avg" % (n, cnt[n], t / 1e3, t / 1e3 / cnt[n]))
    out.append("# one step in detail (kernel, duration us, idle gap in front us)")
    for i in range(steps[-3], steps[-2] + 1):
        n, s0, e0 = rows[i]; out.append("  %-50s %8.2f %8.2f" % (n, (e0 - s0) / 1e3, (s0 - rows[i - 1][2]) / 1e3))
    gaps = sorted(((rows[i + 1][1] - max(r[2] for r in rows[max(a, i - 3):i + 1]), i) for i in range(a, b)), reverse=True)[:14]
    out.append("# largest idle gaps inside the span (us, after kernel -> before kernel)")
    for gp, i in gaps: out.append("  %8.1f  %s -> %s" % (gp / 1e3, rows[i][0][-40:], rows[i + 1][0][-40:]))
    return out


def main():
    db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
    views = [r[0] for r in cur.execute("select name from sqlite_master where type in ('view','table')").fetchall()]
    v = "kernels" if "kernels" in views else [x for x in views if "kernel" in x.lower()][0]
    cols = [d[1] for d in cur.execute("pragma table_info(%s)" % v).fetchall()]
    nm = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    st = "start" if "start" in cols else [c for c in cols if "start" in c][0]
    en = "end" if "end" in cols else [c for c in cols if "end" in c][0]
    rows = cur.execute("select %s, %s, %s from %s order by %s" % (nm, st, en, v, st)).fetchall()
    rows = [(str(n).split("(")[0][-48:], int(a), int(b)) for n, a, b in rows]
    steps = [i for i, r in enumerate(rows) if "env_step_kernel" in r[0]]
    out = []
    roll = [i for i, r in enumerate(rows) if "env_rollout_kernel" in r[0]]
    if len(roll) >= 2:      # round 5: the rollout is ONE launch -> the timeline of one iteration = from one rollout launch to the next
        a, b = roll[-2], roll[-1]
        span = rows[b][1] - rows[a][1]
        out.append("# one iteration, rollout launch to rollout launch: span %.3f ms; the rollout launch itself %.3f ms" % (span / 1e6, (rows[a][2] - rows[a][1]) / 1e6))
        tot = collections.Counter(); cnt = collections.Counter(); gap = 0
        for i in range(a, b):
            n, s0, e0 = rows[i]; tot[n] += e0 - s0; cnt[n] += 1
            gap += max(0, rows[i + 1][1] - max(r[2] for r in rows[max(a, i - 3):i + 1]))
        out.append("# kernel time %.3f ms, idle between kernels %.3f ms" % (sum(tot.values()) / 1e6, gap / 1e6))
        for n, t in tot.most_common(24): out.append("%-50s %5d calls %9.1f us total %8.2f us avg" % (n, cnt[n], t / 1e3, t / 1e3 / cnt[n]))
        out.append("# behind the rollout launch, up to the first optimiser step (kernel, duration us, idle gap in front us)")
        i = a + 1
        while i < b and "ppo_head_kernel" not in rows[i][0] and i < a + 60:
            n, s0, e0 = rows[i]; out.append("  %-50s %8.2f %8.2f" % (n, (e0 - s0) / 1e3, (s0 - rows[i - 1][2]) / 1e3)); i += 1
        gaps = sorted(((rows[i + 1][1] - max(r[2] for r in rows[max(a, i - 3):i + 1]), i) for i in range(a, b)), reverse=True)[:14]
        out.append("# largest idle gaps inside the iteration (us, after kernel -> before kernel)")
        for gp, i in gaps: out.append("  %8.1f  %s -> %s" % (gp / 1e3, rows[i][0][-40:], rows[i + 1][0][-40:]))
        steps = []
    elif len(steps) < 40: out.append("too few env steps"); print("\n".join(out)); return
    if len(steps) >= 40:
      a, b = steps[-34], steps[-2]          # 32 consecutive env steps inside the last iteration(s)
      out += _stepwise(rows, steps, a, b)
    pre = [i for i, r in enumerate(rows) if "ppo_loss_kernel" in r[0]]
    if len(pre) > 3:
        out.append("# one minibatch of the learner in detail (kernel, duration us, idle gap in front us)")
        i0 = pre[-2]; 
        while i0 > 0 and "clip_adam" not in rows[i0 - 1][0]: i0 -= 1
        i1 = pre[-1]
        while i1 > 0 and "clip_adam" not in rows[i1 - 1][0]: i1 -= 1
        for i in range(i0, i1):
            n, s0, e0 = rows[i]; out.append("  %-50s %8.2f %8.2f" % (n, (e0 - s0) / 1e3, (s0 - rows[i - 1][2]) / 1e3))
        out.append("# minibatch span %.1f us, kernel time %.1f us" % ((rows[i1][1] - rows[i0][1]) / 1e3, sum(rows[i][2] - rows[i][1] for i in range(i0, i1)) / 1e3))
    txt = "\n".join(out)
    if len(sys.argv) > 2: open(sys.argv[2], "w").write(txt + "\n")
    print(txt)

if __name__ == "__main__":
    main()
