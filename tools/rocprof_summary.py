"""Turn a rocprofv3 results .db (rocpd sqlite, --kernel-trace --stats) into the per-kernel summary text we commit
under profiles/.  usage: python tools/rocprof_summary.py <results.db> [out.txt]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    rows = cur.execute("select * from top_kernels").fetchall()
    cols = [d[0] for d in cur.description]
    lines = ["# rocprofv3 --kernel-trace --stats : per-kernel summary (view top_kernels)", " | ".join(cols)]
    for r in rows[:40]:
        lines.append(" | ".join(str(x)[:70] for x in r))
    txt = "\n".join(lines)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(txt + "\n")
    print(txt)


if __name__ == "__main__":
    main()
