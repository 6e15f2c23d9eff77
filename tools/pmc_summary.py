"""Per-kernel means of the counters in a rocprofv3 --pmc ... --output-format csv run.
usage: python tools/pmc_summary.py <dir with *_counter_collection.csv> [out.txt] [header line]"""
import collections
import csv
import glob
import sys


def main():
    files = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in files:
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    lines = [sys.argv[3] if len(sys.argv) > 3 else "# rocprofv3 --pmc: per-dispatch means"]
    for k in sorted(acc):
        n = max(len(v) for v in acc[k].values())
        lines.append(k + ": " + ", ".join("%s=%.4g" % (c, sum(v) / len(v)) for c, v in sorted(acc[k].items())) + "  (n=%d)" % n)
    txt = "\n".join(lines)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(txt + "\n")
    print(txt)


if __name__ == "__main__":
    main()
