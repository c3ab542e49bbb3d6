#!/usr/bin/env python3
"""Condense rocprofv3 CSV output into the small summaries kept under profiles/.

  kernel stats : profile_summarise.py stats  <dir> <out.csv>      (copies the *_kernel_stats.csv of --kernel-trace --stats)
  counters     : profile_summarise.py pmc    <dir> [<dir> ...] <out.csv>
                 per kernel name: launches and the per-launch mean of every counter found in *_counter_collection.csv
"""
import csv, glob, os, sys
from collections import defaultdict


def find(d, suffix):
    hits = sorted(glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True))
    if not hits:
        raise SystemExit("no *%s under %s" % (suffix, d))
    return hits[0]


def short(name):
    return name.split("(")[0].replace("void ", "")


def main():
    mode = sys.argv[1]
    if mode == "stats":
        src = find(sys.argv[2], "kernel_stats.csv")
        with open(src) as f, open(sys.argv[3], "w") as g:
            g.write(f.read())
        print("wrote", sys.argv[3])
    elif mode == "pmc":
        dirs, out = sys.argv[2:-1], sys.argv[-1]
        acc = defaultdict(lambda: defaultdict(float))   # kernel -> counter -> sum
        cnt = defaultdict(lambda: defaultdict(set))     # kernel -> counter -> dispatch ids
        for d in dirs:
            with open(find(d, "counter_collection.csv")) as f:
                for row in csv.DictReader(f):
                    k = short(row["Kernel_Name"])
                    c = row["Counter_Name"]
                    acc[k][c] += float(row["Counter_Value"])
                    cnt[k][c].add(row["Dispatch_Id"])
        counters = sorted({c for k in acc for c in acc[k]})
        with open(out, "w") as g:
            g.write("# per-launch means; rocprofv3 --pmc, one pass per directory: %s\n" % " ".join(os.path.basename(os.path.normpath(d)) for d in dirs))
            g.write("kernel,launches," + ",".join(counters) + "\n")
            for k in sorted(acc, key=lambda k: -sum(acc[k].values())):
                n = max(len(cnt[k][c]) for c in cnt[k])
                g.write("%s,%d,%s\n" % (k.replace(",", ";"), n, ",".join("%.6g" % (acc[k][c] / max(len(cnt[k][c]), 1)) if c in acc[k] else "" for c in counters)))
        print("wrote", out)
    else:
        raise SystemExit(__doc__)


if __name__ == "__main__":
    main()
