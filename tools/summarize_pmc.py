#!/usr/bin/env python3
"""Aggregate a rocprofv3 --pmc counter_collection.csv per kernel.

usage: summarize_pmc.py <dir-with-*_counter_collection.csv> <COUNTER> [out.csv]
Prints launches, mean and total of COUNTER per kernel name.  FETCH_SIZE / WRITE_SIZE are KB
(counter_defs.yaml); on gfx950 FETCH_SIZE tallies 128-B requests as 64 B for wide coalesced
reads, so the bytes column doubles it (MI355X_MICROARCH.md, HBM section); WRITE_SIZE is exact.
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def shorten(k):
    k = k.replace("(anonymous namespace)::", "").replace("k2hip::", "").replace("void ", "")
    return k.split("(")[0][:110]


def main():
    d, counter = sys.argv[1], sys.argv[2]
    out = sys.argv[3] if len(sys.argv) > 3 else None
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        sys.exit("no counter_collection.csv under " + d)
    agg = defaultdict(lambda: [0, 0.0])
    for f in files:
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row.get("Counter_Name") != counter:
                    continue
                a = agg[row["Kernel_Name"]]
                a[0] += 1
                a[1] += float(row["Counter_Value"])
    scale = 1024.0 * (2.0 if counter == "FETCH_SIZE" else 1.0)
    rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
    lines = ["kernel,launches,mean_%s_KB,total_%s_KB,mean_bytes_corrected" % (counter, counter)]
    for k, (n, tot) in rows:
        short = shorten(k)
        lines.append('"%s",%d,%.1f,%.1f,%.0f' % (short, n, tot / n, tot, tot / n * scale))
    text = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(text)
    sys.stdout.write(text[:3000])


if __name__ == "__main__":
    main()
