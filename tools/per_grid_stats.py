#!/usr/bin/env python3
"""Dev tool: per (kernel, grid) durations from a rocprofv3 kernel trace CSV.  usage: per_grid_stats.py trace.csv [batches] [filter]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
nb = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
flt = sys.argv[3] if len(sys.argv) > 3 else ""
g = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"]
    if flt and flt not in n:
        continue
    short = n.replace("void ", "").replace("k2hip::", "").replace("(anonymous namespace)::", "").split("(")[0]
    wg = int(r["Workgroup_Size_X"])
    key = (short[:48], int(r["Grid_Size_X"]) // wg, int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]), wg)
    g[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
out = sorted(((sum(v) / nb / 1e3, k, len(v) / nb, sum(v) / len(v) / 1e3) for k, v in g.items()), reverse=True)
print(f"{'us/batch':>9} {'launches':>8} {'avg us':>8}  kernel, grid (workgroups), workgroup size")
for t, k, c, a in out:
    print(f"{t:9.1f} {c:8.1f} {a:8.2f}  {k[0]} {k[1]}x{k[2]}x{k[3]} x{k[4]}")
