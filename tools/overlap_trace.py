#!/usr/bin/env python3
"""Reads a rocprofv3 --kernel-trace CSV of the pipelined offline bench and answers: what does the search (k_greedy, on its own
stream, under the next batch's encoder) cost the encoder?  For every (kernel name, grid) of the encoder it compares the mean
duration of launches that overlap a k_greedy interval with those that do not, and sums the difference per step.
usage: overlap_trace.py <kernel_trace.csv> [steps]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
ker = [(r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Grid_Size_X", "") + "x" + r.get("Grid_Size_Y", "") + "x" + r.get("Grid_Size_Z", ""), r.get("Workgroup_Size_X", "")) for r in rows]
ker.sort(key=lambda k: k[1])
gre = [(s, e) for n, s, e, _, _ in ker if "k_greedy" in n]
print(f"{len(ker)} launches, {len(gre)} k_greedy, mean k_greedy {sum(e - s for s, e in gre) / max(len(gre), 1) / 1e3:.1f} us")


def overlap(s, e):
    return any(s < ge and e > gs for gs, ge in gre)


stat = defaultdict(lambda: [[], []])
for n, s, e, g, w in ker:
    if "k_greedy" in n:
        continue
    stat[(n.replace("void ", "").replace("k2hip::", "").replace("(anonymous namespace)::", "").split("(")[0][:70], g, w)][1 if overlap(s, e) else 0].append((e - s) / 1e3)
tot_extra = 0.0
lines = []
for k, (a, b) in stat.items():
    if a and b:
        ma, mb = sum(a) / len(a), sum(b) / len(b)
        extra = (mb - ma) * len(b)
        tot_extra += extra
        lines.append((extra, k, len(a), ma, len(b), mb))
lines.sort(reverse=True)
for extra, k, na, ma, nb, mb in lines[:25]:
    print(f"{extra / 1e3:8.3f} ms extra  {k[0][:60]:60s} grid {k[1]:>14s} wg {k[2]:>4s}  alone {na:5d} x {ma:8.1f} us   under search {nb:5d} x {mb:8.1f} us  ({mb / ma:.2f}x)")
print(f"sum over kernels: {tot_extra / 1e3:.3f} ms extra in the launches that ran under a search, {tot_extra / 1e3 / max(len(gre), 1):.3f} ms per search")
# gaps: idle time on the device between consecutive launches (any stream) inside / outside search intervals
