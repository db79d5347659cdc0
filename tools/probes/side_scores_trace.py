#!/usr/bin/env python3
"""Reads a rocprofv3 --kernel-trace CSV of the offline bench and lists, for the attention-scores launches, which launches of other
queues ran at the same time and for how long (does the side queue really overlap the first feed-forward?).
usage: side_scores_trace.py <kernel_trace.csv>"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ker = sorted(((r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("k2hip::", "").replace("void ", "").split("(")[0][:48], int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?") + "/s" + r.get("Stream_Id", "?")) for r in rows), key=lambda k: k[1])
sc = [i for i, k in enumerate(ker) if "k_attn_scores_softmax" in k[0]]
print(f"{len(ker)} launches, {len(sc)} scores launches; queues: {sorted(set(k[3] for k in ker))}")
from collections import Counter
for q in sorted(set(k[3] for k in ker)):
    c = Counter(k[0] for k in ker if k[3] == q)
    print(f"queue {q}: " + ", ".join(f"{n} x{v}" for n, v in c.most_common(6)))
ov_tot = dur_tot = 0
shown = 0
for i in sc[len(sc) // 2:]:
    n, s, e, q = ker[i]
    others = [(k[0], k[3], min(e, k[2]) - max(s, k[1])) for k in ker[max(0, i - 6):i + 7] if k is not ker[i] and "k_greedy" not in k[0] and k[1] < e and k[2] > s]
    ov = sum(o[2] for o in others)
    ov_tot += ov
    dur_tot += e - s
    if shown < 12:
        shown += 1
        prev = ker[i - 1]
        print(f"scores q{q} {(e - s) / 1e3:6.1f} us, starts {(s - prev[2]) / 1e3:6.1f} us after the end of {prev[0]} (q{prev[3]}); overlapping: " + ", ".join(f"{o[0]} q{o[1]} {o[2] / 1e3:.1f} us" for o in others))
print(f"overlapped share of the scores launches' time: {ov_tot / max(dur_tot, 1):.2f}")
