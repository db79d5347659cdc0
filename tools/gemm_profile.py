#!/usr/bin/env python3
"""Per-shape GEMM time of one instrumented batch of the benchmark workload (dev tool, GPU box).

usage: gemm_profile.py [preset] [batch] [seconds]
Prints the launches of one offline batch grouped by (M, N, K, batch, act, res, kind), sorted by total time,
with the achieved TFLOP/s of each group -- the list to work down when raising roofline.frac.
"""
import os
import sys
from collections import defaultdict

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import k2transducerasr_amd as pkg  # noqa: E402
from k2transducerasr_amd.synth import synth_utterance, write_synthetic_model  # noqa: E402

preset = sys.argv[1] if len(sys.argv) > 1 else "zipformer2-large-en"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
secs = float(sys.argv[3]) if len(sys.argv) > 3 else 10.0
path = f"/tmp/k2hip_prof_{preset}.k2w"
if not os.path.exists(path):
    write_synthetic_model(path, preset)
m = pkg.Model(path, 0)
s = np.stack([synth_utterance(u, secs) for u in range(B)])
ptr = m.device_alloc(s.nbytes)
m.device_upload(ptr, s)
for _ in range(2):
    m.offline_greedy_from_samples_dev(ptr, s.shape[1], B)
m.set_instrument(True)
m.offline_greedy_from_samples_dev(ptr, s.shape[1], B)
rows = m.gemm_profile()
m.set_instrument(False)
agg = defaultdict(lambda: [0, 0.0])
for r in rows:
    k = tuple(int(x) for x in r[:7])
    agg[k][0] += 1
    agg[k][1] += float(r[7])
tot = sum(v[1] for v in agg.values())
print(f"{len(rows)} launches, {tot / 1e3:.3f} ms")
print(f"{'M':>7} {'N':>5} {'K':>5} {'bat':>4} act res kind |  n   us/launch  total_us   TF/s   cum%")
cum = 0.0
for k, (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    M, N, K, bat, act, res, kind = k
    fl = 2.0 * M * N * K * bat * n
    cum += us
    print(f"{M:7d} {N:5d} {K:5d} {bat:4d} {act:3d} {res:3d} {kind:4d} | {n:2d} {us / n:10.1f} {us:9.1f} {fl / us / 1e6:6.1f} {100 * cum / tot:6.1f}")
