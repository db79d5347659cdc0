#!/usr/bin/env python3
"""Per-shape GEMM time of one instrumented streaming tick (dev tool, GPU box): tools/gemm_profile.py for bench_streaming.py's workload.

usage: gemm_profile_streaming.py [preset] [streams]
Every ready stream gets exactly one chunk; the tick's GEMM launches grouped by (M, N, K, batch, act, res, kind), sorted by total time."""
import os
import sys
from collections import defaultdict

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import k2transducerasr_amd as pkg  # noqa: E402
from k2transducerasr_amd.synth import synth_utterance, write_synthetic_model  # noqa: E402

preset = sys.argv[1] if len(sys.argv) > 1 else "zipformer2-streaming-zh"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 128
path = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"k2hip_bench_{preset}.k2w")
if not os.path.exists(path):
    write_synthetic_model(path, preset)
os.environ.setdefault("K2HIP_MAX_STREAMS", str(max(256, N)))
rec = pkg.OnlineRecognizer(path)
wave = np.stack([synth_utterance(1000 + u, 2.0) for u in range(N)])
need = (rec.chunk_length - 1) * 160 + 400
for rep in range(3):   # two warm ticks, then the instrumented one
    hs = [rec.create_online_stream() for _ in range(N)]
    rec.add_samples_batch(hs, wave[:, :need])
    rec.model.set_instrument(rep == 2)
    dec, _ = rec.get_results(hs)
    assert all(dec)
    rows = rec.model.gemm_profile() if rep == 2 else None
    rec.model.set_instrument(False)
    for h in hs:
        h.close()
agg = defaultdict(lambda: [0, 0.0])
for r in rows:
    k = tuple(int(x) for x in r[:7])
    agg[k][0] += 1
    agg[k][1] += float(r[7])
tot = sum(v[1] for v in agg.values())
print(f"{preset}, {N} streams: {len(rows)} GEMM launches, {tot / 1e3:.3f} ms (HIP events around each launch)")
print(f"{'M':>7} {'N':>5} {'K':>5} {'bat':>4} act res kind |  n   us/launch  total_us   TF/s   cum%")
cum = 0.0
for k, (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    M, Nn, K, bat, act, res, kind = k
    cum += us
    kname = "pipe" if kind & 128 else "ring" if kind & 64 else "skinny" if kind & 32 else "dma" if kind & 16 else "classic"
    print(f"{M:7d} {Nn:5d} {K:5d} {bat:4d} {act:3d} {res:3d} {kname:>7} | {n:2d} {us / n:10.1f} {us:9.1f} {2.0 * M * Nn * K * bat * n / us / 1e6:6.1f} {100 * cum / tot:6.1f}")
