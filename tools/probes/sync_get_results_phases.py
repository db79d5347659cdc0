#!/usr/bin/env python3
"""Where the synchronous GetResults call shape spends its host time (dev tool, GPU box): B x (create + AddSamples), GetResults, token
pulls, destroys -- each phase timed over a few batches of 32 x 10 s.  usage: sync_get_results_phases.py [preset] [B] [seconds]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import k2transducerasr_amd as pkg  # noqa: E402
from k2transducerasr_amd.synth import synth_utterance, write_synthetic_model  # noqa: E402

preset = sys.argv[1] if len(sys.argv) > 1 else "zipformer2-large-en"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
secs = float(sys.argv[3]) if len(sys.argv) > 3 else 10.0
path = f"/tmp/k2hip_prof_{preset}.k2w"
if not os.path.exists(path):
    write_synthetic_model(path, preset)
rec = pkg.OfflineRecognizer(path)
wav = [synth_utterance(u, secs) for u in range(B)]
acc = {"create+add": [], "get_results": [], "pull": [], "destroy": []}
for it in range(8):
    t0 = time.perf_counter()
    ss = [rec.create_offline_stream() for _ in range(B)]
    for s, w in zip(ss, wav):
        s.add_samples(w)
    t1 = time.perf_counter()
    rec.get_results(ss)
    t2 = time.perf_counter()
    tm = rec.model.timing()
    out = [(s.tokens, s.timestamps) for s in ss]
    t3 = time.perf_counter()
    for s in ss:
        s.close()
    t4 = time.perf_counter()
    if it >= 2:
        for k, v in zip(acc, (t1 - t0, t2 - t1, t3 - t2, t4 - t3)):
            acc[k].append(v * 1e3)
print({k: round(float(np.mean(v)), 3) for k, v in acc.items()}, "ms per batch; device stages of the last GetResults:", tm, "; copy alone:", end=" ")
dst = np.empty(sum(w.size for w in wav), np.float32)
t0 = time.perf_counter()
for _ in range(5):
    o = 0
    for w in wav:
        dst[o:o + w.size] = w
        o += w.size
print(round((time.perf_counter() - t0) / 5 * 1e3, 3), "ms for numpy copies of the same arrays into pageable memory")
