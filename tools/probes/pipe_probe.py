"""Dev probe (GPU): ms per pipelined offline step (two batches in flight), as bench.py times it but without its result checks."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import k2transducerasr_amd as pkg
from k2transducerasr_amd.synth import synth_utterance, write_synthetic_model

path = "/tmp/k2hip_probe_large.k2w"
if not os.path.exists(path):
    write_synthetic_model(path, "zipformer2-large-en")
m = pkg.Model(path, 0)
B = 32
s = np.stack([synth_utterance(u, 10.0) for u in range(B)])
ptr = m.device_alloc(s.nbytes)
m.device_upload(ptr, s)


def run(n):
    tk = m.offline_submit_samples_dev(ptr, s.shape[1], B)
    for _ in range(n - 1):
        nxt = m.offline_submit_samples_dev(ptr, s.shape[1], B)
        m.offline_wait(tk)
        tk = nxt
    return m.offline_wait(tk)


run(3)
m.synchronize()
t0 = time.perf_counter()
run(12)
m.synchronize()
print("ms per step: %.3f" % ((time.perf_counter() - t0) * 1e3 / 12))
