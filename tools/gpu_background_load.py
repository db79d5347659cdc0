#!/usr/bin/env python3
"""Dev tool (GPU): keep the card busy with the headline model (offline batches back to back) for N seconds -- run next to the test
suite to shift every kernel's timing (how the race in the pipelined GEMM's fragment reads was found).  usage: gpu_background_load.py [seconds]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import k2transducerasr_amd as pkg
from k2transducerasr_amd.synth import synth_utterance, write_synthetic_model

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
path = "/tmp/k2hip_bench_zipformer2-large-en.k2w"
if not os.path.exists(path):
    tmp = path + f".tmp{os.getpid()}"
    write_synthetic_model(tmp, "zipformer2-large-en")
    os.replace(tmp, path)
m = pkg.Model(path, 0)
utts = [synth_utterance(7 + u, 5.0) for u in range(16)]
t0 = time.time()
n = 0
while time.time() - t0 < secs:
    m.offline_greedy_from_samples(utts)
    n += 1
print(f"background load: {n} batches in {time.time() - t0:.0f} s")
