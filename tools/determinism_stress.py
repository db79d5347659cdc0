#!/usr/bin/env python3
"""Dev tool (GPU): is the encoder bit-reproducible while ANOTHER process keeps the same GPU busy?  (Two ranks sharing one card is how
the multi-rank path is rehearsed on a one-GPU box; kernels of the two processes interleave, which shifts every timing.)
usage: determinism_stress.py [iterations] [switch=value ...]
Runs the headline architecture's encoder on one fixed batch `iterations` times next to a child process looping the same model, and
compares every output with the first one, bit for bit; on a mismatch it reports the first encoder tap (stack output) that differs."""
import os
import subprocess
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import k2transducerasr_amd as pkg
from k2transducerasr_amd.synth import synth_utterance, write_synthetic_model

iters = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 60
child = "--child" in sys.argv
path = "/tmp/k2hip_bench_zipformer2-large-en.k2w"
if not os.path.exists(path):
    write_synthetic_model(path, "zipformer2-large-en")
for a in sys.argv[1:]:
    if "=" in a:
        k, v = a.split("=")
        os.environ[k] = v
B, secs = 16, 6.0
utts = np.stack([synth_utterance(40 + u, secs) for u in range(B)])
if child:
    m = pkg.Model(path, 0)
    while True:
        m.offline_greedy_from_samples(list(utts))
# (the child is started before this process touches the GPU)
proc = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--child"] + [a for a in sys.argv[1:] if "=" in a])
m = pkg.Model(path, 0)
try:
    ref = None
    bad = 0
    for it in range(iters):
        got = m.offline_greedy_from_samples(list(utts))
        if ref is None:
            ref = got
        elif got != ref:
            bad += 1
            d = [i for i in range(B) if got[i] != ref[i]]
            print(f"iteration {it}: tokens of streams {d} differ from the first run")
    print(f"{iters} iterations, {bad} differed")
finally:
    proc.kill()
    proc.wait()
