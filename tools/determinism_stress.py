#!/usr/bin/env python3
"""Dev tool (GPU): is the encoder bit-reproducible while ANOTHER process keeps the same GPU busy?  (Two ranks sharing one card is how
the multi-rank path is rehearsed on a one-GPU box; kernels of the two processes interleave, which shifts every timing.)
usage: determinism_stress.py [iterations] [preset=<offline or streaming preset>] [beam=N] [SWITCH=value ...]
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
preset, beam = "zipformer2-large-en", 0
for a in sys.argv[1:]:
    if a.startswith("preset="):
        preset = a.split("=")[1]
    elif a.startswith("beam="):
        beam = int(a.split("=")[1])
    elif "=" in a:
        k, v = a.split("=")
        os.environ[k] = v
path = f"/tmp/k2hip_bench_{preset}.k2w"
if not os.path.exists(path):
    tmp = path + f".tmp{os.getpid()}"
    write_synthetic_model(tmp, preset)
    os.replace(tmp, path)
streaming = "streaming" in preset
B, secs = (24, 4.0) if streaming else (16, 6.0)
utts = np.stack([synth_utterance(40 + u, secs) for u in range(B)])


def make():
    if streaming:
        rec = pkg.OnlineRecognizer(path)

        def run():
            ss = [rec.create_online_stream() for _ in range(B)]
            step = 3200  # 200 ms pushes
            for off in range(0, utts.shape[1], step):
                rec.add_samples_batch(ss, utts[:, off : off + step])
                rec.get_results(ss)
            out = [(s.tokens, s.timestamps) for s in ss]
            for s_ in ss:
                s_.close()
            return out
        return run
    m = pkg.Model(path, 0)
    if beam > 0:
        m.set_decoding_method("modified_beam_search", beam)
    return lambda: m.offline_greedy_from_samples(list(utts))


if child:
    run = make()
    while True:
        run()
# (the child is started before this process touches the GPU)
proc = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--child"] + [a for a in sys.argv[1:] if "=" in a])
run = make()
try:
    ref = None
    bad = 0
    for it in range(iters):
        got = run()
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
