"""Dev probe (GPU): what two half-size streaming ticks in flight at once buy.  One recognizer with N streams against two recognizers
(two engines, two HIP streams, two host threads) with N / 2 streams each, same audio, wall time per 320 ms of audio of all N streams.
The tick is a chain of ~380 dependent launches of ~10 us that hardly depend on the stream count; if two chains overlap on the GPU
the pair's time is what an engine that splits its ready streams into two groups on two HIP streams could reach.  Not part of the product."""
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import k2transducerasr_amd as pkg
from k2transducerasr_amd.synth import synth_utterance, write_synthetic_model

preset = sys.argv[1] if len(sys.argv) > 1 else "zipformer2-streaming-zh"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 128
G = int(sys.argv[3]) if len(sys.argv) > 3 else 2
seconds = 10.0
weights = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"k2hip_bench_{preset}.k2w")
if not os.path.exists(weights):
    write_synthetic_model(weights, preset)
os.environ.setdefault("K2HIP_MAX_STREAMS", "256")
wave = np.stack([synth_utterance(1000 + u, seconds) for u in range(N)])
n = wave.shape[1]


def drive(rec, w, out, barrier):
    for rep in range(2):   # warm-up, timed
        streams = rec.batch([rec.create_online_stream() for _ in range(w.shape[0])])
        if barrier:
            barrier.wait()
        t0 = time.perf_counter()
        steps = 0
        for pos in range(0, n, 800):
            rec.add_samples_batch(streams, w[:, pos : pos + 800])
            while True:
                dec, _ = rec.get_results(streams)
                steps += any(dec)
                if not any(dec):
                    break
        rec.model.synchronize()
        out[:] = [time.perf_counter() - t0, steps, [list(s.tokens) for s in streams]]
        for s in streams:
            s.close()


one = [0, 0, None]
rec = pkg.OnlineRecognizer(weights)
drive(rec, wave, one, None)
print(f"{preset}: one engine, {N} streams: {one[0] / one[1] * 1e3:.3f} ms per tick ({one[1]} ticks)", flush=True)
recs = [rec] + [pkg.OnlineRecognizer(weights) for _ in range(G - 1)]
outs = [[0, 0, None] for _ in range(G)]
bar = threading.Barrier(G)
th = [threading.Thread(target=drive, args=(recs[g], wave[g * N // G : (g + 1) * N // G], outs[g], bar)) for g in range(G)]
for t in th:
    t.start()
for t in th:
    t.join()
wall = max(o[0] for o in outs)
print(f"{preset}: {G} engines x {N // G} streams on {G} host threads: {wall / outs[0][1] * 1e3:.3f} ms per tick of all {N} streams "
      f"({[round(o[0] / o[1] * 1e3, 3) for o in outs]} each)", flush=True)
same = sum(a == b for o, g in zip(outs, range(G)) for a, b in zip(o[2], one[2][g * N // G : (g + 1) * N // G]))
print(f"tokens equal to the one-engine run: {same} / {N}")
