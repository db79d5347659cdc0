"""Dev tool: choose the synthetic joiner's blank bias so that ~25 % of frames emit.

Uses the CPU oracle (test infrastructure) to look at the logits a seeded
random-weight model produces on seeded synthetic audio; the resulting constant
is then frozen in k2transducerasr_amd/synth.py:BLANK_BIAS.  Not part of the product.
"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from k2transducerasr_amd.synth import write_synthetic_model, synth_utterance
from oracle import Oracle

preset = sys.argv[1]
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 6.0
target = float(sys.argv[3]) if len(sys.argv) > 3 else 0.25
path = f"/tmp/calib_{preset}.k2w"
write_synthetic_model(path, preset, blank_bias=0.0)
o = Oracle(path)
feats = [o.fbank(synth_utterance(u, secs)) for u in range(4)]
xp = o.pad_sequence(feats)
T = xp.shape[1] // 80
e = o.encoder(xp.reshape(4, T, 80))
gaps = []
for ctx in ([-1, 0], [0, 0], [5, 9]):
    d = o.decoder(np.array([ctx], np.int64))
    for b in range(4):
        l = o.joiner(e[b], np.repeat(d, e.shape[1], 0))
        gaps.append(l[:, 1:].max(1) - l[:, 0])
gaps = np.concatenate(gaps)
bias = float(np.quantile(gaps, 1.0 - target))
print(preset, "gap quantiles", np.quantile(gaps, [0.05, 0.25, 0.5, 0.75, 0.95]), "-> blank_bias", round(bias, 3))
write_synthetic_model(path, preset, blank_bias=round(bias, 3))
o = Oracle(path)
res, mg = o.greedy_batch(o.encoder(xp.reshape(4, T, 80)), want_margins=True)
print("emitted", [len(r[0]) for r in res], "of", e.shape[1], "margin q", np.quantile(mg, [0, 0.01, 0.1, 0.5]))
print([r[0][:12] for r in res])
