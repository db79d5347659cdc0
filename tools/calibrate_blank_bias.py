"""Dev tool: choose the synthetic joiner's blank bias so that a target fraction of frames emits under GREEDY decoding.

Random weights make the logits nearly frame-independent, so the emission rate is a steep function of the bias and depends on
the decoder context that the emissions themselves create; a quantile of the blank gap under a fixed context (the first version
of this tool) missed by 3-4x.  Here the bias is bisected on the emission rate of an actual greedy loop (per stream, one symbol
per frame, skip blank / unk -- OfflineRecognizer.cs:127-179) run in numpy on the CPU oracle's encoder / decoder / joiner
outputs (test infrastructure); the constant is then frozen in k2transducerasr_amd/synth.py:BLANK_BIAS and checked with the
oracle's own batch loop.  Not part of the product.

usage: calibrate_blank_bias.py <preset> [seconds] [target emission rate] [streaming: 0|1]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from k2transducerasr_amd.synth import synth_utterance, write_synthetic_model  # noqa: E402
from oracle import Oracle  # noqa: E402

preset = sys.argv[1]
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 10.0
target = float(sys.argv[3]) if len(sys.argv) > 3 else 0.2
NUTT = 4
path = f"/tmp/calib_{preset}.k2w"
meta = write_synthetic_model(path, preset, blank_bias=0.0)
streaming = meta.get("streaming", "0") not in ("0", "", "False")
if streaming:
    from oracle.online import OnlineOracle
    o = OnlineOracle(path)
    encs = []
    for u in range(NUTT):
        f = o.fbank(synth_utterance(1000 + u, secs))
        s = o.create_stream()
        T, S = o.chunk_length, o.shift_length
        outs = [o.encoder_chunk(s, f[k * S : k * S + T]) for k in range((f.shape[0] - T) // S + 1)]
        encs.append(np.concatenate(outs))
    init_ctx, skip = [0, 0], (0, 1, 2)          # OnlineStream.cs:44, OnlineRecognizer.cs:181
else:
    o = Oracle(path)
    feats = [o.fbank(synth_utterance(u, secs)) for u in range(NUTT)]
    xp = o.pad_sequence(feats)
    encs = list(o.encoder(xp.reshape(NUTT, -1, 80)))
    init_ctx, skip = [-1, 0], (0, 2)            # OfflineRecognizer.cs:105,161
dec_cache = {}


def dec(ctx):
    k = tuple(ctx)
    if k not in dec_cache:
        dec_cache[k] = o.decoder(np.array([ctx], np.int64))
    return dec_cache[k]


def emission(bias):
    emitted = frames = 0
    for e in encs:
        ctx = list(init_ctx)
        t = 0
        while t < e.shape[0]:
            # frames up to the next emission share one context: evaluate them in one joiner call
            l = o.joiner(e[t:], np.repeat(dec(ctx), e.shape[0] - t, 0))
            l[:, 0] += bias
            y = l.shape[1] - 1 - np.argmax(l[:, ::-1], axis=1)     # later index wins ties
            hit = np.nonzero(~np.isin(y, skip))[0]
            if hit.size == 0:
                break
            t += int(hit[0])
            ctx = [ctx[1], int(y[hit[0]])]
            emitted += 1
            t += 1
        frames += e.shape[0]
    return emitted / frames


lo, hi = 0.0, 12.0
for _ in range(28):
    mid = 0.5 * (lo + hi)
    if emission(mid) > target:
        lo = mid
    else:
        hi = mid
bias = round(0.5 * (lo + hi), 3)
print(f"{preset}: blank_bias {bias} -> emission {emission(bias):.3f} (target {target}); +-0.05: {emission(bias - 0.05):.3f} / {emission(bias + 0.05):.3f}")
if not streaming:
    write_synthetic_model(path, preset, blank_bias=bias)
    o2 = Oracle(path)
    res = o2.greedy_batch(o2.encoder(xp.reshape(NUTT, -1, 80)))
    print("oracle batch loop emitted", [len(r[0]) for r in res], "of", encs[0].shape[0], "frames each")
