"""Dev tool: choose the synthetic joiner's blank bias so that a target fraction of frames emits under GREEDY decoding.

Random weights make the logits nearly frame-independent, so the emission rate is a steep function of the bias and depends on
the decoder context that the emissions themselves create; a quantile of the blank gap under a fixed context (the first version
of this tool) missed by 3-4x.  Here the bias is bisected on the emission rate of an actual greedy loop (per stream, one symbol
per frame, skip blank / unk -- OfflineRecognizer.cs:127-179) run in numpy on the CPU oracle's encoder / decoder / joiner
outputs (test infrastructure); the constant is then frozen in k2transducerasr_amd/synth.py:BLANK_BIAS and checked with the
oracle's own batch loop.  Not part of the product.

usage: calibrate_blank_bias.py <preset> [seconds] [target emission rate]
       calibrate_blank_bias.py <preset> <seconds> <target> batch <B>   offline presets: bisect on the emission rate of the ORACLE'S OWN
                                                                       batch loop over the benchmark's B utterances (the first-emission
                                                                       context switch couples the streams of a batch); the bias is
                                                                       patched in place in the weights file between evaluations
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
if len(sys.argv) > 5 and sys.argv[4] == "batch":
    import struct
    B = int(sys.argv[5])
    path = f"/tmp/calib_{preset}_b{B}.k2w"
    write_synthetic_model(path, preset, blank_bias=0.0)
    raw = open(path, "rb").read(1 << 22)
    _, n_meta, n_t, data_off = struct.unpack_from("<IIIQ", raw, 4)
    p = 24
    for _ in range(n_meta):
        kl, vl = struct.unpack_from("<II", raw, p)
        p += 8 + kl + vl
    off0 = None
    for _ in range(n_t):
        (nl,) = struct.unpack_from("<I", raw, p)
        p += 4
        name = raw[p : p + nl].decode()
        p += nl
        off = struct.unpack_from("<II4QQQ", raw, p)[6]
        p += 56
        if name == "joiner.output_linear.bias":
            off0 = data_off + off
    with open(path, "rb") as f:
        f.seek(off0)
        base = struct.unpack("<f", f.read(4))[0]
    o = Oracle(path)
    feats = [o.fbank(synth_utterance(u, secs)) for u in range(B)]
    enc = o.encoder(o.pad_sequence(feats).reshape(B, -1, 80))
    o.close()

    def rate(bias):
        with open(path, "r+b") as f:
            f.seek(off0)
            f.write(struct.pack("<f", base + bias))
        oo = Oracle(path)
        res = oo.greedy_batch(enc)
        oo.close()
        return sum(len(r[0]) for r in res) / (B * enc.shape[1])

    lo, hi = 0.0, 12.0
    for _ in range(16):
        mid = 0.5 * (lo + hi)
        if rate(mid) > target:
            lo = mid
        else:
            hi = mid
    b = round(0.5 * (lo + hi), 3)
    print(f"{preset}, batch of {B} x {secs:g} s: blank_bias {b} -> emission {rate(b):.3f} (target {target}); +-0.02: {rate(b - 0.02):.3f} / {rate(b + 0.02):.3f}")
    sys.exit(0)
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
