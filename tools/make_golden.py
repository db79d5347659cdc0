"""Generate tests/golden/tiny_golden.npz.

Inputs are seeded synthetic audio; expected outputs come from the INDEPENDENT torch /
numpy restatement (tests/torch_twin.py), not from the C oracle, so the fixture pins the
oracle from outside.  The reference itself cannot be run (C# + ONNXRuntime),
so there are no reference-generated vectors.  Tokens come from the twin's encoder output
pushed through a literal Python transcription of OfflineRecognizer.cs:202-288.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from k2transducerasr_amd.k2w import read_k2w  # noqa: E402
from k2transducerasr_amd.synth import synth_utterance, write_synthetic_model  # noqa: E402
from torch_twin import Twin, fbank_np  # noqa: E402

LOG_FLOOR = np.float32(-23.025850929940457)


def pad_sequence(feats, tail=19):
    L = max(f.size for f in feats) + 80 * tail
    out = np.zeros((len(feats), L), np.float32)
    for i, f in enumerate(feats):
        out[i, : f.size] = f.reshape(-1)
    out[out == 0] = LOG_FLOOR
    return out


def greedy_batch(tw, enc):  # OfflineRecognizer.cs:189-303
    B, Tp, _ = enc.shape
    blank, unk = 0, 2
    dec = tw.decoder(torch.tensor([[-1, blank]] * B)).numpy()
    tokens = [None] * B
    stamps = [None] * B
    for t in range(Tp):
        logits = tw.joiner(torch.from_numpy(enc[:, t]), torch.from_numpy(dec)).numpy()
        emitted = False
        for m in range(B):
            tok = 0
            for k in range(1, logits.shape[1]):
                tok = tok if logits[m, tok] > logits[m, k] else k
            if tokens[m] is None:
                tokens[m] = [blank] * (2 * B)
                stamps[m] = [0] * (2 * B)
            if tok != blank and tok != unk:
                tokens[m].append(tok)
                stamps[m].append(t)
                emitted = True
        if emitted:
            dec = tw.decoder(torch.tensor([tk[-2:] for tk in tokens])).numpy()
    return [(tk[2 * B:], st[2 * B:]) for tk, st in zip(tokens, stamps)]


def main():
    torch.set_num_threads(4)
    path = "/tmp/golden_tiny.k2w"
    write_synthetic_model(path, "zipformer2-tiny-test")
    meta, tensors = read_k2w(path)
    tw = Twin(meta, tensors)
    utts = [synth_utterance(900 + u, s) for u, s in enumerate([1.0, 0.8, 1.0])]
    feats = [fbank_np(u, meta) for u in utts]
    x = pad_sequence(feats).reshape(len(utts), -1, 80)
    with torch.no_grad():
        enc = tw.encoder(torch.from_numpy(x)).numpy()
        y = np.array([[-1, 0], [0, 0], [5, 7], [36, 1], [3, 3], [-1, -1]], np.int64)
        dec = tw.decoder(torch.from_numpy(y)).numpy()
        logits = tw.joiner(torch.from_numpy(enc[0, :6]), torch.from_numpy(dec[:6])).numpy()
        res = greedy_batch(tw, enc)
    out = dict(samples=utts[0], fbank=feats[0], x=x, encoder_out=enc, y=y, decoder_out=dec, logits=logits,
               w_checksum=np.float64(np.asarray(tensors["joiner.output_linear.weight"], np.float64).sum()))
    for b, (tk, st) in enumerate(res):
        out[f"tok{b}"] = np.array(tk, np.int64)
        out[f"ts{b}"] = np.array(st, np.int32)
    dst = os.path.join(ROOT, "tests", "golden", "tiny_golden.npz")
    np.savez_compressed(dst, **out)
    print("wrote", dst, os.path.getsize(dst), "bytes;", [len(r[0]) for r in res], "tokens")


if __name__ == "__main__":
    main()
