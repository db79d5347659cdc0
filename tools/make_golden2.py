"""Generate tests/golden/{conformer,beam,streaming}_tiny_golden.npz.

Like tools/make_golden.py: expected outputs come from the INDEPENDENT restatements -- tests/torch_twin_conformer.py,
tests/torch_twin_online.py and a literal transcription of icefall's modified_beam_search on torch ops -- never from the C
oracle, so the fixtures pin the oracle (and through it the HIP path) from outside.  No reference-generated vectors exist
(the reference is C# + ONNXRuntime and cannot run here).
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from k2transducerasr_amd.k2w import read_k2w  # noqa: E402
from k2transducerasr_amd.synth import synth_utterance, write_synthetic_model  # noqa: E402
from make_golden import pad_sequence  # noqa: E402
from torch_twin import Twin, fbank_np  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
LOG_FLOOR = np.float32(-23.025850929940457)


def checksum(t, name):
    return np.float64(np.asarray(t[name], np.float64).sum())


def conformer():
    from torch_twin_conformer import ConformerTwin
    path = "/tmp/golden_conformer.k2w"
    write_synthetic_model(path, "conformer-tiny-test")
    meta, tensors = read_k2w(path)
    tw = ConformerTwin(meta, tensors)
    utts = [synth_utterance(910 + u, s) for u, s in enumerate([0.9, 0.7])]
    feats = [fbank_np(u, meta) for u in utts]
    x = pad_sequence(feats).reshape(len(utts), -1, 80)
    enc = tw.forward(x)
    l1 = tw.forward(x, 1)
    y = np.array([[-1, 0], [0, 0], [5, 7], [40, 1], [-1, -1]], np.int64)
    np.savez_compressed(os.path.join(GOLD, "conformer_tiny_golden.npz"), x=x, encoder_out=enc, layer0_out=l1, y=y,
                        decoder_out=tw.decoder(y), w_checksum=checksum(tensors, "joiner.encoder_proj.weight"))
    print("conformer golden: enc", enc.shape, float(np.abs(enc).max()))


def beam():
    """icefall modified_beam_search (beam 4) on the zipformer2-tiny twin's encoder output, decoder / joiner from the twin"""
    path = "/tmp/golden_tiny.k2w"
    write_synthetic_model(path, "zipformer2-tiny-test")
    meta, tensors = read_k2w(path)
    tw = Twin(meta, tensors)
    utts = [synth_utterance(920 + u, s) for u, s in enumerate([1.0, 0.8])]
    feats = [fbank_np(u, meta) for u in utts]
    x = pad_sequence(feats).reshape(len(utts), -1, 80)
    out = dict(x=x, w_checksum=checksum(tensors, "joiner.output_linear.weight"))
    with torch.no_grad():
        enc = tw.encoder(torch.from_numpy(x)).numpy()
        out["encoder_out"] = enc
        for b in range(enc.shape[0]):
            B = {(0, 0): dict(ys=[0, 0], lp=torch.zeros(1), ts=[])}
            for t in range(enc.shape[1]):
                A = list(B.values())
                B = {}
                dec = tw.decoder(torch.tensor([h["ys"][-2:] for h in A]))
                logits = tw.joiner(torch.from_numpy(enc[b, t]).expand(len(A), -1), dec)
                lp = logits.log_softmax(-1) + torch.cat([h["lp"].reshape(1, 1) for h in A])
                V = lp.size(-1)
                vals, idx = lp.reshape(-1).topk(min(4, lp.numel()))
                for v, i in zip(vals, idx.tolist()):
                    h = A[i // V]
                    tok = i % V
                    ys, ts = h["ys"][:], h["ts"][:]
                    if tok not in (0, 2):
                        ys.append(tok)
                        ts.append(t)
                    key = tuple(ys)
                    if key in B:
                        B[key]["lp"] = torch.logaddexp(B[key]["lp"], v.reshape(1))
                    else:
                        B[key] = dict(ys=ys, lp=v.reshape(1), ts=ts)
            best = max(B.values(), key=lambda h: h["lp"] / len(h["ys"]))
            out[f"tok{b}"] = np.array(best["ys"][2:], np.int64)
            out[f"ts{b}"] = np.array(best["ts"], np.int32)
            out[f"score{b}"] = np.float32(best["lp"].item())
            print("beam golden stream", b, best["ys"][2:])
    np.savez_compressed(os.path.join(GOLD, "beam_tiny_golden.npz"), **out)


def streaming():
    from torch_twin_online import OnlineTwin
    path = "/tmp/golden_stream.k2w"
    write_synthetic_model(path, "zipformer2-streaming-tiny-test")
    meta, tensors = read_k2w(path)
    tw = OnlineTwin(meta, tensors)
    f = fbank_np(synth_utterance(930, 1.6), meta)
    T, shift = int(meta["T"]), int(meta["decode_chunk_len"])
    st = tw.init_states(1)
    outs, pos = [], 0
    with torch.no_grad():
        while pos + T <= f.shape[0]:
            x = f[pos : pos + T].copy()
            x[x == 0] = LOG_FLOOR
            o, st = tw.encoder_chunk(torch.from_numpy(x[None]), st)
            outs.append(o[0].numpy())
            pos += shift
    np.savez_compressed(os.path.join(GOLD, "streaming_tiny_golden.npz"), feats=f, chunk_out=np.stack(outs),
                        cached_key0=st[0][:, 0].numpy(), embed_state=st[-2][0].numpy(), processed_len=np.int64(st[-1][0].item()),
                        w_checksum=checksum(tensors, "joiner.encoder_proj.weight"))
    print("streaming golden:", len(outs), "chunks")


if __name__ == "__main__":
    torch.set_num_threads(4)
    conformer()
    beam()
    streaming()
