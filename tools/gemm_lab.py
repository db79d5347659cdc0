#!/usr/bin/env python3
"""Dev tool (GPU): time and CHECK tile configurations of the fp32 MFMA GEMM on the benchmark's shapes.

usage: gemm_lab.py offline|streaming|all [cfg,cfg,...]   (cfg: -1 auto, 0..11 classic / LDS-DMA table, 100+i = ring table entry i)
Each cell: us per launch (max |diff| vs the register-staged kernel when it is not ~1e-6-small)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import k2transducerasr_amd as pkg
from k2transducerasr_amd.synth import write_synthetic_model

path = "/tmp/tune_tiny.k2w"
if not os.path.exists(path):
    write_synthetic_model(path, "zipformer2-tiny-test")
m = pkg.Model(path, 0)
L = pkg.load_library()
L.k2hip_debug_gemm_check.argtypes = [C.c_void_p] + [C.c_int32] * 7 + [C.POINTER(C.c_float), C.POINTER(C.c_float)]

import re
_src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "k2transducerasr_amd", "csrc", "gemm.hip")).read()
_tab = _src[_src.index("#define K2_RING_TABLE(X)") : _src.index("#define X(i, bm, bn, ks, nst, lw, pf) {bm")]
RING = [tuple(int(v) for v in m[1:]) for m in sorted((tuple(int(x) for x in t) for t in re.findall(r"X\((\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (\d+)\)", _tab)))]


def layer_shapes(M, D, F, H):
    # (M, N, K, act, res, count per layer)
    return [(M, 68 * H, D, 0, 0, 1), (M, F * 3 // 4, D, 1, 0, 1), (M, D, F * 3 // 4, 0, 1, 1), (M, F, D, 1, 0, 1), (M, D, F, 0, 1, 1),
            (M, F * 5 // 4, D, 1, 0, 1), (M, D, F * 5 // 4, 0, 1, 1), (M, 9 * D // 4, D, 0, 0, 1), (M, D, 3 * D // 4, 0, 1, 1),
            (M, 12 * H, D, 0, 0, 2), (M, 2 * D, D, 0, 0, 2), (M, D, D, 0, 1, 2)]


# the plain-GEMM launches of one B = 32 x 10 s batch of zipformer2-large-en, from an instrumented pass (tools/gemm_profile.py):
# (M, N, K, act, res), launches per batch
OFFLINE_REAL = [((4064, 1024, 512, 0, 0), 16), ((4064, 1920, 512, 1, 0), 8), ((4064, 512, 1920, 0, 1), 8), ((2048, 1536, 768, 0, 0), 10),
                ((4064, 1536, 512, 1, 0), 8), ((4064, 1424, 512, 1, 0), 8), ((4064, 512, 1536, 0, 1), 8), ((4064, 1152, 512, 0, 0), 8),
                ((2048, 768, 2560, 0, 1), 5), ((4064, 512, 512, 0, 1), 16), ((2048, 2560, 768, 1, 0), 5), ((2048, 768, 2048, 0, 1), 5),
                ((2048, 2080, 768, 1, 0), 5), ((4064, 512, 1152, 0, 1), 8), ((307040, 384, 128, 1, 0), 1), ((2048, 768, 768, 0, 1), 10),
                ((2048, 2048, 768, 1, 0), 5), ((2048, 1728, 768, 0, 0), 5), ((2048, 768, 1536, 0, 1), 5), ((307040, 128, 384, 0, 1), 1),
                ((8096, 512, 256, 0, 0), 8), ((2048, 96, 768, 0, 0), 10), ((8096, 960, 256, 1, 0), 4), ((8096, 848, 256, 1, 0), 4),
                ((4064, 48, 512, 0, 0), 16), ((4064, 512, 384, 0, 1), 8), ((8096, 256, 960, 0, 1), 4), ((16160, 192, 2432, 0, 0), 1),
                ((8096, 256, 256, 0, 1), 8), ((8096, 256, 768, 0, 1), 4), ((8096, 768, 256, 1, 0), 4), ((8096, 576, 256, 0, 0), 4),
                ((2048, 768, 576, 0, 1), 5), ((16160, 384, 192, 0, 0), 4), ((16160, 656, 192, 1, 0), 2), ((8096, 256, 576, 0, 1), 4),
                ((16160, 192, 640, 0, 1), 2), ((16160, 640, 192, 1, 0), 2), ((16160, 192, 512, 0, 1), 2), ((16160, 192, 192, 0, 1), 4),
                ((16160, 512, 192, 1, 0), 2), ((8096, 48, 256, 0, 0), 8), ((16160, 432, 192, 0, 0), 2), ((16160, 192, 384, 0, 1), 2),
                ((8096, 256, 192, 0, 1), 4), ((8096, 512, 768, 0, 0), 1), ((16160, 48, 192, 0, 0), 4), ((8096, 500, 512, 0, 0), 1)]


def streaming_real():
    """plain-GEMM launches of one 128-stream chunk step of zipformer2-streaming-zh, from an instrumented step"""
    import numpy as np
    from collections import Counter
    from k2transducerasr_amd.synth import synth_utterance
    wpath = "/tmp/k2hip_bench_zipformer2-streaming-zh.k2w"
    if not os.path.exists(wpath):
        write_synthetic_model(wpath, "zipformer2-streaming-zh")
    rec = pkg.OnlineRecognizer(wpath)
    ss = [rec.create_online_stream() for _ in range(128)]
    wave = np.stack([synth_utterance(1000 + u, 1.0) for u in range(128)])
    need = (rec.chunk_length - 1) * 160 + 400
    rec.add_samples_batch(ss, wave[:, :need])
    rec.get_results(ss)
    rec.add_samples_batch(ss, wave[:, need : need + 32 * 160])
    rec.model.set_instrument(True)
    rec.get_results(ss)
    rows = rec.model.gemm_profile()
    rec.model.set_instrument(False)
    c = Counter()
    for r in rows:
        Mr, Nr, Kr, bat, act, res, kind = (int(x) for x in r[:7])
        if bat == 1 and (kind & 3) == 0 and Kr % 32 == 0:
            c[(Mr, Nr, Kr, act if act in (0, 1, 2) else 0, res)] += 1
    for s_ in ss:
        s_.close()
    return sorted(c.items(), key=lambda kv: -kv[1] * kv[0][0] * kv[0][1] * kv[0][2])


which = sys.argv[1] if len(sys.argv) > 1 else "offline"
shapes = []
if which == "offline":
    shapes = list(OFFLINE_REAL)
if which == "streaming-real":
    shapes = streaming_real()
if which in ("offline-synth", "all"):
    for M, D, F, H, nl in ((16160, 192, 512, 4, 2), (8096, 256, 768, 4, 4), (4064, 512, 1536, 4, 8), (2048, 768, 2048, 8, 5)):
        shapes += [(s[:5], s[5] * nl) for s in layer_shapes(M, D, F, H)]
    shapes += [((307040, 384, 128, 1, 0), 1), ((307040, 128, 384, 0, 1), 1), ((16160, 192, 2432, 0, 0), 1), ((8096, 512, 768, 0, 0), 1)]
if which in ("streaming", "all"):
    for M, D, F, H, nl in ((2048, 192, 512, 4, 2), (1024, 256, 768, 4, 4), (512, 384, 1024, 4, 6), (256, 512, 1536, 8, 4)):
        shapes += [(s[:5], s[5] * nl) for s in layer_shapes(M, D, F, H)]
cfgs = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [-1, 100, 101, 103, 104]
if len(sys.argv) > 3:  # "MxN,MxN": only these output shapes
    keep = {tuple(int(v) for v in t.split("x")) for t in sys.argv[3].split(",")}
    shapes = [s for s in shapes if (s[0][0], s[0][1]) in keep]




_tab3 = _src[_src.index("#define K2_PIPE_TABLE(X)") : _src.index("const PipeCfg kPipe[]")]
PIPE = [tuple(int(v) for v in t[1:]) for t in sorted((tuple(int(x) for x in t) for t in re.findall(r"X\((\d+), (\d+), (\d+), (\d+), (\d+), (\d+)\)", _tab3)))]


P16 = [(64, 96, 32, 48), (128, 96, 64, 48), (64, 192, 32, 96), (32, 96, 16, 48)]   # gemm.hip launch_p16_idx


def name(c):
    if c >= 3000:
        return "q%dx%d.%dx%d" % P16[c - 3000]
    if c >= 2000:
        return "p%dx%d.%dx%d.%d" % PIPE[c - 2000]
    return "auto" if c < 0 else f"c{c}" if c < 100 else "r%dx%d.%d.%d.%d%s" % (RING[c - 100][:5] + ("p" if RING[c - 100][5] else "",))


print(f"{'M':>7} {'N':>5} {'K':>5} a r  n |" + "".join(f"{name(c):>18}" for c in cfgs) + " | best")
tot = [0.0] * len(cfgs)
best_tot = 0.0
for (M, N, K, act, res), cnt in shapes:
    line = f"{M:7d} {N:5d} {K:5d} {act} {res} {cnt:2d} |"
    row = []
    for ci, cfg in enumerate(cfgs):
        ms, err = C.c_float(), C.c_float()
        ks = RING[cfg - 100][2] if 100 <= cfg < 1000 else 1
        if 100 <= cfg < 1000 and (K % (32 * ks) or K < 32 * ks):
            row.append(None)
            line += f"{'-':>18}"
            continue
        rc = L.k2hip_debug_gemm_check(m.handle, M, N, K, act, res, cfg, 20, C.byref(ms), C.byref(err))
        if rc != 0:
            row.append(None)
            line += f"{'ERR':>18}"
            continue
        us = ms.value * 1e3
        row.append(us)
        tot[ci] += us * cnt
        bad = err.value > 2e-3 * max(1.0, K ** 0.5 / 8)
        line += f"{us:13.1f}{'!%.0e' % err.value if bad else '     '}"
    ok = [u for u in row if u is not None]
    b = min(ok) if ok else 0
    best_tot += b * cnt
    line += f" | {name(cfgs[row.index(b)]) if ok else '-'} {2.0 * M * N * K / (b * 1e-6) / 1e12 if b else 0:5.1f} TF"
    print(line, flush=True)
print("weighted total ms:", " ".join(f"{name(c)}={t / 1e3:.3f}" for c, t in zip(cfgs, tot)), f"| best-of={best_tot / 1e3:.3f}")
