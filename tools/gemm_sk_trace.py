#!/usr/bin/env python3
"""Dev tool (GPU): in-kernel timeline of ONE stream-K GEMM launch (s_memtime stamps of lane 0 of every wave).
usage: gemm_sk_trace.py M N K act res cfg(>=1000)
Per segment of a workgroup: start | pipeline primed (stores drained, barrier, first K step landed) | K loop done | epilogue/publish done."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import k2transducerasr_amd as pkg
from k2transducerasr_amd.synth import write_synthetic_model

M, N, K, act, res, cfg = [int(x) for x in sys.argv[1:7]]
path = "/tmp/tune_tiny.k2w"
if not os.path.exists(path):
    write_synthetic_model(path, "zipformer2-tiny-test")
m = pkg.Model(path, 0)
L = pkg.load_library()
cap = 1 << 22
buf = np.zeros(cap, np.uint64)
nwg, nw = C.c_int32(), C.c_int32()
L.k2hip_debug_gemm_trace.argtypes = [C.c_void_p] + [C.c_int32] * 6 + [C.c_void_p, C.c_int64, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
rc = L.k2hip_debug_gemm_trace(m.handle, M, N, K, act, res, cfg, buf.ctypes.data_as(C.c_void_p), cap, C.byref(nwg), C.byref(nw))
assert rc == 0, L.k2hip_last_error()
t = buf[: nwg.value * nw.value * 64].reshape(nwg.value, nw.value, 64).astype(np.int64)
ns = t[:, :, 62]
rt = (t[:, :, 63] - t[:, :, 60]).astype(np.float64) * 10.0
life = np.array([[t[g, w, ns[g, w] - 1] - t[g, w, 0] for w in range(nw.value)] for g in range(nwg.value)], np.float64)
print(f"{M}x{N}x{K} cfg {cfg}: {nwg.value} workgroups x {nw.value} waves; s_memtime ticks per ns: {np.median(life / rt):.3f}; wave lifetime median {np.median(rt) / 1e3:.1f} us")
t0 = t[:, :, 0].min()
print(f"kernel span {(t[:, :, :60].max() - t0)} ticks; entry spread {(t[:, :, 0].max() - t0)}")
segs = (ns // 4).max()
for s in range(min(segs, 8)):
    ok = ns >= 4 * (s + 1)
    a = t[:, :, 4 * s][ok]; b = t[:, :, 4 * s + 1][ok]; c = t[:, :, 4 * s + 2][ok]; d = t[:, :, 4 * s + 3][ok]
    print(f" segment {s}: waves {ok.sum():5d} | start at {np.mean(a - t0):9.0f} | prime {np.mean(b - a):7.0f} (p95 {np.percentile(b - a, 95):7.0f}) | K loop {np.mean(c - b):8.0f} (p95 {np.percentile(c - b, 95):8.0f}) | "
          f"epilogue {np.mean(d - c):7.0f} (p95 {np.percentile(d - c, 95):7.0f})")
nk = K // 32
print(f"(a full tile's K loop is {nk} steps; ideal MFMA ticks per step and SIMD: see gemm_trace.py -- s_memtime ticks at 100 MHz x ratio above)")
