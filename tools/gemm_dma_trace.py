#!/usr/bin/env python3
"""Dev tool (GPU): in-kernel timeline of ONE launch of the LDS-DMA GEMM kernel (s_memtime stamps of lane 0 of every wave).
usage: gemm_dma_trace.py M N K act res cfg   (cfg: 5 = 128x64, 0 = 128x128, 9 = 64x64, 11 = 64x96; 2000+i = pipelined kernel table)
Stamps: entry | prologue issued (+ residual loads) | barrier of each K step passed (first 40) | loop done | end."""
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
ns = int(t[0, 0, 62])
rt = (t[:, :, 63] - t[:, :, 60]).astype(np.float64) * 10.0
st = t[:, :, :ns]
life = (st[:, :, -1] - st[:, :, 0]).astype(np.float64)
t0 = st[:, :, 0].min()
print(f"{M}x{N}x{K} cfg {cfg}: {nwg.value} workgroups x {nw.value} waves, {ns} stamps; ticks per ns {np.median(life / rt):.3f}; wave lifetime median {np.median(rt) / 1e3:.2f} us")
print(f"wave lifetime (ticks): mean {life.mean():.0f} min {life.min():.0f} max {life.max():.0f}")
# s_memtime counts per XCD (the bases differ), s_memrealtime (stamps 60 / 63: entry / exit, 10 ns) is chip-wide: launch geometry from it
e_rt, x_rt = t[:, :, 60].astype(np.float64) * 10.0, t[:, :, 63].astype(np.float64) * 10.0
z = e_rt.min()
print("chip-wide (s_memrealtime, us): wave entry p0/p50/p90/p100 " + " ".join(f"{np.percentile(e_rt - z, q) / 1e3:.2f}" for q in (0, 50, 90, 100)) +
      " | wave exit p0/p10/p50/p100 " + " ".join(f"{np.percentile(x_rt - z, q) / 1e3:.2f}" for q in (0, 10, 50, 100)))
d = np.diff(st, axis=2)
nk = min(K // 32, 40)
print(f"  prologue (entry -> DMA + residual issued)  mean {d[:, :, 0].mean():7.0f}  p95 {np.percentile(d[:, :, 0], 95):7.0f}")
print(f"  first K step landed (wait + barrier)       mean {d[:, :, 1].mean():7.0f}  p95 {np.percentile(d[:, :, 1], 95):7.0f}")
steps = d[:, :, 2 : 1 + nk]
if steps.size:
    print(f"  K step (barrier to barrier), steps 1..{nk - 1}   mean {steps.mean():7.0f}  p50 {np.percentile(steps, 50):7.0f}  p95 {np.percentile(steps, 95):7.0f}")
    print("   per-step means:", " ".join(f"{steps[:, :, i].mean():.0f}" for i in range(steps.shape[2])))
print(f"  last step(s) + loop exit                     mean {d[:, :, ns - 3].mean():7.0f}")
print(f"  epilogue                                     mean {d[:, :, ns - 2].mean():7.0f}  p95 {np.percentile(d[:, :, ns - 2], 95):7.0f}")
mt = {0: 2, 5: 1, 7: 1, 9: 1, 10: 1, 11: 1}.get(cfg, 0)
print(f"ideal MFMA ticks per K step and wave: {16 * mt * 64} (x waves per SIMD sharing the pipe)")
