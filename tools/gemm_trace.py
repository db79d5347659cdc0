#!/usr/bin/env python3
"""Dev tool (GPU): in-kernel timeline of ONE ring-GEMM launch (s_memtime stamps of lane 0 of every wave).
usage: gemm_trace.py M N K act res cfg(>=100)
Stamps: 0 entry | 1 prologue DMA issued | 2 tile 0 landed + barrier + first fragments | then per K tile: after SYNC, after STEP
(the last one or two tiles have no stamps between them) | after loop | end.  [63] = s_memrealtime (100 MHz) at the end."""
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
rt = (t[:, :, 63] - t[:, :, 60]).astype(np.float64) * 10.0       # ns (s_memrealtime ticks at 100 MHz)
cyc = (t[:, :, ns - 1] - t[:, :, 0]).astype(np.float64)
print(f"s_memtime ticks per ns over a wave's lifetime: median {np.median(cyc / rt):.3f} (wave lifetime median {np.median(rt) / 1e3:.1f} us)")
st = t[:, :, :ns]
t0 = st[:, :, 0].min()
print(f"{M}x{N}x{K} cfg {cfg}: {nwg.value} workgroups x {nw.value} waves, {ns} stamps per wave")
start = st[:, :, 0] - t0
end = st[:, :, ns - 1] - t0
print(f"kernel span (first entry -> last end): {end.max()} cycles; workgroup entry spread: {start.max()} cycles")
d = np.diff(st, axis=2)                      # [wg, wave, ns-1]
life = (st[:, :, ns - 1] - st[:, :, 0])
print(f"wave lifetime: mean {life.mean():.0f}  min {life.min()}  max {life.max()} cycles")
names = ["prologue issue", "wait tile 0 + barrier + frags"]
npairs = (ns - 5) // 2
print(f"  {'prologue (entry -> DMA issued)':40s} mean {d[:, :, 0].mean():8.0f}  p95 {np.percentile(d[:, :, 0], 95):8.0f}")
print(f"  {'tile 0 landed (wait+barrier+frag reads)':40s} mean {d[:, :, 1].mean():8.0f}  p95 {np.percentile(d[:, :, 1], 95):8.0f}")
sync = d[:, :, 2 : 2 + 2 * npairs : 2]
step = d[:, :, 3 : 3 + 2 * npairs : 2]
if npairs:
    print(f"  {'SYNC (vmcnt wait + barrier), per tile':40s} mean {sync.mean():8.0f}  p50 {np.percentile(sync, 50):8.0f}  p95 {np.percentile(sync, 95):8.0f}  max {sync.max()}")
    print(f"  {'STEP (16 MFMAs + next frags + DMA issue)':40s} mean {step.mean():8.0f}  p50 {np.percentile(step, 50):8.0f}  p95 {np.percentile(step, 95):8.0f}  max {step.max()}")
    print("   per-tile means over all waves (SYNC/STEP):", " ".join(f"{sync[:, :, i].mean():.0f}/{step[:, :, i].mean():.0f}" for i in range(min(npairs, 24))))
print(f"  {'tail tiles (after last stamped pair)':40s} mean {d[:, :, ns - 3].mean():8.0f}")
print(f"  {'epilogue':40s} mean {d[:, :, ns - 2].mean():8.0f}  p95 {np.percentile(d[:, :, ns - 2], 95):8.0f}")
if npairs:
    for wg in (0, nwg.value // 2):
        print(f" workgroup {wg}: per wave, tiles 4..9 as SYNC/STEP; then absolute time of the barrier exits of tile 6 relative to wave 0")
        for w in range(nw.value):
            print(f"   wave {w}: " + " ".join(f"{sync[wg, w, i]:5d}/{step[wg, w, i]:5d}" for i in range(4, min(10, npairs))) +
                  f"   | exit6 {st[wg, w, 2 + 2 * 6 + 1] - st[wg, 0, 2 + 2 * 6 + 1]:6d}  end6 {st[wg, w, 2 + 2 * 6 + 2] - st[wg, 0, 2 + 2 * 6 + 1]:6d}")
    print("  STEP histogram (cycles):", np.histogram(step, bins=[0, 1100, 1300, 1500, 1700, 1900, 2100, 2300, 2500, 2800, 3200, 100000])[0].tolist())
    print("  SYNC histogram (cycles):", np.histogram(sync, bins=[0, 100, 200, 400, 600, 800, 1000, 1300, 1600, 2000, 100000])[0].tolist())
ideal = (K // 32) * 16 * 64
print(f"ideal MFMA cycles per wave: {ideal} (x2 waves per SIMD when one workgroup of 8 waves owns the CU: {2 * ideal})")
