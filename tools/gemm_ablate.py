#!/usr/bin/env python3
"""Dev tool (GPU): ablations of the ring GEMM on one shape: full | no in-loop DMA | no MFMA | neither.
usage: gemm_ablate.py M N K ringidx[,ringidx...]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import k2transducerasr_amd as pkg
from k2transducerasr_amd.synth import write_synthetic_model

M, N, K = [int(x) for x in sys.argv[1:4]]
idxs = [int(x) for x in sys.argv[4].split(",")]
path = "/tmp/tune_tiny.k2w"
if not os.path.exists(path):
    write_synthetic_model(path, "zipformer2-tiny-test")
m = pkg.Model(path, 0)
L = pkg.load_library()
L.k2hip_debug_gemm.argtypes = [C.c_void_p] + [C.c_int32] * 7 + [C.POINTER(C.c_float)]
fl = 2.0 * M * N * K
for idx in idxs:
    out = []
    for ab, name in ((0, "full"), (1, "no in-loop DMA"), (2, "no MFMA"), (3, "neither"), (4, "no stores")):
        ms = C.c_float()
        rc = L.k2hip_debug_gemm(m.handle, M, N, K, 0, 0, 100 + idx + 256 * ab, 30, C.byref(ms))
        assert rc == 0, L.k2hip_last_error()
        out.append(f"{name}: {ms.value * 1e3:6.1f} us ({fl / (ms.value * 1e-3) / 1e12:5.1f} TF)")
    print(f"{M}x{N}x{K} ring {idx}: " + " | ".join(out), flush=True)
