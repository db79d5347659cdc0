"""Dev probe (GPU): sustained fp32 matrix rate on ONE large problem (8192^3: prologue / epilogue / launch are negligible) --
vendor library through torch.matmul (TF32 off) next to this repo's kernels.  Separates in-loop efficiency from per-launch overhead."""
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import k2transducerasr_amd as pkg
from k2transducerasr_amd.synth import write_synthetic_model

torch.backends.cuda.matmul.allow_tf32 = False
dev = torch.device("cuda:0")
for M, N, K in ((8192, 8192, 8192), (8192, 8192, 512), (16384, 4096, 2048)):
    a = (torch.rand(M, K, device=dev) * 2 - 1)
    w = (torch.rand(N, K, device=dev) * 2 - 1)
    for _ in range(3):
        c = a @ w.t()
    torch.cuda.synchronize()
    t = time.perf_counter()
    it = 10
    for _ in range(it):
        c = a @ w.t()
    torch.cuda.synchronize()
    us = (time.perf_counter() - t) / it * 1e6
    print(f"vendor {M:6d} {N:5d} {K:5d}  {us:9.1f} us  {2.0 * M * N * K / us / 1e6:6.1f} TF/s", flush=True)
    del a, w, c

path = "/tmp/tune_tiny.k2w"
if not os.path.exists(path):
    write_synthetic_model(path, "zipformer2-tiny-test")
m = pkg.Model(path, 0)
L = pkg.load_library()
L.k2hip_debug_gemm.argtypes = [C.c_void_p] + [C.c_int32] * 7 + [C.POINTER(C.c_float)]
cfgs = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [-1, 118, 100, 1001, 1011, 1021, 1061]
for M, N, K in ((8192, 8192, 8192), (8192, 8192, 512), (16384, 4096, 2048)):
    for cfg in cfgs:
        ms = C.c_float()
        rc = L.k2hip_debug_gemm(m.handle, M, N, K, 0, 0, cfg, 5, C.byref(ms))
        print(f"k2hip cfg {cfg:5d} {M:6d} {N:5d} {K:5d}  {ms.value * 1e3:9.1f} us  {2.0 * M * N * K / ms.value / 1e9:6.1f} TF/s  rc={rc}", flush=True)
