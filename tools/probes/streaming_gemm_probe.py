"""Dev probe (GPU): the streaming chunk step's GEMM shapes (128 streams) under the automatic choice (mostly the 16-row small-problem
kernel) against forced 64x64 tiles (register-staged cfg 3, LDS-DMA cfg 9 / 10)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import k2transducerasr_amd as pkg
from k2transducerasr_amd.synth import write_synthetic_model

path = "/tmp/tune_tiny.k2w"
write_synthetic_model(path, "zipformer2-tiny-test")
m = pkg.Model(path, 0)
L = pkg.load_library()
L.k2hip_debug_gemm.argtypes = [C.c_void_p] + [C.c_int32] * 7 + [C.POINTER(C.c_float)]
shapes = []
for M, D, F, nl in ((2048, 192, 512, 2), (1024, 256, 768, 4), (512, 384, 1024, 6), (256, 512, 1536, 4)):
    H = 8 if D == 512 else 4
    for (N, K, act, res) in ((68 * H, D, 0, 0), (F * 3 // 4, D, 1, 0), (D, F * 3 // 4, 0, 1), (F, D, 1, 0), (D, F, 0, 1), (F * 5 // 4, D, 1, 0), (D, F * 5 // 4, 0, 1),
                             (9 * D // 4, D, 0, 0), (D, 3 * D // 4, 0, 1), (2 * D, D, 0, 0), (D, D, 0, 1)):
        shapes.append((M, N, K, act, res, nl * (2 if (N, K) in ((2 * D, D), (D, D)) else 1)))
cfgs = (-1, 3, 9, 10)
print(f"{'M':>6} {'N':>5} {'K':>5} a r  n |" + "".join(f" cfg{c:>2} us |" for c in cfgs))
tot = [0.0] * len(cfgs)
for (M, N, K, act, res, n) in shapes:
    line = f"{M:6d} {N:5d} {K:5d} {act} {res} {n:2d} |"
    for ci, cfg in enumerate(cfgs):
        ms = C.c_float()
        rc = L.k2hip_debug_gemm(m.handle, M, N, K, act, res, cfg, 30, C.byref(ms))
        assert rc == 0, L.k2hip_last_error()
        tot[ci] += ms.value * 1e3 * n
        line += f" {ms.value * 1e3:8.1f} |"
    print(line, flush=True)
print("us per chunk step over these shapes:", [round(t) for t in tot])
