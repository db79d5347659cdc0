"""Dev probe (GPU): tile configs on the 6.25 Hz stack's shapes (M = 2048 rows), where 128x64 tiles quantise badly over 256 CUs."""
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
shapes = [(2048, 2080, 768, 1, 0, 5), (2048, 768, 1536, 0, 1, 5), (2048, 1728, 768, 0, 0, 5), (2048, 768, 576, 0, 1, 5), (2048, 1536, 768, 0, 0, 10),
          (2048, 768, 768, 0, 1, 10), (2048, 2048, 768, 1, 0, 5), (2048, 768, 2048, 0, 1, 5), (2048, 2560, 768, 1, 0, 5), (2048, 768, 2560, 0, 1, 5),
          (4064, 1424, 512, 1, 0, 8), (4064, 512, 512, 0, 1, 16), (4064, 1024, 512, 0, 0, 16), (8096, 656, 256, 1, 0, 4), (16160, 656, 192, 1, 0, 2)]
cfgs = (5, 9, 10, 11, 3, -1)
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
print("us per batch over these shapes:", [round(t) for t in tot])
