"""Dev probe (GPU): tile configs on the conformer-zh shard's GEMM shapes (M = 8 x 753 = 6024 rows)."""
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
shapes = [(6024, 2048, 512, 6, 0, 24), (6024, 512, 2048, 0, 1, 24), (6024, 1536, 512, 0, 0, 12), (6024, 512, 512, 0, 1, 24), (6024, 1024, 512, 0, 0, 12)]
cfgs = (5, 9, 10, 11, -1)
print(f"{'M':>6} {'N':>5} {'K':>5} a r  n |" + "".join(f" cfg{c:>2} us |" for c in cfgs))
tot = [0.0] * len(cfgs)
for (M, N, K, act, res, n) in shapes:
    line = f"{M:6d} {N:5d} {K:5d} {act} {res} {n:2d} |"
    for ci, cfg in enumerate(cfgs):
        ms = C.c_float()
        assert L.k2hip_debug_gemm(m.handle, M, N, K, act, res, cfg, 30, C.byref(ms)) == 0
        tot[ci] += ms.value * 1e3 * n
        line += f" {ms.value * 1e3:8.1f} |"
    print(line, flush=True)
print("us per batch:", [round(t) for t in tot])
