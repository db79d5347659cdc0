"""Dev tool (GPU): run ONE GEMM shape/config a few times (for rocprofv3 --pmc runs).
python tools/gemm_one.py M N K act res cfg [iters]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import k2transducerasr_amd as pkg
from k2transducerasr_amd.synth import write_synthetic_model

M, N, K, act, res, cfg = [int(x) for x in sys.argv[1:7]]
iters = int(sys.argv[7]) if len(sys.argv) > 7 else 5
path = "/tmp/tune_tiny.k2w"
if not os.path.exists(path):
    write_synthetic_model(path, "zipformer2-tiny-test")
m = pkg.Model(path, 0)
L = pkg.load_library()
L.k2hip_debug_gemm.argtypes = [C.c_void_p] + [C.c_int32] * 7 + [C.POINTER(C.c_float)]
ms = C.c_float()
assert L.k2hip_debug_gemm(m.handle, M, N, K, act, res, cfg, iters, C.byref(ms)) == 0
print(M, N, K, "cfg", cfg, "us", ms.value * 1e3, "TF/s", 2.0 * M * N * K / (ms.value * 1e-3) / 1e12)
