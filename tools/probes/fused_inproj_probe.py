"""Dev probe (GPU): would one GEMM for [feed_forward1.in_proj | self_attn_weights.in_proj] (same input) beat two launches?"""
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


def t(M, N, K, act):
    ms = C.c_float()
    assert L.k2hip_debug_gemm(m.handle, M, N, K, act, 0, -1, 30, C.byref(ms)) == 0
    return ms.value * 1e3


tot_a = tot_b = 0
for M, D, F, nl in ((16160, 192, 512, 2), (8096, 256, 768, 4), (4064, 512, 1536, 8), (2048, 768, 2048, 5)):
    H = 8 if D == 768 else 4
    a, b, c = t(M, F * 3 // 4, D, 1), t(M, 68 * H, D, 0), t(M, F * 3 // 4 + 68 * H, D, 1)
    print(f"M={M} D={D}: ff1 {a:.1f} us + attn {b:.1f} us = {a + b:.1f}; merged {c:.1f} us  (x{nl} layers)")
    tot_a += (a + b) * nl
    tot_b += c * nl
print(f"per batch: separate {tot_a:.0f} us, merged {tot_b:.0f} us")
