"""Dev tool (GPU): time the fp32 MFMA GEMM on the benchmark's shapes, per tile config.
Usage on the GPU box:  python tools/gemm_tune.py > gpurun_out/gemm_tune.txt"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import k2transducerasr_amd as pkg
from k2transducerasr_amd.synth import write_synthetic_model

path = "/tmp/tune_tiny.k2w"
write_synthetic_model(path, "zipformer2-tiny-test")
m = pkg.Model(path, 0)
L = pkg.load_library()
L.k2hip_debug_gemm.argtypes = [C.c_void_p] + [C.c_int32] * 7 + [C.POINTER(C.c_float)]

# (M, N, K, act, res) of the large model at B=32 x 10 s
shapes = []
for M, D, F in ((16160, 192, 512), (8096, 256, 768), (4064, 512, 1536), (2048, 768, 2048)):
    H = 8 if D == 768 else 4
    shapes += [(M, 68 * H, D, 0, 0), (M, F * 3 // 4, D, 1, 0), (M, D, F * 3 // 4, 0, 1), (M, F, D, 1, 0), (M, D, F, 0, 1),
               (M, F * 5 // 4, D, 1, 0), (M, D, F * 5 // 4, 0, 1), (M, 9 * D // 4, D, 0, 0), (M, D, 3 * D // 4, 0, 1),
               (M, 12 * H, D, 0, 0), (M, D, 12 * H, 0, 1), (M, 2 * D, D, 0, 0), (M, D, D, 0, 1)]
shapes += [(307040, 384, 128, 1, 0), (307040, 128, 384, 0, 1), (16160, 192, 2432, 0, 0), (8096, 500, 512, 0, 0), (8096, 512, 768, 0, 0)]
if len(sys.argv) > 1 and sys.argv[1] == "streaming":
    # streaming-zh at 128 streams: rows = 128 x chunk frames of the stack
    shapes = []
    for M, D, F in ((2048, 192, 512), (1024, 256, 768), (512, 384, 1024), (256, 512, 1536)):
        H = 8 if D == 512 else 4
        shapes += [(M, 68 * H, D, 0, 0), (M, F * 3 // 4, D, 1, 0), (M, D, F * 3 // 4, 0, 1), (M, F, D, 1, 0), (M, D, F, 0, 1),
                   (M, F * 5 // 4, D, 1, 0), (M, D, F * 5 // 4, 0, 1), (M, 9 * D // 4, D, 0, 0), (M, D, 3 * D // 4, 0, 1),
                   (M, 12 * H, D, 0, 0), (M, D, 12 * H, 0, 1), (M, 2 * D, D, 0, 0), (M, D, D, 0, 1)]
elif len(sys.argv) > 1:
    shapes = shapes[: int(sys.argv[1])]
print(f"{'M':>7} {'N':>5} {'K':>5} act res |" + "".join(f" cfg{c}: us   TF/s |" for c in range(6)) + " auto")
tot = [0.0] * 7
for (M, N, K, act, res) in shapes:
    line = f"{M:7d} {N:5d} {K:5d} {act:3d} {res:3d} |"
    for ci, cfg in enumerate((0, 1, 2, 3, 4, 5, -1)):
        ms = C.c_float()
        rc = L.k2hip_debug_gemm(m.handle, M, N, K, act, res, cfg, 20, C.byref(ms))
        assert rc == 0, L.k2hip_last_error()
        tf = 2.0 * M * N * K / (ms.value * 1e-3) / 1e12
        tot[ci] += ms.value
        line += f" {ms.value * 1e3:8.1f} {tf:6.1f} |"
    print(line, flush=True)
print("total ms per config (cfg0..5, auto):", [round(t, 3) for t in tot])
