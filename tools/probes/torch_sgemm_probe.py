"""Dev probe (GPU): what the vendor library (rocBLAS / hipBLASLt through torch.matmul, fp32, TF32 off) reaches on the
benchmark's GEMM shapes -- a reference point for gemm_f32_mfma*, not part of the product."""
import time

import torch

torch.backends.cuda.matmul.allow_tf32 = False
shapes = [(4064, 1024, 512), (4064, 512, 1920), (4064, 1536, 512), (4064, 512, 512), (8096, 512, 256), (8096, 256, 768),
          (2048, 768, 2560), (2048, 2048, 768), (16160, 192, 512), (16160, 384, 192), (307040, 384, 128), (16160, 192, 2432)]
dev = torch.device("cuda:0")
for M, N, K in shapes:
    a = torch.randn(M, K, device=dev)
    w = torch.randn(N, K, device=dev)
    for _ in range(3):
        c = a @ w.t()
    torch.cuda.synchronize()
    t = time.perf_counter()
    it = 30
    for _ in range(it):
        c = a @ w.t()
    torch.cuda.synchronize()
    us = (time.perf_counter() - t) / it * 1e6
    print(f"{M:7d} {N:5d} {K:5d}  {us:8.1f} us  {2.0 * M * N * K / us / 1e6:6.1f} TF/s")
