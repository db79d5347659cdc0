"""Dev probe (GPU): an upper bound for split-bf16 products before writing a kernel for them.  An f32 product a.w with both operands
split into three bf16 planes (a = a1 + a2 + a3) and the six largest plane products kept is ONE bf16 GEMM with K' = 6 K over
concatenated planes: [a1|a1|a2|a1|a2|a3] . [w1|w2|w1|w3|w2|w1].  What the vendor library reaches on that K' (operands already
split and resident) against its own f32 GEMM on K says what the matrix pipe could buy on the benchmark's shapes; the error of the
six-product sum against float64 says what it costs.  Not part of the product."""
import time

import torch

torch.backends.cuda.matmul.allow_tf32 = False
dev = torch.device("cuda:0")


def split3(x):
    p1 = x.to(torch.bfloat16)
    r = x - p1.float()
    p2 = r.to(torch.bfloat16)
    p3 = (r - p2.float()).to(torch.bfloat16)
    return p1, p2, p3


def timed(f, it=50):
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(it):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / it * 1e6


shapes = [(4064, 512, 512), (4064, 1024, 512), (4064, 1536, 512), (4064, 512, 1920), (4064, 512, 2048), (8096, 512, 256),
          (8096, 256, 768), (2048, 768, 768), (2048, 768, 2560), (2048, 2048, 768), (16160, 192, 512), (16160, 384, 192)]
for M, N, K in shapes:
    a = torch.randn(M, K, device=dev)
    w = torch.randn(N, K, device=dev) * 0.05
    a1, a2, a3 = split3(a)
    w1, w2, w3 = split3(w)
    A6 = torch.cat([a1, a1, a2, a1, a2, a3], 1).contiguous()
    W6 = torch.cat([w1, w2, w1, w3, w2, w1], 1).contiguous()
    A3 = torch.cat([a1, a1, a2], 1).contiguous()
    W3 = torch.cat([w1, w2, w1], 1).contiguous()
    us32 = timed(lambda: a @ w.t())
    us6 = timed(lambda: A6 @ W6.t())
    us3 = timed(lambda: A3 @ W3.t())
    ref = a.double() @ w.double().t()
    e32 = ((a @ w.t()).double() - ref).abs().max().item()
    # the six products summed smallest first in f32 (what a kernel accumulating plane pairs in that order would hold)
    parts = [(a3, w1), (a1, w3), (a2, w2), (a2, w1), (a1, w2), (a1, w1)]
    acc = torch.zeros(M, N, device=dev, dtype=torch.float64)
    for p, q in parts:
        acc += p.double() @ q.double().t()
    e6 = (acc - ref).abs().max().item()           # what the DROPPED plane pairs cost (the products themselves are exact)
    scale = ref.abs().max().item()
    print(f"{M:6d} {N:5d} {K:5d}  f32 {us32:7.1f} us ({2.0 * M * N * K / us32 / 1e6:6.1f} TF/s)   bf16 K'=6K {us6:7.1f} us "
          f"({12.0 * M * N * K / us6 / 1e6:7.1f} TF/s executed, x{us32 / us6:4.2f})   K'=3K {us3:7.1f} us   "
          f"max|err| f32 {e32:.2e}  six planes {e6:.2e}  (max|c| {scale:.1f})", flush=True)
