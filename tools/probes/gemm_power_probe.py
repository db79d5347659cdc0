"""Dev probe (GPU): shader clock and package power (rocm-smi samples) while one large fp32 GEMM runs in a loop -- the vendor
library next to this repo's kernels.  Tells whether a lower sustained rate comes from stalls or from a lower clock."""
import ctypes as C
import os
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import k2transducerasr_amd as pkg
from k2transducerasr_amd.synth import write_synthetic_model

samples = []
stop = False


def sampler():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--csv"], capture_output=True, text=True, timeout=10).stdout
            samples.append((time.perf_counter(), out))
        except Exception as e:  # noqa
            samples.append((time.perf_counter(), "ERR " + str(e)))
        time.sleep(0.3)


def summarize(t0, t1):
    rows = [o for (t, o) in samples if t0 + 0.5 < t < t1]
    return rows[len(rows) // 2] if rows else "(no sample)"


th = threading.Thread(target=sampler, daemon=True)
th.start()
torch.backends.cuda.matmul.allow_tf32 = False
dev = torch.device("cuda:0")
M = N = K = 8192
a = torch.rand(M, K, device=dev) * 2 - 1
w = torch.rand(N, K, device=dev) * 2 - 1
c = a @ w.t()
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 0
while time.perf_counter() - t0 < 4.0:
    for _ in range(20):
        c = a @ w.t()
    torch.cuda.synchronize()
    n += 20
t1 = time.perf_counter()
print(f"vendor: {2.0 * M * N * K * n / (t1 - t0) / 1e12:.1f} TF/s over {t1 - t0:.1f} s")
print(summarize(t0, t1))
del a, w, c
path = "/tmp/tune_tiny.k2w"
if not os.path.exists(path):
    write_synthetic_model(path, "zipformer2-tiny-test")
m = pkg.Model(path, 0)
L = pkg.load_library()
L.k2hip_debug_gemm.argtypes = [C.c_void_p] + [C.c_int32] * 7 + [C.POINTER(C.c_float)]
for cfg in [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["118", "100", "1011"])]:
    ms = C.c_float()
    t0 = time.perf_counter()
    L.k2hip_debug_gemm(m.handle, M, N, K, 0, 0, cfg, 500, C.byref(ms))
    t1 = time.perf_counter()
    print(f"k2hip cfg {cfg}: {2.0 * M * N * K / ms.value / 1e9:.1f} TF/s over {t1 - t0:.1f} s")
    print(summarize(t0, t1))
stop = True
