"""The fp32 MFMA GEMM has several kernels for the same math (register-staged, LDS-DMA, ring, pipelined);
the product's dispatcher picks by shape.  Here every family is forced (k2hip_debug_gemm_check's tuning hook) over ragged
and edge shapes and compared with the register-staged kernel on the same operands: same fp32 products, another summation order,
so the tolerance is that of an fp32 dot product of K terms in [-1, 1)."""
import ctypes as C

import pytest

pytestmark = pytest.mark.gpu

# (M, N, K): rows / columns that are no multiple of any tile, the shortest K each pipeline depth accepts, a K walk with a ragged
# tail of steps (K / 32 = 2 .. 9 covers every remainder of the unrolled loops for 3 and 4 stages), a few-tile long-K case,
# and the headline's most common shape with activation and residual
SHAPES = [(130, 100, 64), (257, 36, 96), (1000, 500, 160), (64, 64, 64), (333, 260, 128), (2048, 192, 192), (515, 132, 224),
          (700, 96, 256), (129, 520, 288), (300, 1000, 2432 // 4), (4064, 512, 512), (256, 768, 2560)]
PIPE = [2001, 2005, 2008, 2013, 2002, 2004, 2009, 2012]          # 32x32x2 form: 128x64, 64x64, 128x128 (16 waves), 128x32, 128x128 (4 waves), 4 stages, 256x64
OTHER = [-1, 5, 9, 100, 108, 118]                                 # the dispatcher's own choice, LDS-DMA 128x64 / 64x64, ring tiles


@pytest.fixture(scope="module")
def gemm_check(hip_tiny):
    from k2transducerasr_amd import load_library
    L = load_library()
    L.k2hip_debug_gemm_check.argtypes = [C.c_void_p] + [C.c_int32] * 7 + [C.POINTER(C.c_float), C.POINTER(C.c_float)]

    def run(M, N, K, act, res, cfg):
        ms, err = C.c_float(), C.c_float()
        rc = L.k2hip_debug_gemm_check(hip_tiny.handle, M, N, K, act, res, cfg, 1, C.byref(ms), C.byref(err))
        return rc, err.value
    return run


def _fits(cfg, M, N, K):
    if cfg >= 2000:                      # pipelined: K >= 32 (stages - 1)
        stages = 4 if (cfg % 100) in (4, 6, 9) else 3
        return K % 32 == 0 and K >= 32 * (stages - 1)
    if cfg >= 100:                       # ring: K a multiple of 32 x its in-workgroup K split
        ks = {100: 1, 108: 2, 118: 1}[cfg]
        return K % (32 * ks) == 0 and K >= 32 * ks
    return K % 32 == 0 and K >= 64


@pytest.mark.parametrize("family,cfgs", [("pipe", PIPE), ("other", OTHER)])
def test_every_gemm_kernel_agrees_with_the_register_staged_one(gemm_check, family, cfgs):
    ran = 0
    for (M, N, K) in SHAPES:
        tol = 2e-5 * max(1.0, K ** 0.5)  # fp32 rounding of a K-term sum of products below 1, summed in another order
        for cfg in cfgs:
            if not _fits(cfg, M, N, K):
                continue
            for act, res in ((0, 0), (1, 1), (3, 0)):   # none / SwooshL + residual / tanh
                rc, err = gemm_check(M, N, K, act, res, cfg)
                assert rc == 0, (family, cfg, M, N, K, act, res)
                assert err <= tol, (family, cfg, M, N, K, act, res, err, tol)
                ran += 1
    assert ran >= 30


def test_gated_epilogue_in_every_kernel_family(gemm_check):
    """GemmArgs.glu (the conv modules' GLU and NonlinAttention's tanh gate, fused into the producing GEMM): value and gate sit 16
    lanes apart in the 32-column accumulator layout, so every kernel family that can be asked for it is forced here and compared
    with the PLAIN product of the register-staged kernel gated on the host (an independent path: no shared epilogue code decides
    the expected values).  Mode 1: all columns paired, sigmoid; mode 2: the first 2N/3 paired with tanh, the rest passed through."""
    ran = 0
    for (M, N, K) in [(300, 128, 64), (1000, 384, 160), (257, 1152, 96), (4064, 1024, 512), (130, 96 * 4, 256)]:
        for mode in (101, 102):
            if mode == 102 and ((2 * N // 3) // 32 * 32) < 32:
                continue
            tol = 3e-5 * max(1.0, K ** 0.5)
            for cfg in [-1, 5, 9, 100, 118, 2001, 2005, 2008, 2013]:
                if not _fits(cfg, M, N, K):
                    continue
                rc, err = gemm_check(M, N, K, mode, 0, cfg)
                assert rc == 0, (cfg, M, N, K, mode)
                assert err <= tol, (cfg, M, N, K, mode, err, tol)
                ran += 1
    assert ran >= 32
