"""The fp32 MFMA GEMM has several kernels for the same math (register-staged, LDS-DMA, ring, pipelined); the product's dispatcher
picks by shape.  Here every family is forced (k2hip_debug_gemm_run, include/k2hip_debug.h) over ragged and edge shapes on operands
made HERE, and the result that comes back is compared with a float64 product + bias + activation + residual computed on the host
with numpy -- every element of every shape, not another GPU kernel: a wrong operand mapping or a wrong epilogue shared by all
kernels cannot pass.  Tolerance: the fp32 rounding of a K-term sum of products of numbers in [-1, 1) (the kernels keep exact fp32
products and accumulate in fp32), plus the hardware exp / log of the Swoosh activations."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

# (M, N, K): rows / columns that are no multiple of any tile, the shortest K each pipeline depth accepts, a K walk with a ragged
# tail of steps (K / 32 = 2 .. 9 covers every remainder of the unrolled loops for 3 and 4 stages), a few-tile long-K case,
# and the headline's most common shape with activation and residual
SHAPES = [(130, 100, 64), (257, 36, 96), (1000, 500, 160), (64, 64, 64), (333, 260, 128), (2048, 192, 192), (515, 132, 224),
          (700, 96, 256), (129, 520, 288), (300, 1000, 2432 // 4), (4064, 512, 512), (256, 768, 2560)]
PIPE = [2001, 2005, 2008, 2013, 2002, 2004, 2009, 2012]          # 32x32x2 form: 128x64, 64x64, 128x128 (16 waves), 128x32, 128x128 (4 waves), 4 stages, 256x64
OTHER = [-1, 66, 5, 9, 100, 108, 118]                             # the dispatcher's own choice, register-staged 64x64, LDS-DMA 128x64 / 64x64, ring tiles
P16 = [3000, 3001, 3002, 3003]                                    # 16x16x4 form (round 5): 64x96 (4 waves of 32x48), 128x96, 64x192, 32x96

ACT_NONE, ACT_SWOOSH_L, ACT_SWOOSH_R, ACT_TANH, ACT_SIGMOID, ACT_RELU, ACT_DOUBLE_SWISH = range(7)


def act_f64(x, act):
    """csrc/kernels.h enum Act, in float64 (icefall's definitions)"""
    if act == ACT_NONE:
        return x
    if act == ACT_SWOOSH_L:
        return np.logaddexp(0.0, x - 4.0) - 0.08 * x - 0.035
    if act == ACT_SWOOSH_R:
        return np.logaddexp(0.0, x - 1.0) - 0.08 * x - 0.313261687
    if act == ACT_TANH:
        return np.tanh(x)
    if act == ACT_SIGMOID:
        return 1.0 / (1.0 + np.exp(-x))
    if act == ACT_RELU:
        return np.maximum(x, 0.0)
    if act == ACT_DOUBLE_SWISH:
        return x / (1.0 + np.exp(-(x - 1.0)))
    raise ValueError(act)


@pytest.fixture(scope="module")
def gemm_run(hip_tiny):
    from k2transducerasr_amd import load_library
    L = load_library()
    fp = C.POINTER(C.c_float)
    L.k2hip_debug_gemm_run.argtypes = [C.c_void_p, fp, fp, fp, fp, fp] + [C.c_int32] * 7

    def run(A, W, bias, res, act, cfg, glu=0, glu_cols=0):
        M, K = A.shape
        N = W.shape[0]
        gc = (glu_cols or N) if glu else 0
        ldo = gc // 2 + (N - gc) if glu else N
        Cout = np.zeros((M, ldo), np.float32)
        p = lambda a: a.ctypes.data_as(fp) if a is not None else None   # noqa: E731
        rc = L.k2hip_debug_gemm_run(hip_tiny.handle, p(A), p(W), p(bias), p(res), p(Cout), M, N, K, act, glu, glu_cols, cfg)
        assert rc == 0, (L.k2hip_last_error(), M, N, K, act, cfg, glu)
        return Cout
    return run


def operands(M, N, K, seed, ldo=None):
    rng = np.random.default_rng(seed)
    A = rng.uniform(-1, 1, (M, K)).astype(np.float32)
    W = rng.uniform(-1, 1, (N, K)).astype(np.float32)
    bias = rng.uniform(-1, 1, N).astype(np.float32)
    res = rng.uniform(-1, 1, (M, ldo or N)).astype(np.float32)
    return A, W, bias, res


def _fits(cfg, M, N, K):
    if cfg >= 3000:                      # 16x16x4 pipelined: three stages
        return K % 32 == 0 and K >= 64
    if cfg >= 2000:                      # pipelined: K >= 32 (stages - 1)
        stages = 4 if (cfg % 100) in (4, 6, 9) else 3
        return K % 32 == 0 and K >= 32 * (stages - 1)
    if cfg >= 100:                       # ring: K a multiple of 32 x its in-workgroup K split
        ks = {100: 1, 108: 2, 118: 1}[cfg]
        return K % (32 * ks) == 0 and K >= 32 * ks
    if cfg == 66:                        # register-staged: any shape
        return True
    return K % 32 == 0 and K >= 64


@pytest.mark.parametrize("family,cfgs", [("pipe", PIPE), ("other", OTHER), ("p16", P16)])
def test_every_gemm_kernel_against_a_float64_product_on_the_host(gemm_run, family, cfgs):
    ran = 0
    for si, (M, N, K) in enumerate(SHAPES):
        A, W, bias, res = operands(M, N, K, 100 + si)
        z = A.astype(np.float64) @ W.astype(np.float64).T + bias.astype(np.float64)
        # |sum| of K products below 1 reaches ~sqrt(K / 9); fp32 accumulation in another order: a few ulps of that
        tol = 2e-5 * max(1.0, K ** 0.5)
        for cfg in cfgs:
            if not _fits(cfg, M, N, K):
                continue
            for act, with_res in ((ACT_NONE, False), (ACT_SWOOSH_L, True), (ACT_TANH, False), (ACT_SWOOSH_R, False), (ACT_SIGMOID, True)):
                want = act_f64(z, act) + (res.astype(np.float64) if with_res else 0.0)
                got = gemm_run(A, W, bias, res if with_res else None, act, cfg)
                err = np.abs(got.astype(np.float64) - want)
                assert np.isfinite(got).all(), (family, cfg, M, N, K, act, "an element was not written")
                assert err.max() <= tol, (family, cfg, M, N, K, act, with_res, float(err.max()), tol, np.unravel_index(err.argmax(), err.shape))
                ran += 1
    assert ran >= 40


def test_no_bias_and_ragged_k_on_the_register_staged_kernel(gemm_run):
    """the kernel everything else used to be compared WITH: K tails that are no multiple of 32 (rows are float4-aligned: K % 4 == 0
    is the engine's one requirement on a Linear's input width), no bias"""
    for si, (M, N, K) in enumerate([(70, 50, 40), (129, 33, 12), (300, 260, 100), (64, 64, 4)]):
        A, W, _, _ = operands(M, N, K, 300 + si)
        want = A.astype(np.float64) @ W.astype(np.float64).T
        for cfg in (-1, 66):
            got = gemm_run(A, W, None, None, ACT_NONE, cfg)
            assert np.abs(got - want).max() <= 2e-5 * max(1.0, K ** 0.5), (cfg, M, N, K)


def test_gated_epilogue_in_every_kernel_family(gemm_run):
    """GemmArgs.glu (the conv modules' GLU and NonlinAttention's tanh gate, fused into the producing GEMM): value and gate sit 16
    lanes apart in the 32-column accumulator layout, so every kernel family that can be asked for it is forced here and compared
    with the float64 product gated on the host.  Mode 1: all columns paired, sigmoid; mode 2: the first 2N/3 paired with tanh, the
    rest passed through behind them."""
    ran = 0
    for si, (M, N, K) in enumerate([(300, 128, 64), (1000, 384, 160), (257, 1152, 96), (4064, 1024, 512), (130, 96 * 4, 256)]):
        A, W, bias, _ = operands(M, N, K, 200 + si)
        z = A.astype(np.float64) @ W.astype(np.float64).T + bias.astype(np.float64)
        tol = 3e-5 * max(1.0, K ** 0.5)
        for mode in (1, 2):
            gc = N if mode == 1 else (2 * N // 3) // 32 * 32
            if gc < 32:
                continue
            blocks = z[:, :gc].reshape(M, gc // 32, 32)
            val, gate = blocks[:, :, :16], blocks[:, :, 16:]
            gated = val * (act_f64(gate, ACT_SIGMOID) if mode == 1 else np.tanh(gate))
            want = np.concatenate([gated.reshape(M, gc // 2), z[:, gc:]], axis=1)
            for cfg in [-1, 66, 5, 9, 100, 118, 2001, 2005, 2008, 2013]:
                if not _fits(cfg, M, N, K):
                    continue
                got = gemm_run(A, W, bias, None, ACT_NONE, cfg, glu=mode, glu_cols=0 if gc == N else gc)
                assert got.shape == want.shape and np.isfinite(got).all(), (cfg, M, N, K, mode)
                err = np.abs(got - want).max()
                assert err <= tol, (cfg, M, N, K, mode, float(err), tol)
                ran += 1
    assert ran >= 40
