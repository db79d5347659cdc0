// fp32 MFMA GEMM for gfx950:  C = act(A . W^T + bias) (+ residual)
//
// Why fp32 MFMA: greedy decoding must reproduce the reference's fp32 token
// sequence, so products are exact f32 (v_mfma_f32_32x32x2_f32 is bitwise an fmaf
// chain) at the f32 matrix rate (157 TFLOP/s peak on MI355X).
//
// Structure (one workgroup = 4 waves = 256 threads):
//   - block tile BM x BN, K step 32; wave tile WM x WN made of 32x32 MFMA tiles
//   - A and W are both K-contiguous ([M,K] activations, [N,K] torch Linear
//     weights), so both operands stage the same way: coalesced float4 global
//     loads -> registers -> LDS rows padded to 36 floats (conflict-free for the
//     ds_read_b128 fragment reads below)
//   - fragment read: lane (i = lane&31, h = lane>>5) reads 4 consecutive k of row
//     i at k = 8g + 4h as ONE ds_read_b128 and feeds element e to MFMA step e.
//     The MFMA's two k-slots (h = 0, 1) then hold k = 8g+e and 8g+4+e; the order
//     in which k is summed is a permutation of 0..K-1, identical for A and W.
//   - double-buffered LDS, next tile's global loads in flight during the MFMAs,
//     one barrier per K step
//   - epilogue straight from the accumulators: bias, activation, residual add
//   - optional gathers: implicit-conv row/k mapping for A (NHWC), [K,N] B operand
//     (transposed while staging) for attention-weights x values products.
#include "kernels.h"

namespace k2hip {

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int BK = 32;
constexpr int LDSK = 36;  // padded row length (floats): 144-byte rows, 16-byte aligned

__device__ __forceinline__ float logaddexp0(float z) { return fmaxf(z, 0.f) + log1pf(expf(-fabsf(z))); }

__device__ __forceinline__ float apply_act(float v, int act) {
    switch (act) {
        case ACT_SWOOSH_L: return logaddexp0(v - 4.0f) - 0.08f * v - 0.035f;
        case ACT_SWOOSH_R: return logaddexp0(v - 1.0f) - 0.08f * v - 0.313261687f;
        case ACT_TANH: return tanhf(v);
        case ACT_SIGMOID: return 1.0f / (1.0f + expf(-v));
        case ACT_RELU: return fmaxf(v, 0.f);
        default: return v;
    }
}

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256) void gemm_f32_mfma(GemmArgs g) {
    constexpr int MT = WM / 32, NT = WN / 32;
    constexpr int WCOLS = BN / WN;
    static_assert((BM / WM) * (BN / WN) == 4, "4 waves per workgroup");
    constexpr int PA = BM / 32;  // float4 loads per thread for the A tile
    constexpr int PB = BN / 32;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sA = smem;                        // [2][BM][LDSK]
    float* sB = smem + 2 * BM * LDSK;        // [2][BN][LDSK]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wr = wave / WCOLS, wc = wave % WCOLS;
    const int li = lane & 31, lh = lane >> 5;

    const int z = blockIdx.z;
    const int z0 = z % g.nb0, z1 = z / g.nb0;
    const float* __restrict__ A = g.A + z0 * g.sA0 + z1 * g.sA1;
    const float* __restrict__ W = g.W + z0 * g.sW0 + z1 * g.sW1;
    float* __restrict__ C = g.C + z0 * g.sC0 + z1 * g.sC1;
    const float* __restrict__ R = g.res ? g.res + z0 * g.sR0 + z1 * g.sR1 : nullptr;

    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int Kp = (g.K + 3) & ~3;

    // ---- per-thread global-load coordinates
    const int ld_row = tid >> 3;       // 0..31
    const int ld_kq = (tid & 7) * 4;   // k offset of this thread's float4
    long long a_base[PA];
    bool a_ok[PA];
#pragma unroll
    for (int p = 0; p < PA; p++) {
        int row = m0 + ld_row + 32 * p;
        a_ok[p] = row < g.M;
        if (g.cv_Fout > 0) {
            int f = row % g.cv_Fout;
            int bt = row / g.cv_Fout;
            int t = bt % g.cv_Tout, b = bt / g.cv_Tout;
            a_base[p] = (((long long)b * g.cv_Tin + (long long)t * g.cv_st) * g.cv_Fin + (long long)f * g.cv_sf) * g.cv_C;
        } else {
            a_base[p] = (long long)row * g.lda;
        }
    }
    float4 ra[PA], rb[PB];

    auto load_tile = [&](int k0) {
        int k = k0 + ld_kq;
        long long koff = k;
        if (g.seg_len > 0) koff = (long long)(k / g.seg_len) * g.seg_stride + (k % g.seg_len);
#pragma unroll
        for (int p = 0; p < PA; p++) {
            ra[p] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (a_ok[p] && k < Kp) ra[p] = *reinterpret_cast<const float4*>(A + a_base[p] + koff);
        }
        if (!g.w_kn) {
#pragma unroll
            for (int p = 0; p < PB; p++) {
                int col = n0 + ld_row + 32 * p;
                rb[p] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (col < g.N && k < Kp) rb[p] = *reinterpret_cast<const float4*>(W + (long long)col * g.ldw + k);
            }
        } else {
            // W is [K,N]: this thread loads 4 consecutive n of one k row
            constexpr int NQ = BN / 4;           // float4 per k row
            constexpr int KR = 256 / NQ;         // k rows per pass
#pragma unroll
            for (int p = 0; p < PB; p++) {
                int kk = k0 + tid / NQ + KR * p;
                int col = n0 + (tid % NQ) * 4;
                rb[p] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (kk < g.K && col < g.N) rb[p] = *reinterpret_cast<const float4*>(W + (long long)kk * g.ldw + col);
            }
        }
    };
    auto store_tile = [&](int buf) {
        float* a = sA + buf * BM * LDSK;
        float* b = sB + buf * BN * LDSK;
#pragma unroll
        for (int p = 0; p < PA; p++) *reinterpret_cast<float4*>(a + (ld_row + 32 * p) * LDSK + ld_kq) = ra[p];
        if (!g.w_kn) {
#pragma unroll
            for (int p = 0; p < PB; p++) *reinterpret_cast<float4*>(b + (ld_row + 32 * p) * LDSK + ld_kq) = rb[p];
        } else {
            constexpr int NQ = BN / 4;
            constexpr int KR = 256 / NQ;
#pragma unroll
            for (int p = 0; p < PB; p++) {
                int kk = tid / NQ + KR * p;
                int c = (tid % NQ) * 4;
                b[(c + 0) * LDSK + kk] = rb[p].x;
                b[(c + 1) * LDSK + kk] = rb[p].y;
                b[(c + 2) * LDSK + kk] = rb[p].z;
                b[(c + 3) * LDSK + kk] = rb[p].w;
            }
        }
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; i++)
#pragma unroll
        for (int j = 0; j < NT; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

    const int nk = (g.K + BK - 1) / BK;
    load_tile(0);
    store_tile(0);
    __syncthreads();
    int cur = 0;
    for (int kt = 0; kt < nk; kt++) {
        if (kt + 1 < nk) load_tile((kt + 1) * BK);
        const float* a = sA + cur * BM * LDSK + (wr * WM + li) * LDSK + 4 * lh;
        const float* b = sB + cur * BN * LDSK + (wc * WN + li) * LDSK + 4 * lh;
#pragma unroll
        for (int gk = 0; gk < 4; gk++) {
            float4 fa[MT], fb[NT];
#pragma unroll
            for (int i = 0; i < MT; i++) fa[i] = *reinterpret_cast<const float4*>(a + i * 32 * LDSK + gk * 8);
#pragma unroll
            for (int j = 0; j < NT; j++) fb[j] = *reinterpret_cast<const float4*>(b + j * 32 * LDSK + gk * 8);
#pragma unroll
            for (int i = 0; i < MT; i++)
#pragma unroll
                for (int j = 0; j < NT; j++) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].x, fb[j].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].y, fb[j].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].z, fb[j].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].w, fb[j].w, acc[i][j], 0, 0, 0);
                }
        }
        if (kt + 1 < nk) store_tile(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }

    // ---- epilogue: C/D layout of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
    for (int j = 0; j < NT; j++) {
        int col = n0 + wc * WN + j * 32 + li;
        if (col >= g.N) continue;
        float bv = g.bias ? g.bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < MT; i++) {
#pragma unroll
            for (int r = 0; r < 16; r++) {
                int row = m0 + wr * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (row < g.M) {
                    float v = apply_act(acc[i][j][r] + bv, g.act);
                    if (R) v += R[(long long)row * g.ldr + col];
                    C[(long long)row * g.ldc + col] = v;
                }
            }
        }
    }
}

template <int BM, int BN, int WM, int WN>
void launch(const Ctx& ctx, const GemmArgs& a) {
    dim3 grid(cdiv(a.N, BN), cdiv(a.M, BM), a.nb0 * a.nb1);
    size_t lds = sizeof(float) * 2 * (BM + BN) * LDSK;
    static bool attr_set = false;
    if (!attr_set) {
        K2_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f32_mfma<BM, BN, WM, WN>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_f32_mfma<BM, BN, WM, WN>), grid, dim3(256), lds, ctx.stream, a);
}

}  // namespace

void gemm(const Ctx& ctx, const GemmArgs& a) {
    K2_REQUIRE(a.M > 0 && a.N > 0 && a.K > 0, "gemm: empty shape %dx%dx%d", a.M, a.N, a.K);
    K2_REQUIRE(a.cv_Fout > 0 || a.lda % 4 == 0, "gemm: lda %d must be a multiple of 4", a.lda);
    K2_REQUIRE(a.w_kn || a.K % 4 == 0 || a.lda >= ((a.K + 3) & ~3), "gemm: K %d needs zero-padded A rows", a.K);
    K2_REQUIRE(a.ldw % 4 == 0, "gemm: ldw %d must be a multiple of 4", a.ldw);
    K2_REQUIRE(!a.w_kn || a.N % 4 == 0, "gemm: [K,N] operand needs N %% 4 == 0 (N=%d)", a.N);
    K2_REQUIRE(a.seg_len == 0 || a.seg_len % 4 == 0, "gemm: seg_len %d must be a multiple of 4", a.seg_len);
    const double fl = 2.0 * a.M * (double)a.N * a.K * a.nb0 * a.nb1;
    ctx.add_flops(fl, 0.0, 1);
    if (ctx.dry) return;
    if (ctx.instrument) K2_HIP(hipEventRecord(ctx.next_event(), ctx.stream));
    // tile choice: big tiles when they still give every CU >= 2 workgroups
    long long blocks128 = (long long)cdiv(a.M, 128) * cdiv(a.N, 128) * a.nb0 * a.nb1;
    if (a.N > 64 && blocks128 >= 512) launch<128, 128, 64, 64>(ctx, a);
    else launch<64, 64, 32, 32>(ctx, a);
    K2_HIP(hipGetLastError());
    if (ctx.instrument) K2_HIP(hipEventRecord(ctx.next_event(), ctx.stream));
}

void linear(const Ctx& ctx, const float* A, int lda, const float* W, const float* bias, float* C, int ldc, int M, int K,
            int N, int act, const float* res, int ldr) {
    GemmArgs g;
    g.A = A; g.lda = lda; g.W = W; g.ldw = K; g.bias = bias; g.C = C; g.ldc = ldc;
    g.M = M; g.N = N; g.K = K; g.act = act; g.res = res; g.ldr = ldr;
    gemm(ctx, g);
}

}  // namespace k2hip
