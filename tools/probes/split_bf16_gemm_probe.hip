// Dev probe (GPU): an f32 GEMM C = A . W^T whose products run on the bf16 matrix pipe.  Every f32 operand is the exact sum of
// three bf16 planes (x = x1 + x2 + x3: the top 8, the next 8 and the last 8 significant bits -- truncation, no rounding); of the nine
// plane products the six largest are kept (a1 w1, a1 w2, a2 w1, a2 w2, a1 w3, a3 w1: what is dropped is below 2^-23 of |a||w|),
// each an exact bf16 x bf16 product accumulated in f32 by v_mfma_f32_32x32x16_bf16.  Six MFMAs of 16x the f32 rate per product:
// 2.67x the f32 matrix rate if the operands (3 planes each) can be fed.  This probe measures that "if" on the benchmark's shapes:
//   mode 0: A and W pre-split and pre-packed in fragment order in HBM (upper bound: the loop is LDS-DMA + ds_read + MFMA)
//   mode 1: A as the engine has it (f32 row-major), split by the vector ALU inside the kernel
// Usage: split_bf16_gemm_probe M N K [mode] [BM] [BN]      (K % 32 == 0).  Not part of the product.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define CHECK(x)                                                                      \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) {                                                       \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            exit(2);                                                                  \
        }                                                                             \
    } while (0)

struct Args {
    const float* A;      // [M][lda] (mode 1)
    const uint4* Ap;     // packed planes (mode 0): [mblk][K/16][3][64] x 16 B
    const uint4* Wp;     // packed planes: [nblk][K/16][3][64] x 16 B
    float* C;
    int M, N, K, lda, ldc;
};

constexpr int UNIT = 3072;  // bytes of one (32-row block, 16-deep k slice): 3 planes x 64 lanes x 16 B

// the wave's LDS-DMA: 64 lanes x 16 B from per-lane global pointers to LDS [dst, dst + 1 KB)
__device__ __forceinline__ void dma16(const void* gp, unsigned dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gp), "s"(dst)
                 : "memory");
}

// BM x BN tile, waves of 64 x 64, SL k slices of 16 per K step, NST stages
template <int BM, int BN, int SL, int NST, int MODE>
__global__ __launch_bounds__(64 * (BM / 64) * (BN / 64)) void gemm_split(Args g) {
    constexpr int NW = (BM / 64) * (BN / 64), WCOLS = BN / 64;
    constexpr int AU = (BM / 32) * SL, BU = (BN / 32) * SL;  // units per stage
    constexpr int STAGE = (AU + BU) * UNIT;
    constexpr int PF = NST - 1;
    constexpr int DU = MODE == 0 ? AU + BU : BU;             // units that arrive by DMA
    constexpr int NINST = 3 * DU;                            // 1 KB instructions per stage
    static_assert(NINST % NW == 0, "every wave issues the same number of DMAs");
    constexpr int IPW = NINST / NW;
    static_assert(AU % NW == 0, "A units split evenly");
    constexpr int AUW = AU / NW;                             // A units a wave splits per K step (mode 1)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WCOLS, wc = wave % WCOLS;
    const int li = lane & 31, lh = lane >> 5;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int nk = g.K / (16 * SL), KS = g.K / 16;
    const unsigned lds_base = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)smem;

    // DMA sources: instruction q of this wave moves plane p of unit u
    const char* src[IPW];
    unsigned dsto[IPW];
#pragma unroll
    for (int q = 0; q < IPW; q++) {
        const int inst = wave + q * NW, u = inst / 3, p = inst % 3;
        int lu;  // unit index within the stage
        const char* base;
        if (MODE == 0 && u < AU) {
            const int blk = u / SL, s = u % SL;
            base = (const char*)g.Ap + ((size_t)(m0 / 32 + blk) * KS + s) * UNIT;
            lu = u;
        } else {
            const int ub = MODE == 0 ? u - AU : u;
            const int blk = ub / SL, s = ub % SL;
            base = (const char*)g.Wp + ((size_t)(n0 / 32 + blk) * KS + s) * UNIT;
            lu = AU + ub;
        }
        src[q] = base + p * 1024 + lane * 16;
        dsto[q] = lu * UNIT + p * 1024;
    }
    auto issue = [&](int kt) {
        const unsigned st = lds_base + (kt % NST) * STAGE;
#pragma unroll
        for (int q = 0; q < IPW; q++) dma16(src[q] + (size_t)kt * SL * UNIT, __builtin_amdgcn_readfirstlane(st + dsto[q]));
    };
    // mode 1: this wave's A units of a K step: (blk, s) -> rows m0 + 32 blk + li, k = 16 (kt SL + s) + 8 lh ..
    const float* arow[AUW];
    unsigned adst[AUW];
    if (MODE == 1) {
#pragma unroll
        for (int a = 0; a < AUW; a++) {
            const int u = wave * AUW + a, blk = u / SL, s = u % SL;
            arow[a] = g.A + (size_t)min(m0 + 32 * blk + li, g.M - 1) * g.lda + 16 * s + 8 * lh;
            adst[a] = u * UNIT + lane * 16;
        }
    }
    // the A loads go through asm like the DMAs: the compiler's own wait counting knows nothing of the DMAs and would drain the queue
    f32x4 an[AUW][2];
    auto load_a = [&](int kt) {
#pragma unroll
        for (int a = 0; a < AUW; a++) {
            const float* p = arow[a] + kt * 16 * SL;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(an[a][0]) : "v"(p) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, off offset:16" : "=&v"(an[a][1]) : "v"(p) : "memory");
        }
    };
    auto wait_a = [&]() {  // queue: [.., A loads, the IPW DMAs issued behind them]
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(IPW) : "memory");
#pragma unroll
        for (int a = 0; a < AUW; a++) {
            asm volatile("" : "+v"(an[a][0]));
            asm volatile("" : "+v"(an[a][1]));
        }
    };
    auto split_store = [&](int kt) {
        char* st = smem + (kt % NST) * STAGE;
#pragma unroll
        for (int a = 0; a < AUW; a++) {
            unsigned x[8];
#pragma unroll
            for (int e = 0; e < 4; e++) x[e] = __float_as_uint(an[a][0][e]), x[4 + e] = __float_as_uint(an[a][1][e]);
            unsigned r1[8], r2[8];
#pragma unroll
            for (int e = 0; e < 8; e++) {
                r1[e] = __float_as_uint(__uint_as_float(x[e]) - __uint_as_float(x[e] & 0xffff0000u));
                r2[e] = __float_as_uint(__uint_as_float(r1[e]) - __uint_as_float(r1[e] & 0xffff0000u));
            }
            u32x4 p1, p2, p3;
#pragma unroll
            for (int d = 0; d < 4; d++) {
                p1[d] = __builtin_amdgcn_perm(x[2 * d + 1], x[2 * d], 0x07060302u);
                p2[d] = __builtin_amdgcn_perm(r1[2 * d + 1], r1[2 * d], 0x07060302u);
                p3[d] = __builtin_amdgcn_perm(r2[2 * d + 1], r2[2 * d], 0x07060302u);
            }
            *reinterpret_cast<u32x4*>(st + adst[a]) = p1;
            *reinterpret_cast<u32x4*>(st + adst[a] + 1024) = p2;
            *reinterpret_cast<u32x4*>(st + adst[a] + 2048) = p3;
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

    // prologue
#pragma unroll
    for (int p = 0; p < PF; p++) {
        if (p < nk) {
            if (MODE == 1) load_a(p);
            issue(p);
            if (MODE == 1) {
                wait_a();
                split_store(p);
            }
        }
    }
    for (int kt = 0; kt < nk; kt++) {
        if (kt + PF - 1 < nk && PF > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((PF - 1) * IPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (kt + PF < nk) {
            if (MODE == 1) load_a(kt + PF);
            issue(kt + PF);
        }
        const char* st = smem + (kt % NST) * STAGE;
#pragma unroll
        for (int s = 0; s < SL; s++) {
            bf16x8 fa[2][3], fb[2][3];
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int p = 0; p < 3; p++) {
                    fa[i][p] = *reinterpret_cast<const bf16x8*>(st + ((wr * 2 + i) * SL + s) * UNIT + p * 1024 + lane * 16);
                    fb[i][p] = *reinterpret_cast<const bf16x8*>(st + (AU + (wc * 2 + i) * SL + s) * UNIT + p * 1024 + lane * 16);
                }
            // smallest products first
            constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
            for (int p = 0; p < 6; p++)
#pragma unroll
                for (int i = 0; i < 2; i++)
#pragma unroll
                    for (int j = 0; j < 2; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][PA[p]], fb[j][PB[p]], acc[i][j], 0, 0, 0);
        }
        if (MODE == 1 && kt + PF < nk) {
            wait_a();
            split_store(kt + PF);
        }
    }
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int col = n0 + wc * 64 + j * 32 + li;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int row = m0 + wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (row < g.M && col < g.N) g.C[(size_t)row * g.ldc + col] = acc[i][j][r];
            }
        }
}

// ---- host -----------------------------------------------------------------------------------------------------------------
static void split3(float x, uint16_t out[3]) {
    uint32_t b;
    float r = x;
    for (int p = 0; p < 3; p++) {
        memcpy(&b, &r, 4);
        b &= 0xffff0000u;
        out[p] = (uint16_t)(b >> 16);
        float h;
        memcpy(&h, &b, 4);
        r = r - h;
    }
}
// [rows / 32][K / 16][3][64 lanes][8]: lane (i, h) holds row 32 blk + i, k = 16 ks + 8 h + j
static std::vector<uint16_t> pack(const std::vector<float>& X, int rows, int K, int rows_padded) {
    std::vector<uint16_t> P((size_t)rows_padded * K * 3, 0);
    const int KS = K / 16;
    for (int r = 0; r < rows; r++)
        for (int k = 0; k < K; k++) {
            uint16_t pl[3];
            split3(X[(size_t)r * K + k], pl);
            const int blk = r / 32, i = r % 32, ks = k / 16, h = (k % 16) / 8, j = k % 8;
            for (int p = 0; p < 3; p++) P[((((size_t)blk * KS + ks) * 3 + p) * 64 + (h * 32 + i)) * 8 + j] = pl[p];
        }
    return P;
}

template <int BM, int BN, int SL, int NST, int MODE>
static float run(const Args& a, int iters) {
    constexpr int STAGE = ((BM / 32) + (BN / 32)) * SL * UNIT;
    const size_t lds = (size_t)STAGE * NST;
    CHECK(hipFuncSetAttribute((const void*)gemm_split<BM, BN, SL, NST, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    dim3 grid((a.N + BN - 1) / BN, (a.M + BM - 1) / BM), block(64 * (BM / 64) * (BN / 64));
    for (int i = 0; i < 5; i++) gemm_split<BM, BN, SL, NST, MODE><<<grid, block, lds>>>(a);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < iters; i++) gemm_split<BM, BN, SL, NST, MODE><<<grid, block, lds>>>(a);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("  tile %3d x %3d, %d slices x %d stages (%zu KB LDS, %d workgroups of %d): ", BM, BN, SL, NST, lds / 1024, grid.x * grid.y, block.x);
    return ms * 1000.f / iters;
}

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 4064, N = argc > 2 ? atoi(argv[2]) : 512, K = argc > 3 ? atoi(argv[3]) : 512;
    const int mode = argc > 4 ? atoi(argv[4]) : 0;
    if (K % 32 || N % 128) {
        fprintf(stderr, "K %% 32 == 0 and N %% 128 == 0\n");
        return 1;
    }
    const int Mp = (M + 127) / 128 * 128;
    std::vector<float> A((size_t)M * K), W((size_t)N * K);
    uint32_t s = 12345;
    auto rnd = [&]() {
        s = s * 1664525u + 1013904223u;
        return ((s >> 8) & 0xffff) / 65536.f - 0.5f;
    };
    for (auto& v : A) v = 4.f * rnd() * (1.f + rnd());
    for (auto& v : W) v = 0.2f * rnd();
    std::vector<uint16_t> Ap = pack(A, M, K, Mp), Wp = pack(W, N, K, N);
    Args a{};
    float *dA, *dC;
    void *dAp, *dWp;
    CHECK(hipMalloc(&dA, A.size() * 4));
    CHECK(hipMalloc(&dC, (size_t)M * N * 4));
    CHECK(hipMalloc(&dAp, Ap.size() * 2));
    CHECK(hipMalloc(&dWp, Wp.size() * 2));
    CHECK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dAp, Ap.data(), Ap.size() * 2, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dWp, Wp.data(), Wp.size() * 2, hipMemcpyHostToDevice));
    a.A = dA, a.Ap = (const uint4*)dAp, a.Wp = (const uint4*)dWp, a.C = dC, a.M = M, a.N = N, a.K = K, a.lda = K, a.ldc = N;
    printf("%d x %d x %d, mode %d (%s)\n", M, N, K, mode, mode ? "A split in the kernel" : "A pre-split in HBM");
    auto report = [&](float us) {
        printf("%7.1f us  %6.1f TF/s of f32 products (%6.1f executed)\n", us, 2.0 * M * N * K / us / 1e6, 12.0 * M * N * K / us / 1e6);
        // check a sample against float64 on the host
        std::vector<float> C((size_t)M * N);
        CHECK(hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost));
        double worst = 0, scale = 0;
        for (int t = 0; t < 4000; t++) {
            const int r = (int)(((uint64_t)t * 2654435761u) % M), c = (int)(((uint64_t)t * 40503u + 17) % N);
            double ref = 0;
            for (int k = 0; k < K; k++) ref += (double)A[(size_t)r * K + k] * W[(size_t)c * K + k];
            worst = std::max(worst, std::fabs(ref - C[(size_t)r * N + c]));
            scale = std::max(scale, std::fabs(ref));
        }
        // the last row / column too
        for (int c = 0; c < N; c += 37) {
            double ref = 0;
            for (int k = 0; k < K; k++) ref += (double)A[(size_t)(M - 1) * K + k] * W[(size_t)c * K + k];
            worst = std::max(worst, std::fabs(ref - C[(size_t)(M - 1) * N + c]));
        }
        printf("      max |c - float64| over 4000 samples: %.3g (max |c| %.3g)\n", worst, scale);
        CHECK(hipMemset(dC, 0, (size_t)M * N * 4));
    };
    if (mode == 0) {
        report(run<128, 128, 2, 3, 0>(a, 50));
        report(run<128, 128, 1, 4, 0>(a, 50));
        report(run<128, 64, 2, 2, 0>(a, 50));
        report(run<128, 64, 1, 3, 0>(a, 50));
        report(run<64, 128, 1, 3, 0>(a, 50));
        report(run<64, 64, 2, 3, 0>(a, 50));
    } else {
        report(run<128, 128, 2, 3, 1>(a, 50));
        report(run<128, 128, 1, 4, 1>(a, 50));
        report(run<128, 64, 2, 2, 1>(a, 50));
        report(run<128, 64, 1, 3, 1>(a, 50));
        report(run<64, 128, 1, 3, 1>(a, 50));
        report(run<64, 64, 2, 3, 1>(a, 50));
    }
    return 0;
}
