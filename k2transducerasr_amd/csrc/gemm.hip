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
//   - loads are branch-free: out-of-range rows/columns are clamped to a valid row
//     (their results are never stored), the K tail is clamped + zero-selected
//   - epilogue straight from the accumulators: bias, activation (hardware exp/log),
//     residual add
//   - MODE_CONV: implicit-conv row/k mapping for A over an NHWC tensor;
//     MODE_WKN: [K,N] B operand (transposed while staging) for attention-weights x
//     values products.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <map>
#include <mutex>
#include <type_traits>

#include "kernels.h"

// The LDS-DMA inline asm of the pipelined kernels writes M0 (the LDS destination) and SCC (s_add_u32) and says so in its clobber
// list.  M0 is a reserved register for the compiler, which is why clang remarks on seeing it there; the declaration is still what
// keeps the backend from carrying an M0 / SCC value of its own across the statement.
#pragma clang diagnostic ignored "-Winline-asm"

namespace k2hip {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

// K step BK (32 or 64) is a template parameter; LDS rows are padded to BK+4 floats (16-byte
// aligned; 36- and 68-dword strides are both conflict-free for the ds_read_b128 fragment reads)
enum { MODE_PLAIN = 0, MODE_CONV = 1, MODE_WKN = 2 };

// Activations with the hardware transcendental units (v_exp_f32 / v_log_f32, ~1e-6
// relative): the accurate libm forms cost ~100 VALU instructions per element, which
// for K <= 256 made the epilogue longer than the MFMA loop.
__device__ __forceinline__ float fast_softplus(float z) {  // log(1 + e^z)
    return z > 15.f ? z : __logf(1.0f + __expf(z));
}
__device__ __forceinline__ float apply_act(float v, int act) {
    switch (act) {
        case ACT_SWOOSH_L: return fast_softplus(v - 4.0f) - 0.08f * v - 0.035f;
        case ACT_SWOOSH_R: return fast_softplus(v - 1.0f) - 0.08f * v - 0.313261687f;
        case ACT_TANH: {
            float e = __expf(-2.0f * fabsf(v));
            float t = (1.0f - e) / (1.0f + e);
            return v < 0.f ? -t : t;
        }
        case ACT_SIGMOID: return 1.0f / (1.0f + __expf(-v));
        case ACT_RELU: return fmaxf(v, 0.f);
        case ACT_DOUBLE_SWISH: return v / (1.0f + __expf(1.0f - v));  // v * sigmoid(v - 1)
        default: return v;
    }
}

// component-wise select (`ok ? v : zero4` on a float4 is lowered through scratch memory by
// hipcc).  Only the K-tail tile uses it: any VALU touching a loaded value makes hipcc wait
// for the load right there, i.e. BEFORE the MFMAs it is supposed to overlap, so the loads
// of full tiles are left completely untouched until the LDS write of the next iteration.
__device__ __forceinline__ float4 sel4(bool ok, float4 v) {
    return make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
}

// ---- shared epilogue ------------------------------------------------------------------------------------------------------
// One lane's NV accumulator values of output column `col`, rows row0 + ROWS::off(n).  v = act(acc + bias) (+ residual) [* mul]
// [bypass] -> C.  Written so that the NV stores issue BACK TO BACK: every optional per-element load (residual when it was not
// prefetched, mul, the bypass original) is batched in front of the arithmetic, the activation is chosen by ONE wave-uniform switch
// around the element loop, and nothing loads between two stores.  (With the loads and a per-element activation switch inside the
// element loop the compiler drains vmcnt(0) in front of every element, so each store waited for the previous store's round trip:
// ~500 cycles x 16 per 32x32 block -- 8.6k of the 50k cycles a 128x64 tile with K = 512 takes; in-kernel stamps,
// tools/gemm_dma_trace.py.)
struct Rows32 {  // C/D layout of v_mfma_f32_32x32x2_f32: register r of lane (li, lh) is row (r & 3) + 8 (r >> 2) + 4 lh
    static __device__ __forceinline__ int off(int r) { return (r & 3) + 8 * (r >> 2); }
};
struct Rows16 {  // v_mfma_f32_16x16x4_f32: register e of lane (r, q) is row 4 q + e
    static __device__ __forceinline__ int off(int e) { return e; }
};
// residual: prefetched by the caller (rres, when use_rres) or loaded here from R
template <int NV, typename ROWS>
__device__ __forceinline__ void epilogue_rows(const GemmArgs& g, float (&v)[NV], int row0, int col, float* __restrict__ C,
                                              const float* __restrict__ R, const float (&rres)[NV], bool use_rres,
                                              const float* __restrict__ mulp, const float* __restrict__ biasp, bool bias_loaded = false,
                                              float bias_value = 0.f) {
    if (col >= g.N) return;
    // (bias_loaded: the caller fetched this column's bias before its K loop -- the load's latency is then not part of the epilogue)
    const float bv = bias_loaded ? bias_value : (biasp ? biasp[col] : 0.f);
    const bool act_on = g.act_cols == 0 || col < g.act_cols;
    float rv[NV], mv[NV], ov[NV];
    float bs = 0.f;
    const bool has_res = R != nullptr || use_rres;
    if (use_rres) {
#pragma unroll
        for (int n = 0; n < NV; n++) rv[n] = rres[n];
    } else if (R) {
#pragma unroll
        for (int n = 0; n < NV; n++) {
            const int row = min(row0 + ROWS::off(n), g.M - 1);
            rv[n] = R[(long long)(g.res_div > 1 ? row / g.res_div : row) * g.ldr + col];
        }
    }
    if (mulp) {
#pragma unroll
        for (int n = 0; n < NV; n++) mv[n] = mulp[(long long)min(row0 + ROWS::off(n), g.M - 1) * g.ldm + col];
    }
    if (g.byp_orig) {
        bs = g.byp_scale[col];
#pragma unroll
        for (int n = 0; n < NV; n++) ov[n] = g.byp_orig[(long long)min(row0 + ROWS::off(n), g.M - 1) * g.ld_orig + col];
    }
#pragma unroll
    for (int n = 0; n < NV; n++) v[n] += bv;
    if (g.glu) {
        // The first glu_cols columns (all N when 0) come in blocks of 32 = 16 values | their 16 gates (weights interleaved at load):
        // the partner is 16 lanes away (col = lane & 31 in these layouts); value * sigmoid(gate) (1) or value * tanh(gate) (2) goes
        // to column 16 q + p of C.  Columns from glu_cols on pass through to C columns glu_cols / 2 + (col - glu_cols).
        const int gc = g.glu_cols > 0 ? g.glu_cols : g.N;
        const bool paired = col < gc;  // uniform over a 32-column block (gc % 32 == 0)
        const bool is_value = (col & 16) == 0;
        const int ccol = paired ? ((col >> 5) << 4) + (col & 15) : (gc >> 1) + (col - gc);
        float* cp = C + (long long)row0 * g.ldc + ccol;
#pragma unroll
        for (int n = 0; n < NV; n++) {
            const float other = __shfl_xor(v[n], 16, 64);
            const float gate = g.glu == 2 ? apply_act(other, ACT_TANH) : 1.0f / (1.0f + __expf(-other));
            v[n] = paired ? v[n] * gate : v[n];
        }
        if ((paired && !is_value) || (g.ablate & 4)) return;
#pragma unroll
        for (int n = 0; n < NV; n++)
            if (row0 + ROWS::off(n) < g.M) cp[(long long)ROWS::off(n) * g.ldc] = v[n];
        return;
    }
    if (g.act_after_res && has_res) {
#pragma unroll
        for (int n = 0; n < NV; n++) v[n] += rv[n];
    }
    switch (g.act) {  // wave-uniform; the lanes of columns >= act_cols keep their value
#define K2_ACT_CASE(A_)                                                        \
    case A_:                                                                   \
        _Pragma("unroll") for (int n = 0; n < NV; n++) {                       \
            const float t = apply_act(v[n], A_);                               \
            v[n] = act_on ? t : v[n];                                          \
        }                                                                      \
        break;
        K2_ACT_CASE(ACT_SWOOSH_L)
        K2_ACT_CASE(ACT_SWOOSH_R)
        K2_ACT_CASE(ACT_TANH)
        K2_ACT_CASE(ACT_SIGMOID)
        K2_ACT_CASE(ACT_RELU)
        K2_ACT_CASE(ACT_DOUBLE_SWISH)
#undef K2_ACT_CASE
        default: break;
    }
    if (!g.act_after_res && has_res) {
#pragma unroll
        for (int n = 0; n < NV; n++) v[n] += rv[n];
    }
    if (mulp) {
#pragma unroll
        for (int n = 0; n < NV; n++) v[n] *= mv[n];
    }
    if (g.byp_orig) {
#pragma unroll
        for (int n = 0; n < NV; n++) v[n] = ov[n] + (v[n] - ov[n]) * bs;
    }
    if (g.ablate & 4) return;
    // address = (uniform 64-bit base + the row's uniform byte offset) + a 32-bit per-lane offset: the row offsets are scalar work and
    // the store can take its SGPR-base form; when the lane's last row is inside M (all but the bottom tiles) no row is tested
    const unsigned lane_off = (unsigned)(((long long)row0 * g.ldc + col) * 4);  // < 2^32: checked by the launchers that matter (M * ldc < 2^29)
    const bool small = (long long)g.M * g.ldc < (1ll << 29);
    if (small && row0 + ROWS::off(NV - 1) < g.M) {
#pragma unroll
        for (int n = 0; n < NV; n++) {
            char* ub = reinterpret_cast<char*>(C) + (long long)ROWS::off(n) * g.ldc * 4;
            *reinterpret_cast<float*>(ub + lane_off) = v[n];
        }
        return;
    }
    float* cp = C + (long long)row0 * g.ldc + col;
#pragma unroll
    for (int n = 0; n < NV; n++)
        if (row0 + ROWS::off(n) < g.M) cp[(long long)ROWS::off(n) * g.ldc] = v[n];
}

// XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs (b and b+8 share an L2), so the natural x-fastest order
// makes every XCD read all of A and all of W.  Remap so that each XCD owns a contiguous run of a linear tile order in which runs are
// compact blocks: the N tiles are cut into PN panels (1, 2, 4 or 8), a panel is walked row-major, panels follow each other.  PN = 1 is a
// band of M per XCD (its L2 holds A / 8 + W: the right cut while the rows outweigh the columns); with PN panels an XCD holds ~A PN / 8 +
// W / PN, and the cut that minimises the fabric traffic PN x A + (8 / PN) x W -- PN M + (8 / PN) N in rows and columns, K cancels -- is
// taken (round 4: the wide feed-forward GEMMs of the 6.25 / 12.5 Hz stacks used to pull W, the larger operand there, once per XCD).
// Any (M tiles, N tiles, PN) is a bijection: run boundaries need not coincide with panel boundaries.
__device__ __forceinline__ void xcd_tile(int& mb, int& nb, int M = 0, int N = 0) {
    const int gx = gridDim.x, gy = gridDim.y, nwg = gx * gy;
    if (gridDim.z != 1 || nwg < 16) {  // batched launches keep the natural order
        mb = blockIdx.y;
        nb = blockIdx.x;
        return;
    }
    const int orig = blockIdx.x + blockIdx.y * gx;
    const int xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
    int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    int pn = 1;
    // (launches of more than a round of the chip only: on the streaming tick's GEMMs of 100 - 250 workgroups the panels cost 0.06 ms per
    // tick -- there the next launch reads what this one's XCD just wrote, band by band)
    if (M > 0 && N > 0 && nwg >= 320) {
        long long best = (long long)M + 8ll * N;
        const long long worth = best - best / 8;   // (a cut has to save an eighth of the band form's traffic to be taken)
#pragma unroll
        for (int c = 2; c <= 8; c *= 2) {
            const long long cost = (long long)c * M + (8ll / c) * N;
            if (cost < best && cost <= worth && gx >= c) {
                best = cost;
                pn = c;
            }
        }
    }
    // panel p holds columns [n0, n0 + w): widths gx / pn, the first gx % pn panels one wider
    const int wq = gx / pn, wr = gx % pn;
    int n0 = 0, w = wq + (wr > 0);
    for (int p = 0; p < pn - 1 && t >= gy * w; p++) {
        t -= gy * w;
        n0 += w;
        w = wq + (p + 1 < wr);
    }
    mb = t / w;
    nb = n0 + (t - mb * w);
}

template <int BM, int BN, int WM, int WN, int BK, int MODE>
__global__ __launch_bounds__(64 * (BM / WM) * (BN / WN)) void gemm_f32_mfma(GemmArgs g) {
    constexpr int LDSK = BK + 4;
    constexpr int F4R = BK / 4;  // float4 per tile row
    constexpr int MT = WM / 32, NT = WN / 32;
    constexpr int WCOLS = BN / WN;
    constexpr int NTHR = 64 * (BM / WM) * (BN / WN);
    constexpr int RPP = NTHR / F4R;  // tile rows covered by one pass of float4 loads
    static_assert(BM % RPP == 0 && BN % RPP == 0, "tile must be a whole number of load passes");
    constexpr int PA = BM / RPP;  // float4 loads per thread for the A tile
    constexpr int PB = BN / RPP;

    if (g.skip_if_zero && *g.skip_if_zero == 0) return;  // uniform: every wave of every workgroup leaves
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
    const float* __restrict__ W = g.W + (MODE == MODE_WKN && g.wz_map ? (long long)g.wz_map[z0] : (long long)z0) * g.sW0 + z1 * g.sW1;
    float* __restrict__ C = g.C + z0 * g.sC0 + z1 * g.sC1;
    const float* __restrict__ R = g.res ? g.res + z0 * g.sR0 + z1 * g.sR1 : nullptr;

    int mb_, nb_;
    xcd_tile(mb_, nb_, g.xcd_panels == 1 ? 0 : g.M, g.N);
    const int m0 = mb_ * BM, n0 = nb_ * BN;
    const int Kp = (g.K + 3) & ~3;

    // ---- per-thread global-load coordinates (rows / columns clamped into range)
    const int ld_row = tid / F4R;        // 0..RPP-1
    const int ld_kq = (tid % F4R) * 4;   // k offset of this thread's float4
    const float* a_ptr[PA];
#pragma unroll
    for (int p = 0; p < PA; p++) {
        int row = min(m0 + ld_row + RPP * p, g.M - 1);
        if (MODE == MODE_CONV) {
            int f = row % g.cv_Fout;
            int bt = row / g.cv_Fout;
            int t = bt % g.cv_Tout, b = bt / g.cv_Tout;
            a_ptr[p] = A + (((long long)b * g.cv_Tin + (long long)t * g.cv_st) * g.cv_Fin + (long long)f * g.cv_sf) * g.cv_C;
        } else {
            a_ptr[p] = A + (long long)row * g.lda;
        }
    }
    constexpr int NQ = BN / 4;           // MODE_WKN: float4 per k row
    constexpr int KR = NTHR / NQ;        //           k rows per pass
    constexpr int PBK = MODE == MODE_WKN ? BK / KR : PB;  // loads per thread for the B tile
    static_assert(MODE != MODE_WKN || (KR <= BK && BK % KR == 0), "[K,N] staging needs KR | BK");
    const float* w_ptr[PBK];
#pragma unroll
    for (int p = 0; p < PBK; p++) {
        if (MODE == MODE_WKN) {
            int col = min(n0 + (tid % NQ) * 4, g.N - 4);
            w_ptr[p] = W + col;
        } else {
            int col = min(n0 + ld_row + RPP * p, g.N - 1);
            w_ptr[p] = W + (long long)col * g.ldw;
        }
    }
    // Staging registers.  Plain loops over constexpr bounds, no lambdas: captured arrays
    // were placed in scratch memory by hipcc (16 scratch_load/store_dwordx4 per K step).
    float4 ra[PA], rb[PBK];

    // full K tile: unconditional loads, nothing consumes them before the next LDS write
#define K2_LOAD_TILE(k0_)                                                                                         \
    {                                                                                                             \
        const int k = (k0_) + ld_kq;                                                                              \
        long long koff = k;                                                                                       \
        if (MODE == MODE_CONV) koff = (long long)(k / g.seg_len) * g.seg_stride + (k % g.seg_len);                \
        _Pragma("unroll") for (int p = 0; p < PA; p++) ra[p] = *reinterpret_cast<const float4*>(a_ptr[p] + koff); \
        if (MODE != MODE_WKN) {                                                                                   \
            _Pragma("unroll") for (int p = 0; p < PB; p++) rb[p] = *reinterpret_cast<const float4*>(w_ptr[p] + k); \
        } else {                                                                                                  \
            _Pragma("unroll") for (int p = 0; p < PBK; p++) {                                                     \
                const int kk = (k0_) + tid / NQ + KR * p;                                                         \
                rb[p] = *reinterpret_cast<const float4*>(w_ptr[p] + (long long)kk * g.ldw);                       \
            }                                                                                                     \
        }                                                                                                         \
    }
    // last, partial K tile: clamp the address, zero what lies beyond K
#define K2_LOAD_TAIL(k0_)                                                                                         \
    {                                                                                                             \
        const int k = (k0_) + ld_kq;                                                                              \
        const bool kok = k < Kp;                                                                                  \
        const int kc = kok ? k : Kp - 4;                                                                          \
        long long koff = kc;                                                                                      \
        if (MODE == MODE_CONV) koff = (long long)(kc / g.seg_len) * g.seg_stride + (kc % g.seg_len);              \
        _Pragma("unroll") for (int p = 0; p < PA; p++)                                                            \
            ra[p] = sel4(kok, *reinterpret_cast<const float4*>(a_ptr[p] + koff));                                 \
        if (MODE != MODE_WKN) {                                                                                   \
            _Pragma("unroll") for (int p = 0; p < PB; p++)                                                        \
                rb[p] = sel4(kok, *reinterpret_cast<const float4*>(w_ptr[p] + kc));                               \
        } else {                                                                                                  \
            _Pragma("unroll") for (int p = 0; p < PBK; p++) {                                                     \
                const int kk = (k0_) + tid / NQ + KR * p;                                                         \
                const bool ok = kk < g.K;                                                                         \
                rb[p] = sel4(ok, *reinterpret_cast<const float4*>(w_ptr[p] + (long long)(ok ? kk : g.K - 1) * g.ldw)); \
            }                                                                                                     \
        }                                                                                                         \
    }

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; i++)
#pragma unroll
        for (int j = 0; j < NT; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

    // residual operand: fetched now, consumed in the epilogue, so its latency (it was written
    // by the previous kernel and usually sits in L2/MALL) hides behind the whole K loop
    float rres[MT][NT][16];
    if (R) {
#pragma unroll
        for (int j = 0; j < NT; j++) {
            const int col = min(n0 + wc * WN + j * 32 + li, g.N - 1);
#pragma unroll
            for (int i = 0; i < MT; i++)
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int row = min(m0 + wr * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh, g.M - 1);
                    rres[i][j][r] = R[(long long)(g.res_div > 1 ? row / g.res_div : row) * g.ldr + col];
                }
        }
    }

    const int nk = (g.K + BK - 1) / BK;
    const int nfull = g.K / BK;  // tiles [0, nfull) are complete
    if (nfull > 0) K2_LOAD_TILE(0) else K2_LOAD_TAIL(0)
    for (int kt = 0; kt < nk; kt++) {
        // registers -> LDS buffer kt&1.  The buffer was last read in iteration kt-2's
        // MFMA section, which every wave left before passing iteration kt-1's barrier.
        {
            float* a = sA + (kt & 1) * BM * LDSK;
            float* b = sB + (kt & 1) * BN * LDSK;
#pragma unroll
            for (int p = 0; p < PA; p++) *reinterpret_cast<float4*>(a + (ld_row + RPP * p) * LDSK + ld_kq) = ra[p];
            if (MODE != MODE_WKN) {
#pragma unroll
                for (int p = 0; p < PB; p++) *reinterpret_cast<float4*>(b + (ld_row + RPP * p) * LDSK + ld_kq) = rb[p];
            } else {
#pragma unroll
                for (int p = 0; p < PBK; p++) {
                    const int kk = tid / NQ + KR * p;
                    const int c = (tid % NQ) * 4;
                    // a clamped column group (n0 + c > N - 4) re-loads valid data; its outputs are never stored
                    b[(c + 0) * LDSK + kk] = rb[p].x;
                    b[(c + 1) * LDSK + kk] = rb[p].y;
                    b[(c + 2) * LDSK + kk] = rb[p].z;
                    b[(c + 3) * LDSK + kk] = rb[p].w;
                }
            }
        }
        __syncthreads();
        if (!(g.ablate & 1)) {  // next tile's loads stay in flight during the MFMAs below
            if (kt + 1 < nfull) K2_LOAD_TILE((kt + 1) * BK)
            else if (kt + 1 < nk) K2_LOAD_TAIL((kt + 1) * BK)
        }
        const float* a = sA + (kt & 1) * BM * LDSK + (wr * WM + li) * LDSK + 4 * lh;
        const float* b = sB + (kt & 1) * BN * LDSK + (wc * WN + li) * LDSK + 4 * lh;
        if (!(g.ablate & 2))
#pragma unroll
        for (int gk = 0; gk < BK / 8; gk++) {
            float4 fa[MT], fb[NT];
#pragma unroll
            for (int i = 0; i < MT; i++) fa[i] = *reinterpret_cast<const float4*>(a + i * 32 * LDSK + gk * 8);
#pragma unroll
            for (int j = 0; j < NT; j++) fb[j] = *reinterpret_cast<const float4*>(b + j * 32 * LDSK + gk * 8);
            // k-step outer, accumulator tile inner: consecutive MFMAs hit different accumulators
#pragma unroll
            for (int e = 0; e < 4; e++)
#pragma unroll
                for (int i = 0; i < MT; i++)
#pragma unroll
                    for (int j = 0; j < NT; j++) {
                        const float av = e == 0 ? fa[i].x : e == 1 ? fa[i].y : e == 2 ? fa[i].z : fa[i].w;
                        const float bv = e == 0 ? fb[j].x : e == 1 ? fb[j].y : e == 2 ? fb[j].z : fb[j].w;
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i][j], 0, 0, 0);
                    }
        }
    }
#undef K2_LOAD_TILE
#undef K2_LOAD_TAIL

    // ---- epilogue: C/D layout of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
    for (int j = 0; j < NT; j++)
#pragma unroll
        for (int i = 0; i < MT; i++) {
            float vals[16];
#pragma unroll
            for (int r = 0; r < 16; r++) vals[r] = acc[i][j][r];
            epilogue_rows<16, Rows32>(g, vals, m0 + wr * WM + i * 32 + 4 * lh, n0 + wc * WN + j * 32 + li, C, nullptr, rres[i][j], R != nullptr,
                                      g.mul ? g.mul + z0 * g.sM0 + z1 * g.sM1 : nullptr, g.bias ? g.bias + z0 * g.sBias0 : nullptr);
        }
}

// ---------------------------------------------------------------------------------------
// LDS-DMA variant (plain Linear, K % 32 == 0): global -> LDS directly with
// global_load_lds_dwordx4 (no staging VGPRs, no ds_write pass), THREE LDS stages, a counted
// s_waitcnt vmcnt and ONE raw s_barrier per K step, so the DMA of tile kt+2 is in flight
// across the barrier while tile kt feeds the MFMAs.
//   * the LDS image of a stage is lane-linear (a wave instruction writes 64 x 16 B = 8 rows of
//     32 floats); bank conflicts of the 128-byte rows are removed by swizzling on the SOURCE
//     side: slot (row r, chunk c') receives logical chunk c = c' ^ ((r >> 1) & 7), and the
//     fragment reads apply the same XOR.
//   * rows beyond M / N are clamped to the last valid row (never stored).
template <int BM, int BN, int WM, int WN, int NST = 3>
__global__ __launch_bounds__(64 * (BM / WM) * (BN / WN)) void gemm_f32_mfma_dma(GemmArgs g) {
    constexpr int BK = 32, PF = NST - 1;  // PF stages are in flight ahead of the one being multiplied
    constexpr int MT = WM / 32, NT = WN / 32;
    constexpr int WCOLS = BN / WN;
    constexpr int NW = (BM / WM) * (BN / WN);
    constexpr int NINST = (BM + BN) / 8;       // 1 KB wave-instructions per tile
    constexpr int IPW = (NINST + NW - 1) / NW;  // per wave (the last round may be short: wave w issues inst w + q*NW < NINST)
    constexpr int NFULL = NINST % NW == 0 ? NW : NINST % NW;  // waves 0..NFULL-1 issue IPW instructions, the others IPW - 1
    constexpr int STAGE = (BM + BN) * BK;      // floats per stage

    if (g.skip_if_zero && *g.skip_if_zero == 0) return;
    extern __shared__ __attribute__((aligned(16))) float smem[];  // [NST][STAGE]: A rows then W rows

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WCOLS, wc = wave % WCOLS;
    const int li = lane & 31, lh = lane >> 5;
    // blockIdx.z = z0 + nb0 * z1: batched launches (the LSTM layer wavefront: one z per layer, every operand at its own stride)
    const int z0 = blockIdx.z % g.nb0, z1 = blockIdx.z / g.nb0;
    const float* __restrict__ A = g.A + z0 * g.sA0 + z1 * g.sA1;
    const float* __restrict__ W = g.W + z0 * g.sW0 + z1 * g.sW1;
    float* __restrict__ C = g.C + z0 * g.sC0 + z1 * g.sC1;
    const float* __restrict__ R = g.res ? g.res + z0 * g.sR0 + z1 * g.sR1 : nullptr;
    int mb_, nb_;
    xcd_tile(mb_, nb_, g.xcd_panels == 1 ? 0 : g.M, g.N);
    const int m0 = mb_ * BM, n0 = nb_ * BN;
    // tuning only (g.dbg != nullptr): lane 0 of every wave stamps s_memtime: entry | prologue issued | per K step: barrier passed
    // (first 40) | loop done | end; [62] = stamps, [60]/[63] = s_memrealtime
    unsigned long long* stamp = g.dbg ? g.dbg + ((size_t)(blockIdx.x + blockIdx.y * gridDim.x) * NW + wave) * 64 : nullptr;
    int nstamp = 0;
#define K2_DMA_STAMP() \
    if (stamp && lane == 0 && nstamp < 60) stamp[nstamp++] = __builtin_amdgcn_s_memtime();
    if (stamp && lane == 0) stamp[60] = __builtin_amdgcn_s_memrealtime();
    K2_DMA_STAMP()

    // this lane's source pointer for each of the wave's DMA instructions (advances by BK per tile)
    const float* src[IPW];
#pragma unroll
    for (int q = 0; q < IPW; q++) {
        const int inst = min(wave + q * NW, NINST - 1);  // wave-uniform (clamped: the pointer of a missing slot is unused)
        const int slot = inst * 64 + lane;       // 16-byte slot within the stage
        const int r = slot >> 3, cp = slot & 7;  // row of the combined [A;W] tile, physical chunk
        const int c = cp ^ ((r >> 1) & 7);       // logical k chunk this slot holds
        if (inst < BM / 8) src[q] = A + (long long)min(m0 + r, g.M - 1) * g.lda + 4 * c;
        else src[q] = W + (long long)min(n0 + (r - BM), g.N - 1) * g.ldw + 4 * c;
    }
    // The DMA is issued through inline asm: with the builtin, hipcc treats it as a pending LDS
    // write and drains vmcnt(0) in front of every ds_read, which serialises the pipeline.  In asm
    // the instruction is invisible to the compiler's wait-count bookkeeping; completion is
    // tracked by the hand-counted s_waitcnt vmcnt below (M0 = wave-uniform LDS byte address of the
    // 1 KB destination; written and restored inside the same statement).
    const unsigned lds_base = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)smem;
    auto issue = [&](int kt) {
#pragma unroll
        for (int q = 0; q < IPW; q++) {
            const int inst = wave + q * NW;
            if (inst >= NINST) break;  // wave-uniform
            const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + ((kt % NST) * STAGE + inst * 256) * 4);
            const float* gp = src[q] + kt * BK;
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep)
                         : "v"(gp), "s"(dst)
                         : "memory");
        }
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; i++)
#pragma unroll
        for (int j = 0; j < NT; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

    const int nk = g.K / BK;
    // fragment addressing: row-dependent XOR (rows of one lane differ by multiples of 32 -> same XOR)
    const int arow = wr * WM + li, brow = wc * WN + li;
    const int swa = (arow >> 1) & 7, swb = ((BM + brow) >> 1) & 7;

#pragma unroll
    for (int p = 0; p < PF; p++)
        if (p < nk) issue(p);
    // residual operand: fetched behind the prologue's DMAs (so that the two latencies overlap), consumed in the epilogue.  vmcnt
    // retires in issue order: while these NRES loads are younger than the tile being waited for (kt < PF) they are allowed to stay
    // outstanding on top of the newer tiles' DMAs; the asm memory clobbers on both sides keep the loads where they are written.
    constexpr int NRES = 16 * MT * NT;
    float rres[MT][NT][16];
    if (R) {
#pragma unroll
        for (int j = 0; j < NT; j++) {
            const int col = min(n0 + wc * WN + j * 32 + li, g.N - 1);
#pragma unroll
            for (int i = 0; i < MT; i++)
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int row = min(m0 + wr * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh, g.M - 1);
                    rres[i][j][r] = R[(long long)row * g.ldr + col];
                }
        }
    }
    K2_DMA_STAMP()
    for (int kt = 0; kt < nk; kt++) {
        // tile kt landed for this wave once at most the PF-1 newer tiles' instructions are pending
        if (R && kt < PF && kt + PF - 1 < nk) {
            constexpr int X = (PF - 1) * IPW + NRES < 63 ? (PF - 1) * IPW + NRES : 63;
            constexpr int Y = (PF - 1) * (IPW - 1) + NRES < 63 ? (PF - 1) * (IPW - 1) + NRES : 63;
            if (wave < NFULL) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(X) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(Y) : "memory");
        } else if (kt + PF - 1 < nk && (PF - 1) * (IPW - 1) > 0) {
            if (wave < NFULL) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((PF - 1) * IPW) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((PF - 1) * (IPW - 1)) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();  // ... and for every wave; also: everyone is done reading the stage about to be refilled
        if (kt < 40) { K2_DMA_STAMP() }
        if (kt + PF < nk) issue(kt + PF);
        const float* sa = smem + (kt % NST) * STAGE + arow * BK;
        const float* sb = smem + (kt % NST) * STAGE + (BM + brow) * BK;
        // one-group look-ahead: the reads of 8-k group gk+1 are issued between the first and the second MFMA of group gk, so
        // they land while the (dependent, 64-cycle) MFMA chain of gk runs instead of after it.  sched_barrier(0) pins the
        // order (the scheduler otherwise sinks the reads back next to their first use).
        float4 fa[2][MT], fb[2][NT];
#pragma unroll
        for (int i = 0; i < MT; i++) fa[0][i] = *reinterpret_cast<const float4*>(sa + i * 32 * BK + (((0 + lh) ^ swa) << 2));
#pragma unroll
        for (int j = 0; j < NT; j++) fb[0][j] = *reinterpret_cast<const float4*>(sb + j * 32 * BK + (((0 + lh) ^ swb) << 2));
#pragma unroll
        for (int gk = 0; gk < 4; gk++) {
            const int cu = gk & 1, nx = cu ^ 1;
#pragma unroll
            for (int e = 0; e < 4; e++) {
#pragma unroll
                for (int i = 0; i < MT; i++)
#pragma unroll
                    for (int j = 0; j < NT; j++) {
                        const float av = e == 0 ? fa[cu][i].x : e == 1 ? fa[cu][i].y : e == 2 ? fa[cu][i].z : fa[cu][i].w;
                        const float bv = e == 0 ? fb[cu][j].x : e == 1 ? fb[cu][j].y : e == 2 ? fb[cu][j].z : fb[cu][j].w;
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i][j], 0, 0, 0);
                    }
                if (e == 0 && gk < 3) {
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < MT; i++)
                        fa[nx][i] = *reinterpret_cast<const float4*>(sa + i * 32 * BK + (((2 * (gk + 1) + lh) ^ swa) << 2));
#pragma unroll
                    for (int j = 0; j < NT; j++)
                        fb[nx][j] = *reinterpret_cast<const float4*>(sb + j * 32 * BK + (((2 * (gk + 1) + lh) ^ swb) << 2));
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    }

    K2_DMA_STAMP()
#pragma unroll
    for (int j = 0; j < NT; j++)
#pragma unroll
        for (int i = 0; i < MT; i++) {
            float vals[16];
#pragma unroll
            for (int r = 0; r < 16; r++) vals[r] = acc[i][j][r];
            epilogue_rows<16, Rows32>(g, vals, m0 + wr * WM + i * 32 + 4 * lh, n0 + wc * WN + j * 32 + li, C, nullptr, rres[i][j], R != nullptr,
                                      nullptr, g.bias ? g.bias + z0 * g.sBias0 : nullptr);
        }
    K2_DMA_STAMP()
    if (stamp && lane == 0) {
        stamp[62] = nstamp;
        stamp[63] = __builtin_amdgcn_s_memrealtime();
    }
#undef K2_DMA_STAMP
}

// ---------------------------------------------------------------------------------------
// Pipelined variant of the LDS-DMA kernel (plain Linear, K % 32 == 0): the same tile image, swizzle and fragment reads, with a
// K loop that contains nothing but MFMAs, LDS reads, LDS-DMAs and scalar bookkeeping.
//
// What the measurements said (tools/probes/vmem_vs_mfma_probe.hip, tools/gemm_dma_trace.py):
//   * a wave streaming v_mfma_f32_32x32x2_f32 keeps the 64-cycle rate with six LDS-DMAs, global loads or ds_read_b128 per 16 MFMAs
//     in its instruction stream -- they cost nothing -- but every VECTOR-ALU instruction between two MFMAs costs matrix time: a
//     per-lane 64-bit address add in front of each DMA took the loop from 64.1 to 71 - 74 cycles per MFMA, saving / restoring M0
//     around it another 3;
//   * the compiled K loop of gemm_f32_mfma_dma carries ~30 such instructions per K step and wave (pointer + kt * 32, the XOR
//     swizzle of every fragment read, kt % NST, select chains for wave-uniform conditions) and a dozen branches: 2671 cycles per
//     2048 cycles of MFMA work in its in-kernel stamps, whatever the pipeline order.
// So here every address is settled BEFORE the loop:
//   * fragment reads: one LDS byte address per (stage, 8-deep k group) and operand in VGPRs (2 x 4 NST registers), the 32-row
//     block of the wave tile in the instruction's immediate offset;
//   * LDS-DMA: global_load_lds_dwordx4 in its SGPR-base + 32-bit VGPR-offset form -- the per-lane offset (row * ld + swizzled
//     chunk) never changes, the base advances by 128 bytes per K step with scalar adds; M0 = a per-instruction SGPR + the stage's
//     constant;
//   * the K loop is unrolled by NST so that the stage is a compile-time constant; neither its steady state nor the straight-line
//     copies of the last steps test anything (K >= 32 (NST - 1): the launcher checks);
//   * pipeline: fragments double-buffered per k group, the last group of a step reads the first group of the next one; the wait
//     for step kt+1's DMA and the step's one barrier sit between groups 2 and 3, behind queued MFMAs; the DMA of step kt+NST-1
//     is issued behind that barrier, spread over group 3's MFMA slots.
// Wave tiles of 32 MT x 32 NT; NINST % waves == 0 (every wave issues the same number of DMAs, so the counted waits are constants).
// The per-lane byte offsets stay below 2^31: the launcher requires M * lda and N * ldw below 2^29 floats.
template <int BM, int BN, int WM, int WN, int NST>
__global__ __launch_bounds__(64 * (BM / WM) * (BN / WN)) void gemm_f32_mfma_pipe(GemmArgs g) {
    constexpr int BK = 32;
    constexpr int MT = WM / 32, NT = WN / 32;
    constexpr int WCOLS = BN / WN;
    constexpr int NW = (BM / WM) * (BN / WN);
    constexpr int NINST = (BM + BN) / 8;
    constexpr int IPW = NINST / NW;
    constexpr int STAGE = (BM + BN) * BK;
    constexpr int DPE = (IPW + 3) / 4;  // DMA instructions issued behind each of group 3's four MFMA steps
    static_assert(NST == 3 || NST == 4, "pipe: three or four stages");
    static_assert(NINST % NW == 0, "pipe: every wave issues the same number of DMA instructions");
    static_assert((NST - 2) * IPW <= 63 && MT <= 4 && NT <= 2, "pipe: counted wait / wave tile out of range");

    if (g.skip_if_zero && *g.skip_if_zero == 0) return;
    extern __shared__ __attribute__((aligned(16))) float smem[];  // [NST][STAGE]: A rows then W rows

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WCOLS, wc = wave % WCOLS;
    const int li = lane & 31, lh = lane >> 5;
    float* __restrict__ C = g.C;
    const float* __restrict__ R = g.res;
    int mb_, nb_;
    xcd_tile(mb_, nb_, g.xcd_panels == 1 ? 0 : g.M, g.N);
    const int m0 = mb_ * BM, n0 = nb_ * BN;
    const int nk = g.K / BK;

    unsigned long long* stamp = g.dbg ? g.dbg + ((size_t)(blockIdx.x + blockIdx.y * gridDim.x) * NW + wave) * 64 : nullptr;
    int nstamp = 0;
#define K2_PIPE_STAMP()                                                    \
    if (stamp) { /* wave-uniform test and counter: no vector work when off */ \
        if (lane == 0) stamp[nstamp] = __builtin_amdgcn_s_memtime();       \
        nstamp++;                                                          \
    }
    if (stamp && lane == 0) stamp[60] = __builtin_amdgcn_s_memrealtime();
    K2_PIPE_STAMP()

    // ---- addresses, once ----
    const unsigned lds_base = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)smem;
    unsigned voff[IPW];            // per-lane byte offset of this wave's q-th DMA from its operand's base (constant over K)
    unsigned long long sb[IPW];    // that operand's base (wave-uniform: A for instructions < BM / 8, else W)
    unsigned mq[IPW];              // LDS byte address of the instruction's 1 KB inside stage 0
#pragma unroll
    for (int q = 0; q < IPW; q++) {
        const int inst = wave + q * NW;  // wave-uniform
        const int slot = inst * 64 + lane;
        const int r = slot >> 3, cp = slot & 7;
        const int c = cp ^ ((r >> 1) & 7);
        const bool is_a = inst < BM / 8;
        const long long row = is_a ? min(m0 + r, g.M - 1) : min(n0 + (r - BM), g.N - 1);
        voff[q] = (unsigned)((row * (is_a ? g.lda : g.ldw) + 4 * c) * 4);
        const unsigned long long base = (unsigned long long)(is_a ? g.A : g.W);
        // (readfirstlane returns int: without the casts a low half with bit 31 set sign-extends over the high half)
        sb[q] = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(base >> 32)) << 32) |
                (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)base);
        mq[q] = __builtin_amdgcn_readfirstlane(lds_base + inst * 1024);
    }

// instruction q_ of K step KT_ (stage ST_, a literal): M0 = its LDS destination, source = operand base + 128 KT_ bytes + lane offset
#define K2_PIPE_DMA(ST_, KT_, q_)                                                                                          \
    {                                                                                                                      \
        const unsigned long long src_ = sb[q_] + (unsigned long long)(KT_) * (BK * 4);                                    \
        asm volatile("s_add_u32 m0, %2, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"                                    \
                     :                                                                                                     \
                     : "v"(voff[q_]), "s"(src_), "s"(mq[q_]), "n"((ST_) * STAGE * 4)                                       \
                     : "memory", "m0", "scc");                                                                                          \
    }
// fragments of k group G_ of stage ST_ into register set SET_ (the 32-row blocks of the wave tile: 32 * BK floats apart -> the
// instruction's immediate offset)
#define K2_PIPE_READ(SET_, ST_, G_)                                                                                        \
    {                                                                                                                      \
        _Pragma("unroll") for (int i = 0; i < MT; i++) fa[SET_][i] = *reinterpret_cast<const f32x4*>(ra[ST_][G_] + i * 32 * BK); \
        _Pragma("unroll") for (int j = 0; j < NT; j++) fb[SET_][j] = *reinterpret_cast<const f32x4*>(rb[ST_][G_] + j * 32 * BK); \
    }
#define K2_PIPE_FRAGS_READY(SET_) {}
// one K step.  ST_: its stage (literal).  HAS_NEXT_: step KT_+1 exists; DO_ISSUE_: step KT_+NST-1 exists (both `true` in the steady
// state); WAIT_: the vmcnt wait in front of the barrier (instructions of newer steps that may stay in flight)
#define K2_PIPE_STEP(ST_, KT_, HAS_NEXT_, DO_ISSUE_, WAIT_)                                                                \
    _Pragma("unroll") for (int gk = 0; gk < 4; gk++) {                                                                     \
        const int cu = gk & 1;                                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                                                 \
        K2_PIPE_FRAGS_READY(cu)                                                                                            \
        _Pragma("unroll") for (int e = 0; e < 4; e++) {                                                                    \
            _Pragma("unroll") for (int i = 0; i < MT; i++) _Pragma("unroll") for (int j = 0; j < NT; j++)                  \
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cu][i][e], fb[cu][j][e], acc[i][j], 0, 0, 0);          \
            if (e == 0) {                                                                                                  \
                __builtin_amdgcn_sched_barrier(0);                                                                         \
                if (gk == 0) K2_PIPE_READ(1, ST_, 1)                                                                       \
                if (gk == 1) K2_PIPE_READ(0, ST_, 2)                                                                       \
                if (gk == 2) K2_PIPE_READ(1, ST_, 3)                                                                       \
                if (gk == 3 && (HAS_NEXT_)) K2_PIPE_READ(0, ((ST_) + 1) % NST, 0)                                          \
                __builtin_amdgcn_sched_barrier(0);                                                                         \
            }                                                                                                              \
            if (gk == 3 && (DO_ISSUE_)) {                                                                                  \
                __builtin_amdgcn_sched_barrier(0);                                                                         \
                _Pragma("unroll") for (int q = e * DPE; q < (e + 1) * DPE && q < IPW; q++)                                 \
                    K2_PIPE_DMA(((ST_) + NST - 1) % NST, (KT_) + NST - 1, q)                                               \
                __builtin_amdgcn_sched_barrier(0);                                                                         \
            }                                                                                                              \
        }                                                                                                                  \
        if (gk == 2 && (HAS_NEXT_)) { /* step KT_+1 has landed for everyone; everyone has left step KT_-1's stage */       \
            __builtin_amdgcn_sched_barrier(0);                                                                             \
            WAIT_                                                                                                          \
            __builtin_amdgcn_s_barrier();                                                                                  \
            if ((KT_) < 40) { K2_PIPE_STAMP() }                                                                            \
            __builtin_amdgcn_sched_barrier(0);                                                                             \
        }                                                                                                                  \
    }
#define K2_PIPE_WAIT_STEADY asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 3) * IPW) : "memory");
#define K2_PIPE_WAIT_DRAIN asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // ---- prologue: K steps 0 .. NST-2 in flight, step 0 landed, its first fragments on their way ----
#pragma unroll
    for (int p = 0; p < NST - 1; p++)
        if (p < nk) {
#pragma unroll
            for (int q = 0; q < IPW; q++) {
                if (p == 0) K2_PIPE_DMA(0, 0, q)
                if (p == 1) K2_PIPE_DMA(1, 1, q)
                if (p == 2) K2_PIPE_DMA(2, 2, q)
            }
        }
    // (fragment pointers and accumulators only now: the DMAs above are already on their way)
    const int arow = wr * WM + li, brow = wc * WN + li;
    const int swa = (arow >> 1) & 7, swb = ((BM + brow) >> 1) & 7;
    // fragment read pointers: stage, k group.  Plain LDS loads through them (NOT inline-asm ds_reads): the compiler then knows
    // the destination registers are pending until ITS lgkmcnt wait.  With asm reads + a hand-placed wait it was free to copy a
    // fragment register (a phi copy at a loop edge) before the data had arrived -- results changed on rare runs when a second
    // process shared the GPU and LDS returns came late (tools/determinism_stress.py).
    const float* ra[NST][4];
    const float* rb[NST][4];
#pragma unroll
    for (int st = 0; st < NST; st++)
#pragma unroll
        for (int gk = 0; gk < 4; gk++) {
            ra[st][gk] = smem + st * STAGE + arow * BK + (((2 * gk + lh) ^ swa) << 2);
            rb[st][gk] = smem + st * STAGE + (BM + brow) * BK + (((2 * gk + lh) ^ swb) << 2);
        }

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; i++)
#pragma unroll
        for (int j = 0; j < NT; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
    f32x4 fa[2][MT], fb[2][NT];
    K2_PIPE_STAMP()
    if (nk >= NST - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * IPW) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    K2_PIPE_READ(0, 0, 0)
    K2_PIPE_STAMP()
    // Residual and bias are fetched HERE: behind the prologue's DMAs and the first barrier, so that the DMAs leave as early as the
    // kernel can compute their addresses and this block's ~150 address instructions run while the first K steps are in flight
    // (they used to run in front of the first DMA).  vmcnt retires in issue order: these loads are younger than K steps
    // 0 .. NST-2 and older than every later one, so the loop's counted waits also cover them -- conservatively, never wrongly.
    float rres[MT][NT][16];
    if (R) {
#pragma unroll
        for (int j = 0; j < NT; j++) {
            const int col = min(n0 + wc * WN + j * 32 + li, g.N - 1);
#pragma unroll
            for (int i = 0; i < MT; i++)
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int row = min(m0 + wr * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh, g.M - 1);
                    rres[i][j][r] = R[(long long)row * g.ldr + col];
                }
        }
    }
    float bias_pre[NT];
#pragma unroll
    for (int j = 0; j < NT; j++) bias_pre[j] = g.bias ? g.bias[min(n0 + wc * WN + j * 32 + li, g.N - 1)] : 0.f;

    int kt = 0;
    const int nsteady = nk - (NST - 1);  // steps whose step kt+NST-1 exists
    for (; kt + NST <= nsteady; kt += NST) {  // kt % NST == 0 here: the stage of step kt + s is s
        K2_PIPE_STEP(0, kt, true, true, K2_PIPE_WAIT_STEADY)
        K2_PIPE_STEP(1, kt + 1, true, true, K2_PIPE_WAIT_STEADY)
        K2_PIPE_STEP(2, kt + 2, true, true, K2_PIPE_WAIT_STEADY)
        if (NST == 4) { K2_PIPE_STEP(NST == 4 ? 3 : 0, kt + 3, true, true, K2_PIPE_WAIT_STEADY) }
    }
    // What is left starts at a multiple of NST, so the stages stay literals: rs = 0 .. NST-1 more steady steps, then the last NST-1
    // steps (nothing left to issue; the very last has no successor).  One straight-line copy per rs, no flags at run time.
#define K2_PIPE_S(j_) K2_PIPE_STEP((j_) % NST, kt + (j_), true, true, K2_PIPE_WAIT_STEADY)
#define K2_PIPE_T(j_) K2_PIPE_STEP((j_) % NST, kt + (j_), true, false, K2_PIPE_WAIT_DRAIN)
#define K2_PIPE_L(j_) K2_PIPE_STEP((j_) % NST, kt + (j_), false, false, K2_PIPE_WAIT_DRAIN)
    const int rs = nsteady - kt;
    if (NST == 3) {
        switch (rs) {
            case 0: K2_PIPE_T(0) K2_PIPE_L(1) break;
            case 1: K2_PIPE_S(0) K2_PIPE_T(1) K2_PIPE_L(2) break;
            default: K2_PIPE_S(0) K2_PIPE_S(1) K2_PIPE_T(2) K2_PIPE_L(3) break;
        }
    } else {
        switch (rs) {
            case 0: K2_PIPE_T(0) K2_PIPE_T(1) K2_PIPE_L(2) break;
            case 1: K2_PIPE_S(0) K2_PIPE_T(1) K2_PIPE_T(2) K2_PIPE_L(3) break;
            case 2: K2_PIPE_S(0) K2_PIPE_S(1) K2_PIPE_T(2) K2_PIPE_T(3) K2_PIPE_L(4) break;
            default: K2_PIPE_S(0) K2_PIPE_S(1) K2_PIPE_S(2) K2_PIPE_T(3) K2_PIPE_T(4) K2_PIPE_L(5) break;
        }
    }
#undef K2_PIPE_S
#undef K2_PIPE_T
#undef K2_PIPE_L
    K2_PIPE_STAMP()
#undef K2_PIPE_WAIT_DRAIN
#undef K2_PIPE_WAIT_STEADY
#undef K2_PIPE_STEP
#undef K2_PIPE_FRAGS_READY
#undef K2_PIPE_READ
#undef K2_PIPE_DMA

#pragma unroll
    for (int j = 0; j < NT; j++)
#pragma unroll
        for (int i = 0; i < MT; i++) {
            float vals[16];
#pragma unroll
            for (int r = 0; r < 16; r++) vals[r] = acc[i][j][r];
            epilogue_rows<16, Rows32>(g, vals, m0 + wr * WM + i * 32 + 4 * lh, n0 + wc * WN + j * 32 + li, C, nullptr, rres[i][j], R != nullptr,
                                      nullptr, g.bias, true, bias_pre[j]);
        }
    K2_PIPE_STAMP()
    if (stamp && lane == 0) {
        stamp[62] = nstamp;
        stamp[63] = __builtin_amdgcn_s_memrealtime();
    }
#undef K2_PIPE_STAMP
}

// ---------------------------------------------------------------------------------------
// The pipelined kernel on 16x16x4 MFMA tiles (round 5): the same stage image, swizzle, LDS-DMA and vector-ALU-free K loop as
// gemm_f32_mfma_pipe above, with wave tiles of WM x WN = (16 MT) x (16 NT) built from v_mfma_f32_16x16x4_f32 (exact f32 products, the
// same 64 flop / cycle / SIMD as the 32x32x2 shape).  Why a second shape: a 32 x 32 block per wave makes every workgroup tile a multiple
// of 32 x 32 PER WAVE, and the launches that quantise badly on 256 CUs need tiles in between -- 2048 x 768 (the 6.25 Hz stack: 30
// launches, 1.44 ms of the batch) is 192 tiles of 128 x 64 (a quarter of the chip idle) or 256 tiles of 64 x 96, and 64 x 96 in 32 x 32
// blocks is six waves on four SIMDs (two of them carry twice the work: no gain, measured in rounds 1 and 5) while in 16 x 16 blocks it is
// four waves of 32 x 48, one per SIMD, all equal.  Fragments: lane (i = lane & 15, kq = lane >> 4) reads chunk 4 g + kq of row i of a
// 16-k group g as one ds_read_b128 and feeds component e to MFMA step e, for A and for W alike (the k order inside a group is a
// permutation, identical on both sides); rows 16 apart share the XOR swizzle, so a wave tile's blocks are immediate offsets of one
// address per (stage, group), and the 16 lanes of a row block hit all 64 banks once.  C/D: lane (n, q) holds rows 4 q + e of column n
// (epilogue_rows<4, Rows16>; no gated epilogue: that pairs lanes 16 columns apart).
template <int BM, int BN, int WM, int WN, int NST>
__global__ __launch_bounds__(64 * (BM / WM) * (BN / WN)) void gemm_f32_mfma_p16(GemmArgs g) {
    constexpr int BK = 32;
    constexpr int MT = WM / 16, NT = WN / 16;
    constexpr int WCOLS = BN / WN;
    constexpr int NW = (BM / WM) * (BN / WN);
    constexpr int NINST = (BM + BN) / 8;
    constexpr int IPW = NINST / NW;
    constexpr int STAGE = (BM + BN) * BK;
    constexpr int DPE = (IPW + 3) / 4;
    static_assert(NST == 3, "p16: three stages");
    static_assert(NINST % NW == 0, "p16: every wave issues the same number of DMA instructions");
    static_assert(MT >= 1 && NT >= 1 && MT * NT <= 12, "p16: wave tile");

    if (g.skip_if_zero && *g.skip_if_zero == 0) return;
    extern __shared__ __attribute__((aligned(16))) float smem[];  // [NST][STAGE]: A rows then W rows

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WCOLS, wc = wave % WCOLS;
    const int li = lane & 15, kq = lane >> 4;
    float* __restrict__ C = g.C;
    const float* __restrict__ R = g.res;
    int mb_, nb_;
    xcd_tile(mb_, nb_, g.xcd_panels == 1 ? 0 : g.M, g.N);
    const int m0 = mb_ * BM, n0 = nb_ * BN;
    const int nk = g.K / BK;

    const unsigned lds_base = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)smem;
    unsigned voff[IPW];
    unsigned long long sb[IPW];
    unsigned mq[IPW];
#pragma unroll
    for (int q = 0; q < IPW; q++) {
        const int inst = wave + q * NW;  // wave-uniform
        const int slot = inst * 64 + lane;
        const int r = slot >> 3, cp = slot & 7;
        const int c = cp ^ ((r >> 1) & 7);
        const bool is_a = inst < BM / 8;
        const long long row = is_a ? min(m0 + r, g.M - 1) : min(n0 + (r - BM), g.N - 1);
        voff[q] = (unsigned)((row * (is_a ? g.lda : g.ldw) + 4 * c) * 4);
        const unsigned long long base = (unsigned long long)(is_a ? g.A : g.W);
        sb[q] = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(base >> 32)) << 32) |
                (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)base);
        mq[q] = __builtin_amdgcn_readfirstlane(lds_base + inst * 1024);
    }
#define K2_P16_DMA(ST_, KT_, q_)                                                                                           \
    {                                                                                                                      \
        const unsigned long long src_ = sb[q_] + (unsigned long long)(KT_) * (BK * 4);                                    \
        asm volatile("s_add_u32 m0, %2, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"                                    \
                     :                                                                                                     \
                     : "v"(voff[q_]), "s"(src_), "s"(mq[q_]), "n"((ST_) * STAGE * 4)                                       \
                     : "memory", "m0", "scc");                                                                             \
    }
#define K2_P16_READ(SET_, ST_, G_)                                                                                         \
    {                                                                                                                      \
        _Pragma("unroll") for (int i = 0; i < MT; i++) fa[SET_][i] = *reinterpret_cast<const f32x4*>(ra[ST_][G_] + i * 16 * BK); \
        _Pragma("unroll") for (int j = 0; j < NT; j++) fb[SET_][j] = *reinterpret_cast<const f32x4*>(rb[ST_][G_] + j * 16 * BK); \
    }
// one K step = two 16-k groups.  Group 0 out of set 0 (its fragments were read during the previous step's group 1), group 1 out of set 1
// (read behind group 0's first MFMAs); behind group 0: the wait for step KT_+1's DMA + the step's one barrier; behind group 1's first
// MFMAs: the next step's group 0 into set 0; the DMA of step KT_+NST-1 spread over group 1's MFMA slots.
#define K2_P16_STEP(ST_, KT_, HAS_NEXT_, DO_ISSUE_, WAIT_)                                                                 \
    _Pragma("unroll") for (int gk = 0; gk < 2; gk++) {                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                                                 \
        _Pragma("unroll") for (int e = 0; e < 4; e++) {                                                                    \
            _Pragma("unroll") for (int i = 0; i < MT; i++) _Pragma("unroll") for (int j = 0; j < NT; j++)                  \
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[gk][i][e], fb[gk][j][e], acc[i][j], 0, 0, 0);          \
            if (e == 0) {                                                                                                  \
                __builtin_amdgcn_sched_barrier(0);                                                                         \
                if (gk == 0) K2_P16_READ(1, ST_, 1)                                                                        \
                if (gk == 1 && (HAS_NEXT_)) K2_P16_READ(0, ((ST_) + 1) % NST, 0)                                           \
                __builtin_amdgcn_sched_barrier(0);                                                                         \
            }                                                                                                              \
            if (gk == 1 && (DO_ISSUE_)) {                                                                                  \
                __builtin_amdgcn_sched_barrier(0);                                                                         \
                _Pragma("unroll") for (int q = e * DPE; q < (e + 1) * DPE && q < IPW; q++)                                 \
                    K2_P16_DMA(((ST_) + NST - 1) % NST, (KT_) + NST - 1, q)                                                \
                __builtin_amdgcn_sched_barrier(0);                                                                         \
            }                                                                                                              \
        }                                                                                                                  \
        if (gk == 0 && (HAS_NEXT_)) { /* step KT_+1 has landed for everyone; everyone has left step KT_-1's stage */       \
            __builtin_amdgcn_sched_barrier(0);                                                                             \
            WAIT_                                                                                                          \
            __builtin_amdgcn_s_barrier();                                                                                  \
            __builtin_amdgcn_sched_barrier(0);                                                                             \
        }                                                                                                                  \
    }
#define K2_P16_WAIT_STEADY asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 3) * IPW) : "memory");
#define K2_P16_WAIT_DRAIN asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // ---- prologue: K steps 0 .. NST-2 in flight
#pragma unroll
    for (int p = 0; p < NST - 1; p++)
        if (p < nk) {
#pragma unroll
            for (int q = 0; q < IPW; q++) {
                if (p == 0) K2_P16_DMA(0, 0, q)
                if (p == 1) K2_P16_DMA(1, 1, q)
            }
        }
    const int arow = wr * WM + li, brow = wc * WN + li;
    const int swa = (arow >> 1) & 7, swb = ((BM + brow) >> 1) & 7;
    const float* ra[NST][2];
    const float* rb[NST][2];
#pragma unroll
    for (int st = 0; st < NST; st++)
#pragma unroll
        for (int gg = 0; gg < 2; gg++) {
            ra[st][gg] = smem + st * STAGE + arow * BK + (((4 * gg + kq) ^ swa) << 2);
            rb[st][gg] = smem + st * STAGE + (BM + brow) * BK + (((4 * gg + kq) ^ swb) << 2);
        }
    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; i++)
#pragma unroll
        for (int j = 0; j < NT; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 fa[2][MT], fb[2][NT];
    if (nk >= NST - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * IPW) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    K2_P16_READ(0, 0, 0)
    // residual and bias: behind the prologue's DMAs and the first barrier (vmcnt retires in issue order: the loop's counted waits cover
    // these loads conservatively, as in gemm_f32_mfma_pipe)
    float rres[MT][NT][4];
    if (R) {
#pragma unroll
        for (int j = 0; j < NT; j++) {
            const int col = min(n0 + wc * WN + j * 16 + li, g.N - 1);
#pragma unroll
            for (int i = 0; i < MT; i++)
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const int row = min(m0 + wr * WM + i * 16 + 4 * kq + e, g.M - 1);
                    rres[i][j][e] = R[(long long)row * g.ldr + col];
                }
        }
    }
    float bias_pre[NT];
#pragma unroll
    for (int j = 0; j < NT; j++) bias_pre[j] = g.bias ? g.bias[min(n0 + wc * WN + j * 16 + li, g.N - 1)] : 0.f;

    int kt = 0;
    const int nsteady = nk - (NST - 1);
    for (; kt + NST <= nsteady; kt += NST) {
        K2_P16_STEP(0, kt, true, true, K2_P16_WAIT_STEADY)
        K2_P16_STEP(1, kt + 1, true, true, K2_P16_WAIT_STEADY)
        K2_P16_STEP(2, kt + 2, true, true, K2_P16_WAIT_STEADY)
    }
#define K2_P16_S(j_) K2_P16_STEP((j_) % NST, kt + (j_), true, true, K2_P16_WAIT_STEADY)
#define K2_P16_T(j_) K2_P16_STEP((j_) % NST, kt + (j_), true, false, K2_P16_WAIT_DRAIN)
#define K2_P16_L(j_) K2_P16_STEP((j_) % NST, kt + (j_), false, false, K2_P16_WAIT_DRAIN)
    const int rs = nsteady - kt;
    switch (rs) {
        case 0: K2_P16_T(0) K2_P16_L(1) break;
        case 1: K2_P16_S(0) K2_P16_T(1) K2_P16_L(2) break;
        default: K2_P16_S(0) K2_P16_S(1) K2_P16_T(2) K2_P16_L(3) break;
    }
#undef K2_P16_S
#undef K2_P16_T
#undef K2_P16_L
#undef K2_P16_WAIT_DRAIN
#undef K2_P16_WAIT_STEADY
#undef K2_P16_STEP
#undef K2_P16_READ
#undef K2_P16_DMA

#pragma unroll
    for (int j = 0; j < NT; j++)
#pragma unroll
        for (int i = 0; i < MT; i++) {
            float vals[4];
#pragma unroll
            for (int e = 0; e < 4; e++) vals[e] = acc[i][j][e];
            epilogue_rows<4, Rows16>(g, vals, m0 + wr * WM + i * 16 + 4 * kq, n0 + wc * WN + j * 16 + li, C, nullptr, rres[i][j], R != nullptr, nullptr,
                                     g.bias, true, bias_pre[j]);
        }
}

// ---------------------------------------------------------------------------------------
// Ring variant of the LDS-DMA kernel (plain Linear, K % (32 KS) == 0).  Same tile image and swizzle as above; what changes is
// the pipeline and the K split:
//   * the fragments of a K tile are read from LDS into a SECOND register set while the MFMAs of the previous tile run, so
//     after a barrier a wave continues straight into MFMAs whose operands are already in registers.  The barrier of
//     iteration kt then only says "everyone holds tile kt in registers, and tile kt+1 has landed for everyone": it sits in
//     the middle of matrix work instead of in front of an LDS round trip, and the stage of tile kt is free the moment it
//     passes -- NST stages keep NST - 1 tiles in flight (the kernel above: NST - 1 stages in flight, one LDS latency and
//     the DMA issue exposed per K step).
//   * KS wave groups split each 32 KS-deep K step inside the workgroup (group g multiplies k in [32 g, 32 g + 32) of the
//     step), so a problem with few output tiles (streaming chunks: 256 .. 2048 rows) still puts 8 - 16 waves on a CU and
//     walks K in K / (32 KS) steps.  The groups' partial tiles meet in LDS after the loop and are summed in the fixed order
//     g = 0 .. KS - 1 (deterministic); each group finishes 16 / KS of the accumulator registers.
// ---- fused tail of the streaming conv modules' in_proj (GemmArgs::cf_*): GLU + chunk-causal depthwise conv + SwooshR on the tile.
// fin: the lane's finished accumulator values (rows lrow0 + Rows32::off(n) of the tile, GEMM column n0 + 32 wc + li).  LDS (the ring's
// memory, free by now): col [S][CH][CL] = per (stream, channel) the cached frames then the chunk's GLU outputs -- exactly the `col` array
// a lane of k_glu_causal_conv_reg holds in registers -- and the two filters' taps for the tile's CH = BN / 2 channels.
// A thread computes TPT consecutive frames of one (stream, channel) from REGISTERS: its TPT + K - 1 column values and its channel's taps
// are read from LDS once (the first version read two LDS words per FMA: 47 x 2 per output, LDS-bound at ~3 us per tile).  Same sums in
// the same order as the separate kernel.  Tc, CH and TPT are powers of two (shifts; an integer division costs ~40 instructions).
// What the tail needs from memory, requested BEFORE the K loop so that its latencies (the streams' slot indexes, then their cached
// frames; the taps; the scale rows) lie under the GEMM instead of behind it -- three dependent round trips were ~2.5 us of a 12 us launch.
template <int BM, int BN, int NTHR>
struct ConvPre {
    static constexpr int CH = BN / 2, lgCH = CH == 16 ? 4 : 5, TPT0 = BM * CH / NTHR;
    static_assert((CH == 16 || CH == 32) && (TPT0 == 2 || TPT0 == 4), "conv tail: tile form");
    float cv[8], wv[4], uv[4], le[TPT0], re[TPT0], bias, bcv, bwv;
    int slot, lgTc, pad, Kc, npc, lgTPS, cs, cr;
    bool live, full;   // full: every thread owns TPT0 frames of one (stream, channel) (Tc >= TPT0): le / re / bcv / bwv are prefetched
    // independent loads (before the DMA prologue): slot index, taps, biases, scale rows
    __device__ __forceinline__ void first(const GemmArgs& g, int tid, int m0, int n0, int col) {
        const int Tc = g.cf_Tc, K = g.cf_K, D = g.N >> 1, c0 = n0 >> 1;
        lgTc = 31 - __builtin_clz(Tc);
        pad = K >> 1;
        Kc = (K + 1) >> 1;
        npc = CH * pad;
        const int lgS = (BM == 64 ? 6 : 5) - lgTc;
        lgTPS = (NTHR == 512 ? 9 : 8) - lgS;
        cs = tid >> lgTPS;
        cr = tid & ((1 << lgTPS) - 1);
        const int b0 = m0 >> lgTc, Bn = g.M >> lgTc;
        live = b0 + cs < Bn;
        slot = live ? g.cf_slots[b0 + cs] : 0;
        bias = g.bias ? g.bias[col] : 0.f;
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int idx = tid + u * NTHR;
            wv[u] = idx < CH * Kc ? g.cf_wc[(long long)c0 * Kc + idx] : 0.f;
            uv[u] = idx < CH * K ? g.cf_ww[(long long)c0 * K + idx] : 0.f;
        }
        full = Tc >= TPT0;
        if (full) {   // this thread's outputs: (stream, frame group q, channel) as conv_outputs<K, TPT0> maps them
            const int lgQ = lgTc - (TPT0 == 2 ? 1 : 2), ch = tid & (CH - 1), q = (tid >> lgCH) & ((1 << lgQ) - 1), c = c0 + ch, tq = q * TPT0;
            bcv = g.cf_bc[c];
            bwv = g.cf_bw[c];
#pragma unroll
            for (int i = 0; i < TPT0; i++) {
                const int t = tq + i, il = min(t, K - 1), ir = min(max(t - (Tc - K), 0), K - 1);
                const float lv = g.cf_sc[(long long)c * K + il], rv = g.cf_sc[(long long)D * K + (long long)c * K + ir];
                le[i] = t < K ? lv : 0.f;
                re[i] = t >= Tc - K ? rv : 0.f;
            }
        }
    }
    // the streams' cached frames (the slot index has arrived by now: behind the prologue's first barrier)
    __device__ __forceinline__ void second(const GemmArgs& g, int n0) {
        const float* cbase = g.cf_pool + (long long)slot * g.cf_stride + g.cf_off + (long long)(n0 >> 1) * pad;
        const int nper = (npc + (1 << lgTPS) - 1) >> lgTPS;   // elements per thread, workgroup-uniform (<= 8): the unrolled slots behind it are skipped
#pragma unroll
        for (int u = 0; u < 8; u++)
            if (u < nper) cv[u] = (live && cr + (u << lgTPS) < npc) ? cbase[cr + (u << lgTPS)] : 0.f;
    }
};

template <int KT, int TPT, typename PRE>
__device__ __forceinline__ void conv_outputs(const GemmArgs& g, const PRE& pre, bool use_pre, const float* colL, const float* wcl, const float* wwl, int CL,
                                             int KcP, int lgTc, int lgCH, int S, int Bn, int b0, int c0, int m0, int D, int tid, int nthreads) {
    constexpr int K = KT, pad = K >> 1, Kc = (K + 1) >> 1;
    const int Tc = 1 << lgTc, CH = 1 << lgCH, lgQ = lgTc - (TPT == 1 ? 0 : TPT == 2 ? 1 : TPT == 4 ? 2 : 3), Q = 1 << lgQ;
    for (int w = tid; w < (S << (lgCH + lgQ)); w += nthreads) {
        const int ch = w & (CH - 1), sq = w >> lgCH, q = sq & (Q - 1), s = sq >> lgQ;
        if (b0 + s >= Bn) continue;
        const int c = c0 + ch, tq = q * TPT;
        const float* cp = colL + ((s << lgCH) + ch) * CL + tq;
        // causal_conv: all (K + 1) / 2 taps are live -- the TPT + Kc - 1 column values and the taps in registers, unrolled
        float col[TPT + Kc - 1], wc[Kc];
#pragma unroll
        for (int j = 0; j < TPT + Kc - 1; j++) col[j] = cp[j];   // (tq + j <= Tc - 1 + Kc - 1 = CL - 1: inside the column)
#pragma unroll
        for (int k = 0; k < Kc; k++) wc[k] = wcl[ch * KcP + k];
        // chunkwise_conv_scale (k_glu_causal_conv_reg's le / re) and the two biases: prefetched before the K loop where the thread's
        // outputs were known then (use_pre), else unconditional loads from clamped indexes, all in flight together
        float le[TPT], re[TPT], bcv, bwv;
        if (use_pre) {
            bcv = pre.bcv;
            bwv = pre.bwv;
#pragma unroll
            for (int i = 0; i < TPT; i++) {
                le[i] = pre.le[i < PRE::TPT0 ? i : 0];
                re[i] = pre.re[i < PRE::TPT0 ? i : 0];
            }
        } else {
            bcv = g.cf_bc[c];
            bwv = g.cf_bw[c];
#pragma unroll
            for (int i = 0; i < TPT; i++) {
                const int t = tq + i, il = min(t, K - 1), ir = min(max(t - (Tc - K), 0), K - 1);
                const float lv = g.cf_sc[(long long)c * K + il], rv = g.cf_sc[(long long)D * K + (long long)c * K + ir];
                le[i] = t < K ? lv : 0.f;
                re[i] = t >= Tc - K ? rv : 0.f;
            }
        }
        float* yp = g.C + ((long long)m0 + (s << lgTc) + tq) * g.ldc + c;
        // chunkwise_conv: of its K taps only those whose frame t + k - pad lies inside the chunk are live (8 of 31 at Tc = 8); with the
        // frame index a run-time value a tap loop cannot drop the others at compile time (it cost ~3.5 instructions per dead tap: the
        // tail was ~600 instructions per thread, ~2.7 us at two waves per SIMD) -- so the loop runs over the chunk's FRAMES tt, tap
        // k = tt - t + pad read from LDS: ascending tt is ascending k, the sum is formed in k_glu_causal_conv_reg's order.
        float xw[TPT];
#pragma unroll
        for (int i = 0; i < TPT; i++) xw[i] = bwv;
        {
            const float* xcol = colL + ((s << lgCH) + ch) * CL + pad;     // the chunk's frames 0 .. Tc - 1 of this (stream, channel)
            const float* wrow = wwl + ch * K + pad - tq;                  // tap of output i at frame tt: wrow[tt - i]
            for (int tt = 0; tt < Tc; tt++) {
                const float xv = xcol[tt];
#pragma unroll
                for (int i = 0; i < TPT; i++) {
                    const int k = tt - i + pad - tq;                      // (a uniform bound test per output: |tt - t| <= pad)
                    if (k >= 0 && k < K) xw[i] += wrow[tt - i] * xv;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < TPT; i++) {
            float xc = bcv;
#pragma unroll
            for (int k = 0; k < Kc; k++) xc += wc[k] * col[i + k];
            const float z = xw[i] * (1.0f + (le[i] + re[i])) + xc;
            yp[(long long)i * g.ldc] = fast_softplus(z - 1.0f) - 0.08f * z - 0.313261687f;   // SwooshR (online.hip swoosh_r)
        }
    }
}

template <int BM, int BN, int NV, int NTHR>
__device__ __forceinline__ void conv_tail(const GemmArgs& g, const ConvPre<BM, BN, NTHR>& pre, const float (&fin)[NV], float* smem, int m0, int n0, int wc,
                                          int lrow0, int li, int tid) {
    constexpr int CH = BN / 2, lgCH = CH == 16 ? 4 : 5, TPT0 = ConvPre<BM, BN, NTHR>::TPT0;
    const int Tc = g.cf_Tc, lgTc = pre.lgTc, K = g.cf_K, pad = pre.pad, Kc = pre.Kc, CL = pad + Tc, S = BM >> lgTc, KcP = Kc | 1;
    const int Bn = g.M >> lgTc, b0 = m0 >> lgTc, c0 = n0 >> 1, D = g.N >> 1;
    float* colL = smem;                    // [S][CH][CL]   (CL is odd for every (K, Tc) of the model zoo: conflict-free columns)
    float* wcl = colL + S * CH * CL;       // [CH][KcP]
    float* wwl = wcl + CH * KcP;           // [CH][K]       (K odd)
    __syncthreads();                       // every wave has left the ring / the K groups' exchange area
    {   // (a) the chunk's GLU outputs: value lanes (li & 16) == 0, their gates 16 lanes up
        const int ch = wc * 16 + (li & 15);
#pragma unroll
        for (int n = 0; n < NV; n++) {
            const float v = fin[n] + pre.bias;
            const float other = __shfl_xor(v, 16, 64);
            const float gated = v * (1.0f / (1.0f + __expf(-other)));
            const int lrow = lrow0 + Rows32::off(n);
            const int s = lrow >> lgTc, t = lrow & (Tc - 1);
            if ((li & 16) == 0) colL[(s * CH + ch) * CL + pad + t] = gated;
        }
    }
    // (b) the streams' cached frames (contiguous per stream: channels c0 .. c0 + CH - 1, pad frames each) and the taps, out of the
    // registers ConvPre filled before the K loop: thread = (stream cs, element cr) with TPS = NTHR / S threads per stream, <= 8 elements each
    const int npc = pre.npc, TPS = 1 << pre.lgTPS, cs = pre.cs, cr = pre.cr;
    const bool live = pre.live;
    const float inv_pad = 1.0f / (float)pad;
    const int nper = (npc + TPS - 1) >> pre.lgTPS;   // (uniform: unrolled slots past it are skipped, not predicated)
#pragma unroll
    for (int u = 0; u < 8; u++) {
        const int rem = cr + u * TPS;
        if (u < nper && live && rem < npc) {
            const int ch = __float2int_rz(((float)rem + 0.5f) * inv_pad), r = rem - ch * pad;   // (exact: rem < 512, pad in {3, 7, 15})
            colL[(cs * CH + ch) * CL + r] = pre.cv[u];
        }
    }
    {
        const int lgKc = 31 - __builtin_clz(Kc);   // (K + 1) / 2 is 4, 8 or 16
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int idx = tid + u * NTHR;
            if (idx < CH * Kc) wcl[(idx >> lgKc) * KcP + (idx & (Kc - 1))] = pre.wv[u];
            if (idx < CH * K) wwl[idx] = pre.uv[u];
        }
    }
    __syncthreads();
    // (c) outputs
#define K2_CONV_OUT(KT_, TPT_, PRE_) conv_outputs<KT_, TPT_>(g, pre, PRE_, colL, wcl, wwl, CL, KcP, lgTc, lgCH, S, Bn, b0, c0, m0, D, tid, NTHR)
    if (pre.full) {
        switch (K) {
            case 31: K2_CONV_OUT(31, TPT0, true); break;
            case 15: K2_CONV_OUT(15, TPT0, true); break;
            default: K2_CONV_OUT(7, TPT0, true); break;
        }
    } else {   // chunks of fewer frames than a thread's share (Tc = 2 against 4): two frames per thread, two passes over the threads
        switch (K) {
            case 31: K2_CONV_OUT(31, 2, false); break;
            case 15: K2_CONV_OUT(15, 2, false); break;
            default: K2_CONV_OUT(7, 2, false); break;
        }
    }
#undef K2_CONV_OUT
    // (d) cache = the last pad frames of [cache ; chunk] (read from LDS: the old cache in memory is not needed any more)
    float* cbase = g.cf_pool + (long long)pre.slot * g.cf_stride + g.cf_off + (long long)c0 * pad;
#pragma unroll
    for (int u = 0; u < 8; u++) {
        const int rem = cr + u * TPS;
        if (u < nper && live && rem < npc) {
            const int ch = __float2int_rz(((float)rem + 0.5f) * inv_pad), r = rem - ch * pad;
            cbase[rem] = colL[(cs * CH + ch) * CL + Tc + r];
        }
    }
}
struct ConvPreNone {};

template <int BM, int BN, int KS, int NST, int LW, int PF = 0, bool CONV = false>
__global__ __launch_bounds__(64 * ((BM / 32) * (BN / 32) * KS + LW + PF)) void gemm_f32_mfma_ring(GemmArgs g) {
    constexpr int BK = 32;
    constexpr int WCOLS = BN / 32;
    constexpr int TW = (BM / 32) * (BN / 32);  // waves per K group = 32x32 output tiles of the workgroup
    constexpr int NW = TW * KS;                // consumer waves; waves NW .. NW+LW-1 are loaders (LW == 0: consumers load)
    constexpr int NINST = (BM + BN) / 8;       // 1 KB wave-instructions per K group and stage
    // who issues the DMA of a stage: LW == 0 -- wave tw of every K group issues instructions tw, tw + TW, ... of its own group;
    // LW > 0 -- loader j issues instructions j, j + LW, ... of the KS * NINST instructions of the whole stage
    constexpr int ISS = LW > 0 ? LW : TW;                 // issuers sharing one instruction list
    constexpr int NLIST = LW > 0 ? KS * NINST : NINST;    // length of that list
    constexpr int IPW = (NLIST + ISS - 1) / ISS;
    constexpr int NFULL = NLIST % ISS == 0 ? ISS : NLIST % ISS;  // issuers 0..NFULL-1 issue IPW, the others IPW - 1
    constexpr int SUB = (BM + BN) * BK;        // floats per K group and stage: A rows then W rows
    constexpr int STAGE = SUB * KS;
    constexpr int RPG = 16 / KS;               // accumulator registers each group finishes
    static_assert(NST >= 2 && (KS == 1 || KS == 2 || KS == 4), "ring: NST >= 2, KS in {1, 2, 4}");

    if (g.skip_if_zero && *g.skip_if_zero == 0) return;
    extern __shared__ __attribute__((aligned(16))) float smem[];  // [NST][KS][SUB]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool loader = LW > 0 && wave >= NW;
    const int kg = loader ? 0 : wave / TW, tw = loader ? 0 : wave % TW;
    const int wr = tw / WCOLS, wc = tw % WCOLS;
    const int li = lane & 31, lh = lane >> 5;
    const int z0 = blockIdx.z % g.nb0, z1 = blockIdx.z / g.nb0;
    const float* __restrict__ A = g.A + z0 * g.sA0 + z1 * g.sA1;
    const float* __restrict__ W = g.W + z0 * g.sW0 + z1 * g.sW1;
    float* __restrict__ C = g.C + z0 * g.sC0 + z1 * g.sC1;
    const float* __restrict__ R = g.res ? g.res + z0 * g.sR0 + z1 * g.sR1 : nullptr;
    int mb_, nb_;
    xcd_tile(mb_, nb_, g.xcd_panels == 1 ? 0 : g.M, g.N);
    const int m0 = mb_ * BM, n0 = nb_ * BN;
    const int nk = g.K / (BK * KS);
    // (CONV: what the fused conv tail needs from memory goes out first, see ConvPre)
    typename std::conditional<CONV, ConvPre<BM, BN, 64 * NW>, ConvPreNone>::type cpre;
    if constexpr (CONV) {
        if (!loader && !(PF > 0 && wave == NW + LW)) cpre.first(g, tid, m0, n0, n0 + wc * 32 + li);
    }

    // tuning only (g.dbg != nullptr): lane 0 of every wave stamps s_memtime at the phases of the pipeline
    unsigned long long* stamp = g.dbg ? g.dbg + ((size_t)(blockIdx.x + blockIdx.y * gridDim.x) * (NW + LW + PF) + wave) * 64 : nullptr;
    int nstamp = 0;
#define K2_STAMP()                                                                     \
    if (stamp && lane == 0 && nstamp < 60) stamp[nstamp++] = __builtin_amdgcn_s_memtime();
    if (stamp && lane == 0) stamp[60] = __builtin_amdgcn_s_memrealtime();
    K2_STAMP()

    // ---- DMA issue (the loaders, or every consumer when LW == 0)
    const int me = LW > 0 ? wave - NW : tw;  // index among the issuers of my list
    const float* src[IPW];
    unsigned dst0[IPW];
    const unsigned lds_base = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)smem;
    if (LW == 0 || loader) {
#pragma unroll
        for (int q = 0; q < IPW; q++) {
            const int idx = min(me + q * ISS, NLIST - 1);
            const int gi = LW > 0 ? idx / NINST : kg, inst = LW > 0 ? idx % NINST : idx;  // K group and instruction within it
            const int slot = inst * 64 + lane;
            const int r = slot >> 3, cp = slot & 7;
            const int c = cp ^ ((r >> 1) & 7);
            if (inst < BM / 8) src[q] = A + (long long)min(m0 + r, g.M - 1) * g.lda + 4 * c + BK * gi;
            else src[q] = W + (long long)min(n0 + (r - BM), g.N - 1) * g.ldw + 4 * c + BK * gi;
            dst0[q] = lds_base + (gi * SUB + inst * 256) * 4;
        }
    }
    auto issue = [&](int kt) {
#pragma unroll
        for (int q = 0; q < IPW; q++) {
            if (me + q * ISS >= NLIST) break;  // wave-uniform
            const unsigned dst = __builtin_amdgcn_readfirstlane(dst0[q] + (kt % NST) * (STAGE * 4));
            const float* gp = src[q] + kt * (BK * KS);
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep)
                         : "v"(gp), "s"(dst)
                         : "memory");
        }
    };
    // my share of tile kt_next has landed when at most `behind` newer tiles of mine are pending
#define K2_RING_WAIT(behind_)                                                                               \
    {                                                                                                       \
        if ((behind_) == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                \
        else if (me < NFULL) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((behind_) * IPW > 63 ? 63 : (behind_) * IPW) : "memory"); \
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((behind_) * (IPW - 1) > 63 ? 63 : (behind_) * (IPW - 1)) : "memory");     \
    }

    if (PF > 0 && wave == NW + LW) {
        // ---- L2 prefetch wave.  Every workgroup of a launch walks K in lockstep, so a tile's lines are requested by all its
        // sharers (the gridDim.x workgroups of a row of tiles for A, the row tiles of the same XCD for W) at the same moment:
        // one request goes to the Infinity Cache / HBM and the others wait on the same miss -- every DMA pays the full miss
        // latency (~900 cycles), and a CU holds only ~9 KB of misses in flight: ~9 B/clk/CU, below the 12 B/clk a 128x64 tile
        // needs.  This wave touches ONE dword of each line of its 1/sharers share of the tile that the DMA will ask for in the
        // NEXT iteration, so that those DMA requests find the lines in the XCD's L2 (~300 cycles).
        const int gx = gridDim.x, nwg = gx * gridDim.y;
        const int sa = min(gx, BM), ra = (BM + sa - 1) / sa;             // A: sharers, rows of my share
        const int sw = max(1, min((nwg >> 3) / max(gx, 1), BN)), rw = (BN + sw - 1) / sw;  // W: row tiles per XCD
        const int a0 = (nb_ % sa) * ra, w0 = (mb_ % sw) * rw;
        const int nline = (ra + rw) * KS;  // lines per K step of my share (one 128-byte line = 32 floats of a row)
        __builtin_amdgcn_s_barrier();
        for (int kt = 0; kt + 1 < nk; kt++) {
            __builtin_amdgcn_s_barrier();
            const int t = kt + NST + 1;  // the DMA of tile kt + NST + 1 goes out one iteration from now
            if (t < nk) {
                for (int l = lane; l < nline; l += 64) {
                    const int rr = l / KS, kgi = l % KS;
                    const float* p = rr < ra ? A + (long long)min(m0 + min(a0 + rr, BM - 1), g.M - 1) * g.lda
                                             : W + (long long)min(n0 + min(w0 + rr - ra, BN - 1), g.N - 1) * g.ldw;
                    // LDS-destination form (256 scratch bytes behind the ring): no VGPR is written when the data arrives, so
                    // nothing the compiler has since reused can be clobbered
                    constexpr int SINK = NST * STAGE > (KS > 1 ? KS * TW * 1024 : 0) ? NST * STAGE : KS * TW * 1024;  // floats: behind the ring AND the reduce area
                    const unsigned sink = __builtin_amdgcn_readfirstlane(lds_base + SINK * 4);
                    const float* pp = p + (long long)t * (BK * KS) + BK * kgi;
                    unsigned keep;
                    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                                 : "=&s"(keep)
                                 : "v"(pp), "s"(sink)
                                 : "memory");
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }
    if (loader) {
        // ---- loader wave: keeps NST - 1 tiles in flight ahead of the tile being multiplied; one barrier per tile with the
        // consumers ("tile kt+1 has landed" one way, "tile kt is in your registers, its stage is free" the other way)
#pragma unroll
        for (int p = 0; p < NST; p++)
            if (p < nk) issue(p);
        if (nk >= NST) K2_RING_WAIT(NST - 1) else K2_RING_WAIT(0)
        __builtin_amdgcn_s_barrier();
        for (int kt = 0; kt + 1 < nk; kt++) {
            if (kt + NST - 1 < nk && NST > 2) K2_RING_WAIT(NST - 2) else K2_RING_WAIT(0)
            __builtin_amdgcn_s_barrier();
            if (kt + NST < nk) issue(kt + NST);
        }
        return;
    }

    // (one accumulator per wave: a second, independent chain for alternate k steps was measured and changes nothing -- a lone wave
    // issues dependent v_mfma_f32_32x32x2_f32 at the full 64-cycle rate, tools/probes/mfma_loop_probe.hip)
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.f;

    const int arow = wr * 32 + li, brow = wc * 32 + li;
    const int swa = (arow >> 1) & 7, swb = ((BM + brow) >> 1) & 7;
    // this lane's fragment addresses inside a stage (float offsets), one per 8-k group
    const float* fbase = smem + kg * SUB;
    int offa[4], offb[4];
#pragma unroll
    for (int gk = 0; gk < 4; gk++) {
        offa[gk] = arow * BK + (((2 * gk + lh) ^ swa) << 2);
        offb[gk] = (BM + brow) * BK + (((2 * gk + lh) ^ swb) << 2);
    }

    float4 fa[2][4], fb[2][4];
#define K2_RING_READ(set_, st_, gk_)                                                   \
    {                                                                                  \
        fa[set_][gk_] = *reinterpret_cast<const float4*>(fbase + (st_) + offa[gk_]);   \
        fb[set_][gk_] = *reinterpret_cast<const float4*>(fbase + (st_) + offb[gk_]);   \
    }
    // meet the other waves: everyone holds the current tile in registers and the next tile has landed (own DMA: wait for it first)
#define K2_RING_SYNC(kt_)                                                                                        \
    {                                                                                                            \
        if (LW == 0) {                                                                                           \
            if ((kt_) + NST - 1 < nk && NST > 2) K2_RING_WAIT(NST - 2) else K2_RING_WAIT(0)                      \
        }                                                                                                        \
        __builtin_amdgcn_s_barrier();                                                                            \
    }
    // one K tile out of register set cu_; MORE_: the next tile's fragments are read into set nx_ between the MFMAs, and the DMA
    // of tile kt_ + NST goes out behind the first MFMA (the matrix pipe is busy meanwhile)
#define K2_RING_STEP(cu_, nx_, kt_, MORE_)                                                                              \
    {                                                                                                                   \
        const int st_next = (((kt_) + 1) % NST) * STAGE;                                                                \
        _Pragma("unroll") for (int gk = 0; gk < 4; gk++) {                                                              \
            _Pragma("unroll") for (int e = 0; e < 4; e++) {                                                             \
                const float av = e == 0 ? fa[cu_][gk].x : e == 1 ? fa[cu_][gk].y : e == 2 ? fa[cu_][gk].z : fa[cu_][gk].w; \
                const float bv = e == 0 ? fb[cu_][gk].x : e == 1 ? fb[cu_][gk].y : e == 2 ? fb[cu_][gk].z : fb[cu_][gk].w; \
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);                                       \
                if (MORE_ && e == 0) {                                                                                  \
                    __builtin_amdgcn_sched_barrier(0);                                                                  \
                    if (LW == 0 && gk == 0 && (kt_) + NST < nk) issue((kt_) + NST);                                     \
                    K2_RING_READ(nx_, st_next, gk)                                                                      \
                    __builtin_amdgcn_sched_barrier(0);                                                                  \
                }                                                                                                       \
            }                                                                                                           \
        }                                                                                                               \
    }

    if (LW == 0) {
#pragma unroll
        for (int p = 0; p < NST; p++)
            if (p < nk) issue(p);
        K2_STAMP()
        // tile 0 landed for this wave once at most the NST - 1 newer tiles are pending (fewer exist when K is short: drain)
        if (nk >= NST) K2_RING_WAIT(NST - 1) else K2_RING_WAIT(0)
    }
    __builtin_amdgcn_s_barrier();
    if constexpr (CONV) cpre.second(g, n0);   // (its slot index was requested in front of the prologue's DMAs)
#pragma unroll
    for (int gk = 0; gk < 4; gk++) K2_RING_READ(0, 0, gk)

    K2_STAMP()
    int kt = 0;
    for (; kt + 2 < nk; kt += 2) {  // two tiles per trip, so that the register-set index is a compile-time constant
        K2_RING_SYNC(kt)
        K2_STAMP()
        K2_RING_STEP(0, 1, kt, true)
        K2_STAMP()
        K2_RING_SYNC(kt + 1)
        K2_STAMP()
        K2_RING_STEP(1, 0, kt + 1, true)
        K2_STAMP()
    }
    if (kt + 1 < nk) {  // two tiles left
        K2_RING_SYNC(kt)
        K2_RING_STEP(0, 1, kt, true)
        K2_RING_STEP(1, 0, kt + 1, false)
    } else {            // one tile left
        K2_RING_STEP(0, 1, kt, false)
    }
#undef K2_RING_STEP
#undef K2_RING_SYNC
#undef K2_RING_READ
#undef K2_RING_WAIT
    K2_STAMP()

    // ---- the K groups' partial tiles meet in LDS; group g finishes registers [g RPG, (g+1) RPG)
    float fin[RPG];
    if (KS > 1) {
        __syncthreads();  // every wave is done with the ring
        float4* red = reinterpret_cast<float4*>(smem);  // [KS][TW][4][64] float4
#pragma unroll
        for (int q4 = 0; q4 < 4; q4++)
            red[((kg * TW + tw) * 4 + q4) * 64 + lane] = make_float4(acc[4 * q4], acc[4 * q4 + 1], acc[4 * q4 + 2], acc[4 * q4 + 3]);
        __syncthreads();
        const float* redf = smem;
#pragma unroll
        for (int rr = 0; rr < RPG; rr++) {
            const int r = kg * RPG + rr;
            float v = 0.f;
#pragma unroll
            for (int s = 0; s < KS; s++) v += redf[((((s * TW + tw) * 4 + (r >> 2)) * 64 + lane) << 2) + (r & 3)];
            fin[rr] = v;
        }
    } else {
#pragma unroll
        for (int rr = 0; rr < RPG; rr++) fin[rr] = acc[rr];
    }

    // group kg finishes registers [kg RPG, (kg + 1) RPG): RPG = 16 -> the whole block; 8 -> rows (r&3) + 8 (r>>2) from 16 kg; 4 -> from 8 kg
    if constexpr (CONV) {
        static_assert(LW == 0, "conv tail: no loader waves (an L2-prefetch wave has left the kernel by now)");
        conv_tail<BM, BN, RPG, 64 * NW>(g, cpre, fin, smem, m0, n0, wc, wr * 32 + 4 * lh + 8 * ((kg * RPG) >> 2), li, tid);
        return;
    }
    epilogue_rows<RPG, Rows32>(g, fin, m0 + wr * 32 + 4 * lh + 8 * ((kg * RPG) >> 2), n0 + wc * 32 + li, C, R, fin, false, nullptr,
                               g.bias ? g.bias + z0 * g.sBias0 : nullptr);
    K2_STAMP()
    if (stamp && lane == 0) {
        stamp[62] = (unsigned long long)nstamp;
        stamp[63] = __builtin_amdgcn_s_memrealtime();
    }
#undef K2_STAMP
}

// ---------------------------------------------------------------------------------------
// Skinny-N variant (N <= 96, K % 64 == 0): the value projections of the attention modules
// (N = 12 x heads).  With so few columns a tiled launch is a handful of workgroups, each walking
// the whole K serially (22 us for 4064 x 48 x 512).  Here a workgroup owns 16 rows x all N, its
// four waves split K four ways, operands go straight from global memory into the
// v_mfma_f32_16x16x4_f32 fragment layout (lane l: A[l&15][k = l>>4]; a float4 per lane holds 4
// consecutive k, so four MFMAs consume its components -- the k order inside a 16-wide group is
// permuted, which a sum does not care about), and the four partial tiles meet in LDS.
// NW = waves per workgroup = K split: 4, or 8 for long K with few workgroups (K = 1024 .. 1920 against 256 .. 1024 rows in the
// streaming chunk step: the launch is one chain of K / (32 NW) dependent load rounds per wave)
template <int NT, int NW = 4>
__global__ __launch_bounds__(64 * NW) void gemm_f32_mfma_skinny(GemmArgs g) {
    if (g.skip_if_zero && *g.skip_if_zero == 0) return;
    __shared__ __attribute__((aligned(16))) float red[NW][NT][64][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, q = lane >> 4;
    const int m0 = blockIdx.x * 16, n0 = blockIdx.y * (16 * NT);  // grid.y walks N in chunks of 16*NT columns
    const int Kw = g.K / NW, kbeg = wave * Kw;
    const float* __restrict__ ap = g.A + (long long)min(m0 + r, g.M - 1) * g.lda + kbeg + 4 * q;
    const float* __restrict__ wp[NT];
#pragma unroll
    for (int j = 0; j < NT; j++) wp[j] = g.W + (long long)min(n0 + 16 * j + r, g.N - 1) * g.ldw + kbeg + 4 * q;
    f32x4 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; j++) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    constexpr int UN = NT <= 3 ? 4 : 2;  // 16-k groups loaded ahead of their MFMAs
    for (int k = 0; k < Kw; k += 16 * UN) {
        float4 a4[UN], w4[UN][NT];
#pragma unroll
        for (int u = 0; u < UN; u++) {
            // groups past this wave's K slice (Kw is a multiple of 16, not of 16*UN) contribute zero; clamp the address
            const int kk = min(k + 16 * u, Kw - 16);
            const float m = (k + 16 * u < Kw) ? 1.f : 0.f;
            a4[u] = *reinterpret_cast<const float4*>(ap + kk);
            a4[u].x *= m; a4[u].y *= m; a4[u].z *= m; a4[u].w *= m;
#pragma unroll
            for (int j = 0; j < NT; j++) w4[u][j] = *reinterpret_cast<const float4*>(wp[j] + kk);
        }
#pragma unroll
        for (int u = 0; u < UN; u++)
#pragma unroll
            for (int e = 0; e < 4; e++)
#pragma unroll
                for (int j = 0; j < NT; j++) {
                    const float av = e == 0 ? a4[u].x : e == 1 ? a4[u].y : e == 2 ? a4[u].z : a4[u].w;
                    const float bv = e == 0 ? w4[u][j].x : e == 1 ? w4[u][j].y : e == 2 ? w4[u][j].z : w4[u][j].w;
                    acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[j], 0, 0, 0);
                }
    }
#pragma unroll
    for (int j = 0; j < NT; j++) *reinterpret_cast<f32x4*>(&red[wave][j][lane][0]) = acc[j];
    __syncthreads();
    // wave w finishes the column tiles j = w, w + 4, ...: C/D layout col = lane & 15, row = 4 * (lane >> 4) + reg
    for (int j = wave; j < NT; j += NW) {
        float vals[4];
#pragma unroll
        for (int e = 0; e < 4; e++) {
            float v = red[0][j][lane][e] + red[1][j][lane][e] + red[2][j][lane][e] + red[3][j][lane][e];
            if (NW == 8) v += (red[4][j][lane][e] + red[5][j][lane][e]) + (red[6][j][lane][e] + red[7][j][lane][e]);
            vals[e] = v;
        }
        epilogue_rows<4, Rows16>(g, vals, m0 + 4 * q, n0 + 16 * j + r, g.C, g.res, vals, false, nullptr, g.bias);
    }
}

template <int BM, int BN, int WM, int WN, int NST = 3>
void launch_dma(const Ctx& ctx, const GemmArgs& a) {
    dim3 grid(cdiv(a.N, BN), cdiv(a.M, BM), a.nb0 * a.nb1);
    size_t lds = sizeof(float) * NST * (BM + BN) * 32;
    static LdsAttrOnce lds_attr;
    lds_attr.ensure(gemm_f32_mfma_dma<BM, BN, WM, WN, NST>, (int)lds);
    hipLaunchKernelGGL((gemm_f32_mfma_dma<BM, BN, WM, WN, NST>), grid, dim3(64 * (BM / WM) * (BN / WN)), lds, ctx.stream, a);
}

template <int BM, int BN, int KS, int NST, int LW, int PF = 0>
void launch_ring(const Ctx& ctx, const GemmArgs& a) {
    dim3 grid(cdiv(a.N, BN), cdiv(a.M, BM), a.nb0 * a.nb1);
    // the ring, or the K groups' partial tiles after the loop, whichever is larger
    size_t lds = sizeof(float) * std::max((size_t)NST * KS * (BM + BN) * 32, KS > 1 ? (size_t)KS * (BM / 32) * (BN / 32) * 1024 : (size_t)0) + (PF ? 256 : 0);
    static LdsAttrOnce lds_attr;
    lds_attr.ensure(gemm_f32_mfma_ring<BM, BN, KS, NST, LW, PF>, (int)lds);
    hipLaunchKernelGGL((gemm_f32_mfma_ring<BM, BN, KS, NST, LW, PF>), grid, dim3(64 * ((BM / 32) * (BN / 32) * KS + LW + PF)), lds, ctx.stream, a);
}

template <int BM, int BN, int KS, int NST, int PF = 0>
void launch_ring_conv(const Ctx& ctx, const GemmArgs& a) {
    dim3 grid(cdiv(a.N, BN), cdiv(a.M, BM), 1);
    const int pad = a.cf_K >> 1, Kc = (a.cf_K + 1) >> 1, CH = BN / 2;
    const size_t conv_fl = (size_t)(BM / a.cf_Tc) * CH * (pad + a.cf_Tc) + (size_t)CH * (Kc | 1) + (size_t)CH * a.cf_K;
    size_t lds = sizeof(float) * std::max({(size_t)NST * KS * (BM + BN) * 32, KS > 1 ? (size_t)KS * (BM / 32) * (BN / 32) * 1024 : (size_t)0, conv_fl}) + (PF ? 256 : 0);
    static LdsAttrOnce lds_attr;
    lds_attr.ensure(gemm_f32_mfma_ring<BM, BN, KS, NST, 0, PF, true>, (int)lds);
    hipLaunchKernelGGL((gemm_f32_mfma_ring<BM, BN, KS, NST, 0, PF, true>), grid, dim3(64 * ((BM / 32) * (BN / 32) * KS + PF)), lds, ctx.stream, a);
}

template <int BM, int BN, int WM, int WN, int NST>
void launch_pipe(const Ctx& ctx, const GemmArgs& a) {
    dim3 grid(cdiv(a.N, BN), cdiv(a.M, BM), 1);
    size_t lds = sizeof(float) * NST * (BM + BN) * 32;
    static LdsAttrOnce lds_attr;
    lds_attr.ensure(gemm_f32_mfma_pipe<BM, BN, WM, WN, NST>, (int)lds);
    K2_REQUIRE((long long)a.M * a.lda < (1ll << 29) && (long long)a.N * a.ldw < (1ll << 29), "pipe: operand too large for 31-bit lane byte offsets");
    K2_REQUIRE(a.K % 32 == 0 && a.K >= 32 * (NST - 1), "pipe: K %d too short for %d stages", a.K, NST);
    hipLaunchKernelGGL((gemm_f32_mfma_pipe<BM, BN, WM, WN, NST>), grid, dim3(64 * (BM / WM) * (BN / WN)), lds, ctx.stream, a);
}
// tuning table of the pipelined kernel: k2hip_debug_gemm cfg = 2000 + index
struct PipeCfg { int BM, BN, WM, WN, NST; };
#define K2_PIPE_TABLE(X)                                                                                                       \
    X(0, 128, 64, 64, 32, 3) X(1, 128, 64, 32, 32, 3) X(2, 128, 128, 64, 64, 3) X(3, 128, 128, 64, 32, 3) X(4, 128, 64, 64, 32, 4)  \
    X(5, 64, 64, 32, 32, 3) X(6, 128, 128, 64, 64, 4) X(7, 64, 128, 32, 64, 3) X(8, 128, 128, 32, 32, 3) X(9, 128, 64, 32, 32, 4) \
    X(10, 64, 64, 64, 32, 3) X(11, 64, 64, 32, 64, 3) X(12, 256, 64, 64, 32, 3) X(13, 128, 32, 32, 32, 3) X(14, 128, 32, 64, 32, 3)
#define X(i, bm, bn, wm, wn, nst) {bm, bn, wm, wn, nst},
const PipeCfg kPipe[] = {K2_PIPE_TABLE(X)};
#undef X
bool launch_pipe_idx(const Ctx& ctx, const GemmArgs& a, int idx) {
    switch (idx) {
#define X(i, bm, bn, wm, wn, nst) case i: launch_pipe<bm, bn, wm, wn, nst>(ctx, a); break;
        K2_PIPE_TABLE(X)
#undef X
        default: return false;
    }
    return true;
}

template <int BM, int BN, int WM, int WN, int NST>
void launch_p16(const Ctx& ctx, const GemmArgs& a) {
    dim3 grid(cdiv(a.N, BN), cdiv(a.M, BM), 1);
    size_t lds = sizeof(float) * NST * (BM + BN) * 32;
    static LdsAttrOnce lds_attr;
    lds_attr.ensure(gemm_f32_mfma_p16<BM, BN, WM, WN, NST>, (int)lds);
    K2_REQUIRE((long long)a.M * a.lda < (1ll << 29) && (long long)a.N * a.ldw < (1ll << 29), "p16: operand too large for 31-bit lane byte offsets");
    K2_REQUIRE(a.K % 32 == 0 && a.K >= 32 * (NST - 1) && !a.glu, "p16: K %d too short for %d stages, or a gated epilogue", a.K, NST);
    hipLaunchKernelGGL((gemm_f32_mfma_p16<BM, BN, WM, WN, NST>), grid, dim3(64 * (BM / WM) * (BN / WN)), lds, ctx.stream, a);
}
// tuning table of the 16x16x4 pipelined kernel: k2hip_debug_gemm cfg = 3000 + index
bool launch_p16_idx(const Ctx& ctx, const GemmArgs& a, int idx) {
    switch (idx) {
        case 0: launch_p16<64, 96, 32, 48, 3>(ctx, a); break;     // 4 waves of 32 x 48
        case 1: launch_p16<128, 96, 64, 48, 3>(ctx, a); break;    // 4 waves of 64 x 48
        case 2: launch_p16<64, 192, 32, 96, 3>(ctx, a); break;    // 4 waves of 32 x 96
        case 3: launch_p16<32, 96, 16, 48, 3>(ctx, a); break;     // 4 waves of 16 x 48
        default: return false;
    }
    return true;
}

// tuning table of the ring kernel: k2hip_debug_gemm cfg = 100 + index
struct RingCfg { int BM, BN, KS, NST, LW, PF; };
#define K2_RING_TABLE(X)                                                                                        \
    X(0, 128, 64, 1, 2, 0, 0) X(1, 128, 64, 1, 3, 0, 0) X(2, 128, 64, 1, 2, 0, 1) X(3, 128, 64, 1, 3, 0, 1) X(4, 128, 64, 1, 4, 0, 1)   \
    X(5, 64, 64, 1, 3, 0, 0) X(6, 64, 64, 1, 3, 0, 1) X(7, 64, 64, 1, 4, 0, 1) X(8, 64, 64, 2, 3, 0, 0) X(9, 64, 64, 2, 3, 0, 1)      \
    X(10, 64, 64, 4, 2, 0, 0) X(11, 64, 64, 4, 2, 0, 1) X(12, 32, 64, 2, 3, 0, 0) X(13, 32, 64, 4, 3, 0, 0) X(14, 32, 64, 4, 3, 0, 1)  \
    X(15, 32, 32, 4, 3, 0, 0) X(16, 32, 32, 4, 4, 0, 1) X(17, 64, 32, 4, 3, 0, 0) X(18, 128, 128, 1, 2, 0, 0) X(19, 128, 128, 1, 2, 0, 1) \
    X(20, 64, 128, 1, 3, 0, 0) X(21, 64, 128, 1, 3, 0, 1) X(22, 64, 96, 1, 3, 0, 0) X(23, 64, 96, 1, 3, 0, 1) X(24, 128, 96, 1, 2, 0, 1) \
    X(25, 128, 64, 1, 3, 2, 0) X(26, 128, 64, 1, 3, 2, 1) X(27, 128, 64, 2, 2, 0, 1) X(28, 64, 96, 2, 3, 0, 0) X(29, 64, 96, 2, 2, 0, 0) \
    X(30, 64, 96, 2, 2, 0, 1) X(31, 128, 96, 1, 3, 0, 0) X(32, 128, 96, 1, 3, 0, 1)
#define X(i, bm, bn, ks, nst, lw, pf) {bm, bn, ks, nst, lw, pf},
const RingCfg kRing[] = {K2_RING_TABLE(X)};
#undef X
bool launch_ring_idx(const Ctx& ctx, const GemmArgs& a, int idx) {
    switch (idx) {
#define X(i, bm, bn, ks, nst, lw, pf) case i: launch_ring<bm, bn, ks, nst, lw, pf>(ctx, a); break;
        K2_RING_TABLE(X)
#undef X
        default: return false;
    }
    return true;
}

template <int BM, int BN, int WM, int WN, int BK, int MODE>
void launch_cfg(const Ctx& ctx, const GemmArgs& a) {
    constexpr int LDSK = BK + 4;
    dim3 grid(cdiv(a.N, BN), cdiv(a.M, BM), a.nb0 * a.nb1);
    size_t lds = sizeof(float) * 2 * (BM + BN) * LDSK;
    static LdsAttrOnce lds_attr;
    lds_attr.ensure(gemm_f32_mfma<BM, BN, WM, WN, BK, MODE>, (int)lds);
    hipLaunchKernelGGL((gemm_f32_mfma<BM, BN, WM, WN, BK, MODE>), grid, dim3(64 * (BM / WM) * (BN / WN)), lds, ctx.stream, a);
}

template <int MODE>
void launch_mode(const Ctx& ctx, const GemmArgs& a, int cfg) {
    switch (cfg) {
        case 0: launch_cfg<128, 128, 64, 32, 32, MODE>(ctx, a); break;  // 8 waves
        case 1: launch_cfg<128, 128, 64, 32, 64, MODE>(ctx, a); break;  // 8 waves, BK 64
        case 3: launch_cfg<64, 64, 32, 32, 64, MODE>(ctx, a); break;    // 4 waves, BK 64
        case 4: launch_cfg<128, 64, 32, 32, 64, MODE>(ctx, a); break;   // 8 waves, BK 64
        case 5: case 7: case 8: case 11: launch_cfg<128, 64, 32, 32, 32, MODE>(ctx, a); break;   // 8 waves (and the fallback of the DMA-only configs)
        case 12: launch_cfg<128, 32, 32, 32, 32, MODE>(ctx, a); break;  // 4 waves, 32 columns (the second embed convolution: N = 32, 600k rows)
        default: launch_cfg<64, 64, 32, 32, 32, MODE>(ctx, a); break;   // 4 waves
    }
}

int g_forced_cfg = -1;  // debug_force_gemm_cfg (tuning hook); K2HIP_GEMM_CFG comes through tunables()

// Tile choice, from tools/gemm_tune.py on the benchmark's shapes (gpurun_out/gemm_tune_*.txt):
// on this path K is short (192..2560), so a launch is dominated by how well the prologue /
// epilogue of one workgroup overlaps the MFMA loop of its neighbours.  Small wave tiles
// (32x32 per wave, 4-5 waves per SIMD) win almost everywhere; the 128x128 tile only pays
// when the output is large enough to fill every CU several times over.
//   cfg 0: 128x128, 8 waves (64x32 per wave)   cfg 5: 128x64, 8 waves (32x32 per wave)
//   cfg 2:  64x64,  4 waves (32x32 per wave)
int choose_cfg(const GemmArgs& a) {
    if (g_forced_cfg >= 0) return g_forced_cfg;
    if (tunables().gemm_cfg >= 0) return tunables().gemm_cfg;
    // N <= 32 with many rows (encoder_embed.conv.4 as an implicit GEMM: 632 736 x 32 x 72 at the headline shape): a 64-column tile
    // multiplies 32 columns of padding
    if (a.N <= 32 && a.M >= 4096) return 12;
    // (the third embed convolution, 307 040 x 128 x 288, stays on 128x64 with K steps of 32: one 128-column tile that gathers each A row once is
    // 301 against 284 us, K steps of 64 382)
    if (a.N <= 64) return 2;    // 64x64 tiles, 4 waves
    if (a.M <= 64) return 3;    // a handful of rows (per-frame recurrent products, batched over layers): 64x64 tiles, K step 64
    // few output tiles (streaming chunks: 256..2048 rows): the launch is one latency-bound K sweep per
    // workgroup; small tiles with a 64-deep K step are fastest (gpurun_out/gemm_tune_s1.txt)
    if ((long long)cdiv(a.M, 128) * cdiv(a.N, 64) * a.nb0 * a.nb1 < 144) return 3;  // fewer 128x64 tiles than ~half the CUs
    if (a.N <= 128 && a.M < 32768) return 3;   // 64x64 tiles, K step 64 (the ConvNeXt 1x1s have enough rows for 128x64)
    {   // N a multiple of 96 and a multi-round 128x64 grid that leaves the last round mostly empty: 64x96 tiles balance it
        // (measured -9 % on 4064x1152x512, -4 % on 2048x1536x768; no gain on single-round grids, which are bubble-bound)
        const long long b5 = (long long)cdiv(a.M, 128) * cdiv(a.N, 64), b6 = (long long)cdiv(a.M, 64) * (a.N / 96);
        const bool plain = a.cv_Fout == 0 && !a.w_kn && a.nb0 * a.nb1 == 1 && a.K % 32 == 0 && a.K >= 64;
        if (plain && a.N % 96 == 0 && b5 > 256 && cdiv(b6, 256) * 6144 * 100 <= cdiv(b5, 256) * 8192 * 80) return 11;
        // 128x64 tiles that fill the last round of the 256 CUs badly while 64x64 tiles fill it well (the 6.25 Hz stack: 2048 rows x
        // 1536 / 2080 / 2560 columns -> 384 / 528 / 640 tiles): the smaller tile costs ~7 % per tile and wins 7-14 % on balance
        // (tools/probes/m2048_probe.py)
        const long long b9 = (long long)cdiv(a.M, 64) * cdiv(a.N, 64);
        const double e5 = (double)b5 / (double)(cdiv(b5, 256) * 256), e9 = (double)b9 / (double)(cdiv(b9, 256) * 256);
        if (plain && a.K >= 512 && b5 > 256 && e9 >= e5 + 0.12) return 9;
    }
    return 5;                   // 128x64 tiles, 8 waves (LDS-DMA pipeline when K % 32 == 0)
}

}  // namespace

int g_ablate = 0;
int g_use_dma = 1;
void debug_ring_shape(int idx, int* bm, int* bn, int* waves) {
    K2_REQUIRE(idx >= 0 && idx < (int)(sizeof(kRing) / sizeof(kRing[0])), "no ring cfg %d", idx);
    *bm = kRing[idx].BM;
    *bn = kRing[idx].BN;
    *waves = (kRing[idx].BM / 32) * (kRing[idx].BN / 32) * kRing[idx].KS + kRing[idx].LW + kRing[idx].PF;
}
int g_forced_ring = -1;
int g_forced_pipe = -1;
int g_forced_p16 = -1;
void debug_pipe_shape(int cfg, int M, int N, int* n_wg, int* waves) {
    const int idx = (cfg - 2000) % 100;
    K2_REQUIRE(idx >= 0 && idx < (int)(sizeof(kPipe) / sizeof(kPipe[0])), "no pipe cfg %d", idx);
    *n_wg = cdiv(M, kPipe[idx].BM) * cdiv(N, kPipe[idx].BN);
    *waves = (kPipe[idx].BM / kPipe[idx].WM) * (kPipe[idx].BN / kPipe[idx].WN);
}
void debug_force_gemm_cfg(int cfg) {
    const int dma_default = 1;
    g_forced_pipe = -1;
    g_forced_p16 = -1;
    if (cfg >= 3000) {  // 16x16x4 pipelined kernel table
        g_forced_p16 = cfg - 3000;
        g_forced_ring = -1;
        g_forced_cfg = -1;
        g_ablate = 0;
        g_use_dma = dma_default;
        return;
    }
    if (cfg >= 2000) {  // pipelined kernel table
        g_forced_pipe = cfg - 2000;
        g_forced_ring = -1;
        g_forced_cfg = -1;
        g_ablate = 0;
        g_use_dma = dma_default;
        return;
    }
    g_forced_ring = cfg >= 100 ? ((cfg - 100) & 0xff) : -1;
    if (cfg >= 100) {
        g_ablate = (cfg - 100) >> 8;
        g_forced_cfg = -1;
        g_use_dma = dma_default;
        return;
    }
    g_forced_cfg = cfg < 0 ? -1 : (cfg & 0x3f);
    g_ablate = cfg < 0 ? 0 : (cfg >> 8);
    g_use_dma = (cfg >= 0 && (cfg & 0x40)) ? 0 : dma_default;  // +64: classic (register-staged) kernel
}

void gemm(const Ctx& ctx, const GemmArgs& a) {
    K2_REQUIRE(a.M > 0 && a.N > 0 && a.K > 0, "gemm: empty shape %dx%dx%d", a.M, a.N, a.K);
    K2_REQUIRE(!a.wz_map || a.w_kn, "gemm: wz_map is for the [K,N] operand form");
    K2_REQUIRE(!a.glu || ((a.glu_cols > 0 ? a.glu_cols : a.N) % 32 == 0 && a.glu_cols <= a.N && a.N > 96 && !a.res && !a.mul && !a.byp_orig &&
                          a.act == ACT_NONE && a.nb0 * a.nb1 == 1),
               "gemm: the gated epilogue needs its paired columns in whole blocks of 32, N > 96 and no other epilogue term");
    K2_REQUIRE(a.cv_Fout > 0 || a.lda % 4 == 0, "gemm: lda %d must be a multiple of 4", a.lda);
    K2_REQUIRE(a.K >= 4 || (a.w_kn && a.K >= 1), "gemm: K=%d too small", a.K);  // [K,N] form: A rows are zero-padded to 4, W rows k >= K masked
    K2_REQUIRE(a.w_kn || a.K % 4 == 0, "gemm: K %d must be a multiple of 4", a.K);
    K2_REQUIRE(!a.w_kn || a.lda >= ((a.K + 3) & ~3), "gemm: [K,N] form needs A rows zero-padded to a multiple of 4");
    K2_REQUIRE(a.ldw % 4 == 0, "gemm: ldw %d must be a multiple of 4", a.ldw);
    K2_REQUIRE(!a.w_kn || (a.N % 4 == 0 && a.N >= 4), "gemm: [K,N] operand needs N %% 4 == 0 (N=%d)", a.N);
    K2_REQUIRE(!(a.w_kn && a.cv_Fout > 0), "gemm: conv gather and [K,N] operand cannot be combined");
    K2_REQUIRE(a.cv_Fout == 0 || (a.seg_len > 0 && a.seg_len % 4 == 0), "gemm: conv gather needs seg_len %% 4 == 0");
    const double fl = 2.0 * a.M * (double)a.N * a.K * a.nb0 * a.nb1;
    ctx.add_flops(fl, 0.0, 1);
    if (ctx.dry) return;
    if (ctx.instrument) K2_HIP(hipEventRecord(ctx.next_event(), ctx.stream));
    int cfg = choose_cfg(a);
    GemmArgs b = a;
    b.ablate = g_ablate;
    b.xcd_panels = tunables().xcd_panels;
    if (ctx.instrument && ctx.gemm_log)
        ctx.gemm_log->push_back({a.M, a.N, a.K, a.nb0 * a.nb1, a.act, a.res != nullptr, a.cv_Fout > 0 ? 1 : (a.w_kn ? 2 : 0), 0.f});
    // (a) N <= 96: few columns; (b) small problems (streaming chunks, beam search: a 128x64 grid would leave most CUs idle and
    // every workgroup would walk K serially): the same kernel over column chunks of 96
    if (g_forced_p16 >= 0) {  // tuning hook
        K2_REQUIRE(a.cv_Fout == 0 && !a.w_kn && !a.mul && a.res_div == 1 && !a.act_after_res && !a.glu && a.K % 32 == 0 && a.K >= 64 && a.nb0 * a.nb1 == 1,
                   "p16 cfg %d does not fit this GEMM", g_forced_p16);
        K2_REQUIRE(launch_p16_idx(ctx, b, g_forced_p16), "no p16 cfg %d", g_forced_p16);
        K2_HIP(hipGetLastError());
        if (ctx.instrument) K2_HIP(hipEventRecord(ctx.next_event(), ctx.stream));
        return;
    }
    if (g_forced_pipe >= 0) {  // tuning hook
        K2_REQUIRE(a.cv_Fout == 0 && !a.w_kn && !a.mul && a.res_div == 1 && !a.act_after_res && a.K % 32 == 0 && a.K >= 64 && a.nb0 * a.nb1 == 1,
                   "pipe cfg %d does not fit this GEMM", g_forced_pipe);
        K2_REQUIRE(launch_pipe_idx(ctx, b, g_forced_pipe), "no pipe cfg %d", g_forced_pipe);
        K2_HIP(hipGetLastError());
        if (ctx.instrument) K2_HIP(hipEventRecord(ctx.next_event(), ctx.stream));
        return;
    }
    if (g_forced_ring >= 0) {  // tuning hook
        const RingCfg& rc = kRing[g_forced_ring];
        K2_REQUIRE(a.cv_Fout == 0 && !a.w_kn && !a.mul && a.res_div == 1 && !a.act_after_res && a.K % (32 * rc.KS) == 0 && a.K >= 32 * rc.KS,
                   "ring cfg %d does not fit this GEMM", g_forced_ring);
        K2_REQUIRE(launch_ring_idx(ctx, b, g_forced_ring), "no ring cfg %d", g_forced_ring);
        K2_HIP(hipGetLastError());
        if (ctx.instrument) K2_HIP(hipEventRecord(ctx.next_event(), ctx.stream));
        return;
    }
    const Tunables& tn = tunables();
    const bool forced = g_forced_cfg >= 0 || tn.gemm_cfg >= 0, use_dma = g_use_dma != 0;
    const bool plain = a.cv_Fout == 0 && !a.w_kn && a.nb0 * a.nb1 == 1 && a.K % 64 == 0 && a.K >= 64;
    // (16-row workgroups re-read the weight chunk M/16 times: with many rows and a short K the 64x64 tiles are better)
    const bool few_tiles = (long long)cdiv(a.M, 128) * cdiv(a.N, 64) < 144 && a.M <= 4096 && !(a.M >= 2048 && a.K <= 256 && a.N > 272);
    const bool skinny_ok = !forced && plain && !a.mul && ((a.N <= 96 && a.M >= 512) || few_tiles);
    const bool skinny16_ok = skinny_ok && !a.glu;  // the 16-column C/D layout of gemm_f32_mfma_skinny has no lane pair 16 apart
    // Small problems with more than a handful of columns (the streaming chunk step: 256 .. 2048 rows): small ring tiles with the
    // K step split over four (two) wave groups of the workgroup put 4 - 8 waves on ~200 CUs and walk K in K / 128 (K / 64) steps
    // through coalesced LDS-DMA tiles, where the 16-row skinny kernel re-reads the weight chunk M / 16 times straight into
    // fragment layout (half-used cache lines; it is bound by the texture-address path, not by latency).  Choice by grid size,
    // from tools/gemm_lab.py streaming (gpurun_out/lab_str1.txt): 17 % less GEMM time over the chunk step's shapes.
    if (skinny_ok && few_tiles && a.N > 96 && a.res_div == 1 && !a.act_after_res) {
        const long long g32 = (long long)cdiv(a.M, 32) * cdiv(a.N, 32), g6432 = (long long)cdiv(a.M, 64) * cdiv(a.N, 32);
        int ring = -1;
        if (a.K % 128 == 0 && g32 <= 256) ring = 16;          // 32x32 tiles, KS 4, 4 stages + L2 prefetch wave
        else if (a.K % 128 == 0 && g6432 <= 256) ring = 17;   // 64x32 tiles, KS 4, 3 stages
        // K a multiple of 64 but not of 128 (the 256-wide stacks' feed-forward outputs: K = 576 / 960, and K = 192): 32x64 tiles, KS 2 --
        // round 4, tools/gemm_lab.py streaming-real (gpurun_out/r4c/lab_stream.txt): 1024 x 256 x 960 14.7 against 17.3 us (skinny),
        // x 576 10.4 against 12.2, 2048 x 192 x 192 6.2 against 7.9 (64x64 tiles)
        else if (a.K % 64 == 0 && a.K % 128 != 0 && (long long)cdiv(a.M, 32) * cdiv(a.N, 64) <= 256) ring = 12;
        else if (a.K % 64 == 0 && (long long)cdiv(a.M, 64) * cdiv(a.N, 64) >= 96) ring = 8;  // 64x64 tiles, KS 2, 3 stages
        if (ring >= 0) {
            launch_ring_idx(ctx, b, ring);
            K2_HIP(hipGetLastError());
            if (ctx.instrument && ctx.gemm_log) ctx.gemm_log->back().kind += 64;
            if (ctx.instrument) K2_HIP(hipEventRecord(ctx.next_event(), ctx.stream));
            return;
        }
    }
    // a few hundred rows x <= 96 columns with a long enough K (the streaming value projections of the downsampled stacks: 256 x 96 x 512,
    // 512 x 48 x 384): 32x32 ring tiles with the K step split four ways, 6.4 / 5.7 against 9.3 / 6.5 us (same lab run)
    if (skinny16_ok && a.N <= 96 && a.M <= 512 && a.K % 128 == 0 && a.K >= 384 && a.res_div == 1 && !a.act_after_res) {
        launch_ring_idx(ctx, b, 15);
        K2_HIP(hipGetLastError());
        if (ctx.instrument && ctx.gemm_log) ctx.gemm_log->back().kind += 64;
        if (ctx.instrument) K2_HIP(hipEventRecord(ctx.next_event(), ctx.stream));
        return;
    }
    if (skinny16_ok) {
        if (a.N <= 48) hipLaunchKernelGGL(gemm_f32_mfma_skinny<3>, dim3(cdiv(a.M, 16), 1), dim3(256), 0, ctx.stream, b);
        else if (a.K % 128 == 0 && ((a.K >= 1024 && (long long)cdiv(a.M, 16) * cdiv(a.N, 96) <= 384) ||
                                    (a.K >= 512 && (long long)cdiv(a.M, 16) * cdiv(a.N, 96) <= 128)))
            hipLaunchKernelGGL((gemm_f32_mfma_skinny<6, 8>), dim3(cdiv(a.M, 16), cdiv(a.N, 96)), dim3(512), 0, ctx.stream, b);
        else hipLaunchKernelGGL(gemm_f32_mfma_skinny<6>, dim3(cdiv(a.M, 16), cdiv(a.N, 96)), dim3(256), 0, ctx.stream, b);
        K2_HIP(hipGetLastError());
        if (ctx.instrument && ctx.gemm_log) ctx.gemm_log->back().kind += 32;
        if (ctx.instrument) K2_HIP(hipEventRecord(ctx.next_event(), ctx.stream));
        return;
    }
    const bool dma_ok = a.cv_Fout == 0 && !a.w_kn && a.K % 32 == 0 && a.K >= 64 && a.lda % 4 == 0 && !a.mul && a.res_div == 1 && !a.act_after_res;
    if (dma_ok && use_dma && !forced && a.nb0 * a.nb1 > 1 && a.M <= 64) {
        // a handful of rows against many layers' weight matrices (LSTM wavefront): a weight-streaming problem -- 64x64 tiles, three
        // 16 KB stages in flight per workgroup
        // enough workgroups to keep ~3 per CU streaming (bytes in flight are what sets the rate): 32-column tiles when 64-column
        // tiles would give fewer than ~600
        if ((long long)cdiv(a.N, 64) * a.nb0 * a.nb1 < 600) launch_dma<32, 32, 32, 32, 4>(ctx, b);
        else launch_dma<32, 64, 32, 32, 4>(ctx, b);
        K2_HIP(hipGetLastError());
        if (ctx.instrument && ctx.gemm_log) ctx.gemm_log->back().kind += 16;
        if (ctx.instrument) K2_HIP(hipEventRecord(ctx.next_event(), ctx.stream));
        return;
    }
    // One problem per launch, K % 32 == 0: the pipelined kernel, tile by a small cost model fitted to tools/gemm_lab.py offline
    // (gpurun_out/lab_pipe3.txt).  The busiest CU runs ceil(tiles / 256) tiles; a tile costs its K steps plus a fixed part (prologue,
    // last steps, epilogue -- less of it is exposed when several small workgroups share the CU), small tiles pay a few percent
    // for their extra operand traffic.  Examples it reproduces: 4064 x 512 -> 128x64 (256 tiles, one round); 4064 x 1152,
    // 2048 x 2560 / 2080 -> 64x64 (1152 / 1280 / 1056 tiles: 4.5 / 5 / 4.1 rounds of 4096 instead of 2.25 / 2.5 / 2.1 -> 3 of 8192);
    // 4064 x 1024 / 1920, 2048 x 2048 -> 128x128; 16160 x 192 -> 64x64 or 128x32 (3 rounds of 4096 instead of 2 of 8192).
    if (dma_ok && use_dma && !forced && a.nb0 * a.nb1 == 1 && a.K >= 64 && a.M >= 256 &&
        (long long)a.M * a.lda < (1ll << 29) && (long long)a.N * a.ldw < (1ll << 29)) {
        struct Cand { int idx, bm, bn; double fixed_steps, penalty; };
        static const Cand cands[] = {{8, 128, 128, 3.5, 0.0}, {1, 128, 64, 3.3, 0.02}, {5, 64, 64, 4.0, 0.12}, {13, 128, 32, 5.5, 0.08}};
        const double nk = a.K / 32.0;
        int best = -1;
        double best_cost = 0;
        for (const Cand& c : cands) {
            const long long tiles = (long long)cdiv(a.M, c.bm) * cdiv(a.N, c.bn);
            const double cost = (double)cdiv(tiles, 256) * c.bm * c.bn * (nk + c.fixed_steps) * (1.0 + c.penalty);
            if (best < 0 || cost < best_cost) {
                best = c.idx;
                best_cost = cost;
            }
        }
        // The 16x16x4 form's 64 x 96 tile (gemm_f32_mfma_p16, round 5) where N is a multiple of 96 and the same model prefers it: the
        // launches that quantise badly in 32 x 32 blocks -- 2048 x 768 is 256 tiles of 64 x 96 against 192 of 128 x 64 (tools/gemm_lab.py,
        // gpurun_out/r5j/lab_p16.txt: 69.3 against 83.5 us at K = 2560, 57.2 / 67.8 at 2048, 44.7 / 52.5 at 1536, 25.7 / 29.0 at 768;
        // 4064 x 1152 x 512 47.4 / 48.3; it loses where 128 x 64 already divides the output: 8096 x 768 x 256 40.7 / 36.3, which the model
        // reproduces).  One wave per SIMD on a single-round grid, so its prologue and tail are fully exposed: six fixed steps.
        if (a.N % 96 == 0 && !a.glu && a.M >= 1024) {
            const long long tiles = (long long)cdiv(a.M, 64) * (a.N / 96);
            // (a long K loop amortises the form's extra operand traffic: 16160 x 192 x 2432 is 126.8 us on it against 134.8 on 128 x 32,
            // gpurun_out/r5q/lab_p16_all.txt -- 6 % over the other tiles from 64 K steps, 10 % below)
            const double cost = (double)cdiv(tiles, 256) * 64 * 96 * (nk + 6.0) * (nk >= 64 ? 1.06 : 1.10);
            if (tiles >= 128 && cost < best_cost) {
                launch_p16_idx(ctx, b, 0);
                K2_HIP(hipGetLastError());
                // (+128 + 2048: the pipelined kernel on 16x16x4 tiles; BM / 32 = 2 in bits 8-11, BN / 32 = 3 in bits 12-15)
                if (ctx.instrument && ctx.gemm_log) ctx.gemm_log->back().kind += 128 + 256 * 2 + 4096 * 3 + (1 << 20);
                if (ctx.instrument) K2_HIP(hipEventRecord(ctx.next_event(), ctx.stream));
                return;
            }
        }
        launch_pipe_idx(ctx, b, best);
        K2_HIP(hipGetLastError());
        // (+128: the pipelined kernel; its tile in bits 8.. so that a profile can be grouped by instantiation: BM / 32, BN / 32)
        if (ctx.instrument && ctx.gemm_log) ctx.gemm_log->back().kind += 128 + 256 * (kPipe[best].BM / 32) + 4096 * (kPipe[best].BN / 32);
        if (ctx.instrument) K2_HIP(hipEventRecord(ctx.next_event(), ctx.stream));
        return;
    }
    // (batched launches too -- the Conformer's per-(head, stream) score products -- as long as the tile choice is one of the DMA kernel's)
    if (dma_ok && use_dma && (cfg == 5 || cfg == 0 || (cfg >= 7 && cfg <= 11))) {
        if (cfg == 11) launch_dma<64, 96, 32, 32, 2>(ctx, b);        // 64x96 tiles (6 waves): N % 96 == 0 outputs that 128x64 quantises badly
        else if (cfg == 9) launch_dma<64, 64, 32, 32, 2>(ctx, b);    // tuning: 64x64 tiles, 2 stages
        else if (cfg == 10) launch_dma<64, 64, 32, 32, 3>(ctx, b);   // tuning: 64x64 tiles, 3 stages
        else if (cfg == 7) launch_dma<128, 64, 32, 32, 2>(ctx, b);  // tuning: 2 stages, 3 workgroups per CU
        else if (cfg == 8) launch_dma<128, 64, 32, 32, 4>(ctx, b);  // tuning: 4 stages, 1 workgroup per CU
        else if (cfg == 5) launch_dma<128, 64, 32, 32, 2>(ctx, b);  // 2 stages: 3 workgroups per CU
        else launch_dma<128, 128, 64, 32>(ctx, b);
        K2_HIP(hipGetLastError());
        if (ctx.instrument && ctx.gemm_log) ctx.gemm_log->back().kind += 16;
        if (ctx.instrument) K2_HIP(hipEventRecord(ctx.next_event(), ctx.stream));
        return;
    }
    if (a.cv_Fout > 0) launch_mode<MODE_CONV>(ctx, b, cfg);
    else if (a.w_kn) launch_mode<MODE_WKN>(ctx, b, cfg);
    else launch_mode<MODE_PLAIN>(ctx, b, cfg);
    K2_HIP(hipGetLastError());
    if (ctx.instrument) K2_HIP(hipEventRecord(ctx.next_event(), ctx.stream));
}

bool gemm_glu_causal_conv(const Ctx& ctx, const float* x, const float* wg, const float* bg, float* pool, long long slot_stride, long long off,
                          const int* slots, const float* wc, const float* bc, const float* ww, const float* bw, const float* sc, float* y, int B,
                          int Tc, int D, int K) {
    const int M = B * Tc, N = 2 * D;
    // tiles: 32 / 64 rows of whole streams (Tc | 32), 32 / 64 GEMM columns of whole (value | gate) blocks; the K step split over the
    // workgroup's wave groups as the dispatcher of gemm() does for these shapes (streaming chunk steps: a few hundred tiles at most)
    if (Tc < 2 || 32 % Tc != 0 || D % 64 != 0 || (K != 31 && K != 15 && K != 7)) return false;
    const long long g32 = (long long)cdiv(M, 32) * (N / 32), g6432 = (long long)cdiv(M, 64) * (N / 32);
    int form = -1;   // 0: 32x32 KS 4 (4 stages), 1: 64x32 KS 4, 2: 32x64 KS 2, 3: 64x64 KS 2
    if (D % 128 == 0 && g32 <= 256) form = 0;
    else if (D % 128 == 0 && g6432 <= 256) form = 1;
    else if (D % 128 == 0) form = 3;
    else if ((long long)cdiv(M, 32) * (N / 64) <= 256) form = 2;
    else form = 3;
    {   // the tail's cache staging holds at most 8 elements per thread (conv_tail): tile's streams x channels x cached frames
        const int bm = (form == 1 || form == 3) ? 64 : 32, ch = (form >= 2) ? 32 : 16, nthr = (bm / 32) * (ch / 16) * (form >= 2 ? 2 : 4) * 64;
        if ((long long)(bm / Tc) * ch * (K >> 1) > 8ll * nthr) return false;
    }
    const double fl = 2.0 * M * (double)N * D;
    ctx.add_flops(fl, 2.0 * M * (double)D * (K + (K + 1) / 2), 1);
    if (ctx.dry) return true;
    GemmArgs g;
    g.A = x; g.lda = D; g.W = wg; g.ldw = D; g.bias = bg; g.C = y; g.ldc = D; g.M = M; g.N = N; g.K = D;
    g.cf_pool = pool; g.cf_stride = slot_stride; g.cf_off = off; g.cf_slots = slots;
    g.cf_wc = wc; g.cf_bc = bc; g.cf_ww = ww; g.cf_bw = bw; g.cf_sc = sc; g.cf_Tc = Tc; g.cf_K = K;
    g.xcd_panels = tunables().xcd_panels;
    if (ctx.instrument) K2_HIP(hipEventRecord(ctx.next_event(), ctx.stream));
    if (ctx.instrument && ctx.gemm_log) ctx.gemm_log->push_back({M, N, D, 1, 0, 0, 64, 0.f});
    switch (form) {
        case 0: launch_ring_conv<32, 32, 4, 4, 1>(ctx, g); break;   // (+ the L2-prefetch wave, as gemm()'s choice for this grid has it)
        case 1: launch_ring_conv<64, 32, 4, 3>(ctx, g); break;
        case 2: launch_ring_conv<32, 64, 2, 3>(ctx, g); break;
        default: launch_ring_conv<64, 64, 2, 3>(ctx, g); break;
    }
    K2_HIP(hipGetLastError());
    if (ctx.instrument) K2_HIP(hipEventRecord(ctx.next_event(), ctx.stream));
    return true;
}

void linear(const Ctx& ctx, const float* A, int lda, const float* W, const float* bias, float* C, int ldc, int M, int K,
            int N, int act, const float* res, int ldr) {
    GemmArgs g;
    g.A = A; g.lda = lda; g.W = W; g.ldw = K; g.bias = bias; g.C = C; g.ldc = ldc;
    g.M = M; g.N = N; g.K = K; g.act = act; g.res = res; g.ldr = ldr;
    gemm(ctx, g);
}

}  // namespace k2hip
