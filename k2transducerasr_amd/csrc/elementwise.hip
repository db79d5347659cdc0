// HBM-bound kernels of the Zipformer2 path: padding quirk, Conv2dSubsampling's
// first conv and depthwise 7x7, BiasNorm / bypass, GLU / tanh gates, depthwise
// Conv1d, learned down/up-sampling.  All activations are [rows, channels] with
// channels contiguous, so every kernel moves float4 per lane along channels.
#include "kernels.h"

namespace k2hip {
namespace {

constexpr float kLogFloor = -23.025850929940457F;  // PadHelper.cs:58

// hardware exp/log (v_exp_f32 / v_log_f32), ~1e-6 relative -- see gemm.hip apply_act
__device__ __forceinline__ float fast_softplus(float z) { return z > 15.f ? z : __logf(1.0f + __expf(z)); }
__device__ __forceinline__ float swoosh_r(float v) { return fast_softplus(v - 1.0f) - 0.08f * v - 0.313261687f; }
__device__ __forceinline__ float fast_tanh(float v) {
    float e = __expf(-2.0f * fabsf(v));
    float t = (1.0f - e) / (1.0f + e);
    return v < 0.f ? -t : t;
}

__device__ __forceinline__ float sigm(float s) { return 1.0f / (1.0f + __expf(-s)); }
__device__ __forceinline__ float dswish(float v) { return v / (1.0f + __expf(1.0f - v)); }  // v * sigmoid(v - 1)

inline int nblocks(long long n, int per) { return (int)((n + per - 1) / per); }

// ---- F3: PadHelper.PadSequence (PadHelper.cs:20-60) on device ------------------
//  out[b, j] = j < len[b] ? src[b][j] : 0 ; then x == 0 -> log floor (Q1, Q2)
__global__ void k_pad_logfloor(const float* __restrict__ packed, const long long* __restrict__ off,
                               const long long* __restrict__ len, float* __restrict__ out, long long L) {
    int b = blockIdx.y;
    const float* src = packed + off[b];
    long long n = len[b];
    for (long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x; j < L; j += (long long)gridDim.x * blockDim.x) {
        float v = j < n ? src[j] : 0.0f;
        out[(long long)b * L + j] = (v == 0.0f) ? kLogFloor : v;
    }
}
__global__ void k_pad_logfloor_dense(const float* __restrict__ feats, long long n, float* __restrict__ out, long long L) {
    int b = blockIdx.y;
    for (long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x; j < L; j += (long long)gridDim.x * blockDim.x) {
        float v = j < n ? feats[(long long)b * n + j] : 0.0f;
        out[(long long)b * L + j] = (v == 0.0f) ? kLogFloor : v;
    }
}

// ---- Conv2dSubsampling conv.0: 1->8 ch, 3x3, pad (TPAD,1); out NHWC [B,T-2+2*TPAD,F,8]
//      Zipformer2: TPAD 0 + SwooshR; Conformer: TPAD 1 + DoubleSwish
template <int TPAD, bool DSWISH>
__global__ void k_conv0(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                        float* __restrict__ y, int B, int T, int F) {
    __shared__ float sw[72], sb[8];
    if (threadIdx.x < 72) sw[threadIdx.x] = w[threadIdx.x];
    if (threadIdx.x < 8) sb[threadIdx.x] = bias[threadIdx.x];
    __syncthreads();
    int T1 = T - 2 + 2 * TPAD;
    long long n = (long long)B * T1 * F;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int f = (int)(i % F);
    long long bt = i / F;
    int t = (int)(bt % T1), b = (int)(bt / T1);
    float xin[3][3];
#pragma unroll
    for (int kt = 0; kt < 3; kt++)
#pragma unroll
        for (int kf = 0; kf < 3; kf++) {
            int ff = f + kf - 1, tt = t + kt - TPAD;
            xin[kt][kf] = (ff >= 0 && ff < F && tt >= 0 && tt < T) ? x[((long long)b * T + tt) * F + ff] : 0.f;
        }
    float o[8];
#pragma unroll
    for (int co = 0; co < 8; co++) {
        float s = sb[co];
#pragma unroll
        for (int kt = 0; kt < 3; kt++)
#pragma unroll
            for (int kf = 0; kf < 3; kf++) s += sw[(co * 3 + kt) * 3 + kf] * xin[kt][kf];
        o[co] = DSWISH ? dswish(s) : swoosh_r(s);
    }
    float4* yo = reinterpret_cast<float4*>(y + i * 8);
    yo[0] = make_float4(o[0], o[1], o[2], o[3]);
    yo[1] = make_float4(o[4], o[5], o[6], o[7]);
}

// ---- ConvNeXt depthwise 7x7, pad 3, NHWC, weights [49][C]
__global__ void k_dwconv7x7(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                            float* __restrict__ y, int B, int Tin, int T, int tpad, int F, int C) {
    int C4 = C / 4;
    long long n = (long long)B * T * F * C4;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int c = (int)(i % C4) * 4;
    long long p = i / C4;
    int f = (int)(p % F);
    long long bt = p / F;
    int t = (int)(bt % T), b = (int)(bt / T);
    float4 s = *reinterpret_cast<const float4*>(bias + c);
    for (int kt = 0; kt < 7; kt++) {
        int tt = t + kt - tpad;
        if (tt < 0 || tt >= Tin) continue;
        for (int kf = 0; kf < 7; kf++) {
            int ff = f + kf - 3;
            if (ff < 0 || ff >= F) continue;
            float4 xv = *reinterpret_cast<const float4*>(x + (((long long)b * Tin + tt) * F + ff) * C + c);
            float4 wv = *reinterpret_cast<const float4*>(w + (kt * 7 + kf) * C + c);
            s.x += wv.x * xv.x; s.y += wv.y * xv.y; s.z += wv.z * xv.z; s.w += wv.w * xv.w;
        }
    }
    *reinterpret_cast<float4*>(y + p * C + c) = s;
}

// The 7x7 taps of one chunk of DW7_TT output frames for one (bin, channel quad): `col` points at the lane's cell of patch row
// slot 0 (bin fi + 0, quad cq); patch row r of the chunk sits in slot (s0 + r) mod DW7_RING, rows `rowf` floats apart; `wl` points
// at the lane's quad of the workgroup's taps in LDS ([49][32]).  Explicit packed FMAs (v_pk_fma_f32): left to itself the compiler
// fused the x/y halves and split the z/w halves into multiplies, adds and register moves (about twice the instructions), and
// waited for each tap's weights from global memory right in front of its use.  Both kernels below call this, so they agree bit for bit.
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int DW7_TT = 8, DW7_CG = 32, DW7_RING = 22;
__device__ __forceinline__ void dw7_stage_taps(float* wl, const float* __restrict__ w, int C, int c0, int tid) {
    for (int i = tid; i < 49 * 8; i += 192)
        *reinterpret_cast<float4*>(wl + (i >> 3) * 32 + 4 * (i & 7)) = *reinterpret_cast<const float4*>(w + (long long)(i >> 3) * C + c0 + 4 * (i & 7));
}
__device__ __forceinline__ void dw7_chunk(const float* col, int rowf, int s0, const float* wl, f32x2 (&alo)[DW7_TT], f32x2 (&ahi)[DW7_TT]) {
#pragma unroll 1  // fully unrolled, the compiler hoists all 147 loads and spills
    for (int kf = 0; kf < 7; kf++) {
        f32x2 wlo[7], whi[7];
#pragma unroll
        for (int kt = 0; kt < 7; kt++) {
            const float4 t = *reinterpret_cast<const float4*>(wl + (kt * 7 + kf) * 32);
            wlo[kt] = f32x2{t.x, t.y};
            whi[kt] = f32x2{t.z, t.w};
        }
        // all 14 rows of the column in flight before the first FMA (one row at a time, each read's latency sat in front of its FMAs)
        float4 xr[DW7_TT + 6];
#pragma unroll
        for (int r = 0; r < DW7_TT + 6; r++) {
            int slot = s0 + r;
            slot = slot >= DW7_RING ? slot - DW7_RING : slot;
            xr[r] = *reinterpret_cast<const float4*>(col + kf * 32 + slot * rowf);
        }
#pragma unroll
        for (int r = 0; r < DW7_TT + 6; r++) {
            const f32x2 xlo{xr[r].x, xr[r].y}, xhi{xr[r].z, xr[r].w};
#pragma unroll
            for (int kt = 0; kt < 7; kt++) {
                const int to = r - kt;  // output frame fed by patch row r through tap kt
                if (to >= 0 && to < DW7_TT) {
                    alo[to] = __builtin_elementwise_fma(wlo[kt], xlo, alo[to]);
                    ahi[to] = __builtin_elementwise_fma(whi[kt], xhi, ahi[to]);
                }
            }
        }
    }
}

// LDS-tiled form: one workgroup = 8 output frames x all F bins x 32 channels.  The (8+6) x (F+6) x 32 input patch (zero
// halo) is staged once; a thread owns one (bin, channel quad) column of 8 outputs and slides down the patch, so every staged
// value is read 7 times from LDS instead of 49 times from L2 (the per-pixel kernel above fetched 11x the tensor: 1.78 GB per
// launch against 157 MB, profiles/r01_v2_pmc_fetch_summary.csv).
__global__ __launch_bounds__(192) void k_dwconv7x7_tiled(const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ bias, float* __restrict__ y, int Tin, int T,
                                                         int tpad, int F, int C) {
    extern __shared__ __attribute__((aligned(16))) float xs[];  // [DW7_TT + 6][F + 6][32] + taps [49][32]
    const int tid = threadIdx.x, cq = tid & 7, fi = tid >> 3;
    const int c0 = blockIdx.x * DW7_CG, t0 = blockIdx.y * DW7_TT, b = blockIdx.z;
    const int FP = F + 6, ncell = (DW7_TT + 6) * FP * 8;
    float* wl = xs + (DW7_TT + 6) * FP * 32;
    dw7_stage_taps(wl, w, C, c0, tid);
    // five patch cells per lane in flight (one at a time, the staging was a chain of ~15 memory latencies per workgroup)
    for (int base = tid; base < ncell; base += 5 * 192) {
        float4 v[5];
#pragma unroll
        for (int u = 0; u < 5; u++) {
            const int idx = base + u * 192;
            const int q = idx & 7, cell = idx >> 3;
            const int fp = cell % FP, r = cell / FP;
            const int f = fp - 3, tt = t0 + r - tpad;
            v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (idx < ncell && f >= 0 && f < F && tt >= 0 && tt < Tin)
                v[u] = *reinterpret_cast<const float4*>(x + (((long long)b * Tin + tt) * F + f) * C + c0 + 4 * q);
        }
#pragma unroll
        for (int u = 0; u < 5; u++) {
            const int idx = base + u * 192;
            if (idx < ncell) *reinterpret_cast<float4*>(xs + (long long)(idx >> 3) * 32 + 4 * (idx & 7)) = v[u];
        }
    }
    __syncthreads();
    if (fi >= F) return;
    const int c = c0 + 4 * cq;
    const float4 bv = *reinterpret_cast<const float4*>(bias + c);
    f32x2 alo[DW7_TT], ahi[DW7_TT];
#pragma unroll
    for (int i = 0; i < DW7_TT; i++) {
        alo[i] = f32x2{bv.x, bv.y};
        ahi[i] = f32x2{bv.z, bv.w};
    }
    dw7_chunk(xs + fi * 32 + 4 * cq, FP * 32, 0, wl + 4 * cq, alo, ahi);
#pragma unroll
    for (int i = 0; i < DW7_TT; i++) {
        const int t = t0 + i;
        if (t < T) *reinterpret_cast<float4*>(y + (((long long)b * T + t) * F + fi) * C + c) = make_float4(alo[i].x, alo[i].y, ahi[i].x, ahi[i].y);
    }
}

// Sliding form for long inputs: a workgroup walks DW7_TR output frames of one (utterance, 32-channel group) in chunks of 8, keeping
// the 14 patch rows of the current chunk and the 8 new rows of the next one in a 22-row LDS ring.  The new rows arrive by LDS-DMA
// (global_load_lds_dwordx4: a row's F x 32 floats are contiguous by lane in LDS, 128 B per bin in memory) while the current chunk
// is computed, so the input is fetched (TR + 6) / TR times instead of 14 / 8 (274 MB against a 157 MB tensor per launch of the
// tiled kernel, profiles/r01_v5_pmc_fetch_summary.csv) and no fetch latency sits in front of a chunk.  Wave w issues part w of every
// row (cells 64 w .. 64 w + 63 of its F x 8 float4 cells); the halo bins are zeroed once and never written again; a row outside
// the input is zeroed by plain stores when its ring slot comes up.  Sums run in the tiled kernel's order: bit-identical results.
__global__ __launch_bounds__(192) void k_dwconv7x7_slide(const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ bias, float* __restrict__ y, int Tin, int T,
                                                         int tpad, int F, int C, int TR) {
    extern __shared__ __attribute__((aligned(16))) float xs[];  // [DW7_RING][F + 6][32] + taps [49][32]
    const int tid = threadIdx.x, cq = tid & 7, fi = tid >> 3, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c0 = blockIdx.x * DW7_CG, tbeg = blockIdx.y * TR, b = blockIdx.z;
    const int tend = min(T, tbeg + TR), nch = (tend - tbeg + DW7_TT - 1) / DW7_TT;
    const int FP = F + 6, rowf = FP * 32;
    // halo bins of every ring row
    for (int i = tid; i < DW7_RING * 6 * 8; i += 192) {
        const int q = i & 7, hb = (i >> 3) % 6, r = i / 48;
        *reinterpret_cast<float4*>(xs + r * rowf + (hb < 3 ? hb : F + hb) * 32 + 4 * q) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float* wl = xs + DW7_RING * rowf;
    dw7_stage_taps(wl, w, C, c0, tid);
    const unsigned lds_base = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)xs;
    const int ci = 64 * wv + lane;           // this lane's cell of a row: bin ci / 8, channel quad ci % 8
    const bool cell_ok = ci < F * 8;
    const unsigned voff = (unsigned)(((lane >> 3) * C + (lane & 7) * 4) * 4);
    // patch row pr (0 = input frame tbeg - tpad) -> ring slot pr % DW7_RING
    auto fetch_row = [&](int pr) {
        const int tt = tbeg - tpad + pr, slot = pr % DW7_RING;
        if (tt >= 0 && tt < Tin) {
            const unsigned long long src = (unsigned long long)(x + (((long long)b * Tin + tt) * F + 8 * wv) * C + c0);
            const unsigned m0v = lds_base + (unsigned)((slot * rowf + 96 + 256 * wv) * 4);
            if (cell_ok) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(src), "s"(m0v) : "memory");
        } else if (cell_ok) {
            *reinterpret_cast<float4*>(xs + slot * rowf + 96 + 4 * ci) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    for (int pr = 0; pr < DW7_TT + 6; pr++) fetch_row(pr);
    const int c = c0 + 4 * cq;
    const float4 bv = *reinterpret_cast<const float4*>(bias + c);
    const bool mine = fi < F;
    const float* lanep = xs + (mine ? fi : 0) * 32 + 4 * cq;
    for (int n = 0; n < nch; n++) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this chunk's rows have landed (and the previous chunk's stores retired)
        __syncthreads();                                    // ... for every wave; and every wave is done with chunk n - 1
        if (n + 1 < nch)
            for (int k = 0; k < DW7_TT; k++) fetch_row(DW7_TT * (n + 1) + 6 + k);
        if (mine) {
            f32x2 alo[DW7_TT], ahi[DW7_TT];
#pragma unroll
            for (int i = 0; i < DW7_TT; i++) {
                alo[i] = f32x2{bv.x, bv.y};
                ahi[i] = f32x2{bv.z, bv.w};
            }
            dw7_chunk(lanep, rowf, (DW7_TT * n) % DW7_RING, wl + 4 * cq, alo, ahi);
            const int t0 = tbeg + DW7_TT * n;
#pragma unroll
            for (int i = 0; i < DW7_TT; i++)
                if (t0 + i < tend)
                    *reinterpret_cast<float4*>(y + (((long long)b * T + t0 + i) * F + fi) * C + c) = make_float4(alo[i].x, alo[i].y, ahi[i].x, ahi[i].y);
        }
    }
}

// ---- BiasNorm (+ optional bypass): one wave per row, row kept in registers
//   scale = (mean((x-b)^2))^-0.5 * exp(log_scale);  y = x*scale
//   with orig: y = orig + (x*scale - orig) * bscale
template <bool BYPASS>
__global__ void k_biasnorm(const float* __restrict__ x, const float* __restrict__ orig, const float* __restrict__ nbias,
                           const float* __restrict__ log_scale, const float* __restrict__ bscale, float* __restrict__ y,
                           int M, int D) {
    int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= M) return;
    int lane = threadIdx.x & 63;
    const float* xr = x + (long long)row * D;
    float4 v[4];  // D <= 1024
    int nq = D >> 2;
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        int q = lane + 64 * j;
        if (q < nq) {
            v[j] = *reinterpret_cast<const float4*>(xr + 4 * q);
            float4 b = *reinterpret_cast<const float4*>(nbias + 4 * q);
            float a0 = v[j].x - b.x, a1 = v[j].y - b.y, a2 = v[j].z - b.z, a3 = v[j].w - b.w;
            s += a0 * a0 + a1 * a1 + a2 * a2 + a3 * a3;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    float sc = (1.0f / sqrtf(s / (float)D)) * expf(log_scale[0]);
#pragma unroll
    for (int j = 0; j < 4; j++) {
        int q = lane + 64 * j;
        if (q < nq) {
            float4 o = make_float4(v[j].x * sc, v[j].y * sc, v[j].z * sc, v[j].w * sc);
            if (BYPASS) {
                float4 g = *reinterpret_cast<const float4*>(orig + (long long)row * D + 4 * q);
                float4 bs = *reinterpret_cast<const float4*>(bscale + 4 * q);
                o.x = g.x + (o.x - g.x) * bs.x; o.y = g.y + (o.y - g.y) * bs.y;
                o.z = g.z + (o.z - g.z) * bs.z; o.w = g.w + (o.w - g.w) * bs.w;
            }
            *reinterpret_cast<float4*>(y + (long long)row * D + 4 * q) = o;
        }
    }
}

// The last layer of a stack that runs at the input rate, in front of a downsampled stack: BiasNorm + bypass (k_biasnorm<true>'s
// arithmetic, row by row) AND the next stack's SimpleDownsample of the result (k_downsample's weighted sum, same order) in one launch.
// One wave per (stream, downsampled frame): it walks the group's ds2 rows, writes each (y may be `orig`: a row is read whole before it
// is written), and keeps the running sum in registers; a frame past T repeats the last row from registers (never re-read: it has been
// overwritten in place).  xd2 rows are D2 wide: zero-extended (truncated) like convert_channels.
__global__ void k_biasnorm_bypass_downsample(const float* __restrict__ x, const float* __restrict__ orig, const float* __restrict__ nbias,
                                             const float* __restrict__ log_scale, const float* __restrict__ bscale, float* __restrict__ y,
                                             const float* __restrict__ bias2, float* __restrict__ xd2, int B, int T, int D, int ds2, int D2) {
    const int Td2 = (T + ds2 - 1) / ds2;
    const int g = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (g >= B * Td2) return;
    const int lane = threadIdx.x & 63, b = g / Td2, t2 = g - b * Td2;
    float wgt[16], mx = -INFINITY, sum = 0.f;
    for (int k = 0; k < ds2; k++) mx = fmaxf(mx, bias2[k]);
    for (int k = 0; k < ds2; k++) { wgt[k] = expf(bias2[k] - mx); sum += wgt[k]; }
    for (int k = 0; k < ds2; k++) wgt[k] /= sum;
    const int nq = D >> 2, nq2 = D2 >> 2;
    const float es = expf(log_scale[0]);
    float4 acc[4], last[4];
#pragma unroll
    for (int j = 0; j < 4; j++) acc[j] = last[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k = 0; k < ds2; k++) {
        const int t = t2 * ds2 + k;
        if (t < T) {
            const long long row = (long long)b * T + t;
            const float* xr = x + row * D;
            float4 v[4];
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int q = lane + 64 * j;
                if (q < nq) {
                    v[j] = *reinterpret_cast<const float4*>(xr + 4 * q);
                    const float4 bq = *reinterpret_cast<const float4*>(nbias + 4 * q);
                    const float a0 = v[j].x - bq.x, a1 = v[j].y - bq.y, a2 = v[j].z - bq.z, a3 = v[j].w - bq.w;
                    s += a0 * a0 + a1 * a1 + a2 * a2 + a3 * a3;
                }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
            const float sc = (1.0f / sqrtf(s / (float)D)) * es;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int q = lane + 64 * j;
                if (q < nq) {
                    float4 o = make_float4(v[j].x * sc, v[j].y * sc, v[j].z * sc, v[j].w * sc);
                    const float4 gq = *reinterpret_cast<const float4*>(orig + row * D + 4 * q);
                    const float4 bs = *reinterpret_cast<const float4*>(bscale + 4 * q);
                    o.x = gq.x + (o.x - gq.x) * bs.x; o.y = gq.y + (o.y - gq.y) * bs.y;
                    o.z = gq.z + (o.z - gq.z) * bs.z; o.w = gq.w + (o.w - gq.w) * bs.w;
                    *reinterpret_cast<float4*>(y + row * D + 4 * q) = o;
                    last[j] = o;
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            acc[j].x += last[j].x * wgt[k]; acc[j].y += last[j].y * wgt[k]; acc[j].z += last[j].z * wgt[k]; acc[j].w += last[j].w * wgt[k];
        }
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int q = lane + 64 * j;
        if (q < nq2) *reinterpret_cast<float4*>(xd2 + ((long long)b * Td2 + t2) * D2 + 4 * q) = q < nq ? acc[j] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

// ---- BasicNorm (Conformer): y = x * (mean(x^2) + exp(log_eps))^-0.5, one wave per row
__global__ void k_basicnorm(const float* __restrict__ x, const float* __restrict__ log_eps, float* __restrict__ y, int M, int D) {
    int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= M) return;
    int lane = threadIdx.x & 63;
    const float* xr = x + (long long)row * D;
    float4 v[4];  // D <= 1024
    int nq = D >> 2;
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        int q = lane + 64 * j;
        if (q < nq) {
            v[j] = *reinterpret_cast<const float4*>(xr + 4 * q);
            s += v[j].x * v[j].x + v[j].y * v[j].y + v[j].z * v[j].z + v[j].w * v[j].w;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    float sc = 1.0f / sqrtf(s / (float)D + expf(log_eps[0]));
#pragma unroll
    for (int j = 0; j < 4; j++) {
        int q = lane + 64 * j;
        if (q < nq) *reinterpret_cast<float4*>(y + (long long)row * D + 4 * q) = make_float4(v[j].x * sc, v[j].y * sc, v[j].z * sc, v[j].w * sc);
    }
}

// BypassModule: y = orig + (x - orig) * scale[d]
__global__ void k_bypass(const float* __restrict__ orig, const float* __restrict__ x, const float* __restrict__ scale,
                         float* __restrict__ y, long long n4, int D4) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    float4 g = reinterpret_cast<const float4*>(orig)[i], v = reinterpret_cast<const float4*>(x)[i];
    float4 s = reinterpret_cast<const float4*>(scale)[i % D4];
    reinterpret_cast<float4*>(y)[i] = make_float4(g.x + (v.x - g.x) * s.x, g.y + (v.y - g.y) * s.y,
                                                  g.z + (v.z - g.z) * s.z, g.w + (v.w - g.w) * s.w);
}


// y[m, d] = x[m, d] * sigmoid(x[m, D + d])       x: [M, 2D]
__global__ void k_glu(const float* __restrict__ x, float* __restrict__ y, long long n4, int D4) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    long long m = i / D4;
    int q = (int)(i % D4);
    const float4* xr = reinterpret_cast<const float4*>(x) + m * 2 * D4;
    float4 a = xr[q], s = xr[D4 + q];
    reinterpret_cast<float4*>(y)[i] = make_float4(a.x * sigm(s.x), a.y * sigm(s.y), a.z * sigm(s.z), a.w * sigm(s.w));
}
// y[m, c] = x[m, Hc + c] * tanh(x[m, c])         x: [M, 3Hc]
__global__ void k_tanh_gate(const float* __restrict__ x, float* __restrict__ y, long long n4, int H4) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    long long m = i / H4;
    int q = (int)(i % H4);
    const float4* xr = reinterpret_cast<const float4*>(x) + m * 3 * H4;
    float4 s = xr[q], a = xr[H4 + q];
    reinterpret_cast<float4*>(y)[i] = make_float4(a.x * fast_tanh(s.x), a.y * fast_tanh(s.y), a.z * fast_tanh(s.z), a.w * fast_tanh(s.w));
}
// a[m, n] *= x[m*ldx + col0 + n]
__global__ void k_mul_cols(float* __restrict__ a, const float* __restrict__ x, int ldx, int col0, long long n4, int N4) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    long long m = i / N4;
    int q = (int)(i % N4);
    float4 v = reinterpret_cast<float4*>(a)[i];
    float4 s = *reinterpret_cast<const float4*>(x + m * ldx + col0 + 4 * q);
    reinterpret_cast<float4*>(a)[i] = make_float4(v.x * s.x, v.y * s.y, v.z * s.z, v.w * s.w);
}

// ---- ConvolutionModule: GLU + depthwise Conv1d over time (zero pad K/2) + bias + SwooshR
//  x2: [B,T,2D] (in_proj output: value | gate), y: [B,T,D], w: [K][D].
//  One workgroup = 4 waves = 32 output frames x 256 channels.  The GLU'd input strip
//  (32 + K - 1 frames) is staged once in LDS with 1 KB coalesced row loads; lane = channel
//  quad, so every LDS read of a wave is one conflict-free 1 KB row segment.
//  DW_TT = outputs per thread (8, or 4 / 2 when 32-frame workgroups would leave CUs idle: at B x T = 4064 rows x 512 channels there
//  are 256 of them, one per CU, and the launch is one workgroup's stage -> compute -> store chain); 4 DW_TT output frames per workgroup
template <bool DSWISH, bool GLU = true, int DW_TT = 8>
__global__ __launch_bounds__(256) void k_glu_dwconv1d(const float* __restrict__ x2, const float* __restrict__ w,
                                                      const float* __restrict__ bias, float* __restrict__ y, int B, int T,
                                                      int D, int K) {
    extern __shared__ __attribute__((aligned(16))) float sx[];  // [DW_ROWS + K - 1][256]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = blockIdx.x * 256 + lane * 4;
    constexpr int DW_ROWS = 4 * DW_TT;
    const int t0 = blockIdx.y * DW_ROWS, b = blockIdx.z;
    const int half = K >> 1, nrows = DW_ROWS + K - 1;
    const bool cok = c < D;
    // K <= 32 (every recipe: 31, 15): all taps of the lane's channel quad are requested before the staging loop, so that their
    // latency lies under the staging instead of in front of every group of 8 taps (4 x ~0.7 us of a ~10 us launch)
    const bool pre = K <= 32;
    float4 wp[32];
    if (pre) {
#pragma unroll
        for (int q = 0; q < 32; q++)
            wp[q] = (cok && q < K) ? *reinterpret_cast<const float4*>(w + q * D + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // four rows (eight 16-byte loads) in flight per lane: one row at a time made the staging a chain of ~16 memory latencies
    for (int r0 = wave; r0 < nrows; r0 += 16) {
        float4 a[4], sg[4];
        bool ok[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int r = r0 + 4 * i, u = t0 - half + r;
            ok[i] = cok && r < nrows && u >= 0 && u < T;
            if (ok[i]) {
                const float* row = x2 + ((long long)b * T + u) * (GLU ? 2 * D : D);  // !GLU: the gate has been applied upstream
                a[i] = *reinterpret_cast<const float4*>(row + c);
                if (GLU) sg[i] = *reinterpret_cast<const float4*>(row + D + c);
            }
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int r = r0 + 4 * i;
            if (r < nrows) {
                float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
                if (ok[i]) g = GLU ? make_float4(a[i].x * sigm(sg[i].x), a[i].y * sigm(sg[i].y), a[i].z * sigm(sg[i].z), a[i].w * sigm(sg[i].w)) : a[i];
                *reinterpret_cast<float4*>(sx + r * 256 + lane * 4) = g;
            }
        }
    }
    __syncthreads();
    if (!cok) return;
    f32x2 alo[DW_TT], ahi[DW_TT];  // explicit packed FMAs (see dw7_chunk)
    const float4 bv = *reinterpret_cast<const float4*>(bias + c);
#pragma unroll
    for (int i = 0; i < DW_TT; i++) {
        alo[i] = f32x2{bv.x, bv.y};
        ahi[i] = f32x2{bv.z, bv.w};
    }
    const float* base = sx + (wave * DW_TT) * 256 + lane * 4;
    if (pre) {
#pragma unroll
        for (int q = 0; q < 32; q++) {
            if (q < K) {
                const f32x2 wlo{wp[q].x, wp[q].y}, whi{wp[q].z, wp[q].w};
#pragma unroll
                for (int i = 0; i < DW_TT; i++) {
                    const float4 xv = *reinterpret_cast<const float4*>(base + (i + q) * 256);
                    alo[i] = __builtin_elementwise_fma(wlo, f32x2{xv.x, xv.y}, alo[i]);
                    ahi[i] = __builtin_elementwise_fma(whi, f32x2{xv.z, xv.w}, ahi[i]);
                }
            }
        }
    }
    // longer kernels: taps in groups of 8, the 8 weight loads in flight together
    for (int kb = 0; !pre && kb < K; kb += 8) {
        float4 wv[8];
#pragma unroll
        for (int q = 0; q < 8; q++)
            wv[q] = kb + q < K ? *reinterpret_cast<const float4*>(w + (kb + q) * D + c) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int q = 0; q < 8; q++) {
            if (kb + q < K) {
                const f32x2 wlo{wv[q].x, wv[q].y}, whi{wv[q].z, wv[q].w};
#pragma unroll
                for (int i = 0; i < DW_TT; i++) {
                    const float4 xv = *reinterpret_cast<const float4*>(base + (i + kb + q) * 256);
                    alo[i] = __builtin_elementwise_fma(wlo, f32x2{xv.x, xv.y}, alo[i]);
                    ahi[i] = __builtin_elementwise_fma(whi, f32x2{xv.z, xv.w}, ahi[i]);
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < DW_TT; i++) {
        const int t = t0 + wave * DW_TT + i;
        if (t < T)
            *reinterpret_cast<float4*>(y + ((long long)b * T + t) * D + c) =
                DSWISH ? make_float4(dswish(alo[i].x), dswish(alo[i].y), dswish(ahi[i].x), dswish(ahi[i].y))
                       : make_float4(swoosh_r(alo[i].x), swoosh_r(alo[i].y), swoosh_r(ahi[i].x), swoosh_r(ahi[i].y));
    }
}

// ---- SimpleDownsample: softmax(bias)-weighted sum of ds frames, last frame repeated
// x rows are Din4 float4 wide: a narrower (wider) input is zero-extended (truncated) to D4 on the fly (convert_channels)
__global__ void k_downsample(const float* __restrict__ x, const float* __restrict__ bias, float* __restrict__ y, int B,
                             int T, int Td, int D4, int ds, int Din4) {
    long long n = (long long)B * Td * D4;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float wgt[16], mx = -INFINITY, sum = 0.f;
    for (int k = 0; k < ds; k++) mx = fmaxf(mx, bias[k]);
    for (int k = 0; k < ds; k++) { wgt[k] = expf(bias[k] - mx); sum += wgt[k]; }
    for (int k = 0; k < ds; k++) wgt[k] /= sum;
    int q = (int)(i % D4);
    long long bt = i / D4;
    int t = (int)(bt % Td), b = (int)(bt / Td);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k = 0; k < ds; k++) {
        int tt = min(t * ds + k, T - 1);
        float4 v = q < Din4 ? reinterpret_cast<const float4*>(x)[((long long)b * T + tt) * Din4 + q] : make_float4(0.f, 0.f, 0.f, 0.f);
        s.x += v.x * wgt[k]; s.y += v.y * wgt[k]; s.z += v.z * wgt[k]; s.w += v.w * wgt[k];
    }
    reinterpret_cast<float4*>(y)[i] = s;
}
// SimpleUpsample (repeat) + truncate + out_combiner bypass
// _get_full_dim_output + the final SimpleDownsample in one pass: column c of the full-width row comes from the LAST stack output that
// is wider than c (segments sorted by column: [col0, col1) from src with row width ld); y[b, t', c] = sum_k softmax(bias)_k full[b, ds t' + k, c]
__global__ void k_downsample_full(FullDimSegs segs, const float* __restrict__ bias, float* __restrict__ y, int B, int T, int Td, int D4, int ds) {
    long long n = (long long)B * Td * D4;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float wgt[16], mx = -INFINITY, sum = 0.f;
    for (int k = 0; k < ds; k++) mx = fmaxf(mx, bias[k]);
    for (int k = 0; k < ds; k++) { wgt[k] = expf(bias[k] - mx); sum += wgt[k]; }
    for (int k = 0; k < ds; k++) wgt[k] /= sum;
    int q = (int)(i % D4);
    long long bt = i / D4;
    int t = (int)(bt % Td), b = (int)(bt / Td);
    int sg = 0;
    while (sg + 1 < segs.n && 4 * q >= segs.col1[sg]) sg++;
    const float* src = segs.src[sg];
    const int ld4 = segs.ld[sg] >> 2;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool lazy = sg == 0 && segs.lz_orig != nullptr;   // the last stack's out_combiner on the fly (FullDimSegs)
    float4 sc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (lazy) sc = reinterpret_cast<const float4*>(segs.lz_scale)[q];
    const int Do4 = segs.lz_Do >> 2;
    for (int k = 0; k < ds; k++) {
        int tt = min(t * ds + k, T - 1);
        float4 v;
        if (lazy) {
            const float4 o = q < Do4 ? reinterpret_cast<const float4*>(segs.lz_orig)[((long long)b * T + tt) * Do4 + q] : make_float4(0.f, 0.f, 0.f, 0.f);
            const float4 u = reinterpret_cast<const float4*>(segs.lz_xd)[((long long)b * segs.lz_Td + tt / segs.lz_ds) * ld4 + q];
            v = make_float4(o.x + (u.x - o.x) * sc.x, o.y + (u.y - o.y) * sc.y, o.z + (u.z - o.z) * sc.z, o.w + (u.w - o.w) * sc.w);
        } else {
            v = reinterpret_cast<const float4*>(src)[((long long)b * T + tt) * ld4 + q];
        }
        s.x += v.x * wgt[k]; s.y += v.y * wgt[k]; s.z += v.z * wgt[k]; s.w += v.w * wgt[k];
    }
    reinterpret_cast<float4*>(y)[i] = s;
}
__global__ void k_upsample_combine(const float* __restrict__ orig, const float* __restrict__ xd,
                                   const float* __restrict__ scale, float* __restrict__ y, int B, int T, int Td, int D4,
                                   int ds, int Do4) {  // orig rows are Do4 float4 wide (zero-extended / truncated to D4)
    long long n = (long long)B * T * D4;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int q = (int)(i % D4);
    long long bt = i / D4;
    int t = (int)(bt % T), b = (int)(bt / T);
    float4 o = q < Do4 ? reinterpret_cast<const float4*>(orig)[bt * Do4 + q] : make_float4(0.f, 0.f, 0.f, 0.f);
    float4 u = reinterpret_cast<const float4*>(xd)[((long long)b * Td + t / ds) * D4 + q];
    float4 s = reinterpret_cast<const float4*>(scale)[q];
    reinterpret_cast<float4*>(y)[i] = make_float4(o.x + (u.x - o.x) * s.x, o.y + (u.y - o.y) * s.y,
                                                  o.z + (u.z - o.z) * s.z, o.w + (u.w - o.w) * s.w);
}
// One stack's out_combiner and the NEXT stack's SimpleDownsample in one pass (both stacks downsampled: four of the six stack seams of
// Zipformer2): thread (b, t2, q) forms the ds2 combined frames of its group -- y[b, t, q] = orig + (upsample(xd) - orig) * scale, the
// expression of k_upsample_combine -- writes them, and sums them with the next stack's softmax(bias2) weights into xd2[b, t2, q], in
// k_downsample's order (the last frame repeated past T; y is D4 float4 wide, xd2 D2_4: zero-extended / truncated like convert_channels).
__global__ void k_upsample_combine_downsample(const float* __restrict__ orig, const float* __restrict__ xd, const float* __restrict__ scale,
                                              float* __restrict__ y, const float* __restrict__ bias2, float* __restrict__ xd2, int B, int T,
                                              int Td, int D4, int ds, int Do4, int Td2, int D2_4, int ds2) {
    const int W4 = max(D4, D2_4);
    long long n = (long long)B * Td2 * W4;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float wgt[16], mx = -INFINITY, sum = 0.f;
    for (int k = 0; k < ds2; k++) mx = fmaxf(mx, bias2[k]);
    for (int k = 0; k < ds2; k++) { wgt[k] = expf(bias2[k] - mx); sum += wgt[k]; }
    for (int k = 0; k < ds2; k++) wgt[k] /= sum;
    const int q = (int)(i % W4);
    const long long bt = i / W4;
    const int t2 = (int)(bt % Td2), b = (int)(bt / Td2);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (q < D4) {
        const float4 sc = reinterpret_cast<const float4*>(scale)[q];
        for (int k = 0; k < ds2; k++) {
            const int t = t2 * ds2 + k, tt = min(t, T - 1);
            const long long row = (long long)b * T + tt;
            const float4 o = q < Do4 ? reinterpret_cast<const float4*>(orig)[row * Do4 + q] : make_float4(0.f, 0.f, 0.f, 0.f);
            const float4 u = reinterpret_cast<const float4*>(xd)[((long long)b * Td + tt / ds) * D4 + q];
            const float4 v = make_float4(o.x + (u.x - o.x) * sc.x, o.y + (u.y - o.y) * sc.y, o.z + (u.z - o.z) * sc.z, o.w + (u.w - o.w) * sc.w);
            if (t < T) reinterpret_cast<float4*>(y)[row * D4 + q] = v;
            s.x += v.x * wgt[k]; s.y += v.y * wgt[k]; s.z += v.z * wgt[k]; s.w += v.w * wgt[k];
        }
    }
    if (q < D2_4) reinterpret_cast<float4*>(xd2)[((long long)b * Td2 + t2) * D2_4 + q] = s;
}
// convert_num_channels: truncate / zero-pad
__global__ void k_convert_channels(const float* __restrict__ x, float* __restrict__ y, long long M, int Din4, int Dout4) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M * Dout4) return;
    long long m = i / Dout4;
    int q = (int)(i % Dout4);
    reinterpret_cast<float4*>(y)[i] = q < Din4 ? reinterpret_cast<const float4*>(x)[m * Din4 + q] : make_float4(0.f, 0.f, 0.f, 0.f);
}
__global__ void k_copy_cols(const float* __restrict__ x, int ldx, int xc, float* __restrict__ y, int ldy, int yc,
                            long long M, int n4) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M * n4) return;
    long long m = i / n4;
    int q = (int)(i % n4);
    *reinterpret_cast<float4*>(y + m * ldy + yc + 4 * q) = *reinterpret_cast<const float4*>(x + m * ldx + xc + 4 * q);
}

}  // namespace

#define LAUNCH(kernel, grid, block, ...)                                        \
    do {                                                                        \
        if (!ctx.dry) {                                                         \
            hipLaunchKernelGGL(kernel, grid, block, 0, ctx.stream, __VA_ARGS__); \
            K2_HIP(hipGetLastError());                                          \
        }                                                                       \
    } while (0)

void pad_logfloor(const Ctx& ctx, const float* packed, const long long* d_off, const long long* d_len, float* out, int B,
                  long long L) {
    dim3 grid(std::min(nblocks(L, 256), 512), B);
    LAUNCH(k_pad_logfloor, grid, dim3(256), packed, d_off, d_len, out, L);
}
void pad_logfloor_dense(const Ctx& ctx, const float* feats, long long n_each, float* out, int B, long long L) {
    dim3 grid(std::min(nblocks(L, 256), 512), B);
    LAUNCH(k_pad_logfloor_dense, grid, dim3(256), feats, n_each, out, L);
}
void conv0_swoosh(const Ctx& ctx, const float* x, const float* w, const float* b, float* y, int B, int T, int F) {
    long long n = (long long)B * (T - 2) * F;
    LAUNCH((k_conv0<0, false>), dim3(nblocks(n, 256)), dim3(256), x, w, b, y, B, T, F);
}
void conv0_nopad_dswish(const Ctx& ctx, const float* x, const float* w, const float* b, float* y, int B, int T, int F) {
    long long n = (long long)B * (T - 2) * F;
    LAUNCH((k_conv0<0, true>), dim3(nblocks(n, 256)), dim3(256), x, w, b, y, B, T, F);
}
void conv0_pad1_dswish(const Ctx& ctx, const float* x, const float* w, const float* b, float* y, int B, int T, int F) {
    long long n = (long long)B * T * F;
    LAUNCH((k_conv0<1, true>), dim3(nblocks(n, 256)), dim3(256), x, w, b, y, B, T, F);
}
void basicnorm(const Ctx& ctx, const float* x, const float* log_eps, float* y, int M, int D) {
    K2_REQUIRE(D % 4 == 0 && D <= 1024, "basicnorm: D=%d unsupported", D);
    LAUNCH(k_basicnorm, dim3(nblocks(M, 4)), dim3(256), x, log_eps, y, M, D);
}
void dwconv7x7(const Ctx& ctx, const float* x, const float* w_kc, const float* b, float* y, int B, int Tin, int T, int tpad,
               int F, int C) {
    ctx.add_flops(0.0, 2.0 * B * T * (double)F * C * 49, 0);
    if (C % DW7_CG == 0 && F <= 24 && !tunables().dw7_tiled && T >= 48 && (long long)F * C * 4 < (1ll << 31)) {
        if (ctx.dry) return;
        const int TR = T >= 256 ? 64 : 32;
        size_t lds = sizeof(float) * ((size_t)DW7_RING * (F + 6) * 32 + 49 * 32);
        static LdsAttrOnce lds_attr;
        lds_attr.ensure(k_dwconv7x7_slide, 96 * 1024);
        hipLaunchKernelGGL(k_dwconv7x7_slide, dim3(C / DW7_CG, cdiv(T, TR), B), dim3(192), lds, ctx.stream, x, w_kc, b, y, Tin, T, tpad, F,
                           C, TR);
        K2_HIP(hipGetLastError());
        return;
    }
    if (C % DW7_CG == 0 && F <= 24) {
        if (ctx.dry) return;
        size_t lds = sizeof(float) * ((size_t)(DW7_TT + 6) * (F + 6) * 32 + 49 * 32);
        hipLaunchKernelGGL(k_dwconv7x7_tiled, dim3(C / DW7_CG, cdiv(T, DW7_TT), B), dim3(192), lds, ctx.stream, x, w_kc, b, y, Tin, T,
                           tpad, F, C);
        K2_HIP(hipGetLastError());
        return;
    }
    long long n = (long long)B * T * F * (C / 4);
    LAUNCH(k_dwconv7x7, dim3(nblocks(n, 256)), dim3(256), x, w_kc, b, y, B, Tin, T, tpad, F, C);
}
void biasnorm(const Ctx& ctx, const float* x, const float* bias, const float* log_scale, float* y, int M, int D) {
    K2_REQUIRE(D % 4 == 0 && D <= 1024, "biasnorm: D=%d unsupported", D);
    LAUNCH((k_biasnorm<false>), dim3(nblocks(M, 4)), dim3(256), x, nullptr, bias, log_scale, nullptr, y, M, D);
}
void biasnorm_bypass(const Ctx& ctx, const float* x, const float* orig, const float* nbias, const float* log_scale,
                     const float* scale, float* y, int M, int D) {
    K2_REQUIRE(D % 4 == 0 && D <= 1024, "biasnorm: D=%d unsupported", D);
    LAUNCH((k_biasnorm<true>), dim3(nblocks(M, 4)), dim3(256), x, orig, nbias, log_scale, scale, y, M, D);
}
void biasnorm_bypass_downsample(const Ctx& ctx, const float* x, const float* orig, const float* nbias, const float* log_scale, const float* scale,
                                float* y, const float* bias2, float* xd2, int B, int T, int D, int ds2, int D2) {
    K2_REQUIRE(D % 4 == 0 && D <= 1024 && D2 % 4 == 0 && D2 <= 1024 && ds2 >= 1 && ds2 <= 16, "biasnorm + downsample: D=%d D2=%d ds=%d unsupported", D, D2, ds2);
    const int Td2 = (T + ds2 - 1) / ds2;
    LAUNCH(k_biasnorm_bypass_downsample, dim3(nblocks((long long)B * Td2, 4)), dim3(256), x, orig, nbias, log_scale, scale, y, bias2, xd2, B, T, D,
           ds2, D2);
}
void bypass(const Ctx& ctx, const float* orig, const float* x, const float* scale, float* y, int M, int D) {
    long long n4 = (long long)M * D / 4;
    LAUNCH(k_bypass, dim3(nblocks(n4, 256)), dim3(256), orig, x, scale, y, n4, D / 4);
}
void glu_sigmoid(const Ctx& ctx, const float* x, float* y, int M, int D) {
    long long n4 = (long long)M * D / 4;
    LAUNCH(k_glu, dim3(nblocks(n4, 256)), dim3(256), x, y, n4, D / 4);
}
void tanh_gate(const Ctx& ctx, const float* x, float* y, int M, int Hc) {
    long long n4 = (long long)M * Hc / 4;
    LAUNCH(k_tanh_gate, dim3(nblocks(n4, 256)), dim3(256), x, y, n4, Hc / 4);
}
void mul_cols(const Ctx& ctx, float* a, const float* x, int ldx, int col0, int M, int N) {
    long long n4 = (long long)M * N / 4;
    LAUNCH(k_mul_cols, dim3(nblocks(n4, 256)), dim3(256), a, x, ldx, col0, n4, N / 4);
}
// outputs per thread by grid size: 32-frame workgroups unless they would number fewer than ~2 per CU
static int dw1d_tt(int B, int T, int D) {
    const long long g8 = (long long)cdiv(D, 256) * cdiv(T, 32) * B;
    const int force = tunables().dw1d_tt;
    if (force == 8 || force == 4 || force == 2) return force;
    return g8 >= 512 ? 8 : 4;  // (2: measured slower everywhere)
}
template <bool DSWISH, bool GLU, int TT>
static void launch_dw1d(const Ctx& ctx, const float* x, const float* w_kd, const float* b, float* y, int B, int T, int D, int K) {
    size_t lds = sizeof(float) * (size_t)(4 * TT + K - 1) * 256;
    dim3 grid(cdiv(D, 256), cdiv(T, 4 * TT), B);
    static LdsAttrOnce lds_attr;
    lds_attr.ensure(k_glu_dwconv1d<DSWISH, GLU, TT>, 128 * 1024);
    hipLaunchKernelGGL((k_glu_dwconv1d<DSWISH, GLU, TT>), grid, dim3(256), lds, ctx.stream, x, w_kd, b, y, B, T, D, K);
    K2_HIP(hipGetLastError());
}
template <bool DSWISH, bool GLU>
static void glu_dwconv1d_any(const Ctx& ctx, const float* x2, const float* w_kd, const float* b, float* y, int B, int T, int D,
                             int K) {
    K2_REQUIRE(D % 4 == 0, "dwconv1d: D=%d", D);
    ctx.add_flops(0.0, 2.0 * B * T * (double)D * K, 0);
    if (ctx.dry) return;
    // (Round 5 built the same convolution on tall strips -- 64 channels x 128 / 64 frames per workgroup, 1.23x / 1.47x staged input
    // instead of this kernel's 2.9x on 16-frame workgroups -- and measured it slower: 11.6 us per launch at one 128-frame workgroup per
    // CU, 10.0 us at two 64-frame ones, against 9.0 here (profiles/r05_v1 / r05_v2_offline_kernel_stats.csv).  The input was written by
    // the in_proj GEMM a launch earlier and sits in L2 / the memory-side cache: the launch is not bound by the bytes it re-reads.  Removed.)
    switch (dw1d_tt(B, T, D)) {
        case 8: launch_dw1d<DSWISH, GLU, 8>(ctx, x2, w_kd, b, y, B, T, D, K); break;
        case 4: launch_dw1d<DSWISH, GLU, 4>(ctx, x2, w_kd, b, y, B, T, D, K); break;
        default: launch_dw1d<DSWISH, GLU, 2>(ctx, x2, w_kd, b, y, B, T, D, K); break;
    }
}
void dwconv1d_swoosh(const Ctx& ctx, const float* x, const float* w_kd, const float* b, float* y, int B, int T, int D, int K) {
    glu_dwconv1d_any<false, false>(ctx, x, w_kd, b, y, B, T, D, K);
}
void glu_dwconv1d_swoosh(const Ctx& ctx, const float* x2, const float* w_kd, const float* b, float* y, int B, int T, int D,
                          int K) {
    glu_dwconv1d_any<false, true>(ctx, x2, w_kd, b, y, B, T, D, K);
}
void glu_dwconv1d_dswish(const Ctx& ctx, const float* x2, const float* w_kd, const float* b, float* y, int B, int T, int D,
                          int K) {
    glu_dwconv1d_any<true, true>(ctx, x2, w_kd, b, y, B, T, D, K);
}
void downsample(const Ctx& ctx, const float* x, const float* bias, float* y, int B, int T, int D, int ds, int Din) {
    int Td = (T + ds - 1) / ds;
    long long n = (long long)B * Td * (D / 4);
    LAUNCH(k_downsample, dim3(nblocks(n, 256)), dim3(256), x, bias, y, B, T, Td, D / 4, ds, (Din > 0 ? Din : D) / 4);
}
void downsample_full(const Ctx& ctx, const FullDimSegs& segs, const float* bias, float* y, int B, int T, int D, int ds) {
    int Td = (T + ds - 1) / ds;
    long long n = (long long)B * Td * (D / 4);
    LAUNCH(k_downsample_full, dim3(nblocks(n, 256)), dim3(256), segs, bias, y, B, T, Td, D / 4, ds);
}
void upsample_combine(const Ctx& ctx, const float* orig, const float* xd, const float* scale, float* y, int B, int T,
                      int Td, int D, int ds, int Dorig) {
    long long n = (long long)B * T * (D / 4);
    LAUNCH(k_upsample_combine, dim3(nblocks(n, 256)), dim3(256), orig, xd, scale, y, B, T, Td, D / 4, ds, (Dorig > 0 ? Dorig : D) / 4);
}
void upsample_combine_downsample(const Ctx& ctx, const float* orig, const float* xd, const float* scale, float* y, const float* bias2,
                                 float* xd2, int B, int T, int Td, int D, int ds, int Dorig, int D2, int ds2) {
    K2_REQUIRE(ds2 >= 1 && ds2 <= 16 && D % 4 == 0 && D2 % 4 == 0, "upsample_combine_downsample: ds2=%d D=%d D2=%d unsupported", ds2, D, D2);
    const int Td2 = (T + ds2 - 1) / ds2;
    long long n = (long long)B * Td2 * (std::max(D, D2) / 4);
    LAUNCH(k_upsample_combine_downsample, dim3(nblocks(n, 256)), dim3(256), orig, xd, scale, y, bias2, xd2, B, T, Td, D / 4, ds,
           (Dorig > 0 ? Dorig : D) / 4, Td2, D2 / 4, ds2);
}
void convert_channels(const Ctx& ctx, const float* x, float* y, int M, int Din, int Dout) {
    long long n = (long long)M * (Dout / 4);
    LAUNCH(k_convert_channels, dim3(nblocks(n, 256)), dim3(256), x, y, (long long)M, Din / 4, Dout / 4);
}
void copy_cols(const Ctx& ctx, const float* x, int ldx, int xcol0, float* y, int ldy, int ycol0, int M, int n) {
    long long tot = (long long)M * (n / 4);
    LAUNCH(k_copy_cols, dim3(nblocks(tot, 256)), dim3(256), x, ldx, xcol0, y, ldy, ycol0, (long long)M, n / 4);
}

}  // namespace k2hip
