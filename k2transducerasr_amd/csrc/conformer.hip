// Helpers of the offline Conformer's RelPositionMultiheadAttention (icefall
// pruned_transducer_stateless2; reference side: Model_type "conformer" ->
// OfflineProjOfTransducer.EncoderProj, OfflineProjOfTransducer.cs:48-92).
//
// The two score products run on the MFMA GEMM as batched launches:
//   ac[b,h] = (q*s + u_h) . k^T        [T, T]
//   bd[b,h] = (q*s + v_h) . p_h^T      [T, 2T-1]
// and the rel_shift (icefall does it with as_strided) is folded into the softmax as a gather:
//   w[i, j] = softmax_j(ac[i, j] + bd[i, T-1-i+j])
// The softmaxed weights are then the K-contiguous A operand of the value product.
#include <algorithm>
#include <vector>

#include "kernels.h"

namespace k2hip {
namespace {

__global__ void k_conformer_qprep(const float* __restrict__ qkv, const float* __restrict__ bu, const float* __restrict__ bv,
                                  float* __restrict__ qu, float* __restrict__ qv, long long n4, int D4, float scaling, int ldq) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    long long r = i / D4;
    int c = (int)(i % D4) * 4;
    float4 q = *reinterpret_cast<const float4*>(qkv + r * ldq + c);
    float4 u = *reinterpret_cast<const float4*>(bu + c);
    float4 v = *reinterpret_cast<const float4*>(bv + c);
    q.x *= scaling; q.y *= scaling; q.z *= scaling; q.w *= scaling;
    reinterpret_cast<float4*>(qu)[i] = make_float4(q.x + u.x, q.y + u.y, q.z + u.z, q.w + u.w);
    reinterpret_cast<float4*>(qv)[i] = make_float4(q.x + v.x, q.y + v.y, q.z + v.z, q.w + v.w);
}

// one wave per (z, i) row; the row lives in registers (T <= 64 * SM_PER_LANE)
constexpr int SM_PER_LANE = 32;  // T <= 2048
__global__ __launch_bounds__(256) void k_conformer_softmax_shift(float* __restrict__ ac, const float* __restrict__ bd, long long rows,
                                                                 int T, int Tp, int NPp) {
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const int i = (int)(row % T);
    float* a = ac + row * Tp;
    const float* b = bd + row * NPp + (T - 1 - i);
    float v[SM_PER_LANE];
    float mx = -INFINITY;
#pragma unroll
    for (int u = 0; u < SM_PER_LANE; u++) {
        const int j = lane + 64 * u;
        v[u] = j < T ? a[j] + b[j] : -INFINITY;
        mx = fmaxf(mx, v[u]);
    }
    mx = wave_max_dpp(mx);
    float sum = 0.f;
#pragma unroll
    for (int u = 0; u < SM_PER_LANE; u++) {
        const int j = lane + 64 * u;
        v[u] = j < T ? __expf(v[u] - mx) : 0.f;
        sum += v[u];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    const float inv = 1.0f / sum;
#pragma unroll
    for (int u = 0; u < SM_PER_LANE; u++) {
        const int j = lane + 64 * u;
        if (j < Tp) a[j] = v[u] * inv;  // pad columns [T, Tp) become 0
    }
}

// any T: the same row operation with three sweeps over global memory instead of a register-resident row
__global__ __launch_bounds__(256) void k_conformer_softmax_shift_long(float* __restrict__ ac, const float* __restrict__ bd, long long rows,
                                                                      int T, int Tp, int NPp) {
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const int i = (int)(row % T);
    float* a = ac + row * Tp;
    const float* b = bd + row * NPp + (T - 1 - i);
    float mx = -INFINITY;
    for (int j = lane; j < T; j += 64) mx = fmaxf(mx, a[j] + b[j]);
    mx = wave_max_dpp(mx);
    float sum = 0.f;
    for (int j = lane; j < T; j += 64) {
        const float e = __expf(a[j] + b[j] - mx);
        a[j] = e;
        sum += e;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    const float inv = 1.0f / sum;
    for (int j = lane; j < Tp; j += 64) a[j] = j < T ? a[j] * inv : 0.f;
}

// streaming: one wave per (stream, head, query) row over KL = left + Tc keys
__global__ __launch_bounds__(256) void k_conformer_softmax_shift_stream(float* __restrict__ ac, const float* __restrict__ bd,
                                                                         const long long* __restrict__ plen, long long rows, int H, int Tc,
                                                                         int left, int KLp, int NPp) {
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const int i = (int)(row % Tc);
    const int b = (int)(row / ((long long)Tc * H));
    const int KL = left + Tc;
    const long long pl = plen[b];
    float* a = ac + row * KLp;
    const float* sh = bd + row * NPp + (Tc - 1 - i);
    float mx = -INFINITY;
    for (int j = lane; j < KL; j += 64) {
        float s = a[j] + sh[j];
        if (j < left && pl <= (long long)(left - 1 - j)) s = -INFINITY;
        a[j] = s;
        mx = fmaxf(mx, s);
    }
    mx = wave_max_dpp(mx);
    float sum = 0.f;
    for (int j = lane; j < KL; j += 64) {
        const float e = __expf(a[j] - mx);
        a[j] = e;
        sum += e;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    const float inv = 1.0f / sum;
    for (int j = lane; j < KLp; j += 64) a[j] = j < KL ? a[j] * inv : 0.f;
}

__global__ void k_slice_rows(const float* __restrict__ in, float* __restrict__ out, int Tin, int row0, int Tout, int D4, long long n4) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const int q = (int)(i % D4);
    const long long bt = i / D4;
    const int t = (int)(bt % Tout), b = (int)(bt / Tout);
    reinterpret_cast<float4*>(out)[i] = reinterpret_cast<const float4*>(in)[((long long)b * Tin + row0 + t) * D4 + q];
}

__global__ void k_dwconv_valid_dswish(const float* __restrict__ cat, const float* __restrict__ w, const float* __restrict__ bias,
                                      float* __restrict__ y, int Tc, int D4, int K, long long n4) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const int q = (int)(i % D4);
    const long long bt = i / D4;
    const int t = (int)(bt % Tc), b = (int)(bt / Tc);
    const int D = D4 * 4, Tin = K - 1 + Tc;
    float4 s = *reinterpret_cast<const float4*>(bias + 4 * q);
    for (int k = 0; k < K; k++) {
        const float4 xv = *reinterpret_cast<const float4*>(cat + ((long long)b * Tin + t + k) * D + 4 * q);
        const float4 wv = *reinterpret_cast<const float4*>(w + (long long)k * D + 4 * q);
        s.x += wv.x * xv.x; s.y += wv.y * xv.y; s.z += wv.z * xv.z; s.w += wv.w * xv.w;
    }
    auto ds = [](float v) { return v / (1.0f + __expf(1.0f - v)); };
    reinterpret_cast<float4*>(y)[i] = make_float4(ds(s.x), ds(s.y), ds(s.z), ds(s.w));
}


// ---------------------------------------------------------------------------------------
// RelPositionMultiheadAttention scores + softmax, fused (offline Conformer): for one (stream, head) and a strip of 16 query rows
//   s[i,j] = (q_i + u).k_j + (q_i + v).p[T-1-i+j]          w[i,:] = softmax_j s[i,:]
// without the two [T, T] / [T, 2T-1] score tensors of the GEMM form (145 + 290 MB per layer at T = 753, written and re-read).
// Four waves share the 16 x T strip in LDS (48 KB at T = 750, so THREE workgroups share a CU and one strip's softmax and write-out
// run under the others' MFMAs); a wave takes runs of 16-key tiles: the content term is 16x16x4 f32 MFMAs (K = dk); the positional
// term of a tile only involves the table rows n = base .. base + 30, two 16-wide table tiles G = (q+v) . p[base ..]^T whose elements
// land in the strip skewed -- G[r, col] belongs to key j0 + col - 15 + r -- icefall's rel-shift through LDS (each strip element
// receives exactly one of them, so the sum is exact and needs no staging tile: the strip is the only LDS).  Then the row softmax and
// one coalesced write.  NG = dk / 16.
// (Rounds 1 - 4 kept two earlier forms beside this one -- 32-row strips with 32x32x2 tiles, one workgroup per CU: 257 us per layer at
// T = 750; 16-row strips with a read-modify-write scatter: 199 us; this one 185 us -- DESIGN.md "Where the Conformer attention
// kernel's time goes".  Round 5 removed them: one path per operation in the library.)
// ---------------------------------------------------------------------------------------
constexpr int CR = 32;       // (bound of the in-LDS form: 32 x (Tp + 4) floats must fit, conformer_scores_softmax below)
typedef float cf32x4 __attribute__((ext_vector_type(4)));

// The scatter and the operand loads are off the critical path (round 4).  What the ISA of the read-modify-write form
// showed: every one of its 8 positional read-modify-writes per key tile sat in a branch of its own (ds_read, wait, add,
// ds_write: eight exposed LDS round trips per tile), and the loop began with vmcnt(0) on loads issued one scatter earlier.  Here
//   * a positional tile's element (rl, li) belongs to exactly ONE strip position, column 16 (m - 1) + 1 + li + rl of row rl (the "low
//     run" and the "high run" of the form above are the two halves of that one formula): the tile is WRITTEN skewed, four plain
//     ds_write per lane, masked by address (a dump slot per lane) -- no branch, no read;
//   * the content tile is then added on top: four reads in one batch, one wait, four writes (a wave's LDS operations complete in order,
//     and a wave only writes the columns of its own run);
//   * keys and table rows are requested TWO tiles ahead into the register set just consumed (the loop is unrolled by two so that the set
//     is a compile-time choice).
// Same MFMAs, same sums (a position receives one content value and one positional value: the order of that one addition does not
// matter), bit-identical weights.
template <int NG, int NW>
__global__ __launch_bounds__(64 * NW) void k_conformer_scores_softmax16s(const float* __restrict__ qu, const float* __restrict__ qv,
                                                                     const float* __restrict__ kmat, int ldk, const float* __restrict__ pp,
                                                                     float* __restrict__ aw, int B, int H, int T, int Tp, int D,
                                                                     int lds_stride, unsigned long long* __restrict__ stamps, int ldq,
                                                                     const float* __restrict__ bu, const float* __restrict__ bv, float scaling) {
    // bu != null: `qu` is the in_proj output's q block (rows ldq floats apart) and the two query operands are formed here, as
    // k_conformer_qprep forms them (q * scaling + pos_bias_u / pos_bias_v) -- that launch and its two [M, D] tensors are gone (round 5)
    constexpr int DK = 16 * NG;
    extern __shared__ __attribute__((aligned(16))) float csm[];
    float* S = csm;                                   // [16][lds_stride] | 64 dump slots
    // tuning (K2HIP_CONFORMER_STAMPS): per wave s_memtime at entry | queries loaded | tile loop done | barrier passed | end, + s_memrealtime
    unsigned long long* st = stamps ? stamps + ((size_t)((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * NW + (threadIdx.x >> 6)) * 8 : nullptr;
    if (st && (threadIdx.x & 63) == 0) { st[0] = __builtin_amdgcn_s_memtime(); st[6] = __builtin_amdgcn_s_memrealtime(); }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, kq = lane >> 4;
    const int i0 = blockIdx.x * 16, b = blockIdx.y, h = blockIdx.z, NP = 2 * T - 1;
    const long long rowbase = (long long)b * T;
    const int dump = 16 * lds_stride + lane;
    float4 fu[NG], fv[NG];
    {
        const int row = i0 + li;
#pragma unroll
        for (int g = 0; g < NG; g++) {
            fu[g] = fv[g] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < T) {
                if (bu) {
                    float4 q = *reinterpret_cast<const float4*>(qu + (rowbase + row) * ldq + h * DK + 16 * g + 4 * kq);
                    const float4 u = *reinterpret_cast<const float4*>(bu + h * DK + 16 * g + 4 * kq);
                    const float4 v = *reinterpret_cast<const float4*>(bv + h * DK + 16 * g + 4 * kq);
                    q.x *= scaling; q.y *= scaling; q.z *= scaling; q.w *= scaling;
                    fu[g] = make_float4(q.x + u.x, q.y + u.y, q.z + u.z, q.w + u.w);
                    fv[g] = make_float4(q.x + v.x, q.y + v.y, q.z + v.z, q.w + v.w);
                } else {
                    fu[g] = *reinterpret_cast<const float4*>(qu + (rowbase + row) * D + h * DK + 16 * g + 4 * kq);
                    fv[g] = *reinterpret_cast<const float4*>(qv + (rowbase + row) * D + h * DK + 16 * g + 4 * kq);
                }
            }
        }
    }
    const int njt = (T + 15) / 16;
    auto load_k = [&](int jt, float4* k_) {
        const int jr = min(jt * 16 + li, T - 1);
#pragma unroll
        for (int g = 0; g < NG; g++) k_[g] = *reinterpret_cast<const float4*>(kmat + (rowbase + jr) * ldk + h * DK + 16 * g + 4 * kq);
    };
    auto load_p = [&](int nrow, float4* p_) {   // table row nrow (out-of-table rows only meet masked entries)
        const int nr = min(max(nrow, 0), NP - 1);
#pragma unroll
        for (int g = 0; g < NG; g++) p_[g] = *reinterpret_cast<const float4*>(pp + (long long)nr * D + h * DK + 16 * g + 4 * kq);
    };
    auto mma = [](const float4& a, const float4& bq, cf32x4 acc) {
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, bq.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, bq.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, bq.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, bq.w, acc, 0, 0, 0);
        return acc;
    };
    const int tpw = (njt + NW - 1) / NW, jt_beg = wave * tpw, jt_end = min(njt, jt_beg + tpw);
    if (st && lane == 0) st[1] = __builtin_amdgcn_s_memtime();
    if (jt_beg < jt_end) {
        const int col_lo = 16 * jt_beg, col_hi = min(16 * jt_end, T);   // the strip columns this wave owns
        // positional tile m = jt_beg .. jt_end: table rows base0 + 16 (m - jt_beg) + li
        const int base0 = T - 1 - i0 - 15 + jt_beg * 16;
        auto put_pos = [&](const cf32x4& g, int m) {
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int rl = 4 * kq + e, col = 16 * (m - 1) + 1 + li + rl;
                S[(col >= col_lo && col < col_hi) ? rl * lds_stride + col : dump] = g[e];
            }
        };
        // (Tried on top of this form and measured slower, all of them fighting the compiler's wait counting rather than the hardware:
        // separate interior / boundary code paths with addresses as uniform base + fixed per-lane offset -- a third of the vector-ALU
        // instructions, but vmcnt(0) wherever the paths join: 233 us against 195; the same without branches: the register allocator
        // rotates the two operand sets through copies at the loop's latch and waits for the loads there, tile loop 67.8k cycles against
        // 58.6k; the loads as inline asm with hand-counted waits: the copies then read destination registers of loads in flight.)
        auto step = [&](int jt, float4* K_, float4* P_) {   // key tile jt (operands in K_) and positional tile jt + 1 (in P_)
            cf32x4 acc = {0.f, 0.f, 0.f, 0.f}, g = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < NG; q++) {
                acc = mma(fu[q], K_[q], acc);
                g = mma(fv[q], P_[q], g);
            }
            if (jt + 2 < jt_end) {   // (wave-uniform) two tiles ahead, into the registers just consumed
                load_k(jt + 2, K_);
                load_p(base0 + (jt + 3 - jt_beg) * 16 + li, P_);
            }
            put_pos(g, jt + 1);
            const int j = jt * 16 + li;
            int idx[4];
            float sv[4];
#pragma unroll
            for (int e = 0; e < 4; e++) {
                idx[e] = j < T ? (4 * kq + e) * lds_stride + j : dump;
                sv[e] = S[idx[e]];
            }
#pragma unroll
            for (int e = 0; e < 4; e++) S[idx[e]] = sv[e] + acc[e];
        };
        float4 k0[NG], k1[NG], p0[NG], p1[NG];
        load_p(base0 + li, p1);
        load_k(jt_beg, k0);
        load_p(base0 + 16 + li, p0);
        {
            cf32x4 g = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < NG; q++) g = mma(fv[q], p1[q], g);
            put_pos(g, jt_beg);
        }
        if (jt_beg + 1 < jt_end) {
            load_k(jt_beg + 1, k1);
            load_p(base0 + 32 + li, p1);
        }
        for (int jt = jt_beg; jt < jt_end; jt += 2) {
            step(jt, k0, p0);
            if (jt + 1 < jt_end) step(jt + 1, k1, p1);
        }
    }
    if (st && lane == 0) st[2] = __builtin_amdgcn_s_memtime();
    __syncthreads();
    if (st && lane == 0) st[3] = __builtin_amdgcn_s_memtime();
    // row softmax: wave w owns 16 / NW rows; a row's T scores are read ONCE into registers (T <= 64 * 20), four consecutive columns per
    // lane and instruction (ds_read_b128 / global_store_dwordx4: with three workgroups per CU the phase is bound by the instructions
    // the SIMD can issue, and this form has a third of them).  Rows are lds_stride = Tp + 4 floats, 16-byte aligned; Tp % 4 == 0.
    float* out = aw + (((long long)b * H + h) * T) * Tp;
    const int nu = (Tp + 255) >> 8;   // groups of 256 columns (<= 5)
    for (int rr = 0; rr < 16 / NW; rr++) {
        const int rl = wave * (16 / NW) + rr, i = i0 + rl;
        if (i >= T) break;
        const float* srow = S + rl * lds_stride;
        float4 v[5];
        float mx = -INFINITY;
#pragma unroll
        for (int u = 0; u < 5; u++) {
            if (u < nu) {
                const int jj = 4 * lane + 256 * u;
                const float4 x = *reinterpret_cast<const float4*>(srow + min(jj, lds_stride - 4));
                v[u].x = jj < T ? x.x : -INFINITY;
                v[u].y = jj + 1 < T ? x.y : -INFINITY;
                v[u].z = jj + 2 < T ? x.z : -INFINITY;
                v[u].w = jj + 3 < T ? x.w : -INFINITY;
                mx = fmaxf(fmaxf(mx, fmaxf(v[u].x, v[u].y)), fmaxf(v[u].z, v[u].w));
            }
        }
        mx = wave_max_dpp(mx);
        float sum = 0.f;
#pragma unroll
        for (int u = 0; u < 5; u++) {
            if (u < nu) {   // exp(-inf) = 0 for the masked columns
                v[u].x = __expf(v[u].x - mx);
                v[u].y = __expf(v[u].y - mx);
                v[u].z = __expf(v[u].z - mx);
                v[u].w = __expf(v[u].w - mx);
                sum += (v[u].x + v[u].y) + (v[u].z + v[u].w);
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
        const float inv = 1.0f / sum;
        float* orow = out + (long long)i * Tp;
#pragma unroll
        for (int u = 0; u < 5; u++) {
            const int jj = 4 * lane + 256 * u;
            if (u < nu && jj < Tp)   // pad columns [T, Tp) get 0
                *reinterpret_cast<float4*>(orow + jj) = make_float4(v[u].x * inv, v[u].y * inv, v[u].z * inv, v[u].w * inv);
        }
    }
    if (st && lane == 0) { st[4] = __builtin_amdgcn_s_memtime(); st[7] = __builtin_amdgcn_s_memrealtime(); }
}

}  // namespace

void conformer_softmax_shift_stream(const Ctx& ctx, float* ac, const float* bd, const long long* plen, int B, int H, int Tc, int left,
                                    int KLp, int NPp) {
    if (ctx.dry) return;
    const long long rows = (long long)B * H * Tc;
    hipLaunchKernelGGL(k_conformer_softmax_shift_stream, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, ctx.stream, ac, bd, plen, rows, H, Tc,
                       left, KLp, NPp);
    K2_HIP(hipGetLastError());
}
void slice_rows(const Ctx& ctx, const float* in, float* out, int B, int Tin, int row0, int Tout, int D) {
    if (ctx.dry) return;
    const long long n4 = (long long)B * Tout * (D / 4);
    hipLaunchKernelGGL(k_slice_rows, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, ctx.stream, in, out, Tin, row0, Tout, D / 4, n4);
    K2_HIP(hipGetLastError());
}
void dwconv_valid_dswish(const Ctx& ctx, const float* cat, const float* w_kd, const float* bias, float* y, int B, int Tc, int D, int K) {
    ctx.add_flops(0.0, 2.0 * B * Tc * (double)D * K, 0);
    if (ctx.dry) return;
    const long long n4 = (long long)B * Tc * (D / 4);
    hipLaunchKernelGGL(k_dwconv_valid_dswish, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, ctx.stream, cat, w_kd, bias, y, Tc, D / 4, K, n4);
    K2_HIP(hipGetLastError());
}

void conformer_qprep(const Ctx& ctx, const float* qkv, const float* bias_u, const float* bias_v, float* qu, float* qv, int M,
                     int D, float scaling, int ldq) {
    if (ctx.dry) return;
    long long n4 = (long long)M * D / 4;
    hipLaunchKernelGGL(k_conformer_qprep, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, ctx.stream, qkv, bias_u, bias_v, qu, qv,
                       n4, D / 4, scaling, ldq > 0 ? ldq : 3 * D);
    K2_HIP(hipGetLastError());
}

void conformer_softmax_shift(const Ctx& ctx, float* ac, const float* bd, int Z, int T, int Tp, int NPp) {
    if (ctx.dry) return;
    long long rows = (long long)Z * T;
    const bool force_long = tunables().attn_long != 0;
    if (T > 64 * SM_PER_LANE || force_long) {
        hipLaunchKernelGGL(k_conformer_softmax_shift_long, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, ctx.stream, ac, bd, rows, T, Tp, NPp);
        K2_HIP(hipGetLastError());
        return;
    }
    hipLaunchKernelGGL(k_conformer_softmax_shift, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, ctx.stream, ac, bd, rows, T, Tp, NPp);
    K2_HIP(hipGetLastError());
}

template <int NG>
static bool conformer_scores_launch16(const Ctx& ctx, const float* qu, const float* qv, const float* kmat, int ldk, const float* pp, float* aw,
                                      int B, int H, int T, int Tp, int D, int ldq, const float* bu, const float* bv, float scaling) {
    const int lds_stride = Tp + 4;
    const size_t lds = sizeof(float) * ((size_t)16 * lds_stride + 64);   // + a dump slot per lane (masked positional writes)
    static LdsAttrOnce lds_attr;
    // (eight waves per strip: 222 against 205 us per launch; four it is)
    {
        lds_attr.ensure((k_conformer_scores_softmax16s<NG, 4>), 96 * 1024);
        unsigned long long* d_st = nullptr;
        const size_t n_st = (size_t)cdiv(T, 16) * B * H * 4 * 8;
        if (tunables().conformer_stamps) {
            K2_HIP(hipMalloc(&d_st, n_st * 8));
            K2_HIP(hipMemsetAsync(d_st, 0, n_st * 8, ctx.stream));
        }
        hipLaunchKernelGGL((k_conformer_scores_softmax16s<NG, 4>), dim3(cdiv(T, 16), B, H), dim3(256), lds, ctx.stream, qu, qv, kmat, ldk, pp, aw, B, H,
                           T, Tp, D, lds_stride, d_st, ldq, bu, bv, scaling);
        if (d_st) {   // tuning only: synchronous report (cycles of the slowest wave of each workgroup, averaged; launch span in realtime ticks)
            std::vector<unsigned long long> h(n_st);
            K2_HIP(hipStreamSynchronize(ctx.stream));
            K2_HIP(copy_blocking(h.data(), d_st, n_st * 8, hipMemcpyDeviceToHost));
            (void)hipFree(d_st);
            double ph[4] = {0, 0, 0, 0};
            unsigned long long r0 = ~0ull, r1 = 0;
            const size_t nwg = n_st / 32;
            for (size_t w = 0; w < nwg; w++) {
                double worst[4] = {0, 0, 0, 0};
                for (int v = 0; v < 4; v++) {
                    const unsigned long long* q = &h[(w * 4 + v) * 8];
                    for (int k = 0; k < 4; k++) worst[k] = std::max(worst[k], (double)(q[k + 1] - q[k]));
                    r0 = std::min(r0, q[6]);
                    r1 = std::max(r1, q[7]);
                }
                for (int k = 0; k < 4; k++) ph[k] += worst[k];
            }
            fprintf(stderr, "conformer scores stamps: %zu workgroups; slowest wave per workgroup, mean cycles: queries %.0f | tile loop %.0f | barrier wait %.0f | "
                            "softmax + write %.0f; launch span %.1f us (100 MHz ticks)\n",
                    nwg, ph[0] / nwg, ph[1] / nwg, ph[2] / nwg, ph[3] / nwg, (double)(r1 - r0) / 100.0);
        }
    }
    K2_HIP(hipGetLastError());
    return true;
}

// fused scores + softmax of the offline Conformer attention; returns false (nothing launched) when the shape does not fit.
// bias_u != null: `qu` is the q block of the in_proj output (rows ldq floats apart, qv unused) and the kernel forms q * scaling + bias_u /
// bias_v itself (no conformer_qprep launch in front)
bool conformer_scores_softmax(const Ctx& ctx, const float* qu, const float* qv, const float* kmat, int ldk, const float* pp, float* aw, int B, int H,
                              int T, int Tp, int D, int ldq, const float* bias_u, const float* bias_v, float scaling) {
    const int dk = D / H;
    if (tunables().conformer_gemm_scores || Tp % 4 != 0 || D % 4 != 0 || ldk % 4 != 0 || (dk != 16 && dk != 32 && dk != 64)) return false;
    if (bias_u && (ldq % 4 != 0 || !bias_v)) return false;
    if (sizeof(float) * (size_t)CR * (Tp + 4) > 156 * 1024 || Tp > 64 * 20) return false;
    ctx.add_flops(0.0, 2.0 * dk * (double)T * (3.0 * T) * B * H, 0);
    if (ctx.dry) return true;
    if (dk == 64) return conformer_scores_launch16<4>(ctx, qu, qv, kmat, ldk, pp, aw, B, H, T, Tp, D, ldq, bias_u, bias_v, scaling);
    if (dk == 32) return conformer_scores_launch16<2>(ctx, qu, qv, kmat, ldk, pp, aw, B, H, T, Tp, D, ldq, bias_u, bias_v, scaling);
    return conformer_scores_launch16<1>(ctx, qu, qv, kmat, ldk, pp, aw, B, H, T, Tp, D, ldq, bias_u, bias_v, scaling);
}

}  // namespace k2hip
