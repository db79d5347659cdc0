// Kernels of the streaming Zipformer (v1) path: Model_type "zipformer" -> OnlineProjOfZipformer
// (OnlineRecognizer.cs:28-30).  As in online.hip every stream owns a slot of the device-resident state pool laid out like its
// per-stream state in GetEncoderInitStates (OnlineProjOfZipformer.cs:56-111):
//   per layer: cached_len (1) | cached_avg [D] | cached_key [left, att] | cached_val [left, att/2] |
//              cached_val2 [left, att/2] | cached_conv1 [D, K-1] | cached_conv2 [D, K-1]
// so the reference's stack_states / unstack_states (:133-428) become an index array.  The GEMMs, the cache concatenation
// (cat_shift) and the attention-weights x values products are shared with the Zipformer2 path; this file holds what v1 adds.
#include "kernels.h"

namespace k2hip {
namespace {

__device__ __forceinline__ float sigm(float s) { return 1.0f / (1.0f + __expf(-s)); }
__device__ __forceinline__ float dswish(float v) { return v * sigm(v - 1.0f); }

// PoolingModule.streaming_forward: out[b,t,:] = (cumsum_t(x[b]) + cached_avg * cached_len) / (t + 1 + cached_len);
// cached_avg <- out[b, Tc-1], cached_len += Tc.  One workgroup per stream (it owns the stream's cached_len).
__global__ __launch_bounds__(256) void k_z1_pool(const float* __restrict__ x, float* __restrict__ pool, long long slot_stride,
                                                 long long avg_off, long long len_off, const int* __restrict__ slots,
                                                 float* __restrict__ out, int Tc, int D) {
    const int b = blockIdx.x;
    float* st = pool + (long long)slots[b] * slot_stride;
    const float len = st[len_off];
    for (int d = threadIdx.x; d < D; d += blockDim.x) {
        const float base = st[avg_off + d] * len;
        float cum = 0.f, last = 0.f;
        for (int t = 0; t < Tc; t++) {
            cum += x[((long long)b * Tc + t) * D + d];
            last = (cum + base) * (1.0f / ((float)(t + 1) + len));
            out[((long long)b * Tc + t) * D + d] = last;
        }
        st[avg_off + d] = last;
    }
    __syncthreads();  // every lane has read `len`
    if (threadIdx.x == 0) st[len_off] = len + (float)Tc;
}

// RelPositionMultiheadAttention.streaming_multi_head_attention_forward, scores + softmax for one (stream, head):
//   s[i,j] = q_i.k_j + p_i.pos[Tc-1-i+j] over the KL = left + Tc keys (v1 has no mask over an unfilled left context)
//   qkvp rows: q [A] | k [A] | v [A/2] | p [H*4];  kcat [B, KL, A];  pp [2Tc-1+left, H*4];  aw [H][B][Tc][KLp]
__global__ __launch_bounds__(256) void k_z1_attn(const float* __restrict__ qkvp, int ld, const float* __restrict__ kcat,
                                                 const float* __restrict__ pp, float* __restrict__ aw, int B, int Tc, int L,
                                                 int KLp, int H, int A, int hd) {
    extern __shared__ float S[];  // [Tc][KL]
    const int b = blockIdx.x, h = blockIdx.y, KL = L + Tc, PH = 4;
    const int p_off = 2 * A + A / 2;
    for (int e = threadIdx.x; e < Tc * KL; e += blockDim.x) {
        int i = e / KL, j = e - i * KL;
        const float* q = qkvp + ((long long)b * Tc + i) * ld + h * hd;
        const float* p = qkvp + ((long long)b * Tc + i) * ld + p_off + h * PH;
        const float* k = kcat + ((long long)b * KL + j) * A + h * hd;
        float s = 0.f;
        for (int d = 0; d < hd; d += 4) {
            float4 a = *reinterpret_cast<const float4*>(q + d), c = *reinterpret_cast<const float4*>(k + d);
            s += a.x * c.x + a.y * c.y + a.z * c.z + a.w * c.w;
        }
        float4 pv = *reinterpret_cast<const float4*>(p);
        float4 ev = *reinterpret_cast<const float4*>(pp + (long long)(Tc - 1 - i + j) * (H * PH) + h * PH);
        s += pv.x * ev.x + pv.y * ev.y + pv.z * ev.z + pv.w * ev.w;
        S[e] = s;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* out = aw + (((long long)h * B + b) * Tc) * KLp;
    for (int i = wave; i < Tc; i += 4) {
        float* row = S + i * KL;
        float mx = -INFINITY;
        for (int j = lane; j < KL; j += 64) mx = fmaxf(mx, row[j]);
        mx = wave_max_dpp(mx);
        float sum = 0.f;
        for (int j = lane; j < KL; j += 64) {
            float e = __expf(row[j] - mx);
            row[j] = e;
            sum += e;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
        float inv = 1.0f / sum;
        for (int j = lane; j < KLp; j += 64) out[(long long)i * KLp + j] = j < KL ? row[j] * inv : 0.f;
    }
}

// ConvolutionModule.streaming_forward core: GLU -> [cache (K-1) ; chunk] -> depthwise K (valid) + bias -> DoubleSwish
//   x2 [B*Tc, 2D]; cache [D][K-1] per stream; w [D][K]; y [B*Tc, D].  KT = kernel size as a compile-time constant (0 = any): the
//   channel's taps then sit in registers for the whole chunk.
template <int KT>
__global__ __launch_bounds__(64) void k_z1_glu_conv(const float* __restrict__ x2, float* __restrict__ pool, long long slot_stride,
                                                    long long off, const int* __restrict__ slots, const float* __restrict__ w,
                                                    const float* __restrict__ bias, float* __restrict__ y, int Tc, int D, int Krt) {
    extern __shared__ float cat[];  // [(K-1 + Tc)][64]
    const int K = KT ? KT : Krt;
    const int c = blockIdx.x * 64 + threadIdx.x, b = blockIdx.y, lc = threadIdx.x, lo = K - 1;
    if (c >= D) return;  // whole trailing lanes only; each lane touches its own LDS column, no barrier needed
    float* cache = pool + (long long)slots[b] * slot_stride + off + (long long)c * lo;
    for (int r = 0; r < lo; r++) cat[r * 64 + lc] = cache[r];
    for (int t = 0; t < Tc; t++) {
        const float* row = x2 + ((long long)b * Tc + t) * 2 * D;
        cat[(lo + t) * 64 + lc] = row[c] * sigm(row[D + c]);
    }
    for (int r = 0; r < lo; r++) cache[r] = cat[(Tc + r) * 64 + lc];  // cache = cat[..., -(K-1):]
    const float bv = bias[c];
    constexpr int KR = KT ? KT : 1;
    float wr[KR];
    if (KT) {
#pragma unroll
        for (int k = 0; k < KR; k++) wr[k] = w[c * KR + k];
    }
    for (int t = 0; t < Tc; t++) {
        float s = bv;
        if (KT) {
#pragma unroll
            for (int k = 0; k < KR; k++) s += wr[k] * cat[(t + k) * 64 + lc];
        } else {
            for (int k = 0; k < K; k++) s += w[c * K + k] * cat[(t + k) * 64 + lc];
        }
        y[((long long)b * Tc + t) * D + c] = dswish(s);
    }
}

// layer tail: y = orig + (BasicNorm(x) - orig) * bypass_scale ; one wave per row
__global__ __launch_bounds__(256) void k_z1_norm_bypass(const float* __restrict__ x, const float* __restrict__ orig,
                                                        const float* __restrict__ log_eps, const float* __restrict__ bscale,
                                                        float* __restrict__ y, int M, int D) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= M) return;
    const float* xr = x + (long long)row * D;
    float ss = 0.f;
    for (int d = lane; d < D; d += 64) ss += xr[d] * xr[d];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
    const float sc = 1.0f / sqrtf(ss / (float)D + __expf(log_eps[0])), bs = bscale[0];
    for (int d = lane; d < D; d += 64) {
        float o = orig[(long long)row * D + d];
        y[(long long)row * D + d] = o + (xr[d] * sc - o) * bs;
    }
}

// AttentionDownsample.forward (first Din output channels): per (stream, output frame) softmax over the ds frames' scores
// <frame, query>, weighted sum.  Frames past the end repeat the last one.  x [B,T,Din] -> y [B,Td,ldy] (cols 0..Din)
__global__ __launch_bounds__(64) void k_z1_attn_downsample(const float* __restrict__ x, const float* __restrict__ query,
                                                           float* __restrict__ y, int T, int Td, int Din, int ldy, int ds) {
    const int td = blockIdx.x, b = blockIdx.y, lane = threadIdx.x;
    float sc[16];
    float mx = -INFINITY;
    for (int k = 0; k < ds; k++) {
        int t = min(td * ds + k, T - 1);
        const float* r = x + ((long long)b * T + t) * Din;
        float s = 0.f;
        for (int d = lane; d < Din; d += 64) s += r[d] * query[d];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        sc[k] = s;
        mx = fmaxf(mx, s);
    }
    float sum = 0.f;
    for (int k = 0; k < ds; k++) {
        sc[k] = __expf(sc[k] - mx);
        sum += sc[k];
    }
    const float inv = 1.0f / sum;
    for (int d = lane; d < Din; d += 64) {
        float a = 0.f;
        for (int k = 0; k < ds; k++) {
            int t = min(td * ds + k, T - 1);
            a += x[((long long)b * T + t) * Din + d] * (sc[k] * inv);
        }
        y[((long long)b * Td + td) * ldy + d] = a;
    }
}

// SimpleCombiner.forward(src1 [.,d1], src2 [.,d2]) = src1 * w (zero-padded / truncated to d2) + src2 * (1 - w);
// with ub != null, src2 is SimpleUpsample(xd)[: T]: xd[b, t / ds] + ub[t % ds]
__global__ void k_z1_combine(const float* __restrict__ s1, int d1, const float* __restrict__ s2, int d2, const float* __restrict__ w1,
                             const float* __restrict__ ub, int ds, int T, int Td, float* __restrict__ y, long long n) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int d = (int)(i % d2);
    const long long bt = i / d2;
    const float w = w1[0];
    float v2;
    if (ub) {
        const int t = (int)(bt % T);
        const long long b = bt / T;
        v2 = s2[(b * Td + t / ds) * d2 + d] + ub[(long long)(t % ds) * d2 + d];
    } else {
        v2 = s2[i];
    }
    const float a = d < d1 ? s1[bt * d1 + d] * w : 0.f;
    y[i] = a + v2 * (1.0f - w);
}

// ---- offline graph (OfflineRecognizer.cs:40: Model_type "zipformer" through OfflineProjOfTransducer) ----
// PoolingModule.forward with x_lens = T: mean over the utterance's frames, sum_t x[b,t,d] * (1/T) in frame order
__global__ void k_z1_mean(const float* __restrict__ x, float* __restrict__ mean, int B, int T, int D) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * D) return;
    const int b = i / D, d = i - b * D;
    const float w = 1.0f / (float)T;
    const float* p = x + (long long)b * T * D + d;
    float a = 0.f;
    for (int t = 0; t < T; t++) a += p[(long long)t * D] * w;
    mean[i] = a;
}
// x[b, t, :] += v[b, :]
__global__ void k_z1_add_bcast(float* __restrict__ x, const float* __restrict__ v, int T, int D4, long long n4) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const int q = (int)(i % D4);
    const long long b = i / ((long long)T * D4);
    float4 a = reinterpret_cast<float4*>(x)[i];
    const float4 c = reinterpret_cast<const float4*>(v)[b * D4 + q];
    reinterpret_cast<float4*>(x)[i] = make_float4(a.x + c.x, a.y + c.y, a.z + c.z, a.w + c.w);
}
// grp[b, td, k*Din + d] = x[b, min(td*ds + k, T-1), d]: the ds frames of a group side by side, the last frame repeated as padding
__global__ void k_z1_group_rows(const float* __restrict__ x, float* __restrict__ grp, int T, int Td, int Din, int ds, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int d = (int)(i % Din);
    long long r = i / Din;
    const int k = (int)(r % ds);
    r /= ds;
    const int td = (int)(r % Td);
    const long long b = r / Td;
    grp[i] = x[(b * T + min(td * ds + k, T - 1)) * Din + d];
}

inline int nb(long long n, int per) { return (int)((n + per - 1) / per); }

}  // namespace

void z1_mean(const Ctx& ctx, const float* x, float* mean, int B, int T, int D) {
    if (ctx.dry) return;
    hipLaunchKernelGGL(k_z1_mean, dim3(nb((long long)B * D, 64)), dim3(64), 0, ctx.stream, x, mean, B, T, D);
    K2_HIP(hipGetLastError());
}
void z1_add_bcast(const Ctx& ctx, float* x, const float* v, int B, int T, int D) {
    K2_REQUIRE(D % 4 == 0, "z1_add_bcast: D=%d", D);
    if (ctx.dry) return;
    const long long n4 = (long long)B * T * (D / 4);
    hipLaunchKernelGGL(k_z1_add_bcast, dim3(nb(n4, 256)), dim3(256), 0, ctx.stream, x, v, T, D / 4, n4);
    K2_HIP(hipGetLastError());
}
void z1_group_rows(const Ctx& ctx, const float* x, float* grp, int B, int T, int Din, int ds) {
    if (ctx.dry) return;
    const int Td = (T + ds - 1) / ds;
    const long long n = (long long)B * Td * ds * Din;
    hipLaunchKernelGGL(k_z1_group_rows, dim3(nb(n, 256)), dim3(256), 0, ctx.stream, x, grp, T, Td, Din, ds, n);
    K2_HIP(hipGetLastError());
}

void z1_pool(const Ctx& ctx, const float* x, float* pool, long long slot_stride, long long avg_off, long long len_off, const int* slots,
             float* out, int B, int Tc, int D) {
    if (ctx.dry) return;
    hipLaunchKernelGGL(k_z1_pool, dim3(B), dim3(256), 0, ctx.stream, x, pool, slot_stride, avg_off, len_off, slots, out, Tc, D);
    K2_HIP(hipGetLastError());
}
void z1_attn(const Ctx& ctx, const float* qkvp, int ld, const float* kcat, const float* pp, float* aw, int B, int Tc, int L, int KLp,
             int H, int A) {
    const int hd = A / H;
    K2_REQUIRE(A % H == 0 && hd % 4 == 0 && A % 8 == 0 && ld % 4 == 0, "zipformer attention: attention_dim %d / heads %d unsupported", A, H);
    ctx.add_flops(0.0, 2.0 * (hd + 4) * (double)Tc * (L + Tc) * B * H, 0);
    if (ctx.dry) return;
    size_t lds = sizeof(float) * Tc * (L + Tc);
    K2_REQUIRE(lds <= 64 * 1024, "zipformer attention: %d x %d scores do not fit the workgroup's LDS", Tc, L + Tc);
    hipLaunchKernelGGL(k_z1_attn, dim3(B, H), dim3(256), lds, ctx.stream, qkvp, ld, kcat, pp, aw, B, Tc, L, KLp, H, A, hd);
    K2_HIP(hipGetLastError());
}
void z1_glu_conv(const Ctx& ctx, const float* x2, float* pool, long long slot_stride, long long off, const int* slots, const float* w,
                 const float* bias, float* y, int B, int Tc, int D, int K) {
    ctx.add_flops(0.0, 2.0 * B * Tc * (double)D * K, 0);
    if (ctx.dry) return;
    size_t lds = sizeof(float) * (K - 1 + Tc) * 64;
    const dim3 grid(cdiv(D, 64), B);
#define K2_ZGC(KT) hipLaunchKernelGGL(k_z1_glu_conv<KT>, grid, dim3(64), lds, ctx.stream, x2, pool, slot_stride, off, slots, w, bias, y, Tc, D, K)
    switch (K) {
        case 31: K2_ZGC(31); break;
        case 15: K2_ZGC(15); break;
        case 7: K2_ZGC(7); break;
        case 5: K2_ZGC(5); break;
        default: K2_ZGC(0); break;
    }
#undef K2_ZGC
    K2_HIP(hipGetLastError());
}
void z1_norm_bypass(const Ctx& ctx, const float* x, const float* orig, const float* log_eps, const float* bscale, float* y, int M, int D) {
    if (ctx.dry) return;
    hipLaunchKernelGGL(k_z1_norm_bypass, dim3(cdiv(M, 4)), dim3(256), 0, ctx.stream, x, orig, log_eps, bscale, y, M, D);
    K2_HIP(hipGetLastError());
}
void z1_attn_downsample(const Ctx& ctx, const float* x, const float* query, float* y, int B, int T, int Din, int ldy, int ds) {
    K2_REQUIRE(ds >= 1 && ds <= 16, "AttentionDownsample: factor %d unsupported", ds);
    if (ctx.dry) return;
    const int Td = (T + ds - 1) / ds;
    hipLaunchKernelGGL(k_z1_attn_downsample, dim3(Td, B), dim3(64), 0, ctx.stream, x, query, y, T, Td, Din, ldy, ds);
    K2_HIP(hipGetLastError());
}
void z1_combine(const Ctx& ctx, const float* s1, int d1, const float* s2, int d2, const float* w1, const float* ub, int ds, int B, int T,
                int Td, float* y) {
    if (ctx.dry) return;
    long long n = (long long)B * T * d2;
    hipLaunchKernelGGL(k_z1_combine, dim3(nb(n, 256)), dim3(256), 0, ctx.stream, s1, d1, s2, d2, w1, ub, ds, T, Td, y, n);
    K2_HIP(hipGetLastError());
}

}  // namespace k2hip
