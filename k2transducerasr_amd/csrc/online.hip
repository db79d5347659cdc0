// Kernels of the streaming (OnlineRecognizer / OnlineProjOfZipformer2) path.
//
// The reference keeps every stream's caches in managed arrays and, per chunk, interleaves them
// into batch-major ONNX inputs and back (stack_states / unstack_states,
// OnlineProjOfZipformer2.cs:144-489: ~1.85 MB per stream per direction per chunk on one host
// thread).  Here every stream owns a slot of one device-resident pool laid out exactly like the
// per-stream state of GetEncoderInitStates (:63-111); kernels index the pool by slot, so
// "stacking" is an index array and nothing is copied to or from the host.
//   per layer: cached_key [L, 32H] | cached_nonlin_attn [L, 3D/4] | cached_val1 [L, 12H] |
//              cached_val2 [L, 12H] | cached_conv1 [D, K/2] | cached_conv2 [D, K/2]
//   then embed_states [128, 3, 19].  processed_lens lives on the host (it is 16 x chunks).
#include "kernels.h"

namespace k2hip {
namespace {

__device__ __forceinline__ float fast_softplus(float z) { return z > 15.f ? z : __logf(1.0f + __expf(z)); }
__device__ __forceinline__ float swoosh_r(float v) { return fast_softplus(v - 1.0f) - 0.08f * v - 0.313261687f; }
__device__ __forceinline__ float sigm(float s) { return 1.0f / (1.0f + __expf(-s)); }

// ConvNeXt.streaming_forward in one pass over x (round 5; three launches before: this one, the cache update, a 2-D copy):
//   cat[b] = [cached_left_pad (3 frames) ; x (T3 frames)], NHWC
//   cached_left_pad <- cat[:, :, Tc:Tc+3] = x frames Tc-3 .. Tc-1   (a cache element is read -- into cat -- and rewritten by the SAME
//                                                                    thread, in that order: nobody else touches it)
//   byp[b] = x[b, :Tc]                                               (the module's bypass operand)
__global__ void k_convnext_cat(const float* __restrict__ a3, float* __restrict__ pool, long long slot_stride,
                               long long embed_off, const int* __restrict__ slots, float* __restrict__ cat, float* __restrict__ byp, int B,
                               int T3, int Tc, int F, int C) {
    long long n = (long long)B * (T3 + 3) * F * C;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int c = (int)(i % C);
    long long p = i / C;
    int f = (int)(p % F);
    long long bt = p / F;
    int t = (int)(bt % (T3 + 3)), b = (int)(bt / (T3 + 3));
    float v;
    if (t < 3) {
        float* cp = pool + (long long)slots[b] * slot_stride + embed_off + ((long long)c * 3 + t) * F + f;
        v = *cp;
        *cp = a3[(((long long)b * T3 + (Tc - 3 + t)) * F + f) * C + c];
    } else {
        v = a3[(((long long)b * T3 + (t - 3)) * F + f) * C + c];
        if (t - 3 < Tc) byp[(((long long)b * Tc + (t - 3)) * F + f) * C + c] = v;
    }
    cat[i] = v;
}

// cat[b] = [cache[slot_b] (L rows) ; new[b] (Tc rows)] and cache[slot_b] <- cat[b][Tc:], in ONE pass: a thread owns the rows
// r0, r0+Tc, r0+2Tc, ... of one 16-byte column of one stream.  Row r of the new cache is row r+Tc of the old one (or a new row), so
// walking the chain upwards every old value is read exactly once -- it goes to cat[r] and, one step later, into cache[r-Tc] -- and
// no other thread touches these addresses.  (Two launches before: build cat, then copy the cache back out of it.)
// GATE: the new rows are not read as they are but formed on the fly as x[width + c] * tanh(x[c]) from rows of >= 2*width floats
// (NonlinAttention's gated input, which is what its cache holds) -- saves the separate gating kernel and its round trip.
template <bool GATE>
__global__ void k_cat_shift(float* __restrict__ pool, long long slot_stride, long long off, const int* __restrict__ slots,
                            const float* __restrict__ newrows, int ldn, float* __restrict__ cat, int B, int L, int Tc, int width) {
    auto newrow = [&](const float* p) {
        float4 v = *reinterpret_cast<const float4*>(GATE ? p + width : p);
        if (GATE) {
            const float4 sg = *reinterpret_cast<const float4*>(p);
            v = make_float4(v.x * tanhf(sg.x), v.y * tanhf(sg.y), v.z * tanhf(sg.z), v.w * tanhf(sg.w));
        }
        return v;
    };
    const int w4 = width >> 2;
    const long long n = (long long)B * Tc * w4;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int c = (int)(i % w4) * 4;
    const long long br = i / w4;
    const int r0 = (int)(br % Tc), b = (int)(br / Tc);
    float* cache = pool + (long long)slots[b] * slot_stride + off + c;
    const float* nw = newrows + (long long)b * Tc * ldn + c;
    float* ct = cat + (long long)b * (L + Tc) * width + c;
    const float4 mine = newrow(nw + (long long)r0 * ldn);
    *reinterpret_cast<float4*>(ct + (long long)(L + r0) * width) = mine;
    if (r0 >= L) return;
    // the chain's values are fetched in groups of 8 independent loads before any store (the stores go through the same pointer, so
    // load-after-store would otherwise serialise the walk into one memory latency per row)
    float4 cur = *reinterpret_cast<const float4*>(cache + (long long)r0 * width);  // old cache[r]
    for (int rb = r0; rb < L; rb += 8 * Tc) {
        float4 nx[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int q = rb + (u + 1) * Tc;  // row of [old cache ; new] that becomes cache[rb + u*Tc]
            if (q - Tc < L)
                nx[u] = q < L ? *reinterpret_cast<const float4*>(cache + (long long)q * width)
                              : newrow(nw + (long long)(q - L) * ldn);
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int r = rb + u * Tc;
            if (r < L) {
                *reinterpret_cast<float4*>(ct + (long long)r * width) = cur;
                *reinterpret_cast<float4*>(cache + (long long)r * width) = nx[u];
                cur = nx[u];
            }
        }
    }
}

// RelPositionMultiheadAttentionWeights.streaming_forward for one (stream, head), keys in the stream's ring (RingRef): the
// chunk's own key rows go into the ring first (this workgroup's 32 columns of them; nobody else reads or writes those), then every
// key is read from ring row p.  Column p of aw = ring row p; the reference's key index of that row (for the positional term and the
// left-context mask) is j = (p - head - Tc) mod KL.
// a lane's value moved within its row of 16 lanes (DPP: quad_perm [1,0,3,2] 0xB1, [2,3,0,1] 0x4E, row_half_mirror 0x141, row_mirror 0x140)
template <int CTRL>
__device__ __forceinline__ float row16_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
__global__ __launch_bounds__(256) void k_attn_stream_ring(const float* __restrict__ qkp, int ld, RingRef keys, const float* __restrict__ pp,
                                                          const long long* __restrict__ plen, float* __restrict__ aw, int B, int Tc,
                                                          int L, int KLp, int H, int ds, int left50) {
    extern __shared__ float S[];  // [Tc][KL]
    const int b = blockIdx.x, h = blockIdx.y, KL = L + Tc;
    const int QH = 32, PH = 4;
    const long long pl = plen[b];
    const int head = (int)(((long long)keys.chunks[b] * Tc) % KL);
    float* ring = keys.pool + (long long)keys.slots[b] * keys.slot_stride + keys.off;
    for (int e = threadIdx.x; e < Tc * (QH / 4); e += blockDim.x) {
        const int r = e / (QH / 4), c = (e % (QH / 4)) * 4;
        *reinterpret_cast<float4*>(ring + (long long)((head + r) % KL) * (H * QH) + h * QH + c) =
            *reinterpret_cast<const float4*>(qkp + ((long long)b * Tc + r) * ld + H * QH + h * QH + c);
    }
    __syncthreads();  // this workgroup's stores are visible to its own loads below (no line of the ring was cached before them)
    // thread = one query row i for a column of keys p = p0, p0 + blockDim / Tc, ...: the row's q (32 floats) and positional query (4)
    // are read once into registers (a flat (i, p) index re-read both for every score: half of the kernel's loads)
    if (Tc <= (int)blockDim.x && blockDim.x % Tc == 0) {
        const int i = threadIdx.x % Tc, pstep = blockDim.x / Tc;
        const float* q = qkp + ((long long)b * Tc + i) * ld + h * QH;
        float4 qv[QH / 4];
#pragma unroll
        for (int d = 0; d < QH / 4; d++) qv[d] = *reinterpret_cast<const float4*>(q + 4 * d);
        const float4 pv = *reinterpret_cast<const float4*>(qkp + ((long long)b * Tc + i) * ld + 2 * H * QH + h * PH);
        for (int p = threadIdx.x / Tc; p < KL; p += pstep) {
            int j = (p - head - Tc) % KL;
            if (j < 0) j += KL;
            const float* k = ring + (long long)p * (H * QH) + h * QH;
            float s = 0.f;
#pragma unroll
            for (int d = 0; d < QH / 4; d++) {
                const float4 c = *reinterpret_cast<const float4*>(k + 4 * d);
                s += qv[d].x * c.x + qv[d].y * c.y + qv[d].z * c.z + qv[d].w * c.w;
            }
            const float4 ev = *reinterpret_cast<const float4*>(pp + (long long)(Tc - 1 - i + j) * (H * PH) + h * PH);
            s += pv.x * ev.x + pv.y * ev.y + pv.z * ev.z + pv.w * ev.w;
            // src_key_padding_mask[..., ::ds]: left-context slot j (50 Hz slot j*ds) is valid only once processed
            if (j < L && pl <= (long long)(left50 - 1 - j * ds)) s = -1000.0f;
            S[i * KL + p] = s;
        }
    } else
    for (int e = threadIdx.x; e < Tc * KL; e += blockDim.x) {
        const int i = e / KL, p = e - i * KL;
        int j = (p - head - Tc) % KL;
        if (j < 0) j += KL;
        const float* q = qkp + ((long long)b * Tc + i) * ld + h * QH;
        const float* pq = qkp + ((long long)b * Tc + i) * ld + 2 * H * QH + h * PH;
        const float* k = ring + (long long)p * (H * QH) + h * QH;
        float s = 0.f;
#pragma unroll
        for (int d = 0; d < QH; d += 4) {
            float4 a = *reinterpret_cast<const float4*>(q + d), c = *reinterpret_cast<const float4*>(k + d);
            s += a.x * c.x + a.y * c.y + a.z * c.z + a.w * c.w;
        }
        float4 pv = *reinterpret_cast<const float4*>(pq);
        float4 ev = *reinterpret_cast<const float4*>(pp + (long long)(Tc - 1 - i + j) * (H * PH) + h * PH);
        s += pv.x * ev.x + pv.y * ev.y + pv.z * ev.z + pv.w * ev.w;
        // src_key_padding_mask[..., ::ds]: left-context slot j (50 Hz slot j*ds) is valid only once processed
        if (j < L && pl <= (long long)(left50 - 1 - j * ds)) s = -1000.0f;
        S[e] = s;
    }
    __syncthreads();
    // softmax: a row per group of 16 lanes (4 rows per wave at once; the reductions stay inside a DPP row of 16: no ds_bpermute)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, sl = lane & 15;
    float* out = aw + (((long long)h * B + b) * Tc) * KLp;
    for (int i0 = 0; i0 < Tc; i0 += 16) {
        const int i = i0 + wave * 4 + (lane >> 4);
        const bool live = i < Tc;
        float* row = S + (live ? i : 0) * KL;
        float mx = -INFINITY;
        for (int j = sl; j < KL; j += 16) mx = fmaxf(mx, row[j]);
        mx = fmaxf(mx, row16_f<0xB1>(mx));
        mx = fmaxf(mx, row16_f<0x4E>(mx));
        mx = fmaxf(mx, row16_f<0x141>(mx));
        mx = fmaxf(mx, row16_f<0x140>(mx));
        float sum = 0.f;
        for (int j = sl; j < KL; j += 16) {
            const float e = __expf(row[j] - mx);
            if (live) row[j] = e;
            sum += e;
        }
        sum += row16_f<0xB1>(sum);
        sum += row16_f<0x4E>(sum);
        sum += row16_f<0x141>(sum);
        sum += row16_f<0x140>(sum);
        const float inv = 1.0f / sum;
        if (live)
            for (int j = sl; j < KLp; j += 16) out[(long long)i * KLp + j] = j < KL ? row[j] * inv : 0.f;
    }
}

// RelPositionMultiheadAttentionWeights.streaming_forward for one (stream, head):
//   scores[i,j] = q_i.k_j + p_i.pos[Tc-1-i+j]; key j < L masked (-1000) while the left context is not
//   yet filled (processed_lens); softmax over j.  aw: [H][B][Tc][KLp]
__global__ __launch_bounds__(256) void k_attn_stream(const float* __restrict__ qkp, int ld, const float* __restrict__ kcat,
                                                     const float* __restrict__ pp, const long long* __restrict__ plen,
                                                     float* __restrict__ aw, int B, int Tc, int L, int KLp, int H, int ds,
                                                     int left50) {
    extern __shared__ float S[];  // [Tc][KL]
    const int b = blockIdx.x, h = blockIdx.y, KL = L + Tc;
    const int QH = 32, PH = 4;
    const long long pl = plen[b];
    for (int e = threadIdx.x; e < Tc * KL; e += blockDim.x) {
        int i = e / KL, j = e - i * KL;
        const float* q = qkp + ((long long)b * Tc + i) * ld + h * QH;
        const float* p = qkp + ((long long)b * Tc + i) * ld + 2 * H * QH + h * PH;
        const float* k = kcat + ((long long)b * KL + j) * (H * QH) + h * QH;
        float s = 0.f;
#pragma unroll
        for (int d = 0; d < QH; d += 4) {
            float4 a = *reinterpret_cast<const float4*>(q + d), c = *reinterpret_cast<const float4*>(k + d);
            s += a.x * c.x + a.y * c.y + a.z * c.z + a.w * c.w;
        }
        float4 pv = *reinterpret_cast<const float4*>(p);
        float4 ev = *reinterpret_cast<const float4*>(pp + (long long)(Tc - 1 - i + j) * (H * PH) + h * PH);
        s += pv.x * ev.x + pv.y * ev.y + pv.z * ev.z + pv.w * ev.w;
        // src_key_padding_mask[..., ::ds]: left-context slot j (50 Hz slot j*ds) is valid only once processed
        if (j < L && pl <= (long long)(left50 - 1 - j * ds)) s = -1000.0f;
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

// ConvolutionModule.streaming_forward core: GLU -> ChunkCausalDepthwiseConv1d.streaming_forward -> SwooshR
//   x2: [B*Tc, 2D]; cache [D][pad] per stream; wc [D][Kc], ww [D][K], sc [2][D][K]; y: [B*Tc, D]
// KT = the kernel size as a compile-time constant (31 / 15 / 7; 0 = any): the channel's causal and chunk-wise taps are then held in
// registers for the whole chunk (read per frame and tap from memory they were a chain of ~(K + Kc) Tc dependent loads per lane).
template <int KT>
__global__ __launch_bounds__(64) void k_glu_causal_conv(const float* __restrict__ x2, float* __restrict__ pool,
                                                        long long slot_stride, long long off, const int* __restrict__ slots,
                                                        const float* __restrict__ wc, const float* __restrict__ bc,
                                                        const float* __restrict__ ww, const float* __restrict__ bw,
                                                        const float* __restrict__ sc, float* __restrict__ y, int B, int Tc,
                                                        int D, int Krt) {
    extern __shared__ float cat[];  // [(pad + Tc)][64]
    const int K = KT ? KT : Krt;
    const int c = blockIdx.x * 64 + threadIdx.x, b = blockIdx.y, lc = threadIdx.x;
    const int pad = K >> 1, Kc = (K + 1) >> 1;
    if (c >= D) return;  // whole trailing lanes only; no barrier below depends on them (each lane uses its own column)
    float* cache = pool + (long long)slots[b] * slot_stride + off + (long long)c * pad;
    for (int r = 0; r < pad; r++) cat[r * 64 + lc] = cache[r];
    for (int t = 0; t < Tc; t++) {
        const float* row = x2 + ((long long)b * Tc + t) * 2 * D;
        cat[(pad + t) * 64 + lc] = row[c] * sigm(row[D + c]);
    }
    for (int r = 0; r < pad; r++) cache[r] = cat[(Tc + r) * 64 + lc];  // cache = cat[..., -pad:]
    const float bcv = bc[c], bwv = bw[c];
    constexpr int KR = KT ? KT : 1, KCR = KT ? (KT + 1) / 2 : 1;
    float wcr[KCR], wwr[KR];
    if (KT) {
#pragma unroll
        for (int k = 0; k < KCR; k++) wcr[k] = wc[c * KCR + k];
#pragma unroll
        for (int k = 0; k < KR; k++) wwr[k] = ww[c * KR + k];
    }
    for (int t = 0; t < Tc; t++) {
        float xc = bcv, xw = bwv;
        if (KT) {
#pragma unroll
            for (int k = 0; k < KCR; k++) xc += wcr[k] * cat[(t + k) * 64 + lc];
#pragma unroll
            for (int k = 0; k < KR; k++) {
                const int tt = t + k - (KR >> 1);
                if (tt >= 0 && tt < Tc) xw += wwr[k] * cat[((KR >> 1) + tt) * 64 + lc];
            }
        } else {
            for (int k = 0; k < Kc; k++) xc += wc[c * Kc + k] * cat[(t + k) * 64 + lc];
            for (int k = 0; k < K; k++) {
                int tt = t + k - pad;
                if (tt >= 0 && tt < Tc) xw += ww[c * K + k] * cat[(pad + tt) * 64 + lc];
            }
        }
        float le, re;
        if (Tc < K) {
            le = sc[(long long)c * K + t];
            re = sc[(long long)D * K + (long long)c * K + (K - Tc) + t];
        } else {
            le = t < K ? sc[(long long)c * K + t] : 0.f;
            re = t >= Tc - K ? sc[(long long)D * K + (long long)c * K + (t - (Tc - K))] : 0.f;
        }
        y[((long long)b * Tc + t) * D + c] = swoosh_r(xw * (1.0f + (le + re)) + xc);
    }
}

// The same with the chunk length a compile-time constant too: the lane's whole column (cache + chunk, pad + TC values) stays in
// registers, the frame loop is unrolled and no value goes through LDS (the LDS form reads one word per FMA: 47 per frame and lane
// at K = 31).  Same sums in the same order: bit-identical outputs and caches.
template <int KT, int TC>
__global__ __launch_bounds__(64) void k_glu_causal_conv_reg(const float* __restrict__ x2, float* __restrict__ pool,
                                                            long long slot_stride, long long off, const int* __restrict__ slots,
                                                            const float* __restrict__ wc, const float* __restrict__ bc,
                                                            const float* __restrict__ ww, const float* __restrict__ bw,
                                                            const float* __restrict__ sc, float* __restrict__ y, int B, int D) {
    constexpr int K = KT, pad = K >> 1, Kc = (K + 1) >> 1, Tc = TC;
    const int c = blockIdx.x * 64 + threadIdx.x, b = blockIdx.y;
    if (c >= D) return;
    float* cache = pool + (long long)slots[b] * slot_stride + off + (long long)c * pad;
    float col[pad + Tc];
#pragma unroll
    for (int r = 0; r < pad; r++) col[r] = cache[r];
    float xv[Tc], gv[Tc];
#pragma unroll
    for (int t = 0; t < Tc; t++) {
        const float* row = x2 + ((long long)b * Tc + t) * 2 * D;
        xv[t] = row[c];
        gv[t] = row[D + c];
    }
    float wcr[Kc], wwr[K];
#pragma unroll
    for (int k = 0; k < Kc; k++) wcr[k] = wc[c * Kc + k];
#pragma unroll
    for (int k = 0; k < K; k++) wwr[k] = ww[c * K + k];
    const float bcv = bc[c], bwv = bw[c];
    float le[Tc], re[Tc];
#pragma unroll
    for (int t = 0; t < Tc; t++) {
        if (Tc < K) {
            le[t] = sc[(long long)c * K + t];
            re[t] = sc[(long long)D * K + (long long)c * K + (K - Tc) + t];
        } else {
            le[t] = t < K ? sc[(long long)c * K + t] : 0.f;
            re[t] = t >= Tc - K ? sc[(long long)D * K + (long long)c * K + (t - (Tc - K))] : 0.f;
        }
    }
#pragma unroll
    for (int t = 0; t < Tc; t++) col[pad + t] = xv[t] * sigm(gv[t]);
#pragma unroll
    for (int r = 0; r < pad; r++) cache[r] = col[Tc + r];  // cache = cat[..., -pad:]
#pragma unroll
    for (int t = 0; t < Tc; t++) {
        float xc = bcv, xw = bwv;
#pragma unroll
        for (int k = 0; k < Kc; k++) xc += wcr[k] * col[t + k];
#pragma unroll
        for (int k = 0; k < K; k++) {
            const int tt = t + k - pad;
            if (tt >= 0 && tt < Tc) xw += wwr[k] * col[pad + tt];
        }
        y[((long long)b * Tc + t) * D + c] = swoosh_r(xw * (1.0f + (le[t] + re[t])) + xc);
    }
}

// ---- device mirror of the feature FIFOs: float4 per lane along the feature dimension (feat % 4 == 0)
__global__ void k_fifo_append(float* __restrict__ fifo, int cap, int f4, const float* __restrict__ src, const int* __restrict__ slots,
                              const int* __restrict__ pos, int nf) {
    const int g = blockIdx.y, p0 = pos[g];
    if (p0 < 0) return;
    const long long n = (long long)nf * f4;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int fr = (int)(i / f4), q = (int)(i % f4);
        reinterpret_cast<float4*>(fifo)[((long long)slots[g] * cap + (p0 + fr) % cap) * f4 + q] =
            reinterpret_cast<const float4*>(src)[((long long)g * nf + fr) * f4 + q];
    }
}
__global__ void k_fifo_gather(const float* __restrict__ fifo, int cap, int f4, const int* __restrict__ slots, const int* __restrict__ head,
                              float* __restrict__ x, int T) {
    const int b = blockIdx.y, h = head[b];
    const long long n = (long long)T * f4;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int t = (int)(i / f4), q = (int)(i % f4);
        float4 v = reinterpret_cast<const float4*>(fifo)[((long long)slots[b] * cap + (h + t) % cap) * f4 + q];
        // the online PadSequence's floor of genuine zeros (PadHelper.cs:9-13,58; k_logfloor) on the way through: one launch fewer per tick
        v.x = v.x == 0.0f ? -23.025850929940457F : v.x;
        v.y = v.y == 0.0f ? -23.025850929940457F : v.y;
        v.z = v.z == 0.0f ? -23.025850929940457F : v.z;
        v.w = v.w == 0.0f ? -23.025850929940457F : v.w;
        reinterpret_cast<float4*>(x)[((long long)b * T + t) * f4 + q] = v;
    }
}

// zero one stream's slot (GetEncoderInitStates: all caches start at 0)
__global__ void k_zero(float* __restrict__ p, long long n) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0.f;
}

// x == 0 -> log floor, in place (online PadSequence, PadHelper.cs:9-13,58)
__global__ void k_logfloor(float* __restrict__ x, long long n) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && x[i] == 0.0f) x[i] = -23.025850929940457F;
}

inline int nb(long long n, int per) { return (int)((n + per - 1) / per); }

}  // namespace

void convnext_cat(const Ctx& ctx, const float* a3, float* pool, long long slot_stride, long long embed_off, const int* slots, float* cat,
                  float* byp, int B, int T3, int Tc, int F, int C) {
    K2_REQUIRE(Tc >= 3 && Tc <= T3, "convnext_cat: chunk of %d frames out of %d", Tc, T3);
    if (ctx.dry) return;
    long long n = (long long)B * (T3 + 3) * F * C;
    hipLaunchKernelGGL(k_convnext_cat, dim3(nb(n, 256)), dim3(256), 0, ctx.stream, a3, pool, slot_stride, embed_off, slots, cat, byp, B, T3, Tc,
                       F, C);
    K2_HIP(hipGetLastError());
}
void cat_shift(const Ctx& ctx, float* pool, long long slot_stride, long long off, const int* slots, const float* newrows,
               int ldn, float* cat, int B, int L, int Tc, int width, bool tanh_gated) {
    K2_REQUIRE(width % 4 == 0 && ldn % 4 == 0, "cat_shift: width %d / ld %d must be multiples of 4", width, ldn);
    if (ctx.dry) return;
    const dim3 grid(nb((long long)B * Tc * (width / 4), 256));
    if (tanh_gated)
        hipLaunchKernelGGL(k_cat_shift<true>, grid, dim3(256), 0, ctx.stream, pool, slot_stride, off, slots, newrows, ldn, cat, B, L, Tc, width);
    else
        hipLaunchKernelGGL(k_cat_shift<false>, grid, dim3(256), 0, ctx.stream, pool, slot_stride, off, slots, newrows, ldn, cat, B, L, Tc, width);
    K2_HIP(hipGetLastError());
}
namespace {
// phase 0: cat[b][r] = r < L ? cache[slot b][r] : new[b][r - L];  phase 1: cache[slot b][r] = cat[b][row0 + r], r < L
template <int PHASE>
__global__ void k_cat_keep(float* __restrict__ pool, long long slot_stride, long long off, const int* __restrict__ slots,
                           const float* __restrict__ newrows, int ldn, float* __restrict__ cat, int L, int Tc, int w4, int row0, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int c = (int)(i % w4) * 4;
    const long long br = i / w4;
    const int rows = PHASE == 0 ? L + Tc : L;
    const int r = (int)(br % rows), b = (int)(br / rows);
    float* cache = pool + (long long)slots[b] * slot_stride + off;
    float* ct = cat + (long long)b * (L + Tc) * (w4 * 4);
    if (PHASE == 0) {
        const float4 v = r < L ? *reinterpret_cast<const float4*>(cache + (long long)r * (w4 * 4) + c)
                               : *reinterpret_cast<const float4*>(newrows + ((long long)b * Tc + (r - L)) * ldn + c);
        *reinterpret_cast<float4*>(ct + (long long)r * (w4 * 4) + c) = v;
    } else {
        *reinterpret_cast<float4*>(cache + (long long)r * (w4 * 4) + c) = *reinterpret_cast<const float4*>(ct + (long long)(row0 + r) * (w4 * 4) + c);
    }
}
}  // namespace
void cat_keep(const Ctx& ctx, float* pool, long long slot_stride, long long off, const int* slots, const float* newrows, int ldn, float* cat,
              int B, int L, int Tc, int width, int keep_back) {
    K2_REQUIRE(width % 4 == 0 && ldn % 4 == 0 && keep_back >= 0 && keep_back <= Tc, "cat_keep: width %d / ld %d / keep_back %d", width, ldn, keep_back);
    if (ctx.dry) return;
    const int w4 = width / 4;
    const long long n0 = (long long)B * (L + Tc) * w4, n1 = (long long)B * L * w4;
    hipLaunchKernelGGL(k_cat_keep<0>, dim3(nb(n0, 256)), dim3(256), 0, ctx.stream, pool, slot_stride, off, slots, newrows, ldn, cat, L, Tc, w4, 0, n0);
    hipLaunchKernelGGL(k_cat_keep<1>, dim3(nb(n1, 256)), dim3(256), 0, ctx.stream, pool, slot_stride, off, slots, newrows, ldn, cat, L, Tc, w4,
                       Tc - keep_back, n1);
    K2_HIP(hipGetLastError());
}
void attn_stream_ring(const Ctx& ctx, const float* qkp, int ld, const RingRef& keys, const float* pp, const long long* plen, float* aw, int B,
                      int Tc, int L, int KLp, int H, int ds, int left50) {
    ctx.add_flops(0.0, 2.0 * 36 * (double)Tc * (L + Tc) * B * H, 0);
    if (ctx.dry) return;
    size_t lds = sizeof(float) * Tc * (L + Tc);
    hipLaunchKernelGGL(k_attn_stream_ring, dim3(B, H), dim3(256), lds, ctx.stream, qkp, ld, keys, pp, plen, aw, B, Tc, L, KLp, H, ds, left50);
    K2_HIP(hipGetLastError());
}
void attn_stream(const Ctx& ctx, const float* qkp, int ld, const float* kcat, const float* pp, const long long* plen,
                 float* aw, int B, int Tc, int L, int KLp, int H, int ds, int left50) {
    ctx.add_flops(0.0, 2.0 * 36 * (double)Tc * (L + Tc) * B * H, 0);
    if (ctx.dry) return;
    size_t lds = sizeof(float) * Tc * (L + Tc);
    hipLaunchKernelGGL(k_attn_stream, dim3(B, H), dim3(256), lds, ctx.stream, qkp, ld, kcat, pp, plen, aw, B, Tc, L, KLp, H, ds, left50);
    K2_HIP(hipGetLastError());
}
void glu_causal_conv(const Ctx& ctx, const float* x2, float* pool, long long slot_stride, long long off, const int* slots,
                     const float* wc, const float* bc, const float* ww, const float* bw, const float* sc, float* y, int B,
                     int Tc, int D, int K) {
    ctx.add_flops(0.0, 2.0 * B * Tc * (double)D * (K + (K + 1) / 2), 0);
    if (ctx.dry) return;
    size_t lds = sizeof(float) * (K / 2 + Tc) * 64;
    const dim3 grid(cdiv(D, 64), B);
    {   // the lane's column in registers for the (K, Tc) pairs of the model zoo; any other pair: the LDS form below
#define K2_GCR(KT, TC)                                                                                                              \
    if (K == KT && Tc == TC) {                                                                                                      \
        hipLaunchKernelGGL((k_glu_causal_conv_reg<KT, TC>), grid, dim3(64), 0, ctx.stream, x2, pool, slot_stride, off, slots, wc, bc, ww, bw, \
                           sc, y, B, D);                                                                                            \
        K2_HIP(hipGetLastError());                                                                                                  \
        return;                                                                                                                     \
    }
        K2_GCR(31, 16) K2_GCR(31, 8) K2_GCR(31, 4) K2_GCR(31, 2) K2_GCR(15, 16) K2_GCR(15, 8) K2_GCR(15, 4) K2_GCR(15, 2)
#undef K2_GCR
    }
#define K2_GCC(KT) hipLaunchKernelGGL(k_glu_causal_conv<KT>, grid, dim3(64), lds, ctx.stream, x2, pool, slot_stride, off, slots, wc, bc, ww, bw, sc, y, B, Tc, D, K)
    switch (K) {
        case 31: K2_GCC(31); break;
        case 15: K2_GCC(15); break;
        case 7: K2_GCC(7); break;
        default: K2_GCC(0); break;
    }
#undef K2_GCC
    K2_HIP(hipGetLastError());
}
void fifo_append(const Ctx& ctx, float* fifo, int cap, int feat, const float* src, const int* slots, const int* pos, int G, int nf) {
    K2_REQUIRE(feat % 4 == 0, "fifo: feature dimension %d", feat);
    if (ctx.dry || G <= 0 || nf <= 0) return;
    hipLaunchKernelGGL(k_fifo_append, dim3(std::max(1, std::min(8, (nf * (feat / 4) + 255) / 256)), G), dim3(256), 0, ctx.stream, fifo, cap, feat / 4, src,
                       slots, pos, nf);
    K2_HIP(hipGetLastError());
}
void fifo_gather(const Ctx& ctx, const float* fifo, int cap, int feat, const int* slots, const int* head, float* x, int B, int T) {
    K2_REQUIRE(feat % 4 == 0, "fifo: feature dimension %d", feat);
    if (ctx.dry || B <= 0) return;
    hipLaunchKernelGGL(k_fifo_gather, dim3(std::max(1, std::min(8, (T * (feat / 4) + 255) / 256)), B), dim3(256), 0, ctx.stream, fifo, cap, feat / 4,
                       slots, head, x, T);
    K2_HIP(hipGetLastError());
}
void zero_floats(const Ctx& ctx, float* p, long long n) {
    if (ctx.dry || n <= 0) return;
    hipLaunchKernelGGL(k_zero, dim3(nb(n, 256)), dim3(256), 0, ctx.stream, p, n);
    K2_HIP(hipGetLastError());
}
void logfloor_inplace(const Ctx& ctx, float* x, long long n) {
    if (ctx.dry || n <= 0) return;
    hipLaunchKernelGGL(k_logfloor, dim3(nb(n, 256)), dim3(256), 0, ctx.stream, x, n);
    K2_HIP(hipGetLastError());
}

}  // namespace k2hip
