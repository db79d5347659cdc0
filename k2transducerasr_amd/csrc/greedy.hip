// Stateless decoder, joiner and the on-device greedy search loop.
//
// The reference crosses managed<->ONNXRuntime once per frame (JoinerProj) plus once
// per emitting frame (DecoderProj), with a host argmax in between
// (OfflineRecognizer.cs:216-288).  Here one workgroup owns one stream and runs all
// T' frames without leaving the GPU: tanh(enc_t + dec) -> vocab projection from a
// k-major weight matrix (coalesced 16-byte loads, L2 resident) -> block argmax with
// the reference's tie-break -> emit filter -> token append -> decoder update.
//
// Reference semantics kept (file:line in K2TransducerAsr/OfflineRecognizer.cs):
//   - argmax: `tok = l[tok] > l[k] ? tok : k` for k ascending (:237-240): among
//     equal maxima the LATER index wins.
//   - emit when y != blank(0) && y != unk(2) (:268); the online loop also skips id 1
//     (OnlineRecognizer.cs:181).
//   - at most one symbol per frame (:216-288 has no inner loop; single path :129-134)
//   - batch path: every stream starts from decoder([-1, blank]) (:202-208); when ANY
//     stream emits, the decoder is re-run for ALL streams on the last two entries of
//     their token lists, which for a not-yet-emitting stream are the seeded blanks
//     (:250-258, :278-286).  So such a stream's context is [-1,0] up to and including
//     the batch's first emitting frame t0 and [0,0] after it.  t0 is found by a
//     fully parallel pass (first_emit_frame) and handed to the loop.
#include <climits>

#include "kernels.h"
#include "sweep.h"

namespace k2hip {
namespace {

// h[co] = relu(grouped Conv1d(k = 2) over (emb[y0], emb[y1])), id < 0 -> zero embedding.
//   cpg <= 4 (Zipformer recipes, groups = DD/4): weights [DD][cpg][2], a handful of loads per output.
//   cpg == DD (stateless2 decoder, groups = 1): weights k-major [2*DD][DD], k = ci*2 + tap; the stacked
//   embeddings are staged in xe (LDS, [2*DD]) and every weight load is coalesced across outputs.
__device__ void decoder_conv_narrow(const DecJoinW& w, long long y0, long long y1, float* h, float* xe) {
    const int tid = threadIdx.x, nt = blockDim.x;
    if (w.cpg <= 4) {
        for (int co = tid; co < w.DD; co += nt) {
            const int g0 = (co / w.cpg) * w.cpg;
            float s = 0.f;
            for (int ci = 0; ci < w.cpg; ci++) {
                float e0 = y0 >= 0 ? w.emb[y0 * w.DD + g0 + ci] : 0.f;
                float e1 = y1 >= 0 ? w.emb[y1 * w.DD + g0 + ci] : 0.f;
                s += w.conv[(co * w.cpg + ci) * 2 + 0] * e0;
                s += w.conv[(co * w.cpg + ci) * 2 + 1] * e1;
            }
            h[co] = fmaxf(s, 0.f);
        }
    } else if (w.ptab) {
        const float* p0 = w.ptab + y0 * w.DD;
        const float* p1 = w.ptab + ((long long)w.V + y1) * w.DD;
        for (int co = tid; co < w.DD; co += nt) h[co] = fmaxf((y0 >= 0 ? p0[co] : 0.f) + (y1 >= 0 ? p1[co] : 0.f), 0.f);
    } else {
        for (int c = tid; c < w.DD; c += nt) {
            xe[2 * c] = y0 >= 0 ? w.emb[y0 * w.DD + c] : 0.f;
            xe[2 * c + 1] = y1 >= 0 ? w.emb[y1 * w.DD + c] : 0.f;
        }
        __syncthreads();
        for (int co = tid; co < w.DD; co += nt) {
            float s = 0.f;
            for (int k = 0; k < 2 * w.DD; k++) s += xe[k] * w.conv[(long long)k * w.DD + co];
            h[co] = fmaxf(s, 0.f);
        }
    }
    __syncthreads();
}

// dec_out[J] = decoder_proj(relu(grouped_conv(emb[y0], emb[y1])))   (block-wide, 256 threads)
//   h: LDS scratch [3*DD]; out: LDS or global [J]
__device__ void decoder_block(const DecJoinW& w, long long y0, long long y1, float* h, float* out) {
    const int tid = threadIdx.x;
    decoder_conv_narrow(w, y0, y1, h, h + w.DD);
    // eight weight loads in flight per thread; the sum keeps its k-ascending order (one dependent load per k made the single
    // workgroup of the t0 pre-pass a 118 us kernel)
    for (int n = tid; n < w.J; n += blockDim.x) {
        float s = w.dproj_b[n];
        const float* __restrict__ wk = w.dproj_kn + n;
        int k = 0;
        for (; k + 8 <= w.DD; k += 8) {
            float wv[8];
#pragma unroll
            for (int u = 0; u < 8; u++) wv[u] = wk[(long long)(k + u) * w.J];
#pragma unroll
            for (int u = 0; u < 8; u++) s += h[k + u] * wv[u];
        }
        for (; k < w.DD; k++) s += h[k] * wk[(long long)k * w.J];
        out[n] = s;
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void k_decoder(DecJoinW w, const long long* __restrict__ y, float* __restrict__ dec_out) {
    extern __shared__ float sm[];
    int n = blockIdx.x;
    decoder_block(w, y[n * 2], y[n * 2 + 1], sm, dec_out + (long long)n * w.J);
}

__global__ void k_tanh_add(const float* __restrict__ enc, const float* __restrict__ dec, int dec_stride,
                           float* __restrict__ y, long long n4, int J4) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    long long r = i / J4;
    int q = (int)(i % J4);
    float4 e = reinterpret_cast<const float4*>(enc)[i];
    float4 d = *reinterpret_cast<const float4*>(dec + r * dec_stride + 4 * q);
    reinterpret_cast<float4*>(y)[i] = make_float4(tanhf(e.x + d.x), tanhf(e.y + d.y), tanhf(e.z + d.z), tanhf(e.w + d.w));
}

// (value, index) max with "later index wins on ties"
__device__ __forceinline__ void amax_merge(float& v, int& i, float ov, int oi) {
    if (ov > v || (ov == v && oi > i)) { v = ov; i = oi; }
}
// Data-parallel-primitive lane moves (row = 16 lanes): no LDS crossbar, unlike __shfl_xor's ds_bpermute_b32 (a 6-step butterfly over
// two values is a dependent chain of ~900 cycles)
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) {
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, false);
}
// the wave's (value, index) maximum in every lane (amax_merge is a total order: the sequence of merges does not matter)
__device__ __forceinline__ void amax_wave(float& v, int& i) {
    amax_merge(v, i, dpp_f<0xB1>(v), dpp_i<0xB1>(i));     // quad_perm [1,0,3,2]
    amax_merge(v, i, dpp_f<0x4E>(v), dpp_i<0x4E>(i));     // quad_perm [2,3,0,1]
    amax_merge(v, i, dpp_f<0x141>(v), dpp_i<0x141>(i));   // row_half_mirror
    amax_merge(v, i, dpp_f<0x140>(v), dpp_i<0x140>(i));   // row_mirror: every lane holds its row of 16
    float rv = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
    int ri = __builtin_amdgcn_readlane(i, 0);
#pragma unroll
    for (int row = 1; row < 4; row++)
        amax_merge(rv, ri, __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16 * row)), __builtin_amdgcn_readlane(i, 16 * row));
    v = rv;
    i = ri;
}

// ---- the reference's argmax WITH its NaN behaviour --------------------------------------------------------------
// OfflineRecognizer.cs:150-154 / :236-240 / OnlineRecognizer.cs:159-163 scan a row with
//     token_num = logits[j, token_num] > logits[j, k] ? token_num : k;        (k = 1 .. V-1, token_num = 0 at the start)
// A comparison with a NaN is false, so a NaN at index p takes the scan to p and the element behind it takes it on to p + 1 whatever
// its value: the result is the later-wins argmax over the indexes BEHIND the row's last NaN, or that NaN's own index if it is the
// row's last element ([5, NaN, 1] -> 2, [0, 1, NaN] -> 2, all NaN -> V - 1).  amax_merge above ignores NaNs (it is a total order over
// the non-NaN elements, which is what lets lanes take interleaved columns); two ways to put the NaN rule on top of it:
//   * rows that can be read again (k_argmax_rows, k_rounds_step): amax_row_fix -- the lanes also track the largest NaN index they
//     saw; without one (every row of a healthy model) the total-order result stands, otherwise the row is scanned once more behind it;
//   * values that exist only once (k_greedy's sweep): rmax_* -- a (value, index, saw-a-NaN) state per CONTIGUOUS index range that is
//     the reference's scan of that range, merged in index order.
constexpr int kRmaxNan = 1 << 30, kRmaxIdx = kRmaxNan - 1;
// the scan's next element (ranges are scanned in increasing index); empty state: v = -inf, i = -1
__device__ __forceinline__ void rmax_scan(float& v, int& i, float x, int col) {
    const int seen = (i < 0 ? 0 : (i & kRmaxNan)) | ((x != x) ? kRmaxNan : 0);
    if (!(v > x)) { v = x; i = col; }   // the reference's comparison: false for a tie (later index wins) and for a NaN on either side
    i = (i & kRmaxIdx) | seen;
}
// two scanned ranges -> the scan of their union.  The ranges are disjoint and contiguous, so the states' own indexes say which is the
// earlier one; the merge is symmetric and associative over ranges in any grouping (butterflies, trees).
__device__ __forceinline__ void rmax_merge(float& v, int& i, float ov, int oi) {
    if (oi < 0) return;
    if (i < 0) { v = ov; i = oi; return; }
    const bool o_later = (oi & kRmaxIdx) > (i & kRmaxIdx);
    const float av = o_later ? v : ov, bv = o_later ? ov : v;   // a = earlier range, b = later range
    const int ai = o_later ? i : oi, bi = o_later ? oi : i;
    if (bi & kRmaxNan) { v = bv; i = bi; }                      // a NaN in the later range: nothing in front of it survives
    else if (!(av > bv)) { v = bv; i = bi | (ai & kRmaxNan); }  // (av NaN = the earlier range ended on its NaN: the scan moves on)
    else { v = av; i = ai; }
}
__device__ __forceinline__ void rmax_wave(float& v, int& i) {   // lanes = consecutive ranges; every lane gets the wave's state
    rmax_merge(v, i, dpp_f<0xB1>(v), dpp_i<0xB1>(i));
    rmax_merge(v, i, dpp_f<0x4E>(v), dpp_i<0x4E>(i));
    rmax_merge(v, i, dpp_f<0x141>(v), dpp_i<0x141>(i));
    rmax_merge(v, i, dpp_f<0x140>(v), dpp_i<0x140>(i));
    float rv = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
    int ri = __builtin_amdgcn_readlane(i, 0);
#pragma unroll
    for (int row = 1; row < 4; row++)
        rmax_merge(rv, ri, __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16 * row)), __builtin_amdgcn_readlane(i, 16 * row));
    v = rv;
    i = ri;
}
__device__ __forceinline__ int rmax_index(int i) { return i < 0 ? i : (i & kRmaxIdx); }

// the wave's largest value of a per-lane int (DPP moves, every lane gets it)
__device__ __forceinline__ int imax_wave(int x) {
    x = max(x, dpp_i<0xB1>(x));
    x = max(x, dpp_i<0x4E>(x));
    x = max(x, dpp_i<0x141>(x));
    x = max(x, dpp_i<0x140>(x));
    int r = __builtin_amdgcn_readlane(x, 0);
#pragma unroll
    for (int row = 1; row < 4; row++) r = max(r, __builtin_amdgcn_readlane(x, 16 * row));
    return r;
}
// One wave, one row l[0 .. V) that can be read again: (v, idx) = the wave's NaN-ignoring later-wins argmax (amax_wave's result),
// pnan = this lane's largest NaN index (-1: none).  Returns the reference's token (wave-uniform).
__device__ __forceinline__ int amax_row_fix(const float* __restrict__ l, int V, int lane, int idx, int pnan) {
    const int p = imax_wave(pnan);
    if (p < 0) return idx;          // no NaN in the row: every row of a healthy model
    if (p == V - 1) return p;       // the scan ends on the NaN
    float v = -INFINITY;
    int i2 = -1;
    for (int k = p + 1 + lane; k < V; k += 64) amax_merge(v, i2, l[k], k);   // (NaN-free by the choice of p)
    amax_wave(v, i2);
    return i2;
}

// one wave per row; FIRST = false: the transducer loops' scan (later index wins ties, NaN rule above); true: CTC's
// Array.IndexOf(row, row.Max()) (OfflineRecognizer.cs:335,396) -- the first index of the maximum; Enumerable.Max orders NaN below
// every number, so a NaN only wins a row of nothing but NaNs, where IndexOf finds the first element
template <bool FIRST>
__global__ void k_argmax_rows(const float* __restrict__ logits, int ld, int N, int V, int* __restrict__ tok) {
    int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= N) return;
    int lane = threadIdx.x & 63;
    const float* l = logits + (long long)row * ld;
    float v = -INFINITY;
    int idx = -1;
    if (!FIRST) {
        int pnan = -1;
        for (int k = lane; k < V; k += 64) {
            const float x = l[k];
            amax_merge(v, idx, x, k);
            pnan = (x != x) ? k : pnan;
        }
        amax_wave(v, idx);
        idx = amax_row_fix(l, V, lane, idx, pnan);
    } else {
        for (int k = lane; k < V; k += 64) {
            const float x = l[k];
            if (idx < 0 || x > v || (v != v && x == x)) { v = x; idx = k; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            float ov = __shfl_xor(v, o);
            int oi = __shfl_xor(idx, o);
            const bool onan = ov != ov, mnan = v != v;
            if (oi >= 0 && (idx < 0 || ov > v || (mnan && !onan) || ((ov == v || (mnan && onan)) && oi < idx))) { v = ov; idx = oi; }
        }
    }
    if (lane == 0) tok[row] = idx;
}

// in-place row log_softmax: x - max - log(sum(exp(x - max))), one wave per row
__global__ void k_log_softmax_rows(float* __restrict__ x, int M, int V) {
    int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= M) return;
    int lane = threadIdx.x & 63;
    float* l = x + (long long)row * V;
    float mx = -INFINITY;
    for (int k = lane; k < V; k += 64) mx = fmaxf(mx, l[k]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float s = 0.f;
    for (int k = lane; k < V; k += 64) s += expf(l[k] - mx);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float lse = logf(s);
    for (int k = lane; k < V; k += 64) l[k] = l[k] - mx - lse;
}

// CTC collapse, one thread per stream (T' is a few hundred frames; the argmax above is the parallel part)
__global__ void k_ctc_collapse(const int* __restrict__ tok, int B, int Tp, const int* __restrict__ frame_off,
                               long long* __restrict__ tokens, int* __restrict__ timestamps, int* __restrict__ n_tokens,
                               int max_tokens, int* __restrict__ trail, int* __restrict__ any, int* __restrict__ overflow) {
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    int prev = -1, n = 0, tr = 0, an = 0;
    const int fo = frame_off ? frame_off[b] : 0;
    for (int t = 0; t < Tp; t++) {
        const int y = tok[(long long)b * Tp + t];
        if (y == K2HIP_BLANK_ID) tr++;
        else { tr = 0; an = 1; }
        if (y != K2HIP_BLANK_ID && y != prev) {
            if (n < max_tokens) {
                tokens[(long long)b * max_tokens + n] = y;
                timestamps[(long long)b * max_tokens + n] = t + fo;
            } else {
                *overflow = 1;
            }
            n++;
        }
        prev = y;
    }
    n_tokens[b] = n < max_tokens ? n : max_tokens;
    trail[b] = tr;
    any[b] = an;
}

__global__ void k_first_emit(const int* __restrict__ tok, int B, int Tp, int skip1, int* __restrict__ t0) {
    __shared__ int best;
    if (threadIdx.x == 0) best = INT_MAX;
    __syncthreads();
    int mine = INT_MAX;
    for (long long i = threadIdx.x; i < (long long)B * Tp; i += blockDim.x) {
        int y = tok[i];
        bool emit = (y != K2HIP_BLANK_ID && y != K2HIP_UNK_ID && !(skip1 && y == 1));
        if (emit) mine = min(mine, (int)(i % Tp));
    }
    atomicMin(&best, mine);
    __syncthreads();
    if (threadIdx.x == 0) *t0 = best;
}

// ---- multi-frame greedy loop ---------------------------------------------------------
// One or more workgroups of GT threads per stream.  The joiner matrix (J x V f32, 1 MB for the large-en model) does not fit in
// LDS, and sweeping it from L2 once per frame made the loop ~50 us per frame.  Blank wins most frames and the decoder context
// only changes on an emission, so each ROUND evaluates the next GF frames against the CURRENT context in one sweep of the matrix
// (on the matrix pipe: sweep.h), then accepts frames in order up to and including the first one that emits.  Frames after an
// emission are re-evaluated in the next round under the new context, so the result is exactly the frame-by-frame loop of
// OfflineRecognizer.cs:216-288.

// out[n] = f(bias[n] + sum_k x[k] * W[k*N + n]) for a k-major matrix, GT threads: 8 k slices x N/4 column
// groups, 8 float4 weight loads in flight per thread (a dependent load per FMA made this ~50 us per
// emission), partials combined through LDS.  x: LDS [K]; scratch: LDS >= 8*N floats; N % 4 == 0.
__device__ void gemv_kn_wide(const float* x, int K, const float* __restrict__ W, const float* __restrict__ bias, int N,
                             float* scratch, float* out, bool relu) {
    const int tid = threadIdx.x;
    const int ncg = N >> 2, kslice = (K + 7) >> 3;
    for (int u = tid; u < 8 * ncg; u += GT) {
        const int ks = u / ncg, cg = u - ks * ncg;
        const int k0 = ks * kslice, k1 = min(k0 + kslice, K);
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int kb = k0; kb < k1; kb += 16) {  // 16 weight loads in flight (8: 19 us per 512 x 512 product, once per emission)
            float4 wv[16];
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const int k = min(kb + i, k1 - 1);
                wv[i] = *reinterpret_cast<const float4*>(W + (long long)k * N + 4 * cg);
            }
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const float hv = (kb + i < k1) ? x[kb + i] : 0.f;
                s.x += hv * wv[i].x; s.y += hv * wv[i].y; s.z += hv * wv[i].z; s.w += hv * wv[i].w;
            }
        }
        *reinterpret_cast<float4*>(scratch + ks * N + 4 * cg) = s;
    }
    __syncthreads();
    for (int n = tid; n < N; n += GT) {
        float s = bias ? bias[n] : 0.f;
#pragma unroll
        for (int ks = 0; ks < 8; ks++) s += scratch[ks * N + n];
        out[n] = relu ? fmaxf(s, 0.f) : s;
    }
    __syncthreads();
}

// h: LDS [3*DD] (h | stacked embeddings); scratch: LDS >= 8*max(J, DD) floats
// the decoder's front end (embedding + grouped Conv1d(k = 2) + ReLU) -> h[DD]
__device__ void decoder_conv_wide(const DecJoinW& w, long long y0, long long y1, float* h, float* scratch) {
    const int tid = threadIdx.x;
    // (the embedding rows are loaded UNCONDITIONALLY from clamped ids and masked afterwards: a load under `y >= 0 ? ... : 0` is a
    // branch with its own wait, and a thread's 2 cpg loads then run one after the other)
    const long long z0 = y0 >= 0 ? y0 : 0, z1 = y1 >= 0 ? y1 : 0;
    const float m0 = y0 >= 0 ? 1.f : 0.f, m1 = y1 >= 0 ? 1.f : 0.f;
    if (w.cpg <= 4) {
        for (int co = tid; co < w.DD; co += GT) {
            const int g0 = (co / w.cpg) * w.cpg;
            float e0[4], e1[4];
#pragma unroll
            for (int ci = 0; ci < 4; ci++) {
                const int cc = min(ci, w.cpg - 1);
                e0[ci] = w.emb[z0 * w.DD + g0 + cc];
                e1[ci] = w.emb[z1 * w.DD + g0 + cc];
            }
            float s = 0.f;
            for (int ci = 0; ci < w.cpg; ci++) {
                s += w.conv[(co * w.cpg + ci) * 2 + 0] * (e0[ci] * m0);
                s += w.conv[(co * w.cpg + ci) * 2 + 1] * (e1[ci] * m1);
            }
            h[co] = fmaxf(s, 0.f);
        }
        __syncthreads();
    } else if (w.ptab) {
        const float* p0 = w.ptab + z0 * w.DD;
        const float* p1 = w.ptab + ((long long)w.V + z1) * w.DD;
        for (int co = tid; co < w.DD; co += GT) h[co] = fmaxf(p0[co] * m0 + p1[co] * m1, 0.f);
        __syncthreads();
    } else {
        float* xe = h + w.DD;
        for (int c = tid; c < w.DD; c += GT) {
            xe[2 * c] = y0 >= 0 ? w.emb[y0 * w.DD + c] : 0.f;
            xe[2 * c + 1] = y1 >= 0 ? w.emb[y1 * w.DD + c] : 0.f;
        }
        __syncthreads();
        gemv_kn_wide(xe, 2 * w.DD, w.conv, nullptr, w.DD, scratch, h, true);
    }
}
__device__ void decoder_block_wide(const DecJoinW& w, long long y0, long long y1, float* h, float* scratch, float* out) {
    decoder_conv_wide(w, y0, y1, h, scratch);
    gemv_kn_wide(h, w.DD, w.dproj_kn, w.dproj_b, w.J, scratch, out, false);
}
// Small vocabularies: every context's decoder output is in the model's table (decoder_table below, built by this file's own
// arithmetic: the same bits) and a decoder update is one row read.
__device__ __forceinline__ bool decoder_in_table(const DecJoinW& w, long long y0, long long y1) {
    return w.dec_table && y0 >= -1 && y0 < w.V && y1 >= 0 && y1 < w.V;
}
__device__ void decoder_lookup(const DecJoinW& w, long long y0, long long y1, float* out) {
    const float* row = w.dec_table + ((y0 + 1) * w.V + y1) * (long long)w.J;
    for (int n = 4 * threadIdx.x; n < w.J; n += 4 * GT) *reinterpret_cast<float4*>(out + n) = *reinterpret_cast<const float4*>(row + n);
    __syncthreads();
}
// out[J] (LDS, 16-byte aligned) = decoder(y0, y1): the table's row, or computed here
__device__ void decoder_context(const DecJoinW& w, long long y0, long long y1, float* h, float* scratch, float* out) {
    if (decoder_in_table(w, y0, y1)) decoder_lookup(w, y0, y1, out);
    else decoder_block_wide(w, y0, y1, h, scratch, out);
}

// ---- vocabulary-parallel exchange ----------------------------------------------------------------
// With a large vocabulary (V = 5537: an 11 MB joiner matrix) one workgroup per stream spends ~300 us per
// sweep pulling the matrix through a single CU.  The sweep is therefore split over `parts` workgroups per
// stream (column slabs); after each sweep every part publishes its per-frame (max, argmax) as two 8-byte
// {epoch, value} granules -- relaxed agent-scope atomic stores (write-through), the tag IS the flag
// (cdna_hip_programming.md Guideline 16, recipe R2) -- and wave 0 of every part polls the stream's
// granules until all carry this round's epoch.  All parts then take the same decision and run the same
// decoder update, so no further exchange is needed.  Granules are double-buffered by round parity: a part
// can only be one round ahead of its slowest peer.  Spins are bounded; a timeout sets *overflow = 2.
typedef __attribute__((address_space(1))) unsigned long long gu64;
constexpr int kMaxParts = 16;
constexpr unsigned kSpinLimit = 1u << 22;

__device__ __forceinline__ void store_granule(unsigned long long* g, unsigned epoch, unsigned value) {
    __hip_atomic_store((gu64*)g, ((unsigned long long)epoch << 32) | value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long load_granule(const unsigned long long* g) {
    return __hip_atomic_load((gu64*)g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- large vocabularies: the round's sweep as an f16 SCREEN + an exact f32 re-check of the few columns that can win ------------
// A round's three f32 passes over a 1.4 MB slab (V = 5537, 8 parts) are 70 % of that search.  Only the argmax of each frame's logits
// is wanted, so the slab is first swept in f16 on v_mfma_f32_16x16x32_f16 (half the bytes, 1/16 of the matrix instructions, no k
// split, no exchange of partial sums through LDS): l~[f][v] with |l~ - l| <= eps[v], a bound fixed at load time (model.cpp) for
// the f32 logit l the passes would compute.  A column can only hold frame f's maximum if l~ + eps >= max_u (l~_u - eps_u); those
// candidates -- a handful -- are recomputed EXACTLY as the passes compute them (the k slices' fma chains in k order, the slices
// summed in the passes' tree, + bias: bit-identical values), and the later-wins argmax over them is the argmax over the slab.
// Non-finite screen values (a NaN sample upstream, a NaN / Inf / f16-overflowing weight: the reference's NaN rule needs every column)
// or more than kScreenCand candidates send the round to the f32 passes.  Per round and part: 708 KB of L2 traffic instead of 1.4 MB
// x the three passes' MFMA-issue time.
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
constexpr int kScreenCand = 64;          // (frame, column) pairs re-checked per round and part: one per (pair, k slice) thread
constexpr int kScreenMaxCols = 1536;     // scr[GF][cols] + the lists below live in the psum area (kPsumFloats)
static_assert(GF * kScreenMaxCols + 64 + 8 + kScreenCand + kScreenCand * 8 <= kPsumFloats, "screen: LDS layout");

// actH: the round's activations as f16 A fragments [J / 32][64 lanes][8] (rows >= GF of the 16-row tile stay zero).
// Returns true (workgroup-uniform) when the screen decided the round: then lane 0 of wave f < nf holds frame f's (bestv, besti) over
// the part's columns [4 cg0, 4 cg1); false: nothing was decided, run the f32 passes.
// (a real call: inlined, the four instantiations' loop-invariant addresses stay live across the whole search loop and the kernel spills)
template <int NS>   // NS = J / 32 K steps: a compile-time constant, so that the tile loop is straight-line code with counted waits
__device__ __noinline__ bool screen_round(const DecJoinW& w, const float* actT, const _Float16* actH, float* area, int cg0, int cg1, int nf,
                                             float& bestv, int& besti, unsigned long long* st) {
    unsigned long long tprev = st ? __builtin_amdgcn_s_memrealtime() : 0;
    auto stamp = [&](int slot) {   // (wave-uniform: st is a kernel argument)
        if (st) {
            const unsigned long long now = __builtin_amdgcn_s_memrealtime();
            if (threadIdx.x == 0) st[slot] += now - tprev;
            tprev = now;
        }
    };
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (uniform: the tile loop's tests stay scalar)
    const int c_lo = 4 * cg0, c_hi = min(4 * cg1, w.V), ncols = c_hi - c_lo;
    // (this is a real call: the LDS pointers arrive as generic pointers -- cast back, or every access is a flat instruction whose
    // wait also drains the global loads in flight)
    typedef __attribute__((address_space(3))) float lfloat;
    typedef __attribute__((address_space(3))) int lint;
    lfloat* scr = (lfloat*)area;                                    // [GF][ncols]: upper bounds l~ + eps (-inf outside the slab)
    lfloat* wlb = scr + GF * kScreenMaxCols;                        // [8 waves][GF]: the waves' largest lower bounds l~ - eps
    lint* ctl = (lint*)(wlb + 64);                                  // [0] candidate count, [1] non-finite flag
    lint* cand = ctl + 8;                                           // [kScreenCand]: frame | local column << 3
    lfloat* part = (lfloat*)(cand + kScreenCand);                   // [kScreenCand][8 slices]
    const lfloat* actL = (const lfloat*)actT;
    // (ctl[0 .. 1] were zeroed by the caller IN FRONT of the barrier that precedes this call: zeroed here, a wave that finished its tile
    // sweep early could raise the non-finite flag before wave 0's late store cleared it again)
    // ---- the screen: wave = every 8th 16-column tile; the A fragments (the activations) are re-read from LDS per tile -- held in
    // registers they are 64 VGPRs on top of the B ring's 64 and the kernel spills
    typedef __attribute__((address_space(3))) const h16x8 lh16x8;   // (LDS, not flat: a flat load's wait also drains the global loads in flight)
    lh16x8* afr = (lh16x8*)actH + lane;
    const int t0 = c_lo >> 4, t1 = (c_hi + 15) >> 4;
    typedef __attribute__((address_space(1))) const h16x8 gh16x8;   // (global, not flat: flat loads count on lgkmcnt too)
    gh16x8* wh = (gh16x8*)w.out_h16;
    float lbmax[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    bool bad = false;
    // bias and bound of the lane's column: requested one tile ahead, like the B fragments (fetched behind the MFMAs they were a
    // dependent global round trip per tile)
    const float* bias_g = w.out_b;
    const float* eps_g = w.out_eps;
    auto colof = [&](int t) { return min(16 * t + (lane & 15), w.V - 1); };
    float bvn = bias_g[colof(min(t0 + wave, t1 - 1))], en = eps_g[colof(min(t0 + wave, t1 - 1))];
    // B fragments: a ring of one register set -- step s of the NEXT tile is requested as soon as step s of this tile has been
    // used, so a wave keeps a whole tile (16 KB at J = 512) in flight without a second set (two sets spilled: 256 VGPRs + scratch)
    // (no test inside the loop: the wave's last tile re-requests itself; a conditional load costs the counted waits)
    h16x8 bfr[NS];
    auto tile = [&](int t) {
        const int tn = min(t + GT / 64, t1 - 1);
        gh16x8* wn = wh + ((size_t)tn * NS) * 64 + lane;
        const float bv = bvn, e = en;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        constexpr int AG = NS >= 4 ? 4 : NS;   // A fragments per LDS round trip
#pragma unroll
        for (int s0 = 0; s0 < NS; s0 += AG) {
            h16x8 ag[AG];
#pragma unroll
            for (int q = 0; q < AG; q++) ag[q] = afr[(s0 + q) * 64];
#pragma unroll
            for (int q = 0; q < AG; q++) {
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ag[q], bfr[s0 + q], acc, 0, 0, 0);
                bfr[s0 + q] = wn[(s0 + q) * 64];
            }
            __builtin_amdgcn_sched_barrier(0);   // (keeps the reloads behind their MFMAs: hoisted, they need a second register set)
        }
        bvn = bias_g[colof(tn)];
        en = eps_g[colof(tn)];
        // C/D: column = lane & 15, row (frame) = 4 (lane >> 4) + i: frames 0 .. 7 sit in lanes 0 .. 31
        const int col = 16 * t + (lane & 15);
        if (lane < 32 && col >= c_lo && col < c_hi) {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int f = 4 * (lane >> 4) + i;
                const float v = acc[i] + bv;
                bad = bad || !(fabsf(v) <= 3.0e38f) || !(e <= 3.0e38f);
                scr[f * ncols + (col - c_lo)] = v + e;
                lbmax[i] = fmaxf(lbmax[i], v - e);
            }
        }
    };
    {
        gh16x8* w0 = wh + ((size_t)min(t0 + wave, t1 - 1) * NS) * 64 + lane;
#pragma unroll
        for (int s = 0; s < NS; s++) bfr[s] = w0[s * 64];
    }
#pragma unroll 1
    for (int t = t0 + wave; t < t1; t += GT / 64) tile(t);
#pragma unroll
    for (int i = 0; i < 4; i++) {   // the row of 16 lanes that shares frames 4 (lane >> 4) + i
        float v = lbmax[i];
        v = fmaxf(v, dpp_f<0xB1>(v));
        v = fmaxf(v, dpp_f<0x4E>(v));
        v = fmaxf(v, dpp_f<0x141>(v));
        v = fmaxf(v, dpp_f<0x140>(v));
        if ((lane & 15) == 0 && lane < 32) wlb[wave * GF + 4 * (lane >> 4) + i] = v;
    }
    if (__any(bad) && lane == 0) ctl[1] = 1;
    __syncthreads();
    stamp(2);
    // ---- candidates: every (frame, column) whose upper bound reaches the frame's largest lower bound
    {
        // the eight thresholds first (independent reads), then a thread per column over the frames: as one loop over (frame, column)
        // with the threshold re-read per element this was a chain of ~70 dependent LDS round trips (2.4 us of the round)
        float tau[GF];
#pragma unroll
        for (int f = 0; f < GF; f++) tau[f] = wlb[f];
#pragma unroll
        for (int q = 1; q < GT / 64; q++)
#pragma unroll
            for (int f = 0; f < GF; f++) tau[f] = fmaxf(tau[f], wlb[q * GF + f]);
        for (int c = tid; c < ncols; c += GT) {
            float ub[GF];
#pragma unroll
            for (int f = 0; f < GF; f++) ub[f] = f < nf ? scr[f * ncols + c] : -INFINITY;
#pragma unroll
            for (int f = 0; f < GF; f++)
                if (f < nf && ub[f] >= tau[f]) {
                    const int slot = atomicAdd((int*)&ctl[0], 1);
                    if (slot < kScreenCand) cand[slot] = f | (c << 3);
                }
        }
    }
    __syncthreads();
    stamp(3);
    const int P = ctl[0];
    const bool give_up = ctl[1] || P > kScreenCand || P <= 0;   // (uniform: read from LDS behind the barrier)
    if (give_up) {
        __syncthreads();   // everyone has read the verdict before the f32 passes reuse this area for their partial sums
        return false;
    }
    // ---- the candidates' logits exactly as the f32 passes form them: thread = (pair, k slice), the slice's fma chain in k order
    {
        const int p = tid >> 3, s = tid & 7;
        if (p < P) {
            // the column's weights from the [V][J] layout: the slice is KP contiguous floats, all requested before the chain starts
            // (from the k-major matrix every weight was its own 64-byte sector, fetched in four dependent groups: ~8 us per round)
            constexpr int KP = 4 * NS;   // = kper = J / 8
            const int f = cand[p] & 7, col = c_lo + (cand[p] >> 3);
            const float4* wp = reinterpret_cast<const float4*>(w.out_vj + (long long)col * w.J + s * KP);
            const lfloat* ap = actL + (s * KP) * GF + s * APAD + f;
            float4 wv[KP / 4];
#pragma unroll
            for (int i = 0; i < KP / 4; i++) wv[i] = wp[i];
            float acc = 0.f;
#pragma unroll
            for (int i = 0; i < KP / 4; i++) {
                acc = fmaf(ap[(4 * i + 0) * GF], wv[i].x, acc);
                acc = fmaf(ap[(4 * i + 1) * GF], wv[i].y, acc);
                acc = fmaf(ap[(4 * i + 2) * GF], wv[i].z, acc);
                acc = fmaf(ap[(4 * i + 3) * GF], wv[i].w, acc);
            }
            part[p * 8 + s] = acc;
        }
    }
    __syncthreads();
    bestv = -INFINITY;
    besti = -1;
    if (lane < P && (cand[lane] & 7) == wave) {   // wave f: frame f's candidates, one per lane
        const lfloat* ps = part + lane * 8;
        const int col = c_lo + (cand[lane] >> 3);
        bestv = (((ps[0] + ps[1]) + (ps[2] + ps[3])) + ((ps[4] + ps[5]) + (ps[6] + ps[7]))) + w.out_b[col];
        besti = col;
    }
    amax_wave(bestv, besti);   // finite values: the total order (later index wins ties) is the reference's scan
    stamp(4);
    return true;
}

// LDS: actT[J][GF] | psum[8][GF][256] | dec_a[J] | dec_b[J] | dec_own[J] | h[3*DD] (h | stacked embeddings) | fin[GF] | exchange scratch |
// actH (large vocabularies: the activations as f16 MFMA fragments, J x 32 bytes)
// SCREEN: the instantiation for large vocabularies (w.out_h16 set) -- kept apart so that the screen's 128 fragment registers do
// not weigh on the small-vocabulary kernel's allocation (172 VGPRs, no scratch)
template <bool SCREEN>
__global__ __launch_bounds__(GT) void k_greedy(DecJoinW w, GreedyArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* actT = sm;
    float* psum = actT + w.J * GF + 8 * APAD;   // the k slices' partial sums of one pass (kPsumFloats)
    float* dec_a = psum + kPsumFloats;
    float* dec_b = dec_a + w.J;
    float* dec_own = dec_b + w.J;
    float* h = dec_own + w.J;
    int* fin = reinterpret_cast<int*>(h + 3 * w.DD);
    // exchange scratch lives in the dynamic region too (static LDS would shift its 16-byte alignment)
    float* pv = reinterpret_cast<float*>(fin + GF + 8);
    int* pi = reinterpret_cast<int*>(pv + kMaxParts * GF);
    int* xf = pi + kMaxParts * GF;  // [0] = exchange timed out
    _Float16* actH = reinterpret_cast<_Float16*>(xf + 4);   // (only allocated when the screen is on: greedy_lds_bytes)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int G = a.parts, b = blockIdx.x / G, part = blockIdx.x - b * G;
    unsigned epoch = 0, epoch2 = 0;
    if (tid == 0) xf[0] = 0;
    const float* enc = a.enc + (long long)b * a.Tp * w.J;
    const int t0 = a.t0 ? *a.t0 : INT_MAX;
    const int ncg = w.Vp >> 2, kper = w.J >> 3;
    // this part's slab of column groups (whole 8-wide wave chunks)
    const int cper = ((ncg + G - 1) / G + 7) & ~7;
    const int cg0 = part * cper, cg1 = min(ncg, cg0 + cper);

    // the f16 screen (screen_round): large vocabularies whose slab fits the psum area
    const int nks = w.J >> 5;
    const bool screen = SCREEN && w.out_h16 != nullptr && (w.J & 31) == 0 && (nks == 16 || nks == 8 || nks == 4 || nks == 2) &&
                        4 * (cg1 - cg0) <= kScreenMaxCols && cg1 > cg0;
    if (screen) {
        // (J x 32 bytes = J x 8 words: exactly what greedy_lds_bytes allots -- actH is the last region of the dynamic LDS)
        for (int i = tid; i < w.J * 8; i += GT) reinterpret_cast<unsigned*>(actH)[i] = 0u;   // rows 8 .. 15 of the A tile stay zero
        __syncthreads();
    }

    long long y0 = -1, y1 = K2HIP_BLANK_ID;
    int n_tok = 0, t = 0;
    bool own = false;
    if (a.init_ctx) {
        // online: the loop starts from the stream's own last two tokens (OnlineRecognizer.cs:122-136)
        y0 = a.init_ctx[2 * b];
        y1 = a.init_ctx[2 * b + 1];
        own = true;
        decoder_context(w, y0, y1, h, actT, dec_own);
    } else if (a.dec_init) {
        for (int k = tid; k < 2 * w.J; k += GT) dec_a[k] = a.dec_init[k];  // dec_a | dec_b are adjacent
        __syncthreads();
    } else {
        decoder_context(w, -1, K2HIP_BLANK_ID, h, actT, dec_a);
        if (a.t0) decoder_context(w, K2HIP_BLANK_ID, K2HIP_BLANK_ID, h, actT, dec_b);
    }

    unsigned long long* st = (a.stamps && blockIdx.x == 0) ? a.stamps : nullptr;
    unsigned long long tprev = 0;
    auto stamp = [&](int slot) {
        if (st) {
            const unsigned long long now = __builtin_amdgcn_s_memrealtime();
            if (tid == 0) st[slot] += now - tprev;
            tprev = now;
        }
    };
    while (t < a.Tp && n_tok < a.max_sym) {
        const int nf = min(GF, a.Tp - t);
        if (st) {
            tprev = __builtin_amdgcn_s_memrealtime();
            if (tid == 0) st[0] += 1;
        }
        // activations of the next GF frames under the current context
        // thread = joiner channel k, the GF frames' encoder values requested together (one frame per iteration of a flat
        // index loop was a chain of J GF / GT = 8 dependent global-load latencies per round, ~12 us of a ~55 us round)
        for (int k = tid; k < w.J; k += GT) {
            float e[GF];
#pragma unroll
            for (int f = 0; f < GF; f++) e[f] = enc[(long long)min(t + f, a.Tp - 1) * w.J + k];   // (unconditional, clamped; dropped below when f >= nf)
            const float da = dec_a[k], db = a.t0 ? dec_b[k] : 0.f, dn = dec_own[k];
            float* dst = actT + k * GF + (k / kper) * APAD;
            // (screen) element j = k & 7 of lane f + 16 ((k & 31) >> 3) of K step k >> 5
            _Float16* dh = actH + (((size_t)(k >> 5) * 64 + 16 * ((k & 31) >> 3)) * 8 + (k & 7));
#pragma unroll
            for (int f = 0; f < GF; f++) {
                const float d = own ? dn : ((t + f) > t0 ? db : da);
                const float v = f < nf ? tanhf(e[f] + d) : 0.f;
                dst[f] = v;
                if (screen) dh[f * 8] = (_Float16)v;
            }
        }
        // the screen's candidate count and non-finite flag (screen_round's ctl, behind scr and wlb in the psum area): every thread has
        // left the previous round's reads of them (barriers behind both of its exits), and the barrier below orders this store in front
        // of every wave's `ctl[1] = 1`
        if (screen && tid < 2) reinterpret_cast<int*>(psum + GF * kScreenMaxCols + 64)[tid] = 0;
        __syncthreads();
        stamp(1);
        // The sweep (mfma_sweep_rows): wave = k slice (J / 8 rows), lane = 4 columns -- a pass covers 256 columns of the slab and a
        // wave's load instruction reads 1 KB of one weight row.
        // wave f, after the passes: the reference's scan (rmax_*) of frame f's logits over this part's slab.  Lanes hold consecutive
        // column groups within a pass and the passes walk the slab in order, so a pass is reduced over the wave before the next one
        // joins it (slabs of one pass -- small vocabularies -- keep the single reduction behind the loop)
        float bestv = -INFINITY;
        int besti = -1;
        const bool one_pass = cg1 - cg0 <= 64;
        bool screened = false;
        if constexpr (SCREEN) if (screen) {
            switch (nks) {
                case 16: screened = screen_round<16>(w, actT, actH, psum, cg0, cg1, nf, bestv, besti, st); break;
                case 8: screened = screen_round<8>(w, actT, actH, psum, cg0, cg1, nf, bestv, besti, st); break;
                case 4: screened = screen_round<4>(w, actT, actH, psum, cg0, cg1, nf, bestv, besti, st); break;
                default: screened = screen_round<2>(w, actT, actH, psum, cg0, cg1, nf, bestv, besti, st); break;
            }
        }
        if (screened && cg0 < cg1) __syncthreads();   // (the passes below are skipped: nothing else reads the psum area this round)
        if (st) tprev = __builtin_amdgcn_s_memrealtime();
        for (int cgb = cg0; cgb < (screened ? cg0 : cg1); cgb += 64) {
            const int cg = cgb + lane;
            const bool valid = cg < cg1;
            f32x4 c[2][4];
#pragma unroll
            for (int hh = 0; hh < 2; hh++)
#pragma unroll
                for (int q = 0; q < 4; q++) c[hh][q] = f32x4{0.f, 0.f, 0.f, 0.f};
            mfma_sweep_rows(w.out_kn + (long long)(wave * kper) * w.Vp + 4 * min(cg, cg1 - 1),   // (lanes past the slab: its last group, dropped below)
                            w.Vp, actT + (wave * kper) * GF + wave * APAD + (lane & 3), kper, c);
            psum_store(psum, wave, lane, c);
            __syncthreads();
            {
                // wave f takes frame f: slices summed as the pairwise tree the lane shuffles of the vector version formed
                float4 ps[8];
#pragma unroll
                for (int q = 0; q < 8; q++) ps[q] = *reinterpret_cast<const float4*>(psum + (q * GF + wave) * 256 + 4 * lane);
                float sj[4];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const float p0 = (&ps[0].x)[j], p1 = (&ps[1].x)[j], p2 = (&ps[2].x)[j], p3 = (&ps[3].x)[j];
                    const float p4 = (&ps[4].x)[j], p5 = (&ps[5].x)[j], p6 = (&ps[6].x)[j], p7 = (&ps[7].x)[j];
                    sj[j] = ((p0 + p1) + (p2 + p3)) + ((p4 + p5) + (p6 + p7));
                }
                float pv_ = -INFINITY;
                int pi_ = -1;
                if (valid) {
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const int col = 4 * cg + j;
                        if (col < w.V) rmax_scan(pv_, pi_, sj[j] + w.out_b[col], col);
                    }
                }
                if (one_pass) {
                    bestv = pv_;
                    besti = pi_;
                } else {
                    rmax_wave(pv_, pi_);
                    rmax_merge(bestv, besti, pv_, pi_);
                }
            }
            if (cgb + 64 < cg1) __syncthreads();   // psum is rewritten by the next pass
        }
        if (one_pass && !screened) rmax_wave(bestv, besti);
        stamp(5);
        if (lane == 0) {
            fin[wave] = rmax_index(besti);
            if (G > 1) {  // publish this slab's candidate for frame `wave` (round parity buffer)
                unsigned long long* gr = a.gran + ((((long long)b * 2 + (epoch & 1)) * G + part) * GF + wave) * 2;
                store_granule(gr, epoch + 1, __float_as_uint(bestv));
                store_granule(gr + 1, epoch + 1, (unsigned)besti);
            }
        }
        if (G > 1) {
            epoch++;
            if (wave == 0) {  // one wave polls the stream's G*GF*2 granules of this round
                const unsigned long long* gr = a.gran + (((long long)b * 2 + ((epoch - 1) & 1)) * G) * GF * 2;
                const int ng = G * GF * 2;
                unsigned vals[(kMaxParts * GF * 2) / 64];
                bool ok_all = false;
                for (unsigned spins = 0; spins < kSpinLimit; spins++) {
                    bool ok = true;
#pragma unroll
                    for (int k = 0; k < (kMaxParts * GF * 2) / 64; k++) {
                        const int q = lane + 64 * k;
                        if (q < ng) {
                            const unsigned long long x = load_granule(gr + q);
                            vals[k] = (unsigned)x;
                            ok &= (unsigned)(x >> 32) == epoch;
                        }
                    }
                    if (__all(ok)) { ok_all = true; break; }
                    __builtin_amdgcn_s_sleep(2);
                }
                if (!ok_all && lane == 0) xf[0] = 1;
#pragma unroll
                for (int k = 0; k < (kMaxParts * GF * 2) / 64; k++) {
                    const int q = lane + 64 * k;
                    if (q < ng) {
                        if (q & 1) pi[q >> 1] = (int)vals[k];
                        else pv[q >> 1] = __uint_as_float(vals[k]);
                    }
                }
            }
            __syncthreads();
            if (xf[0]) {  // a peer never arrived: give up (every part of the stream times out the same way)
                if (tid == 0) *a.overflow = 2;
                return;
            }
            if (tid < GF) {
                float v = pv[tid];
                int i = pi[tid];
                for (int p = 1; p < G; p++) rmax_merge(v, i, pv[p * GF + tid], pi[p * GF + tid]);   // (the parts' slabs: consecutive ranges)
                fin[tid] = rmax_index(i);
            }
        }
        __syncthreads();
        stamp(6);
        // accept frames in order up to and including the first emission
        int adv = nf;
        bool emitted = false;
        for (int f = 0; f < nf; f++) {
            const int y = fin[f];
            if (y != K2HIP_BLANK_ID && y != K2HIP_UNK_ID && !(a.skip1 && y == 1)) {
                if (n_tok < a.max_tokens) {
                    if (tid == 0 && part == 0) {
                        a.tokens[(long long)b * a.max_tokens + n_tok] = y;
                        a.timestamps[(long long)b * a.max_tokens + n_tok] = t + f;
                    }
                } else if (tid == 0) {
                    *a.overflow = 1;
                }
                n_tok++;
                y0 = y1;
                y1 = y;
                own = true;
                adv = f + 1;
                emitted = true;
                break;
            }
        }
        t += adv;
        if (emitted && decoder_in_table(w, y0, y1)) {
            decoder_lookup(w, y0, y1, dec_own);   // (every part reads the row: nothing to exchange)
        } else if (emitted && G > 1) {
            // Every part takes the same decision, so the decoder update is shared out as well: a part computes its J / G outputs of
            // decoder_proj (the same k slices summed in the same order as gemv_kn_wide: bit-identical values), publishes them as
            // {emission epoch, value} granules and collects the whole vector.  (Each of the 16 parts of a V = 5537 stream pulling
            // the whole 1 MB matrix through its CU for every emission was ~40 % of that search.)
            decoder_conv_wide(w, y0, y1, h, actT);
            const int ncgJ = w.J >> 2, ca = (part * ncgJ) / G, nc = ((part + 1) * ncgJ) / G - ca, kslice = (w.DD + 7) >> 3;
            for (int u = tid; u < 8 * nc; u += GT) {
                const int ksl = u / nc, cgi = u - ksl * nc;
                const int k0 = ksl * kslice, k1 = min(k0 + kslice, w.DD);
                float4 sacc = make_float4(0.f, 0.f, 0.f, 0.f);
                for (int kb = k0; kb < k1; kb += 16) {
                    float4 wv[16];
#pragma unroll
                    for (int i = 0; i < 16; i++) wv[i] = *reinterpret_cast<const float4*>(w.dproj_kn + (long long)min(kb + i, k1 - 1) * w.J + 4 * (ca + cgi));
#pragma unroll
                    for (int i = 0; i < 16; i++) {
                        const float hv = (kb + i < k1) ? h[kb + i] : 0.f;
                        sacc.x += hv * wv[i].x; sacc.y += hv * wv[i].y; sacc.z += hv * wv[i].z; sacc.w += hv * wv[i].w;
                    }
                }
                *reinterpret_cast<float4*>(actT + ksl * (4 * nc) + 4 * cgi) = sacc;
            }
            __syncthreads();
            unsigned long long* g2 = a.gran2 + ((long long)b * 2 + (epoch2 & 1)) * w.J;
            for (int n = tid; n < 4 * nc; n += GT) {
                float sv = w.dproj_b[4 * ca + n];
#pragma unroll
                for (int q = 0; q < 8; q++) sv += actT[q * (4 * nc) + n];
                store_granule(g2 + 4 * ca + n, epoch2 + 1, __float_as_uint(sv));
            }
            epoch2++;
            bool ok = true;
            for (int n = tid; n < w.J; n += GT) {
                unsigned long long x = load_granule(g2 + n);
                for (unsigned spins = 0; (unsigned)(x >> 32) != epoch2 && spins < kSpinLimit; spins++) {
                    __builtin_amdgcn_s_sleep(1);
                    x = load_granule(g2 + n);
                }
                ok = ok && (unsigned)(x >> 32) == epoch2;
                dec_own[n] = __uint_as_float((unsigned)x);
            }
            if (!ok) xf[0] = 1;
            __syncthreads();
            if (xf[0]) {
                if (tid == 0) *a.overflow = 2;
                return;
            }
        } else if (emitted) {
            decoder_block_wide(w, y0, y1, h, actT, dec_own);
        } else {
            __syncthreads();  // actT / fin are rewritten by the next round
        }
        stamp(7);
    }
    if (tid == 0 && part == 0) a.n_tokens[b] = n_tok < a.max_tokens ? n_tok : a.max_tokens;
}

// decoder outputs of the offline loops' two start contexts, by k_greedy's own routine (so a kernel that loads them computes exactly
// what it would have computed itself)
__global__ __launch_bounds__(GT) void k_decoder_start(DecJoinW w, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* scratch = sm;                       // >= 8 max(J, DD)
    float* h = sm + 8 * max(w.J, w.DD);        // [3 DD]
    float* o = h + 3 * w.DD;                   // [J]
    decoder_context(w, blockIdx.x == 0 ? -1 : K2HIP_BLANK_ID, K2HIP_BLANK_ID, h, scratch, o);
    for (int k = threadIdx.x; k < w.J; k += GT) out[(long long)blockIdx.x * w.J + k] = o[k];
}

// ---- decoder table (small vocabularies) ----------------------------------------------------------------------
// table[(y0 + 1) V + y1][J] = decoder(y0, y1) for every context, y0 in -1 .. V-1.  One workgroup takes GF consecutive contexts: the
// front end of each by decoder_conv_wide itself, then decoder_proj as the search kernels' sweep with contexts in place of frames --
// gemv_kn_wide's 8 k slices, each the same fma chain, added onto the bias in slice order: the bits decoder_block_wide gives.
__global__ __launch_bounds__(GT) void k_decoder_table(DecJoinW w, float* __restrict__ table, long long n_ctx) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* hT = sm;                                   // [DD][GF] (+ APAD per k slice)
    float* psum = hT + w.DD * GF + 8 * APAD;
    float* h = psum + kPsumFloats;                    // [3 DD]
    float* scratch = h + 3 * w.DD;                    // [8 max(J, DD)] (decoder_conv_wide's GEMV branch)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kslice = (w.DD + 7) >> 3;
    const long long c0 = (long long)blockIdx.x * GF;
    for (int f = 0; f < GF; f++) {
        const long long cx = min(c0 + f, n_ctx - 1);
        decoder_conv_wide(w, cx / w.V - 1, cx % w.V, h, scratch);
        for (int k = tid; k < w.DD; k += GT) hT[k * GF + (k / kslice) * APAD + f] = h[k];
        __syncthreads();
    }
    const int k0 = min(wave * kslice, w.DD), nrows = min(k0 + kslice, w.DD) - k0;
    const int ncg = w.J >> 2;
    for (int cgb = 0; cgb < ncg; cgb += 64) {
        const int cg = cgb + lane;
        f32x4 c[2][4];
#pragma unroll
        for (int hh = 0; hh < 2; hh++)
#pragma unroll
            for (int q = 0; q < 4; q++) c[hh][q] = f32x4{0.f, 0.f, 0.f, 0.f};
        mfma_sweep_rows(w.dproj_kn + (long long)k0 * w.J + 4 * min(cg, ncg - 1), w.J, hT + k0 * GF + wave * APAD + (lane & 3), nrows, c);
        psum_store(psum, wave, lane, c);
        __syncthreads();
        if (cg < ncg && c0 + wave < n_ctx) {   // wave f: context c0 + f
            float4 sv = *reinterpret_cast<const float4*>(w.dproj_b + 4 * cg);
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const float4 ps = *reinterpret_cast<const float4*>(psum + (q * GF + wave) * 256 + 4 * lane);
                sv.x += ps.x; sv.y += ps.y; sv.z += ps.z; sv.w += ps.w;
            }
            *reinterpret_cast<float4*>(table + (c0 + wave) * w.J + 4 * cg) = sv;
        }
        __syncthreads();
    }
}

// out[n][J] = decoder_block_wide(y[n][0], y[n][1]) -- the search kernels' routine itself, never the table (the table's test)
__global__ __launch_bounds__(GT) void k_decoder_rows_wide(DecJoinW w, const long long* __restrict__ y, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* scratch = sm;                       // >= 8 max(J, DD)
    float* h = sm + 8 * max(w.J, w.DD);        // [3 DD]
    float* o = h + 3 * w.DD;                   // [J]
    decoder_block_wide(w, y[2 * blockIdx.x], y[2 * blockIdx.x + 1], h, scratch, o);
    for (int k = threadIdx.x; k < w.J; k += GT) out[(long long)blockIdx.x * w.J + k] = o[k];
}

// ---- batched rounds (greedy_rounds) ------------------------------------------------------------------------------
struct RoundsState {
    const float* enc;         // [B, Tp, J]
    int B, Tp, S;
    const int* t0;            // device scalar (offline batch quirk) or nullptr
    int skip1;
    const long long* init_ctx;  // [B][2] (online) or nullptr
    long long* tokens;
    int* timestamps;
    int* n_tokens;
    int max_tokens;
    int* overflow;
    int* next;                // [B] next frame to decide
    int* nemit;               // [B] symbols emitted so far
    long long* ctx;           // [B][2] context the stream's dec row was computed with
    int* win;                 // [B] frames of the current window
    float* dec;               // [B][J]
    float* act;               // [B*S][J] joiner input of the current windows
    const float* logits;      // [B*S][ldl]
    int ldl;
    int* active;              // [rounds + 1]: streams still running at the start of each round
};

// r < 0: set up every stream and its first window; r >= 0: decide round r's windows and set up the next ones
__global__ __launch_bounds__(GT) void k_rounds_step(DecJoinW w, RoundsState a, int r) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* h = sm;                                   // [3*DD]
    float* scratch = h + 3 * w.DD;                   // [8*max(J, DD)]
    float* decl = scratch + 8 * max(w.J, w.DD);      // [J]
    int* toks = reinterpret_cast<int*>(decl + w.J);  // [S]
    __shared__ int s_t, s_n, s_changed;
    __shared__ long long s_y0, s_y1;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = blockIdx.x;
    const int t0 = a.t0 ? *a.t0 : INT_MAX;
    int t, n;
    long long y0, y1;
    bool have_dec;  // dec[b] holds decoder(ctx[b])
    if (r < 0) {
        t = 0; n = 0;
        have_dec = false;
        if (a.init_ctx) { y0 = a.init_ctx[2 * b]; y1 = a.init_ctx[2 * b + 1]; }
        else { y0 = -1; y1 = K2HIP_BLANK_ID; }
    } else {
        t = a.next[b];
        if (t >= a.Tp) return;  // finished in an earlier round (uniform for the workgroup)
        n = a.nemit[b];
        y0 = a.ctx[2 * b];
        y1 = a.ctx[2 * b + 1];
        have_dec = true;
        const int wl = a.win[b];
        // argmax of each window row, one wave per row (S <= 8 = waves): later index wins ties, as the reference's loop
        for (int s = wave; s < wl; s += GT / 64) {
            const float* row = a.logits + (long long)(b * a.S + s) * a.ldl;
            float bv = -INFINITY;
            int bi = -1, pnan = -1;   // pnan: the largest NaN index this lane saw (amax_row_fix applies the reference's NaN rule)
            if ((a.ldl & 3) == 0) {   // 16-byte loads, 4 rows' worth of them independent (the merge is a total order: any sequence)
                for (int j4 = 4 * lane; j4 < w.V; j4 += 256) {
                    const float4 t4 = *reinterpret_cast<const float4*>(row + j4);
                    amax_merge(bv, bi, t4.x, j4);
                    if (j4 + 1 < w.V) amax_merge(bv, bi, t4.y, j4 + 1);
                    if (j4 + 2 < w.V) amax_merge(bv, bi, t4.z, j4 + 2);
                    if (j4 + 3 < w.V) amax_merge(bv, bi, t4.w, j4 + 3);
                    pnan = (t4.x != t4.x) ? j4 : pnan;
                    pnan = (j4 + 1 < w.V && t4.y != t4.y) ? j4 + 1 : pnan;
                    pnan = (j4 + 2 < w.V && t4.z != t4.z) ? j4 + 2 : pnan;
                    pnan = (j4 + 3 < w.V && t4.w != t4.w) ? j4 + 3 : pnan;
                }
            } else {
                for (int j = lane; j < w.V; j += 64) {
                    const float x = row[j];
                    amax_merge(bv, bi, x, j);
                    pnan = (x != x) ? j : pnan;
                }
            }
            amax_wave(bv, bi);
            bi = amax_row_fix(row, w.V, lane, bi, pnan);
            if (lane == 0) toks[s] = bi;
        }
        __syncthreads();
        if (tid == 0) {
            int adv = wl, changed = 0;
            for (int s = 0; s < wl; s++) {
                const int y = toks[s];
                if (y != K2HIP_BLANK_ID && y != K2HIP_UNK_ID && !(a.skip1 && y == 1)) {
                    if (n < a.max_tokens) {
                        a.tokens[(long long)b * a.max_tokens + n] = y;
                        a.timestamps[(long long)b * a.max_tokens + n] = t + s;
                    } else {
                        *a.overflow = 1;
                    }
                    n++;
                    y0 = y1;
                    y1 = y;
                    adv = s + 1;
                    changed = 1;
                    break;
                }
            }
            s_t = t + adv; s_n = n; s_y0 = y0; s_y1 = y1; s_changed = changed;
        }
        __syncthreads();
        t = s_t; n = s_n; y0 = s_y0; y1 = s_y1;
        if (s_changed) have_dec = false;
    }
    // a stream of the offline batch that has not emitted yet decodes frames <= t0 under [-1, blank] and frames > t0 under
    // [blank, blank] (the batch's first emission re-runs the decoder for every stream on its seeded blanks, OfflineRecognizer.cs
    // :250-258,278-286); its windows never straddle t0
    const bool own = n > 0 || a.init_ctx != nullptr;
    if (!own && t < a.Tp) {
        const long long w0 = t > t0 ? K2HIP_BLANK_ID : -1;
        if (w0 != y0) { y0 = w0; y1 = K2HIP_BLANK_ID; have_dec = false; }
    }
    if (tid == 0) {
        a.next[b] = t;
        a.nemit[b] = n;
        a.n_tokens[b] = n < a.max_tokens ? n : a.max_tokens;
        a.ctx[2 * b] = y0;
        a.ctx[2 * b + 1] = y1;
    }
    if (t >= a.Tp) return;
    int wl = min(a.S, a.Tp - t);
    // (t0 == INT_MAX: no scalar, or no stream of the batch ever emits; t0 - t + 1 would overflow)
    if (!own && t0 != INT_MAX && t <= t0) wl = min(wl, t0 - t + 1);
    if (tid == 0) {
        a.win[b] = wl;
        atomicAdd(&a.active[r + 2], 1);  // r = -1 fills active[1]... see greedy_rounds: slot k+1 counts the streams alive in round k
    }
    if (!have_dec) {
        decoder_context(w, y0, y1, h, scratch, decl);
        for (int k = tid; k < w.J; k += GT) a.dec[(long long)b * w.J + k] = decl[k];
    } else {
        for (int k = tid; k < w.J; k += GT) decl[k] = a.dec[(long long)b * w.J + k];
        __syncthreads();
    }
    const float* enc = a.enc + (long long)b * a.Tp * w.J;
    for (int idx = tid; idx < a.S * (w.J >> 2); idx += GT) {
        const int s = idx / (w.J >> 2), k = (idx - s * (w.J >> 2)) * 4;
        const int row = min(t + s, a.Tp - 1);  // rows past the window are computed on a valid frame and ignored
        const float4 e = *reinterpret_cast<const float4*>(enc + (long long)row * w.J + k);
        *reinterpret_cast<float4*>(a.act + (long long)(b * a.S + s) * w.J + k) =
            make_float4(tanhf(e.x + decl[k]), tanhf(e.y + decl[k + 1]), tanhf(e.z + decl[k + 2]), tanhf(e.w + decl[k + 3]));
    }
}

}  // namespace

void greedy_rounds(const Ctx& ctx, const DecJoinW& w, const float* out_w, const GreedyArgs& a0) {
    if (a0.B <= 0) return;
    const int B = a0.B, Tp = a0.Tp, S = std::min(GF, Tp), R = Tp;  // a round advances every running stream by >= 1 frame
    Arena& ar = *ctx.arena;
    RoundsState st;
    st.enc = a0.enc; st.B = B; st.Tp = Tp; st.S = S; st.t0 = a0.t0; st.skip1 = a0.skip1; st.init_ctx = a0.init_ctx;
    st.tokens = a0.tokens; st.timestamps = a0.timestamps; st.n_tokens = a0.n_tokens; st.max_tokens = a0.max_tokens; st.overflow = a0.overflow;
    st.next = ar.take<int>(B);
    st.nemit = ar.take<int>(B);
    st.ctx = ar.take<long long>(2 * B);
    st.win = ar.take<int>(B);
    st.dec = ar.take<float>((int64_t)B * w.J);
    st.act = ar.take<float>((int64_t)B * S * w.J);
    st.ldl = w.Vp;
    float* logits = ar.take<float>((int64_t)B * S * st.ldl);
    st.logits = logits;
    st.active = ar.take<int>(R + 2);
    const size_t lds = sizeof(float) * (3 * (size_t)w.DD + 8 * (size_t)std::max(w.J, w.DD) + w.J + 16);
    if (!ctx.dry) {
        K2_REQUIRE(w.J % 4 == 0 && w.DD % 4 == 0 && S <= GT / 64, "greedy_rounds: joiner %d / decoder %d widths must be multiples of 4", w.J, w.DD);
        K2_HIP(hipMemsetAsync(st.active, 0, sizeof(int) * (R + 2), ctx.stream));
        static LdsAttrOnce lds_attr;
        lds_attr.ensure(k_rounds_step, (int)lds);
        hipLaunchKernelGGL(k_rounds_step, dim3(B), dim3(GT), lds, ctx.stream, w, st, -1);  // counts the streams of round 0 into active[1]
        K2_HIP(hipGetLastError());
    }
    ctx.add_flops(0.0, 2.0 * B * (double)Tp * w.J * w.V, 0);  // one joiner evaluation per frame, as the frame-by-frame loop
    Ctx q = ctx;  // the rounds' GEMMs are not part of the encoder-GEMM statistics (most of them return at once)
    q.stats = nullptr;
    q.instrument = false;
    q.gemm_log = nullptr;
    for (int r = 0; r < R; r++) {
        GemmArgs g;
        g.A = st.act; g.lda = w.J; g.W = out_w; g.ldw = w.J; g.bias = w.out_b; g.C = logits; g.ldc = st.ldl;
        g.M = B * S; g.N = w.V; g.K = w.J;
        g.skip_if_zero = st.active + r + 1;
        gemm(q, g);
        if (!ctx.dry) {
            hipLaunchKernelGGL(k_rounds_step, dim3(B), dim3(GT), lds, ctx.stream, w, st, r);
            K2_HIP(hipGetLastError());
        }
    }
}

void decoder_table(const Ctx& ctx, const DecJoinW& w, float* table) {
    if (ctx.dry) return;
    K2_REQUIRE(w.J % 4 == 0 && w.DD % 4 == 0, "decoder table: joiner %d / decoder %d widths must be multiples of 4", w.J, w.DD);
    const long long n_ctx = ((long long)w.V + 1) * w.V;
    const size_t lds = sizeof(float) * ((size_t)w.DD * GF + 8 * APAD + kPsumFloats + 3 * (size_t)w.DD + 8 * (size_t)std::max(w.J, w.DD));
    K2_REQUIRE(lds <= 150 * 1024, "decoder table: joiner %d / decoder %d need %zu B of LDS", w.J, w.DD, lds);
    static LdsAttrOnce lds_attr;
    lds_attr.ensure(k_decoder_table, 150 * 1024);
    DecJoinW wn = w;
    wn.dec_table = nullptr;
    hipLaunchKernelGGL(k_decoder_table, dim3((unsigned)((n_ctx + GF - 1) / GF)), dim3(GT), lds, ctx.stream, wn, table, n_ctx);
    K2_HIP(hipGetLastError());
}
void decoder_rows_wide(const Ctx& ctx, const DecJoinW& w, const long long* y, int N, float* out) {
    if (ctx.dry || N <= 0) return;
    const size_t lds = sizeof(float) * (8 * (size_t)std::max(w.J, w.DD) + 3 * (size_t)w.DD + w.J);
    static LdsAttrOnce lds_attr;
    lds_attr.ensure(k_decoder_rows_wide, 150 * 1024);
    hipLaunchKernelGGL(k_decoder_rows_wide, dim3(N), dim3(GT), lds, ctx.stream, w, y, out);
    K2_HIP(hipGetLastError());
}
void decoder_start_contexts(const Ctx& ctx, const DecJoinW& w, float* out) {
    if (ctx.dry) return;
    const size_t lds = sizeof(float) * (8 * (size_t)std::max(w.J, w.DD) + 3 * (size_t)w.DD + w.J);
    hipLaunchKernelGGL(k_decoder_start, dim3(2), dim3(GT), lds, ctx.stream, w, out);
    K2_HIP(hipGetLastError());
}
void decoder(const Ctx& ctx, const DecJoinW& w, const long long* y, int N, float* dec_out) {
    if (ctx.dry || N <= 0) return;
    hipLaunchKernelGGL(k_decoder, dim3(N), dim3(256), sizeof(float) * 3 * w.DD, ctx.stream, w, y, dec_out);
    K2_HIP(hipGetLastError());
}
void tanh_add(const Ctx& ctx, const float* enc, const float* dec, int dec_stride, float* y, int N, int J) {
    if (ctx.dry || N <= 0) return;
    long long n4 = (long long)N * J / 4;
    hipLaunchKernelGGL(k_tanh_add, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, ctx.stream, enc, dec, dec_stride, y, n4, J / 4);
    K2_HIP(hipGetLastError());
}
void argmax_rows(const Ctx& ctx, const float* logits, int ld, int N, int V, int* tok) {
    if (ctx.dry || N <= 0) return;
    hipLaunchKernelGGL(k_argmax_rows<false>, dim3(cdiv(N, 4)), dim3(256), 0, ctx.stream, logits, ld, N, V, tok);
    K2_HIP(hipGetLastError());
}
void argmax_first_rows(const Ctx& ctx, const float* logits, int ld, int N, int V, int* tok) {
    if (ctx.dry || N <= 0) return;
    hipLaunchKernelGGL(k_argmax_rows<true>, dim3(cdiv(N, 4)), dim3(256), 0, ctx.stream, logits, ld, N, V, tok);
    K2_HIP(hipGetLastError());
}
void log_softmax_rows(const Ctx& ctx, float* x, int M, int V) {
    if (ctx.dry || M <= 0) return;
    hipLaunchKernelGGL(k_log_softmax_rows, dim3(cdiv(M, 4)), dim3(256), 0, ctx.stream, x, M, V);
    K2_HIP(hipGetLastError());
}
void ctc_collapse(const Ctx& ctx, const int* tok, int B, int Tp, const int* frame_off, long long* tokens, int* timestamps,
                  int* n_tokens, int max_tokens, int* trail, int* any, int* overflow) {
    if (ctx.dry || B <= 0) return;
    hipLaunchKernelGGL(k_ctc_collapse, dim3(cdiv(B, 64)), dim3(64), 0, ctx.stream, tok, B, Tp, frame_off, tokens, timestamps, n_tokens,
                       max_tokens, trail, any, overflow);
    K2_HIP(hipGetLastError());
}
void first_emit_frame(const Ctx& ctx, const int* tok, int B, int Tp, int skip1, int* t0) {
    if (ctx.dry) return;
    hipLaunchKernelGGL(k_first_emit, dim3(1), dim3(1024), 0, ctx.stream, tok, B, Tp, skip1, t0);
    K2_HIP(hipGetLastError());
}
static size_t greedy_lds_bytes(const DecJoinW& w) {
    return sizeof(float) * ((size_t)w.J * GF + 8 * APAD + kPsumFloats + 3 * (size_t)w.J + 3 * (size_t)w.DD + GF + 8 + 2 * kMaxParts * GF + 4) +
           (w.out_h16 ? (size_t)w.J * 32 : 0);   // + actH
}

// column slabs per stream of the persistent search (see greedy_loop)
static int greedy_parts(const DecJoinW& w, int B, bool streaming, bool one_part) {
    if (one_part) return 1;
    const int ncg = w.Vp >> 2, pass = (GT / 64) * 8;
    const int bc = std::max(B, 1);
    const int budget = streaming ? device_cu_count() : std::max(device_cu_count() / 4, 1);
    int parts = std::min({(ncg + pass - 1) / pass, kMaxParts, budget / bc});
    if (tunables().greedy_parts > 0) parts = std::max(1, std::min({tunables().greedy_parts, kMaxParts, budget / bc}));  // tuning only
    if (parts < 2 || tunables().greedy_one_part) parts = 1;
    return parts;
}
// would greedy_loop's rounds go through the f16 screen for a batch of B streams?  (The streaming tick picks its search form by this:
// with the screen the persistent search beats the rounds of joiner GEMMs -- zipformer2-streaming-zh, V = 2000, 128 streams: 4.18
// against 4.31 ms per tick; without it -- conformer-streaming-zh, V = 5537: two slabs of 2770 columns per stream do not fit the
// screen's LDS area -- the rounds win, 7.20 against 7.99 ms.)
bool greedy_loop_screens(const DecJoinW& w, int B, bool streaming, bool one_part) {
    if (!w.out_h16 || (w.J & 31) != 0) return false;
    const int nks = w.J >> 5;
    if (!(nks == 16 || nks == 8 || nks == 4 || nks == 2)) return false;
    const int parts = greedy_parts(w, B, streaming, one_part), ncg = w.Vp >> 2;
    const int cper = ((ncg + parts - 1) / parts + 7) & ~7;
    return 4 * cper <= kScreenMaxCols;
}

void greedy_loop(const Ctx& ctx, const DecJoinW& w, const GreedyArgs& a0) {
    if (a0.B <= 0) return;
    GreedyArgs a = a0;
    // parts per stream: enough that a part's slab is about one pass of the workgroup's (GT / 64) * 8 column groups; all B*parts
    // workgroups must be co-resident (they wait for each other), and offline the search runs under the next batch's encoder, which
    // gets no workgroup onto a CU that hosts one: a quarter of the chip.  Measured, pipelined step / search alone: conformer-zh
    // (V = 5537, B = 8) 4 parts 25.7 / 22.6 ms, 6: 21.3 / 18.9, 8: 16.7 / 13.9, 10: 18.1 / 14.2, 12: 17.1 / 11.7, 16: 17.05 / 10.0;
    // zipformer2-large-en (V = 500, B = 32) 1 part 15.13 / 5.45, 2: 14.44 / 3.79, 3: 14.57 / 3.67, 4: 14.56 / 3.35.
    // co-residency budget: a quarter of the chip's CUs offline, the whole chip for a streaming chunk step (nothing else is in
    // flight; one 150 KB workgroup per CU).  From the device's own CU count (256 on MI355X -> 64 / 256); a GPU shared with other
    // processes can still leave the parts of a stream apart: then the bounded waits time out and the engine repeats the search with
    // one part per stream (greedy_relaunch_one_part).
    const int parts = greedy_parts(w, a.B, a.init_ctx != nullptr, ctx.one_part);
    a.parts = parts;
    // (one block, one memset: the two exchange areas were two fills of ~5 us each in front of the search)
    const size_t gran_words = (size_t)a.B * 2 * parts * GF * 2, gran2_words = (size_t)a.B * 2 * w.J;
    a.gran = parts > 1 ? ctx.arena->take<unsigned long long>((int64_t)(gran_words + gran2_words)) : nullptr;
    a.gran2 = parts > 1 ? a.gran + gran_words : nullptr;
    // the launch is noted for the one-part repeat BEFORE the dry return (the sizing pass writes null pointers here and is always
    // followed by a pass with real ones)
    if (ctx.greedy_rec) {
        ctx.greedy_rec->valid = parts > 1;
        ctx.greedy_rec->beam = false;
        ctx.greedy_rec->w = w;
        ctx.greedy_rec->a = a;
    }
    if (ctx.dry) return;
    K2_REQUIRE(w.J % 8 == 0, "greedy: joiner_dim %d must be a multiple of 8", w.J);
    K2_REQUIRE(w.DD % 4 == 0 && 8 * w.DD <= w.J * GF, "greedy: decoder_dim %d too wide for the LDS scratch", w.DD);
    const size_t lds = greedy_lds_bytes(w);
    K2_REQUIRE(lds <= 150 * 1024, "greedy: vocab %d / joiner %d need %zu B of LDS", w.V, w.J, lds);
    static LdsAttrOnce lds_attr;
    static LdsAttrOnce lds_attr_s;
    lds_attr.ensure(k_greedy<false>, 150 * 1024);
    lds_attr_s.ensure(k_greedy<true>, 150 * 1024);
    if (parts > 1) {
        K2_HIP(hipMemsetAsync(a.gran, 0, sizeof(unsigned long long) * (gran_words + gran2_words), ctx.stream));
    }
    static unsigned long long* d_stamps = nullptr;
    if (tunables().greedy_stamps) {
        if (!d_stamps) K2_HIP(hipMalloc(&d_stamps, 8 * sizeof(unsigned long long)));
        K2_HIP(hipMemsetAsync(d_stamps, 0, 8 * sizeof(unsigned long long), ctx.stream));
        a.stamps = d_stamps;
    }
    if (w.out_h16) hipLaunchKernelGGL(k_greedy<true>, dim3(a.B * parts), dim3(GT), lds, ctx.stream, w, a);
    else hipLaunchKernelGGL(k_greedy<false>, dim3(a.B * parts), dim3(GT), lds, ctx.stream, w, a);
    if (a.stamps) {   // tuning: synchronous report of workgroup 0's round
        unsigned long long h[8];
        K2_HIP(hipStreamSynchronize(ctx.stream));
        K2_HIP(copy_blocking(h, d_stamps, sizeof h, hipMemcpyDeviceToHost));
        const double r = h[0] ? (double)h[0] : 1.0;
        fprintf(stderr, "[k_greedy stamps] %llu rounds, %d parts; us per round: activations %.2f, screen tiles %.2f, candidate scan %.2f, re-check %.2f, "
                        "sweep passes %.2f, publish + exchange %.2f, decision + decoder %.2f\n", h[0], parts, h[1] / r / 100.0, h[2] / r / 100.0, h[3] / r / 100.0,
                h[4] / r / 100.0, h[5] / r / 100.0, h[6] / r / 100.0, h[7] / r / 100.0);
        a.stamps = nullptr;
    }
    K2_HIP(hipGetLastError());
    if (parts > 1 && tunables().test_greedy_timeout) K2_HIP(hipMemsetD32Async((hipDeviceptr_t)a.overflow, 2, 1, ctx.stream));
}

void greedy_relaunch_one_part(hipStream_t stream, const GreedyLaunch& rec) {
    if (rec.beam) {   // the one-kernel beam search with two slabs per stream: once more with one
        beam_relaunch_one_slab(stream, rec);
        return;
    }
    K2_REQUIRE(rec.valid, "greedy retry: no repeatable launch on record");
    GreedyArgs a = rec.a;
    a.parts = 1;
    a.gran = nullptr;
    a.gran2 = nullptr;
    const DecJoinW& w = rec.w;
    const size_t lds = greedy_lds_bytes(w);
    K2_HIP(hipMemsetAsync(a.overflow, 0, sizeof(int), stream));
    if (tunables().test_greedy_timeout) {
        // the test hook raises the timeout flag behind a search that FINISHED: wipe what it wrote, so that the tokens the caller gets
        // can only be the repeat's (a repeat that wrote nothing would otherwise pass on the first launch's output)
        K2_HIP(hipMemsetAsync(a.tokens, 0xEE, sizeof(long long) * (size_t)a.B * a.max_tokens, stream));
        K2_HIP(hipMemsetAsync(a.timestamps, 0xEE, sizeof(int) * (size_t)a.B * a.max_tokens, stream));
        K2_HIP(hipMemsetAsync(a.n_tokens, 0xEE, sizeof(int) * (size_t)a.B, stream));
    }
    if (w.out_h16) hipLaunchKernelGGL(k_greedy<true>, dim3(a.B), dim3(GT), lds, stream, w, a);
    else hipLaunchKernelGGL(k_greedy<false>, dim3(a.B), dim3(GT), lds, stream, w, a);
    K2_HIP(hipGetLastError());
}

}  // namespace k2hip
