// Modified beam search on device (BASELINE.json configs[2]: "modified-beam-search beam=4").
//
// The reference has no beam search (OfflineRecognizer.cs:54-68 only dispatches "greedy_search"); the
// semantics are icefall's beam_search.py modified_beam_search, restated and tie-broken in
// DESIGN.md.  Frame-synchronous, so the whole search is enqueued up front (5 launches
// per frame, no host round trip): for frame t
//   decoder(ctx of every hypothesis)              k_decoder           [B*K rows]
//   act = tanh(enc[b,t] + dec[b,k])               k_beam_act
//   logits = act . W^T + bias                     gemm_f32_mfma       [B*K, V]
//   per stream: log_softmax + hyp score, top-K over K*V by (score desc, flat index asc), expand,
//   merge equal token sequences by logaddexp (first-inserted hypothesis keeps its timestamps)   k_beam_step
// Hypotheses live in double-buffered device arrays [2][B][K][cap]; ys excludes the ctx-blank prefix.
#include "kernels.h"

namespace k2hip {
namespace {

constexpr int BT = 256;  // threads per stream workgroup

__global__ void k_beam_init(BeamState s, int B) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * s.K) return;
    int k = i % s.K;
    s.lp[i] = k == 0 ? 0.f : -INFINITY;
    s.n[i] = 0;
    s.n[B * s.K + i] = 0;
    s.ctx[2 * i] = K2HIP_BLANK_ID;
    s.ctx[2 * i + 1] = K2HIP_BLANK_ID;
    if (k == 0) s.nhyp[i / s.K] = 1;
}

__global__ void k_beam_act(const float* __restrict__ enc, const float* __restrict__ dec, float* __restrict__ act, int B, int K,
                           int Tp, int t, int J4) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)B * K * J4) return;
    int q = (int)(i % J4);
    int m = (int)(i / J4), b = m / K;
    float4 e = *reinterpret_cast<const float4*>(enc + ((long long)b * Tp + t) * (J4 * 4) + 4 * q);
    float4 d = reinterpret_cast<const float4*>(dec)[i];
    reinterpret_cast<float4*>(act)[i] = make_float4(tanhf(e.x + d.x), tanhf(e.y + d.y), tanhf(e.z + d.z), tanhf(e.w + d.w));
}

__device__ __forceinline__ float block_max(float v, float* red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = red[0];
    for (int w = 1; w < BT / 64; w++) r = fmaxf(r, red[w]);
    __syncthreads();
    return r;
}
__device__ __forceinline__ float block_sum(float v, float* red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = red[0];
    for (int w = 1; w < BT / 64; w++) r += red[w];
    __syncthreads();
    return r;
}

// one workgroup per stream; `cur` = buffer holding frame t's input hypotheses
__global__ __launch_bounds__(BT) void k_beam_step(BeamState s, const float* __restrict__ logits, int V, int t, int cur, int B) {
    __shared__ float red[BT / 64];
    __shared__ float mxs[kMaxBeam], lses[kMaxBeam];  // per hypothesis: max logit, log(sum(exp(l - max)))
    __shared__ int taken[kMaxBeam];
    __shared__ float topv[kMaxBeam];
    __shared__ int topi[kMaxBeam];
    __shared__ float rv[BT / 64];
    __shared__ int ri[BT / 64];
    __shared__ int dupof[kMaxBeam];           // for new entry r: index of the earlier entry it merges into, or -1
    __shared__ int same;
    const int b = blockIdx.x, tid = threadIdx.x, K = s.K;
    const int nA = s.nhyp[b];
    const float* lg = logits + (long long)b * K * V;
    // ---- log_softmax offsets per hypothesis
    for (int k = 0; k < nA; k++) {
        float mx = -INFINITY;
        for (int v = tid; v < V; v += BT) mx = fmaxf(mx, lg[(long long)k * V + v]);
        mx = block_max(mx, red);
        float sm = 0.f;
        for (int v = tid; v < V; v += BT) sm += expf(lg[(long long)k * V + v] - mx);
        sm = block_sum(sm, red);
        if (tid == 0) { mxs[k] = mx; lses[k] = logf(sm); }
        __syncthreads();
    }
    // ---- top `want` of nA*V by (score desc, flat index asc)
    const int nc = nA * V, want = min(K, nc);
    for (int r = 0; r < want; r++) {
        float bv = -INFINITY;
        int bi = -1;
        for (int i = tid; i < nc; i += BT) {
            bool tk = false;
            for (int q = 0; q < r; q++) tk |= (taken[q] == i);
            if (tk) continue;
            const int k = i / V;
            const float sc = (lg[i] - mxs[k] - lses[k]) + s.lp[b * K + k];  // the oracle's order of operations
            if (bi < 0 || sc > bv) { bv = sc; bi = i; }   // ascending i per thread: first maximum wins
        }
        // block argmax: value desc, then flat index asc (bi < 0 loses)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            float ov = __shfl_xor(bv, o);
            int oi = __shfl_xor(bi, o);
            if (oi >= 0 && (bi < 0 || ov > bv || (ov == bv && oi < bi))) { bv = ov; bi = oi; }
        }
        if ((tid & 63) == 0) { rv[tid >> 6] = bv; ri[tid >> 6] = bi; }
        __syncthreads();
        if (tid == 0) {
            float v = rv[0];
            int i = ri[0];
            for (int w = 1; w < BT / 64; w++)
                if (ri[w] >= 0 && (i < 0 || rv[w] > v || (rv[w] == v && ri[w] < i))) { v = rv[w]; i = ri[w]; }
            taken[r] = i;
            topv[r] = v;
            topi[r] = i;
        }
        __syncthreads();
    }
    // ---- expand + merge (HypothesisList.add) into the other buffer
    const int nxt = cur ^ 1;
    const long long BK = (long long)B * K;
    const int* ys_c = s.ys + ((long long)cur * BK + (long long)b * K) * s.cap;
    const int* ts_c = s.ts + ((long long)cur * BK + (long long)b * K) * s.cap;
    int* ys_n = s.ys + ((long long)nxt * BK + (long long)b * K) * s.cap;
    int* ts_n = s.ts + ((long long)nxt * BK + (long long)b * K) * s.cap;
    const int* n_c = s.n + cur * BK + b * K;
    int* n_n = s.n + nxt * BK + b * K;
    // sequence of candidate r = ys_c[parent_r][0..n) (+ tok_r if real)
    for (int r = 0; r < want; r++) {
        const int hr = topi[r] / V, tr = topi[r] % V;
        const bool realr = tr != K2HIP_BLANK_ID && tr != K2HIP_UNK_ID;
        const int lenr = n_c[hr] + (realr ? 1 : 0);
        int found = -1;
        for (int q = 0; q < r && found < 0; q++) {
            if (dupof[q] >= 0) continue;  // q itself was merged away; its target is tested on its own
            const int hq = topi[q] / V, tq = topi[q] % V;
            const bool realq = tq != K2HIP_BLANK_ID && tq != K2HIP_UNK_ID;
            const int lenq = n_c[hq] + (realq ? 1 : 0);
            if (lenq != lenr) continue;
            if (tid == 0) same = 1;
            __syncthreads();
            for (int i = tid; i < lenr; i += BT) {
                const int a = (i < n_c[hr]) ? ys_c[(long long)hr * s.cap + i] : tr;
                const int c = (i < n_c[hq]) ? ys_c[(long long)hq * s.cap + i] : tq;
                if (a != c) same = 0;
            }
            __syncthreads();
            if (same) found = q;
            __syncthreads();
        }
        if (tid == 0) dupof[r] = found;
        __syncthreads();
    }
    if (tid == 0) {
        // slot assignment in insertion order; merged scores accumulate in candidate order (logaddexp)
        int slot_of[kMaxBeam];
        int nN = 0;
        for (int r = 0; r < want; r++) {
            if (dupof[r] < 0) {
                slot_of[r] = nN++;
                s.lp_next[b * K + slot_of[r]] = topv[r];
            } else {
                slot_of[r] = slot_of[dupof[r]];
                const float a = s.lp_next[b * K + slot_of[r]], c = topv[r];
                const float mx = fmaxf(a, c);
                s.lp_next[b * K + slot_of[r]] = (isinf(mx) && mx < 0) ? mx : mx + log1pf(expf(-fabsf(a - c)));
            }
            taken[r] = dupof[r] < 0 ? slot_of[r] : -1;  // reuse: destination slot of a fresh hypothesis
        }
        for (int k = nN; k < K; k++) {
            s.lp_next[b * K + k] = -INFINITY;
            // empty slots still go through the batched decoder launch: give them a valid context
            s.ctx_next[2 * (b * K + k)] = K2HIP_BLANK_ID;
            s.ctx_next[2 * (b * K + k) + 1] = K2HIP_BLANK_ID;
            n_n[k] = 0;
        }
        s.nhyp_next[b] = nN;
    }
    __syncthreads();
    for (int r = 0; r < want; r++) {
        const int slot = taken[r];
        if (slot < 0) continue;
        const int hr = topi[r] / V, tr = topi[r] % V;
        const bool realr = tr != K2HIP_BLANK_ID && tr != K2HIP_UNK_ID;
        const int n0 = n_c[hr];
        for (int i = tid; i < n0; i += BT) {
            ys_n[(long long)slot * s.cap + i] = ys_c[(long long)hr * s.cap + i];
            ts_n[(long long)slot * s.cap + i] = ts_c[(long long)hr * s.cap + i];
        }
        if (tid == 0) {
            int nn = n0;
            if (realr) {
                if (nn < s.cap) {
                    ys_n[(long long)slot * s.cap + nn] = tr;
                    ts_n[(long long)slot * s.cap + nn] = t;
                }
                nn++;
            }
            n_n[slot] = nn;
            // decoder context of the new hypothesis: last two of [blank, blank] + ys
            long long y1 = nn >= 1 ? (realr ? tr : ys_c[(long long)hr * s.cap + n0 - 1]) : K2HIP_BLANK_ID;
            long long y0 = K2HIP_BLANK_ID;
            if (nn >= 2) y0 = realr ? ys_c[(long long)hr * s.cap + n0 - 1] : ys_c[(long long)hr * s.cap + n0 - 2];
            s.ctx_next[2 * (b * K + slot)] = y0;
            s.ctx_next[2 * (b * K + slot) + 1] = y1;
        }
    }
}

// lp/ctx/nhyp are single-buffered inputs of the next frame's decoder + step: commit after the step
__global__ void k_beam_commit(BeamState s, int B) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < B * s.K) {
        s.lp[i] = s.lp_next[i];
        s.ctx[2 * i] = s.ctx_next[2 * i];
        s.ctx[2 * i + 1] = s.ctx_next[2 * i + 1];
    }
    if (i < B) s.nhyp[i] = s.nhyp_next[i];
}

// get_most_probable(length_norm=True): max of log_prob / len(ys) (len counts the 2 ctx blanks), first maximum
__global__ void k_beam_final(BeamState s, int fin, int B, long long* __restrict__ tokens, int* __restrict__ timestamps,
                             int* __restrict__ n_tokens, float* __restrict__ scores, int max_tokens, int* __restrict__ overflow) {
    const int b = blockIdx.x, K = s.K;
    const long long BK = (long long)B * K;
    const int* n_f = s.n + fin * BK + b * K;
    int best = 0;
    float bs = s.lp[b * K] / (float)(n_f[0] + 2);
    for (int k = 1; k < s.nhyp[b]; k++) {
        float v = s.lp[b * K + k] / (float)(n_f[k] + 2);
        if (v > bs) { bs = v; best = k; }
    }
    const int n = n_f[best];
    if (n > max_tokens || n > s.cap) {
        if (threadIdx.x == 0) *overflow = 1;
        return;
    }
    const int* ys = s.ys + (fin * BK + (long long)b * K + best) * s.cap;
    const int* ts = s.ts + (fin * BK + (long long)b * K + best) * s.cap;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        tokens[(long long)b * max_tokens + i] = ys[i];
        timestamps[(long long)b * max_tokens + i] = ts[i];
    }
    if (threadIdx.x == 0) {
        n_tokens[b] = n;
        if (scores) scores[b] = s.lp[b * K + best];
    }
}

}  // namespace

void beam_search(const Ctx& ctx, const DecJoinW& w, const BeamArgs& a) {
    K2_REQUIRE(a.beam >= 1 && a.beam <= kMaxBeam, "beam search: beam %d out of range [1,%d]", a.beam, kMaxBeam);
    K2_REQUIRE(a.B > 0 && a.Tp > 0, "beam search: bad shape");
    Arena& ar = *ctx.arena;
    const int B = a.B, K = a.beam, M = B * K, cap = a.Tp + 1;
    BeamState s;
    s.K = K;
    s.cap = cap;
    s.ys = ar.take<int>((int64_t)2 * M * cap);
    s.ts = ar.take<int>((int64_t)2 * M * cap);
    s.n = ar.take<int>((int64_t)2 * M);
    s.lp = ar.take<float>(M);
    s.lp_next = ar.take<float>(M);
    s.ctx = ar.take<long long>((int64_t)2 * M);
    s.ctx_next = ar.take<long long>((int64_t)2 * M);
    s.nhyp = ar.take<int>(B);
    s.nhyp_next = ar.take<int>(B);
    float* dec = ar.take<float>((int64_t)M * w.J);
    float* act = ar.take<float>((int64_t)M * w.J);
    float* logits = ar.take<float>((int64_t)M * w.V);
    if (!ctx.dry) {
        K2_HIP(hipMemsetAsync(a.overflow, 0, sizeof(int), ctx.stream));
        hipLaunchKernelGGL(k_beam_init, dim3(cdiv(M, 256)), dim3(256), 0, ctx.stream, s, B);
        K2_HIP(hipGetLastError());
    }
    for (int t = 0; t < a.Tp; t++) {
        decoder(ctx, w, s.ctx, M, dec);
        if (!ctx.dry) {
            long long n4 = (long long)M * w.J / 4;
            hipLaunchKernelGGL(k_beam_act, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, ctx.stream, a.enc, dec, act, B, K, a.Tp, t,
                               w.J / 4);
        }
        linear(ctx, act, w.J, a.out_w, w.out_b, logits, w.V, M, w.J, w.V);
        if (!ctx.dry) {
            hipLaunchKernelGGL(k_beam_step, dim3(B), dim3(BT), 0, ctx.stream, s, logits, w.V, t, t & 1, B);
            hipLaunchKernelGGL(k_beam_commit, dim3(cdiv(M, 256)), dim3(256), 0, ctx.stream, s, B);
            K2_HIP(hipGetLastError());
        }
    }
    if (!ctx.dry) {
        hipLaunchKernelGGL(k_beam_final, dim3(B), dim3(256), 0, ctx.stream, s, a.Tp & 1, B, a.tokens, a.timestamps, a.n_tokens, a.scores,
                           a.max_tokens, a.overflow);
        K2_HIP(hipGetLastError());
    }
}

}  // namespace k2hip
