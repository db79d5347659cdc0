// Modified beam search on device (BASELINE.json configs[2]: "modified-beam-search beam=4").
//
// The reference has no beam search (OfflineRecognizer.cs:54-68 only dispatches "greedy_search"); the
// semantics are icefall's beam_search.py modified_beam_search, restated and tie-broken in
// DESIGN.md.  Frame-synchronous, so the whole search is enqueued up front (4 launches
// per frame, no host round trip): for frame t
//   h = relu(conv(emb(ctx)))  of every hypothesis  k_beam_embconv      [B*K rows]
//   act = tanh(enc[b,t] + decoder_proj(h)[b,k])   one GEMM, the encoder frame added (row / K) before the tanh in its epilogue
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

// relu(Conv1d(k = 2) over (emb[y0], emb[y1]))[co]  (id < 0 -> zero embedding)
__device__ __forceinline__ float embconv1(const DecJoinW& w, long long y0, long long y1, int co) {
    float s = 0.f;
    if (w.cpg <= 4) {
        const int g0 = (co / w.cpg) * w.cpg;
        for (int ci = 0; ci < w.cpg; ci++) {
            const float e0 = y0 >= 0 ? w.emb[y0 * w.DD + g0 + ci] : 0.f;
            const float e1 = y1 >= 0 ? w.emb[y1 * w.DD + g0 + ci] : 0.f;
            s += w.conv[(co * w.cpg + ci) * 2 + 0] * e0;
            s += w.conv[(co * w.cpg + ci) * 2 + 1] * e1;
        }
    } else {
        s = (y0 >= 0 ? w.ptab[y0 * w.DD + co] : 0.f) + (y1 >= 0 ? w.ptab[((long long)w.V + y1) * w.DD + co] : 0.f);
    }
    return fmaxf(s, 0.f);
}
// h[m][co] for every hypothesis row m
__global__ void k_beam_embconv(DecJoinW w, const long long* __restrict__ y, float* __restrict__ h, int M) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)M * w.DD) return;
    const int m = (int)(i / w.DD), co = (int)(i - (long long)m * w.DD);
    h[i] = embconv1(w, y[2 * m], y[2 * m + 1], co);
}

// one workgroup per stream; `cur` = buffer holding frame t's input hypotheses
__global__ __launch_bounds__(BT) void k_beam_step(BeamState s, const float* __restrict__ logits, int V, int t, int cur, int B) {
    __shared__ int taken[kMaxBeam];
    __shared__ float topv[kMaxBeam];
    __shared__ int topi[kMaxBeam];
    __shared__ int dupof[kMaxBeam];           // for new entry r: index of the earlier entry it merges into, or -1
    __shared__ int same;
    const int b = blockIdx.x, tid = threadIdx.x, K = s.K;
    const int nA = s.nhyp[b];
    const float* lg = logits + (long long)b * K * V;
    // ---- one wave per hypothesis (4 waves, K <= 8): log_softmax statistics, then the hypothesis' own top `want` candidates by
    //      (score desc, token asc) with wave shuffles only; the global top `want` is the top of the union (no block-wide
    //      reductions, which is what made this step cost more than the joiner GEMM)
    const int nc = nA * V, want = min(K, nc);
    const int lane = tid & 63, wave = tid >> 6;
    __shared__ float candv[kMaxBeam * kMaxBeam];
    __shared__ int candi[kMaxBeam * kMaxBeam];
    if (V <= 64 * 16) {
        // the hypothesis' logits, V / 64 per lane, are read ONCE into registers: the max, the sum and the `want` selection rounds below
        // went through global memory 2 + want times before (a chain of dependent loads that made this step 20 us)
        for (int k = wave; k < nA; k += BT / 64) {
            const float* l = lg + (long long)k * V;
            float lv[16];
#pragma unroll
            for (int i = 0; i < 16; i++) lv[i] = lane + 64 * i < V ? l[lane + 64 * i] : -INFINITY;
            float mx = -INFINITY;
#pragma unroll
            for (int i = 0; i < 16; i++) mx = fmaxf(mx, lv[i]);
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
            float sm = 0.f;
#pragma unroll
            for (int i = 0; i < 16; i++)
                if (lane + 64 * i < V) sm += expf(lv[i] - mx);
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) sm += __shfl_xor(sm, o);
            const float lse = logf(sm), lpk = s.lp[b * K + k];
            unsigned used = 0;  // bit i: this lane's element i was already selected
            for (int r = 0; r < want; r++) {
                float bv = -INFINITY;
                int bi = -1;
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    const int v = lane + 64 * i;
                    if (v >= V || (used >> i & 1)) continue;
                    const float sc = (lv[i] - mx - lse) + lpk;  // the oracle's order of operations
                    if (bi < 0 || sc > bv) { bv = sc; bi = v; }   // ascending v per lane: first maximum wins
                }
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) {
                    const float ov = __shfl_xor(bv, o);
                    const int oi = __shfl_xor(bi, o);
                    if (oi >= 0 && (bi < 0 || ov > bv || (ov == bv && oi < bi))) { bv = ov; bi = oi; }
                }
                if (bi >= 0 && (bi & 63) == lane) used |= 1u << (bi >> 6);
                if (lane == 0) { candv[k * kMaxBeam + r] = bv; candi[k * kMaxBeam + r] = bi < 0 ? -1 : k * V + bi; }
            }
        }
    } else
    for (int k = wave; k < nA; k += BT / 64) {
        const float* l = lg + (long long)k * V;
        float mx = -INFINITY;
        for (int v = lane; v < V; v += 64) mx = fmaxf(mx, l[v]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
        float sm = 0.f;
        for (int v = lane; v < V; v += 64) sm += expf(l[v] - mx);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sm += __shfl_xor(sm, o);
        const float lse = logf(sm), lpk = s.lp[b * K + k];
        int excl[kMaxBeam];
        for (int r = 0; r < want; r++) {
            float bv = -INFINITY;
            int bi = -1;
            for (int v = lane; v < V; v += 64) {
                bool tk = false;
                for (int q = 0; q < r; q++) tk |= (excl[q] == v);
                if (tk) continue;
                const float sc = (l[v] - mx - lse) + lpk;  // the oracle's order of operations
                if (bi < 0 || sc > bv) { bv = sc; bi = v; }   // ascending v per lane: first maximum wins
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const float ov = __shfl_xor(bv, o);
                const int oi = __shfl_xor(bi, o);
                if (oi >= 0 && (bi < 0 || ov > bv || (ov == bv && oi < bi))) { bv = ov; bi = oi; }
            }
            excl[r] = bi;
            if (lane == 0) { candv[k * kMaxBeam + r] = bv; candi[k * kMaxBeam + r] = bi < 0 ? -1 : k * V + bi; }
        }
    }
    __syncthreads();
    if (wave == 0) {  // merge the nA x want candidates: lane = k * kMaxBeam + r
        const bool has = (lane / kMaxBeam) < nA && (lane % kMaxBeam) < want;
        float myv = has ? candv[lane] : -INFINITY;
        int myi = has ? candi[lane] : -1;
        for (int r = 0; r < want; r++) {
            float bv = myv;
            int bi = myi;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const float ov = __shfl_xor(bv, o);
                const int oi = __shfl_xor(bi, o);
                if (oi >= 0 && (bi < 0 || ov > bv || (ov == bv && oi < bi))) { bv = ov; bi = oi; }
            }
            if (lane == 0) { topv[r] = bv; topi[r] = bi; taken[r] = bi; }
            if (myi == bi) myi = -1;  // that candidate is used up (flat indexes are unique)
        }
    }
    __syncthreads();
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
    s.lp_next = s.lp;  // in place: within k_beam_step every read of a stream's lp / nhyp precedes (barrier-separated) tid 0's write
    s.ctx = ar.take<long long>((int64_t)2 * M);
    s.ctx_next = s.ctx;
    s.nhyp = ar.take<int>(B);
    s.nhyp_next = s.nhyp;
    float* hbuf = ar.take<float>((int64_t)M * w.DD);
    float* act = ar.take<float>((int64_t)M * w.J);
    float* logits = ar.take<float>((int64_t)M * w.V);
    if (!ctx.dry) {
        K2_HIP(hipMemsetAsync(a.overflow, 0, sizeof(int), ctx.stream));
        hipLaunchKernelGGL(k_beam_init, dim3(cdiv(M, 256)), dim3(256), 0, ctx.stream, s, B);
        K2_HIP(hipGetLastError());
    }
    for (int t = 0; t < a.Tp; t++) {
        // four launches per frame: the decoder front end (embedding + conv + ReLU, elementwise); act = tanh(enc_t + decoder_proj(h)) as
        // ONE small-problem GEMM whose epilogue adds the stream's encoder frame (row / K) before the tanh; the joiner GEMM; the
        // per-stream step.  (Folding the front end into the step's tail only moved its time into the 32-workgroup step kernel.)
        if (!ctx.dry) hipLaunchKernelGGL(k_beam_embconv, dim3(cdiv((long long)M * w.DD, 256)), dim3(256), 0, ctx.stream, w, s.ctx, hbuf, M);
        {
            GemmArgs g;
            g.A = hbuf; g.lda = w.DD; g.W = a.dproj_w; g.ldw = w.DD; g.bias = w.dproj_b;
            g.C = act; g.ldc = w.J; g.M = M; g.N = w.J; g.K = w.DD;
            g.act = ACT_TANH; g.act_after_res = 1; g.res = a.enc + (long long)t * w.J; g.ldr = a.Tp * w.J; g.res_div = K;
            gemm(ctx, g);
        }
        linear(ctx, act, w.J, a.out_w, w.out_b, logits, w.V, M, w.J, w.V);
        if (!ctx.dry) {
            hipLaunchKernelGGL(k_beam_step, dim3(B), dim3(BT), 0, ctx.stream, s, logits, w.V, t, t & 1, B);
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
