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
#include <type_traits>

#include "kernels.h"
#include "sweep.h"

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

// ---- wave-wide reductions on data-parallel-primitive lane moves (no LDS crossbar: a ds_bpermute butterfly of 6 steps x 2 values is a
// chain of ~900 cycles, and the step runs ~10 of them per frame) ------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) {
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, false);
}
constexpr int kDppXor1 = 0xB1, kDppXor2 = 0x4E, kDppHalfMirror = 0x141, kDppMirror = 0x140;  // quad_perm [1,0,3,2] / [2,3,0,1], row_half_mirror, row_mirror
// candidate order of the search: higher score first, then the LOWER flat index; index < 0 = no candidate
__device__ __forceinline__ void best_merge(float& bv, int& bi, float ov, int oi) {
    if (oi >= 0 && (bi < 0 || ov > bv || (ov == bv && oi < bi))) { bv = ov; bi = oi; }
}
// the best (score, index) of the wave, in every lane (a total order: the sequence of merges does not matter)
__device__ __forceinline__ void wave_best(float& bv, int& bi) {
    best_merge(bv, bi, dpp_f<kDppXor1>(bv), dpp_i<kDppXor1>(bi));
    best_merge(bv, bi, dpp_f<kDppXor2>(bv), dpp_i<kDppXor2>(bi));
    best_merge(bv, bi, dpp_f<kDppHalfMirror>(bv), dpp_i<kDppHalfMirror>(bi));
    best_merge(bv, bi, dpp_f<kDppMirror>(bv), dpp_i<kDppMirror>(bi));   // every lane: its row of 16
    float rv = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, bv), 0));
    int ri = __builtin_amdgcn_readlane(bi, 0);
#pragma unroll
    for (int row = 1; row < 4; row++)
        best_merge(rv, ri, __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, bv), 16 * row)), __builtin_amdgcn_readlane(bi, 16 * row));
    bv = rv;
    bi = ri;
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, dpp_f<kDppXor1>(v));
    v = fmaxf(v, dpp_f<kDppXor2>(v));
    v = fmaxf(v, dpp_f<kDppHalfMirror>(v));
    v = fmaxf(v, dpp_f<kDppMirror>(v));
    float r = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
#pragma unroll
    for (int row = 1; row < 4; row++) r = fmaxf(r, __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16 * row)));
    return r;
}

// One stream's hypotheses as the step sees them: frame t's input buffers (_c), the other parity's (_n), and the in-place state.
// Global memory (k_beam_step: one launch per frame) or LDS (k_beam_loop: the whole search in one kernel).
struct HypView {
    int K, cap;
    const int *ys_c, *ts_c, *n_c;
    int *ys_n, *ts_n, *n_n;
    float* lp;         // [K] in place: every read of it precedes (barrier-separated) tid 0's write
    long long* ctx;    // [K][2]
    int* nhyp;
    // debug tap (k2hip_debug.h, K2HIP_BEAM_TRACE) or null: this frame's record of this stream, 2 K + 1 words = the selected candidates'
    // flat indexes (slot * V + token) in rank order, their scores (float bits), the number of hypotheses after the merges
    int* trace = nullptr;
};
constexpr int kStepScratchInts = 4 * kMaxBeam + 4 + 2 * kMaxBeam * kMaxBeam;
// one workgroup of NT threads per stream; lg: the hypotheses' logits, ldl floats per row; scratch: kStepScratchInts ints of LDS
template <int NT>
__device__ void beam_step_body(const HypView& hv, const float* lg, int ldl, int V, int t, int* scratch) {
    constexpr int BT = NT;
    int* taken = scratch;
    float* topv = reinterpret_cast<float*>(scratch + kMaxBeam);
    int* topi = scratch + 2 * kMaxBeam;
    float* candv = reinterpret_cast<float*>(scratch + 4 * kMaxBeam + 4);
    int* candi = scratch + 4 * kMaxBeam + 4 + kMaxBeam * kMaxBeam;
    const int tid = threadIdx.x, K = hv.K;
    const int nA = *hv.nhyp;
    // ---- one wave per hypothesis (4 waves, K <= 8): log_softmax statistics, then the hypothesis' own top `want` candidates by
    //      (score desc, token asc) with wave shuffles only; the global top `want` is the top of the union (no block-wide
    //      reductions, which is what made this step cost more than the joiner GEMM)
    const int nc = nA * V, want = min(K, nc);
    const int lane = tid & 63, wave = tid >> 6;
    if (V <= 64 * 16) {
        // the hypothesis' logits, V / 64 per lane, are read ONCE into registers: the max, the sum and the `want` selection rounds below
        // went through global memory 2 + want times before (a chain of dependent loads that made this step 20 us)
        auto select = [&](auto ne_tag) {
            constexpr int NE = decltype(ne_tag)::value;   // elements per lane
            for (int k = wave; k < nA; k += BT / 64) {
                const float* l = lg + (long long)k * ldl;
                float lv[NE];
#pragma unroll
                for (int i = 0; i < NE; i++) lv[i] = lane + 64 * i < V ? l[lane + 64 * i] : -INFINITY;
                float mx = -INFINITY;
#pragma unroll
                for (int i = 0; i < NE; i++) mx = fmaxf(mx, lv[i]);
                mx = wave_max(mx);
                float sm = 0.f;
#pragma unroll
                for (int i = 0; i < NE; i++)
                    if (lane + 64 * i < V) sm += expf(lv[i] - mx);
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) sm += __shfl_xor(sm, o);
                const float lse = logf(sm), lpk = hv.lp[k];
#pragma unroll
                for (int i = 0; i < NE; i++) lv[i] = (lv[i] - mx - lse) + lpk;  // the scores, in the oracle's order of operations
                unsigned used = 0;  // bit i: this lane's element i was already selected
                for (int r = 0; r < want; r++) {
                    float bv = -INFINITY;
                    int bi = -1;
#pragma unroll
                    for (int i = 0; i < NE; i++) {
                        const int v = lane + 64 * i;
                        if (v >= V || (used >> i & 1)) continue;
                        if (bi < 0 || lv[i] > bv) { bv = lv[i]; bi = v; }   // ascending v per lane: first maximum wins
                    }
                    wave_best(bv, bi);
                    if (bi >= 0 && (bi & 63) == lane) used |= 1u << (bi >> 6);
                    if (lane == 0) { candv[k * kMaxBeam + r] = bv; candi[k * kMaxBeam + r] = bi < 0 ? -1 : k * V + bi; }
                }
            }
        };
        if (V <= 64 * 8) select(std::integral_constant<int, 8>{});
        else select(std::integral_constant<int, 16>{});
    } else
    for (int k = wave; k < nA; k += BT / 64) {
        const float* l = lg + (long long)k * ldl;
        float mx = -INFINITY;
        for (int v = lane; v < V; v += 64) mx = fmaxf(mx, l[v]);
        mx = wave_max(mx);
        float sm = 0.f;
        for (int v = lane; v < V; v += 64) sm += expf(l[v] - mx);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sm += __shfl_xor(sm, o);
        const float lse = logf(sm), lpk = hv.lp[k];
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
            wave_best(bv, bi);
            excl[r] = bi;
            if (lane == 0) { candv[k * kMaxBeam + r] = bv; candi[k * kMaxBeam + r] = bi < 0 ? -1 : k * V + bi; }
        }
    }
    __syncthreads();
    if (wave == 0) {  // merge the nA x want candidates: lane = k * kMaxBeam + r
        // Each candidate's RANK among all of them (how many beat it: higher score, then lower flat index -- a total order, flat
        // indexes are unique): every candidate is broadcast in turn with v_readlane and compared by all lanes at once, no dependent
        // rounds (`want` rounds of a wave-wide best, each waiting for the one before, were 2 us of the step).
        const bool has = (lane / kMaxBeam) < nA && (lane % kMaxBeam) < want;
        const float myv = has ? candv[lane] : -INFINITY;
        const int myi = has ? candi[lane] : -1;
        int rank = 0;
        for (int k = 0; k < nA; k++)
            for (int r = 0; r < want; r++) {
                const int src = k * kMaxBeam + r;   // (uniform)
                const float ov = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, myv), src));
                const int oi = __builtin_amdgcn_readlane(myi, src);
                if (oi >= 0 && (ov > myv || (ov == myv && oi < myi))) rank++;
            }
        if (myi >= 0 && rank < want) { topv[rank] = myv; topi[rank] = myi; taken[rank] = myi; }
    }
    __syncthreads();
    // ---- expand + merge (HypothesisList.add) into the other buffer
    const int *ys_c = hv.ys_c, *ts_c = hv.ts_c, *n_c = hv.n_c;
    int *ys_n = hv.ys_n, *ts_n = hv.ts_n, *n_n = hv.n_n;
    // sequence of candidate r = ys_c[parent_r][0..n) (+ tok_r if real).  Which candidates spell the same sequence: every pair (r, q < r)
    // compared by a wave of its own (one barrier for all pairs; a loop over the pairs with three block barriers each was a third of
    // the step), then tid 0 resolves the merges in insertion order
    int* eq = candi;   // [r][q] (the per-hypothesis candidates are used up)
    {
        const int npairs = want * (want - 1) / 2;
        for (int pi = wave; pi < npairs; pi += BT / 64) {
            int r = 1, base = 0;
            while (base + r <= pi) { base += r; r++; }
            const int q = pi - base;
            const int hr = topi[r] / V, tr = topi[r] % V, hq = topi[q] / V, tq = topi[q] % V;
            const bool realr = tr != K2HIP_BLANK_ID && tr != K2HIP_UNK_ID, realq = tq != K2HIP_BLANK_ID && tq != K2HIP_UNK_ID;
            const int lenr = n_c[hr] + (realr ? 1 : 0), lenq = n_c[hq] + (realq ? 1 : 0);
            bool e = lenr == lenq;
            if (e) {
                for (int i = lane; i < lenr; i += 64) {
                    const int x = (i < n_c[hr]) ? ys_c[(long long)hr * hv.cap + i] : tr;
                    const int y = (i < n_c[hq]) ? ys_c[(long long)hq * hv.cap + i] : tq;
                    e = e && x == y;
                }
            }
            e = __all(e);
            if (lane == 0) eq[r * kMaxBeam + q] = e ? 1 : 0;
        }
    }
    __syncthreads();
    if (tid == 0) {
        // One thread resolves the merges in insertion order.  Everything it needs is read up front (independent LDS loads) and kept in
        // registers -- fully unrolled, selects instead of indexed arrays: as a loop over LDS this was a chain of ~40 dependent
        // accesses, 2 us of the step.
        float tv[kMaxBeam], lpn[kMaxBeam];
        unsigned eqm[kMaxBeam];   // bit q of eqm[r]: candidates r and q < r spell the same sequence
#pragma unroll
        for (int r = 0; r < kMaxBeam; r++) {
            tv[r] = r < want ? topv[r] : -INFINITY;
            lpn[r] = -INFINITY;
            eqm[r] = 0;
#pragma unroll
            for (int q = 0; q < r; q++)
                if (r < want && eq[r * kMaxBeam + q]) eqm[r] |= 1u << q;
        }
        unsigned merged = 0;      // bit q: candidate q was merged into an earlier one
        int slot_of[kMaxBeam];
        int nN = 0;
#pragma unroll
        for (int r = 0; r < kMaxBeam; r++) {
            if (r < want) {
                // the earliest earlier entry with the same sequence that was not merged away itself, or none
                const unsigned cand = eqm[r] & ~merged;
                const int found = cand ? __ffs(cand) - 1 : -1;
                int sl = nN;
                if (found >= 0) {
                    merged |= 1u << r;
#pragma unroll
                    for (int q = 0; q < r; q++)
                        if (q == found) sl = slot_of[q];
                }
                slot_of[r] = sl;
                // slot assignment in insertion order; merged scores accumulate in candidate order (logaddexp)
                float cur = -INFINITY;
#pragma unroll
                for (int q = 0; q < kMaxBeam; q++)
                    if (q == sl) cur = lpn[q];
                float nv = tv[r];
                if (found >= 0) {
                    const float mx = fmaxf(cur, tv[r]);
                    nv = (isinf(mx) && mx < 0) ? mx : mx + log1pf(expf(-fabsf(cur - tv[r])));
                } else {
                    nN++;
                }
#pragma unroll
                for (int q = 0; q < kMaxBeam; q++)
                    if (q == sl) lpn[q] = nv;
                taken[r] = found < 0 ? sl : -1;  // reuse: destination slot of a fresh hypothesis
            }
        }
#pragma unroll
        for (int q = 0; q < kMaxBeam; q++)
            if (q < nN) hv.lp[q] = lpn[q];
        for (int k = nN; k < K; k++) {
            hv.lp[k] = -INFINITY;
            // empty slots still go through the batched decoder launch: give them a valid context
            hv.ctx[2 * k] = K2HIP_BLANK_ID;
            hv.ctx[2 * k + 1] = K2HIP_BLANK_ID;
            n_n[k] = 0;
        }
        *hv.nhyp = nN;
        if (hv.trace) hv.trace[2 * K] = nN;
    }
    if (hv.trace && tid < K) {
        hv.trace[tid] = tid < want ? topi[tid] : -1;
        hv.trace[K + tid] = __float_as_int(tid < want ? topv[tid] : -INFINITY);
    }
    __syncthreads();
    for (int r = wave; r < want; r += BT / 64) {   // a wave per surviving candidate (distinct destination slots)
        const int slot = taken[r];
        if (slot < 0) continue;
        const int hr = topi[r] / V, tr = topi[r] % V;
        const bool realr = tr != K2HIP_BLANK_ID && tr != K2HIP_UNK_ID;
        const int n0 = n_c[hr];
        for (int i = lane; i < n0; i += 64) {
            ys_n[(long long)slot * hv.cap + i] = ys_c[(long long)hr * hv.cap + i];
            ts_n[(long long)slot * hv.cap + i] = ts_c[(long long)hr * hv.cap + i];
        }
        if (lane == 0) {
            int nn = n0;
            if (realr) {
                if (nn < hv.cap) {
                    ys_n[(long long)slot * hv.cap + nn] = tr;
                    ts_n[(long long)slot * hv.cap + nn] = t;
                }
                nn++;
            }
            n_n[slot] = nn;
            // decoder context of the new hypothesis: last two of [blank, blank] + ys
            long long y1 = nn >= 1 ? (realr ? tr : ys_c[(long long)hr * hv.cap + n0 - 1]) : K2HIP_BLANK_ID;
            long long y0 = K2HIP_BLANK_ID;
            if (nn >= 2) y0 = realr ? ys_c[(long long)hr * hv.cap + n0 - 1] : ys_c[(long long)hr * hv.cap + n0 - 2];
            hv.ctx[2 * slot] = y0;
            hv.ctx[2 * slot + 1] = y1;
        }
    }
}

// one workgroup per stream; `cur` = buffer holding frame t's input hypotheses
__global__ __launch_bounds__(BT) void k_beam_step(BeamState s, const float* __restrict__ logits, int V, int t, int cur, int B, int* trace, int Tp) {
    __shared__ int scratch[kStepScratchInts];
    const int b = blockIdx.x, K = s.K, nxt = cur ^ 1;
    const long long BK = (long long)B * K;
    HypView hv;
    hv.K = K; hv.cap = s.cap;
    hv.ys_c = s.ys + ((long long)cur * BK + (long long)b * K) * s.cap;
    hv.ts_c = s.ts + ((long long)cur * BK + (long long)b * K) * s.cap;
    hv.ys_n = s.ys + ((long long)nxt * BK + (long long)b * K) * s.cap;
    hv.ts_n = s.ts + ((long long)nxt * BK + (long long)b * K) * s.cap;
    hv.n_c = s.n + cur * BK + b * K;
    hv.n_n = s.n + nxt * BK + b * K;
    hv.lp = s.lp + b * K;
    hv.ctx = s.ctx + 2 * (long long)b * K;
    hv.nhyp = s.nhyp + b;
    hv.trace = trace ? trace + ((long long)b * Tp + t) * (2 * K + 1) : nullptr;
    beam_step_body<BT>(hv, logits + (long long)b * K * V, V, V, t, scratch);
}

// ---- the whole search of a stream in one kernel (small vocabularies: the model's all-contexts decoder table) -----------------
// One workgroup of GT threads per stream walks the T' frames: the K hypotheses' decoder outputs are rows of the table, their
// joiner inputs tanh(enc_t + dec_k) take the place of k_greedy's 8 speculated frames in the same sweep of the joiner matrix on the
// matrix pipe (mfma_sweep_rows: 8 hypotheses x 256 columns per pass), the logits stay in LDS, and the step (log-softmax, top K,
// expand, merge) is beam_step_body on hypotheses that live in LDS as well.  Replaces 4 launches per frame (1012 for the headline
// batch, each waiting for slots between the next batch's encoder GEMMs) by one launch per batch.
// Forms of the sweep: beam <= 4 uses the lower 4 x 4 half of each MFMA; with V <= 512 both 256-column chunks go through one pass
// (mfma_sweep_rows_2c), or -- when 2 B workgroups fit the co-residency budget -- each chunk gets a workgroup of its own and the two
// exchange their 4 x 256 logits per frame as tagged granules (BeamLoopArgs::xg; bounded waits, *overflow = 2 on a timeout and a repeat
// with one workgroup per stream, as for k_greedy's parts).
typedef __attribute__((address_space(1))) unsigned long long bgu64;
__device__ __forceinline__ void bstore_granule(unsigned long long* g, unsigned epoch, unsigned value) {
    __hip_atomic_store((bgu64*)g, ((unsigned long long)epoch << 32) | value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long bload_granule(const unsigned long long* g) {
    return __hip_atomic_load((bgu64*)g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// LDS (floats): actT[J GF] | psum | lg[GF][Vp] | ctx[2 GF] (long long) | lp[GF] | n[2][GF] | nhyp, pad | scratch | ys[2][K][cap] | ts[2][K][cap]
__host__ __device__ inline size_t beam_loop_lds_floats(int J, int Vp, int K, int cap, bool hyp_in_lds) {
    return (size_t)J * GF + kPsumFloats + (size_t)GF * Vp + 4 * GF + GF + 2 * GF + 4 + kStepScratchInts + 4 + (hyp_in_lds ? 4 * (size_t)K * cap : 0);
}
template <int NH>   // NH = 1: beam <= 4, the sweep forms only rows 0..3
__global__ __launch_bounds__(GT) void k_beam_loop(DecJoinW w, BeamLoopArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* actT = sm;
    float* psum = actT + w.J * GF;
    float* lg = psum + kPsumFloats;
    long long* ctx = reinterpret_cast<long long*>(lg + GF * w.Vp);   // (J GF, the psum size and GF Vp are multiples of 4 floats)
    float* lp = reinterpret_cast<float*>(ctx + 2 * GF);
    int* nbuf = reinterpret_cast<int*>(lp + GF);       // [2][GF]
    int* nhyp = nbuf + 2 * GF;
    int* scratch = nhyp + 4;
    int* ys = a.ys_g ? a.ys_g + (size_t)blockIdx.x * 2 * a.K * a.cap : scratch + kStepScratchInts + 4;
    int* ts = a.ys_g ? a.ts_g + (size_t)blockIdx.x * 2 * a.K * a.cap : ys + 2 * a.K * a.cap;
    __shared__ int xfail;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, K = a.K;
    if (tid == 0) xfail = 0;
    const int b = a.xg ? blockIdx.x >> 1 : blockIdx.x, slab = a.xg ? blockIdx.x & 1 : 0;
    const float* enc = a.enc + (long long)b * a.Tp * w.J;
    if (tid < GF) {   // k_beam_init
        lp[tid] = tid == 0 ? 0.f : -INFINITY;
        nbuf[tid] = 0;
        nbuf[GF + tid] = 0;
        ctx[2 * tid] = K2HIP_BLANK_ID;
        ctx[2 * tid + 1] = K2HIP_BLANK_ID;
        if (tid == 0) *nhyp = 1;
    }
    __syncthreads();
    const int kper = w.J >> 3, ncg = w.Vp >> 2;
    for (int t = 0; t < a.Tp; t++) {
        const int nA = *nhyp, cur = t & 1;
        // joiner inputs of the live hypotheses (row q of the sweep = hypothesis q); rows past nA are zero and ignored
        for (int k = tid; k < w.J; k += GT) {
            const float e = enc[(long long)t * w.J + k];
            float d[4 * NH];
#pragma unroll
            for (int q = 0; q < 4 * NH; q++) {
                const long long y0 = ctx[2 * min(q, K - 1)], y1 = ctx[2 * min(q, K - 1) + 1];   // (empty slots hold [blank, blank])
                d[q] = w.dec_table[((y0 + 1) * w.V + y1) * (long long)w.J + k];
            }
#pragma unroll
            for (int q = 0; q < GF; q++) actT[k * GF + q] = (q < 4 * NH && q < nA) ? tanhf(e + d[q < 4 * NH ? q : 0]) : 0.f;
        }
        __syncthreads();
        if (NH == 1 && a.xg) {
            // two slabs: this workgroup's 256-column chunk, then the peer's logits through tagged granules
            f32x4 c[2][4];
#pragma unroll
            for (int hh = 0; hh < 2; hh++)
#pragma unroll
                for (int q = 0; q < 4; q++) c[hh][q] = f32x4{0.f, 0.f, 0.f, 0.f};
            const int cg = 64 * slab + lane;
            mfma_sweep_rows<1>(w.out_kn + (long long)(wave * kper) * w.Vp + 4 * min(cg, ncg - 1), w.Vp, actT + (wave * kper) * GF + (lane & 3), kper, c);
            psum_store<1>(psum, wave, lane, c);
            __syncthreads();
            unsigned long long* xo = a.xg + ((((long long)b * 2 + (t & 1)) * 2 + slab) * 4) * 256;
            const unsigned long long* xi = a.xg + ((((long long)b * 2 + (t & 1)) * 2 + (slab ^ 1)) * 4) * 256;
            if (wave < 4) {   // wave q: hypothesis q's 256 logits of this slab (computed for every slot: the peer must not wait on nA)
                float4 ps[8];
#pragma unroll
                for (int sl = 0; sl < 8; sl++) ps[sl] = *reinterpret_cast<const float4*>(psum + (sl * GF + wave) * 256 + 4 * lane);
                float sj[4];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const float p0 = (&ps[0].x)[j], p1 = (&ps[1].x)[j], p2 = (&ps[2].x)[j], p3 = (&ps[3].x)[j];
                    const float p4 = (&ps[4].x)[j], p5 = (&ps[5].x)[j], p6 = (&ps[6].x)[j], p7 = (&ps[7].x)[j];
                    const int col = 4 * min(cg, ncg - 1) + j;
                    sj[j] = (((p0 + p1) + (p2 + p3)) + ((p4 + p5) + (p6 + p7))) + (col < w.V ? w.out_b[col] : 0.f);
                }
                if (cg < ncg) *reinterpret_cast<float4*>(lg + wave * w.Vp + 4 * cg) = make_float4(sj[0], sj[1], sj[2], sj[3]);
#pragma unroll
                for (int j = 0; j < 4; j++) bstore_granule(xo + wave * 256 + 4 * lane + j, (unsigned)(t + 1), __float_as_uint(sj[j]));
            }
            // every thread collects 2 of the peer's 1024 granules
            for (int e = tid; e < 4 * 256; e += GT) {
                unsigned long long x = bload_granule(xi + e);
                for (unsigned spins = 0; (unsigned)(x >> 32) != (unsigned)(t + 1) && spins < (1u << 22); spins++) {
                    __builtin_amdgcn_s_sleep(1);
                    x = bload_granule(xi + e);
                }
                if ((unsigned)(x >> 32) != (unsigned)(t + 1)) xfail = 1;
                const int q = e >> 8, col = 256 * (slab ^ 1) + (e & 255);
                if (col < w.Vp) lg[q * w.Vp + col] = __uint_as_float((unsigned)x);
            }
            __syncthreads();
            if (xfail) {
                if (tid == 0) *a.overflow = 2;
                return;
            }
        } else if (NH == 1 && ncg <= 128) {
            // beam <= 4 and V <= 512: both 256-column chunks in ONE pass -- the 4 live rows leave half of the accumulators and of the
            // exchange area free, so the second chunk rides along: one LDS exchange and one pair of barriers per frame instead of two
            f32x4 c[2][4];
#pragma unroll
            for (int hh = 0; hh < 2; hh++)
#pragma unroll
                for (int q = 0; q < 4; q++) c[hh][q] = f32x4{0.f, 0.f, 0.f, 0.f};
            const float* wrow = w.out_kn + (long long)(wave * kper) * w.Vp;
            mfma_sweep_rows_2c(wrow + 4 * min(lane, ncg - 1), wrow + 4 * min(64 + lane, ncg - 1), w.Vp, actT + (wave * kper) * GF + (lane & 3), kper, c);
            // psum[slice][row 0..3][512 columns]
#pragma unroll
            for (int ch = 0; ch < 2; ch++)
#pragma unroll
                for (int i = 0; i < 4; i++)
                    *reinterpret_cast<float4*>(psum + ((wave * 4) + i) * 512 + 256 * ch + 4 * lane) = make_float4(c[ch][0][i], c[ch][1][i], c[ch][2][i], c[ch][3][i]);
            __syncthreads();
            {
                const int q = wave & 3, ch = wave >> 2, cg = 64 * ch + lane;   // wave: hypothesis q's chunk ch
                if (cg < ncg && q < nA) {
                    float4 ps[8];
#pragma unroll
                    for (int sl = 0; sl < 8; sl++) ps[sl] = *reinterpret_cast<const float4*>(psum + ((sl * 4) + q) * 512 + 256 * ch + 4 * lane);
                    float sj[4];
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const float p0 = (&ps[0].x)[j], p1 = (&ps[1].x)[j], p2 = (&ps[2].x)[j], p3 = (&ps[3].x)[j];
                        const float p4 = (&ps[4].x)[j], p5 = (&ps[5].x)[j], p6 = (&ps[6].x)[j], p7 = (&ps[7].x)[j];
                        const int col = 4 * cg + j;
                        sj[j] = (((p0 + p1) + (p2 + p3)) + ((p4 + p5) + (p6 + p7))) + (col < w.V ? w.out_b[col] : 0.f);
                    }
                    *reinterpret_cast<float4*>(lg + q * w.Vp + 4 * cg) = make_float4(sj[0], sj[1], sj[2], sj[3]);
                }
            }
            __syncthreads();
        } else
        for (int cgb = 0; cgb < ncg; cgb += 64) {
            const int cg = cgb + lane;
            f32x4 c[2][4];
#pragma unroll
            for (int hh = 0; hh < 2; hh++)
#pragma unroll
                for (int q = 0; q < 4; q++) c[hh][q] = f32x4{0.f, 0.f, 0.f, 0.f};
            mfma_sweep_rows<NH>(w.out_kn + (long long)(wave * kper) * w.Vp + 4 * min(cg, ncg - 1), w.Vp, actT + (wave * kper) * GF + (lane & 3), kper, c);
            psum_store<NH>(psum, wave, lane, c);
            __syncthreads();
            if (cg < ncg && wave < nA) {   // wave q: hypothesis q's 256 logits of this pass, slices added in k_greedy's order
                float4 ps[8];
#pragma unroll
                for (int q = 0; q < 8; q++) ps[q] = *reinterpret_cast<const float4*>(psum + (q * GF + wave) * 256 + 4 * lane);
                float sj[4];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const float p0 = (&ps[0].x)[j], p1 = (&ps[1].x)[j], p2 = (&ps[2].x)[j], p3 = (&ps[3].x)[j];
                    const float p4 = (&ps[4].x)[j], p5 = (&ps[5].x)[j], p6 = (&ps[6].x)[j], p7 = (&ps[7].x)[j];
                    const int col = 4 * cg + j;
                    sj[j] = (((p0 + p1) + (p2 + p3)) + ((p4 + p5) + (p6 + p7))) + (col < w.V ? w.out_b[col] : 0.f);
                }
                *reinterpret_cast<float4*>(lg + wave * w.Vp + 4 * cg) = make_float4(sj[0], sj[1], sj[2], sj[3]);
            }
            __syncthreads();
        }
        HypView hv;
        hv.K = K; hv.cap = a.cap;
        hv.ys_c = ys + (size_t)cur * K * a.cap; hv.ts_c = ts + (size_t)cur * K * a.cap; hv.n_c = nbuf + cur * GF;
        hv.ys_n = ys + (size_t)(cur ^ 1) * K * a.cap; hv.ts_n = ts + (size_t)(cur ^ 1) * K * a.cap; hv.n_n = nbuf + (cur ^ 1) * GF;
        hv.lp = lp; hv.ctx = ctx; hv.nhyp = nhyp;
        hv.trace = (a.trace && slab == 0) ? a.trace + ((long long)b * a.Tp + t) * (2 * K + 1) : nullptr;
        beam_step_body<GT>(hv, lg, w.Vp, w.V, t, scratch);
        __syncthreads();
    }
    // k_beam_final: max of log_prob / len(ys) (len counts the 2 ctx blanks), first maximum
    const int fin = a.Tp & 1;
    const int* n_f = nbuf + fin * GF;
    int best = 0;
    float bs = lp[0] / (float)(n_f[0] + 2);
    for (int k = 1; k < *nhyp; k++) {
        const float v = lp[k] / (float)(n_f[k] + 2);
        if (v > bs) { bs = v; best = k; }
    }
    const int n = n_f[best];
    if (n > a.max_tokens || n > a.cap) {
        if (tid == 0) *a.overflow = 1;
        return;
    }
    const int* ysf = ys + ((size_t)fin * K + best) * a.cap;
    const int* tsf = ts + ((size_t)fin * K + best) * a.cap;
    if (slab != 0) return;
    for (int i = tid; i < n; i += GT) {
        a.tokens[(long long)b * a.max_tokens + i] = ysf[i];
        a.timestamps[(long long)b * a.max_tokens + i] = tsf[i];
    }
    if (tid == 0) {
        a.n_tokens[b] = n;
        if (a.scores) a.scores[b] = lp[best];
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
    {
        // the one-kernel form: needs the decoder table (small vocabulary) and the stream's logits and hypotheses in LDS
        // hypotheses in LDS when they fit beside the logits (T' <= ~600 at beam 4 with the large-en shapes), else in device memory
        const bool hyp_in_lds = sizeof(float) * beam_loop_lds_floats(w.J, w.Vp, K, cap, true) <= 150 * 1024 && !tunables().beam_hyp_global;
        const size_t lds = sizeof(float) * beam_loop_lds_floats(w.J, w.Vp, K, cap, hyp_in_lds);
        if (w.dec_table && !tunables().beam_launches && lds <= 150 * 1024 && w.J % 8 == 0 && w.Vp % 4 == 0 && K <= GF) {
            // Two column slabs per stream where one workgroup would sweep two chunks (beam <= 4, 256 < V <= 512) and 2 B workgroups fit
            // the offline co-residency budget: the frame's sweep is bound by ONE CU's fetch of the 1 MB matrix, two CUs halve it, and the
            // exchange of the 4 x 256 logits costs less than the half sweep (search alone 5.5 -> 4.45 ms for the headline batch).  The
            // slabs wait for each other (bounded; *overflow = 2 on a timeout): the launch is kept for a repeat with one slab.
            const bool two = !ctx.one_part && tunables().beam_parts != 1 && K <= 4 && (w.Vp >> 2) <= 128 && (w.Vp >> 2) > 64 && 2 * B <= std::max(device_cu_count() / 4, 2);
            unsigned long long* xg = two ? ar.take<unsigned long long>((int64_t)B * 2 * 2 * 4 * 256) : nullptr;
            // (hypotheses in device memory: every workgroup keeps its own copy, so two slabs need twice the space)
            int* ys_g = hyp_in_lds ? nullptr : ar.take<int>((int64_t)(two ? 2 : 1) * 2 * M * cap);
            int* ts_g = hyp_in_lds ? nullptr : ar.take<int>((int64_t)(two ? 2 : 1) * 2 * M * cap);
            if (ctx.dry) return;
            BeamLoopArgs la;
            la.enc = a.enc; la.Tp = a.Tp; la.K = K; la.cap = cap;
            la.tokens = a.tokens; la.timestamps = a.timestamps; la.n_tokens = a.n_tokens; la.scores = a.scores;
            la.max_tokens = a.max_tokens; la.overflow = a.overflow;
            la.ys_g = ys_g; la.ts_g = ts_g;
            la.xg = xg;
            la.trace = a.trace;
            K2_HIP(hipMemsetAsync(a.overflow, 0, sizeof(int), ctx.stream));
            static LdsAttrOnce lds_attr;
            static LdsAttrOnce lds_attr1;
            lds_attr.ensure(k_beam_loop<2>, 150 * 1024);
            lds_attr1.ensure(k_beam_loop<1>, 150 * 1024);
            if (two) K2_HIP(hipMemsetAsync(xg, 0, sizeof(unsigned long long) * (size_t)B * 2 * 2 * 4 * 256, ctx.stream));
            if (K <= 4) hipLaunchKernelGGL(k_beam_loop<1>, dim3(two ? 2 * B : B), dim3(GT), lds, ctx.stream, w, la);
            else hipLaunchKernelGGL(k_beam_loop<2>, dim3(B), dim3(GT), lds, ctx.stream, w, la);
            K2_HIP(hipGetLastError());
            if (ctx.greedy_rec) {
                ctx.greedy_rec->valid = two;
                ctx.greedy_rec->beam = true;
                ctx.greedy_rec->w = w;
                ctx.greedy_rec->a.B = B;
                ctx.greedy_rec->a.overflow = a.overflow;
                ctx.greedy_rec->ba = la;
                ctx.greedy_rec->beam_lds = lds;
            }
            if (two && tunables().test_greedy_timeout) K2_HIP(hipMemsetD32Async((hipDeviceptr_t)a.overflow, 2, 1, ctx.stream));
            return;
        }
    }
    if (ctx.greedy_rec) ctx.greedy_rec->valid = false;
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
            hipLaunchKernelGGL(k_beam_step, dim3(B), dim3(BT), 0, ctx.stream, s, logits, w.V, t, t & 1, B, a.trace, a.Tp);
            K2_HIP(hipGetLastError());
        }
    }
    if (!ctx.dry) {
        hipLaunchKernelGGL(k_beam_final, dim3(B), dim3(256), 0, ctx.stream, s, a.Tp & 1, B, a.tokens, a.timestamps, a.n_tokens, a.scores,
                           a.max_tokens, a.overflow);
        K2_HIP(hipGetLastError());
    }
}

void beam_relaunch_one_slab(hipStream_t stream, const GreedyLaunch& rec) {
    K2_REQUIRE(rec.valid && rec.beam, "beam retry: no repeatable launch on record");
    BeamLoopArgs la = rec.ba;
    la.xg = nullptr;
    K2_HIP(hipMemsetAsync(la.overflow, 0, sizeof(int), stream));
    if (tunables().test_greedy_timeout) {   // (as greedy_relaunch_one_part: the caller must get the REPEAT's output)
        K2_HIP(hipMemsetAsync(la.tokens, 0xEE, sizeof(long long) * (size_t)rec.a.B * la.max_tokens, stream));
        K2_HIP(hipMemsetAsync(la.timestamps, 0xEE, sizeof(int) * (size_t)rec.a.B * la.max_tokens, stream));
        K2_HIP(hipMemsetAsync(la.n_tokens, 0xEE, sizeof(int) * (size_t)rec.a.B, stream));
        if (la.scores) K2_HIP(hipMemsetAsync(la.scores, 0xEE, sizeof(float) * (size_t)rec.a.B, stream));
    }
    hipLaunchKernelGGL(k_beam_loop<1>, dim3(rec.a.B), dim3(GT), rec.beam_lds, stream, rec.w, la);
    K2_HIP(hipGetLastError());
}

}  // namespace k2hip
