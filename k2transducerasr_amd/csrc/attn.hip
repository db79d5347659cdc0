// RelPositionMultiheadAttentionWeights on gfx950: one kernel from the in_proj
// output to the softmaxed attention weights.
//
//   scores[i,j] = q_i . k_j  +  p_i . pos[T-1-i+j]      (no 1/sqrt(d): folded into the weights)
//   aw[i,:]     = softmax_j(scores[i,:])
//
// One workgroup (8 waves) owns 32 query rows of one (head, batch) pair and ALL T
// keys: q.k^T runs on the f32 MFMA (32x32x2, exact f32 products; K = 32), the
// 4-wide positional term is added on the VALU in the accumulator layout, the
// 32 x T score strip lives in LDS, and the row softmax is done in place before a
// single coalesced write of the weights.  The strip never touches HBM; the
// weights are written once and read three times (non-linear attention and the
// two value paths).
#include "kernels.h"

namespace k2hip {

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int PH = 4;   // pos head dim
constexpr int R = 32;   // query rows per workgroup

// NG = query / key head dim / 8 (4: Zipformer2's 32; 3 and 2: Zipformer v1's attention_dim / heads = 24, 16); koff0 / poff0 = float
// offsets of head 0's key / positional query inside a projected row (the query sits at h * 8 NG)
template <int NG>
__global__ __launch_bounds__(512, 4) void k_attn_scores_softmax(const float* __restrict__ qkp, int ld,
                                                             const float* __restrict__ pp, float* __restrict__ aw, int B,
                                                             int T, int Tp, int H, int lds_stride, int koff0, int poff0) {
    constexpr int QH = 8 * NG;
    extern __shared__ __attribute__((aligned(16))) float S[];  // [R][lds_stride]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int i0 = blockIdx.x * R, b = blockIdx.y, h = blockIdx.z;
    const float* base = qkp + (long long)b * T * ld;
    const int qoff = h * QH, koff = koff0 + h * QH, poff = poff0 + h * PH;
    const int ppld = H * PH;

    // Q fragments of this workgroup's 32 rows (every wave holds all of them)
    float4 fq[NG];
    {
        int row = i0 + li;
#pragma unroll
        for (int g = 0; g < NG; g++) {
            fq[g] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < T) fq[g] = *reinterpret_cast<const float4*>(base + (long long)row * ld + qoff + 8 * g + 4 * lh);
        }
    }
    const int njt = (T + 31) / 32;
    // the next key tile's fragments are requested before this tile's MFMAs and positional adds
    float4 fkn[NG];
    auto load_keys = [&](int jt) {
        if (jt >= njt) return;                 // (wave-uniform)
        const int j = min(jt * 32 + li, T - 1);   // (a key past T reads key T-1: its column is never written)
#pragma unroll
        for (int g = 0; g < NG; g++) fkn[g] = *reinterpret_cast<const float4*>(base + (long long)j * ld + koff + 8 * g + 4 * lh);
    };
    load_keys(wave);   // (requested in front of the window staging: one memory latency for both)
    // the positional rows this workgroup can touch, n = T-1-i+j for its rows i and every key j: one window of <= T+R-1
    // consecutive rows of pp (16 B each for this head) -> LDS, so the per-element gather below is an LDS read
    float4* PW = reinterpret_cast<float4*>(S + R * lds_stride);
    const int nlo = T - 1 - min(i0 + R - 1, T - 1), nwin = (2 * T - 2 - i0) - nlo + 1;
    for (int wdx = tid; wdx < nwin; wdx += 512) PW[wdx] = *reinterpret_cast<const float4*>(pp + (long long)(nlo + wdx) * ppld + h * PH);
    // the positional queries p_i of the 32 rows (read back as LDS broadcasts: all lanes of a half-wave share the row)
    float4* PR = PW + (T + R);
    if (tid < R) PR[tid] = i0 + tid < T ? *reinterpret_cast<const float4*>(base + (long long)(i0 + tid) * ld + poff) : make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();

    for (int jt = wave; jt < njt; jt += 8) {
        int j = jt * 32 + li;  // this lane's key (as B-operand column and as accumulator column)
        float4 fk[NG];
#pragma unroll
        for (int g = 0; g < NG; g++) fk[g] = fkn[g];
        load_keys(jt + 8);
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; r++) acc[r] = 0.f;
#pragma unroll
        for (int g = 0; g < NG; g++) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fq[g].x, fk[g].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fq[g].y, fk[g].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fq[g].z, fk[g].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fq[g].w, fk[g].w, acc, 0, 0, 0);
        }
        if (j < T) {
            // rel-shift in gather form: relative index n = T-1-i+j in [0, 2T-2].  No condition on the row (a row past T reads the window
            // entry of row T-1 and is never used): under `if (i < T)` every one of the 16 window reads sat in its own block behind its
            // own LDS wait.  Eight reads per wait now (round 5: 75.5 -> 73.6 us per launch at T = 505 -- the phase is bound by MFMA +
            // vector issue at two workgroups per CU, not by these round trips; DESIGN "Round 5")
#pragma unroll
            for (int r0 = 0; r0 < 16; r0 += 8) {
                float4 e[8];
#pragma unroll
                for (int r = 0; r < 8; r++) {
                    const int rl = ((r0 + r) & 3) + 8 * ((r0 + r) >> 2) + 4 * lh;
                    e[r] = PW[T - 1 - min(i0 + rl, T - 1) + j - nlo];
                }
#pragma unroll
                for (int r = 0; r < 8; r++) {
                    const int rl = ((r0 + r) & 3) + 8 * ((r0 + r) >> 2) + 4 * lh;
                    const float4 pq = PR[rl];
                    S[rl * lds_stride + j] = acc[r0 + r] + (pq.x * e[r].x + pq.y * e[r].y + pq.z * e[r].z + pq.w * e[r].w);
                }
            }
        }
    }
    __syncthreads();

    // row softmax: wave w (of 8) owns rows 4w..4w+3
    float* out = aw + (((long long)h * B + b) * T) * Tp;
    if (Tp <= 1024) {
        // the row passes through registers once: lane = columns 4 lane + 256 k (float4 LDS reads, float4 stores); max, exp, sum and
        // the scaling happen in between, and all four rows' reads are in flight before the first reduction (the in-place form below
        // walked every row three times through LDS, one dependent read -> exp -> write chain per 64 columns)
        float4 v[4][4];
        const int nk = (Tp + 255) >> 8;  // 256-column chunks in use (wave-uniform)
#pragma unroll
        for (int rr = 0; rr < 4; rr++) {
            const float* srow = S + (wave * 4 + rr) * lds_stride;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int j4 = lane * 4 + 256 * k;
                v[rr][k] = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
                if (k < nk && j4 < T) {  // (the strip is padded to a multiple of 4 columns; columns >= T are masked here)
                    const float4 t = *reinterpret_cast<const float4*>(srow + j4);
                    v[rr][k].x = t.x;
                    if (j4 + 1 < T) v[rr][k].y = t.y;
                    if (j4 + 2 < T) v[rr][k].z = t.z;
                    if (j4 + 3 < T) v[rr][k].w = t.w;
                }
            }
        }
#pragma unroll
        for (int rr = 0; rr < 4; rr++) {
            const int i = i0 + wave * 4 + rr;
            if (i >= T) break;
            float mx = -INFINITY;
#pragma unroll
            for (int k = 0; k < 4; k++)
                if (k < nk) mx = fmaxf(fmaxf(fmaxf(mx, v[rr][k].x), fmaxf(v[rr][k].y, v[rr][k].z)), v[rr][k].w);
            mx = wave_max_dpp(mx);
            float sum = 0.f;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if (k >= nk) break;
                v[rr][k].x = __expf(v[rr][k].x - mx); v[rr][k].y = __expf(v[rr][k].y - mx);
                v[rr][k].z = __expf(v[rr][k].z - mx); v[rr][k].w = __expf(v[rr][k].w - mx);
                sum += (v[rr][k].x + v[rr][k].y) + (v[rr][k].z + v[rr][k].w);
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
            const float inv = 1.0f / sum;
            float* orow = out + (long long)i * Tp;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int j4 = lane * 4 + 256 * k;
                if (k < nk && j4 < Tp) *reinterpret_cast<float4*>(orow + j4) = make_float4(v[rr][k].x * inv, v[rr][k].y * inv, v[rr][k].z * inv, v[rr][k].w * inv);
            }
        }
        return;
    }
    for (int rr = 0; rr < 4; rr++) {
        int rl = wave * 4 + rr, i = i0 + rl;
        if (i >= T) break;
        float* srow = S + rl * lds_stride;
        float mx = -INFINITY;
        for (int j = lane; j < T; j += 64) mx = fmaxf(mx, srow[j]);
        mx = wave_max_dpp(mx);
        float sum = 0.f;
        for (int j = lane; j < T; j += 64) {
            float e = __expf(srow[j] - mx);
            srow[j] = e;
            sum += e;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
        float inv = 1.0f / sum;
        float* orow = out + (long long)i * Tp;
        for (int j4 = lane * 4; j4 < Tp; j4 += 256) {
            float4 v;
            v.x = (j4 + 0 < T) ? srow[j4 + 0] * inv : 0.f;
            v.y = (j4 + 1 < T) ? srow[j4 + 1] * inv : 0.f;
            v.z = (j4 + 2 < T) ? srow[j4 + 2] * inv : 0.f;
            v.w = (j4 + 3 < T) ? srow[j4 + 3] * inv : 0.f;
            *reinterpret_cast<float4*>(orow + j4) = v;
        }
    }
}

// Long-sequence form (T beyond the LDS strip, i.e. > ~1120 frames at this stack's rate): two passes over the key tiles with
// nothing but the row statistics kept -- pass 1 builds each row's running (max, sum of exp) per lane, combined once across
// lanes and waves; pass 2 recomputes the scores and writes exp(s - max) / sum straight to the weights.  Twice the (small)
// score arithmetic, no strip.
template <int NG>
__global__ __launch_bounds__(256) void k_attn_scores_softmax_long(const float* __restrict__ qkp, int ld, const float* __restrict__ pp,
                                                                  float* __restrict__ aw, int B, int T, int Tp, int H, int koff0, int poff0) {
    constexpr int QH = 8 * NG;
    __shared__ float smx[4][R], ssm[4][R];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int i0 = blockIdx.x * R, b = blockIdx.y, h = blockIdx.z;
    const float* base = qkp + (long long)b * T * ld;
    const int qoff = h * QH, koff = koff0 + h * QH, poff = poff0 + h * PH;
    const int ppld = H * PH;
    float4 fq[NG];
    {
        int row = i0 + li;
#pragma unroll
        for (int g = 0; g < NG; g++) {
            fq[g] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < T) fq[g] = *reinterpret_cast<const float4*>(base + (long long)row * ld + qoff + 8 * g + 4 * lh);
        }
    }
    float4 pr[16];
#pragma unroll
    for (int r = 0; r < 16; r++) {
        int row = i0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        pr[r] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row < T) pr[r] = *reinterpret_cast<const float4*>(base + (long long)row * ld + poff);
    }
    const int njt = (T + 31) / 32;
    float rm[16], rs[16];
#pragma unroll
    for (int r = 0; r < 16; r++) { rm[r] = -INFINITY; rs[r] = 0.f; }
    float* out = aw + (((long long)h * B + b) * T) * Tp;
    for (int pass = 0; pass < 2; pass++) {
        for (int jt = wave; jt < njt; jt += 4) {
            const int j = jt * 32 + li;
            float4 fk[NG];
#pragma unroll
            for (int g = 0; g < NG; g++) {
                fk[g] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (j < T) fk[g] = *reinterpret_cast<const float4*>(base + (long long)j * ld + koff + 8 * g + 4 * lh);
            }
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; r++) acc[r] = 0.f;
#pragma unroll
            for (int g = 0; g < NG; g++) {
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fq[g].x, fk[g].x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fq[g].y, fk[g].y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fq[g].z, fk[g].z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fq[g].w, fk[g].w, acc, 0, 0, 0);
            }
            if (j < T) {
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int i = i0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (i >= T) continue;
                    const float4 e = *reinterpret_cast<const float4*>(pp + (long long)(T - 1 - i + j) * ppld + h * PH);
                    const float sc = acc[r] + pr[r].x * e.x + pr[r].y * e.y + pr[r].z * e.z + pr[r].w * e.w;
                    if (pass == 0) {
                        const float m = fmaxf(rm[r], sc);
                        rs[r] = rs[r] * __expf(rm[r] - m) + __expf(sc - m);
                        rm[r] = m;
                    } else {
                        out[(long long)i * Tp + j] = __expf(sc - rm[r]) * rs[r];  // rs holds 1 / sum in pass 2
                    }
                }
            }
        }
        if (pass == 0) {
            // combine the per-lane statistics of a row: across the 32 lanes of the half-wave, then across the 4 waves
#pragma unroll
            for (int r = 0; r < 16; r++) {
                float m = rm[r], sv = rs[r];
#pragma unroll
                for (int o = 16; o > 0; o >>= 1) {
                    const float om = __shfl_xor(m, o), os = __shfl_xor(sv, o);
                    const float nm = fmaxf(m, om);
                    sv = (m == -INFINITY ? 0.f : sv * __expf(m - nm)) + (om == -INFINITY ? 0.f : os * __expf(om - nm));
                    m = nm;
                }
                if (li == 0) {
                    const int rl = (r & 3) + 8 * (r >> 2) + 4 * lh;
                    smx[wave][rl] = m;
                    ssm[wave][rl] = sv;
                }
            }
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int rl = (r & 3) + 8 * (r >> 2) + 4 * lh;
                float m = -INFINITY;
                for (int w = 0; w < 4; w++) m = fmaxf(m, smx[w][rl]);
                float sv = 0.f;
                for (int w = 0; w < 4; w++)
                    if (smx[w][rl] != -INFINITY) sv += ssm[w][rl] * __expf(smx[w][rl] - m);
                rm[r] = m;
                rs[r] = 1.0f / sv;
            }
        }
    }
    // pad columns [T, Tp) of this workgroup's rows
    for (int idx = tid; idx < R * (Tp - T); idx += 256) {
        const int rl = idx / (Tp - T), c = T + idx % (Tp - T);
        if (i0 + rl < T) out[(long long)(i0 + rl) * Tp + c] = 0.f;
    }
}


// ---------------------------------------------------------------------------------------
// SelfAttention.forward after its value projection, fused:  x += out_proj(concat_h(aw_h . v_h)) + bias.
//   aw [H][B][T][Tp] softmax weights over KL keys (pad columns zero), v [B*KL, HV] (HV = H*vh, head h = columns h*vh ..),
//   Wout [D, HV] (torch Linear layout), x [B*T, D] updated in place.  Offline KL = T; streaming KL = left context + chunk.
// One workgroup = 16 rows of one stream.  Phase A: the stream's values sit in LDS ([Tk][HV+2], zero rows past T); wave w takes
// heads w, w+4, ...: 16x16x4 f32 MFMAs over the keys, A = a 16-row strip of aw read straight from HBM (two float4 per lane per
// 32 keys -- the launch is bound by that read: every attention-weight element is fetched exactly once), B = the head's values
// (vh <= 16 columns of the 16-wide tile).  Phase B: the 16 x HV context rows go through LDS.  Phase C: the waves split the D/16
// column tiles of out_proj (K = HV, the operand Wout comes from L2), add bias and residual, store.
// Replaces a batched N=12 GEMM (13 us .. 56 us at 4 .. 14 TFLOP/s) plus a K=48 GEMM (15 us) per use.
// ---------------------------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));

// RING: streaming form -- the stream's values live in a ring of KL rows inside its slot of the state pool (RingRef, kernels.h), aw's
// columns are in ring order, and the chunk's T new value rows (newrows [B*T, HV]) are written into the ring by this workgroup before
// it reads them (every column slice of a stream writes the same bytes; nobody reads those rows before its own copy is stored).
// PROJ (with RING): the value projection itself runs here too -- newrows = xin[b rows] . win^T + bin, the chunk's <= 16 rows of one
// stream against [HV, D] weights: (HV / 16) x (D / 4) 16x16x4 MFMAs split over the four waves (~1.5 - 4 us) instead of a launch of its
// own for 256 .. 2048 rows x 48 / 96 columns (8 - 15 us, 32 of them per streaming tick).  Every column slice of a stream computes the same
// rows (same code, same bits) and writes them into the ring as before.  The projection reads the WHOLE input rows while other slices of
// the stream already store their output columns, so input and output are different buffers in this form (xin: projection + residual;
// x: written).
template <bool RING, bool PROJ = false>
__global__ __launch_bounds__(256) void k_attn_av_out(const float* __restrict__ aw, const float* __restrict__ v,
                                                     const float* __restrict__ wout, const float* __restrict__ bias,
                                                     float* __restrict__ x, int B, int T, int KL, int Tp, int Tk, int H, int vh,
                                                     int D, int tiles_per_z, RingRef ring, const float* __restrict__ newrows,
                                                     const float* __restrict__ xin = nullptr, const float* __restrict__ win = nullptr,
                                                     const float* __restrict__ bin = nullptr) {
    extern __shared__ float avs[];  // [16][HV + 1]
    const int HV = H * vh, AS = HV + 1;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.y, i0 = blockIdx.x * 16;
    const int n = lane & 15, kq = lane >> 4;
    const float* xres = PROJ ? xin : x;   // the residual operand of phase C
    if (RING) {
        float* rb = ring.pool + (long long)ring.slots[b] * ring.slot_stride + ring.off;
        const int head = (int)(((long long)ring.chunks[b] * T) % KL), h4 = HV >> 2;
        if (PROJ) {
            // the stream's T input rows -> LDS (behind the context tile, rows of D + 4 floats: conflict-free ds_read_b128 for 16 rows x 4
            // k groups), one coalesced pass of the whole workgroup; rows past T are zero
            float* xs = avs + 16 * AS + ((4 - ((16 * AS) & 3)) & 3);
            const int XS = D + 4, d4 = D >> 2;
            for (int e = tid; e < 16 * d4; e += 256) {
                const int r = e / d4, c = (e % d4) * 4;
                *reinterpret_cast<float4*>(xs + r * XS + c) =
                    r < T ? *reinterpret_cast<const float4*>(xin + ((long long)b * T + r) * D + c) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
            __syncthreads();
            // A: lane (row n, k group kq) holds x[row][k0 + 4 kq ..] (from LDS); B: lane (column n, kq) holds win[column][k0 + 4 kq ..]
            // (from L2, four 16-deep steps requested at a time); MFMA c of a step takes component c of both (every k once); C: column =
            // lane & 15, rows 4 kq + e
            const float* xr = xs + n * XS + 4 * kq;
            for (int ct = wave; 16 * ct < HV; ct += 4) {
                const int col = 16 * ct + n;
                const float* wr = win + (long long)min(col, HV - 1) * D + 4 * kq;
                f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
                for (int k0 = 0; k0 < D; k0 += 64) {   // D % 32 == 0: the last round may hold two steps
                    float4 w4[4], a4[4];
                    const int ns = min(4, (D - k0) >> 4);
#pragma unroll
                    for (int q = 0; q < 4; q++) w4[q] = q < ns ? *reinterpret_cast<const float4*>(wr + k0 + 16 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                    for (int q = 0; q < 4; q++) a4[q] = q < ns ? *reinterpret_cast<const float4*>(xr + k0 + 16 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                    for (int q = 0; q < 4; q += 2) {
                        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[q].x, w4[q].x, acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[q + 1].x, w4[q + 1].x, acc1, 0, 0, 0);
                        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[q].y, w4[q].y, acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[q + 1].y, w4[q + 1].y, acc1, 0, 0, 0);
                        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[q].z, w4[q].z, acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[q + 1].z, w4[q + 1].z, acc1, 0, 0, 0);
                        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[q].w, w4[q].w, acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[q + 1].w, w4[q + 1].w, acc1, 0, 0, 0);
                    }
                }
                if (col < HV) {
                    const float bc = bin[col];
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const int r = 4 * kq + e;
                        if (r < T) rb[(long long)((head + r) % KL) * HV + col] = (acc0[e] + acc1[e]) + bc;
                    }
                }
            }
        } else {
            for (int e = tid; e < T * h4; e += 256) {
                const int r = e / h4, c = (e % h4) * 4;
                *reinterpret_cast<float4*>(rb + (long long)((head + r) % KL) * HV + c) =
                    *reinterpret_cast<const float4*>(newrows + ((long long)b * T + r) * HV + c);
            }
        }
        __syncthreads();
        v = rb;  // this stream's ring: rows 0 .. KL-1
    } else {
        v += (long long)b * KL * HV;
    }
    // ---- phase A: 64 keys per step = 4 groups of 16; lane (n, kq) holds keys k0 + 16 g + 4 kq + {0..3} of row i0 + n (A operand,
    // one float4 of aw) and of value column h*vh + n (B operand, four scalar loads: the stream's values, T x HV floats, stay in L2).
    const int row_a = min(i0 + n, T - 1);  // clamped rows are computed and dropped
    const bool ncol = n < vh;
    for (int h = wave; h < H; h += 4) {
        const float* arow = aw + (((long long)h * B + b) * T + row_a) * Tp + 4 * kq;
        const float* vb = v + (long long)(4 * kq) * HV + h * vh + (ncol ? n : 0);
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        float4 a4[4], a4n[4];
        float bv[4][4], bvn[4][4];
        auto load_step = [&](float4* ad, float (*bd)[4], int k0) {
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const int kb = k0 + 16 * g + 4 * kq;
                ad[g] = kb < Tp ? *reinterpret_cast<const float4*>(arow + k0 + 16 * g) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int m = 0; m < 4; m++) bd[g][m] = (ncol && kb + m < KL) ? vb[(long long)(k0 + 16 * g + m) * HV] : 0.f;
            }
        };
        load_step(a4, bv, 0);
        for (int k0 = 0; k0 < Tk; k0 += 64) {
            if (k0 + 64 < Tk) load_step(a4n, bvn, k0 + 64);  // in flight under this step's MFMAs
#pragma unroll
            for (int g = 0; g < 4; g++) {
                if (g & 1) {
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[g].x, bv[g][0], acc1, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[g].y, bv[g][1], acc1, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[g].z, bv[g][2], acc1, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[g].w, bv[g][3], acc1, 0, 0, 0);
                } else {
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[g].x, bv[g][0], acc0, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[g].y, bv[g][1], acc0, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[g].z, bv[g][2], acc0, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[g].w, bv[g][3], acc0, 0, 0, 0);
                }
            }
#pragma unroll
            for (int g = 0; g < 4; g++) {
                a4[g] = a4n[g];
#pragma unroll
                for (int m = 0; m < 4; m++) bv[g][m] = bvn[g][m];
            }
        }
        if (ncol) {  // C layout: col = lane & 15, row = 4 * (lane >> 4) + e
#pragma unroll
            for (int e = 0; e < 4; e++) avs[(4 * kq + e) * AS + h * vh + n] = acc0[e] + acc1[e];
        }
    }
    __syncthreads();
    // ---- phase C: out[16, D] = avs[16, HV] . wout^T; the A operand (this lane's context values) is the same for every column tile.
    // blockIdx.z splits the column tiles when there are too few row strips to fill the chip (phase A is then repeated per slice).
    constexpr int MAXC = 8;  // HV <= 128
    const int nch = (HV + 15) >> 4;
    float4 av4[MAXC];
#pragma unroll
    for (int c = 0; c < MAXC; c++) {
        const int kb = 16 * c + 4 * kq;
        const float* ar = avs + n * AS + kb;
        av4[c] = (c < nch && kb < HV) ? make_float4(ar[0], ar[1], ar[2], ar[3]) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const int ct0 = blockIdx.z * tiles_per_z, ct1 = min(D >> 4, ct0 + tiles_per_z);
    float4 wc[MAXC], wn[MAXC];
    auto load_w = [&](float4* dst, int ct) {
        const float* wr = wout + (long long)(ct * 16 + n) * HV + 4 * kq;
#pragma unroll
        for (int c = 0; c < MAXC; c++)
            dst[c] = (c < nch && 16 * c + 4 * kq < HV) ? *reinterpret_cast<const float4*>(wr + 16 * c) : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    // Two register sets, no copies: a tile's weights, residual and bias are loaded one tile ahead, i.e. IN FRONT of the previous
    // tile's stores (vmcnt retires in issue order: a load issued behind those stores would make its wait drain them too, and the
    // tiles of a wave would run a store round trip apart).
    float xa[4], xb[4], ba = 0.f, bb = 0.f;
    auto load_x = [&](float* dst, float& bdst, int ct) {
        const int col = ct * 16 + n;
        bdst = bias[col];
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const int i = i0 + 4 * kq + e;
            dst[e] = i < T ? xres[((long long)b * T + i) * D + col] : 0.f;
        }
    };
    auto tile = [&](const float4* w, const float* xr, float bv, int ct) {
        const int col = ct * 16 + n;
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < MAXC; c++) {
            if (c < nch) {
                if (c & 1) {
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av4[c].x, w[c].x, acc1, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av4[c].y, w[c].y, acc1, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av4[c].z, w[c].z, acc1, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av4[c].w, w[c].w, acc1, 0, 0, 0);
                } else {
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av4[c].x, w[c].x, acc0, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av4[c].y, w[c].y, acc0, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av4[c].z, w[c].z, acc0, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av4[c].w, w[c].w, acc0, 0, 0, 0);
                }
            }
        }
        float o[4];  // all four values first, then four stores with nothing loaded in between (else: a vmcnt(0) per store)
#pragma unroll
        for (int e = 0; e < 4; e++) o[e] = acc0[e] + acc1[e] + bv + xr[e];
        asm volatile("" : "+v"(o[0]), "+v"(o[1]), "+v"(o[2]), "+v"(o[3]));  // (keeps the sums out of the conditional store blocks)
        float* xp = x + ((long long)b * T + i0 + 4 * kq) * D + col;
#pragma unroll
        for (int e = 0; e < 4; e++)
            if (i0 + 4 * kq + e < T) xp[(long long)e * D] = o[e];
    };
    int ct = ct0 + wave;
    if (ct < ct1) {
        load_w(wc, ct);
        load_x(xa, ba, ct);
    }
    while (ct < ct1) {
        if (ct + 4 < ct1) {
            load_w(wn, ct + 4);
            load_x(xb, bb, ct + 4);
        }
        tile(wc, xa, ba, ct);
        ct += 4;
        if (ct >= ct1) break;
        if (ct + 4 < ct1) {
            load_w(wc, ct + 4);
            load_x(xa, ba, ct + 4);
        }
        tile(wn, xb, bb, ct);
        ct += 4;
    }
}


// ---- NonlinAttention.streaming_forward behind its in_proj, one launch (k_ring_put + a batched [T, KL] x [KL, Hc] GEMM + the out_proj
// GEMM before: three launches per layer).  One workgroup per (stream, slice of out_proj's column tiles):
//   0. the chunk's rows x * tanh(s) go into this stream's ring rows (hid row = s | x | y, Hc columns each);
//   A. ctx[T, Hc] = aw_head0[T, KL] . ring[KL, Hc] (ring order on both sides), times the gate y, into LDS -- 16x16x4 MFMAs, a wave per
//      16-column tile, the weight row of a lane as float4 (keys contiguous), the ring values as scalars (L2);
//   C. x[T, D] += ctx . wout^T + bias: the column tiles of this slice over the four waves, K = Hc walked in 64-deep trips.
// Every slice of a stream repeats 0 and A (identical values: benign), as in k_attn_av_out.
// MULTI: more than NS steps of keys (KL > 256, e.g. left_context_len 256 + the chunk): trips of NS steps each, a trip's loads issued
// together, the accumulators carried across trips
template <int NS, int NT, bool MULTI = false>  // NS = 64-key steps (all of a tile's loads are issued before its first MFMA), NT = threads
__global__ __launch_bounds__(NT) void k_nonlin_av_out(const float* __restrict__ aw, const float* __restrict__ hid, int ldh,
                                                      const float* __restrict__ wout, const float* __restrict__ bias,
                                                      float* __restrict__ x, int B, int T, int KL, int Tp, int Hc, int D,
                                                      int tiles_per_z, RingRef ring) {
    extern __shared__ float cs_[];  // [16][Hc rounded up to 16, + 4]; the pad columns hold zeros
    constexpr int NWV = NT / 64;
    const int CS = ((Hc + 15) & ~15) + 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.y;
    const int n = lane & 15, kq = lane >> 4;
    float* rb = ring.pool + (long long)ring.slots[b] * ring.slot_stride + ring.off;
    {
        const int head = (int)(((long long)ring.chunks[b] * T) % KL), h4 = Hc >> 2;
        for (int e = tid; e < T * h4; e += NT) {
            const int r = e / h4, c = (e % h4) * 4;
            const float* p = hid + ((long long)b * T + r) * ldh + c;
            const float4 sg = *reinterpret_cast<const float4*>(p), v = *reinterpret_cast<const float4*>(p + Hc);
            *reinterpret_cast<float4*>(rb + (long long)((head + r) % KL) * Hc + c) =
                make_float4(v.x * tanhf(sg.x), v.y * tanhf(sg.y), v.z * tanhf(sg.z), v.w * tanhf(sg.w));
        }
    }
    __syncthreads();
    // ---- phase A: a tile is one latency deep -- the ring rows of a stream were written chunks ago and come from HBM / MALL (~1 us a
    // round trip): with the loads of one 64-key step at a time and four waves the kernel was a chain of 18 such round trips (15 us)
    const int row_a = min(n, T - 1);  // clamped rows are computed and dropped
    const float* arow = aw + ((long long)b * T + row_a) * Tp + 4 * kq;   // head 0: aw[0][b][row][.]
    for (int ct = wave; ct * 16 < Hc; ct += NWV) {
        const bool cok = ct * 16 + n < Hc;  // (Hc % 16 != 0: the last tile is partial)
        const float* vb = rb + (long long)(4 * kq) * Hc + (cok ? ct * 16 + n : 0);
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        const int trips = MULTI ? (KL + 64 * NS - 1) / (64 * NS) : 1;
        for (int tr = 0; tr < trips; tr++) {
        const int k0 = 64 * NS * tr;
        float4 a4[NS][4];
        float bv[NS][4][4];
#pragma unroll
        for (int s = 0; s < NS; s++)
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const int kb = k0 + 64 * s + 16 * g + 4 * kq;
                a4[s][g] = kb < Tp ? *reinterpret_cast<const float4*>(arow + k0 + 64 * s + 16 * g) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int m = 0; m < 4; m++) bv[s][g][m] = (cok && kb + m < KL) ? vb[(long long)(k0 + 64 * s + 16 * g + m) * Hc] : 0.f;
            }
#pragma unroll
        for (int s = 0; s < NS; s++)
#pragma unroll
            for (int g = 0; g < 4; g++) {
                if (g & 1) {
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[s][g].x, bv[s][g][0], acc1, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[s][g].y, bv[s][g][1], acc1, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[s][g].z, bv[s][g][2], acc1, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[s][g].w, bv[s][g][3], acc1, 0, 0, 0);
                } else {
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[s][g].x, bv[s][g][0], acc0, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[s][g].y, bv[s][g][1], acc0, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[s][g].z, bv[s][g][2], acc0, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[s][g].w, bv[s][g][3], acc0, 0, 0, 0);
                }
            }
        }
        // C layout: col = lane & 15, row = 4 * (lane >> 4) + e; the gate y of (row, col) multiplies here; rows >= T hold zeros
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const int r = 4 * kq + e;
            const float gate = (cok && r < T) ? hid[((long long)b * T + r) * ldh + 2 * Hc + ct * 16 + n] : 0.f;
            cs_[r * CS + ct * 16 + n] = (acc0[e] + acc1[e]) * gate;
        }
    }
    __syncthreads();
    if (!wout) {  // the gated context rows themselves: x is [B*T, Hc] here and out_proj runs as a GEMM over all streams' rows
        const int h4 = Hc >> 2;
        for (int e = tid; e < T * h4; e += NT) {
            const int r = e / h4, c = (e % h4) * 4;
            *reinterpret_cast<float4*>(x + ((long long)b * T + r) * Hc + c) = *reinterpret_cast<const float4*>(cs_ + r * CS + c);
        }
        return;
    }
    // ---- phase C
    const int ct0 = blockIdx.z * tiles_per_z, ct1 = min(D >> 4, ct0 + tiles_per_z);
    const float* ar = cs_ + n * CS + 4 * kq;
    for (int ct = ct0 + wave; ct < ct1; ct += NWV) {
        const int col = ct * 16 + n;
        const float* wr = wout + (long long)col * Hc + 4 * kq;
        const float bvv = bias[col];
        float xr[4];
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const int i = 4 * kq + e;
            xr[e] = i < T ? x[((long long)b * T + i) * D + col] : 0.f;
        }
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        for (int c0 = 0; c0 < Hc; c0 += 64) {  // Hc % 4 == 0: a lane's float4 is wholly inside or outside the row
            float4 w4[4], a4[4];
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const bool ok = c0 + 16 * g + 4 * kq < Hc;
                w4[g] = ok ? *reinterpret_cast<const float4*>(wr + c0 + 16 * g) : make_float4(0.f, 0.f, 0.f, 0.f);
                a4[g] = ok ? *reinterpret_cast<const float4*>(ar + c0 + 16 * g) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int g = 0; g < 4; g++) {
                if (g & 1) {
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[g].x, w4[g].x, acc1, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[g].y, w4[g].y, acc1, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[g].z, w4[g].z, acc1, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[g].w, w4[g].w, acc1, 0, 0, 0);
                } else {
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[g].x, w4[g].x, acc0, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[g].y, w4[g].y, acc0, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[g].z, w4[g].z, acc0, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[g].w, w4[g].w, acc0, 0, 0, 0);
                }
            }
        }
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; e++) o[e] = acc0[e] + acc1[e] + bvv + xr[e];
        float* xp = x + ((long long)b * T + 4 * kq) * D + col;
#pragma unroll
        for (int e = 0; e < 4; e++)
            if (4 * kq + e < T) xp[(long long)e * D] = o[e];
    }
}

}  // namespace

template <int NG>
static void attn_scores_launch(const Ctx& ctx, const float* qkp, int ld, const float* pp, float* aw, int B, int T, int Tp, int H, int koff0,
                               int poff0) {
    int lds_stride = Tp + 4;  // rows 16 B aligned; +4 floats de-phases the 4-row-apart writers of one MFMA register
    size_t lds = sizeof(float) * (R * lds_stride + 4 * (T + R) + 4 * R);  // score strip + the positional window + the rows' p
    const bool force_long = tunables().attn_long != 0;
    if (lds > 160 * 1024 || force_long) {  // > ~1120 frames: two-pass form without the LDS strip
        hipLaunchKernelGGL(k_attn_scores_softmax_long<NG>, dim3(cdiv(T, R), B, H), dim3(256), 0, ctx.stream, qkp, ld, pp, aw, B, T, Tp, H, koff0,
                           poff0);
        K2_HIP(hipGetLastError());
        return;
    }
    static LdsAttrOnce lds_attr;
    lds_attr.ensure(k_attn_scores_softmax<NG>, 160 * 1024);
    dim3 grid(cdiv(T, R), B, H);
    hipLaunchKernelGGL(k_attn_scores_softmax<NG>, grid, dim3(512), lds, ctx.stream, qkp, ld, pp, aw, B, T, Tp, H, lds_stride, koff0, poff0);
    K2_HIP(hipGetLastError());
}

void attn_scores_softmax(const Ctx& ctx, const float* qkp, int ld, const float* pp, float* aw, int B, int T, int Tp, int H, int qh, int koff0,
                         int poff0) {
    K2_REQUIRE(Tp % 4 == 0 && Tp >= T, "attn: Tp=%d must be T=%d rounded up to 4", Tp, T);
    if (koff0 < 0) koff0 = H * qh;       // Zipformer2 row: q [H*qh] | k [H*qh] | p [H*4]
    if (poff0 < 0) poff0 = 2 * H * qh;
    K2_REQUIRE(koff0 % 4 == 0 && poff0 % 4 == 0 && ld % 4 == 0, "attn: operand offsets must be multiples of 4 floats");
    ctx.add_flops(0.0, 2.0 * (qh + PH) * (double)T * T * B * H, 0);
    if (ctx.dry) return;
    switch (qh) {
        case 32: attn_scores_launch<4>(ctx, qkp, ld, pp, aw, B, T, Tp, H, koff0, poff0); break;
        case 24: attn_scores_launch<3>(ctx, qkp, ld, pp, aw, B, T, Tp, H, koff0, poff0); break;
        case 16: attn_scores_launch<2>(ctx, qkp, ld, pp, aw, B, T, Tp, H, koff0, poff0); break;
        default: failf(K2HIP_ERR_UNSUPPORTED, "attention head size %d unsupported (16, 24, 32)", qh);
    }
}

bool attn_av_out(const Ctx& ctx, const float* aw, const float* v, const float* wout, const float* bias, float* x, int B, int T, int KL, int Tp,
                 int H, int vh, int D) {
    const int HV = H * vh, Tk = (KL + 63) & ~63;
    const bool off = tunables().no_fused_av != 0;  // the tests compare both paths in one process (k2hip_debug_set_switch)
    if (off || vh > 16 || HV % 4 != 0 || HV > 128 || D % 16 != 0 || Tp % 4 != 0 || Tp < KL) return false;
    ctx.add_flops(0.0, 2.0 * B * (double)T * HV * (KL + D), 0);
    if (ctx.dry) return true;
    // few row strips (short sequences): split out_proj's column tiles over blockIdx.z so that ~2 workgroups per CU exist
    const int strips = cdiv(T, 16) * B, ntile = D / 16;
    int cs = std::max(1, std::min(4, 512 / std::max(1, strips)));
    cs = std::min(cs, ntile);
    const int tiles_per_z = cdiv(ntile, cs);
    cs = cdiv(ntile, tiles_per_z);
    const size_t lds = sizeof(float) * 16 * (HV + 1);
    hipLaunchKernelGGL(k_attn_av_out<false>, dim3(cdiv(T, 16), B, cs), dim3(256), lds, ctx.stream, aw, v, wout, bias, x, B, T, KL, Tp, Tk, H, vh,
                       D, tiles_per_z, RingRef(), nullptr);
    K2_HIP(hipGetLastError());
    return true;
}

bool attn_av_out_ring(const Ctx& ctx, const float* aw, const RingRef& vals, const float* newrows, const float* wout, const float* bias, float* x,
                      int B, int T, int KL, int Tp, int H, int vh, int D) {
    const int HV = H * vh, Tk = (KL + 63) & ~63;
    // (no GEMM fallback in the ring form: the shapes of the streaming recipe always fit; anything else is refused)
    K2_REQUIRE(vh <= 16 && HV % 4 == 0 && HV <= 128 && D % 16 == 0 && Tp % 4 == 0 && Tp >= KL && T <= 16,
               "attn_av_out_ring: shape H=%d vh=%d D=%d T=%d KL=%d unsupported", H, vh, D, T, KL);
    ctx.add_flops(0.0, 2.0 * B * (double)T * HV * (KL + D), 0);
    if (ctx.dry) return true;
    const int strips = B, ntile = D / 16;
    int cs = std::max(1, std::min(4, 512 / std::max(1, strips)));
    cs = std::min(cs, ntile);
    const int tiles_per_z = cdiv(ntile, cs);
    cs = cdiv(ntile, tiles_per_z);
    const size_t lds = sizeof(float) * 16 * (HV + 1);
    hipLaunchKernelGGL(k_attn_av_out<true>, dim3(1, B, cs), dim3(256), lds, ctx.stream, aw, nullptr, wout, bias, x, B, T, KL, Tp, Tk, H, vh, D,
                       tiles_per_z, vals, newrows);
    K2_HIP(hipGetLastError());
    return true;
}

// the streaming self-attention module behind the attention weights, value projection included (k_attn_av_out<true, true>):
// xout = xin + out_proj(aw . [ring ; xin . win^T + bin]); xin and xout must be different buffers
void attn_proj_av_out_ring(const Ctx& ctx, const float* aw, const RingRef& vals, const float* xin, const float* win, const float* bin,
                           const float* wout, const float* bias, float* xout, int B, int T, int KL, int Tp, int H, int vh, int D) {
    const int HV = H * vh, Tk = (KL + 63) & ~63;
    K2_REQUIRE(vh <= 16 && HV % 4 == 0 && HV <= 128 && D % 32 == 0 && Tp % 4 == 0 && Tp >= KL && T <= 16,
               "attn_proj_av_out_ring: shape H=%d vh=%d D=%d T=%d KL=%d unsupported", H, vh, D, T, KL);
    K2_REQUIRE(ctx.dry || xin != xout, "attn_proj_av_out_ring: input and output must be different buffers");
    ctx.add_flops(0.0, 2.0 * B * (double)T * HV * (KL + D) + 2.0 * B * (double)T * HV * D, 0);
    if (ctx.dry) return;
    const int strips = B, ntile = D / 16;
    int cs = std::max(1, std::min(4, 512 / std::max(1, strips)));
    cs = std::min(cs, ntile);
    const int tiles_per_z = cdiv(ntile, cs);
    cs = cdiv(ntile, tiles_per_z);
    const size_t lds = sizeof(float) * (16 * (HV + 1) + 4 + 16 * (D + 4));   // context tile | the stream's input rows
    K2_REQUIRE(lds <= 64 * 1024, "attn_proj_av_out_ring: D=%d too wide", D);
    hipLaunchKernelGGL((k_attn_av_out<true, true>), dim3(1, B, cs), dim3(256), lds, ctx.stream, aw, nullptr, wout, bias, xout, B, T, KL, Tp, Tk, H, vh,
                       D, tiles_per_z, vals, nullptr, xin, win, bin);
    K2_HIP(hipGetLastError());
}

void nonlin_av_out_ring(const Ctx& ctx, const float* aw, const RingRef& cache, const float* hid, int ldh, const float* wout, const float* bias,
                        float* x, int B, int T, int KL, int Tp, int Hc, int D) {
    K2_REQUIRE(Hc % 4 == 0 && D % 16 == 0 && Tp % 4 == 0 && Tp >= KL && T <= 16 && ldh % 4 == 0 && ldh >= 3 * Hc,
               "nonlin_av_out_ring: shape Hc=%d D=%d T=%d KL=%d unsupported", Hc, D, T, KL);
    ctx.add_flops(0.0, 2.0 * B * (double)T * Hc * (KL + D), 0);
    // (every shape check sits in front of the dry return: a call sizes its arena with a dry pass before its first launch, so a shape
    // the kernel cannot take is refused before any layer has advanced a cache in place)
    const size_t lds = sizeof(float) * 16 * (((Hc + 15) & ~15) + 4);
    K2_REQUIRE(lds <= 64 * 1024, "nonlin_av_out_ring: Hc=%d too wide", Hc);
    const int ns = (KL + 63) / 64;
    K2_REQUIRE(ns >= 1, "nonlin_av_out_ring: no keys");
    if (ctx.dry) return;
    const int ntile = D / 16;
    int cs = std::max(1, std::min(4, 512 / std::max(1, B)));
    cs = std::min(cs, ntile);
    if (!wout) cs = 1;  // context rows only: nothing to slice
    const int tiles_per_z = cdiv(ntile, cs);
    cs = cdiv(ntile, tiles_per_z);
    // 16-column tiles of the context over the waves: 16 waves when there are that many tiles (the 50 / 25 Hz stacks), else 8
    const bool wide = (Hc + 15) / 16 > 8;
#define K2_NONLIN_LAUNCH(NS_, NT_)                                                                                                   \
    hipLaunchKernelGGL((k_nonlin_av_out<NS_, NT_>), dim3(1, B, cs), dim3(NT_), lds, ctx.stream, aw, hid, ldh, wout, bias, x, B, T, KL, Tp, Hc, \
                       D, tiles_per_z, cache)
    switch (ns) {
        case 1: if (wide) K2_NONLIN_LAUNCH(1, 1024); else K2_NONLIN_LAUNCH(1, 512); break;
        case 2: if (wide) K2_NONLIN_LAUNCH(2, 1024); else K2_NONLIN_LAUNCH(2, 512); break;
        case 3: if (wide) K2_NONLIN_LAUNCH(3, 1024); else K2_NONLIN_LAUNCH(3, 512); break;
        case 4: K2_NONLIN_LAUNCH(4, 512); break;  // (193 .. 256 keys: 16 waves would spill)
        default:                                   // > 256 keys (left_context_len >= 256): trips of 4 steps
            hipLaunchKernelGGL((k_nonlin_av_out<4, 512, true>), dim3(1, B, cs), dim3(512), lds, ctx.stream, aw, hid, ldh, wout, bias, x, B, T, KL,
                               Tp, Hc, D, tiles_per_z, cache);
            break;
    }
#undef K2_NONLIN_LAUNCH
    K2_HIP(hipGetLastError());
}

}  // namespace k2hip
