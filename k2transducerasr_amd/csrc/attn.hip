// RelPositionMultiheadAttentionWeights on gfx950: one kernel from the in_proj
// output to the softmaxed attention weights.
//
//   scores[i,j] = q_i . k_j  +  p_i . pos[T-1-i+j]      (no 1/sqrt(d): folded into the weights)
//   aw[i,:]     = softmax_j(scores[i,:])
//
// One workgroup (4 waves) owns 32 query rows of one (head, batch) pair and ALL T
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

constexpr int QH = 32;  // query/key head dim (config-checked at load)
constexpr int PH = 4;   // pos head dim
constexpr int R = 32;   // query rows per workgroup

__global__ __launch_bounds__(256) void k_attn_scores_softmax(const float* __restrict__ qkp, int ld,
                                                             const float* __restrict__ pp, float* __restrict__ aw, int B,
                                                             int T, int Tp, int H, int lds_stride) {
    extern __shared__ __attribute__((aligned(16))) float S[];  // [R][lds_stride]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int i0 = blockIdx.x * R, b = blockIdx.y, h = blockIdx.z;
    const float* base = qkp + (long long)b * T * ld;
    const int qoff = h * QH, koff = H * QH + h * QH, poff = 2 * H * QH + h * PH;
    const int ppld = H * PH;

    // Q fragments of this workgroup's 32 rows (every wave holds all of them)
    float4 fq[4];
    {
        int row = i0 + li;
#pragma unroll
        for (int g = 0; g < 4; g++) {
            fq[g] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < T) fq[g] = *reinterpret_cast<const float4*>(base + (long long)row * ld + qoff + 8 * g + 4 * lh);
        }
    }
    // p vectors of the 16 accumulator rows of this lane
    float4 pr[16];
#pragma unroll
    for (int r = 0; r < 16; r++) {
        int row = i0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        pr[r] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row < T) pr[r] = *reinterpret_cast<const float4*>(base + (long long)row * ld + poff);
    }

    const int njt = (T + 31) / 32;
    for (int jt = wave; jt < njt; jt += 4) {
        int j = jt * 32 + li;  // this lane's key (as B-operand column and as accumulator column)
        float4 fk[4];
#pragma unroll
        for (int g = 0; g < 4; g++) {
            fk[g] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (j < T) fk[g] = *reinterpret_cast<const float4*>(base + (long long)j * ld + koff + 8 * g + 4 * lh);
        }
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; r++) acc[r] = 0.f;
#pragma unroll
        for (int g = 0; g < 4; g++) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fq[g].x, fk[g].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fq[g].y, fk[g].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fq[g].z, fk[g].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fq[g].w, fk[g].w, acc, 0, 0, 0);
        }
        if (j < T) {
#pragma unroll
            for (int r = 0; r < 16; r++) {
                int rl = (r & 3) + 8 * (r >> 2) + 4 * lh;
                int i = i0 + rl;
                float s = acc[r];
                if (i < T) {
                    // rel-shift in gather form: relative index n = T-1-i+j in [0, 2T-2]
                    float4 e = *reinterpret_cast<const float4*>(pp + (long long)(T - 1 - i + j) * ppld + h * PH);
                    s += pr[r].x * e.x + pr[r].y * e.y + pr[r].z * e.z + pr[r].w * e.w;
                }
                S[rl * lds_stride + j] = s;
            }
        }
    }
    __syncthreads();

    // row softmax: wave w owns rows 8w..8w+7
    float* out = aw + (((long long)h * B + b) * T) * Tp;
    for (int rr = 0; rr < 8; rr++) {
        int rl = wave * 8 + rr, i = i0 + rl;
        if (i >= T) break;
        float* srow = S + rl * lds_stride;
        float mx = -INFINITY;
        for (int j = lane; j < T; j += 64) mx = fmaxf(mx, srow[j]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
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

// Long-sequence form (T beyond the LDS strip, i.e. > 1275 frames at this stack's rate): two passes over the key tiles with
// nothing but the row statistics kept -- pass 1 builds each row's running (max, sum of exp) per lane, combined once across
// lanes and waves; pass 2 recomputes the scores and writes exp(s - max) / sum straight to the weights.  Twice the (small)
// score arithmetic, no strip.
__global__ __launch_bounds__(256) void k_attn_scores_softmax_long(const float* __restrict__ qkp, int ld, const float* __restrict__ pp,
                                                                  float* __restrict__ aw, int B, int T, int Tp, int H) {
    __shared__ float smx[4][R], ssm[4][R];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int i0 = blockIdx.x * R, b = blockIdx.y, h = blockIdx.z;
    const float* base = qkp + (long long)b * T * ld;
    const int qoff = h * QH, koff = H * QH + h * QH, poff = 2 * H * QH + h * PH;
    const int ppld = H * PH;
    float4 fq[4];
    {
        int row = i0 + li;
#pragma unroll
        for (int g = 0; g < 4; g++) {
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
            float4 fk[4];
#pragma unroll
            for (int g = 0; g < 4; g++) {
                fk[g] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (j < T) fk[g] = *reinterpret_cast<const float4*>(base + (long long)j * ld + koff + 8 * g + 4 * lh);
            }
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; r++) acc[r] = 0.f;
#pragma unroll
            for (int g = 0; g < 4; g++) {
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

}  // namespace

void attn_scores_softmax(const Ctx& ctx, const float* qkp, int ld, const float* pp, float* aw, int B, int T, int Tp, int H) {
    K2_REQUIRE(Tp % 4 == 0 && Tp >= T, "attn: Tp=%d must be T=%d rounded up to 4", Tp, T);
    int lds_stride = Tp + 4;  // rows 16 B aligned; +4 floats de-phases the 4-row-apart writers of one MFMA register
    size_t lds = sizeof(float) * R * lds_stride;
    ctx.add_flops(0.0, 2.0 * (QH + PH) * (double)T * T * B * H, 0);
    if (ctx.dry) return;
    static const bool force_long = getenv("K2HIP_ATTN_LONG") != nullptr;
    if (lds > 160 * 1024 || force_long) {  // > 1275 frames: two-pass form without the LDS strip
        hipLaunchKernelGGL(k_attn_scores_softmax_long, dim3(cdiv(T, R), B, H), dim3(256), 0, ctx.stream, qkp, ld, pp, aw, B, T, Tp, H);
        K2_HIP(hipGetLastError());
        return;
    }
    static LdsAttrOnce lds_attr;
    lds_attr.ensure(k_attn_scores_softmax, 160 * 1024);
    dim3 grid(cdiv(T, R), B, H);
    hipLaunchKernelGGL(k_attn_scores_softmax, grid, dim3(256), lds, ctx.stream, qkp, ld, pp, aw, B, T, Tp, H, lds_stride);
    K2_HIP(hipGetLastError());
}

}  // namespace k2hip
