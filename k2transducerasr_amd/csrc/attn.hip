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

}  // namespace

void attn_scores_softmax(const Ctx& ctx, const float* qkp, int ld, const float* pp, float* aw, int B, int T, int Tp, int H) {
    K2_REQUIRE(Tp % 4 == 0 && Tp >= T, "attn: Tp=%d must be T=%d rounded up to 4", Tp, T);
    int lds_stride = Tp + 4;  // rows 16 B aligned; +4 floats de-phases the 4-row-apart writers of one MFMA register
    size_t lds = sizeof(float) * R * lds_stride;
    K2_REQUIRE(lds <= 160 * 1024, "attn: sequence of %d frames needs %zu B of LDS (max 160 KiB)", T, lds);
    ctx.add_flops(0.0, 2.0 * (QH + PH) * (double)T * T * B * H, 0);
    if (ctx.dry) return;
    static bool attr_set = false;
    if (!attr_set) {
        K2_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_attn_scores_softmax),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    dim3 grid(cdiv(T, R), B, H);
    hipLaunchKernelGGL(k_attn_scores_softmax, grid, dim3(256), lds, ctx.stream, qkp, ld, pp, aw, B, T, Tp, H, lds_stride);
    K2_HIP(hipGetLastError());
}

}  // namespace k2hip
