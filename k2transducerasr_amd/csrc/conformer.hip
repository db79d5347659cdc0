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
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
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
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
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
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
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
    static const bool force_long = getenv("K2HIP_ATTN_LONG") != nullptr;
    if (T > 64 * SM_PER_LANE || force_long) {
        hipLaunchKernelGGL(k_conformer_softmax_shift_long, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, ctx.stream, ac, bd, rows, T, Tp, NPp);
        K2_HIP(hipGetLastError());
        return;
    }
    hipLaunchKernelGGL(k_conformer_softmax_shift, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, ctx.stream, ac, bd, rows, T, Tp, NPp);
    K2_HIP(hipGetLastError());
}

}  // namespace k2hip
