// Small kernels of the LSTM transducer path (Model_type "lstm": OfflineProjOfTransducer offline, OnlineProjOfLstm streaming).
// The matrix work (input projections for all frames, the per-frame recurrent and projection products, feed-forward) runs on
// the MFMA GEMMs; these are the cell non-linearity and the state gather / scatter of the streaming pool.
#include "kernels.h"

namespace k2hip {
namespace {

__device__ __forceinline__ float sigm(float s) { return 1.0f / (1.0f + expf(-s)); }

// torch.nn.LSTM gate order i, f, g, o
__global__ void k_lstm_cell(const float* __restrict__ gx, long long ldgx, const float* __restrict__ gh, int ldgh, float* __restrict__ c,
                            float* __restrict__ hf, int B, int Hh) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * Hh) return;
    const int b = i / Hh, j = i - b * Hh;
    const float* x = gx + (long long)b * ldgx;
    const float* h = gh + (long long)b * ldgh;
    const float ig = sigm(x[j] + h[j]);
    const float fg = sigm(x[Hh + j] + h[Hh + j]);
    const float gg = tanhf(x[2 * Hh + j] + h[2 * Hh + j]);
    const float og = sigm(x[3 * Hh + j] + h[3 * Hh + j]);
    const float cn = fg * c[i] + ig * gg;
    c[i] = cn;
    hf[i] = og * tanhf(cn);
}

// wavefront form: gates already hold x.W_ih + b_ih + h.W_hh + b_hh
__global__ void k_lstm_cell_rows(const float* __restrict__ gates, float* __restrict__ c, float* __restrict__ hf, int rows, int Hh) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * Hh) return;
    const int r = i / Hh, j = i - r * Hh;
    const float* g = gates + (long long)r * 4 * Hh;
    const float ig = sigm(g[j]), fg = sigm(g[Hh + j]), gg = tanhf(g[2 * Hh + j]), og = sigm(g[3 * Hh + j]);
    const float cn = fg * c[i] + ig * gg;
    c[i] = cn;
    hf[i] = og * tanhf(cn);
}
// h[z][b] = sum of the S split-K partials hp[s][z][b] of the projection; x1 = x_t + h
__global__ void k_lstm_add_frame(const float* __restrict__ Y, long long SY, const float* __restrict__ hp, long long pstride, int S,
                                 float* __restrict__ h, float* __restrict__ x1, int n, int B, int T, int D, int lo, int s) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * B * D) return;
    const int d = i % D, r = i / D, b = r % B, z = r / B;
    const int t = s - lo - z;
    float hv = hp[i];
    for (int q = 1; q < S; q++) hv += hp[(long long)q * pstride + i];
    h[i] = hv;
    x1[i] = Y[(long long)(lo + z) * SY + ((long long)b * T + t) * D + d] + hv;
}
// one wave per row: x2 = x1 + b2 + sum of the S split-K partials of feed_forward.4; Y[l+1][frame] = BasicNorm(x2).  float4 per lane,
// every load of the row issued before the first use (D <= 1024: at most 4 float4 per lane, S <= 4 partials)
__global__ __launch_bounds__(256) void k_lstm_norm_frame(const float* __restrict__ x1, const float* __restrict__ fp, long long pstride, int S,
                                                         const float* __restrict__ b2_0, const float* __restrict__ eps0, long long lstride,
                                                         float* __restrict__ Y, long long SY, int n, int B, int T, int D, int lo, int s) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= n * B) return;
    const int b = r % B, z = r / B, t = s - lo - z, nq = D >> 2;
    const float4* b2 = reinterpret_cast<const float4*>(b2_0 + (long long)(lo + z) * lstride);
    const float4* xr = reinterpret_cast<const float4*>(x1 + (long long)r * D);
    float4 pv[4][4], xv[4], bv[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int q = lane + 64 * k;
        if (q < nq) {
            xv[k] = xr[q];
            bv[k] = b2[q];
#pragma unroll
            for (int u = 0; u < 4; u++)
                if (u < S) pv[k][u] = reinterpret_cast<const float4*>(fp + (long long)u * pstride + (long long)r * D)[q];
        }
    }
    float4 v[4];
    float ss = 0.f;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (lane + 64 * k < nq) {
            float4 a = pv[k][0];
#pragma unroll
            for (int u = 1; u < 4; u++)
                if (u < S) { a.x += pv[k][u].x; a.y += pv[k][u].y; a.z += pv[k][u].z; a.w += pv[k][u].w; }
            v[k] = make_float4((a.x + bv[k].x) + xv[k].x, (a.y + bv[k].y) + xv[k].y, (a.z + bv[k].z) + xv[k].z, (a.w + bv[k].w) + xv[k].w);
            ss += v[k].x * v[k].x + v[k].y * v[k].y + v[k].z * v[k].z + v[k].w * v[k].w;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
    const float sc = 1.0f / sqrtf(ss / (float)D + expf(eps0[(long long)(lo + z) * lstride]));
    float4* yr = reinterpret_cast<float4*>(Y + (long long)(lo + z + 1) * SY + ((long long)b * T + t) * D);
#pragma unroll
    for (int k = 0; k < 4; k++)
        if (lane + 64 * k < nq) yr[lane + 64 * k] = make_float4(v[k].x * sc, v[k].y * sc, v[k].z * sc, v[k].w * sc);
}

__global__ void k_add_inplace(float* __restrict__ a, const float* __restrict__ b, long long n4) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    float4 x = reinterpret_cast<float4*>(a)[i], y = reinterpret_cast<const float4*>(b)[i];
    reinterpret_cast<float4*>(a)[i] = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
}

__global__ void k_gather_rows(const float* __restrict__ pool, long long slot_stride, long long off, const int* __restrict__ slots,
                              float* __restrict__ out, int B, int width) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * width) return;
    const int b = i / width, j = i - b * width;
    out[i] = pool[(long long)slots[b] * slot_stride + off + j];
}
__global__ void k_scatter_rows(float* __restrict__ pool, long long slot_stride, long long off, const int* __restrict__ slots,
                               const float* __restrict__ in, int ldin, int B, int width) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * width) return;
    const int b = i / width, j = i - b * width;
    pool[(long long)slots[b] * slot_stride + off + j] = in[(long long)b * ldin + j];
}

}  // namespace

void lstm_cell(const Ctx& ctx, const float* gx, long long ldgx, const float* gh, int ldgh, float* c, float* hf, int B, int Hh) {
    if (ctx.dry) return;
    hipLaunchKernelGGL(k_lstm_cell, dim3(cdiv((long long)B * Hh, 256)), dim3(256), 0, ctx.stream, gx, ldgx, gh, ldgh, c, hf, B, Hh);
    K2_HIP(hipGetLastError());
}
void lstm_cell_rows(const Ctx& ctx, const float* gates, float* c, float* hf, int rows, int Hh) {
    if (ctx.dry) return;
    hipLaunchKernelGGL(k_lstm_cell_rows, dim3(cdiv((long long)rows * Hh, 256)), dim3(256), 0, ctx.stream, gates, c, hf, rows, Hh);
    K2_HIP(hipGetLastError());
}
void lstm_add_frame(const Ctx& ctx, const float* Y, long long SY, const float* hp, long long pstride, int S, float* h, float* x1, int n, int B,
                    int T, int D, int lo, int s) {
    if (ctx.dry) return;
    hipLaunchKernelGGL(k_lstm_add_frame, dim3(cdiv((long long)n * B * D, 256)), dim3(256), 0, ctx.stream, Y, SY, hp, pstride, S, h, x1, n, B, T,
                       D, lo, s);
    K2_HIP(hipGetLastError());
}
void lstm_norm_frame(const Ctx& ctx, const float* x1, const float* fp, long long pstride, int S, const float* b2_0, const float* eps0,
                     long long lstride, float* Y, long long SY, int n, int B, int T, int D, int lo, int s) {
    K2_REQUIRE(D <= 1024 && D % 4 == 0 && S <= 4 && pstride % 4 == 0, "lstm: d_model %d / %d partials unsupported", D, S);
    if (ctx.dry) return;
    hipLaunchKernelGGL(k_lstm_norm_frame, dim3(cdiv(n * B, 4)), dim3(256), 0, ctx.stream, x1, fp, pstride, S, b2_0, eps0, lstride, Y, SY, n, B, T,
                       D, lo, s);
    K2_HIP(hipGetLastError());
}
void add_inplace(const Ctx& ctx, float* a, const float* b, long long n) {
    if (ctx.dry) return;
    K2_REQUIRE(n % 4 == 0, "add_inplace: n %% 4 != 0");
    hipLaunchKernelGGL(k_add_inplace, dim3(cdiv(n / 4, 256)), dim3(256), 0, ctx.stream, a, b, n / 4);
    K2_HIP(hipGetLastError());
}
void gather_rows(const Ctx& ctx, const float* pool, long long slot_stride, long long off, const int* slots, float* out, int B, int width) {
    if (ctx.dry) return;
    hipLaunchKernelGGL(k_gather_rows, dim3(cdiv((long long)B * width, 256)), dim3(256), 0, ctx.stream, pool, slot_stride, off, slots, out, B, width);
    K2_HIP(hipGetLastError());
}
void scatter_rows(const Ctx& ctx, float* pool, long long slot_stride, long long off, const int* slots, const float* in, int ldin, int B,
                  int width) {
    if (ctx.dry) return;
    hipLaunchKernelGGL(k_scatter_rows, dim3(cdiv((long long)B * width, 256)), dim3(256), 0, ctx.stream, pool, slot_stride, off, slots, in, ldin, B, width);
    K2_HIP(hipGetLastError());
}

}  // namespace k2hip
