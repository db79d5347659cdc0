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
