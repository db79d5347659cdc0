// Probe: what rate does v_mfma_f32_32x32x2_f32 sustain on gfx950 under different issue patterns?
// hipcc --offload-arch=gfx950 -O3 mfma_f32_probe.hip -o mfma_probe && ./mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, bool LDS>
__global__ __launch_bounds__(256) void probe(float* out, int iters) {
    __shared__ float sm[4096];
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; i++)
        for (int r = 0; r < 16; r++) acc[i][r] = 0.f;
    float a = threadIdx.x * 0.001f, b = threadIdx.x * 0.002f;
    if (LDS) { sm[threadIdx.x] = a; sm[threadIdx.x + 256] = b; __syncthreads(); }
    for (int it = 0; it < iters; it++) {
        if (LDS) {
            float4 fa = *reinterpret_cast<float4*>(&sm[(threadIdx.x & 63) * 4]);
            float4 fb = *reinterpret_cast<float4*>(&sm[256 + (threadIdx.x & 63) * 4]);
#pragma unroll
            for (int e = 0; e < 4; e++)
#pragma unroll
                for (int i = 0; i < NACC; i++)
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(e == 0 ? fa.x : e == 1 ? fa.y : e == 2 ? fa.z : fa.w,
                                                                 e == 0 ? fb.x : e == 1 ? fb.y : e == 2 ? fb.z : fb.w, acc[i], 0, 0, 0);
        } else {
#pragma unroll
            for (int e = 0; e < 4; e++)
#pragma unroll
                for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
        }
    }
    float s = 0;
    for (int i = 0; i < NACC; i++)
        for (int r = 0; r < 16; r++) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC, bool LDS>
void run(const char* name, int blocks_per_cu) {
    float* out;
    int nblk = 256 * blocks_per_cu;
    hipMalloc(&out, sizeof(float) * nblk * 256);
    int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    probe<NACC, LDS><<<nblk, 256>>>(out, 10);
    hipEventRecord(e0);
    probe<NACC, LDS><<<nblk, 256>>>(out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)nblk * 4 * iters * 4 * NACC * 4096.0;
    printf("%-28s blocks/CU %d: %.3f ms  %.1f TFLOP/s\n", name, blocks_per_cu, ms, flops / (ms * 1e-3) / 1e12);
    hipFree(out);
}

int main() {
    for (int bpc = 1; bpc <= 4; bpc *= 2) {
        run<1, false>("1 acc, reg operands", bpc);
        run<2, false>("2 acc, reg operands", bpc);
        run<4, false>("4 acc, reg operands", bpc);
        run<4, true>("4 acc, lds operands", bpc);
        run<1, true>("1 acc, lds operands", bpc);
    }
    return 0;
}
