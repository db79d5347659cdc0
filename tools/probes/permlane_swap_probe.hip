// Probe (gfx950): do v_permlane16_swap / v_permlane32_swap reproduce v + __shfl_xor(v, 16) and v + __shfl_xor(v, 32)?
// build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 -o /tmp/plp tools/probes/permlane_swap_probe.hip && /tmp/plp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
__global__ void k(const float* in, float* ref16, float* got16, float* ref32, float* got32) {
    const int t = threadIdx.x;
    const float v = in[t];
    ref16[t] = v + __shfl_xor(v, 16);
    ref32[t] = v + __shfl_xor(v, 32);
    // The builtins (__builtin_amdgcn_permlane16_swap / 32_swap) return {vdst, src0} after the swap; fed the same value twice, this
    // compiler treats the two results as equal and emits v_add v, r0, r0 (2 v in every lane, with or without an opaque copy of the
    // operand).  As inline asm with two read-write operands the swap keeps its two registers: new a = [a.r0, b.r0, a.r2, b.r2],
    // new b = [a.r1, b.r1, a.r3, b.r3] (rows of 16 lanes) for 16_swap; [a.lo32, b.lo32] / [a.hi32, b.hi32] for 32_swap.
    float a16 = v, b16 = v;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a16), "+v"(b16));
    got16[t] = a16 + b16;
    float a32 = v, b32 = v;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a32), "+v"(b32));
    got32[t] = a32 + b32;
}
int main() {
    float h[64], *d, *o;
    unsigned s = 12345u;
    for (int i = 0; i < 64; i++) { s = s * 1664525u + 1013904223u; h[i] = (float)(s >> 8) / 8388608.0f - 1.0f; }
    hipMalloc(&d, 256); hipMalloc(&o, 4 * 256);
    hipMemcpy(d, h, 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, o + 64, o + 128, o + 192);
    float r[256];
    hipMemcpy(r, o, 1024, hipMemcpyDeviceToHost);
    int bad16 = 0, bad32 = 0;
    for (int i = 0; i < 64; i++) {
        bad16 += memcmp(&r[i], &r[64 + i], 4) != 0;
        bad32 += memcmp(&r[128 + i], &r[192 + i], 4) != 0;
    }
    for (int i = 0; i < 64; i += 9) printf("lane %2d: in %+.6f ref16 %+.6f got16 %+.6f ref32 %+.6f got32 %+.6f\n", i, h[i], r[i], r[64 + i], r[128 + i], r[192 + i]);
    printf("permlane16_swap vs shfl_xor 16: %d lanes differ; permlane32_swap vs shfl_xor 32: %d lanes differ\n", bad16, bad32);
    return bad16 + bad32 != 0;
}
