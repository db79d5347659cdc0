// Probe: package power and sustained rate of a pure register-to-register fp32 MFMA stream, by instruction shape:
// v_mfma_f32_32x32x2_f32 (4096 flop, 64 cycles, 16 accumulator registers) against v_mfma_f32_16x16x4_f32 (2048 flop, 32 cycles,
// 4 accumulator registers).  Same flop rate; the 32x32 shape moves twice the accumulator bytes per flop through the register file.
// Run next to a rocm-smi sampling loop (tools/probes/run_mfma_shape_power.sh).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE>
__global__ __launch_bounds__(256) void spin(float* out, int iters, float seed) {
    const int lane = threadIdx.x & 63;
    float a = seed + lane * 0.37f, b = 0.61f - lane * 0.011f;
    float s = 0;
    if (SHAPE == 32) {
        f32x16 acc[4];
        for (int t = 0; t < 4; t++)
            for (int r = 0; r < 16; r++) acc[t][r] = 0.f;
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int m = 0; m < 16; m++) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[m & 3], 0, 0, 0);
            a = -a;  // keeps the products from settling (operand toggling is part of the power)
        }
        for (int t = 0; t < 4; t++)
            for (int r = 0; r < 16; r++) s += acc[t][r];
    } else {
        f32x4 acc[16];
        for (int t = 0; t < 16; t++) acc[t] = f32x4{0, 0, 0, 0};
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int m = 0; m < 32; m++) acc[m & 15] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[m & 15], 0, 0, 0);
            a = -a;
        }
        for (int t = 0; t < 16; t++) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int SHAPE>
void run(double seconds) {
    float* out;
    hipMalloc(&out, sizeof(float) * 256 * 256 * 2);
    const int iters = 20000;
    auto t0 = std::chrono::steady_clock::now();
    long launches = 0;
    double el = 0;
    do {
        for (int i = 0; i < 4; i++) spin<SHAPE><<<512, 256>>>(out, iters, 1.0f + launches);  // 2 workgroups of 4 waves per CU
        hipDeviceSynchronize();
        launches += 4;
        el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    } while (el < seconds);
    const double flop = (double)launches * 512 * 4 * iters * 16 * 4096.0;  // both shapes: 65536 flop per wave and iteration
    printf("shape %dx%d: %.1f TF/s over %.1f s\n", SHAPE, SHAPE, flop / el / 1e12, el);
    fflush(stdout);
    hipFree(out);
}

int main(int argc, char** argv) {
    const double sec = argc > 1 ? atof(argv[1]) : 5.0;
    run<32>(sec);
    run<16>(sec);
    run<32>(sec);
    run<16>(sec);
    return 0;
}
