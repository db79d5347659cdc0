// Probe: which element of the GEMM main loop keeps a SIMD pair below the 64-cycle MFMA rate?
// One workgroup per CU; per iteration every wave issues 16 v_mfma_f32_32x32x2_f32 whose operands come from 8 ds_read_b128
// of the PREVIOUS iteration (register double buffer, as in gemm_f32_mfma_ring).  Variants: waves per workgroup (4 = one per
// SIMD, 8 = two per SIMD), s_barrier per iteration or not, LDS reads or not, one or two accumulators.
// hipcc --offload-arch=gfx950 -O3 mfma_loop_probe.hip -o /tmp/mfma_loop_probe && /tmp/mfma_loop_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int WAVES, bool BARRIER, bool LDSREAD, int NACC, bool SETPRIO, int DMA = 0, int NDMA = 3>
__global__ __launch_bounds__(64 * WAVES) void probe(float* out, int iters, unsigned long long* cyc, const float* big = nullptr, long long big_floats = 0) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 2 * 192 * 32; i += blockDim.x) sm[i] = (float)(i % 97) * 0.01f;
    __syncthreads();
    f32x16 acc[2];
    for (int r = 0; r < 16; r++) acc[0][r] = acc[1][r] = 0.f;
    const float* base = sm + (wave % 4) * 32 * 32 + (lane & 31) * 32 + 4 * (lane >> 5);
    float4 fa[2][4], fb[2][4];
    for (int g = 0; g < 4; g++) { fa[0][g] = *(const float4*)(base + 8 * g); fb[0][g] = *(const float4*)(base + 4096 + 8 * g); }
    if (SETPRIO && wave >= 4) __builtin_amdgcn_s_setprio(1);
    // DMA: every wave issues NDMA global_load_lds_dwordx4 (1 KB each) per iteration into a scratch ring behind the fragment area and
    // waits (counted vmcnt) for the ones issued two iterations earlier -- the traffic of a 128x64x32 GEMM tile when NDMA = 3, WAVES = 8
    const unsigned lds_base = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)sm;
    const long long span = (DMA == 1 || DMA == 3 || DMA >= 5) ? 64 * 1024 : big_floats;  // floats this launch walks through (1: L2-resident, 2: streaming)
    long long pos = ((long long)blockIdx.x * WAVES + wave) * 256 * NDMA % (span > 0 ? span : 1);
    auto dma = [&](int it) {
        _Pragma("unroll") for (int q = 0; q < NDMA; q++) {
            const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + (12288 + ((it % 3) * WAVES * NDMA + wave * NDMA + q) * 256) * 4);
            const float* gp = big + (pos + q * 256 + lane * 4) % span;
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(gp), "s"(dst) : "memory");
        }
        pos = (pos + (long long)gridDim.x * WAVES * 256 * NDMA) % span;
    };
    // DMA == 3: the same bytes register-staged (global_load_dwordx4 -> VGPR, ds_write_b128 one iteration later), as the vendor
    // library's fp32 kernels do
    float4 stage[NDMA];
    _Pragma("unroll") for (int q = 0; q < NDMA; q++) stage[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    auto staged = [&](int it) {
        _Pragma("unroll") for (int q = 0; q < NDMA; q++) {
            *(float4*)(sm + 12288 + ((it % 3) * WAVES * NDMA + wave * NDMA + q) * 256 + lane * 4) = stage[q];
            stage[q] = *(const float4*)(big + (pos + q * 256 + lane * 4) % span);
        }
        pos = (pos + (long long)gridDim.x * WAVES * 256 * NDMA) % span;
    };
    // DMA == 5: as 1 but M0 is not saved / restored around each DMA (clobbered);  DMA == 6: as 5 and the NDMA instructions of an
    // iteration are issued one per MFMA slot (after MFMAs 0, 1, 2, ...) instead of back to back
    auto dma1 = [&](int it, int q) {
        const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + (12288 + ((it % 3) * WAVES * NDMA + wave * NDMA + q) * 256) * 4);
        const float* gp = big + (pos + q * 256 + lane * 4) % span;
        if (DMA == 8) asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" : : "v"(gp), "s"(dst) : "memory", "m0");
        else if (DMA == 9) {
            const unsigned off = (unsigned)(((pos + q * 256 + lane * 4) % span) * 4);
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %2" : : "v"(off), "s"(dst), "s"(big) : "memory", "m0");
        } else asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(gp), "s"(dst) : "memory", "m0");
        if (q == NDMA - 1) pos = (pos + (long long)gridDim.x * WAVES * 256 * NDMA) % span;
    };
    if (DMA >= 5) { _Pragma("unroll") for (int q = 0; q < NDMA; q++) dma1(0, q); _Pragma("unroll") for (int q = 0; q < NDMA; q++) dma1(1, q); }
    if (DMA == 1 || DMA == 2) { dma(0); dma(1); }
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
#define STEP(cu, nx, st)                                                                                       \
    _Pragma("unroll") for (int g = 0; g < 4; g++) {                                                            \
        _Pragma("unroll") for (int e = 0; e < 4; e++) {                                                        \
            const float av = e == 0 ? fa[cu][g].x : e == 1 ? fa[cu][g].y : e == 2 ? fa[cu][g].z : fa[cu][g].w; \
            const float bv = e == 0 ? fb[cu][g].x : e == 1 ? fb[cu][g].y : e == 2 ? fb[cu][g].z : fb[cu][g].w; \
            acc[NACC == 2 ? (e & 1) : 0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[NACC == 2 ? (e & 1) : 0], 0, 0, 0); \
            if (e == 0 && g == 0 && (DMA == 1 || DMA == 2)) { __builtin_amdgcn_sched_barrier(0); dma(it + 2 + ((st) == 0)); } \
            if (e == 0 && g == 0 && (DMA == 5 || DMA >= 7)) { __builtin_amdgcn_sched_barrier(0); _Pragma("unroll") for (int q = 0; q < NDMA; q++) dma1(it + 2 + ((st) == 0), q); __builtin_amdgcn_sched_barrier(0); } \
            if (DMA == 6 && g * 4 + e < NDMA) { __builtin_amdgcn_sched_barrier(0); dma1(it + 2 + ((st) == 0), g * 4 + e); __builtin_amdgcn_sched_barrier(0); } \
            if (e == 0 && g == 0 && (DMA == 3 || DMA == 4)) { __builtin_amdgcn_sched_barrier(0); staged(it + 2 + ((st) == 0)); __builtin_amdgcn_sched_barrier(0); } \
            if (e == 0 && LDSREAD) {                                                                           \
                __builtin_amdgcn_sched_barrier(0);                                                             \
                fa[nx][g] = *(const float4*)(base + (st) + 8 * g);                                             \
                fb[nx][g] = *(const float4*)(base + (st) + 4096 + 8 * g);                                      \
                __builtin_amdgcn_sched_barrier(0);                                                             \
            }                                                                                                  \
        }                                                                                                      \
    }
    for (int it = 0; it < iters; it += 2) {
        if (DMA == 1 || DMA == 2 || (DMA >= 5 && DMA != 7)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");
        if (BARRIER) __builtin_amdgcn_s_barrier();
        STEP(0, 1, 6144)
        if (DMA == 1 || DMA == 2 || (DMA >= 5 && DMA != 7)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");
        if (BARRIER) __builtin_amdgcn_s_barrier();
        STEP(1, 0, 0)
    }
    if (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int r = 0; r < 16; r++) s += acc[0][r] + acc[1][r];
    _Pragma("unroll") for (int q = 0; q < NDMA; q++) s += stage[q].x;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * WAVES + wave] = t1 - t0;
}

template <int WAVES, bool BARRIER, bool LDSREAD, int NACC, bool SETPRIO = false, int DMA = 0, int NDMA = 3>
void run(const char* name) {
    float* out;
    unsigned long long* cyc;
    const int nblk = 256, iters = 400;
    static float* big = nullptr;
    const long long big_floats = 64ll << 20;  // 256 MB
    if (!big) { hipMalloc(&big, big_floats * 4); hipMemset(big, 0, big_floats * 4); }
    hipMalloc(&out, sizeof(float) * nblk * 64 * WAVES);
    hipMalloc(&cyc, sizeof(unsigned long long) * nblk * WAVES);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipFuncSetAttribute((const void*)probe<WAVES, BARRIER, LDSREAD, NACC, SETPRIO, DMA, NDMA>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    probe<WAVES, BARRIER, LDSREAD, NACC, SETPRIO, DMA, NDMA><<<nblk, 64 * WAVES, 128 * 1024>>>(out, 20, cyc, big, big_floats);
    hipEventRecord(e0);
    probe<WAVES, BARRIER, LDSREAD, NACC, SETPRIO, DMA, NDMA><<<nblk, 64 * WAVES, 128 * 1024>>>(out, iters, cyc, big, big_floats);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    static unsigned long long h[256 * 16];
    hipMemcpy(h, cyc, sizeof(unsigned long long) * nblk * WAVES, hipMemcpyDeviceToHost);
    double lo = 0, hi = 0;  // mean cycles of waves 0..3 and of the rest
    for (int b = 0; b < nblk; b++)
        for (int w = 0; w < WAVES; w++) (w < 4 ? lo : hi) += (double)h[b * WAVES + w];
    lo /= nblk * 4.0;
    hi /= WAVES > 4 ? nblk * (WAVES - 4.0) : 1.0;
    const double mfma_per_simd = (double)iters * 16 * (WAVES / 4);
    printf("%-58s %.3f ms  %6.1f TF | cycles per MFMA per SIMD: %.1f (waves 0-3 finish at %.0f, 4+ at %.0f cycles)\n", name, ms,
           (double)nblk * WAVES * iters * 16 * 4096.0 / (ms * 1e-3) / 1e12, (WAVES > 4 ? hi : lo) / mfma_per_simd, lo, hi);
    hipFree(out); hipFree(cyc);
}

int main(int argc, char** argv) {
    if (argc > 1) {  // round 2: what does the operand fetch cost, by mechanism
        if (argv[1][0] == 'i') {  // is it the issue of the instruction, the wait, or the bytes?
            run<4, true, true, 2, false, 5, 6>("4 waves + 6 LDS-DMA b128, counted wait");
            run<4, true, true, 2, false, 7, 6>("4 waves + 6 LDS-DMA b128, NO wait in the loop");
            run<4, true, true, 2, false, 8, 6>("4 waves + 6 LDS-DMA b32 (a quarter of the bytes), counted wait");
            run<4, true, true, 2, false, 9, 6>("4 waves + 6 LDS-DMA b128, SGPR base + 32-bit VGPR offset");
            run<4, false, true, 2, false, 5, 6>("4 waves + 6 LDS-DMA b128, counted wait, NO barrier");
            run<8, true, true, 2, false, 7, 3>("8 waves + 3 LDS-DMA b128, NO wait in the loop");
            run<8, true, true, 2, false, 8, 3>("8 waves + 3 LDS-DMA b32, counted wait");
            run<8, true, true, 2, false, 9, 3>("8 waves + 3 LDS-DMA b128, SGPR base + 32-bit VGPR offset");
            return 0;
        }
        if (argv[1][0] == 'm') {  // M0 handling and placement of the LDS-DMA issue
            run<4, true, true, 2, false, 1, 2>("4 waves + 2 LDS-DMA/wave/iter from L2, m0 saved/restored");
            run<4, true, true, 2, false, 5, 2>("4 waves + 2 LDS-DMA/wave/iter from L2, m0 clobbered");
            run<4, true, true, 2, false, 6, 2>("4 waves + 2 LDS-DMA/wave/iter from L2, m0 clobbered, one per MFMA slot");
            run<4, true, true, 2, false, 1, 6>("4 waves + 6 LDS-DMA/wave/iter from L2, m0 saved/restored");
            run<4, true, true, 2, false, 5, 6>("4 waves + 6 LDS-DMA/wave/iter from L2, m0 clobbered");
            run<4, true, true, 2, false, 6, 6>("4 waves + 6 LDS-DMA/wave/iter from L2, m0 clobbered, one per MFMA slot");
            run<8, true, true, 2, false, 1, 3>("8 waves + 3 LDS-DMA/wave/iter from L2, m0 saved/restored");
            run<8, true, true, 2, false, 5, 3>("8 waves + 3 LDS-DMA/wave/iter from L2, m0 clobbered");
            run<8, true, true, 2, false, 6, 3>("8 waves + 3 LDS-DMA/wave/iter from L2, m0 clobbered, one per MFMA slot");
            return 0;
        }
        run<8, true, true, 2>("8 waves, barrier, lds reads, 2 acc (no fetch)");
        run<8, true, true, 2, false, 1, 3>("8 waves + 3 LDS-DMA/wave/iter from L2");
        run<8, true, false, 2, false, 1, 3>("8 waves + 3 LDS-DMA/wave/iter from L2, NO lds reads");
        run<8, true, true, 2, false, 3, 3>("8 waves + 3 register-staged loads/wave/iter from L2");
        run<8, true, true, 2, false, 4, 3>("8 waves + 3 register-staged loads/wave/iter streaming");
        run<8, true, true, 2, false, 2, 3>("8 waves + 3 LDS-DMA/wave/iter streaming");
        run<4, true, true, 2>("4 waves, barrier, lds reads, 2 acc (no fetch)");
        run<4, true, true, 2, false, 1, 6>("4 waves + 6 LDS-DMA/wave/iter from L2");
        run<4, true, true, 2, false, 3, 6>("4 waves + 6 register-staged loads/wave/iter from L2");
        run<4, true, true, 2, false, 1, 3>("4 waves + 3 LDS-DMA/wave/iter from L2 (12 KB per 1024 cycles)");
        run<4, true, true, 2, false, 3, 3>("4 waves + 3 register-staged loads/wave/iter from L2");
        run<4, true, true, 2, false, 1, 2>("4 waves + 2 LDS-DMA/wave/iter from L2 (8 KB per 1024 cycles)");
        return 0;
    }
    run<4, false, false, 1>("4 waves, no barrier, no lds, 1 acc");
    run<4, false, true, 1>("4 waves, no barrier, lds reads, 1 acc");
    run<4, false, true, 2>("4 waves, no barrier, lds reads, 2 acc");
    run<4, true, true, 2>("4 waves, barrier, lds reads, 2 acc");
    run<8, false, false, 1>("8 waves, no barrier, no lds, 1 acc");
    run<8, false, true, 1>("8 waves, no barrier, lds reads, 1 acc");
    run<8, false, true, 2>("8 waves, no barrier, lds reads, 2 acc");
    run<8, true, false, 2>("8 waves, barrier, no lds, 2 acc");
    run<8, true, true, 1>("8 waves, barrier, lds reads, 1 acc");
    run<8, true, true, 2>("8 waves, barrier, lds reads, 2 acc");
    run<8, true, true, 2, true>("8 waves, barrier, lds reads, 2 acc, setprio 1 on waves 4-7");
    run<16, true, true, 2>("16 waves, barrier, lds reads, 2 acc");
    run<8, true, true, 2, false, 1, 3>("8 waves, barrier, lds, 2 acc + 3 DMA/wave/iter from L2 (24 KB per CU per iter)");
    run<8, true, true, 2, false, 2, 3>("8 waves, barrier, lds, 2 acc + 3 DMA/wave/iter streaming 256 MB");
    run<8, true, true, 2, false, 1, 2>("8 waves, barrier, lds, 2 acc + 2 DMA/wave/iter from L2 (16 KB per CU per iter)");
    run<8, true, true, 2, false, 2, 2>("8 waves, barrier, lds, 2 acc + 2 DMA/wave/iter streaming");
    run<8, true, true, 2, false, 1, 1>("8 waves, barrier, lds, 2 acc + 1 DMA/wave/iter from L2");
    run<8, true, true, 2, false, 2, 1>("8 waves, barrier, lds, 2 acc + 1 DMA/wave/iter streaming");
    run<8, false, true, 2, false, 1, 3>("8 waves, NO barrier, lds, 2 acc + 3 DMA/wave/iter from L2");
    run<4, true, true, 2, false, 1, 6>("4 waves, barrier, lds, 2 acc + 6 DMA/wave/iter from L2 (24 KB per CU per 1024 cycles)");
    return 0;
}
