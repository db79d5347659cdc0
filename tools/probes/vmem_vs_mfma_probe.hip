// Probe: what does issuing ONE vector-memory instruction cost a wave that is streaming v_mfma_f32_32x32x2_f32?
// One workgroup of 4 waves (one per SIMD) per CU; each wave issues ITER x 16 MFMAs (two accumulators, operands in registers) and,
// per 16 MFMAs, NV vector-memory instructions whose address needs NO vector arithmetic in the loop (SGPR base advanced with
// s_add, constant per-lane VGPR offset), so the only thing added to the MFMA stream is the instruction itself.
// KIND: 0 none | 1 global_load_lds_dwordx4 (LDS-DMA) | 2 global_load_dwordx4 to VGPRs | 3 global_load_lds_dword | 4 ds_read_b128
// | 5 ds_write_b128 | 6 buffer_load_dwordx4 ... lds.  SPREAD: instructions back to back after MFMA 0, or one per MFMA slot.
// hipcc --offload-arch=gfx950 -O3 vmem_vs_mfma_probe.hip -o /tmp/vmem_vs_mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int KIND, int NV, bool SPREAD, int MF>
__global__ __launch_bounds__(256) void probe(float* out, int iters, unsigned long long* cyc, const float* big) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x16 acc[2];
    for (int r = 0; r < 16; r++) acc[0][r] = acc[1][r] = 0.f;
    float a = 1.0f + lane, b = 0.5f;
    const unsigned lds_base = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)sm;
    const unsigned m0v = __builtin_amdgcn_readfirstlane(lds_base + wave * 8192);
    asm volatile("s_mov_b32 m0, %0" ::"s"(m0v) : "m0");
    const unsigned voff = lane * 16;                                    // constant per-lane byte offset
    // wave-uniform base inside a 4 MB window of the buffer (kept as two 32-bit halves so that the compiler keeps it in SGPRs)
    const unsigned big_lo = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)big);
    const unsigned big_hi = __builtin_amdgcn_readfirstlane((unsigned)((unsigned long long)big >> 32));
    unsigned woff = __builtin_amdgcn_readfirstlane((unsigned)((blockIdx.x * 4 + wave) * 4096));
    unsigned long long sbase = ((unsigned long long)big_hi << 32) | (big_lo + woff);
    const float* lanep = big + lane * 4;  // per-lane pointer (KIND 8, 9)
    f32x4 sink[8];
    for (int q = 0; q < 8; q++) sink[q] = f32x4{0, 0, 0, 0};
    float* lp = sm + wave * 2048 + lane * 4;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int m = 0; m < 16; m++) {
            if (MF) acc[m & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[m & 1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < NV; q++) {
                const bool here = SPREAD ? (m == q) : (m == 0);
                if (!here) continue;
                if (KIND == 1) asm volatile("global_load_lds_dwordx4 %0, %1 offset:%2" ::"v"(voff), "s"(sbase), "n"(0) : "memory");
                if (KIND == 3) asm volatile("global_load_lds_dword %0, %1" ::"v"(voff), "s"(sbase) : "memory");
                if (KIND == 2) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(sink[q]) : "v"(voff), "s"(sbase) : "memory");
                if (KIND == 7) {  // + a new LDS destination per instruction (M0 rewritten), as a tile fill needs
                    const unsigned dst = __builtin_amdgcn_readfirstlane(m0v + q * 1024);
                    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(dst) : "memory", "m0");
                }
                if (KIND == 8) {  // + a per-lane 64-bit address advanced by vector arithmetic (global_load_lds ... off)
                    const unsigned dst = __builtin_amdgcn_readfirstlane(m0v + q * 1024);
                    const float* gp = lanep + (size_t)woff / 4 + q * 64;
                    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gp), "s"(dst) : "memory", "m0");
                }
                if (KIND == 9) {  // as 8, M0 saved and restored around the instruction (the kernels' form until now)
                    const unsigned dst = __builtin_amdgcn_readfirstlane(m0v + q * 1024);
                    const float* gp = lanep + (size_t)woff / 4 + q * 64;
                    unsigned keep;
                    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                                 : "=&s"(keep) : "v"(gp), "s"(dst) : "memory");
                }
                if (KIND == 4) asm volatile("ds_read_b128 %0, %1" : "=v"(sink[q]) : "v"((unsigned)(wave * 8192 + lane * 16 + q * 1024)) : "memory");
                if (KIND == 5) asm volatile("ds_write_b128 %0, %1" ::"v"((unsigned)(wave * 8192 + lane * 16 + q * 1024)), "v"(sink[q]) : "memory");
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (KIND == 1 || KIND == 2 || KIND == 3 || KIND >= 7) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NV) : "memory");  // the previous iteration's have landed
            woff = (woff + 0x100000u) & 0x3fffffu;  // next 1 MB of the window
            sbase = ((unsigned long long)big_hi << 32) | (big_lo + woff);
        }
        if (KIND == 4 || KIND == 5) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = lp[0];
    for (int r = 0; r < 16; r++) s += acc[0][r] + acc[1][r];
    for (int q = 0; q < 8; q++) s += sink[q][0];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
}

static float* big = nullptr;
static float* out = nullptr;
static unsigned long long* cyc = nullptr;

template <int KIND, int NV, bool SPREAD, int MF = 1>
void run(const char* name) {
    const int nblk = 256, iters = 400;
    if (!big) {
        float* raw;
        hipMalloc(&raw, 72 << 20);  // the window arithmetic in the kernel assumes the low 22 bits of `big` are 0: round up to 4 MB
        hipMemset(raw, 0, 72 << 20);
        big = (float*)(((unsigned long long)raw + 0x3fffff) & ~0x3fffffull);
        hipMalloc(&out, sizeof(float) * nblk * 256);
        hipMalloc(&cyc, sizeof(unsigned long long) * nblk * 4);
    }
    hipMemset(cyc, 0, sizeof(unsigned long long) * nblk * 4);
    hipFuncSetAttribute((const void*)probe<KIND, NV, SPREAD, MF>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    probe<KIND, NV, SPREAD, MF><<<nblk, 256, 64 * 1024>>>(out, 10, cyc, big);
    probe<KIND, NV, SPREAD, MF><<<nblk, 256, 64 * 1024>>>(out, iters, cyc, big);
    hipError_t err = hipDeviceSynchronize();
    if (err != hipSuccess || hipGetLastError() != hipSuccess) { printf("%s: launch failed: %s\n", name, hipGetErrorString(err)); return; }
    static unsigned long long h[1024];
    hipMemcpy(h, cyc, sizeof(unsigned long long) * nblk * 4, hipMemcpyDeviceToHost);
    double mean = 0;
    for (int i = 0; i < nblk * 4; i++) mean += (double)h[i];
    mean /= nblk * 4.0;
    printf("%-78s %7.1f cycles per MFMA slot (%.0f per 16)\n", name, mean / (iters * 16.0), mean / iters);
}

int main() {
    run<0, 0, false>("16 MFMAs, nothing else");
    run<1, 1, false>("+ 1 LDS-DMA b128");
    run<1, 2, false>("+ 2 LDS-DMA b128, back to back");
    run<1, 2, true>("+ 2 LDS-DMA b128, one per MFMA slot");
    run<1, 6, false>("+ 6 LDS-DMA b128, back to back");
    run<1, 6, true>("+ 6 LDS-DMA b128, one per MFMA slot");
    run<3, 6, true>("+ 6 LDS-DMA b32, one per MFMA slot");
    run<2, 6, false>("+ 6 global_load_dwordx4 to VGPRs, back to back");
    run<2, 6, true>("+ 6 global_load_dwordx4 to VGPRs, one per MFMA slot");
    run<4, 6, true>("+ 6 ds_read_b128, one per MFMA slot");
    run<5, 6, true>("+ 6 ds_write_b128, one per MFMA slot");
    run<7, 6, false>("+ 6 LDS-DMA b128 back to back, M0 rewritten for each");
    run<7, 6, true>("+ 6 LDS-DMA b128 one per slot, M0 rewritten for each");
    run<8, 6, false>("+ 6 LDS-DMA b128 back to back, M0 rewritten, per-lane 64-bit address from VALU");
    run<8, 6, true>("+ 6 LDS-DMA b128 one per slot, M0 rewritten, per-lane 64-bit address from VALU");
    run<9, 6, false>("+ 6 LDS-DMA b128 back to back, M0 saved/restored, per-lane 64-bit address from VALU");
    run<9, 6, true>("+ 6 LDS-DMA b128 one per slot, M0 saved/restored, per-lane 64-bit address from VALU");
    run<1, 6, true, 0>("NO MFMAs: 6 LDS-DMA b128 per iteration alone");
    run<2, 6, true, 0>("NO MFMAs: 6 global_load_dwordx4 per iteration alone");
    return 0;
}
