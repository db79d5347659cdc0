// Probe (GPU box): what ONE CU can fetch from L2 for the search's joiner sweep, by access pattern.
// 64 workgroups x 512 threads (as k_greedy's two slabs x 32 streams), each re-reading "its" slab of a k-major [J][Vp] f32 matrix
// (J = 512, Vp = 500: 1 MB, slab = 250 columns = 0.5 MB) R times.
//   mode 0: k_greedy's pattern -- lane = (k slice 0..7) x (column group 0..7), a wave instruction = 8 rows x 128 B, 8 loads in flight
//   mode 1: row-contiguous -- a wave instruction = 1 row x 1 KB (256 columns), waves = k slices, 8 loads in flight
//   mode 2: as 1 with 16 loads in flight
//   mode 3: LDS-DMA (global_load_lds_dwordx4), a wave instruction = 1 KB of one row into LDS, 8 in flight per wave, nothing consumed
// build: hipcc -O3 --offload-arch=gfx950 sweep_fetch_probe.hip -o sweep_fetch_probe && ./sweep_fetch_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
constexpr int J = 512, VP = 500, GT = 512;

template <int MODE>
__global__ __launch_bounds__(GT) void k_probe(const float* __restrict__ W, float* __restrict__ out, int R) {
    __shared__ __attribute__((aligned(16))) float lds[8 * 8 * 256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int part = blockIdx.x & 1;
    const int c0 = part * 250;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int r = 0; r < R; r++) {
        asm volatile("" ::: "memory");   // the sweep's loads are not loop-invariant to the compiler
        if (MODE == 0) {
            const int ks = lane >> 3, cgl = lane & 7, kper = J / 8;
            for (int cgb = 0; cgb < 63; cgb += 64) {
                const int cg = min(cgb + wave * 8 + cgl, 61);
                const float* wp = W + (long long)(ks * kper) * VP + c0 + 4 * cg;
                for (int kb = 0; kb < kper; kb += 8) {
                    float4 wv[8];
#pragma unroll
                    for (int i = 0; i < 8; i++) wv[i] = *reinterpret_cast<const float4*>(wp + (long long)(kb + i) * VP);
#pragma unroll
                    for (int i = 0; i < 8; i++) { acc.x += wv[i].x; acc.y += wv[i].y; acc.z += wv[i].z; acc.w += wv[i].w; }
                }
            }
        } else if (MODE == 1 || MODE == 2) {
            constexpr int NL = MODE == 1 ? 8 : 16;
            const int kper = J / 8;
            const float* wp = W + (long long)(wave * kper) * VP + c0 + min(4 * lane, 246);
            for (int kb = 0; kb < kper; kb += NL) {
                float4 wv[NL];
#pragma unroll
                for (int i = 0; i < NL; i++) wv[i] = *reinterpret_cast<const float4*>(wp + (long long)(kb + i) * VP);
#pragma unroll
                for (int i = 0; i < NL; i++) { acc.x += wv[i].x; acc.y += wv[i].y; acc.z += wv[i].z; acc.w += wv[i].w; }
            }
        } else {
            const int kper = J / 8;
            const float* wp = W + (long long)(wave * kper) * VP + c0 + min(4 * lane, 246);
            const unsigned lds_base = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)lds;
            for (int kb = 0; kb < kper; kb += 8) {
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const float* gp = wp + (long long)(kb + i) * VP;
                    const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + ((wave * 8 + i) * 256) * 4);
                    unsigned keep;
                    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                                 : "=&s"(keep) : "v"(gp), "s"(dst) : "memory");
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        }
    }
    if (MODE == 3) acc.x = lds[tid];
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[blockIdx.x * GT + tid] = acc.x;
}

template <int MODE>
int run(const float* W, float* out, int nwg) {
    const int R = 200;
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL(k_probe<MODE>, dim3(nwg), dim3(GT), 0, 0, W, out, 5);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(k_probe<MODE>, dim3(nwg), dim3(GT), 0, 0, W, out, R);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    const double bytes = (double)R * 512 * 250 * 4;   // per workgroup
    printf("mode %d, %3d workgroups: %.3f ms, %.1f us per sweep, %.1f GB/s per CU, %.2f TB/s total\n", MODE, nwg, ms, ms * 1e3 / R,
           bytes / (ms * 1e-3) / 1e9, bytes * nwg / (ms * 1e-3) / 1e12);
    return 0;
}

int main() {
    float *W, *out;
    std::vector<float> h((size_t)J * VP, 0.5f);
    CK(hipMalloc(&W, h.size() * 4 + 4096));
    CK(hipMalloc(&out, 256 * GT * 4));
    CK(hipMemcpy(W, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    for (int nwg : {64, 16, 256}) {
        if (run<0>(W, out, nwg) || run<1>(W, out, nwg) || run<2>(W, out, nwg) || run<3>(W, out, nwg)) return 1;
    }
    return 0;
}
