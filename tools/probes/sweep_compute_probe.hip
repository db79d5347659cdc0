// Probe (GPU box): what bounds k_greedy's joiner sweep (csrc/greedy.hip) -- one round's [GF = 8 frames x J = 512] x [512 x 250-column slab]
// product per workgroup of 512 threads, 64 workgroups (32 streams x 2 slabs), repeated R times.
//   mode 0: the kernel's mapping -- lane = (k slice 0..7) x (column group 0..7), activations from LDS (8 addresses per instruction),
//           GL = 8 weight rows requested together, k slices summed by DPP / ds_bpermute
//   mode 1: as 0 without the LDS activation reads      mode 2: as 0 without the global weight loads      mode 3: as 0 without the shuffles
//   mode 4: wave = k slice, lane = 4 columns (a wave instruction reads 1 KB of one row); activations read from LDS at a wave-uniform
//           address; the 8 waves' partial sums meet in LDS
//   mode 5: as 4, activations held one k row per lane (8 VGPRs) and broadcast with v_readlane
//   mode 6: as 4 on the matrix pipe: v_mfma_f32_4x4x1_16B_f32, block = lane / 4, A = the row's activations (lane % 4 = frame), B = the
//           lane's weight; 8 instructions per k row cover 8 frames x 256 columns.  Its sums are compared bit for bit with mode 4's.
//   mode 7 / 8: as 6 with a ring of 8 / 16 weight rows per lane (row k + ring requested when row k has been used)
// build: hipcc -O3 --offload-arch=gfx950 [-fno-slp-vectorize] sweep_compute_probe.hip -o sweep_compute_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <type_traits>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
constexpr int J = 512, VP = 500, GT = 512, GF = 8, APAD = 8, GL = 8;

template <int MODE>
__global__ __launch_bounds__(GT) void k_probe(const float* __restrict__ W, const float* __restrict__ act_in, float* __restrict__ out, int R, float* __restrict__ dump = nullptr) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* actT = sm;                           // [J][GF] (+ APAD per k slice)
    float* part = sm + J * GF + 8 * APAD;       // mode 4/5: [8 waves][GF][256]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int slab = blockIdx.x & 1, c0 = slab * 250, kper = J / 8;
    for (int k = tid; k < J; k += GT)
        for (int f = 0; f < GF; f++) actT[k * GF + (k / kper) * APAD + f] = act_in[k * GF + f];
    __syncthreads();
    float keep = 0.f;
    for (int r = 0; r < R; r++) {
        asm volatile("" ::: "memory");
        float acc[GF][4];
#pragma unroll
        for (int f = 0; f < GF; f++)
#pragma unroll
            for (int j = 0; j < 4; j++) acc[f][j] = 0.f;
        if (MODE <= 3) {
            const int ks = lane >> 3, cgl = lane & 7;
            const int cg = min(wave * 8 + cgl, 61);
            const float* wp = W + (long long)(ks * kper) * VP + c0 + 4 * cg;
            const float* ap = actT + (ks * kper) * GF + ks * APAD;
            float4 a0c = *reinterpret_cast<const float4*>(ap), a1c = *reinterpret_cast<const float4*>(ap + 4);
            float4 wc[GL];
            if (MODE == 2) {
#pragma unroll
                for (int i = 0; i < GL; i++) wc[i] = *reinterpret_cast<const float4*>(wp + (long long)i * VP);
            }
            for (int kb = 0; kb < kper; kb += GL) {
                float4 wv[GL];
#pragma unroll
                for (int i = 0; i < GL; i++) {
                    if (MODE == 2) { wv[i] = wc[i]; asm volatile("" : "+v"(wv[i].x), "+v"(wv[i].y), "+v"(wv[i].z), "+v"(wv[i].w)); }
                    else wv[i] = *reinterpret_cast<const float4*>(wp + (long long)(kb + i) * VP);
                }
#pragma unroll
                for (int i = 0; i < GL; i++) {
                    const int k = kb + i;
                    float4 a0, a1;
                    if (MODE == 1) { a0 = a0c; a1 = a1c; asm volatile("" : "+v"(a0.x), "+v"(a0.y), "+v"(a0.z), "+v"(a0.w), "+v"(a1.x), "+v"(a1.y), "+v"(a1.z), "+v"(a1.w)); }
                    else { a0 = *reinterpret_cast<const float4*>(ap + k * GF); a1 = *reinterpret_cast<const float4*>(ap + k * GF + 4); }
                    const float av[GF] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
#pragma unroll
                    for (int f = 0; f < GF; f++) {
                        acc[f][0] += av[f] * wv[i].x; acc[f][1] += av[f] * wv[i].y; acc[f][2] += av[f] * wv[i].z; acc[f][3] += av[f] * wv[i].w;
                    }
                }
            }
            if (MODE != 3) {
#pragma unroll
                for (int f = 0; f < GF; f++)
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        float v = acc[f][j];
                        v += __shfl_xor(v, 8); v += __shfl_xor(v, 16); v += __shfl_xor(v, 32);
                        acc[f][j] = v;
                    }
            }
#pragma unroll
            for (int f = 0; f < GF; f++) keep += acc[f][0] + acc[f][1] + acc[f][2] + acc[f][3];
        } else {
            const float* wp = W + (long long)(wave * kper) * VP + c0 + min(4 * lane, 246);
            const float* ap = actT + (wave * kper) * GF + wave * APAD;
            float arow[GF];
            if (MODE == 5) {
                const float4 a0 = *reinterpret_cast<const float4*>(ap + lane * GF), a1 = *reinterpret_cast<const float4*>(ap + lane * GF + 4);
                arow[0] = a0.x; arow[1] = a0.y; arow[2] = a0.z; arow[3] = a0.w; arow[4] = a1.x; arow[5] = a1.y; arow[6] = a1.z; arow[7] = a1.w;
            }
            if (MODE >= 6) {
                typedef float f4 __attribute__((ext_vector_type(4)));
                f4 c[2][4];
#pragma unroll
                for (int h = 0; h < 2; h++)
#pragma unroll
                    for (int q = 0; q < 4; q++) c[h][q] = f4{0.f, 0.f, 0.f, 0.f};
                const float* al = ap + (lane & 3);
                if (MODE == 6) {
                    for (int kb = 0; kb < kper; kb += GL) {
                        float4 wv[GL];
#pragma unroll
                        for (int i = 0; i < GL; i++) wv[i] = *reinterpret_cast<const float4*>(wp + (long long)(kb + i) * VP);
#pragma unroll
                        for (int i = 0; i < GL; i++) {
                            const int k = kb + i;
                            const float alo = al[k * GF], ahi = al[k * GF + 4];
                            c[0][0] = __builtin_amdgcn_mfma_f32_4x4x1f32(alo, wv[i].x, c[0][0], 0, 0, 0);
                            c[0][1] = __builtin_amdgcn_mfma_f32_4x4x1f32(alo, wv[i].y, c[0][1], 0, 0, 0);
                            c[0][2] = __builtin_amdgcn_mfma_f32_4x4x1f32(alo, wv[i].z, c[0][2], 0, 0, 0);
                            c[0][3] = __builtin_amdgcn_mfma_f32_4x4x1f32(alo, wv[i].w, c[0][3], 0, 0, 0);
                            c[1][0] = __builtin_amdgcn_mfma_f32_4x4x1f32(ahi, wv[i].x, c[1][0], 0, 0, 0);
                            c[1][1] = __builtin_amdgcn_mfma_f32_4x4x1f32(ahi, wv[i].y, c[1][1], 0, 0, 0);
                            c[1][2] = __builtin_amdgcn_mfma_f32_4x4x1f32(ahi, wv[i].z, c[1][2], 0, 0, 0);
                            c[1][3] = __builtin_amdgcn_mfma_f32_4x4x1f32(ahi, wv[i].w, c[1][3], 0, 0, 0);
                        }
                    }
                } else {
                    constexpr int RG = MODE == 7 ? 8 : 16;   // (modes 9 / 10: no weight requests / no activation reads either -- the matrix pipe alone)
                    float4 wv[RG];
#pragma unroll
                    for (int i = 0; i < RG; i++) wv[i] = *reinterpret_cast<const float4*>(wp + (long long)i * VP);
                    float anext[2] = {al[0], al[4]};
                    auto rows = [&](int kb, auto reload) {
#pragma unroll
                        for (int i = 0; i < RG; i++) {
                            const int k = kb + i;
                            const float alo = anext[0], ahi = anext[1];
                            const int kn = min(k + 1, kper - 1);
                            if (MODE < 10) { anext[0] = al[kn * GF]; anext[1] = al[kn * GF + 4]; }   // (the next row's activations, one row ahead)
                            else asm volatile("" : "+v"(anext[0]), "+v"(anext[1]));
                            const float4 wk = wv[i];
                            if (decltype(reload)::value && MODE < 9) wv[i] = *reinterpret_cast<const float4*>(wp + (long long)(k + RG) * VP);
                            if (MODE >= 9) asm volatile("" : "+v"(wv[i].x), "+v"(wv[i].y));
                            c[0][0] = __builtin_amdgcn_mfma_f32_4x4x1f32(alo, wk.x, c[0][0], 0, 0, 0);
                            c[0][1] = __builtin_amdgcn_mfma_f32_4x4x1f32(alo, wk.y, c[0][1], 0, 0, 0);
                            c[0][2] = __builtin_amdgcn_mfma_f32_4x4x1f32(alo, wk.z, c[0][2], 0, 0, 0);
                            c[0][3] = __builtin_amdgcn_mfma_f32_4x4x1f32(alo, wk.w, c[0][3], 0, 0, 0);
                            c[1][0] = __builtin_amdgcn_mfma_f32_4x4x1f32(ahi, wk.x, c[1][0], 0, 0, 0);
                            c[1][1] = __builtin_amdgcn_mfma_f32_4x4x1f32(ahi, wk.y, c[1][1], 0, 0, 0);
                            c[1][2] = __builtin_amdgcn_mfma_f32_4x4x1f32(ahi, wk.z, c[1][2], 0, 0, 0);
                            c[1][3] = __builtin_amdgcn_mfma_f32_4x4x1f32(ahi, wk.w, c[1][3], 0, 0, 0);
                            __builtin_amdgcn_sched_barrier(0);   // (the scheduler otherwise sinks the ring's requests to just before their use)
                        }
                    };
                    int kb = 0;
                    for (; kb + RG < kper; kb += RG) rows(kb, std::true_type{});
                    rows(kb, std::false_type{});
                }
#pragma unroll
                for (int h = 0; h < 2; h++)
#pragma unroll
                    for (int q = 0; q < 4; q++)
#pragma unroll
                        for (int i = 0; i < 4; i++) acc[4 * h + i][q] = c[h][q][i];
            } else
            for (int kb = 0; kb < kper; kb += GL) {
                float4 wv[GL];
#pragma unroll
                for (int i = 0; i < GL; i++) wv[i] = *reinterpret_cast<const float4*>(wp + (long long)(kb + i) * VP);
#pragma unroll
                for (int i = 0; i < GL; i++) {
                    const int k = kb + i;
                    float av[GF];
                    if (MODE == 4) {
                        const float4 a0 = *reinterpret_cast<const float4*>(ap + k * GF), a1 = *reinterpret_cast<const float4*>(ap + k * GF + 4);
                        av[0] = a0.x; av[1] = a0.y; av[2] = a0.z; av[3] = a0.w; av[4] = a1.x; av[5] = a1.y; av[6] = a1.z; av[7] = a1.w;
                    } else {
#pragma unroll
                        for (int f = 0; f < GF; f++) av[f] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, arow[f]), k));
                    }
#pragma unroll
                    for (int f = 0; f < GF; f++) {
                        acc[f][0] += av[f] * wv[i].x; acc[f][1] += av[f] * wv[i].y; acc[f][2] += av[f] * wv[i].z; acc[f][3] += av[f] * wv[i].w;
                    }
                }
            }
            // the 8 waves' partial sums meet in LDS: part[wave][f][4 lane .. 4 lane + 3]
#pragma unroll
            for (int f = 0; f < GF; f++)
                *reinterpret_cast<float4*>(part + (wave * GF + f) * 256 + 4 * lane) = make_float4(acc[f][0], acc[f][1], acc[f][2], acc[f][3]);
            __syncthreads();
            // wave f sums frame f's 256 columns over the 8 slices (in slice order) and would take their argmax
            float4 s = *reinterpret_cast<const float4*>(part + (0 * GF + wave) * 256 + 4 * lane);
#pragma unroll
            for (int q = 1; q < 8; q++) {
                const float4 p = *reinterpret_cast<const float4*>(part + (q * GF + wave) * 256 + 4 * lane);
                s.x += p.x; s.y += p.y; s.z += p.z; s.w += p.w;
            }
            keep += s.x + s.y + s.z + s.w;
            if (r == 0 && dump && blockIdx.x < 2) *reinterpret_cast<float4*>(dump + (blockIdx.x * GF + wave) * 256 + 4 * lane) = s;
            __syncthreads();
        }
    }
    out[blockIdx.x * GT + tid] = keep;
}

template <int MODE>
int run(const float* W, const float* act, float* out, int nwg, float* dump = nullptr) {
    const int R = 200;
    const size_t lds = sizeof(float) * (J * GF + 8 * APAD + 8 * GF * 256);
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_probe<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL(k_probe<MODE>, dim3(nwg), dim3(GT), lds, 0, W, act, out, 5, dump);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(k_probe<MODE>, dim3(nwg), dim3(GT), lds, 0, W, act, out, R, (float*)nullptr);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    std::vector<float> h(GT);
    CK(hipMemcpy(h.data(), out, GT * 4, hipMemcpyDeviceToHost));
    printf("mode %d, %3d workgroups: %.2f us per sweep (checksum %.6g)\n", MODE, nwg, ms * 1e3 / R, (double)h[0] + h[100] + h[511]);
    return 0;
}

int main() {
    float *W, *act, *out;
    std::vector<float> h((size_t)J * VP), ha(J * GF);
    for (size_t i = 0; i < h.size(); i++) h[i] = (float)(((i * 2654435761ull) >> 7) % 100003) * 1e-5f - 0.5f;
    for (size_t i = 0; i < ha.size(); i++) ha[i] = (float)(((i * 40503ull) >> 3) % 10007) * 1e-4f - 0.5f;
    CK(hipMalloc(&W, h.size() * 4 + 4096));
    CK(hipMalloc(&act, ha.size() * 4));
    CK(hipMalloc(&out, 256 * GT * 4));
    float *d4, *d6;
    CK(hipMalloc(&d4, 2 * GF * 256 * 4));
    CK(hipMalloc(&d6, 2 * GF * 256 * 4));
    CK(hipMemcpy(W, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(act, ha.data(), ha.size() * 4, hipMemcpyHostToDevice));
    for (int nwg : {64}) {
        if (run<0>(W, act, out, nwg) || run<1>(W, act, out, nwg) || run<2>(W, act, out, nwg) || run<3>(W, act, out, nwg) ||
            run<4>(W, act, out, nwg, d4) || run<5>(W, act, out, nwg) || run<6>(W, act, out, nwg, d6) || run<7>(W, act, out, nwg) || run<8>(W, act, out, nwg, d6) || run<9>(W, act, out, nwg) || run<10>(W, act, out, nwg)) return 1;
    }
    std::vector<unsigned> h4(2 * GF * 256), h6(2 * GF * 256);
    CK(hipMemcpy(h4.data(), d4, h4.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(h6.data(), d6, h6.size() * 4, hipMemcpyDeviceToHost));
    size_t diff = 0;
    for (size_t i = 0; i < h4.size(); i++) diff += h4[i] != h6[i];
    printf("mode 8 (and 6) vs mode 4: %zu of %zu sums differ in their bits\n", diff, h4.size());
    return 0;
}
