// Probe: what does a chain of DEPENDENT kernel launches cost per launch on one HIP stream -- enqueued eagerly, replayed as a captured
// hipGraph, or walked by ONE persistent launch with a grid-wide barrier between the steps?  The encoder pass is such a chain
// (~400 launches per offline batch, ~385 per streaming tick); profiles/r03_* put ~2.4 us between two launches and a ~4.6 us floor
// under a small launch.  Kernel body: `work` rounds of FMAs per thread over a buffer the next launch reads (so the launches really
// depend on each other), grids of 1 x 256 .. 1024 x 256 threads.
// hipcc --offload-arch=gfx950 -O3 graph_chain_probe.hip -o /tmp/graph_chain_probe && /tmp/graph_chain_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

__global__ void k_step(const float* __restrict__ in, float* __restrict__ out, int n, int work) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    float v = in[(i + 1) % n];
    for (int r = 0; r < work; r++) v = fmaf(v, 1.0001f, 0.5f);
    out[i] = v;
}

// the same chain inside ONE launch: every workgroup does its slice of step s, then all meet at a counter barrier (agent scope)
__global__ void k_chain(float* a, float* b, int n, int work, int steps, unsigned* bar) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    for (int s = 0; s < steps; s++) {
        const float* in = (s & 1) ? b : a;
        float* out = (s & 1) ? a : b;
        float v = __hip_atomic_load(&in[(i + 1) % n], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int r = 0; r < work; r++) v = fmaf(v, 1.0001f, 0.5f);
        __hip_atomic_store(&out[i], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned want = (unsigned)(s + 1) * gridDim.x;
            while (__hip_atomic_load(bar, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < want) __builtin_amdgcn_s_sleep(1);
        }
        __syncthreads();
    }
}

int main() {
    const int steps = 400;
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    printf("%8s %6s | %10s %10s %10s   (us per dependent step, %d steps)\n", "blocks", "work", "eager", "graph", "persistent", steps);
    for (int blocks : {1, 64, 256, 1024}) {
        for (int work : {1, 2000, 8000}) {
            const int n = blocks * 256;
            float *a, *b;
            unsigned* bar;
            CK(hipMalloc(&a, n * 4));
            CK(hipMalloc(&b, n * 4));
            CK(hipMalloc(&bar, 4));
            CK(hipMemset(a, 0, n * 4));
            auto chain = [&]() {
                for (int s = 0; s < steps; s++) hipLaunchKernelGGL(k_step, dim3(blocks), dim3(256), 0, st, (s & 1) ? b : a, (s & 1) ? a : b, n, work);
            };
            float ms_e = 0, ms_g = 0, ms_p = 0;
            chain();   // warm
            CK(hipStreamSynchronize(st));
            for (int rep = 0; rep < 3; rep++) {
                CK(hipEventRecord(e0, st));
                chain();
                CK(hipEventRecord(e1, st));
                CK(hipStreamSynchronize(st));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                ms_e = rep == 0 ? ms : (ms < ms_e ? ms : ms_e);
            }
            hipGraph_t g;
            hipGraphExec_t ge;
            CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
            chain();
            CK(hipStreamEndCapture(st, &g));
            CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            CK(hipGraphLaunch(ge, st));
            CK(hipStreamSynchronize(st));
            for (int rep = 0; rep < 3; rep++) {
                CK(hipEventRecord(e0, st));
                CK(hipGraphLaunch(ge, st));
                CK(hipEventRecord(e1, st));
                CK(hipStreamSynchronize(st));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                ms_g = rep == 0 ? ms : (ms < ms_g ? ms : ms_g);
            }
            CK(hipGraphExecDestroy(ge));
            CK(hipGraphDestroy(g));
            if (blocks <= 256) {   // co-resident grids only (one workgroup per CU at most)
                for (int rep = 0; rep < 3; rep++) {
                    CK(hipMemsetAsync(bar, 0, 4, st));
                    CK(hipEventRecord(e0, st));
                    hipLaunchKernelGGL(k_chain, dim3(blocks), dim3(256), 0, st, a, b, n, work, steps, bar);
                    CK(hipEventRecord(e1, st));
                    CK(hipStreamSynchronize(st));
                    float ms;
                    CK(hipEventElapsedTime(&ms, e0, e1));
                    ms_p = rep == 0 ? ms : (ms < ms_p ? ms : ms_p);
                }
            }
            printf("%8d %6d | %10.2f %10.2f %10.2f\n", blocks, work, ms_e * 1e3 / steps, ms_g * 1e3 / steps, ms_p * 1e3 / steps);
            CK(hipFree(a));
            CK(hipFree(b));
            CK(hipFree(bar));
        }
    }
    return 0;
}
