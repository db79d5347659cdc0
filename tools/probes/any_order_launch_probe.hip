// Probe: does hipExtLaunchKernelGGL(..., hipExtAnyOrderLaunch) drop the queue's barrier between two launches on gfx950?  (hip_ext.h says
// the flag "is not supported on AMD GFX9xx boards".)  A chain of 400 launches of a ~30 us kernel over 64 workgroups -- a quarter of the
// chip, so that launches without a barrier between them WOULD overlap -- timed with and without the flag.  Same time = the flag is
// ignored: the next launch's dispatch cannot be moved under the previous kernel's tail from the API.
// hipcc --offload-arch=gfx950 -O3 any_order_launch_probe.hip -o /tmp/any_order_probe && /tmp/any_order_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

__global__ void k_work(float* __restrict__ out, int work) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    float v = (float)i;
    for (int r = 0; r < work; r++) v = fmaf(v, 1.0001f, 0.5f);
    out[i] = v;
}

int main(int argc, char** argv) {
    const int steps = 400, blocks = argc > 1 ? atoi(argv[1]) : 64, work = argc > 2 ? atoi(argv[2]) : 8000;
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float* buf;
    CK(hipMalloc(&buf, (size_t)steps * blocks * 256 * 4));
    for (int flag = 0; flag < 2; flag++) {
        float best = 0;
        for (int rep = 0; rep < 4; rep++) {
            CK(hipEventRecord(e0, st));
            for (int s = 0; s < steps; s++)
                hipExtLaunchKernelGGL(k_work, dim3(blocks), dim3(256), 0, st, nullptr, nullptr, flag ? hipExtAnyOrderLaunch : 0, buf + (size_t)s * blocks * 256, work);
            CK(hipEventRecord(e1, st));
            CK(hipStreamSynchronize(st));
            CK(hipGetLastError());
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep == 1 || (rep > 1 && ms < best)) best = ms;
        }
        printf("%s: %.2f us per launch (%d launches of %d workgroups)\n", flag ? "hipExtAnyOrderLaunch" : "in order            ", best * 1000.f / steps, steps, blocks);
    }
    return 0;
}
