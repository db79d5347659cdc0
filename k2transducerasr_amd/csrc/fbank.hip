// F1: kaldi-style log-mel filterbank on gfx950 (replaces WavFrontend.GetFbank ->
// SpeechFeatures.OnlineFbank.GetFbank, K2TransducerAsr/WavFrontend.cs:32-36).
//
// One workgroup (256 threads) per 25 ms frame: remove DC, pre-emphasis, window,
// 512-point radix-2 FFT in LDS, power spectrum, 80 triangular mel filters, floor at
// FLT_EPSILON, log.  Everything up to the log runs in f64: the whole front-end is
// ~1.5 GFLOP per 32 x 10 s batch, so f64 costs nothing, and it removes the f32
// round-off (a one-ulp difference in pre-processing moves low-energy mel bins by
// ~1e-4) that would otherwise be the largest difference between two correct
// implementations of the same front-end.
#include <cfloat>
#include <mutex>
#include <vector>

#include "kernels.h"

namespace k2hip {
namespace {

constexpr int NFFT = 512, NBIN = 256, LOG2N = 9;

__global__ __launch_bounds__(256) void k_fbank(FbankArgs a, const double2* __restrict__ tw, int num_mel) {
    __shared__ double re[NFFT], im[NFFT];
    __shared__ double red[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long f = blockIdx.x;
    const int u = blockIdx.y;
    const float* s = a.samples + (long long)u * a.utt_stride + f * a.frame_shift;
    const int N = a.frame_len;

    // two samples per thread: i0 = tid, i1 = tid + 256 (p = previous sample, for pre-emphasis)
    double x0 = 0.0, x1 = 0.0, p0 = 0.0, p1 = 0.0;
    const double scale = (double)a.input_scale;
    int i0 = tid, i1 = tid + 256;
    if (i0 < N) { x0 = (double)s[i0] * scale; p0 = (double)s[i0 > 0 ? i0 - 1 : 0] * scale; }
    if (i1 < N) { x1 = (double)s[i1] * scale; p1 = (double)s[i1 - 1] * scale; }
    if (a.remove_dc) {
        double part = x0 + x1;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
        if (lane == 0) red[wave] = part;
        __syncthreads();
        double mean = (red[0] + red[1] + red[2] + red[3]) / (double)N;
        x0 -= mean; x1 -= mean; p0 -= mean; p1 -= mean;
    }
    if (a.preemph != 0.f) {
        // kaldi: w[i] -= c*w[i-1] (i = N-1..1), w[0] -= c*w[0]
        x0 = x0 - (double)a.preemph * p0;
        x1 = x1 - (double)a.preemph * p1;
    }
    if (i0 < N) x0 *= (double)a.window[i0]; else x0 = 0.0;
    if (i1 < N) x1 *= (double)a.window[i1]; else x1 = 0.0;
    // bit-reversed scatter
    re[__brev((unsigned)i0) >> (32 - LOG2N)] = x0;
    re[__brev((unsigned)i1) >> (32 - LOG2N)] = x1;
    im[i0] = 0.0;
    im[i1] = 0.0;
    __syncthreads();
    // radix-2 DIT: stage with half-length hl: butterfly t pairs (i, i+hl), twiddle index k*(256/hl)
#pragma unroll
    for (int st = 0; st < LOG2N; st++) {
        int hl = 1 << st;
        int k = tid & (hl - 1);
        int i = ((tid >> st) << (st + 1)) + k;
        int j = i + hl;
        double2 w = tw[k << (LOG2N - 1 - st)];
        double xr = re[j] * w.x - im[j] * w.y, xi = re[j] * w.y + im[j] * w.x;
        double ar = re[i], ai = im[i];
        re[j] = ar - xr; im[j] = ai - xi;
        re[i] = ar + xr; im[i] = ai + xi;
        __syncthreads();
    }
    // power spectrum of bins 0..255 (the mel filters never reach the Nyquist bin)
    double pw = re[tid] * re[tid] + im[tid] * im[tid];
    __syncthreads();
    re[tid] = pw;
    __syncthreads();
    float* out = a.feats + ((long long)u * a.n_frames + f) * num_mel;
    // one lane per mel filter: a triangle spans a few FFT bins (2 .. ~30), summed in index order
    if (tid < num_mel) {
        const int lo = (int)a.melrange[2 * tid], hi = (int)a.melrange[2 * tid + 1];
        const float* w = a.melw + tid * NBIN;
        double e = 0.0;
        for (int k = lo; k < hi; k++) e += (double)w[k] * re[k];
        float ef = (float)e;
        if (ef < FLT_EPSILON) ef = FLT_EPSILON;
        out[tid] = logf(ef);
    }
}

// one table per device, shared by every model handle of the process: built under the lock and published only when its upload has
// completed (handles opened from several host threads reach this together -- a pointer published before the copy let a neighbour's
// first fbank read an empty table)
std::mutex g_tw_mu;
double2* g_tw[16] = {nullptr};

const double2* twiddles(int device) {
    if (device < 0 || device >= 16) failf(K2HIP_ERR_INVALID, "device index %d out of range", device);
    std::lock_guard<std::mutex> lk(g_tw_mu);
    if (!g_tw[device]) {
        std::vector<double2> h(NBIN);
        for (int k = 0; k < NBIN; k++) {
            double ang = -2.0 * M_PI * k / NFFT;
            h[k].x = cos(ang);
            h[k].y = sin(ang);
        }
        double2* d = nullptr;
        K2_HIP(hipMalloc(&d, sizeof(double2) * NBIN));
        if (hipError_t e = copy_blocking(d, h.data(), sizeof(double2) * NBIN, hipMemcpyHostToDevice); e != hipSuccess) {
            (void)hipFree(d);
            K2_HIP(e);
        }
        g_tw[device] = d;
    }
    return g_tw[device];
}

}  // namespace

void fbank(const Ctx& ctx, const FbankArgs& a) {
    if (a.n_frames <= 0 || a.n_utts <= 0) return;
    K2_REQUIRE(a.frame_len <= NFFT && a.frame_len > 256, "fbank: frame_len %d unsupported", a.frame_len);
    K2_REQUIRE(a.melrange != nullptr, "fbank: mel filter extents missing");
    if (ctx.dry) return;
    int dev = 0;
    K2_HIP(hipGetDevice(&dev));
    const double2* tw = twiddles(dev);
    dim3 grid((unsigned)a.n_frames, (unsigned)a.n_utts);
    hipLaunchKernelGGL(k_fbank, grid, dim3(256), 0, ctx.stream, a, tw, 80);
    K2_HIP(hipGetLastError());
}

// The native OfflineStreams' sample queues (pinned host memory) -> the dense [B, nmax] block the batched fbank reads.  One float4 per
// thread and trip; the reads are coalesced 16 B per lane out of host memory, the grid is sized so that enough of them are in flight
// for the link (a 32 x 10 s batch: 20 MB).
__global__ __launch_bounds__(256) void k_gather_samples(const float* const* __restrict__ src, const long long* __restrict__ n,
                                                        float* __restrict__ dst, long long nmax) {
    const int b = blockIdx.y;
    const float* s = src[b];
    const long long cnt = n[b];
    float* d = dst + (long long)b * nmax;
    const bool al = ((reinterpret_cast<unsigned long long>(s) | reinterpret_cast<unsigned long long>(d)) & 15) == 0;
    for (long long i4 = (long long)blockIdx.x * 256 + threadIdx.x; 4 * i4 < nmax; i4 += (long long)gridDim.x * 256) {
        const long long i = 4 * i4;
        if (al && i + 3 < cnt && i + 3 < nmax) {
            *reinterpret_cast<float4*>(d + i) = *reinterpret_cast<const float4*>(s + i);
        } else {
            for (int k = 0; k < 4 && i + k < nmax; k++) d[i + k] = i + k < cnt ? s[i + k] : 0.f;
        }
    }
}

void gather_samples(const Ctx& ctx, const float* const* src, const long long* n, float* dst, int B, long long nmax) {
    if (ctx.dry) return;
    K2_REQUIRE(B > 0 && nmax > 0, "gather_samples: empty batch");
    const int gx = (int)std::min<long long>(cdiv((int)std::min<long long>((nmax + 3) / 4, 1 << 30), 256), 64);   // 64 x B workgroups, ~10 trips each at 10 s
    hipLaunchKernelGGL(k_gather_samples, dim3(gx, B), dim3(256), 0, ctx.stream, src, n, dst, nmax);
    K2_HIP(hipGetLastError());
}

}  // namespace k2hip
