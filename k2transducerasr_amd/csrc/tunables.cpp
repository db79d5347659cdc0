// Development switches of libk2hip.  They select alternative kernels for the same math (used by the parity tests to compare
// both paths) or tuning variants.  The K2HIP_* environment is read exactly ONCE, when the first model of the process is created;
// after that the launch paths only read this struct (no libc lookups per launch, no behaviour change from a setenv mid-run).
// k2hip_debug_set_switch is the test-only way to flip one afterwards.
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "kernels.h"

namespace k2hip {

// see common.h: blocking transfers on a per-device utility stream.  One mutex and one stream PER DEVICE: handles on different GPUs
// (one host thread each, INTEGRATION.md "More than one GPU") never wait for each other's copies -- a model upload on device 3 does
// not stall a streaming tick on device 0.
namespace {
constexpr int kMaxDev = 64;
std::mutex g_util_mu[kMaxDev];
hipStream_t g_util_stream[kMaxDev] = {};
hipError_t util_device(int* dev) {
    hipError_t e = hipGetDevice(dev);
    if (e != hipSuccess) return e;
    return (*dev < 0 || *dev >= kMaxDev) ? hipErrorInvalidDevice : hipSuccess;
}
hipError_t util_stream(int dev, hipStream_t* s) {   // g_util_mu[dev] held
    if (!g_util_stream[dev]) {
        hipError_t e = hipStreamCreateWithFlags(&g_util_stream[dev], hipStreamNonBlocking);
        if (e != hipSuccess) return e;
    }
    *s = g_util_stream[dev];
    return hipSuccess;
}
}  // namespace
hipError_t copy_blocking(void* dst, const void* src, size_t bytes, hipMemcpyKind kind) {
    if (bytes == 0) return hipSuccess;
    int dev = 0;
    hipError_t e = util_device(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lk(g_util_mu[dev]);
    hipStream_t s = nullptr;
    e = util_stream(dev, &s);
    if (e != hipSuccess) return e;
    e = hipMemcpyAsync(dst, src, bytes, kind, s);
    if (e != hipSuccess) return e;
    return hipStreamSynchronize(s);
}
hipError_t fill_blocking(void* dst, int value, size_t bytes) {
    if (bytes == 0) return hipSuccess;
    int dev = 0;
    hipError_t e = util_device(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lk(g_util_mu[dev]);
    hipStream_t s = nullptr;
    e = util_stream(dev, &s);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(dst, value, bytes, s);
    if (e != hipSuccess) return e;
    return hipStreamSynchronize(s);
}

namespace {
Tunables g_t;
bool g_init = false;
struct Entry {
    const char* env;
    int Tunables::*field;
    bool flag;  // presence of the variable means 1
};
const Entry kEntries[] = {
    {"K2HIP_NO_GLU_EPILOGUE", &Tunables::no_glu_epilogue, true},
    {"K2HIP_ATTN_LONG", &Tunables::attn_long, true},
    {"K2HIP_NO_FUSED_AV", &Tunables::no_fused_av, true},
    {"K2HIP_NO_FUSED_VPROJ", &Tunables::no_fused_vproj, true},
    {"K2HIP_FUSED_VPROJ_MIN_T", &Tunables::fused_vproj_min_t, false},
    {"K2HIP_NO_FUSED_CONV", &Tunables::no_fused_conv, true},
    {"K2HIP_CONFORMER_GEMM_SCORES", &Tunables::conformer_gemm_scores, true},
    {"K2HIP_DW7_TILED", &Tunables::dw7_tiled, true},
    {"K2HIP_LSTM_SEQ", &Tunables::lstm_seq, true},
    {"K2HIP_GREEDY_ONE_PART", &Tunables::greedy_one_part, true},
    {"K2HIP_GREEDY_PARTS", &Tunables::greedy_parts, false},
    {"K2HIP_DECODER_TABLE_MB", &Tunables::decoder_table_mb, false},
    {"K2HIP_BEAM_LAUNCHES", &Tunables::beam_launches, true},
    {"K2HIP_BEAM_HYP_GLOBAL", &Tunables::beam_hyp_global, true},
    {"K2HIP_BEAM_PARTS", &Tunables::beam_parts, false},
    {"K2HIP_BEAM_TRACE", &Tunables::beam_trace, true},
    {"K2HIP_SCREEN_MIN_V", &Tunables::screen_min_v, false},
    {"K2HIP_TEST_GREEDY_TIMEOUT", &Tunables::test_greedy_timeout, true},
    {"K2HIP_SEARCH_ROUNDS", &Tunables::search_rounds, false},
    {"K2HIP_MAX_STREAMS", &Tunables::max_streams, false},
#ifdef K2HIP_DEV   // tuning probes: a -DK2HIP_DEV build only (make DEV=1)
    {"K2HIP_GEMM_CFG", &Tunables::gemm_cfg, false},
    {"K2HIP_XCD_PANELS", &Tunables::xcd_panels, false},
    {"K2HIP_DW1D_TT", &Tunables::dw1d_tt, false},
    {"K2HIP_GREEDY_STAMPS", &Tunables::greedy_stamps, true},
    {"K2HIP_CONFORMER_STAMPS", &Tunables::conformer_stamps, true},
#endif
};
}  // namespace

void tunables_init_from_env() {
    // once per process, whichever threads open the first handles (two at the same moment raced on the flag: ThreadSanitizer,
    // tests/native/tsan_api_driver.cpp)
    static std::once_flag once;
    std::call_once(once, [] {
    g_init = true;
    for (const Entry& e : kEntries)
        if (const char* v = getenv(e.env)) g_t.*(e.field) = e.flag ? 1 : atoi(v);
    });
}

const Tunables& tunables() { return g_t; }
bool tunables_set(const char* env_name, int value) {
    for (const Entry& e : kEntries)
        if (!strcmp(e.env, env_name)) {
            g_t.*(e.field) = value;
            return true;
        }
    return false;
}

}  // namespace k2hip
