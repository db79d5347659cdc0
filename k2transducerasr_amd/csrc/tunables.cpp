// Development switches of libk2hip.  They select alternative kernels for the same math (used by the parity tests to compare
// both paths) or tuning variants.  The K2HIP_* environment is read exactly ONCE, when the first model of the process is created;
// after that the launch paths only read this struct (no libc lookups per launch, no behaviour change from a setenv mid-run).
// k2hip_debug_set_switch is the test-only way to flip one afterwards.
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "kernels.h"

namespace k2hip {

// see common.h: blocking transfers that stay off the legacy stream
namespace {
std::mutex g_util_mu;
hipStream_t g_util_stream[64] = {};
hipError_t util_stream(hipStream_t* s) {   // g_util_mu held
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (!g_util_stream[dev]) {
        e = hipStreamCreateWithFlags(&g_util_stream[dev], hipStreamNonBlocking);
        if (e != hipSuccess) return e;
    }
    *s = g_util_stream[dev];
    return hipSuccess;
}
}  // namespace
hipError_t copy_blocking(void* dst, const void* src, size_t bytes, hipMemcpyKind kind) {
    if (bytes == 0) return hipSuccess;
    std::lock_guard<std::mutex> lk(g_util_mu);
    hipStream_t s = nullptr;
    hipError_t e = util_stream(&s);
    if (e != hipSuccess) return e;
    e = hipMemcpyAsync(dst, src, bytes, kind, s);
    if (e != hipSuccess) return e;
    return hipStreamSynchronize(s);
}
hipError_t fill_blocking(void* dst, int value, size_t bytes) {
    if (bytes == 0) return hipSuccess;
    std::lock_guard<std::mutex> lk(g_util_mu);
    hipStream_t s = nullptr;
    hipError_t e = util_stream(&s);
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
    {"K2HIP_GEMM_CFG", &Tunables::gemm_cfg, false},
    {"K2HIP_GEMM_NO_DMA", &Tunables::gemm_no_dma, true},
    {"K2HIP_GEMM_NO_SKINNY", &Tunables::gemm_no_skinny, true},
    {"K2HIP_GEMM_NST3", &Tunables::gemm_nst3, true},
    {"K2HIP_GEMM_V1", &Tunables::gemm_v1, false},  // 1: everything older; 2: only the pipelined kernel off; 4: only the small ring tiles off
    {"K2HIP_NO_GLU_EPILOGUE", &Tunables::no_glu_epilogue, true},
    {"K2HIP_ATTN_LONG", &Tunables::attn_long, true},
    {"K2HIP_NO_FUSED_AV", &Tunables::no_fused_av, true},
    {"K2HIP_NO_FUSED_VPROJ", &Tunables::no_fused_vproj, true},
    {"K2HIP_XCD_PANELS", &Tunables::xcd_panels, false},
    {"K2HIP_FUSED_VPROJ_MIN_T", &Tunables::fused_vproj_min_t, false},
    {"K2HIP_CONFORMER_GEMM_SCORES", &Tunables::conformer_gemm_scores, true},
    {"K2HIP_CONFORMER_STRIP32", &Tunables::conformer_strip32, true},
    {"K2HIP_CONFORMER_SCATTER_V1", &Tunables::conformer_scatter_v1, true},
    {"K2HIP_CONFORMER_STAMPS", &Tunables::conformer_stamps, true},
    {"K2HIP_DW7_SIMPLE", &Tunables::dw7_simple, true},
    {"K2HIP_DW7_TILED", &Tunables::dw7_tiled, true},
    {"K2HIP_CAUSAL_CONV_LDS", &Tunables::causal_conv_lds, true},
    {"K2HIP_DW1D_TT", &Tunables::dw1d_tt, false},
    {"K2HIP_LSTM_SEQ", &Tunables::lstm_seq, true},
    {"K2HIP_GREEDY_ONE_PART", &Tunables::greedy_one_part, true},
    {"K2HIP_GREEDY_PARTS", &Tunables::greedy_parts, false},
    {"K2HIP_DECODER_TABLE_MB", &Tunables::decoder_table_mb, false},
    {"K2HIP_BEAM_LAUNCHES", &Tunables::beam_launches, true},
    {"K2HIP_BEAM_HYP_GLOBAL", &Tunables::beam_hyp_global, true},
    {"K2HIP_BEAM_PARTS", &Tunables::beam_parts, false},
    {"K2HIP_BEAM_TRACE", &Tunables::beam_trace, true},
    {"K2HIP_SCREEN_MIN_V", &Tunables::screen_min_v, false},
    {"K2HIP_GREEDY_STAMPS", &Tunables::greedy_stamps, true},
    {"K2HIP_NO_GRAPHS", &Tunables::no_graphs, true},
    {"K2HIP_GRAPH_CAPTURE_MODE", &Tunables::graph_capture_mode, false},
    {"K2HIP_GRAPH_OFFLINE", &Tunables::graph_offline, false},
    {"K2HIP_GRAPH_STREAMING", &Tunables::graph_streaming, false},
    {"K2HIP_TEST_GREEDY_TIMEOUT", &Tunables::test_greedy_timeout, true},
    {"K2HIP_SEARCH_ROUNDS", &Tunables::search_rounds, false},
    {"K2HIP_PIPE_MODE", &Tunables::pipe_mode, false},
    {"K2HIP_MAX_STREAMS", &Tunables::max_streams, false},
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
    // Under rocprofv3 (it announces itself with ROCP_TOOL_LIBRARIES) every launch stays eager: the profiler's kernel tracer of ROCm 7.2
    // segfaults inside the HIP runtime when a stream it traces is captured into / replayed from a hipGraph (seen on
    // bench_streaming.py; profiles/README.md).  A profile then shows the eager chain, which is the same kernels.
    if (getenv("ROCP_TOOL_LIBRARIES") && !getenv("K2HIP_GRAPHS_UNDER_PROFILER")) g_t.no_graphs = 1;
    });
}

const Tunables& tunables() { return g_t; }
namespace {
std::atomic<int> g_generation{0};
}
int tunables_generation() { return g_generation.load(std::memory_order_relaxed); }

bool tunables_set(const char* env_name, int value) {
    for (const Entry& e : kEntries)
        if (!strcmp(e.env, env_name)) {
            // a switch that changes what a chain of launches IS makes every recorded graph one of the old form (the switches that
            // only say whether graphs are used do not)
            if (g_t.*(e.field) != value && strncmp(env_name, "K2HIP_NO_GRAPHS", 15) && strncmp(env_name, "K2HIP_GRAPH", 11))
                g_generation.fetch_add(1, std::memory_order_relaxed);
            g_t.*(e.field) = value;
            return true;
        }
    return false;
}

}  // namespace k2hip
