// CPU stand-in for k2hip::Engine + the handful of HIP entry points the host layer names -- TEST INFRASTRUCTURE for the
// AddressSanitizer / UBSan build of csrc/api.cpp (`make -C k2transducerasr_amd/csrc san`, tests/test_sanitizers.py).
//
// GPU sanitizers do not exist on this pool, but everything api.cpp owns is host logic: the OfflineStream / OnlineStream mirrors
// (feature FIFO, lazy fbank of pending samples, RemoveChunk, IsFinished, the device-mirror bookkeeping, poisoning after a failed
// step), argument checking and the error transport across the C ABI.  This file replaces the device work behind it by
// deterministic fakes with the SAME memory contract (every output buffer is written exactly as far as the real engine writes it),
// so that api.cpp itself -- unmodified -- runs under the sanitizers.  The container is still parsed by the real Model class
// (csrc/model.cpp, host only); nothing is uploaded.  Never linked into libk2hip.so.
#include <cstdlib>
#include <cstring>

#include "../../k2transducerasr_amd/csrc/engine.h"

// ---- HIP entry points referenced by api.cpp / model.cpp / common.h (host allocations stand in for device memory) ---------------
extern "C" {
hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
const char* hipGetErrorString(hipError_t) { return "stub"; }
hipError_t hipMalloc(void** p, size_t n) { *p = malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipFree(void* p) { free(p); return hipSuccess; }
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = reinterpret_cast<hipStream_t>(malloc(1)); return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t, int) { *v = 256; return hipSuccess; }
}

namespace k2hip {

namespace {
// test hook of the stand-in: the next online_step / online_encoder throws (a "device failure"), see san_api_driver.cpp
int g_fail_next_step = 0;
int g_fail_next_gather_finish = 0;   // the next collection of a deferred fbank download fails (fbank_gather_finish)
unsigned mix(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
}  // namespace
extern "C" void k2hip_stub_fail_next_step(int n) { g_fail_next_step = n; }
extern "C" void k2hip_stub_fail_next_gather_finish(int n) { g_fail_next_gather_finish = n; }

Engine::Engine(const std::string& weights, const char* overrides, int device) : device_(device) {
    tunables_init_from_env();
    model_.reset(new Model(weights, overrides));   // real host-side parse + validation; no upload
    if (device != 0) failf(K2HIP_ERR_NO_DEVICE, "device %d out of range (have 1)", device);
    tunables_init_from_env();
}
Engine::~Engine() {}

int Engine::encoder_out_frames(int T) const {
    const int T50 = (T - 7) / 2;
    return T50 <= 0 ? 0 : (T50 + 1) / 2;
}
int64_t Engine::fbank_num_frames(int64_t n) const {
    const FbankOpts& f = model_->cfg().fbank;
    return n < f.frame_len ? 0 : 1 + (n - f.frame_len) / f.frame_shift;
}
int Engine::online_frames_per_chunk() const { return (model_->cfg().shift / 2 + 1) / 2; }

// frame values: a function of the frame's first sample, so that the driver can predict them
static void fake_frames(const float* s, int64_t n, const FbankOpts& f, int feat, float* out, int64_t nf) {
    (void)n;
    for (int64_t t = 0; t < nf; t++)
        for (int c = 0; c < feat; c++) out[t * feat + c] = s[t * f.frame_shift] * 0.5f + (float)c;
}
void Engine::fbank_host(const float* samples, int64_t n, float* feats, int64_t cap_frames, int64_t* n_frames) {
    const int64_t nf = fbank_num_frames(n);
    if (nf > cap_frames) failf(K2HIP_ERR_CAPACITY, "fbank: %lld frames exceed capacity %lld", (long long)nf, (long long)cap_frames);
    *n_frames = nf;
    fake_frames(samples, n, model_->cfg().fbank, model_->cfg().feat, feats, nf);
}
void Engine::fbank_host_gather(const float* const* head, const int64_t* n_head, const float* const* tail, const int64_t* n_tail, int64_t n, int G,
                               float* const* dst, int64_t nf, const int* fifo_slot, const int* fifo_pos, bool defer) {
    (void)fifo_slot; (void)fifo_pos; (void)defer;   // (the stand-in fills dst at once: a deferred gather with nothing left to finish)
    std::vector<float> cat((size_t)n);
    for (int g = 0; g < G; g++) {
        K2_REQUIRE(n_head[g] + n_tail[g] == n, "stub: gather lengths");
        if (n_head[g]) memcpy(cat.data(), head[g], sizeof(float) * (size_t)n_head[g]);
        if (n_tail[g]) memcpy(cat.data() + n_head[g], tail[g], sizeof(float) * (size_t)n_tail[g]);
        fake_frames(cat.data(), n, model_->cfg().fbank, model_->cfg().feat, dst[g], nf);
    }
}
void Engine::fbank_gather_finish() {
    if (g_fail_next_gather_finish > 0) {
        g_fail_next_gather_finish--;
        failf(K2HIP_ERR_HIP, "stub: the deferred fbank download failed");
    }
}
void Engine::pad_host(const float* const* speech, const int64_t* n_floats, int B, int tail, float* out, int64_t cap, int64_t* Lout) {
    int64_t mx = 0;
    for (int b = 0; b < B; b++) mx = std::max(mx, n_floats[b]);
    const int64_t L = mx + 80 * tail;
    *Lout = L;
    if (!out) return;
    if ((int64_t)B * L > cap) failf(K2HIP_ERR_CAPACITY, "pad: capacity");
    for (int b = 0; b < B; b++)
        for (int64_t i = 0; i < L; i++) {
            const float v = i < n_floats[b] ? speech[b][i] : 0.f;
            out[b * L + i] = v == 0.f ? -23.025850929940457f : v;
        }
}
void Engine::encoder_host(const float* x, int B, int T, float* enc_out, int64_t cap, int* Tp) {
    (void)x;
    const int tp = encoder_out_frames(T), J = model_->cfg().enc_dim();
    if ((int64_t)B * tp * J > cap) failf(K2HIP_ERR_CAPACITY, "encoder: capacity");
    *Tp = tp;
    for (int64_t i = 0; i < (int64_t)B * tp * J; i++) enc_out[i] = 0.25f;
}
void Engine::encoder_tap_host(const float*, int B, int T, int, float* out, int64_t cap, int64_t* n) {
    *n = (int64_t)B * T;
    if (out && *n <= cap) memset(out, 0, sizeof(float) * (size_t)*n);
}
void Engine::decoder_host(const int64_t* y, int N, float* dec_out) {
    const Config& c = model_->cfg();
    for (int i = 0; i < N; i++)
        for (int j = 0; j < c.J; j++) dec_out[(size_t)i * c.J + j] = (float)(y[(size_t)i * c.ctx] + y[(size_t)i * c.ctx + c.ctx - 1]);
}
void Engine::joiner_host(const float* enc, const float* dec, int N, float* logits) {
    const Config& c = model_->cfg();
    for (int i = 0; i < N; i++)
        for (int v = 0; v < c.V; v++) logits[(size_t)i * c.V + v] = enc[(size_t)i * c.J] + dec[(size_t)i * c.J] + (float)v;
}
// fake search: stream b emits token 3 + (b + t) % 5 on every third frame
static void fake_search(int B, int Tp, int64_t* tokens, int32_t* ts, int32_t* n_tokens, int max_tokens) {
    for (int b = 0; b < B; b++) {
        int n = 0;
        for (int t = b % 3; t < Tp; t += 3) {
            if (n >= max_tokens) failf(K2HIP_ERR_CAPACITY, "a stream emitted more than max_tokens=%d symbols", max_tokens);
            tokens[(size_t)b * max_tokens + n] = 3 + (b + t) % 5;
            ts[(size_t)b * max_tokens + n] = t;
            n++;
        }
        n_tokens[b] = n;
    }
}
void Engine::greedy_host(const float*, int B, int Tp, bool, int64_t* tokens, int32_t* ts, int32_t* n_tokens, int max_tokens) {
    fake_search(B, Tp, tokens, ts, n_tokens, max_tokens);
}
void Engine::offline_greedy_feats(const float* const* feats, const int64_t* n_floats, int B, bool, int64_t* tokens, int32_t* ts, int32_t* n_tokens,
                                  int max_tokens) {
    int64_t mx = 0;
    volatile float sink = 0.f;
    for (int b = 0; b < B; b++) {
        mx = std::max(mx, n_floats[b]);
        if (n_floats[b]) sink = sink + feats[b][0] + feats[b][n_floats[b] - 1];   // both ends must be readable
    }
    fake_search(B, encoder_out_frames((int)(mx / model_->cfg().feat) + 19), tokens, ts, n_tokens, max_tokens);
}
void Engine::offline_greedy_samples(const float* const* samples, const int64_t* n_samples, int B, int64_t* tokens, int32_t* ts, int32_t* n_tokens,
                                    int max_tokens, bool, bool) {
    int64_t mx = 0;
    volatile float sink = 0.f;
    for (int b = 0; b < B; b++) {
        mx = std::max(mx, fbank_num_frames(n_samples[b]));
        if (n_samples[b]) sink = sink + samples[b][0] + samples[b][n_samples[b] - 1];
    }
    fake_search(B, encoder_out_frames((int)mx + 19), tokens, ts, n_tokens, max_tokens);
}
void Engine::offline_greedy_samples_dev(const float* s, int64_t n_each, int B, int64_t* tokens, int32_t* ts, int32_t* n_tokens, int max_tokens) {
    volatile float sink = s[0] + s[(size_t)B * n_each - 1];
    (void)sink;
    fake_search(B, encoder_out_frames((int)fbank_num_frames(n_each) + 19), tokens, ts, n_tokens, max_tokens);
}
int Engine::submit_samples_dev(const float* s, int64_t n_each, int B, int max_tokens) {
    for (int k = 0; k < 2; k++)
        if (!slots_[k].busy) {
            slots_[k].busy = true;
            slots_[k].B = B;
            slots_[k].max_tokens = max_tokens;
            slots_[k].pin_cap = fbank_num_frames(n_each);
            (void)s;
            return k;
        }
    failf(K2HIP_ERR_INVALID, "offline_submit: 2 batches already in flight; wait for one first");
}
int Engine::submit_samples_host(const float* s, int64_t n_each, int B, int max_tokens) { return submit_samples_dev(s, n_each, B, max_tokens); }
void Engine::wait_ticket(int ticket, int64_t* tokens, int32_t* ts, int32_t* n_tokens) {
    K2_REQUIRE(ticket >= 0 && ticket < kSlots && slots_[ticket].busy, "offline_wait: ticket %d is not in flight", ticket);
    Slot& sl = slots_[ticket];
    sl.busy = false;
    fake_search(sl.B, encoder_out_frames((int)sl.pin_cap + 19), tokens, ts, n_tokens, sl.max_tokens);
    sl.pin_cap = 0;
}
float Engine::debug_gemm(int, int, int, int, bool, int, int, float* e) { if (e) *e = 0.f; return 0.f; }
void Engine::decoder_table_check(int, unsigned, long long* rows, long long* mismatched) { *rows = 0; *mismatched = 0; }
void Engine::debug_gemm_trace(int, int, int, int, bool, int, unsigned long long*, int64_t, int* a, int* b) { *a = 0; *b = 0; }
void Engine::debug_gemm_host(const float*, const float*, const float*, const float*, float*, int, int, int, int, int, int, int) {}
void* Engine::dev_alloc(int64_t bytes) { return malloc((size_t)std::max<int64_t>(bytes, 1)); }
void Engine::dev_free(void* p) { free(p); }
void Engine::dev_upload(void* dst, const void* src, int64_t bytes) { memcpy(dst, src, (size_t)bytes); }
void* Engine::host_alloc(int64_t bytes) { return malloc((size_t)std::max<int64_t>(bytes, 1)); }
void Engine::host_free(void* p) { free(p); }
void Engine::synchronize() {}

// ---- streaming -----------------------------------------------------------------------------------------------------------------
int Engine::online_alloc_slot() {
    if (!free_slots_.empty()) {
        const int s = free_slots_.back();
        free_slots_.pop_back();
        return s;
    }
    return online_cap_++;
}
void Engine::online_free_slot(int slot) {
    if (slot >= 0) free_slots_.push_back(slot);
}
void Engine::online_fifo_write(int slot, int pos, const float* feats, int64_t n_frames) {
    K2_REQUIRE(slot >= 0 && pos >= 0 && pos < kFifoFrames && n_frames >= 0 && n_frames <= kFifoFrames, "stub: fifo_write(%d, %d, %lld)", slot, pos,
               (long long)n_frames);
    volatile float sink = 0.f;
    if (n_frames) sink = feats[0] + feats[(size_t)n_frames * model_->cfg().feat - 1];   // the whole block must be readable
    (void)sink;
}
void Engine::online_read_state(int, int, int, long long, float* out, int64_t cap, int64_t* n) {
    *n = 16;
    if (out && cap >= 16) memset(out, 0, sizeof(float) * 16);
}
void Engine::online_step(const int* slots, const float* const* chunks, const long long* hyps, const long long* plens, const int* nchunks, int B,
                         int64_t* tokens, int32_t* ts, int32_t* n_tokens, const int* fifo_heads) {
    if (g_fail_next_step > 0) {
        g_fail_next_step--;
        failf(K2HIP_ERR_HIP, "stub: device failure in the chunk step");
    }
    const Config& c = model_->cfg();
    const int Tp = online_frames_per_chunk();
    for (int b = 0; b < B; b++) {
        K2_REQUIRE(slots[b] >= 0 && slots[b] < online_cap_, "stub: slot %d", slots[b]);
        // the whole chunk must be readable (GetDecodeChunk hands over ChunkLength frames)
        volatile float sink = chunks[b][0] + chunks[b][(size_t)c.chunk_T * c.feat - 1];
        (void)sink;
        if (fifo_heads) K2_REQUIRE(fifo_heads[b] >= 0 && fifo_heads[b] < kFifoFrames, "stub: fifo head %d", fifo_heads[b]);
        const unsigned h = mix((unsigned)(nchunks[b] * 31 + (int)plens[b] + (int)hyps[2 * b + 1]));
        const int n = (int)(h % 3u);   // 0 .. 2 tokens in this chunk (<= Tp)
        for (int k = 0; k < n && k < Tp; k++) {
            tokens[(size_t)b * Tp + k] = 3 + (h >> (4 * k)) % 7;
            ts[(size_t)b * Tp + k] = 2 * k;
        }
        n_tokens[b] = std::min(n, Tp);
    }
}
void Engine::online_encoder(const int* slots, const float* feats, const long long*, const int*, int B, float* enc_out) {
    if (g_fail_next_step > 0) {
        g_fail_next_step--;
        failf(K2HIP_ERR_HIP, "stub: device failure in EncoderProj");
    }
    const Config& c = model_->cfg();
    const int Tp = online_frames_per_chunk();
    volatile float sink = feats[0] + feats[(size_t)B * c.chunk_T * c.feat - 1];
    (void)sink;
    for (int b = 0; b < B; b++) K2_REQUIRE(slots[b] >= 0 && slots[b] < online_cap_, "stub: slot %d", slots[b]);
    for (int64_t i = 0; i < (int64_t)B * Tp * c.enc_dim(); i++) enc_out[i] = 0.5f;
}

}  // namespace k2hip
