// C ABI of libk2hip.so (include/k2hip.h) + the host-side mirror of the reference's
// OfflineStream / OfflineRecognizer bookkeeping (K2TransducerAsr/OfflineStream.cs,
// OfflineRecognizer.cs:77-91,289-296).
#include <algorithm>
#include <map>
#include <mutex>
#include <new>

#include "text.h"
#include "engine.h"
#include "../../include/k2hip_debug.h"

using namespace k2hip;

// The sample queue of an OfflineStream: a float array in PINNED host memory (Engine::host_alloc), which the GPU reads in place -- the
// batch's GetResults gathers the streams' samples straight out of these buffers (no staging copy on the host, no separate upload).
struct PinVec {
    Engine* e = nullptr;
    float* p = nullptr;
    size_t n = 0, cap = 0;
    PinVec() = default;
    PinVec(const PinVec&) = delete;
    PinVec& operator=(const PinVec&) = delete;
    PinVec(PinVec&& o) noexcept : e(o.e), p(o.p), n(o.n), cap(o.cap) { o.p = nullptr; o.n = o.cap = 0; }
    PinVec& operator=(PinVec&& o) noexcept {
        if (this != &o) {
            release();
            e = o.e; p = o.p; n = o.n; cap = o.cap;
            o.p = nullptr; o.n = o.cap = 0;
        }
        return *this;
    }
    ~PinVec() { release(); }
    void release() noexcept {
        if (p && e) {
            try { e->host_free(p); } catch (...) {}
        }
        p = nullptr;
        n = cap = 0;
    }
    const float* data() const { return p; }
    size_t size() const { return n; }
    bool empty() const { return n == 0; }
    size_t capacity() const { return cap; }
    void clear() { n = 0; }
    void append(Engine& eng, const float* src, size_t cnt) {
        if (n + cnt > cap) {   // (a stream normally gets ONE AddSamples call, and its buffer comes from the pool: growth is the rare path)
            const size_t want = std::max(n + cnt, cap + cap / 2);
            const size_t ncap = (want + 65535) & ~(size_t)65535;
            float* q = static_cast<float*>(eng.host_alloc((int64_t)(ncap * sizeof(float))));
            if (n) memcpy(q, p, n * sizeof(float));
            float* old = p;
            e = &eng; p = q; cap = ncap;
            if (old) {
                try { eng.host_free(old); } catch (...) {}   // (the queue is consistent either way)
            }
        }
        memcpy(p + n, src, cnt * sizeof(float));
        n += cnt;
    }
    void erase_front(size_t cnt) {
        if (cnt >= n) { n = 0; return; }
        memmove(p, p + cnt, (n - cnt) * sizeof(float));
        n -= cnt;
    }
};

struct k2hip_model {
    Engine engine;
    // Sample buffers of destroyed OfflineStreams, handed to the next streams (a GetResults batch is B x create / AddSamples / destroy
    // of ~640 KB each: pinned allocations cost a system call and a page-table update each, and fresh pages a first-touch fault per 4 KB
    // -- ~1 ms per 32 x 10 s batch on the host thread that drives the GPU).  At most kPoolMax buffers are kept.
    static constexpr size_t kPoolMax = 128;
    std::mutex wav_mu;
    std::vector<PinVec> wav_pool;   // (declared behind `engine`: released first)
    k2hip_model(const char* path, const char* ov, int dev) : engine(path, ov, dev) {}
};

// OfflineStream.cs:7-99
struct k2hip_offline_stream {
    k2hip_model* model;
    std::vector<float> speech;     // OfflineInputEntity.Speech (frame-major features) -- what has been MATERIALISED of it
    // Samples whose frames are not in `speech` yet: [the streaming fbank's left-over of the frames already materialised ; everything
    // AddSamples has accepted since].  AddSamples only appends here (no fbank launch, no lock on the model): a frame depends on its own
    // samples only (no dither, no cross-frame state), so the frames of this buffer ARE what per-call fbank would have appended, and
    // SpeechLength counts them.  GetResults hands the buffer to the engine's from-samples path (one batched fbank launch for the whole
    // batch, on the device); k2hip_offline_stream_get_speech and a batch that mixes in already materialised features run the fbank here.
    PinVec wav;
    std::vector<int64_t> tokens;   // Tokens, initialised to [blank, blank] (:34)
    std::vector<int32_t> timestamps;
    int32_t frame_offset = 0;        // FrameOffset (:39), read by the CTC search (OfflineRecognizer.cs:376,402)
    int32_t num_trailing_blank = 0;  // NumTrailingBlank (:40)
};

// OnlineStream.cs:7-199
struct k2hip_online_stream {
    k2hip_model* model;
    int slot = -1;
    std::vector<float> speech;     // OnlineInputEntity.Speech (feature FIFO)
    std::vector<float> remainder;  // streaming fbank state: samples of the last, incomplete frame shift
    // Samples accepted since the features were last materialised.  AddSamples only appends here; the fbank of
    // [remainder ; pending] runs when somebody needs the frames (the next chunk step, IsFinished, a feature read) -- for all
    // streams of a step in ONE batched launch instead of one launch per 50 ms push.  Frames depend only on their own samples
    // (no dither, no cross-frame state), so Speech is what per-call fbank would have appended; SpeechLength counts the frames
    // the pending samples will give.
    std::vector<float> pending;
    long long hyp[2] = {K2HIP_BLANK_ID, K2HIP_BLANK_ID};                 // :44
    std::vector<int64_t> tokens{K2HIP_BLANK_ID, K2HIP_BLANK_ID};         // :45
    std::vector<int32_t> timestamps;
    long long processed_len = 0;   // processed_lens state (16 per chunk)
    long long chunks_done = 0;     // chunks decoded so far: position of the stream's attention rings in its device slot
    // Device mirror of `speech` (Engine::kFifoFrames ring rows in the stream's slot): while mir_ok the ring holds exactly the frames of
    // `speech`, frame 0 at ring row mir_head, so a chunk step gathers its input on the device.  It goes stale only if the FIFO would
    // outgrow the ring; the step then reads `speech` as before, and the mirror is valid again once the FIFO has drained.
    bool mir_ok = true;
    int mir_head = 0;
    // A chunk step that failed part-way (a HIP error) may already have advanced the slot's convolution / embed caches in place (the
    // attention rings are idempotent: their position comes from chunks_done); feeding the same chunk again would then silently
    // corrupt the transcript.  Such a stream refuses every further step until k2hip_online_stream_reset.
    bool poisoned = false;
};

// one stream's encoder caches as the operator-level API sees them (IOnlineProj's List<List<float[]>>): a slot of the device pool
// plus the two scalars of the state that live on the host (processed_lens, ring position)
struct k2hip_online_state {
    k2hip_model* model;
    int slot = -1;
    long long processed_len = 0;
    long long chunks_done = 0;
    bool poisoned = false;  // as k2hip_online_stream::poisoned: a failed EncoderProj leaves the caches undefined; destroy and re-create
};

namespace {

thread_local std::string g_last_error;

template <typename F>
int32_t guard(F&& f) {
    try {
        f();
        return K2HIP_OK;
    } catch (const Error& e) {
        g_last_error = e.what();
        return e.code;
    } catch (const std::bad_alloc&) {
        g_last_error = "out of host memory";
        return K2HIP_ERR_INVALID;
    } catch (const std::exception& e) {
        g_last_error = e.what();
        return K2HIP_ERR_INVALID;
    } catch (...) {
        g_last_error = "unknown error";
        return K2HIP_ERR_INVALID;
    }
}

#define NEED(p)                                                        \
    do {                                                               \
        if (!(p)) failf(K2HIP_ERR_INVALID, "%s: null argument '%s'", __func__, #p); \
    } while (0)

}  // namespace

extern "C" {

const char* k2hip_version(void) { return "k2hip 0.1.0 (gfx950)"; }
const char* k2hip_last_error(void) { return g_last_error.c_str(); }

int32_t k2hip_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// "path.k2w" | "path.k2w@N" (include/k2hip.h)
static void parse_model_spec(const char* spec, std::string* path, int* device) {
    const std::string s(spec);
    *path = s;
    *device = 0;
    const size_t at = s.rfind('@');
    if (at == std::string::npos || at == 0 || at + 1 >= s.size() || s.size() - at - 1 > 4) return;   // (at most four digits)
    int d = 0;
    for (size_t i = at + 1; i < s.size(); i++) {
        if (s[i] < '0' || s[i] > '9') return;
        d = d * 10 + (s[i] - '0');
    }
    *path = s.substr(0, at);
    *device = d;
}
int32_t k2hip_parse_model_spec(const char* spec, char* path, int32_t cap, int32_t* device) {
    return guard([&] {
        NEED(spec); NEED(path); NEED(device);
        std::string p;
        int d = 0;
        parse_model_spec(spec, &p, &d);
        if ((int64_t)p.size() + 1 > (int64_t)cap) failf(K2HIP_ERR_CAPACITY, "model spec: the path needs %zu bytes", p.size() + 1);
        memcpy(path, p.c_str(), p.size() + 1);
        *device = d;
    });
}
int32_t k2hip_model_create_spec(const char* spec, const char* overrides, k2hip_model_t** out) {
    if (!spec) return k2hip_model_create(nullptr, overrides, 0, out);
    std::string p;
    int d = 0;
    parse_model_spec(spec, &p, &d);
    return k2hip_model_create(p.c_str(), overrides, d, out);
}
int32_t k2hip_model_create(const char* weights_path, const char* overrides, int32_t device, k2hip_model_t** out) {
    return guard([&] {
        NEED(weights_path);
        NEED(out);
        *out = nullptr;
        *out = new k2hip_model(weights_path, overrides, device);
    });
}
int32_t k2hip_model_destroy(k2hip_model_t* model) {
    return guard([&] { delete model; });
}
int32_t k2hip_model_get_info(const k2hip_model_t* model, k2hip_model_info* info) {
    return guard([&] {
        NEED(model);
        NEED(info);
        const Config& c = model->engine.model().cfg();
        info->vocab_size = c.V;
        info->context_size = c.ctx;
        info->joiner_dim = c.J;
        info->reserved = c.enc_dim();  // width of the encoder entry points' output (vocab_size for zipformer2ctc)
        info->feature_dim = c.feat;
        info->sample_rate = c.fbank.sample_rate;
        info->num_stacks = c.ns;
        info->device = model->engine.model().device();
    });
}
int32_t k2hip_model_meta(const k2hip_model_t* model, const char* key, char* buf, int32_t cap) {
    return guard([&] {
        NEED(model);
        NEED(key);
        NEED(buf);
        auto& mm = model->engine.model().meta();
        auto it = mm.find(key);
        if (it == mm.end()) failf(K2HIP_ERR_INVALID, "metadata key '%s' not present", key);
        if ((int)it->second.size() + 1 > cap) failf(K2HIP_ERR_CAPACITY, "metadata value needs %zu bytes", it->second.size() + 1);
        memcpy(buf, it->second.c_str(), it->second.size() + 1);
    });
}
int32_t k2hip_get_gemm_profile(k2hip_model_t* model, float* rows, int32_t cap_rows, int32_t* n_rows) {
    return guard([&] {
        NEED(model); NEED(n_rows);
        EngineLock lk(model->engine);
        const auto& lg = model->engine.gemm_log();
        *n_rows = (int32_t)lg.size();
        if (!rows) return;
        for (int i = 0; i < std::min<int>(cap_rows, (int)lg.size()); i++) {
            const GemmLaunchRec& r = lg[i];
            float* o = rows + (size_t)i * 8;
            o[0] = (float)r.M; o[1] = (float)r.N; o[2] = (float)r.K; o[3] = (float)r.batch;
            o[4] = (float)r.act; o[5] = (float)r.res; o[6] = (float)r.kind; o[7] = r.us;
        }
    });
}
int32_t k2hip_set_instrument(k2hip_model_t* model, int32_t on) {
    return guard([&] {
        NEED(model);
        EngineLock lk(model->engine);
        model->engine.set_instrument(on != 0);
    });
}
int32_t k2hip_get_timing(const k2hip_model_t* model, k2hip_timing* timing) {
    return guard([&] {
        NEED(model);
        NEED(timing);
        *timing = model->engine.timing();
    });
}

int64_t k2hip_fbank_num_frames(const k2hip_model_t* model, int64_t n_samples) {
    if (!model) return -1;
    return model->engine.fbank_num_frames(n_samples);
}
int32_t k2hip_fbank(k2hip_model_t* model, const float* samples, int64_t n_samples, float* feats, int64_t cap_frames,
                    int64_t* n_frames) {
    return guard([&] {
        NEED(model);
        NEED(n_frames);
        if (n_samples > 0) NEED(samples);
        if (cap_frames > 0) NEED(feats);
        EngineLock lk(model->engine);
        model->engine.fbank_host(samples, n_samples, feats, cap_frames, n_frames);
    });
}
int32_t k2hip_pad_sequence(k2hip_model_t* model, const float* const* speech, const int64_t* n_floats, int32_t B,
                           int32_t tail_frames, float* out, int64_t cap_floats, int64_t* padded_len) {
    return guard([&] {
        NEED(model);
        NEED(speech);
        NEED(n_floats);
        NEED(out);
        NEED(padded_len);
        EngineLock lk(model->engine);
        model->engine.pad_host(speech, n_floats, B, tail_frames, out, cap_floats, padded_len);
    });
}
int32_t k2hip_encoder_out_frames(const k2hip_model_t* model, int32_t T) {
    if (!model) return -1;
    return model->engine.encoder_out_frames(T);
}
int32_t k2hip_offline_encoder(k2hip_model_t* model, const float* x, const int64_t* x_lens, int32_t B, int32_t T,
                              float* enc_out, int64_t cap_floats, int64_t* enc_out_lens, int32_t* Tprime) {
    return guard([&] {
        NEED(model);
        NEED(x);
        NEED(enc_out);
        NEED(Tprime);
        (void)x_lens;  // always T in the reference (OfflineProjOfTransducer.cs:66-70); no masking on this path
        EngineLock lk(model->engine);
        int tp = 0;
        model->engine.encoder_host(x, B, T, enc_out, cap_floats, &tp);
        *Tprime = tp;
        if (enc_out_lens)
            for (int b = 0; b < B; b++) enc_out_lens[b] = tp;
    });
}
int32_t k2hip_offline_encoder_tap(k2hip_model_t* model, const float* x, int32_t B, int32_t T, int32_t tap, float* out,
                                  int64_t cap_floats, int64_t* n_floats) {
    return guard([&] {
        NEED(model);
        NEED(x);
        NEED(out);
        NEED(n_floats);
        EngineLock lk(model->engine);
        model->engine.encoder_tap_host(x, B, T, tap, out, cap_floats, n_floats);
    });
}
int32_t k2hip_decoder(k2hip_model_t* model, const int64_t* y, int32_t N, float* dec_out) {
    return guard([&] {
        NEED(model);
        NEED(dec_out);
        EngineLock lk(model->engine);
        model->engine.decoder_host(y, N, dec_out);
    });
}
int32_t k2hip_joiner(k2hip_model_t* model, const float* enc, const float* dec, int32_t N, float* logits) {
    return guard([&] {
        NEED(model);
        NEED(enc);
        NEED(dec);
        NEED(logits);
        EngineLock lk(model->engine);
        model->engine.joiner_host(enc, dec, N, logits);
    });
}
int32_t k2hip_greedy_batch(k2hip_model_t* model, const float* enc_out, int32_t B, int32_t Tprime, int64_t* tokens,
                           int32_t* timestamps, int32_t* n_tokens, int32_t max_tokens) {
    return guard([&] {
        NEED(model); NEED(enc_out); NEED(tokens); NEED(timestamps); NEED(n_tokens);
        EngineLock lk(model->engine);
        model->engine.greedy_host(enc_out, B, Tprime, false, tokens, timestamps, n_tokens, max_tokens);
    });
}
int32_t k2hip_greedy_single(k2hip_model_t* model, const float* enc_out, int32_t Tprime, int64_t* tokens,
                            int32_t* timestamps, int32_t* n_tokens, int32_t max_tokens) {
    return guard([&] {
        NEED(model); NEED(enc_out); NEED(tokens); NEED(timestamps); NEED(n_tokens);
        EngineLock lk(model->engine);
        model->engine.greedy_host(enc_out, 1, Tprime, true, tokens, timestamps, n_tokens, max_tokens);
    });
}
// ---- token ids -> text (text.cpp) ----------------------------------------------------------------
struct k2hip_tokens {
    k2hip::TokenTable* tab;
};
int32_t k2hip_tokens_load(const char* tokens_path, k2hip_tokens_t** out) {
    return guard([&] {
        NEED(tokens_path); NEED(out);
        auto* t = new k2hip_tokens{token_table_load(tokens_path)};
        *out = t;
    });
}
int32_t k2hip_tokens_destroy(k2hip_tokens_t* t) {
    return guard([&] {
        if (t) {
            token_table_free(t->tab);
            delete t;
        }
    });
}
int32_t k2hip_tokens_size(const k2hip_tokens_t* t) { return t ? token_table_size(t->tab) : -1; }
int32_t k2hip_bbpe_char(int32_t byte) { return bbpe_char_of_byte(byte); }
int32_t k2hip_bbpe_byte(int32_t code_point) { return code_point < 0 ? -1 : bbpe_byte_of_char((uint32_t)code_point); }
int32_t k2hip_decode_text(const k2hip_tokens_t* t, const int64_t* ids, int32_t n, int32_t online, char* out, int32_t cap,
                          int32_t* len) {
    return guard([&] {
        NEED(t); NEED(len);
        K2_REQUIRE(n == 0 || ids != nullptr, "decode_text: ids is NULL");
        std::string s = decode_tokens(*t->tab, ids, n, online != 0);
        *len = (int32_t)s.size();
        if (!out) return;
        if ((int32_t)s.size() + 1 > cap) failf(K2HIP_ERR_CAPACITY, "decode_text: text needs %zu bytes", s.size() + 1);
        memcpy(out, s.c_str(), s.size() + 1);
    });
}
int32_t k2hip_ctc_greedy(k2hip_model_t* model, const float* log_probs, int32_t B, int32_t Tprime, const int32_t* frame_offsets,
                         int64_t* tokens, int32_t* timestamps, int32_t* n_tokens, int32_t max_tokens, int32_t* num_trailing_blank) {
    return guard([&] {
        NEED(model); NEED(log_probs); NEED(tokens); NEED(timestamps); NEED(n_tokens);
        Engine& e = model->engine;
        K2_REQUIRE(e.model().cfg().ctc, "ctc_greedy: model_type '%s' has no CTC head", e.model().cfg().model_type.c_str());
        EngineLock lk(e);
        e.greedy_host(log_probs, B, Tprime, false, tokens, timestamps, n_tokens, max_tokens);
        for (int b = 0; b < B; b++) {
            if (frame_offsets)
                for (int k = 0; k < n_tokens[b]; k++) timestamps[(size_t)b * max_tokens + k] += frame_offsets[b];
            if (num_trailing_blank) num_trailing_blank[b] = e.last_any()[b] ? e.last_trail()[b] : num_trailing_blank[b] + e.last_trail()[b];
        }
    });
}
int32_t k2hip_offline_stream_get_ctc_state(const k2hip_offline_stream_t* s, int32_t* frame_offset, int32_t* num_trailing_blank) {
    return guard([&] {
        NEED(s);
        if (frame_offset) *frame_offset = s->frame_offset;
        if (num_trailing_blank) *num_trailing_blank = s->num_trailing_blank;
    });
}
int32_t k2hip_set_decoding_method(k2hip_model_t* model, const char* method, int32_t beam) {
    return guard([&] {
        NEED(model); NEED(method);
        EngineLock lk(model->engine);
        if (!strcmp(method, "greedy_search")) {
            model->engine.set_beam(0);
        } else if (!strcmp(method, "modified_beam_search")) {
            K2_REQUIRE(beam >= 1 && beam <= kMaxBeam, "modified_beam_search: beam %d out of range [1,%d]", beam, kMaxBeam);
            model->engine.set_beam(beam);
        } else {
            failf(K2HIP_ERR_UNSUPPORTED, "decoding method '%s' (have: greedy_search, modified_beam_search)", method);
        }
    });
}
int32_t k2hip_beam_search(k2hip_model_t* model, const float* enc_out, int32_t B, int32_t Tprime, int32_t beam, int64_t* tokens,
                          int32_t* timestamps, int32_t* n_tokens, int32_t max_tokens, float* scores) {
    return guard([&] {
        NEED(model); NEED(enc_out); NEED(tokens); NEED(timestamps); NEED(n_tokens);
        K2_REQUIRE(beam >= 1 && beam <= kMaxBeam, "beam search: beam %d out of range [1,%d]", beam, kMaxBeam);
        Engine& e = model->engine;
        EngineLock lk(e);
        struct Restore {
            Engine& e; int old;
            ~Restore() { e.set_beam(old); }
        } restore{e, e.beam()};
        e.set_beam(beam);
        e.greedy_host(enc_out, B, Tprime, false, tokens, timestamps, n_tokens, max_tokens);
        if (scores) memcpy(scores, e.last_scores().data(), sizeof(float) * B);
    });
}
int32_t k2hip_last_scores(k2hip_model_t* model, float* scores, int32_t B) {
    return guard([&] {
        NEED(model); NEED(scores);
        EngineLock lk(model->engine);
        K2_REQUIRE((size_t)B == model->engine.last_scores().size(), "last_scores: the last beam-search call had %zu streams, not %d",
                   model->engine.last_scores().size(), B);
        memcpy(scores, model->engine.last_scores().data(), sizeof(float) * B);
    });
}
int32_t k2hip_offline_greedy(k2hip_model_t* model, const float* const* feats, const int64_t* n_floats, int32_t B,
                             int64_t* tokens, int32_t* timestamps, int32_t* n_tokens, int32_t max_tokens) {
    return guard([&] {
        NEED(model); NEED(feats); NEED(n_floats); NEED(tokens); NEED(timestamps); NEED(n_tokens);
        EngineLock lk(model->engine);
        model->engine.offline_greedy_feats(feats, n_floats, B, false, tokens, timestamps, n_tokens, max_tokens);
    });
}
int32_t k2hip_offline_greedy_single(k2hip_model_t* model, const float* feats, int64_t n_floats, int64_t* tokens,
                                    int32_t* timestamps, int32_t* n_tokens, int32_t max_tokens) {
    return guard([&] {
        NEED(model); NEED(feats); NEED(tokens); NEED(timestamps); NEED(n_tokens);
        EngineLock lk(model->engine);
        const float* p[1] = {feats};
        int64_t n[1] = {n_floats};
        model->engine.offline_greedy_feats(p, n, 1, true, tokens, timestamps, n_tokens, max_tokens);
    });
}
int32_t k2hip_offline_greedy_from_samples(k2hip_model_t* model, const float* const* samples, const int64_t* n_samples,
                                          int32_t B, int64_t* tokens, int32_t* timestamps, int32_t* n_tokens,
                                          int32_t max_tokens) {
    return guard([&] {
        NEED(model); NEED(samples); NEED(n_samples); NEED(tokens); NEED(timestamps); NEED(n_tokens);
        EngineLock lk(model->engine);
        model->engine.offline_greedy_samples(samples, n_samples, B, tokens, timestamps, n_tokens, max_tokens);
    });
}
int32_t k2hip_offline_greedy_from_samples_dev(k2hip_model_t* model, const float* samples_dev, int64_t n_samples_each,
                                              int32_t B, int64_t* tokens, int32_t* timestamps, int32_t* n_tokens,
                                              int32_t max_tokens) {
    return guard([&] {
        NEED(model); NEED(samples_dev); NEED(tokens); NEED(timestamps); NEED(n_tokens);
        EngineLock lk(model->engine);
        model->engine.offline_greedy_samples_dev(samples_dev, n_samples_each, B, tokens, timestamps, n_tokens, max_tokens);
    });
}

int32_t k2hip_offline_submit_samples_dev(k2hip_model_t* model, const float* samples_dev, int64_t n_samples_each, int32_t B,
                                         int32_t max_tokens, int32_t* ticket) {
    return guard([&] {
        NEED(model); NEED(samples_dev); NEED(ticket);
        EngineLock lk(model->engine);
        *ticket = model->engine.submit_samples_dev(samples_dev, n_samples_each, B, max_tokens);
    });
}
int32_t k2hip_offline_submit_samples(k2hip_model_t* model, const float* samples_host, int64_t n_samples_each, int32_t B,
                                     int32_t max_tokens, int32_t* ticket) {
    return guard([&] {
        NEED(model); NEED(samples_host); NEED(ticket);
        EngineLock lk(model->engine);
        *ticket = model->engine.submit_samples_host(samples_host, n_samples_each, B, max_tokens);
    });
}
int32_t k2hip_host_alloc(k2hip_model_t* model, int64_t bytes, void** host_ptr) {
    return guard([&] {
        NEED(model); NEED(host_ptr);
        K2_REQUIRE(bytes > 0, "host_alloc: %lld bytes", (long long)bytes);
        *host_ptr = model->engine.host_alloc(bytes);
    });
}
int32_t k2hip_host_free(k2hip_model_t* model, void* host_ptr) {
    return guard([&] {
        NEED(model);
        if (host_ptr) model->engine.host_free(host_ptr);
    });
}
int32_t k2hip_offline_wait(k2hip_model_t* model, int32_t ticket, int64_t* tokens, int32_t* timestamps, int32_t* n_tokens) {
    return guard([&] {
        NEED(model); NEED(tokens); NEED(timestamps); NEED(n_tokens);
        EngineLock lk(model->engine);
        model->engine.wait_ticket(ticket, tokens, timestamps, n_tokens);
    });
}

int32_t k2hip_device_alloc(k2hip_model_t* model, int64_t bytes, void** dev_ptr) {
    return guard([&] {
        NEED(model); NEED(dev_ptr);
        *dev_ptr = model->engine.dev_alloc(bytes);
    });
}
int32_t k2hip_device_free(k2hip_model_t* model, void* dev_ptr) {
    return guard([&] {
        NEED(model);
        if (dev_ptr) model->engine.dev_free(dev_ptr);
    });
}
int32_t k2hip_device_upload(k2hip_model_t* model, void* dev_dst, const void* host_src, int64_t bytes) {
    return guard([&] {
        NEED(model); NEED(dev_dst); NEED(host_src);
        model->engine.dev_upload(dev_dst, host_src, bytes);
    });
}
int32_t k2hip_synchronize(k2hip_model_t* model) {
    return guard([&] {
        NEED(model);
        model->engine.synchronize();
    });
}

// ---- OnlineStream / OnlineRecognizer ----------------------------------------------------------
int32_t k2hip_online_stream_create(k2hip_model_t* model, k2hip_online_stream_t** out) {
    return guard([&] {
        NEED(model); NEED(out);
        *out = nullptr;
        EngineLock lk(model->engine);
        auto* s = new k2hip_online_stream();
        s->model = model;
        try {
            s->slot = model->engine.online_alloc_slot();
        } catch (...) {
            delete s;
            throw;
        }
        // OnlineProjOfConformer.GetEncoderInitStates (:76-77) starts processed_lens at 2, not 0
        if (model->engine.model().cfg().conformer) s->processed_len = 2;
        *out = s;
    });
}
// Back to the state of a freshly created stream (new utterance on the same object): caches zeroed in place, FIFOs, tokens,
// timestamps and Hyp cleared.  The reference has no such method -- a host would drop the OnlineStream and create a new one
// (OnlineRecognizer.cs:60-64); this keeps the slot and skips the allocation.
int32_t k2hip_online_stream_reset(k2hip_online_stream_t* s) {
    return guard([&] {
        NEED(s);
        k2hip_model* model = s->model;
        {
            EngineLock lk(model->engine);
            model->engine.online_free_slot(s->slot);
            s->slot = -1;
            s->slot = model->engine.online_alloc_slot();  // (the slot just freed: re-zeroed like GetEncoderInitStates)
        }
        *s = k2hip_online_stream{model, s->slot};
        if (model->engine.model().cfg().conformer) s->processed_len = 2;
    });
}
int32_t k2hip_online_stream_destroy(k2hip_online_stream_t* s) {
    return guard([&] {
        if (!s) return;
        {
            EngineLock lk(s->model->engine);
            s->model->engine.online_free_slot(s->slot);
        }
        delete s;
    });
}
int32_t k2hip_online_chunk_info(const k2hip_model_t* model, int32_t* chunk_length, int32_t* shift_length, int32_t* frames_per_chunk) {
    return guard([&] {
        NEED(model);
        const Config& c = model->engine.model().cfg();
        K2_REQUIRE(c.streaming || c.lstm, "this model is not a streaming export");
        if (chunk_length) *chunk_length = c.chunk_T;
        if (shift_length) *shift_length = c.shift;
        if (frames_per_chunk) *frames_per_chunk = model->engine.online_frames_per_chunk();
    });
}
static void online_add_samples(k2hip_online_stream* s, const float* samples, int64_t n) {
    if (n > 0) s->pending.insert(s->pending.end(), samples, samples + n);  // OnlineStream.AddSamples (:57-79), fbank deferred
}
// frames the stream's pending samples will add to Speech
static int64_t online_pending_frames(const k2hip_online_stream* s) {
    return s->pending.empty() ? 0 : s->model->engine.fbank_num_frames((int64_t)(s->remainder.size() + s->pending.size()));
}
static int64_t online_logical_floats(const k2hip_online_stream* s) {
    return (int64_t)s->speech.size() + online_pending_frames(s) * s->model->engine.model().cfg().feat;
}
// run the deferred fbank of `n` streams of one model: streams at the same position share one batched launch
// may_defer: the caller runs a chunk step on the same engine next and collects the frames after it (Engine::fbank_gather_finish);
// taken only when ONE batched launch covers every stream with queued samples.  Returns whether a gather was left outstanding.
static bool online_materialize(k2hip_online_stream* const* streams, int n, bool may_defer = false) {
    if (n <= 0) return false;
    bool deferred = false;
    Engine& e = streams[0]->model->engine;
    const Config& c = e.model().cfg();
    std::map<int64_t, std::vector<k2hip_online_stream*>> groups;  // by length of [remainder ; pending]
    for (int i = 0; i < n; i++) {
        k2hip_online_stream* s = streams[i];
        if (s->pending.empty()) continue;
        groups[(int64_t)(s->remainder.size() + s->pending.size())].push_back(s);
    }
    for (auto& kv : groups) {
        const int64_t len = kv.first, nf = e.fbank_num_frames(len);
        std::vector<k2hip_online_stream*>& g = kv.second;
        const int G = (int)g.size();
        if (nf == 0) {  // OnlineStream.cs:67: nothing is appended until the fbank produces a frame
            for (k2hip_online_stream* s : g) {
                s->remainder.insert(s->remainder.end(), s->pending.begin(), s->pending.end());
                s->pending.clear();
            }
            continue;
        }
        // [remainder ; pending] of every stream goes straight into the engine's pinned staging buffer, the frames come straight
        // out of it into the streams' Speech
        std::vector<const float*> hp(G), tp(G);
        std::vector<int64_t> hn(G), tn(G);
        std::vector<float*> dst(G);
        std::vector<int> fslot(G), fpos(G);
        for (int i = 0; i < G; i++) {
            k2hip_online_stream* s = g[i];
            hp[i] = s->remainder.data(); hn[i] = (int64_t)s->remainder.size();
            tp[i] = s->pending.data(); tn[i] = (int64_t)s->pending.size();
            const size_t old = s->speech.size();
            const int64_t have = (int64_t)(old / (size_t)c.feat);
            if (s->mir_ok && have + nf > Engine::kFifoFrames) s->mir_ok = false;  // the FIFO outgrows its device mirror
            fslot[i] = s->slot;
            fpos[i] = s->mir_ok ? (int)((s->mir_head + have) % Engine::kFifoFrames) : -1;
            s->speech.resize(old + (size_t)nf * c.feat);
            dst[i] = s->speech.data() + old;
        }
        try {
            EngineLock lk(e);
            const bool defer = may_defer && groups.size() == 1;
            e.fbank_host_gather(hp.data(), hn.data(), tp.data(), tn.data(), len, G, dst.data(), nf, fslot.data(), fpos.data(), defer);
            deferred = defer;
        } catch (...) {
            for (k2hip_online_stream* s : g) s->speech.resize(s->speech.size() - (size_t)nf * c.feat);  // nothing was appended
            throw;
        }
        const int64_t used = nf * c.fbank.frame_shift;  // samples consumed by whole frame shifts
        for (int i = 0; i < G; i++) {
            k2hip_online_stream* s = g[i];
            std::vector<float> rem((size_t)(len - used));
            // the new remainder is the tail of [remainder ; pending]
            for (int64_t k = used; k < len; k++) rem[(size_t)(k - used)] = k < hn[i] ? s->remainder[(size_t)k] : s->pending[(size_t)(k - hn[i])];
            s->remainder.swap(rem);
            s->pending.clear();
        }
    }
    return deferred;
}
int32_t k2hip_online_stream_accept_samples(k2hip_online_stream_t* s, const float* samples, int64_t n) {
    return guard([&] {
        NEED(s);
        if (n > 0) NEED(samples);
        online_add_samples(s, samples, n);
    });
}
int32_t k2hip_online_accept_samples_batch(k2hip_model_t* model, k2hip_online_stream_t* const* streams, int32_t B,
                                          const float* const* samples, const int64_t* n) {
    return guard([&] {
        NEED(model); NEED(streams); NEED(samples); NEED(n);
        K2_REQUIRE(B > 0, "accept_samples_batch: empty list");
        for (int i = 0; i < B; i++) {
            NEED(streams[i]);
            if (n[i] > 0) NEED(samples[i]);
            online_add_samples(streams[i], samples[i], n[i]);
        }
    });
}
// B AddSamples calls over the rows of one [B, n] matrix (row stride in floats): what a host that receives audio in lock step
// holds anyway; saves building B pointers per 50 ms push
int32_t k2hip_online_accept_samples_matrix(k2hip_model_t* model, k2hip_online_stream_t* const* streams, int32_t B, const float* samples,
                                           int64_t row_stride, int64_t n) {
    return guard([&] {
        NEED(model); NEED(streams);
        K2_REQUIRE(B > 0 && n >= 0 && row_stride >= n, "accept_samples_matrix: bad shape");
        if (n > 0) NEED(samples);
        for (int i = 0; i < B; i++) {
            NEED(streams[i]);
            online_add_samples(streams[i], samples + (size_t)i * row_stride, n);
        }
    });
}
int32_t k2hip_online_stream_accept_features(k2hip_online_stream_t* s, const float* feats, int64_t n_frames) {
    return guard([&] {
        NEED(s);
        if (n_frames > 0) NEED(feats);
        const int feat = s->model->engine.model().cfg().feat;
        online_materialize(&s, 1);  // frames of samples accepted earlier come first
        if (n_frames > 0) {
            const int64_t have = (int64_t)(s->speech.size() / (size_t)feat);
            if (s->mir_ok && have + n_frames > Engine::kFifoFrames) s->mir_ok = false;
            if (s->mir_ok) {
                Engine& e = s->model->engine;
                EngineLock lk(e);
                e.online_fifo_write(s->slot, (int)((s->mir_head + have) % Engine::kFifoFrames), feats, n_frames);
            }
        }
        s->speech.insert(s->speech.end(), feats, feats + n_frames * feat);
    });
}
int64_t k2hip_online_stream_speech_length(const k2hip_online_stream_t* s) { return s ? online_logical_floats(s) : -1; }
// OnlineStream.IsFinished (:124-161)
int32_t k2hip_online_stream_is_finished(k2hip_online_stream_t* s, int32_t is_endpoint, int32_t* finished) {
    return guard([&] {
        NEED(s); NEED(finished);
        const Config& c = s->model->engine.model().cfg();
        *finished = 0;
        if (!is_endpoint) return;
        online_materialize(&s, 1);  // the test below looks at the feature VALUES
        const size_t oLen = s->speech.size();
        if (oLen == 0) { *finished = 1; return; }
        double sum = 0.0;   // LINQ Average() over float[] accumulates in double and returns float
        for (float v : s->speech) sum += v;
        const float avg = (float)(sum / (double)oLen);
        size_t num = 0;
        for (float v : s->speech) num += (v != avg);
        if (num == 0) { *finished = 1; return; }
        if ((long long)oLen <= (long long)c.chunk_T * c.feat) {
            std::vector<float> z(400, 0.f);  // AddSamples(new float[400]) (:146)
            online_add_samples(s, z.data(), 400);
        }
    });
}
// OnlineRecognizer.ForwardBatchGreedySearch (:85-219)
int32_t k2hip_online_step(k2hip_model_t* model, k2hip_online_stream_t* const* streams, int32_t B, int32_t* decoded, int32_t* n_new_tokens) {
    return guard([&] {
        NEED(model); NEED(streams); NEED(decoded); NEED(n_new_tokens);
        Engine& e = model->engine;
        const Config& c = e.model().cfg();
        K2_REQUIRE(c.streaming || c.lstm, "this model is not a streaming export");
        const size_t chunk_floats = (size_t)c.chunk_T * c.feat, shift_floats = (size_t)c.shift * c.feat;
        std::vector<int> idx;
        for (int i = 0; i < B; i++) {
            NEED(streams[i]);
            K2_REQUIRE(streams[i]->model == model, "stream %d belongs to another model", i);
            K2_REQUIRE(!streams[i]->poisoned, "stream %d took part in a chunk step that failed: its caches are undefined, reset it first", i);
            decoded[i] = 0;
            n_new_tokens[i] = 0;
            if ((size_t)online_logical_floats(streams[i]) >= chunk_floats) idx.push_back(i);   // GetDecodeChunk (:82-100)
        }
        {   // one stream twice in the list would consume two chunks against one state slot in the same launch
            std::vector<const k2hip_online_stream*> seen(streams, streams + B);
            std::sort(seen.begin(), seen.end());
            K2_REQUIRE(std::adjacent_find(seen.begin(), seen.end()) == seen.end(), "GetResults: the same stream appears twice in the list");
        }
        if (idx.empty()) return;   // :113-116
        // the new frames' host copy is collected after the step (one wait for the device per tick); whatever way this call ends -- an
        // allocation failure below included -- the outstanding download is collected before the streams' Speech can move
        // If the collection itself fails (the device work behind the download did), the streams whose Speech was extended with
        // placeholder frames for it are poisoned: their FIFOs hold zeros where audio should be, and a later step must not decode them.
        struct GatherFinisher {
            Engine& e;
            std::vector<k2hip_online_stream*> owners;   // the streams whose frames the outstanding download carries
            bool armed = false;
            void finish() {
                if (!armed) return;
                armed = false;
                try {
                    EngineLock lk(e);
                    e.fbank_gather_finish();
                } catch (...) {
                    for (k2hip_online_stream* s : owners) s->poisoned = true;
                    throw;
                }
            }
            ~GatherFinisher() {
                try { finish(); } catch (...) {}   // (already unwinding, or the caller is done: the streams are poisoned, the error of record is the first one)
            }
        } fb{e};
        {   // the deferred fbank of the streams that decode now: one batched launch when they are at the same position
            std::vector<k2hip_online_stream*> ready(idx.size());
            for (size_t r = 0; r < idx.size(); r++) ready[r] = streams[idx[r]];
            fb.owners = ready;
            fb.armed = online_materialize(ready.data(), (int)ready.size(), true);
        }
        const int R = (int)idx.size(), Tp = e.online_frames_per_chunk();
        std::vector<const float*> chunks(R);   // GetDecodeChunk: the first ChunkLength frames of each FIFO
        std::vector<int> slots(R);
        std::vector<long long> hyps(2 * (size_t)R), plens(R);
        std::vector<int> nch(R), heads(R);
        bool all_mirrored = true;
        for (int r = 0; r < R; r++) {
            k2hip_online_stream* s = streams[idx[r]];
            heads[r] = s->mir_head;
            all_mirrored = all_mirrored && s->mir_ok;
            chunks[r] = s->speech.data();
            slots[r] = s->slot;
            hyps[2 * r] = s->hyp[0];
            hyps[2 * r + 1] = s->hyp[1];
            plens[r] = s->processed_len;
            nch[r] = (int)(s->chunks_done % (1LL << 30));  // the kernels only need it modulo the ring lengths; (1 << 30) % KL drift is
                                                           // irrelevant before 2^30 chunks (~10 years of audio)
        }
        std::vector<int64_t> tok((size_t)R * Tp);
        std::vector<int32_t> ts((size_t)R * Tp), n(R);
        if (!all_mirrored) fb.finish();   // the step reads some stream's chunk from host memory: the frames must be there
        try {
            EngineLock lk(e);
            e.online_step(slots.data(), chunks.data(), hyps.data(), plens.data(), nch.data(), R, tok.data(), ts.data(), n.data(),
                          all_mirrored ? heads.data() : nullptr);
        } catch (...) {
            // Host-side nothing has moved (RemoveChunk happens below), but the device caches of these streams may have: the step
            // updates the conv / embed caches in place.  A search exchange timeout is retried inside the engine and does not come
            // here; what does is a HIP failure.  The streams are unusable until reset.
            for (int r = 0; r < R; r++) streams[idx[r]]->poisoned = true;
            throw;   // (fb's destructor collects the frames; the streams are poisoned either way)
        }
        fb.finish();   // the step's token download has synchronised the stream: copies only
        // RemoveChunk (:102-117) only after success
        std::vector<k2hip_online_stream*> remirror;
        for (int r = 0; r < R; r++) {
            k2hip_online_stream* s = streams[idx[r]];
            s->speech.erase(s->speech.begin(), s->speech.begin() + shift_floats);
            s->mir_head = (s->mir_head + c.shift) % Engine::kFifoFrames;
            if (!s->mir_ok && s->speech.size() / (size_t)c.feat <= (size_t)Engine::kFifoFrames) {
                // the FIFO fits its device ring again (a host that pushed a whole file before decoding, as the reference's example
                // does, never drains it to empty: a step needs ChunkLength frames and removes only ShiftLength): re-upload what is
                // left at ring row 0 and the stream is back on the one-upload-per-tick path
                remirror.push_back(s);
            }
            s->chunks_done++;
            for (int k = 0; k < n[r]; k++) {
                s->tokens.push_back(tok[(size_t)r * Tp + k]);          // :183
                s->timestamps.push_back(ts[(size_t)r * Tp + k]);       // :184 (chunk-relative frame index)
            }
            if (!c.ctc) {  // the CTC delegate leaves Hyp alone (OnlineRecognizer.cs:302-310)
                s->hyp[0] = s->tokens[s->tokens.size() - 2];           // :208
                s->hyp[1] = s->tokens[s->tokens.size() - 1];
            }
            if (c.conformer) s->processed_len = R;   // OnlineProjOfConformer.unstack_states (:229) stores the BATCH SIZE (sic), not the model's output
            else if (c.zip1) s->processed_len += (c.chunk_T - 7) / 2;  // no processed_lens state in v1; kept as a frame counter
            else s->processed_len += (c.chunk_T - 7) / 2 - 3;      // new_processed_lens = processed_lens + x_lens
            decoded[idx[r]] = 1;
            n_new_tokens[idx[r]] = n[r];
        }
        if (!remirror.empty()) {
            EngineLock lk(e);
            for (k2hip_online_stream* s : remirror) {
                const int64_t have = (int64_t)(s->speech.size() / (size_t)c.feat);
                if (have > 0) e.online_fifo_write(s->slot, 0, s->speech.data(), have);
                s->mir_head = 0;
                s->mir_ok = true;
            }
        }
    });
}
// ---- operator level of the streaming path (IOnlineProj.cs:65-71) --------------------------------------------------------------
// GetEncoderInitStates for ONE stream (OnlineProjOfZipformer2.cs:144-238): a zeroed slot
int32_t k2hip_online_state_create(k2hip_model_t* model, k2hip_online_state_t** out) {
    return guard([&] {
        NEED(model); NEED(out);
        const Config& c = model->engine.model().cfg();
        K2_REQUIRE(c.streaming && !c.lstm && !c.conformer && !c.zip1 && !c.ctc, "online states: streaming Zipformer2 transducer models only");
        auto* st = new k2hip_online_state();
        st->model = model;
        try {
            EngineLock lk(model->engine);
            st->slot = model->engine.online_alloc_slot();
        } catch (...) {
            delete st;
            throw;
        }
        *out = st;
    });
}
int32_t k2hip_online_state_destroy(k2hip_online_state_t* st) {
    return guard([&] {
        if (!st) return;
        {
            EngineLock lk(st->model->engine);
            st->model->engine.online_free_slot(st->slot);
        }
        delete st;
    });
}
int64_t k2hip_online_state_processed_len(const k2hip_online_state_t* st) { return st ? (int64_t)st->processed_len : -1; }
// EncoderProj (OnlineProjOfZipformer2.cs:491-618) over B states: feats [B, ChunkLength, FeatureDim] (raw fbank: the log floor of the
// online PadSequence is applied inside), encoder_out [B, T', joiner_dim]; the states advance in place, so stack_states /
// unstack_states (:240-489) are the identity on the handles
int32_t k2hip_online_encoder(k2hip_model_t* model, k2hip_online_state_t* const* states, int32_t B, const float* feats, float* encoder_out,
                             int64_t cap_floats) {
    return guard([&] {
        NEED(model); NEED(states); NEED(feats); NEED(encoder_out);
        K2_REQUIRE(B > 0, "EncoderProj: empty batch");
        Engine& e = model->engine;
        const Config& c = e.model().cfg();
        const int Tp = e.online_frames_per_chunk();
        if (cap_floats < (int64_t)B * Tp * c.enc_dim()) failf(K2HIP_ERR_CAPACITY, "encoder_out needs %lld floats", (long long)B * Tp * c.enc_dim());
        std::vector<int> slots(B), nch(B);
        std::vector<long long> plens(B);
        {
            std::vector<const k2hip_online_state*> seen(states, states + B);
            for (int b = 0; b < B; b++) {
                NEED(states[b]);
                K2_REQUIRE(states[b]->model == model, "state %d belongs to another model", b);
                K2_REQUIRE(!states[b]->poisoned, "state %d took part in an EncoderProj call that failed: its caches are undefined, re-create it", b);
            }
            std::sort(seen.begin(), seen.end());
            K2_REQUIRE(std::adjacent_find(seen.begin(), seen.end()) == seen.end(), "EncoderProj: the same state appears twice in the batch");
        }
        for (int b = 0; b < B; b++) {
            slots[b] = states[b]->slot;
            plens[b] = states[b]->processed_len;
            nch[b] = (int)(states[b]->chunks_done % (1LL << 30));
        }
        try {
            EngineLock lk(e);
            e.online_encoder(slots.data(), feats, plens.data(), nch.data(), B, encoder_out);
        } catch (...) {
            for (int b = 0; b < B; b++) states[b]->poisoned = true;  // the conv / embed caches may have advanced in place
            throw;
        }
        for (int b = 0; b < B; b++) {  // only after success, as in k2hip_online_step
            states[b]->chunks_done++;
            states[b]->processed_len += (c.chunk_T - 7) / 2 - 3;
        }
    });
}

int64_t k2hip_online_stream_processed_len(const k2hip_online_stream_t* s) { return s ? (int64_t)s->processed_len : -1; }
int32_t k2hip_online_stream_num_tokens(const k2hip_online_stream_t* s) { return s ? (int32_t)s->tokens.size() : -1; }
int32_t k2hip_online_stream_num_timestamps(const k2hip_online_stream_t* s) { return s ? (int32_t)s->timestamps.size() : -1; }
int32_t k2hip_online_stream_get_tokens(const k2hip_online_stream_t* s, int64_t* tokens, int32_t cap) {
    return guard([&] {
        NEED(s);
        if ((int)s->tokens.size() > cap) failf(K2HIP_ERR_CAPACITY, "stream holds %zu tokens", s->tokens.size());
        if (!s->tokens.empty()) { NEED(tokens); memcpy(tokens, s->tokens.data(), sizeof(int64_t) * s->tokens.size()); }
    });
}
int32_t k2hip_online_stream_get_timestamps(const k2hip_online_stream_t* s, int32_t* timestamps, int32_t cap) {
    return guard([&] {
        NEED(s);
        if ((int)s->timestamps.size() > cap) failf(K2HIP_ERR_CAPACITY, "stream holds %zu timestamps", s->timestamps.size());
        if (!s->timestamps.empty()) { NEED(timestamps); memcpy(timestamps, s->timestamps.data(), sizeof(int32_t) * s->timestamps.size()); }
    });
}
int32_t k2hip_online_stream_get_hyp(const k2hip_online_stream_t* s, int64_t* hyp2) {
    return guard([&] {
        NEED(s); NEED(hyp2);
        hyp2[0] = s->hyp[0];
        hyp2[1] = s->hyp[1];
    });
}
int32_t k2hip_online_stream_state(k2hip_online_stream_t* s, int32_t layer, int32_t kind, float* out, int64_t cap, int64_t* n) {
    return guard([&] {
        NEED(s); NEED(n);
        EngineLock lk(s->model->engine);
        s->model->engine.online_read_state(s->slot, layer, kind, s->chunks_done, out, cap, n);
    });
}

// one-part repeats of the vocabulary-parallel search since the model was created (test / monitoring hook: include/k2hip_debug.h)
int32_t k2hip_debug_search_retries(k2hip_model_t* model, int32_t* n) {
    return guard([&] {
        NEED(model); NEED(n);
        EngineLock lk(model->engine);
        *n = model->engine.search_retries();
    });
}
int32_t k2hip_debug_beam_trace(k2hip_model_t* model, int32_t* trace, int64_t cap_words, int32_t* B, int32_t* Tprime, int32_t* beam) {
    return guard([&] {
        NEED(model); NEED(B); NEED(Tprime); NEED(beam);
        EngineLock lk(model->engine);
        int b = 0, tp = 0, k = 0;
        const std::vector<int>& t = model->engine.last_beam_trace(&b, &tp, &k);
        K2_REQUIRE(!t.empty(), "beam trace: no traced beam search on record (set K2HIP_BEAM_TRACE=1 before the call)");
        *B = b; *Tprime = tp; *beam = k;
        if (!trace) return;
        if ((int64_t)t.size() > cap_words) failf(K2HIP_ERR_CAPACITY, "beam trace: %zu words exceed capacity %lld", t.size(), (long long)cap_words);
        memcpy(trace, t.data(), sizeof(int) * t.size());
    });
}
// test hook (include/k2hip_debug.h): the all-contexts decoder table against the decoder itself on sampled contexts
int32_t k2hip_debug_decoder_table_check(k2hip_model_t* model, int32_t n_samples, uint32_t seed,
                                                                                int64_t* rows, int64_t* mismatched) {
    return guard([&] {
        NEED(model); NEED(rows); NEED(mismatched);
        K2_REQUIRE(n_samples >= 0 && n_samples <= 65536, "decoder table check: %d samples", n_samples);
        EngineLock lk(model->engine);
        long long r = 0, m = 0;
        model->engine.decoder_table_check(n_samples, seed, &r, &m);
        *rows = r;
        *mismatched = m;
    });
}
// test hook (include/k2hip_debug.h): mark a stream as if a chunk step over it had failed on the device
int32_t k2hip_debug_poison_stream(k2hip_online_stream_t* s) {
    return guard([&] {
        NEED(s);
        s->poisoned = true;
    });
}

// test hook (include/k2hip_debug.h): does the stream's device mirror of its feature FIFO hold the FIFO right now?
int32_t k2hip_debug_stream_mirrored(const k2hip_online_stream_t* s, int32_t* ok) {
    return guard([&] {
        NEED(s); NEED(ok);
        *ok = s->mir_ok ? 1 : 0;
    });
}

// ---- tuning hook (include/k2hip_debug.h): time one GEMM shape/config on random data
int32_t k2hip_debug_set_switch(const char* env_name, int32_t value) {
    return guard([&] {
        NEED(env_name);
        tunables_init_from_env();  // so that a later model creation does not overwrite what is set here
        if (!tunables_set(env_name, value)) failf(K2HIP_ERR_INVALID, "unknown switch '%s'", env_name);
    });
}
int32_t k2hip_debug_gemm(k2hip_model_t* model, int32_t M, int32_t N, int32_t K,
                                                                 int32_t act, int32_t with_res, int32_t cfg, int32_t iters, float* ms) {
    return guard([&] {
        NEED(model); NEED(ms);
        EngineLock lk(model->engine);
        *ms = model->engine.debug_gemm(M, N, K, act, with_res != 0, iters, cfg, nullptr);
    });
}
// the same, and max_err = largest |difference| from the register-staged kernel on the same operands
int32_t k2hip_debug_gemm_check(k2hip_model_t* model, int32_t M, int32_t N, int32_t K,
                                                                       int32_t act, int32_t with_res, int32_t cfg, int32_t iters, float* ms,
                                                                       float* max_err) {
    return guard([&] {
        NEED(model); NEED(ms); NEED(max_err);
        EngineLock lk(model->engine);
        *ms = model->engine.debug_gemm(M, N, K, act, with_res != 0, iters, cfg, max_err);
    });
}

int32_t k2hip_debug_gemm_run(k2hip_model_t* model, const float* A, const float* W, const float* bias, const float* res, float* C, int32_t M,
                             int32_t N, int32_t K, int32_t act, int32_t glu, int32_t glu_cols, int32_t cfg) {
    return guard([&] {
        NEED(model); NEED(A); NEED(W); NEED(C);
        EngineLock lk(model->engine);
        model->engine.debug_gemm_host(A, W, bias, res, C, M, N, K, act, glu, glu_cols, cfg);
    });
}

int32_t k2hip_debug_gemm_trace(k2hip_model_t* model, int32_t M, int32_t N, int32_t K, int32_t act,
                                                                       int32_t with_res, int32_t cfg, unsigned long long* out, int64_t cap,
                                                                       int32_t* n_wg, int32_t* n_waves) {
    return guard([&] {
        NEED(model); NEED(out); NEED(n_wg); NEED(n_waves);
        K2_REQUIRE((cfg >= 100 && cfg < 1000) || cfg >= 2000 || cfg == 0 || cfg == 5 || cfg == 7 || (cfg >= 9 && cfg <= 11),
                   "trace: LDS-DMA (0, 5, 7, 9, 10, 11), ring (100+) and pipelined (2000+) configurations only");
        EngineLock lk(model->engine);
        model->engine.debug_gemm_trace(M, N, K, act, with_res != 0, cfg, out, cap, n_wg, n_waves);
    });
}

// ---- OfflineStream ---------------------------------------------------------------
int32_t k2hip_offline_stream_create(k2hip_model_t* model, k2hip_offline_stream_t** out) {
    return guard([&] {
        NEED(model); NEED(out);
        auto* s = new k2hip_offline_stream();
        s->model = model;
        s->tokens = {K2HIP_BLANK_ID, K2HIP_BLANK_ID};  // OfflineStream.cs:34
        *out = s;
    });
}
int32_t k2hip_offline_stream_destroy(k2hip_offline_stream_t* s) {
    return guard([&] {
        if (s && s->wav.capacity() > 0) {   // the buffer goes back to the model's pool (touched pages: no first-touch faults next time)
            std::lock_guard<std::mutex> lk(s->model->wav_mu);
            if (s->model->wav_pool.size() < k2hip_model::kPoolMax) {
                s->wav.clear();
                s->model->wav_pool.push_back(std::move(s->wav));
            }
        }
        delete s;
    });
}
// frames the stream's unmaterialised samples will add to Speech
static int64_t offline_pending_frames(const k2hip_offline_stream* s) {
    return s->wav.empty() ? 0 : s->model->engine.fbank_num_frames((int64_t)s->wav.size());
}
// run the fbank of the stream's unmaterialised samples now and append the frames to Speech; the samples behind the last whole frame
// shift stay queued (the streaming framing of an OnlineFbank)
static void offline_materialize(k2hip_offline_stream* s) {
    Engine& e = s->model->engine;
    const Config& c = e.model().cfg();
    const int64_t nf = offline_pending_frames(s);
    if (nf <= 0) return;
    const size_t old = s->speech.size();
    s->speech.resize(old + (size_t)nf * c.feat);
    try {
        int64_t got = 0;
        EngineLock lk(e);
        e.fbank_host(s->wav.data(), (int64_t)s->wav.size(), s->speech.data() + old, nf, &got);
    } catch (...) {
        s->speech.resize(old);
        throw;
    }
    s->wav.erase_front((size_t)nf * c.fbank.frame_shift);
}
// OfflineStream.AddSamples (:43-57): the samples are queued; their frames count in SpeechLength at once and are computed when
// somebody needs them (GetResults: on the device, for the whole batch in one launch)
int32_t k2hip_offline_stream_accept_samples(k2hip_offline_stream_t* s, const float* samples, int64_t n) {
    return guard([&] {
        NEED(s);
        if (n > 0) NEED(samples);
        if (n <= 0) return;
        if (s->wav.capacity() == 0) {   // first samples of the stream: a buffer of the model's pool that is large enough, if there is one
            std::lock_guard<std::mutex> lk(s->model->wav_mu);
            auto& pool = s->model->wav_pool;
            for (size_t i = pool.size(); i-- > 0;)
                if (pool[i].capacity() >= (size_t)n) {
                    s->wav = std::move(pool[i]);
                    pool.erase(pool.begin() + (long)i);
                    break;
                }
        }
        s->wav.append(s->model->engine, samples, (size_t)n);
    });
}
int64_t k2hip_offline_stream_speech_length(const k2hip_offline_stream_t* s) {
    return s ? (int64_t)s->speech.size() + offline_pending_frames(s) * s->model->engine.model().cfg().feat : -1;
}
int32_t k2hip_offline_stream_get_speech(const k2hip_offline_stream_t* s_, float* out, int64_t cap) {
    return guard([&] {
        NEED(s_);
        k2hip_offline_stream* s = const_cast<k2hip_offline_stream*>(s_);   // (materialising the frames does not change what the stream IS)
        offline_materialize(s);
        if ((int64_t)s->speech.size() > cap) failf(K2HIP_ERR_CAPACITY, "speech has %zu floats", s->speech.size());
        if (!s->speech.empty()) {
            NEED(out);
            memcpy(out, s->speech.data(), sizeof(float) * s->speech.size());
        }
    });
}

// OfflineStream.RemoveSamples (:58-68) after a search that consumed the stream's samples straight from `wav`: Speech = null when
// Tokens.Count > Context_size; otherwise the reference keeps the features -- they are materialised then (a rare path: one stream
// that emitted nothing).  Either way the samples behind the last whole frame shift stay queued.
static void offline_consume(k2hip_offline_stream* s, int ctx) {
    if ((int)s->tokens.size() > ctx) {
        const int64_t nf = offline_pending_frames(s);
        if (nf > 0) s->wav.erase_front((size_t)nf * s->model->engine.model().cfg().fbank.frame_shift);
        s->speech.clear();
        s->speech.shrink_to_fit();
    } else {
        offline_materialize(s);
    }
}

// OfflineRecognizer.GetResults -> ForwardBatchGreedySearch (:85-91, :189-303)
int32_t k2hip_offline_recognizer_get_results(k2hip_model_t* model, k2hip_offline_stream_t* const* streams, int32_t B) {
    return guard([&] {
        NEED(model); NEED(streams);
        K2_REQUIRE(B > 0, "GetResults: empty stream list");
        Engine& e = model->engine;
        const Config& c = e.model().cfg();
        bool from_samples = true;   // no stream holds materialised features: the whole batch goes samples -> tokens on the device
        for (int b = 0; b < B; b++) {
            NEED(streams[b]);
            K2_REQUIRE(streams[b]->model == model, "stream %d belongs to another model", b);
            from_samples = from_samples && streams[b]->speech.empty() && offline_pending_frames(streams[b]) > 0;
        }
        {   // one stream twice in the list would be decoded twice and have its samples consumed twice
            std::vector<const k2hip_offline_stream*> seen(streams, streams + B);
            std::sort(seen.begin(), seen.end());
            K2_REQUIRE(std::adjacent_find(seen.begin(), seen.end()) == seen.end(), "GetResults: the same stream appears twice in the list");
        }
        std::vector<const float*> ptrs(B);
        std::vector<int64_t> nfl(B);
        int64_t mx = 0;
        if (!from_samples)
            for (int b = 0; b < B; b++) offline_materialize(streams[b]);
        for (int b = 0; b < B; b++) {
            ptrs[b] = from_samples ? streams[b]->wav.data() : streams[b]->speech.data();
            nfl[b] = from_samples ? (int64_t)streams[b]->wav.size() : (int64_t)streams[b]->speech.size();
            mx = std::max(mx, from_samples ? offline_pending_frames(streams[b]) * c.feat : nfl[b]);
        }
        int T = (int)((mx + 80 * 19) / c.feat);
        int max_tokens = std::max(1, e.encoder_out_frames(T));
        std::vector<int64_t> tok((size_t)B * max_tokens);
        std::vector<int32_t> ts((size_t)B * max_tokens), n(B);
        {
            EngineLock lk(e);
            if (from_samples) e.offline_greedy_samples(ptrs.data(), nfl.data(), B, tok.data(), ts.data(), n.data(), max_tokens, false, /*pinned_src=*/true);
            else e.offline_greedy_feats(ptrs.data(), nfl.data(), B, false, tok.data(), ts.data(), n.data(), max_tokens);
        }
        auto remove_samples = [&](k2hip_offline_stream* s) {   // RemoveSamples (:294 / :418, OfflineStream.cs:58-68)
            if (from_samples) {
                offline_consume(s, c.ctx);
            } else if ((int)s->tokens.size() > c.ctx) {
                s->speech.clear();
                s->speech.shrink_to_fit();
            }
        };
        if (c.ctc) {
            // ForwardBatchGreedySearchCTC (:366-424): new symbols are appended to the stream's OWN lists (Tokens starts as
            // [blank, blank], OfflineStream.cs:34), timestamps carry FrameOffset, NumTrailingBlank accumulates
            const std::vector<int>&trail = e.last_trail(), &any = e.last_any();
            for (int b = 0; b < B; b++) {
                k2hip_offline_stream* s = streams[b];
                for (int k = 0; k < n[b]; k++) {
                    s->tokens.push_back(tok[(size_t)b * max_tokens + k]);
                    s->timestamps.push_back(ts[(size_t)b * max_tokens + k] + s->frame_offset);
                }
                s->num_trailing_blank = any[b] ? trail[b] : s->num_trailing_blank + trail[b];
                remove_samples(s);
            }
            return;
        }
        for (int b = 0; b < B; b++) {
            k2hip_offline_stream* s = streams[b];
            // tokens[m] / timestamps[m] are seeded with 2*B blanks / zeros (:250-267)
            s->tokens.assign((size_t)2 * B, K2HIP_BLANK_ID);
            s->tokens.insert(s->tokens.end(), tok.begin() + (size_t)b * max_tokens, tok.begin() + (size_t)b * max_tokens + n[b]);
            s->timestamps.insert(s->timestamps.end(), (size_t)2 * B, 0);  // Timestamps.AddRange (:293)
            s->timestamps.insert(s->timestamps.end(), ts.begin() + (size_t)b * max_tokens, ts.begin() + (size_t)b * max_tokens + n[b]);
            remove_samples(s);
        }
    });
}
// OfflineRecognizer.GetResult -> ForwardGreedySearch (:77-83, :93-187)
int32_t k2hip_offline_recognizer_get_result(k2hip_model_t* model, k2hip_offline_stream_t* s) {
    return guard([&] {
        NEED(model); NEED(s);
        K2_REQUIRE(s->model == model, "stream belongs to another model");
        Engine& e = model->engine;
        const Config& c = e.model().cfg();
        const bool from_samples = s->speech.empty() && offline_pending_frames(s) > 0;
        const int64_t n_fl = from_samples ? offline_pending_frames(s) * c.feat : (int64_t)s->speech.size();
        int T = (int)((n_fl + 80 * 19) / c.feat);
        int max_tokens = std::max(1, e.encoder_out_frames(T));
        std::vector<int64_t> tok(max_tokens);
        std::vector<int32_t> ts(max_tokens);
        int32_t n = 0;
        const float* p[1] = {from_samples ? s->wav.data() : s->speech.data()};
        int64_t nfl[1] = {from_samples ? (int64_t)s->wav.size() : (int64_t)s->speech.size()};
        {
            EngineLock lk(e);
            if (from_samples) e.offline_greedy_samples(p, nfl, 1, tok.data(), ts.data(), &n, max_tokens, true, /*pinned_src=*/true);
            else e.offline_greedy_feats(p, nfl, 1, true, tok.data(), ts.data(), &n, max_tokens);
        }
        s->tokens = {-1, K2HIP_BLANK_ID};  // hypList (:115-117, :180); the CTC single path seeds the same pair (:318-320)
        s->tokens.insert(s->tokens.end(), tok.begin(), tok.begin() + n);
        s->timestamps.insert(s->timestamps.end(), ts.begin(), ts.begin() + n);  // (:181; CTC :355 with frameOffset = 0)
        if (c.ctc) s->num_trailing_blank = e.last_any()[0] ? e.last_trail()[0] : e.last_trail()[0];  // local counter from 0 (:316,:354)
        // (the single paths never call RemoveSamples, :93-187 / :305-364: Speech stays -- materialised now if the search took the samples)
        if (from_samples) offline_materialize(s);
    });
}
int32_t k2hip_offline_stream_num_tokens(const k2hip_offline_stream_t* s) { return s ? (int32_t)s->tokens.size() : -1; }
int32_t k2hip_offline_stream_num_timestamps(const k2hip_offline_stream_t* s) { return s ? (int32_t)s->timestamps.size() : -1; }
int32_t k2hip_offline_stream_get_tokens(const k2hip_offline_stream_t* s, int64_t* tokens, int32_t cap) {
    return guard([&] {
        NEED(s);
        if ((int)s->tokens.size() > cap) failf(K2HIP_ERR_CAPACITY, "stream holds %zu tokens", s->tokens.size());
        if (!s->tokens.empty()) {
            NEED(tokens);
            memcpy(tokens, s->tokens.data(), sizeof(int64_t) * s->tokens.size());
        }
    });
}
int32_t k2hip_offline_stream_get_timestamps(const k2hip_offline_stream_t* s, int32_t* timestamps, int32_t cap) {
    return guard([&] {
        NEED(s);
        if ((int)s->timestamps.size() > cap) failf(K2HIP_ERR_CAPACITY, "stream holds %zu timestamps", s->timestamps.size());
        if (!s->timestamps.empty()) {
            NEED(timestamps);
            memcpy(timestamps, s->timestamps.data(), sizeof(int32_t) * s->timestamps.size());
        }
    });
}

}  // extern "C"
