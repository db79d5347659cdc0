// Engine = one model replica on one GPU: weights, HIP stream, device arena and the
// host-side orchestration of the offline path (pad -> Zipformer2 -> greedy).
#pragma once
#include <array>
#include <condition_variable>
#include <functional>
#include <thread>
#include <map>
#include <memory>
#include <mutex>
#include <tuple>

#include "kernels.h"
#include "model.h"

namespace k2hip {

// float offsets of one stream's caches inside its slot of the state pool (OnlineProjOfZipformer2.cs:63-111)
struct OnlineLayout {
    std::vector<long long> key, nonlin, val1, val2, conv1, conv2;  // per layer (Zipformer v1: nonlin = cached_avg, val1 = cached_val)
    std::vector<long long> clen;                                   // Zipformer v1: cached_len (one float) per layer
    std::vector<std::array<long long, 6>> sizes;                   // per layer, float counts in the reference's order
    long long embed = 0;
    long long floats_per_stream = 0;
    int nl = 0;
};

struct OfflineResult {
    std::vector<std::vector<int64_t>> tokens;   // emitted symbols per stream
    std::vector<std::vector<int32_t>> timestamps;
};

class Engine {
  public:
    Engine(const std::string& weights, const char* overrides, int device);
    ~Engine();

    const Model& model() const { return *model_; }
    std::mutex& mutex() { return mu_; }
    int device() const { return device_; }
    int search_retries() const { return search_retries_; }
    // test hook: rows of the all-contexts decoder table (0: the model has none) and how many floats of `n_samples` sampled rows
    // (+ the first, the start contexts' and the last) differ in their bits from the decoder run on those contexts
    void decoder_table_check(int n_samples, unsigned seed, long long* rows, long long* mismatched);
    hipStream_t stream() const { return stream_; }

    int encoder_out_frames(int T) const;
    int64_t fbank_num_frames(int64_t n_samples) const;

    // ---- operator-level entry points (host buffers in/out) ----
    void fbank_host(const float* samples, int64_t n, float* feats, int64_t cap_frames, int64_t* n_frames);
    // n_utts equal-length signals [n_utts, n] -> feats [n_utts, nf, feat] in one launch
    void fbank_host_batch(const float* samples, int64_t n, int n_utts, float* feats, int64_t nf);
    // the same for G signals that each arrive in two pieces ([head_g ; tail_g], n_head[g] + n_tail[g] == n for all g) and whose
    // frames go to G separate destinations: gathered straight into the pinned staging buffer and scattered straight out of it
    // (one host copy each way instead of three)
    // device mirror of the streams' feature FIFOs (see kernels.h fifo_*): rings of kFifoFrames frames per slot
    static constexpr int kFifoFrames = 512;
    void online_fifo_write(int slot, int pos, const float* feats, int64_t n_frames);  // host frames -> ring rows pos ..
    // (fifo_slots / fifo_pos: when given, the frames of signal g are also appended to slot fifo_slots[g]'s ring at fifo_pos[g] (< 0: not))
    void fbank_host_gather(const float* const* head, const int64_t* n_head, const float* const* tail, const int64_t* n_tail, int64_t n, int G,
                           float* const* dst, int64_t nf, const int* fifo_slots = nullptr, const int* fifo_pos = nullptr, bool defer = false);
    // defer = true: everything is enqueued (upload, fbank, the append to the device FIFOs, the download into pinned memory) but the
    // frames reach `dst` only in fbank_gather_finish(), which the caller runs after the next synchronisation of the stream it enqueues
    // on anyway (the chunk step's token download) -- a streaming tick then waits for the device once, not twice.  `dst` must stay
    // valid until then; at most one deferred gather is outstanding.
    void fbank_gather_finish();
    bool fbank_gather_pending() const { return fb_pending_.active; }
    void pad_host(const float* const* speech, const int64_t* n_floats, int B, int tail, float* out, int64_t cap, int64_t* L);
    void encoder_host(const float* x, int B, int T, float* enc_out, int64_t cap, int* Tp);
    void encoder_tap_host(const float* x, int B, int T, int tap, float* out, int64_t cap, int64_t* n);
    void decoder_host(const int64_t* y, int N, float* dec_out);
    void joiner_host(const float* enc, const float* dec, int N, float* logits);
    void greedy_host(const float* enc_out, int B, int Tp, bool single, int64_t* tokens, int32_t* ts, int32_t* n_tokens,
                     int max_tokens);
    // ---- fused paths ----
    void offline_greedy_feats(const float* const* feats, const int64_t* n_floats, int B, bool single, int64_t* tokens,
                              int32_t* ts, int32_t* n_tokens, int max_tokens);
    void offline_greedy_samples(const float* const* samples, const int64_t* n_samples, int B, int64_t* tokens,
                                int32_t* ts, int32_t* n_tokens, int max_tokens, bool single = false, bool pinned_src = false);
    void offline_greedy_samples_dev(const float* samples_dev, int64_t n_each, int B, int64_t* tokens, int32_t* ts,
                                    int32_t* n_tokens, int max_tokens);

    // ---- pipelined (asynchronous) form of offline_greedy_samples_dev ----
    // submit() enqueues fbank + pad + encoder on the encoder stream and the greedy loop + D2H
    // on a second stream, then returns; wait() blocks on that batch only.  With two batches in
    // flight the latency-bound greedy loop (one or two workgroups per stream) of batch i overlaps the MFMA-bound
    // encoder of batch i+1.  At most kSlots batches may be outstanding.  The third slot is for searches that are LONGER than an
    // encoder pass once they share the GPU with one (modified beam search: 4 launches per frame, each waiting for workgroup slots
    // between the encoder's whole-chip launches): every slot's search then runs on the slot's own stream, so the searches of
    // batches i and i+1 overlap each other and the encoder of batch i+2 (pipe mode 2, automatic with set_beam > 0).
    static constexpr int kSlots = 3;
    int submit_samples_dev(const float* samples_dev, int64_t n_each, int B, int max_tokens);
    int submit_samples_host(const float* samples_host, int64_t n_each, int B, int max_tokens);
    void wait_ticket(int ticket, int64_t* tokens, int32_t* ts, int32_t* n_tokens);

    // ---- streaming (OnlineRecognizer) path: online_engine.cpp ----
    int online_alloc_slot();   // GetEncoderInitStates: a zeroed slot of the device state pool
    void online_free_slot(int slot);
    void online_read_state(int slot, int layer, int kind, long long chunks_done, float* out, int64_t cap, int64_t* n);
    int online_tc50() const;
    int online_frames_per_chunk() const;
    // one tick over B streams that each have a full chunk: chunks [B][T*feat] (host), hyps [B][2], plens [B]
    // nchunks [B]: chunks each stream has decoded before this step (position of its attention rings)
    // chunks: [B] pointers to each stream's T*feat chunk floats (gathered into pinned staging by the engine)
    // operator level (IOnlineProj.EncoderProj): one chunk of the streaming Zipformer2 encoder for B state slots, encoder_out to host
    void online_encoder(const int* slots, const float* feats, const long long* plens, const int* nchunks, int B, float* enc_out);
    // fifo_heads (optional): ring row of every stream's first FIFO frame in the device mirror -- the chunk inputs are then gathered
    // on the device and `chunks` is not read
    void online_step(const int* slots, const float* const* chunks, const long long* hyps, const long long* plens, const int* nchunks, int B,
                     int64_t* tokens, int32_t* ts, int32_t* n_tokens, const int* fifo_heads = nullptr);

    void set_instrument(bool on) { instrument_ = on; }
    const std::vector<GemmLaunchRec>& gemm_log() const { return gemm_log_; }
    // decoding method of the batch entry points: 0 = greedy_search (the reference's only method), K >= 1 = modified beam
    // search with beam K (BASELINE.json configs[2]); the single-stream path is always greedy
    void set_beam(int k) { beam_ = k; }
    int beam() const { return beam_; }
    const std::vector<int>& last_trail() const { return last_trail_; }
    const std::vector<int>& last_any() const { return last_any_; }
    const std::vector<float>& last_scores() const { return last_scores_; }
    // K2HIP_BEAM_TRACE: [B][Tp][2 beam + 1] words of the last synchronous beam search (BeamArgs::trace), and its B / Tp / beam
    const std::vector<int>& last_beam_trace(int* B, int* Tp, int* K) const {
        *B = trace_B_; *Tp = trace_Tp_; *K = trace_K_;
        return last_beam_trace_;
    }
    const k2hip_timing& timing() const { return timing_; }

    float debug_gemm(int M, int N, int K, int act, bool with_res, int iters, int cfg, float* max_err);
    // test hook: ONE launch of configuration `cfg` (-1 = the dispatcher's own choice) on the caller's operands, result back to the host --
    // what tests/test_gemm_gpu.py holds against a float64 product computed on the host (k2hip_debug.h: k2hip_debug_gemm_run)
    void debug_gemm_host(const float* A, const float* W, const float* bias, const float* res, float* C, int M, int N, int K, int act,
                         int glu, int glu_cols, int cfg);
    void debug_gemm_trace(int M, int N, int K, int act, bool with_res, int cfg, unsigned long long* out, int64_t cap, int* n_wg, int* n_waves);
    void* dev_alloc(int64_t bytes);
    void dev_free(void* p);
    void dev_upload(void* dst, const void* src, int64_t bytes);
    void* host_alloc(int64_t bytes);
    void host_free(void* p);
    void synchronize();

  private:
    // device-side building blocks; all take a Ctx (dry run = sizing only)
    float* encoder_embed(const Ctx& c, const float* x, int B, int T, int* T50);
    void encoder_layer(const Ctx& c, int si, int li, float* x, const float* pe, int B, int T, const LayerTail* tail = nullptr);
    float* encoder_stacks(const Ctx& c, float* x0, int B, int T50, int tap, float** tap_ptr, int* tap_dim, bool* tapped, FullDimSegs* segs_out = nullptr);
    float* encoder_forward(const Ctx& c, const float* x, int B, int T, int* Tp, int tap, float** tap_ptr, int* tap_rows,
                           int* tap_dim);
    void greedy_device(const Ctx& c, const float* enc, int B, int Tp, bool single, long long* d_tok, int* d_ts, int* d_n,
                       int max_tokens, int* d_overflow);
    const float* pos_emb(int T);  // cached CompactRelPositionalEncoding table on device
    // linear_pos(pos_emb) of one layer: [rows, ncols] = pe [rows, pe_dim] . W^T.  It does not depend on the audio, so it is computed
    // once per (layer, rows) and kept on the device (a handful of utterance lengths in flight; the cache is dropped when it grows)
    const float* pos_proj_cached(const Ctx& c, int layer, const float* pe, int pe_dim, const float* W, int rows, int ncols);
    std::map<std::pair<int, int>, float*> pp_cache_;
    size_t pp_cache_bytes_ = 0;
    void ctc_device(const Ctx& c, const float* logp, int B, int Tp, long long* d_tok, int* d_ts, int* d_n, int max_tokens,
                    int* d_overflow);
    void beam_device(const Ctx& c, const float* enc, int B, int Tp, long long* d_tok, int* d_ts, int* d_n, int max_tokens,
                     int* d_overflow);
    // LSTM transducer (lstm_engine.cpp)
    int lstm_out_frames(int T) const;
    float* lstm_embed(const Ctx& c, const float* x, int B, int T, int* T_out);
    void lstm_layer(const Ctx& c, int li, float* x, const float* h0, int ldh0, float* cst, int B, int T, float* y);
    float* lstm_forward_seq(const Ctx& c, float* xe, int B, int T3, float* enc_out, int tap, float** tap_ptr, int* tap_dim);
    float* lstm_forward(const Ctx& c, const float* x, int B, int T, int* Tp, int tap, float** tap_ptr, int* tap_rows, int* tap_dim);
    float* lstm_chunk(const Ctx& c, const float* x, const int* d_slots, int B);
    // streaming Zipformer v1 (zipformer1_engine.cpp)
    const float* sinus_pos_emb(int Tc, int left, int D);
    std::map<std::tuple<int, int, int>, float*> sinus_cache_;  // (frames, left context, width) -> device table
    float* zip1_embed(const Ctx& c, const float* x, int B, int T, int* Tc_out);
    void zip1_layer(const Ctx& c, int si, const std::string& pfx, int l, float* x, const float* pp, const int* d_slots, int B, int Tc, int L);
    float* zip1_chunk(const Ctx& c, const float* x, const int* d_slots, int B, int* Tp_out);
    void zip1_layer_offline(const Ctx& c, int si, const std::string& pfx, int l, float* x, const float* pe, int B, int T);
    float* zip1_forward(const Ctx& c, const float* x, int B, int T, int* Tp, int tap, float** tap_ptr, int* tap_rows, int* tap_dim);
    // offline Conformer (conformer_engine.cpp)
    const float* conformer_pos_emb_left(int Tc, int left);
    float* conformer_chunk(const Ctx& c, const float* x, const int* d_slots, const long long* d_plen, int B, int* Tc_out);
    int conformer_out_frames(int T) const;
    const float* conformer_pos_emb(int T);
    float* conformer_embed(const Ctx& c, const float* x, int B, int T, int* T_out);
    void conformer_layer(const Ctx& c, int li, float* x, const float* pe, int B, int T);
    float* conformer_forward(const Ctx& c, const float* x, int B, int T, int* Tp, int tap, float** tap_ptr, int* tap_rows,
                             int* tap_dim);
    const float* pos_emb_stream(int Tc, int L);
    void online_ensure_pool();
    float* encoder_embed_stream(const Ctx& c, const float* x, const int* d_slots, int B, int T, int* Tc);
    void encoder_layer_stream(const Ctx& c, int si, int li, int l, float* x, const float* pe, const int* d_slots,
                              const long long* d_plen, const int* d_chunks, int B, int Tc, int L, const LayerTail* tail = nullptr);
    std::mutex cache_mu_;  // pos_proj / pos_emb / decjoin tables are built lazily
    float* online_encoder_zip2(const Ctx& c, const float* d_x, const int* d_slots, const long long* d_plen, const int* d_chunks, int B);
    DecJoinW decjoin();
    float* d_dec_table_ = nullptr;  // [(V + 1) V][J] decoder output of every context (small vocabularies), built on first use
    bool dec_table_tried_ = false;
    float* d_ptab_ = nullptr;  // [2][V][DD] per-token decoder-conv table (groups = 1 models), built on first use

    // run `body` once dry to size the arena, then for real
    template <typename F>
    void run_sized(F&& body);
    int submit_impl(const float* samples_dev, const float* samples_host, int64_t n_each, int B, int max_tokens);
    void finish_tokens(const long long* d_tok, const int* d_ts, const int* d_n, const int* d_ovf, int B, int max_tokens,
                       int64_t* tokens, int32_t* ts, int32_t* n_tokens);
    Ctx make_ctx(bool dry);

    std::unique_ptr<Model> model_;
    int device_;
    hipStream_t stream_ = nullptr;
    Arena arena_;
    Arena* cur_arena_ = &arena_;
    struct Slot {
        Arena arena;
        hipEvent_t enc_done = nullptr, done = nullptr, h2d_done = nullptr;
        hipStream_t stream = nullptr;  // pipe mode 1: the slot's whole pipeline (fbank .. search .. D2H) runs here
        void* pin = nullptr;
        int64_t pin_cap = 0;
        long long* d_tok = nullptr;
        int *d_ts = nullptr, *d_n = nullptr, *d_ovf = nullptr;
        int B = 0, max_tokens = 0;
        bool busy = false;
        GreedyLaunch greedy;            // the slot's search launch, kept for the one-part retry
        hipStream_t search_stream = nullptr;
    } slots_[kSlots];
    GreedyLaunch last_greedy_;          // the same for the synchronous entries (searches on stream_)
    void note_search(bool parted, bool timed_out);
    int one_part_left_ = 0, one_part_span_ = 0;   // searches still to run with one workgroup per stream / the current back-off span
    int search_retries_ = 0;            // one-part retries since the model was created (k2hip_debug_search_retries)
    int next_slot_ = 0;
    hipStream_t stream2_ = nullptr;
    // submit/wait overlap: 0 = search of batch i (stream2) under the encoder of batch i+1 (stream); 1 = every slot owns a
    // stream, so whole batches run concurrently and each other's GEMM prologues / epilogues / tails are filled; 2 = encoders in
    // order on `stream`, every slot's search on the slot's stream (chosen by itself for the beam search)
    float* online_pool_ = nullptr;
    float* online_fifo_ = nullptr;  // [online_cap_][kFifoFrames][feat]
    int online_cap_ = 0;
    std::vector<int> free_slots_;
    OnlineLayout lay_;
    std::mutex mu_;
    std::map<int, float*> pe_cache_;
    bool instrument_ = false;
    int beam_ = 0;
    // CTC search by-products of the last synchronous call (NumTrailingBlank bookkeeping, OfflineRecognizer.cs:392-397)
    int *d_trail_ = nullptr, *d_any_ = nullptr;
    std::vector<int> last_trail_, last_any_;
    float* d_scores_ = nullptr;
    int* d_beam_trace_ = nullptr;
    int trace_B_ = 0, trace_Tp_ = 0, trace_K_ = 0;
    std::vector<int> last_beam_trace_;
    float* d_dec_start_ = nullptr;  // [2][J]: decoder outputs of the start contexts [-1, blank], [blank, blank] (model constants)
    const float* decoder_start(const Ctx& c);
    std::vector<float> last_scores_;
    GemmStats stats_;
    k2hip_timing timing_{};
    hipEvent_t ev_[8] = {nullptr};
    std::vector<hipEvent_t> evpool_;
    std::vector<GemmLaunchRec> gemm_log_;
    int evused_ = 0;
    // pinned staging for results
    void* pin_ = nullptr;
    int64_t pin_cap_ = 0;
    void* pinned(int64_t bytes);
    void* pin_in_ = nullptr;   // pinned staging of a step's inputs (chunks), separate from the result staging above
    int64_t pin_in_cap_ = 0;
    void* pinned_in(int64_t bytes);
    void* pin_fb_ = nullptr;   // pinned staging of a deferred fbank gather (its frames are collected after the step that follows)
    int64_t pin_fb_cap_ = 0;
    void* pinned_fb(int64_t bytes);
    struct PendingFeats {
        bool active = false;
        std::vector<float*> dst;
        const char* src = nullptr;
        size_t per_out = 0;
    } fb_pending_;
};

// What every C-ABI entry holds while it works on a model: the model's mutex (calls on one handle are serialised) AND the model's
// device made current on the calling thread -- BEFORE anything is created lazily (slot streams, pinned staging, pool growth), so a
// host thread whose current device is another GPU (the C# "8 handles from 8 pool threads" story, IOnlineProj.cs:65-71 /
// SURVEY 8b "hipSetDevice per call") can never place a stream or a buffer on the wrong card.
class EngineLock {
  public:
    explicit EngineLock(Engine& e) : lk_(e.mutex()) { K2_HIP(hipSetDevice(e.device())); }

  private:
    std::lock_guard<std::mutex> lk_;
};

template <typename F>
void Engine::run_sized(F&& body) {
    K2_HIP(hipSetDevice(device_));
    stats_ = GemmStats();
    cur_arena_->reset();
    cur_arena_->set_dry(true);
    try {
        Ctx d = make_ctx(true);
        body(d);
    } catch (...) {
        cur_arena_->set_dry(false);
        cur_arena_->reset();
        throw;
    }
    cur_arena_->set_dry(false);
    int64_t need = cur_arena_->high_water();
    cur_arena_->reset();
    if (need > cur_arena_->capacity()) {
        K2_HIP(hipStreamSynchronize(stream_));
        cur_arena_->reserve(need + need / 8);
    }
    Ctx c = make_ctx(false);
    evused_ = 0;
    gemm_log_.clear();
    body(c);
    if (instrument_) {
        K2_HIP(hipStreamSynchronize(stream_));
        for (int i = 0; i + 1 < evused_; i += 2) {
            float ms = 0;
            K2_HIP(hipEventElapsedTime(&ms, evpool_[i], evpool_[i + 1]));
            stats_.ms += ms;
            if ((size_t)(i / 2) < gemm_log_.size()) gemm_log_[i / 2].us = ms * 1e3f;
        }
    }
    timing_.gemm_ms = stats_.ms;
    timing_.gemm_launches = stats_.launches;
    timing_.gemm_flops = stats_.flops;
    timing_.total_flops = stats_.total_flops;
}


}  // namespace k2hip
