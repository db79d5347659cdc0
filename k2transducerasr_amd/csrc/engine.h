// Engine = one model replica on one GPU: weights, HIP stream, device arena and the
// host-side orchestration of the offline path (pad -> Zipformer2 -> greedy).
#pragma once
#include <map>
#include <memory>
#include <mutex>

#include "kernels.h"
#include "model.h"

namespace k2hip {

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
    hipStream_t stream() const { return stream_; }

    int encoder_out_frames(int T) const;
    int64_t fbank_num_frames(int64_t n_samples) const;

    // ---- operator-level entry points (host buffers in/out) ----
    void fbank_host(const float* samples, int64_t n, float* feats, int64_t cap_frames, int64_t* n_frames);
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
                                int32_t* ts, int32_t* n_tokens, int max_tokens);
    void offline_greedy_samples_dev(const float* samples_dev, int64_t n_each, int B, int64_t* tokens, int32_t* ts,
                                    int32_t* n_tokens, int max_tokens);

    // ---- pipelined (asynchronous) form of offline_greedy_samples_dev ----
    // submit() enqueues fbank + pad + encoder on the encoder stream and the greedy loop + D2H
    // on a second stream, then returns; wait() blocks on that batch only.  With two batches in
    // flight the latency-bound greedy loop (32 workgroups) of batch i overlaps the MFMA-bound
    // encoder of batch i+1.  At most kSlots batches may be outstanding.
    static constexpr int kSlots = 2;
    int submit_samples_dev(const float* samples_dev, int64_t n_each, int B, int max_tokens);
    void wait_ticket(int ticket, int64_t* tokens, int32_t* ts, int32_t* n_tokens);

    void set_instrument(bool on) { instrument_ = on; }
    const k2hip_timing& timing() const { return timing_; }

    float debug_gemm(int M, int N, int K, int act, bool with_res, int iters);
    void* dev_alloc(int64_t bytes);
    void dev_free(void* p);
    void dev_upload(void* dst, const void* src, int64_t bytes);
    void synchronize();

  private:
    // device-side building blocks; all take a Ctx (dry run = sizing only)
    float* encoder_embed(const Ctx& c, const float* x, int B, int T, int* T50);
    void encoder_layer(const Ctx& c, int si, int li, float* x, const float* pe, int B, int T);
    float* encoder_stacks(const Ctx& c, float* x0, int B, int T50, int tap, float** tap_ptr, int* tap_dim, bool* tapped);
    float* encoder_forward(const Ctx& c, const float* x, int B, int T, int* Tp, int tap, float** tap_ptr, int* tap_rows,
                           int* tap_dim);
    void greedy_device(const Ctx& c, const float* enc, int B, int Tp, bool single, long long* d_tok, int* d_ts, int* d_n,
                       int max_tokens, int* d_overflow);
    const float* pos_emb(int T);  // cached CompactRelPositionalEncoding table on device
    DecJoinW decjoin() const;

    // run `body` once dry to size the arena, then for real
    template <typename F>
    void run_sized(F&& body);
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
        hipEvent_t enc_done = nullptr, done = nullptr;
        void* pin = nullptr;
        int64_t pin_cap = 0;
        long long* d_tok = nullptr;
        int *d_ts = nullptr, *d_n = nullptr, *d_ovf = nullptr;
        int B = 0, max_tokens = 0;
        bool busy = false;
    } slots_[kSlots];
    int next_slot_ = 0;
    hipStream_t stream2_ = nullptr;
    std::mutex mu_;
    std::map<int, float*> pe_cache_;
    bool instrument_ = false;
    GemmStats stats_;
    k2hip_timing timing_{};
    hipEvent_t ev_[8] = {nullptr};
    std::vector<hipEvent_t> evpool_;
    int evused_ = 0;
    // pinned staging for results
    void* pin_ = nullptr;
    int64_t pin_cap_ = 0;
    void* pinned(int64_t bytes);
};

}  // namespace k2hip
