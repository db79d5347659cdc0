// Model = parsed metadata map + every initializer resident in HBM.
//
// Replaces OfflineModel (K2TransducerAsr/OfflineModel.cs:23-73): where the
// reference creates three ONNXRuntime sessions and reads their custom-metadata
// maps, this loads one .k2w container (same string->string map, same keys) and
// uploads the weights once.  Some tensors are additionally repacked at load into
// the layouts the gfx950 kernels want (NHWC conv filters, k-major joiner matrix).
#pragma once
#include <map>
#include <string>
#include <unordered_map>

#include <memory>

#include "common.h"
#include "k2w_file.h"

namespace k2hip {

constexpr int kMaxStacks = 8;

struct Tensor {
    float* dev = nullptr;          // device pointer
    const float* host = nullptr;   // host view (mmap or repack buffer), valid for the model's lifetime
    int ndim = 0;
    int64_t dims[4] = {1, 1, 1, 1};
    int64_t numel() const { return dims[0] * dims[1] * dims[2] * dims[3]; }
};

struct FbankOpts {
    int sample_rate = 16000, frame_len = 400, frame_shift = 160, padded = 512, num_bins = 80;
    float preemph = 0.97f, low_freq = 20.f, high_freq = 0.f, input_scale = 1.f;
    int remove_dc = 1, snip_edges = 1;
    std::string window_type = "hamming";
};

struct Config {
    std::string model_type;
    int ns = 0;
    int dim[kMaxStacks] = {0}, nlayer[kMaxStacks] = {0}, ff[kMaxStacks] = {0}, heads[kMaxStacks] = {0},
        kern[kMaxStacks] = {0}, ds[kMaxStacks] = {0}, qhd[kMaxStacks] = {0}, vhd[kMaxStacks] = {0}, phd[kMaxStacks] = {0};
    int pos_dim = 48, J = 512, DD = 512, V = 500, ctx = 2, feat = 80, dmax = 0;
    int Vp = 0;  // vocab padded to a multiple of 4 for the k-major joiner matrix
    // streaming export (OnlineModel.cs:38-110): ChunkLength = T, ShiftLength = decode_chunk_len,
    // left_context_len per stack (already divided by the stack's downsampling factor)
    // model_type "conformer" (offline; OfflineRecognizer.cs:38-53 routes it to OfflineProjOfTransducer): one stack,
    // dim[0] / nlayer[0] / ff[0] / heads[0] / kern[0]; see conformer_engine.cpp
    bool conformer = false;
    // model_type "zipformer2ctc": Zipformer2 encoder + CTC head (ctc_output.1), no decoder / joiner; the encoder entry points
    // return log_probs [B,T',V] (OfflineProjOfZipformer2ctc.cs:48-92, OnlineProjOfZipformer2ctc)
    bool ctc = false;
    // model_type "lstm" (icefall lstm_transducer_stateless2; offline via OfflineProjOfTransducer, streaming via OnlineProjOfLstm):
    // dim[0] = d_model, ff[0], nlayer[0], rnn_hidden; chunk_T = 9, shift = 4, one encoder frame per chunk
    bool lstm = false;
    int rnn_hidden = 0;
    // model_type "zipformer" (streaming Zipformer v1, OnlineProjOfZipformer): att[] = attention_dims; qhd = att / heads,
    // vhd = att / 2 / heads, phd = pos_dim (4); chunk_T = shift + 7; see zipformer1_engine.cpp
    bool zip1 = false;
    int att[kMaxStacks] = {0};
    int enc_dim() const { return ctc ? V : J; }
    int conv_cpg = 4;  // decoder conv input channels per group (4: Zipformer recipes; DD: stateless2 decoder, groups = 1)
    bool streaming = false;
    int chunk_T = 0, shift = 0, left[kMaxStacks] = {0};
    int right = 0;   // streaming Conformer: right_context (OnlineModel.cs:161-165), encoder frames of look-ahead per chunk
    FbankOpts fbank;
};

class Model {
  public:
    Model(const std::string& path, const char* overrides);  // host only: parse, validate, repack
    void upload(int device);                                 // copy everything into HBM of `device`
    ~Model();
    Model(const Model&) = delete;
    Model& operator=(const Model&) = delete;

    const Config& cfg() const { return cfg_; }
    int device() const { return device_; }
    const std::map<std::string, std::string>& meta() const { return meta_; }

    // device pointer of a named tensor; throws if absent
    const float* w(const std::string& name) const;
    const float* wf(const char* fmt, ...) const __attribute__((format(printf, 2, 3)));
    const Tensor& tensor(const std::string& name) const;
    bool has(const std::string& name) const { return t_.count(name) != 0; }

    // fbank tables (device): window [frame_len], mel weights [num_bins, padded/2]
    const float* d_window = nullptr;
    const float* d_melw = nullptr;
    const float* d_melrange = nullptr;  // [num_bins][2]: first / one-past-last FFT bin with a non-zero weight

  private:
    void parse_config();
    void add_repacks(std::vector<std::pair<std::string, std::vector<float>>>& extra,
                     std::vector<std::pair<std::string, std::vector<int64_t>>>& shapes);
    void validate_shapes() const;
    int device_ = -1;
    std::vector<std::pair<std::string, std::vector<float>>> extra_;            // repacks waiting for upload()
    std::vector<std::pair<std::string, std::vector<int64_t>>> extra_shapes_;
    std::unique_ptr<K2wFile> file_;  // the mapped container (host views of the tensors point into it)
    struct DevBlob {                 // [file data region | repacked tensors] in HBM
        void* p = nullptr;
        int dev = 0;
        void alloc(int device, size_t bytes);
        ~DevBlob();
    } blob_;
    std::map<std::string, std::string> meta_;
    std::unordered_map<std::string, Tensor> t_;
    std::vector<std::vector<float>> host_keep_;
    Config cfg_;
};

}  // namespace k2hip
