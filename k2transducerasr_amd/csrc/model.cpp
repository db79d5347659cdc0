#include "model.h"
#include "kernels.h"

#include <cmath>
#include <cstdlib>

namespace k2hip {

namespace {

std::vector<int> csv_ints(const std::string& s) {
    std::vector<int> v;
    const char* p = s.c_str();
    while (*p) {
        char* e;
        long x = strtol(p, &e, 10);
        if (e == p) break;
        v.push_back((int)x);
        p = (*e == ',') ? e + 1 : e;
    }
    return v;
}

// MelScale in f64 (rounded once to f32 below): see add_repacks
double mel_scale(double f) { return 1127.0 * log(1.0 + f / 700.0); }

}  // namespace

Model::Model(const std::string& path, const char* overrides) {
    // Every resource acquired below is owned by a member (file_: mmap, blob_: device memory), so a throw anywhere in this
    // constructor releases them (members are destroyed; ~Model itself does not run for a half-built object).
    file_.reset(new K2wFile(path));
    meta_ = file_->meta;
    if (overrides && *overrides) {
        std::string s(overrides);
        size_t a = 0;
        while (a < s.size()) {
            size_t b = s.find(';', a);
            if (b == std::string::npos) b = s.size();
            std::string kv = s.substr(a, b - a);
            size_t eq = kv.find('=');
            if (eq == std::string::npos) failf(K2HIP_ERR_INVALID, "bad override '%s' (want key=value)", kv.c_str());
            meta_[kv.substr(0, eq)] = kv.substr(eq + 1);
            a = b + 1;
        }
    }
    const uint8_t* data = file_->data();
    // tensors the engine never reads (an ONNX export's int64 shape constants that slipped through an importer) stay in
    // the table but are not views: only f32 entries become Tensors
    for (const K2wTensorRec& r : file_->tensors) {
        if (r.dtype != 0) continue;
        Tensor t;
        t.ndim = r.ndim;
        for (int k = 0; k < 4; k++) t.dims[k] = r.dims[k];
        t.host = reinterpret_cast<const float*>(data + r.off);
        t_[r.name] = t;
    }
    parse_config();
    validate_shapes();  // before anything indexes a tensor by the CONFIG's dimensions
    add_repacks(extra_, extra_shapes_);  // repacked copies (host side)
}

// Second phase: everything above is host-only, so a bad container is reported (K2HIP_ERR_IO / _INVALID) on a machine without a
// GPU too; the upload needs the device.
void Model::upload(int device) {
    device_ = device;
    auto& extra = extra_;
    auto& shapes = extra_shapes_;
    const uint8_t* data = file_->data();
    // device upload: [file data region | repacks]
    K2_HIP(hipSetDevice(device_));
    size_t file_bytes = file_->data_bytes();
    size_t extra_bytes = 0;
    for (auto& e : extra) extra_bytes += (size_t)align_up((int64_t)e.second.size() * 4, 256);
    size_t total = (size_t)align_up((int64_t)file_bytes, 256) + extra_bytes;
    blob_.alloc(device_, total);
    void* dev_blob_ = blob_.p;
    K2_HIP(copy_blocking(dev_blob_, data, file_bytes, hipMemcpyHostToDevice));
    for (const K2wTensorRec& r : file_->tensors)
        if (r.dtype == 0) t_[r.name].dev = reinterpret_cast<float*>(static_cast<char*>(dev_blob_) + r.off);
    size_t off = (size_t)align_up((int64_t)file_bytes, 256);
    for (size_t i = 0; i < extra.size(); i++) {
        auto& e = extra[i];
        K2_HIP(copy_blocking(static_cast<char*>(dev_blob_) + off, e.second.data(), e.second.size() * 4, hipMemcpyHostToDevice));
        Tensor t;
        t.dev = reinterpret_cast<float*>(static_cast<char*>(dev_blob_) + off);
        t.ndim = (int)shapes[i].second.size();
        for (int k = 0; k < t.ndim; k++) t.dims[k] = shapes[i].second[k];
        host_keep_.push_back(std::move(e.second));
        t.host = host_keep_.back().data();
        t_[e.first] = t;
        off += (size_t)align_up((int64_t)host_keep_.back().size() * 4, 256);
    }
    d_window = w("#fbank.window");
    d_melw = w("#fbank.melw");
    d_melrange = w("#fbank.melrange");
    extra_.clear();
    extra_shapes_.clear();
}

Model::~Model() = default;  // file_ unmaps, blob_ frees

void Model::DevBlob::alloc(int device, size_t bytes) {
    dev = device;
    K2_HIP(hipMalloc(&p, bytes));
}
Model::DevBlob::~DevBlob() {
    if (p) {
        (void)hipSetDevice(dev);
        (void)hipFree(p);
    }
}

// The tensors whose shapes the kernels take from the CONFIG (metadata) rather than from the tensor: a container whose
// metadata and weights disagree must fail here (K2HIP_ERR_IO), not read out of bounds on the device.
void Model::validate_shapes() const {
    const Config& c = cfg_;
    auto want = [&](const char* name, std::initializer_list<int64_t> dims) {
        if (!has(name)) return;
        const Tensor& t = tensor(name);
        bool ok = t.ndim == (int)dims.size();
        int k = 0;
        for (int64_t d : dims) ok = ok && t.dims[k++] == d;
        if (!ok)
            failf(K2HIP_ERR_IO, "tensor %s is [%lld,%lld,%lld,%lld] (ndim %d), which does not match the metadata", name, (long long)t.dims[0],
                  (long long)t.dims[1], (long long)t.dims[2], (long long)t.dims[3], t.ndim);
    };
    want("decoder.embedding.weight", {c.V, c.DD});
    want("joiner.decoder_proj.weight", {c.J, c.DD});
    want("joiner.decoder_proj.bias", {c.J});
    want("joiner.output_linear.weight", {c.V, c.J});
    want("joiner.output_linear.bias", {c.V});
    want("joiner.encoder_proj.bias", {c.J});
    if (has("joiner.encoder_proj.weight")) {
        const Tensor& t = tensor("joiner.encoder_proj.weight");
        if (t.ndim != 2 || t.dims[0] != c.J) failf(K2HIP_ERR_IO, "joiner.encoder_proj.weight has %lld rows, joiner_dim is %d", (long long)t.dims[0], c.J);
    }
    if (c.ctc) {
        want("ctc_output.1.bias", {c.V});
        if (has("ctc_output.1.weight") && tensor("ctc_output.1.weight").dims[0] != c.V)
            failf(K2HIP_ERR_IO, "ctc_output.1.weight has %lld rows, vocab_size is %d", (long long)tensor("ctc_output.1.weight").dims[0], c.V);
    }
}

const Tensor& Model::tensor(const std::string& name) const {
    auto it = t_.find(name);
    if (it == t_.end()) failf(K2HIP_ERR_IO, "weights file has no tensor '%s'", name.c_str());
    return it->second;
}
const float* Model::w(const std::string& name) const { return tensor(name).dev; }
const float* Model::wf(const char* fmt, ...) const {
    char buf[256];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    return w(buf);
}

void Model::parse_config() {
    auto get = [&](const char* k, const char* dflt) -> std::string {
        auto it = meta_.find(k);
        return it == meta_.end() ? std::string(dflt) : it->second;
    };
    auto geti = [&](const char* k, int d) { return atoi(get(k, std::to_string(d).c_str()).c_str()); };
    auto getf = [&](const char* k, float d) {
        auto it = meta_.find(k);
        return it == meta_.end() ? d : (float)atof(it->second.c_str());
    };
    Config& c = cfg_;
    // ---- Model_type as the reference derives it ------------------------------------------------------------------------------
    // OfflineModel.cs:51-63: Model_type = encoder metadata "model_type" (or ""); a `comment` whose LOWER-CASED text contains both
    // "ctc" and "zipformer2" overrides it with "zipformer2ctc" (icefall's CTC exports carry model_type "zipformer2" and say "ctc" only
    // in the comment).  OnlineModel.cs:96-106: the same test WITHOUT the lower-casing, and the result is model_type + "ctc".
    // OfflineRecognizer.cs:38-53 then routes "zipformer2ctc" to the CTC operator and EVERYTHING else -- unknown and empty strings
    // included (:50-52) -- to the transducer operator; OnlineRecognizer.cs:26-44 has no default case (an unknown type leaves
    // _onlineProj null and the first GetResults dereferences it).  The reference's operator is graph-agnostic (the ONNX file is the
    // graph); here the type also names the encoder graph, so the transducer default is resolved from the architecture keys the
    // container carries.  The derived value is written back: k2hip_model_meta("model_type") is what CustomMetadata.Model_type holds.
    const bool streaming_file = get("streaming", "0") == "1";
    std::string mt = get("model_type", "");
    {
        std::string comment = get("comment", "");
        if (!streaming_file)
            for (auto& ch : comment) ch = (char)tolower((unsigned char)ch);
        if (!comment.empty() && comment.find("ctc") != std::string::npos && comment.find("zipformer2") != std::string::npos) {
            if (!streaming_file) mt = "zipformer2ctc";
            else if (mt.size() < 3 || mt.compare(mt.size() - 3, 3, "ctc") != 0) mt += "ctc";   // (the reference would make "...ctcctc" of a type that already says ctc and then find no operator for it)
        }
    }
    const bool known = mt == "zipformer2" || mt == "zipformer2ctc" || mt == "zipformer" || mt == "conformer" || mt == "lstm";
    if (!known) {
        if (streaming_file)
            failf(K2HIP_ERR_UNSUPPORTED, "model_type '%s' has no streaming operator (OnlineRecognizer.cs:26-44 knows zipformer, zipformer2, zipformer2ctc, lstm, conformer and has no default case)",
                  mt.c_str());
        // OfflineRecognizer.cs:50-52: the transducer operator; which graph, the architecture keys say
        const char* graph = meta_.count("rnn_hidden_size") ? "lstm" : meta_.count("attention_dims") ? "zipformer" : meta_.count("query_head_dims") ? "zipformer2" : csv_ints(get("encoder_dims", "")).size() == 1 ? "conformer" : nullptr;
        if (!graph)
            failf(K2HIP_ERR_UNSUPPORTED, "model_type '%s' routes to the transducer operator (OfflineRecognizer.cs:50-52) but the container's metadata does not say which encoder graph it holds (have: zipformer2, zipformer, conformer, lstm)",
                  mt.c_str());
        meta_["model_type_as_given"] = mt;
        mt = graph;
    }
    meta_["model_type"] = mt;
    c.model_type = mt;
    c.conformer = c.model_type == "conformer";
    c.ctc = c.model_type == "zipformer2ctc";
    c.lstm = c.model_type == "lstm";
    c.zip1 = c.model_type == "zipformer";
    auto fill = [&](const char* k, int* dst) {
        auto v = csv_ints(get(k, ""));
        if ((int)v.size() > kMaxStacks) failf(K2HIP_ERR_INVALID, "metadata %s has too many entries", k);
        for (size_t i = 0; i < v.size(); i++) dst[i] = v[i];
        return (int)v.size();
    };
    c.ns = fill("encoder_dims", c.dim);
    if (c.ns <= 0) failf(K2HIP_ERR_INVALID, "metadata encoder_dims missing");
    const char* keys[] = {"num_encoder_layers", "feedforward_dims", "num_heads",      "cnn_module_kernels",
                          "downsampling_factors", "query_head_dims", "value_head_dims", "pos_head_dims"};
    int* dsts[] = {c.nlayer, c.ff, c.heads, c.kern, c.ds, c.qhd, c.vhd, c.phd};
    if (c.lstm) {
        K2_REQUIRE(c.ns == 1 && fill("num_encoder_layers", c.nlayer) == 1 && fill("feedforward_dims", c.ff) == 1,
                   "lstm: encoder_dims / num_encoder_layers / feedforward_dims must have one entry");
        c.rnn_hidden = geti("rnn_hidden_size", 0);
        K2_REQUIRE(c.rnn_hidden > 0 && c.rnn_hidden % 4 == 0 && c.dim[0] == geti("d_model", c.dim[0]), "lstm: bad d_model / rnn_hidden_size");
        c.heads[0] = 1; c.kern[0] = 1; c.ds[0] = 1; c.qhd[0] = 32; c.phd[0] = 4; c.vhd[0] = 12;  // unused
    }
    if (c.zip1) {
        K2_REQUIRE(fill("attention_dims", c.att) == c.ns, "metadata attention_dims must have %d entries", c.ns);
        for (int i = 0; i < 5; i++)
            if (fill(keys[i], dsts[i]) != c.ns) failf(K2HIP_ERR_INVALID, "metadata %s must have %d entries", keys[i], c.ns);
        K2_REQUIRE(geti("pos_dim", 4) == 4, "zipformer: pos_dim %d unsupported (4)", geti("pos_dim", 4));
        for (int i = 0; i < c.ns; i++) {
            K2_REQUIRE(c.heads[i] > 0 && c.att[i] % (2 * c.heads[i]) == 0 && (c.att[i] / c.heads[i]) % 4 == 0 && c.att[i] % 8 == 0,
                       "zipformer: attention_dims[%d]=%d with %d heads unsupported", i, c.att[i], c.heads[i]);
            K2_REQUIRE(i == 0 || c.dim[i] >= c.dim[i - 1], "zipformer: encoder_dims must not shrink (stack %d: %d -> %d)", i, c.dim[i - 1], c.dim[i]);
            K2_REQUIRE(c.ds[i] > 1 || i == 0 || c.dim[i] == c.dim[i - 1], "zipformer: stack %d has downsampling 1 but changes width", i);
            c.qhd[i] = 32; c.phd[i] = 4; c.vhd[i] = c.att[i] / 2 / c.heads[i];  // qhd: keep the Zipformer2 checks below quiet
        }
    }
    for (int i = 0; i < (c.lstm || c.zip1 ? 0 : c.conformer ? 4 : 8); i++)
        if (fill(keys[i], dsts[i]) != c.ns) failf(K2HIP_ERR_INVALID, "metadata %s must have %d entries", keys[i], c.ns);
    if (c.conformer) {
        K2_REQUIRE(c.ns == 1, "conformer: encoder_dims must have one entry");
        K2_REQUIRE(c.dim[0] % c.heads[0] == 0 && (c.dim[0] / c.heads[0]) % 4 == 0, "conformer: head size must be a multiple of 4");
        c.ds[0] = 1;
        c.qhd[0] = 32; c.phd[0] = 4; c.vhd[0] = 12;  // unused; keep the Zipformer checks below quiet
    }
    c.pos_dim = geti("pos_dim", 48);
    c.J = geti("joiner_dim", 512);
    c.DD = geti("decoder_dim", 512);
    c.V = geti("vocab_size", 500);
    c.ctx = geti("context_size", 2);
    c.feat = geti("feature_dim", 80);
    c.Vp = (int)align_up(c.V, 4);
    if (c.lstm) {  // the same file serves the offline and the streaming operator (OnlineModel.cs:48-49: ChunkLength = T, ShiftLength)
        c.chunk_T = geti("T", 9);
        c.shift = geti("decode_chunk_len", 4);
        K2_REQUIRE(c.chunk_T == 9 && c.shift == 4, "lstm streaming geometry T=%d, decode_chunk_len=%d unsupported (9 / 4)", c.chunk_T, c.shift);
    }
    c.streaming = get("streaming", "0") == "1";
    if (c.streaming) {
        c.chunk_T = geti("T", 45);
        c.shift = geti("decode_chunk_len", 32);
        if (c.conformer) {
            // OnlineProjOfConformer (OnlineModel.cs:131-166): one left_context, T = (chunk_size + 2 + right_context) * 4 + 3
            c.left[0] = geti("left_context", 64);
            const int chunk = geti("chunk_size", 16);
            c.right = geti("right_context", 0);
            K2_REQUIRE(c.right >= 0 && c.right <= 64, "streaming conformer: right_context %d out of range", c.right);
            K2_REQUIRE(c.shift == 4 * chunk && c.chunk_T == (chunk + 2 + c.right) * 4 + 3 && c.left[0] > 0 && c.left[0] % 4 == 0,
                       "streaming conformer geometry T=%d, decode_chunk_len=%d, chunk_size=%d, left_context=%d, right_context=%d unsupported",
                       c.chunk_T, c.shift, chunk, c.left[0], c.right);
        } else if (c.zip1) {
            if (fill("left_context_len", c.left) != c.ns) failf(K2HIP_ERR_INVALID, "metadata left_context_len must have %d entries", c.ns);
            K2_REQUIRE(c.chunk_T == c.shift + 7 && c.shift % 4 == 0, "zipformer streaming geometry T=%d, decode_chunk_len=%d unsupported", c.chunk_T, c.shift);
            for (int i = 0; i < c.ns; i++) K2_REQUIRE(c.left[i] > 0, "zipformer: left_context_len[%d] must be positive", i);
        } else {
            if (fill("left_context_len", c.left) != c.ns) failf(K2HIP_ERR_INVALID, "metadata left_context_len must have %d entries", c.ns);
            K2_REQUIRE(c.chunk_T == c.shift + 13 && c.shift % 4 == 0, "streaming geometry T=%d, decode_chunk_len=%d unsupported", c.chunk_T, c.shift);
        }
    }
    c.dmax = 0;
    for (int i = 0; i < c.ns; i++) {
        c.dmax = std::max(c.dmax, c.dim[i]);
        K2_REQUIRE(c.dim[i] % 16 == 0, "encoder_dims[%d]=%d must be a multiple of 16", i, c.dim[i]);
        K2_REQUIRE(c.qhd[i] == 32, "query_head_dims[%d]=%d: kernels are built for 32", i, c.qhd[i]);
        K2_REQUIRE(c.phd[i] == 4, "pos_head_dims[%d]=%d: kernels are built for 4", i, c.phd[i]);
        K2_REQUIRE(c.kern[i] % 2 == 1 && c.kern[i] <= 63, "cnn_module_kernels[%d]=%d unsupported", i, c.kern[i]);
        K2_REQUIRE(c.ds[i] >= 1 && c.ds[i] <= 16, "downsampling_factors[%d]=%d unsupported", i, c.ds[i]);
    }
    K2_REQUIRE(c.ctx == 2, "context_size %d: the reference's greedy loops seed a 2-entry hyp", c.ctx);
    K2_REQUIRE(c.feat == 80, "feature_dim %d: Conv2dSubsampling geometry is built for 80 bins", c.feat);
    K2_REQUIRE(c.J % 4 == 0 && c.DD % 4 == 0, "joiner_dim/decoder_dim must be multiples of 4");
    FbankOpts& f = c.fbank;
    f.sample_rate = geti("sample_rate", 16000);
    f.frame_len = f.sample_rate * geti("frame_length_ms", 25) / 1000;
    f.frame_shift = f.sample_rate * geti("frame_shift_ms", 10) / 1000;
    f.padded = 1;
    while (f.padded < f.frame_len) f.padded <<= 1;
    f.num_bins = c.feat;
    f.preemph = getf("preemph_coeff", 0.97f);
    f.low_freq = getf("low_freq", 20.f);
    f.high_freq = getf("high_freq", 0.f);
    f.input_scale = getf("input_scale", 1.f);
    f.remove_dc = geti("remove_dc_offset", 1);
    f.snip_edges = geti("snip_edges", 1);
    f.window_type = get("window_type", "hamming");
    K2_REQUIRE(f.snip_edges == 1, "snip_edges=0 (whisper framing, OfflineStream.cs:27-32) is out of scope");
    K2_REQUIRE(f.padded == 512 && f.frame_len <= 512, "fbank kernel is built for a 512-point FFT (got %d)", f.padded);
}

// float -> IEEE binary16 bits, round to nearest even (overflow -> inf, NaN kept)
static uint16_t f32_to_f16_rne(float f) {
    uint32_t x;
    memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    x &= 0x7fffffffu;
    if (x >= 0x7f800000u) return (uint16_t)(sign | 0x7c00u | (x > 0x7f800000u ? 0x200u : 0));   // inf / NaN
    if (x >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);                                      // >= 65520 rounds to inf
    if (x < 0x33000001u) return (uint16_t)sign;                                                   // < 2^-25 (+ tie at 2^-25 -> even 0) rounds to zero
    int e = (int)(x >> 23) - 127;
    uint32_t m = (x & 0x7fffffu) | 0x800000u;   // 24-bit significand
    int shift = e < -14 ? (13 + (-14 - e)) : 13;  // bits dropped (subnormal results drop more)
    uint32_t q = m >> shift, rem = m & ((1u << shift) - 1), half = 1u << (shift - 1);
    if (rem > half || (rem == half && (q & 1))) q++;
    if (e < -14) return (uint16_t)(sign | q);     // subnormal (q may carry into the smallest normal: the encoding is continuous)
    return (uint16_t)(sign | (((uint32_t)(e + 15) << 10) + (q - 0x400u)));   // q in [0x400, 0x800]: a carry bumps the exponent
}

void Model::add_repacks(std::vector<std::pair<std::string, std::vector<float>>>& extra,
                        std::vector<std::pair<std::string, std::vector<int64_t>>>& shapes) {
    const Config& c = cfg_;
    auto push = [&](const std::string& name, std::vector<float>&& v, std::vector<int64_t> shp) {
        extra.emplace_back(name, std::move(v));
        shapes.emplace_back(name, std::move(shp));
    };
    // A container may hold the decoder + joiner only (the reference also loads three separate
    // sessions, OfflineModel.cs:25-27); encoder entry points then fail with "no tensor ...".
    if (has("decoder.conv.weight")) {
        const Tensor& t = tensor("decoder.conv.weight");
        K2_REQUIRE(t.ndim == 3 && t.dims[0] == c.DD && t.dims[2] == c.ctx && c.DD % t.dims[1] == 0,
                   "decoder.conv.weight is [%lld,%lld,%lld], config says [%d,*,%d]", (long long)t.dims[0], (long long)t.dims[1],
                   (long long)t.dims[2], c.DD, c.ctx);
        cfg_.conv_cpg = (int)t.dims[1];
        if (cfg_.conv_cpg > 4) {  // wide groups (stateless2: groups = 1): k-major [cpg*ctx][DD] for a coalesced GEMV
            const int cpg = cfg_.conv_cpg, KK = cpg * c.ctx;
            std::vector<float> v((size_t)KK * c.DD);
            for (int co = 0; co < c.DD; co++)
                for (int k = 0; k < KK; k++) v[(size_t)k * c.DD + co] = t.host[(size_t)co * KK + k];
            push("decoder.conv.weight#kn", std::move(v), {KK, c.DD});
            // per-tap matrices [co][ci] for the per-token table build (Engine::decjoin)
            for (int tap = 0; tap < c.ctx; tap++) {
                std::vector<float> u((size_t)c.DD * cpg);
                for (int co = 0; co < c.DD; co++)
                    for (int ci = 0; ci < cpg; ci++) u[(size_t)co * cpg + ci] = t.host[((size_t)co * cpg + ci) * c.ctx + tap];
                push("decoder.conv.weight#tap" + std::to_string(tap), std::move(u), {c.DD, cpg});
            }
        }
    }
    if ((c.conformer || c.lstm || c.zip1) && has("encoder.encoder_embed.conv.0.weight")) {
        for (const char* nm : {"encoder.encoder_embed.conv.3.weight", "encoder.encoder_embed.conv.6.weight"}) {
            const Tensor& t = tensor(nm);
            int Co = (int)t.dims[0], Ci = (int)t.dims[1];
            std::vector<float> v((size_t)Co * 9 * Ci);
            for (int co = 0; co < Co; co++)
                for (int ci = 0; ci < Ci; ci++)
                    for (int kt = 0; kt < 3; kt++)
                        for (int kf = 0; kf < 3; kf++)
                            v[((size_t)co * 9 + kt * 3 + kf) * Ci + ci] = t.host[(((size_t)co * Ci + ci) * 3 + kt) * 3 + kf];
            push(std::string(nm) + "#ohwi", std::move(v), {Co, 9 * Ci});
        }
        {   // out Linear [D, c*F3+f] -> [D, f*128+c] (NHWC flatten order)
            const Tensor& t = tensor("encoder.encoder_embed.out.weight");
            int D0 = (int)t.dims[0], KK = (int)t.dims[1], C = 128, F3 = KK / C;
            std::vector<float> v((size_t)D0 * KK);
            for (int d = 0; d < D0; d++)
                for (int ch = 0; ch < C; ch++)
                    for (int f = 0; f < F3; f++) v[(size_t)d * KK + f * C + ch] = t.host[(size_t)d * KK + ch * F3 + f];
            push("encoder.encoder_embed.out.weight#fc", std::move(v), {D0, KK});
        }
        for (int li = 0; li < (c.conformer ? c.nlayer[0] : 0); li++) {
            char nm[192];
            snprintf(nm, sizeof nm, "encoder.encoder.layers.%d.conv_module.depthwise_conv.weight", li);
            const Tensor& t = tensor(nm);
            int D = (int)t.dims[0], K = (int)t.dims[2];
            K2_REQUIRE(D == c.dim[0] && K == c.kern[0], "%s has shape [%d,1,%d], config says [%d,1,%d]", nm, D, K, c.dim[0], c.kern[0]);
            std::vector<float> v((size_t)K * D);
            for (int d = 0; d < D; d++)
                for (int kk = 0; kk < K; kk++) v[(size_t)kk * D + d] = t.host[(size_t)d * K + kk];
            push(std::string(nm) + "#kd", std::move(v), {K, D});
        }
    }
    if (c.zip1 && !c.streaming) {  // offline v1: conv_module depthwise [D,1,K] -> [K][D] for the LDS-tiled GLU + depthwise kernel
        for (int si = 0; si < c.ns; si++)
            for (int li = 0; li < c.nlayer[si]; li++)
                for (int k = 1; k <= 2; k++) {
                    char nm[192];
                    snprintf(nm, sizeof nm, "encoder.encoders.%d.%slayers.%d.conv_module%d.depthwise_conv.weight", si, c.ds[si] > 1 ? "encoder." : "", li, k);
                    const Tensor& t = tensor(nm);
                    const int D = (int)t.dims[0], K = (int)t.dims[2];
                    K2_REQUIRE(D == c.dim[si] && K == c.kern[si], "%s has shape [%d,1,%d], config says [%d,1,%d]", nm, D, K, c.dim[si], c.kern[si]);
                    std::vector<float> v((size_t)K * D);
                    for (int d = 0; d < D; d++)
                        for (int kk = 0; kk < K; kk++) v[(size_t)kk * D + d] = t.host[(size_t)d * K + kk];
                    push(std::string(nm) + "#kd", std::move(v), {K, D});
                }
    }
    const bool has_encoder = !c.conformer && !c.zip1 && has("encoder_embed.conv.0.weight");
    if (has_encoder) {
    // conv filters [Co,Ci,3,3] -> [Co][kt][kf][ci]  (K index of the implicit GEMM over NHWC input)
    for (const char* nm : {"encoder_embed.conv.4.weight", "encoder_embed.conv.7.weight"}) {
        const Tensor& t = tensor(nm);
        int Co = (int)t.dims[0], Ci = (int)t.dims[1];
        std::vector<float> v((size_t)Co * 9 * Ci);
        for (int co = 0; co < Co; co++)
            for (int ci = 0; ci < Ci; ci++)
                for (int kt = 0; kt < 3; kt++)
                    for (int kf = 0; kf < 3; kf++)
                        v[((size_t)co * 9 + kt * 3 + kf) * Ci + ci] = t.host[(((size_t)co * Ci + ci) * 3 + kt) * 3 + kf];
        push(std::string(nm) + "#ohwi", std::move(v), {Co, 9 * Ci});
    }
    {   // depthwise 7x7 [C,1,7,7] -> [49][C]
        const Tensor& t = tensor("encoder_embed.convnext.depthwise_conv.weight");
        int C = (int)t.dims[0];
        std::vector<float> v((size_t)49 * C);
        for (int ch = 0; ch < C; ch++)
            for (int k = 0; k < 49; k++) v[(size_t)k * C + ch] = t.host[(size_t)ch * 49 + k];
        push("encoder_embed.convnext.depthwise_conv.weight#kc", std::move(v), {49, C});
    }
    {   // out Linear [D0, c*F3+f] -> [D0, f*128+c] (NHWC flatten order)
        const Tensor& t = tensor("encoder_embed.out.weight");
        int D0 = (int)t.dims[0], KK = (int)t.dims[1], C = 128, F3 = KK / C;
        std::vector<float> v((size_t)D0 * KK);
        for (int d = 0; d < D0; d++)
            for (int ch = 0; ch < C; ch++)
                for (int f = 0; f < F3; f++) v[(size_t)d * KK + f * C + ch] = t.host[(size_t)d * KK + ch * F3 + f];
        push("encoder_embed.out.weight#fc", std::move(v), {D0, KK});
    }
    // conv_module depthwise [D,1,K] -> [K][D]  (offline models; the streaming model's causal/chunkwise
    // filters are read in their native layout)
    for (int si = 0; si < (c.streaming ? 0 : c.ns); si++)
        for (int li = 0; li < c.nlayer[si]; li++)
            for (int k = 1; k <= 2; k++) {
                char nm[192];
                snprintf(nm, sizeof nm, "encoder.encoders.%d.layers.%d.conv_module%d.depthwise_conv.weight", si, li, k);
                const Tensor& t = tensor(nm);
                int D = (int)t.dims[0], K = (int)t.dims[2];
                K2_REQUIRE(D == c.dim[si] && K == c.kern[si], "%s has shape [%d,1,%d], config says [%d,1,%d]", nm, D, K,
                           c.dim[si], c.kern[si]);
                std::vector<float> v((size_t)K * D);
                for (int d = 0; d < D; d++)
                    for (int kk = 0; kk < K; kk++) v[(size_t)kk * D + d] = t.host[(size_t)d * K + kk];
                push(std::string(nm) + "#kd", std::move(v), {K, D});
            }
    // nonlin_attention in_proj [3 Hc, D] (s | x | y): tmp = x * tanh(s) in the GEMM's epilogue -- rows 32 q + p = x of channel
    // 16 q + p (the value), 32 q + 16 + p = its s (the gate), y rows unchanged behind them
    for (int si = 0; si < (c.streaming ? 0 : c.ns); si++)
        for (int li = 0; li < c.nlayer[si]; li++) {
            char nm[192], nb[192];
            snprintf(nm, sizeof nm, "encoder.encoders.%d.layers.%d.nonlin_attention.in_proj.weight", si, li);
            snprintf(nb, sizeof nb, "encoder.encoders.%d.layers.%d.nonlin_attention.in_proj.bias", si, li);
            const Tensor& tw = tensor(nm);
            const Tensor& tb = tensor(nb);
            const int D = c.dim[si], Hc = (int)tw.dims[0] / 3;
            K2_REQUIRE((int)tw.dims[0] == 3 * Hc && (int)tw.dims[1] == D && (int)tb.dims[0] == 3 * Hc, "%s: expected [3 Hc, %d]", nm, D);
            if (Hc % 16 != 0) continue;  // the engine then keeps the separate gate kernel
            std::vector<float> vw((size_t)3 * Hc * D), vb((size_t)3 * Hc);
            for (int ch = 0; ch < Hc; ch++) {
                const int rv = 32 * (ch / 16) + ch % 16, rg = rv + 16;
                memcpy(&vw[(size_t)rv * D], &tw.host[(size_t)(Hc + ch) * D], sizeof(float) * D);  // x
                memcpy(&vw[(size_t)rg * D], &tw.host[(size_t)ch * D], sizeof(float) * D);         // s
                vb[rv] = tb.host[Hc + ch];
                vb[rg] = tb.host[ch];
            }
            memcpy(&vw[(size_t)2 * Hc * D], &tw.host[(size_t)2 * Hc * D], sizeof(float) * (size_t)Hc * D);
            memcpy(&vb[(size_t)2 * Hc], &tb.host[(size_t)2 * Hc], sizeof(float) * Hc);
            push(std::string(nm) + "#glu", std::move(vw), {3 * Hc, D});
            push(std::string(nb) + "#glu", std::move(vb), {3 * Hc});
        }
    // conv_module in_proj [2D, D] (value rows | gate rows) -> rows interleaved in blocks of 16 channels: new row 32 q + p = value of
    // channel 16 q + p, row 32 q + 16 + p = its gate, so that one 32-column block of the GEMM output holds 16 channels' values and
    // gates and the GLU runs in the GEMM's epilogue (a lane pair 16 apart), halving what the conv kernel has to read back
    // (streaming Zipformer2 as well since round 5: there the same epilogue goes on to the chunk-causal depthwise conv, gemm_glu_causal_conv)
    for (int si = 0; si < c.ns; si++)
        for (int li = 0; li < c.nlayer[si]; li++)
            for (int k = 1; k <= 2; k++) {
                char nm[192], nb[192];
                snprintf(nm, sizeof nm, "encoder.encoders.%d.layers.%d.conv_module%d.in_proj.weight", si, li, k);
                snprintf(nb, sizeof nb, "encoder.encoders.%d.layers.%d.conv_module%d.in_proj.bias", si, li, k);
                const Tensor& tw = tensor(nm);
                const Tensor& tb = tensor(nb);
                const int D = c.dim[si];
                K2_REQUIRE((int)tw.dims[0] == 2 * D && (int)tw.dims[1] == D && (int)tb.dims[0] == 2 * D && D % 16 == 0,
                           "%s: expected [%d,%d] with D %% 16 == 0", nm, 2 * D, D);
                std::vector<float> vw((size_t)2 * D * D), vb((size_t)2 * D);
                for (int ch = 0; ch < D; ch++) {
                    const int rv = 32 * (ch / 16) + ch % 16, rg = rv + 16;
                    memcpy(&vw[(size_t)rv * D], &tw.host[(size_t)ch * D], sizeof(float) * D);
                    memcpy(&vw[(size_t)rg * D], &tw.host[(size_t)(D + ch) * D], sizeof(float) * D);
                    vb[rv] = tb.host[ch];
                    vb[rg] = tb.host[D + ch];
                }
                push(std::string(nm) + "#glu", std::move(vw), {2 * D, D});
                push(std::string(nb) + "#glu", std::move(vb), {2 * D});
            }
    // [feed_forward1.in_proj ; self_attn_weights.in_proj] stacked: both read the layer input, so the engine runs them as ONE GEMM
    // (SwooshL on the first F1 columns only) -- one launch and one pass over x instead of two
    for (int si = 0; si < c.ns; si++)
        for (int li = 0; li < c.nlayer[si]; li++) {
            char pf[160];
            snprintf(pf, sizeof pf, "encoder.encoders.%d.layers.%d.", si, li);
            const std::string P(pf);
            const Tensor& wf = tensor(P + "feed_forward1.in_proj.weight");
            const Tensor& wa = tensor(P + "self_attn_weights.in_proj.weight");
            const Tensor& bf = tensor(P + "feed_forward1.in_proj.bias");
            const Tensor& ba = tensor(P + "self_attn_weights.in_proj.bias");
            const int F1 = (int)wf.dims[0], NA = (int)wa.dims[0], D = (int)wf.dims[1];
            K2_REQUIRE(D == c.dim[si] && (int)wa.dims[1] == D, "%s: in_proj widths disagree", pf);
            std::vector<float> wv((size_t)(F1 + NA) * D), bv((size_t)F1 + NA);
            std::copy(wf.host, wf.host + (size_t)F1 * D, wv.begin());
            std::copy(wa.host, wa.host + (size_t)NA * D, wv.begin() + (size_t)F1 * D);
            std::copy(bf.host, bf.host + F1, bv.begin());
            std::copy(ba.host, ba.host + NA, bv.begin() + F1);
            push(P + "#ff1_attn_in.weight", std::move(wv), {F1 + NA, D});
            push(P + "#ff1_attn_in.bias", std::move(bv), {F1 + NA});
        }
    }  // has_encoder
    if (has("joiner.output_linear.weight")) {   // joiner.output_linear [V,J] -> k-major [J][Vp]
        const Tensor& t = tensor("joiner.output_linear.weight");
        K2_REQUIRE(t.dims[0] == c.V && t.dims[1] == c.J, "joiner.output_linear.weight is [%lld,%lld], config says [%d,%d]",
                   (long long)t.dims[0], (long long)t.dims[1], c.V, c.J);
        std::vector<float> v((size_t)c.J * c.Vp, 0.f);
        for (int n = 0; n < c.V; n++)
            for (int k = 0; k < c.J; k++) v[(size_t)k * c.Vp + n] = t.host[(size_t)n * c.J + k];
        push("joiner.output_linear.weight#kn", std::move(v), {c.J, c.Vp});
        // Large vocabularies: an f16 copy for the search's SCREENING pass (greedy.hip screen_round) and, per column, a rigorous bound
        // on how far the f16 product can be from the f32 logit the search kernels compute.  Layout: 16-column tiles x 32-deep K steps
        // in v_mfma_f32_16x16x32_f16's B-fragment order -- element j of lane l of (tile t, step s) is W[16 t + (l & 15)][32 s +
        // 8 (l >> 4) + j] -- so a wave's load instruction reads 1 KB in one piece.
        //   a_k = tanh(..) in [-1, 1]; a~, w~ = fp16(a), fp16(w):  |x~ - x| <= 2^-11 |x| + 2^-25  (round to nearest, subnormals)
        //   |sum a~ w~ - sum a w| <= c (2^-10 + 2^-20) + J 2^-24,  c = sum_k |w_k|
        //   either accumulation (f32, any order) adds at most (J + 8) 2^-24 c (1 + 2^-10), the bias add one ulp of the result
        const int screen_min_v = tunables().screen_min_v;
        if (screen_min_v > 0 && c.V >= screen_min_v && (c.J == 512 || c.J == 256 || c.J == 128 || c.J == 64) && has("joiner.output_linear.bias")) {
            const int nt = (int)(align_up(c.Vp, 16) / 16), ns = c.J / 32;
            std::vector<float> hv((size_t)nt * ns * 64 * 8 / 2, 0.f);   // two halfs per float slot
            uint16_t* h = reinterpret_cast<uint16_t*>(hv.data());
            std::vector<float> eps((size_t)nt * 16, 0.f);
            const float* bias = tensor("joiner.output_linear.bias").host;
            for (int n = 0; n < c.V; n++) {
                double cs = 0.0;
                bool finite = true;
                for (int k = 0; k < c.J; k++) {
                    const float wv = t.host[(size_t)n * c.J + k];
                    cs += std::fabs((double)wv);
                    finite = finite && std::isfinite(wv) && std::fabs(wv) < 65504.0f;
                    const int tt = n >> 4, s = k >> 5, l = (n & 15) + 16 * ((k & 31) >> 3), j = k & 7;
                    h[(((size_t)tt * ns + s) * 64 + l) * 8 + j] = f32_to_f16_rne(wv);
                }
                const double bound = cs * (std::ldexp(1.0, -10) + std::ldexp(1.0, -20) + 2.2 * (c.J + 8) * std::ldexp(1.0, -24)) +
                                     (std::fabs((double)bias[n]) + 1.0) * std::ldexp(1.0, -20) + c.J * std::ldexp(1.0, -24);
                // (a column with a NaN / Inf / f16-overflowing weight gets an infinite bound: it is always a candidate, the candidate
                // list overflows or the screen sees a non-finite value, and the round falls back to the full f32 sweep)
                eps[n] = (finite && std::isfinite(bound)) ? std::nextafter((float)(bound * 1.01), INFINITY) : INFINITY;
            }
            push("joiner.output_linear.weight#h16", std::move(hv), {nt, ns, 64, 4});
            push("joiner.output_linear.weight#eps", std::move(eps), {nt * 16});
        }
    }
    if (has("joiner.decoder_proj.weight")) {   // joiner.decoder_proj [J,DD] -> k-major [DD][J]
        const Tensor& t = tensor("joiner.decoder_proj.weight");
        K2_REQUIRE(t.dims[0] == c.J && t.dims[1] == c.DD, "joiner.decoder_proj.weight shape mismatch");
        std::vector<float> v((size_t)c.DD * c.J);
        for (int n = 0; n < c.J; n++)
            for (int k = 0; k < c.DD; k++) v[(size_t)k * c.J + n] = t.host[(size_t)n * c.DD + k];
        push("joiner.decoder_proj.weight#kn", std::move(v), {c.DD, c.J});
    }
    // fbank tables (kaldi feature-window.cc / mel-computations.cc semantics)
    const FbankOpts& f = c.fbank;
    {
        std::vector<float> win(f.frame_len);
        double a = 2.0 * M_PI / (f.frame_len - 1);
        for (int i = 0; i < f.frame_len; i++) {
            double wv;
            if (f.window_type == "hamming") wv = 0.54 - 0.46 * cos(a * i);
            else if (f.window_type == "hanning") wv = 0.5 - 0.5 * cos(a * i);
            else if (f.window_type == "povey") wv = pow(0.5 - 0.5 * cos(a * i), 0.85);
            else if (f.window_type == "rectangular") wv = 1.0;
            else failf(K2HIP_ERR_INVALID, "unknown window_type '%s'", f.window_type.c_str());
            win[i] = (float)wv;
        }
        push("#fbank.window", std::move(win), {f.frame_len});
        int nb = f.padded / 2;
        std::vector<float> mw((size_t)f.num_bins * nb, 0.f);
        // kaldi mel-computations.cc triangles, evaluated in f64: mel - left cancels ~4
        // digits in f32, which would make the table depend on contraction choices.
        double nyq = 0.5 * f.sample_rate, hi = f.high_freq;
        if (hi <= 0.0) hi += nyq;
        double bin_w = (double)f.sample_rate / f.padded;
        double mel_low = mel_scale(f.low_freq), mel_high = mel_scale(hi);
        double delta = (mel_high - mel_low) / (f.num_bins + 1);
        for (int b = 0; b < f.num_bins; b++) {
            double left = mel_low + b * delta, center = mel_low + (b + 1) * delta, right = mel_low + (b + 2) * delta;
            for (int i = 0; i < nb; i++) {
                double mel = mel_scale(bin_w * i);
                if (mel > left && mel < right)
                    mw[(size_t)b * nb + i] = (float)((mel <= center) ? (mel - left) / (center - left) : (right - mel) / (right - center));
            }
        }
        // extent [lo, hi) of each filter's non-zero weights (a triangle covers a few FFT bins): the kernel sums only those
        std::vector<float> mr((size_t)f.num_bins * 2, 0.f);
        for (int b = 0; b < f.num_bins; b++) {
            int lo = nb, hi2 = 0;
            for (int i = 0; i < nb; i++)
                if (mw[(size_t)b * nb + i] != 0.f) { lo = std::min(lo, i); hi2 = i + 1; }
            if (hi2 <= lo) lo = hi2 = 0;
            mr[2 * b] = (float)lo;
            mr[2 * b + 1] = (float)hi2;
        }
        push("#fbank.melw", std::move(mw), {f.num_bins, nb});
        push("#fbank.melrange", std::move(mr), {f.num_bins, 2});
    }
}

}  // namespace k2hip
