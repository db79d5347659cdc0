// Streaming (OnlineRecognizer) path of the engine: device-resident per-stream state pool and one
// batched chunk step.  Replaces, per tick of OnlineRecognizer.ForwardBatchGreedySearch
// (OnlineRecognizer.cs:85-219): stack_states (:131) + EncoderProj (:132) + the T'-frame greedy loop
// (:141-202) + unstack_states (:204).  States never leave the GPU.
#include <climits>
#include <cmath>

#include "engine.h"

namespace k2hip {

void Engine::online_ensure_pool() {
    if (online_pool_) return;
    const Config& c = model_->cfg();
    K2_REQUIRE(c.streaming || c.lstm, "this model is not a streaming export (metadata 'streaming' != 1)");
    OnlineLayout& L = lay_;
    long long off = 0;
    if (c.conformer) {  // OnlineProjOfConformer.GetEncoderInitStates (:55-82): cached_attn [L][left][D] then cached_conv [L][K-1][D]
        L.nl = c.nlayer[0];
        L.floats_per_stream = ((long long)c.nlayer[0] * (c.left[0] + c.kern[0] - 1) * c.dim[0] + 63) / 64 * 64;
        online_cap_ = 256;
        if (tunables().max_streams > 0) online_cap_ = tunables().max_streams;
        K2_HIP(hipSetDevice(device_));
        K2_HIP(hipMalloc(&online_pool_, sizeof(float) * (size_t)L.floats_per_stream * online_cap_));
        for (int i = online_cap_ - 1; i >= 0; i--) free_slots_.push_back(i);
        return;
    }
    if (c.lstm) {  // OnlineProjOfLstm.GetEncoderInitStates (:55-75): h [layers][d_model] then c [layers][rnn_hidden]
        L.nl = c.nlayer[0];
        L.floats_per_stream = ((long long)c.nlayer[0] * (c.dim[0] + c.rnn_hidden) + 63) / 64 * 64;
        online_cap_ = 256;
        if (tunables().max_streams > 0) online_cap_ = tunables().max_streams;
        K2_HIP(hipSetDevice(device_));
        K2_HIP(hipMalloc(&online_pool_, sizeof(float) * (size_t)L.floats_per_stream * online_cap_));
        for (int i = online_cap_ - 1; i >= 0; i--) free_slots_.push_back(i);
        return;
    }
    auto put = [&](std::vector<long long>& v, long long n) {
        v.push_back(off);
        off += (n + 3) / 4 * 4;  // keep every cache 16-byte aligned
    };
    if (c.zip1) {  // OnlineProjOfZipformer.GetEncoderInitStates (:56-111), batch 1, layer by layer
        for (int si = 0; si < c.ns; si++)
            for (int li = 0; li < c.nlayer[si]; li++) {
                const long long D = c.dim[si], A = c.att[si], left = c.left[si], K = c.kern[si];
                put(L.clen, 1);
                put(L.nonlin, D);  // cached_avg
                put(L.key, left * A);
                put(L.val1, left * (A / 2));
                put(L.val2, left * (A / 2));
                put(L.conv1, D * (K - 1));
                put(L.conv2, D * (K - 1));
                L.sizes.push_back({left * A, D, left * (A / 2), left * (A / 2), D * (K - 1), D * (K - 1)});
                L.nl++;
            }
        L.floats_per_stream = (off + 63) / 64 * 64;
        online_cap_ = 256;
        if (tunables().max_streams > 0) online_cap_ = tunables().max_streams;
        K2_HIP(hipSetDevice(device_));
        K2_HIP(hipMalloc(&online_pool_, sizeof(float) * (size_t)L.floats_per_stream * online_cap_));
        for (int i = online_cap_ - 1; i >= 0; i--) free_slots_.push_back(i);
        return;
    }
    // The four attention caches of a layer are RINGS of KL = left + Tc rows (kernels.h: RingRef): a chunk's rows overwrite the oldest
    // ones in place instead of the whole cache being rolled every chunk.  What GetEncoderInitStates / unstack_states define -- the
    // newest `left` rows in order -- is what online_read_state returns.
    const int tc50 = online_tc50();
    for (int si = 0; si < c.ns; si++) {
        const int D = c.dim[si], H = c.heads[si], left = c.left[si], kl = left + (tc50 + c.ds[si] - 1) / c.ds[si];
        for (int li = 0; li < c.nlayer[si]; li++) {
            put(L.key, (long long)kl * c.qhd[si] * H);
            put(L.nonlin, (long long)kl * (3 * D / 4));
            put(L.val1, (long long)kl * c.vhd[si] * H);
            put(L.val2, (long long)kl * c.vhd[si] * H);
            put(L.conv1, (long long)D * (c.kern[si] / 2));
            put(L.conv2, (long long)D * (c.kern[si] / 2));
            L.sizes.push_back({(long long)left * c.qhd[si] * H, (long long)left * (3 * D / 4), (long long)left * c.vhd[si] * H,
                               (long long)left * c.vhd[si] * H, (long long)D * (c.kern[si] / 2), (long long)D * (c.kern[si] / 2)});
            L.nl++;
        }
    }
    L.embed = off;
    off += 128 * 3 * 19;
    L.floats_per_stream = (off + 63) / 64 * 64;
    online_cap_ = 256;
    if (tunables().max_streams > 0) online_cap_ = tunables().max_streams;
    K2_HIP(hipSetDevice(device_));
    K2_HIP(hipMalloc(&online_pool_, sizeof(float) * (size_t)L.floats_per_stream * online_cap_));
    for (int i = online_cap_ - 1; i >= 0; i--) free_slots_.push_back(i);
}

int Engine::online_alloc_slot() {
    online_ensure_pool();
    if (!online_fifo_ && model_->cfg().feat % 4 == 0) {  // device mirror of the feature FIFOs (42 MB for 256 slots x 512 frames x 80)
        K2_HIP(hipSetDevice(device_));
        K2_HIP(hipMalloc(&online_fifo_, sizeof(float) * (size_t)online_cap_ * kFifoFrames * model_->cfg().feat));
    }
    if (free_slots_.empty())
        failf(K2HIP_ERR_CAPACITY, "all %d stream slots are in use (raise K2HIP_MAX_STREAMS before creating the model)", online_cap_);
    int slot = free_slots_.back();
    free_slots_.pop_back();
    // GetEncoderInitStates (OnlineProjOfZipformer2.cs:63-111): every cache starts at zero
    K2_HIP(hipSetDevice(device_));
    K2_HIP(hipMemsetAsync(online_pool_ + (size_t)slot * lay_.floats_per_stream, 0, sizeof(float) * (size_t)lay_.floats_per_stream, stream_));
    K2_HIP(hipStreamSynchronize(stream_));
    return slot;
}
// host frames into ring rows pos, pos + 1, ... (mod kFifoFrames) of a slot's device FIFO (OnlineStream fed with ready-made features)
void Engine::online_fifo_write(int slot, int pos, const float* feats, int64_t n_frames) {
    if (!online_fifo_ || n_frames <= 0) return;
    K2_REQUIRE(slot >= 0 && slot < online_cap_ && pos >= 0 && pos < kFifoFrames && n_frames <= kFifoFrames, "fifo write out of range");
    const int feat = model_->cfg().feat;
    K2_HIP(hipSetDevice(device_));
    const int64_t first = std::min<int64_t>(n_frames, kFifoFrames - pos);
    float* base = online_fifo_ + (size_t)slot * kFifoFrames * feat;
    K2_HIP(copy_blocking(base + (size_t)pos * feat, feats, sizeof(float) * (size_t)first * feat, hipMemcpyHostToDevice));
    if (n_frames > first) K2_HIP(copy_blocking(base, feats + (size_t)first * feat, sizeof(float) * (size_t)(n_frames - first) * feat, hipMemcpyHostToDevice));
}
void Engine::online_free_slot(int slot) {
    if (slot >= 0) free_slots_.push_back(slot);
}

// frames of one chunk after Conv2dSubsampling + ConvNeXt (50 Hz): T = 45 -> 16
int Engine::online_tc50() const {
    const int T = model_->cfg().chunk_T, T1 = T - 2, T2 = (T1 - 3) / 2 + 1, T3 = T2 - 2;
    return T3 - 3;
}

void Engine::online_read_state(int slot, int layer, int kind, long long chunks_done, float* out, int64_t cap, int64_t* n) {
    online_ensure_pool();
    K2_REQUIRE(slot >= 0 && slot < online_cap_, "bad slot %d", slot);
    long long off, cnt;
    if (model_->cfg().conformer) {  // kind 0: cached_attn of `layer` [left, D]; kind 1: cached_conv of `layer` [K-1, D]
        const Config& cf = model_->cfg();
        K2_REQUIRE(layer >= 0 && layer < cf.nlayer[0] && (kind == 0 || kind == 1), "bad conformer state index layer=%d kind=%d", layer, kind);
        const long long na = (long long)cf.left[0] * cf.dim[0], ncv = (long long)(cf.kern[0] - 1) * cf.dim[0];
        cnt = kind == 0 ? na : ncv;
        off = kind == 0 ? layer * na : cf.nlayer[0] * na + layer * ncv;
    } else if (model_->cfg().lstm) {  // kind 0: h of `layer` [d_model]; kind 1: c of `layer` [rnn_hidden]
        const Config& cf = model_->cfg();
        K2_REQUIRE(layer >= 0 && layer < cf.nlayer[0] && (kind == 0 || kind == 1), "bad lstm state index layer=%d kind=%d", layer, kind);
        cnt = kind == 0 ? cf.dim[0] : cf.rnn_hidden;
        off = kind == 0 ? (long long)layer * cf.dim[0] : (long long)cf.nlayer[0] * cf.dim[0] + (long long)layer * cf.rnn_hidden;
    } else if (model_->cfg().zip1 && kind == 7) {  // cached_len of `layer`
        K2_REQUIRE(layer >= 0 && layer < lay_.nl, "bad state index layer=%d", layer);
        off = lay_.clen[layer];
        cnt = 1;
    } else if (kind == 6) {
        K2_REQUIRE(!model_->cfg().zip1, "a zipformer (v1) stream has no embed state");
        off = lay_.embed;
        cnt = 128 * 3 * 19;
    } else {
        K2_REQUIRE(layer >= 0 && layer < lay_.nl && kind >= 0 && kind < 6, "bad state index layer=%d kind=%d", layer, kind);
        const std::vector<long long>* v[6] = {&lay_.key, &lay_.nonlin, &lay_.val1, &lay_.val2, &lay_.conv1, &lay_.conv2};
        off = (*v[kind])[layer];
        cnt = lay_.sizes[layer][kind];
    }
    *n = cnt;
    if (!out) return;
    if (cnt > cap) failf(K2HIP_ERR_CAPACITY, "state needs %lld floats", cnt);
    K2_HIP(hipSetDevice(device_));
    K2_HIP(hipStreamSynchronize(stream_));
    const Config& cfz = model_->cfg();
    if (!cfz.conformer && !cfz.lstm && !cfz.zip1 && kind >= 0 && kind < 4) {
        // attention caches are rings: the cache the reference would hold after `chunks_done` chunks is rows Tc .. KL-1 of the last
        // [cache ; chunk] concatenation, i.e. ring rows (head + 2 Tc + j) % KL, j = 0 .. left-1, head = ((chunks_done - 1) Tc) % KL
        int si = 0, acc = 0;
        while (si < cfz.ns && layer >= acc + cfz.nlayer[si]) acc += cfz.nlayer[si++];
        const int left = cfz.left[si], tc = (online_tc50() + cfz.ds[si] - 1) / cfz.ds[si], kl = left + tc;
        const long long width = cnt / left;
        std::vector<float> ring((size_t)kl * width);
        K2_HIP(copy_blocking(ring.data(), online_pool_ + (size_t)slot * lay_.floats_per_stream + off, sizeof(float) * ring.size(), hipMemcpyDeviceToHost));
        // before the first chunk the ring is all zeros and any rotation of it is the reference's zero cache
        const long long head = chunks_done > 0 ? ((chunks_done - 1) * tc) % kl : 0;
        for (int j = 0; j < left; j++) {
            const long long p = (head + 2LL * tc + j) % kl;
            memcpy(out + (size_t)j * width, ring.data() + (size_t)p * width, sizeof(float) * (size_t)width);
        }
        return;
    }
    K2_HIP(copy_blocking(out, online_pool_ + (size_t)slot * lay_.floats_per_stream + off, sizeof(float) * (size_t)cnt, hipMemcpyDeviceToHost));
}

int Engine::online_frames_per_chunk() const {
    const Config& c = model_->cfg();
    if (c.lstm) return lstm_out_frames(c.chunk_T);
    if (c.conformer) return conformer_out_frames(c.chunk_T) - 2 - c.right;
    if (c.zip1) return ((c.chunk_T - 7) / 2 + 1) / 2;
    return (c.shift / 2 + 1) / 2;
}

// CompactRelPositionalEncoding.forward(x, left_context_len): row n <-> relative position n - (Tc+L-1)
const float* Engine::pos_emb_stream(int Tc, int L) {
    std::lock_guard<std::mutex> lk(cache_mu_);
    const int key = -(Tc * 100000 + L);  // negative keys: streaming tables
    auto it = pe_cache_.find(key);
    if (it != pe_cache_.end()) return it->second;
    const int pd = model_->cfg().pos_dim, n2 = 2 * Tc - 1 + L;
    std::vector<float> pe((size_t)n2 * pd);
    const float cl = sqrtf((float)pd), ls = (float)pd / (2.0f * (float)M_PI), logcl = logf(cl);
    for (int n = 0; n < n2; n++) {
        float x = (float)(n - (Tc + L - 1));
        float sg = (float)((x > 0.f) - (x < 0.f));
        float xa = atanf(cl * sg * (logf(fabsf(x) + cl) - logcl) / ls);
        for (int k = 0; k < pd / 2; k++) {
            pe[(size_t)n * pd + 2 * k] = cosf(xa * (float)(k + 1));
            pe[(size_t)n * pd + 2 * k + 1] = sinf(xa * (float)(k + 1));
        }
        pe[(size_t)n * pd + pd - 1] = 1.0f;
    }
    float* d = nullptr;
    K2_HIP(hipMalloc(&d, pe.size() * sizeof(float)));
    K2_HIP(copy_blocking(d, pe.data(), pe.size() * sizeof(float), hipMemcpyHostToDevice));
    pe_cache_[key] = d;
    return d;
}

// Conv2dSubsampling.streaming_forward, NHWC.  x: [B,T,80] -> [B*Tc, D0]
float* Engine::encoder_embed_stream(const Ctx& c, const float* x, const int* d_slots, int B, int T, int* Tc_out) {
    const Model& m = *model_;
    const int F0 = 80, T1 = T - 2, T2 = (T1 - 3) / 2 + 1, F2 = (F0 - 3) / 2 + 1, T3 = T2 - 2, F3 = (F2 - 3) / 2 + 1, Tc = T3 - 3;
    K2_REQUIRE(Tc > 0 && F3 == 19, "streaming embed: chunk of %d frames unsupported", T);
    Arena& ar = *c.arena;
    const int D0 = m.cfg().dim[0];
    float* out = ar.take<float>((int64_t)B * Tc * D0);
    int64_t mark = ar.mark();
    float* a1 = ar.take<float>((int64_t)B * T1 * F0 * 8);
    conv0_swoosh(c, x, m.w("encoder_embed.conv.0.weight"), m.w("encoder_embed.conv.0.bias"), a1, B, T, F0);
    float* a2 = ar.take<float>((int64_t)B * T2 * F2 * 32);
    {
        GemmArgs g;
        g.A = a1; g.W = m.w("encoder_embed.conv.4.weight#ohwi"); g.ldw = 72; g.bias = m.w("encoder_embed.conv.4.bias");
        g.C = a2; g.ldc = 32; g.M = B * T2 * F2; g.N = 32; g.K = 72; g.act = ACT_SWOOSH_R;
        g.cv_Fout = F2; g.cv_Tout = T2; g.cv_Tin = T1; g.cv_Fin = F0; g.cv_C = 8; g.cv_st = 2; g.cv_sf = 2;
        g.seg_len = 24; g.seg_stride = F0 * 8;
        gemm(c, g);
    }
    float* a3 = ar.take<float>((int64_t)B * T3 * F3 * 128);
    {
        GemmArgs g;
        g.A = a2; g.W = m.w("encoder_embed.conv.7.weight#ohwi"); g.ldw = 288; g.bias = m.w("encoder_embed.conv.7.bias");
        g.C = a3; g.ldc = 128; g.M = B * T3 * F3; g.N = 128; g.K = 288; g.act = ACT_SWOOSH_R;
        g.cv_Fout = F3; g.cv_Tout = T3; g.cv_Tin = T2; g.cv_Fin = F2; g.cv_C = 32; g.cv_st = 1; g.cv_sf = 2;
        g.seg_len = 96; g.seg_stride = F2 * 32;
        gemm(c, g);
    }
    // ConvNeXt.streaming_forward: [cache(3) ; a3(T3)] -> valid 7-tap time conv -> Tc frames
    float* cat = ar.take<float>((int64_t)B * (T3 + 3) * F3 * 128);
    const int npix = B * Tc * F3;
    float* byp = ar.take<float>((int64_t)npix * 128);  // bypass = x[:, :, :Tc]
    convnext_cat(c, a3, online_pool_, lay_.floats_per_stream, lay_.embed, d_slots, cat, byp, B, T3, Tc, F3, 128);
    float* dw = ar.take<float>((int64_t)npix * 128);
    dwconv7x7(c, cat, m.w("encoder_embed.convnext.depthwise_conv.weight#kc"), m.w("encoder_embed.convnext.depthwise_conv.bias"), dw, B,
              T3 + 3, Tc, 0, F3, 128);
    float* hid = ar.take<float>((int64_t)npix * 384);
    linear(c, dw, 128, m.w("encoder_embed.convnext.pointwise_conv1.weight"), m.w("encoder_embed.convnext.pointwise_conv1.bias"), hid, 384,
           npix, 128, 384, ACT_SWOOSH_L);
    linear(c, hid, 384, m.w("encoder_embed.convnext.pointwise_conv2.weight"), m.w("encoder_embed.convnext.pointwise_conv2.bias"), byp, 128,
           npix, 384, 128, ACT_NONE, byp, 128);
    float* lin = ar.take<float>((int64_t)B * Tc * D0);
    linear(c, byp, F3 * 128, m.w("encoder_embed.out.weight#fc"), m.w("encoder_embed.out.bias"), lin, D0, B * Tc, F3 * 128, D0);
    biasnorm(c, lin, m.w("encoder_embed.out_norm.bias"), m.w("encoder_embed.out_norm.log_scale"), out, B * Tc, D0);
    ar.rewind(mark);
    *Tc_out = Tc;
    return out;
}

// Zipformer2EncoderLayer.streaming_forward, in place on x [B*Tc, D]; l = global layer index
void Engine::encoder_layer_stream(const Ctx& c, int si, int li, int l, float* x, const float* pe, const int* d_slots,
                                  const long long* d_plen, const int* d_chunks, int B, int Tc, int L, const LayerTail* tail) {
    const Model& m = *model_;
    const Config& cf = m.cfg();
    const int D = cf.dim[si], F = cf.ff[si], H = cf.heads[si], vh = cf.vhd[si], K = cf.kern[si], qh = cf.qhd[si], ph = cf.phd[si];
    const int M = B * Tc, KL = L + Tc, KLp = (KL + 3) & ~3, inproj = (2 * qh + ph) * H, Hc = 3 * D / 4, HV = H * vh;
    const int n2 = 2 * Tc - 1 + L, left50 = cf.left[0] * cf.ds[0];
    char p[96];
    snprintf(p, sizeof p, "encoder.encoders.%d.layers.%d.", si, li);
    auto w = [&](const char* suffix) { return m.w(std::string(p) + suffix); };
    Arena& ar = *c.arena;
    int64_t mark = ar.mark();
    const long long SS = lay_.floats_per_stream;

    // [ff1.in_proj | attention-weights in_proj] of the layer input in one GEMM, as in the offline layer (model.cpp stacks the
    // two weight matrices): one launch fewer per layer
    const int F1 = F * 3 / 4, ldcat = F1 + inproj;
    float* cat = ar.take<float>((int64_t)M * ldcat);
    {
        GemmArgs g;
        g.A = x; g.lda = D; g.W = w("#ff1_attn_in.weight"); g.ldw = D; g.bias = w("#ff1_attn_in.bias");
        g.C = cat; g.ldc = ldcat; g.M = M; g.N = ldcat; g.K = D; g.act = ACT_SWOOSH_L; g.act_cols = F1;
        gemm(c, g);
    }
    const float* qkp = cat + F1;
    auto ring = [&](long long off) {
        RingRef r;
        r.pool = online_pool_; r.slot_stride = SS; r.off = off; r.slots = d_slots; r.chunks = d_chunks;
        return r;
    };
    const float* pp = pos_proj_cached(c, 1000 + l, pe, cf.pos_dim, w("self_attn_weights.linear_pos.weight"), n2, ph * H);
    float* aw = ar.take<float>((int64_t)H * B * Tc * KLp);  // columns in ring order
    attn_stream_ring(c, qkp, ldcat, ring(lay_.key[l]), pp, d_plen, aw, B, Tc, L, KLp, H, cf.ds[si], left50);

    float* src = ar.take<float>((int64_t)M * D);
    // the self-attention modules with their value projection inside (attn_proj_av_out_ring) read one buffer and write the other
    const bool vproj_fused = !tunables().no_fused_vproj && Tc >= tunables().fused_vproj_min_t && D % 32 == 0 && vh <= 16 && HV % 4 == 0 && HV <= 128;
    float* src_alt = vproj_fused ? ar.take<float>((int64_t)M * D) : nullptr;
    float* hid = ar.take<float>((int64_t)M * std::max({F * 5 / 4, 3 * Hc, 2 * D}));
    float* tmp2 = ar.take<float>((int64_t)M * std::max(D, Hc));

    auto feed_forward = [&](int k, int Fk, const float* in, float* out) {
        char a[48], b[48], cc[48], d[48];
        snprintf(a, sizeof a, "feed_forward%d.in_proj.weight", k);
        snprintf(b, sizeof b, "feed_forward%d.in_proj.bias", k);
        snprintf(cc, sizeof cc, "feed_forward%d.out_proj.weight", k);
        snprintf(d, sizeof d, "feed_forward%d.out_proj.bias", k);
        linear(c, in, D, w(a), w(b), hid, Fk, M, D, Fk, ACT_SWOOSH_L);
        linear(c, hid, Fk, w(cc), w(d), out, D, M, Fk, D, ACT_NONE, in, D);
    };
    auto self_attn = [&](int k, long long cache_off) {
        char a[48], b[48], cc[48], d[48];
        snprintf(a, sizeof a, "self_attn%d.in_proj.weight", k);
        snprintf(b, sizeof b, "self_attn%d.in_proj.bias", k);
        snprintf(cc, sizeof cc, "self_attn%d.out_proj.weight", k);
        snprintf(d, sizeof d, "self_attn%d.out_proj.bias", k);
        if (vproj_fused) {   // one launch: value projection of the chunk's rows, ring update, attention apply, out_proj, residual
            attn_proj_av_out_ring(c, aw, ring(cache_off), src, w(a), w(b), w(cc), w(d), src_alt, B, Tc, KL, KLp, H, vh, D);
            std::swap(src, src_alt);
            return;
        }
        linear(c, src, D, w(a), w(b), hid, HV, M, D, HV);
        // fused: chunk rows into the value ring, attention apply over the ring, out_proj, residual
        attn_av_out_ring(c, aw, ring(cache_off), hid, w(cc), w(d), src, B, Tc, KL, KLp, H, vh, D);
    };
    auto conv_module = [&](int k, long long cache_off) {
        char a[80], b[80], e[80], f[80], n1[96], n2_[96], n3[96], n4[96], n5[96];
        snprintf(a, sizeof a, "conv_module%d.in_proj.weight", k);
        snprintf(b, sizeof b, "conv_module%d.in_proj.bias", k);
        snprintf(e, sizeof e, "conv_module%d.out_proj.weight", k);
        snprintf(f, sizeof f, "conv_module%d.out_proj.bias", k);
        snprintf(n1, sizeof n1, "conv_module%d.depthwise_conv.causal_conv.weight", k);
        snprintf(n2_, sizeof n2_, "conv_module%d.depthwise_conv.causal_conv.bias", k);
        snprintf(n3, sizeof n3, "conv_module%d.depthwise_conv.chunkwise_conv.weight", k);
        snprintf(n4, sizeof n4, "conv_module%d.depthwise_conv.chunkwise_conv.bias", k);
        snprintf(n5, sizeof n5, "conv_module%d.depthwise_conv.chunkwise_conv_scale", k);
        // in_proj + GLU + chunk-causal depthwise conv + SwooshR in ONE launch where the shape has the fused form (round 5): the Tc rows
        // of a stream and a 16-channel (value | gate) block sit in one tile of the in_proj GEMM, so its epilogue has everything the
        // convolution needs (conv_module 3 -> 2 launches, 32 per tick; K2HIP_NO_FUSED_CONV keeps the two launches for the cross-check)
        bool fused = false;
        if (!tunables().no_fused_conv) {
            const std::string wa = std::string(a) + "#glu", wb = std::string(b) + "#glu";
            fused = gemm_glu_causal_conv(c, src, w(wa.c_str()), w(wb.c_str()), online_pool_, SS, cache_off, d_slots, w(n1), w(n2_), w(n3), w(n4), w(n5),
                                         tmp2, B, Tc, D, K);
        }
        if (!fused) {
            linear(c, src, D, w(a), w(b), hid, 2 * D, M, D, 2 * D);
            glu_causal_conv(c, hid, online_pool_, SS, cache_off, d_slots, w(n1), w(n2_), w(n3), w(n4), w(n5), tmp2, B, Tc, D, K);
        }
        linear(c, tmp2, D, w(e), w(f), src, D, M, D, D, ACT_NONE, src, D);
    };

    // src = x + ff1(x): the hidden activations are the first F1 columns of `cat`
    linear(c, cat, ldcat, w("feed_forward1.out_proj.weight"), w("feed_forward1.out_proj.bias"), src, D, M, F1, D, ACT_NONE, x, D);
    {   // NonlinAttention.streaming_forward
        linear(c, src, D, w("nonlin_attention.in_proj.weight"), w("nonlin_attention.in_proj.bias"), hid, 3 * Hc, M, D, 3 * Hc);
        // x * tanh(s) into the ring rows of this chunk and ctx = (aw_head0 . ring) * y in one per-stream launch (k_ring_put + a batched
        // GEMM before), then out_proj over all rows
        nonlin_av_out_ring(c, aw, ring(lay_.nonlin[l]), hid, 3 * Hc, nullptr, nullptr, tmp2, B, Tc, KL, KLp, Hc, D);
        linear(c, tmp2, Hc, w("nonlin_attention.out_proj.weight"), w("nonlin_attention.out_proj.bias"), src, D, M, Hc, D, ACT_NONE, src, D);
    }
    self_attn(1, lay_.val1[l]);
    conv_module(1, lay_.conv1[l]);
    {   // src = bypass_mid(x, src + ff2(src)): the bypass mix runs in the out_proj GEMM's epilogue
        linear(c, src, D, w("feed_forward2.in_proj.weight"), w("feed_forward2.in_proj.bias"), hid, F, M, D, F, ACT_SWOOSH_L);
        GemmArgs g;
        g.A = hid; g.lda = F; g.W = w("feed_forward2.out_proj.weight"); g.ldw = F; g.bias = w("feed_forward2.out_proj.bias");
        g.C = src; g.ldc = D; g.M = M; g.N = D; g.K = F; g.res = src; g.ldr = D;
        g.byp_orig = x; g.ld_orig = D; g.byp_scale = w("bypass_mid.bypass_scale");
        gemm(c, g);
    }
    self_attn(2, lay_.val2[l]);
    conv_module(2, lay_.conv2[l]);
    feed_forward(3, F * 5 / 4, src, src);
    if (tail && tail->bias2)   // the stack's last layer in front of a downsampled stack: that stack's downsample in the same launch
        biasnorm_bypass_downsample(c, src, x, w("norm.bias"), w("norm.log_scale"), w("bypass.bypass_scale"), x, tail->bias2, tail->xd2, B, Tc, D,
                                   tail->ds2, tail->D2);
    else
        biasnorm_bypass(c, src, x, w("norm.bias"), w("norm.log_scale"), w("bypass.bypass_scale"), x, M, D);
    ar.rewind(mark);
}

// The streaming Zipformer2 encoder of one (sub-)batch: x [B, T, feat] (log-floored) -> encoder_out [B*Tp, enc_dim]; caches of
// the B slots advanced in place (OnlineProjOfZipformer2.EncoderProj :491-618 without stack / unstack)
float* Engine::online_encoder_zip2(const Ctx& c, const float* d_x, const int* d_slots, const long long* d_plen, const int* d_chunks, int B) {
    const Model& m = *model_;
    const Config& cf = m.cfg();
    Arena& ar = *c.arena;
    const int T = cf.chunk_T, Tp = online_frames_per_chunk();
    int Tc = 0;
    float* x = encoder_embed_stream(c, d_x, d_slots, B, T, &Tc);
    const int M = B * Tc;
    float* outputs[kMaxStacks] = {nullptr};
    int Dcur = cf.dim[0], l = 0;
    float *pre_y = nullptr, *pre_xd = nullptr;
    FullDimSegs lz;   // (only its lz_* fields are used)
    for (int si = 0; si < cf.ns; si++) {
        const int D = cf.dim[si], ds = cf.ds[si], L = cf.left[si];
        // the stack's input is the previous output zero-extended / truncated to D channels (convert_channels): a stack that runs at
        // the input rate works in place on a converted copy (or on x itself when the width does not change); a downsampled stack
        // never materialises it -- its downsample and its out_combiner read x at its own width
        const int Din = Dcur;
        Dcur = D;
        if (ds == 1) {
            float* xi = x;
            if (D != Din) {
                xi = ar.take<float>((int64_t)M * D);
                convert_channels(c, x, xi, M, Din, D);
            }
            const float* pe = c.dry ? nullptr : pos_emb_stream(Tc, L);
            // in front of a downsampled stack the last layer's BiasNorm launch forms that stack's input as well (LayerTail)
            LayerTail tail;
            if (si + 1 < cf.ns && cf.ds[si + 1] > 1 && cf.nlayer[si] > 0) {
                tail.D2 = cf.dim[si + 1]; tail.ds2 = cf.ds[si + 1];
                pre_y = ar.take<float>((int64_t)M * tail.D2);
                pre_xd = ar.take<float>((int64_t)B * ((Tc + tail.ds2 - 1) / tail.ds2) * tail.D2);
                tail.xd2 = pre_xd;
                tail.bias2 = m.wf("encoder.encoders.%d.downsample.bias", si + 1);
            }
            for (int li = 0; li < cf.nlayer[si]; li++, l++)
                encoder_layer_stream(c, si, li, l, xi, pe, d_slots, d_plen, d_chunks, B, Tc, L, li == cf.nlayer[si] - 1 ? &tail : nullptr);
            x = xi;
        } else {
            const int Td = (Tc + ds - 1) / ds;
            // (as in the offline stacks: this stack's out_combiner + the next stack's downsample in one launch where both are downsampled)
            float* y = pre_y ? pre_y : ar.take<float>((int64_t)M * D);
            float* xd_ready = pre_xd;
            pre_y = pre_xd = nullptr;
            const bool fuse_next = si + 1 < cf.ns && cf.ds[si + 1] > 1;
            const int D2 = fuse_next ? cf.dim[si + 1] : 0, ds2 = fuse_next ? cf.ds[si + 1] : 1;
            if (fuse_next) {
                pre_y = ar.take<float>((int64_t)M * D2);
                pre_xd = ar.take<float>((int64_t)B * ((Tc + ds2 - 1) / ds2) * D2);
            }
            int64_t mark = ar.mark();
            float* xd = xd_ready ? xd_ready : ar.take<float>((int64_t)B * Td * D);
            if (!xd_ready) downsample(c, x, m.wf("encoder.encoders.%d.downsample.bias", si), xd, B, Tc, D, ds, Din);
            const float* pe = c.dry ? nullptr : pos_emb_stream(Td, L);
            for (int li = 0; li < cf.nlayer[si]; li++, l++) encoder_layer_stream(c, si, li, l, xd, pe, d_slots, d_plen, d_chunks, B, Td, L);
            const bool lazy = si == cf.ns - 1;   // the last stack's out_combiner runs inside the final downsample (FullDimSegs::lz_*)
            if (fuse_next)
                upsample_combine_downsample(c, x, xd, m.wf("encoder.encoders.%d.out_combiner.bypass_scale", si), y,
                                            m.wf("encoder.encoders.%d.downsample.bias", si + 1), pre_xd, B, Tc, Td, D, ds, Din, D2, ds2);
            else if (lazy) {
                lz.lz_orig = x; lz.lz_xd = xd; lz.lz_scale = m.wf("encoder.encoders.%d.out_combiner.bypass_scale", si);
                lz.lz_Td = Td; lz.lz_ds = ds; lz.lz_Do = Din;
            } else
                upsample_combine(c, x, xd, m.wf("encoder.encoders.%d.out_combiner.bypass_scale", si), y, B, Tc, Td, D, ds, Din);
            if (!lazy) ar.rewind(mark);   // (lazy: xd is read by the final downsample -- it stays allocated)
            x = y;
        }
        outputs[si] = x;
    }
    const int Dmax = cf.dmax;
    const int Tpp = (Tc + 1) / 2;
    K2_REQUIRE(Tpp == Tp, "internal: chunk yields %d frames, expected %d", Tpp, Tp);
    float* dsd = ar.take<float>((int64_t)B * Tp * Dmax);
    {   // _get_full_dim_output + downsample_output in one launch (no concatenated tensor)
        FullDimSegs segs;
        int cur = cf.dim[cf.ns - 1];
        segs.src[0] = outputs[cf.ns - 1]; segs.ld[0] = cur; segs.col1[0] = cur; segs.n = 1;
        for (int i = cf.ns - 2; i >= 0; i--) {
            const int d = cf.dim[i];
            if (d > cur) {
                K2_REQUIRE(segs.n < 8, "too many stack widths");
                segs.src[segs.n] = outputs[i]; segs.ld[segs.n] = d; segs.col1[segs.n] = d; segs.n++;
                cur = d;
            }
        }
        segs.lz_orig = c.dry ? nullptr : lz.lz_orig; segs.lz_xd = lz.lz_xd; segs.lz_scale = lz.lz_scale;
        segs.lz_Td = lz.lz_Td; segs.lz_ds = lz.lz_ds; segs.lz_Do = lz.lz_Do;
        downsample_full(c, segs, m.w("encoder.downsample_output.bias"), dsd, B, Tc, Dmax, 2);
    }
    float* enc = ar.take<float>((int64_t)B * Tp * cf.enc_dim());
    if (cf.ctc) {
        linear(c, dsd, Dmax, m.w("ctc_output.1.weight"), m.w("ctc_output.1.bias"), enc, cf.V, B * Tp, Dmax, cf.V);
        log_softmax_rows(c, enc, B * Tp, cf.V);
    } else {
        linear(c, dsd, Dmax, m.w("joiner.encoder_proj.weight"), m.w("joiner.encoder_proj.bias"), enc, cf.J, B * Tp, Dmax, cf.J);
    }
    return enc;
}

void Engine::online_step(const int* slots, const float* const* chunks, const long long* hyps, const long long* plens, const int* nchunks, int B,
                         int64_t* tokens, int32_t* ts, int32_t* n_tokens, const int* fifo_heads) {
    online_ensure_pool();
    K2_REQUIRE(B > 0, "online_step: no ready stream");
    const Model& m = *model_;
    const Config& cf = m.cfg();
    const int T = cf.chunk_T, Tp = online_frames_per_chunk();
    long long* d_tok = nullptr;
    int *d_ts = nullptr, *d_n = nullptr, *d_ovf = nullptr;
    // the streams' chunks, gathered once into pinned staging (one host copy; the H2D below is then a real asynchronous DMA)
    const size_t chunk_floats = (size_t)T * cf.feat;
    K2_HIP(hipSetDevice(device_));
    const bool from_fifo = fifo_heads != nullptr && online_fifo_ != nullptr;  // the chunks are already on the device (FIFO mirror)
    // ONE upload per tick: [chunks' frames (only when they are not on the device yet) | plens | hyps | slots | chunk counts | FIFO heads |
    // overflow flag = 0], packed in pinned staging in the device block's layout (five small copies from pageable memory + a memset were
    // six blit launches of ~4 us each at the head of the step); ONE download: [tokens | timestamps | counts | overflow flag]
    const int64_t nb_x = from_fifo ? 0 : (int64_t)sizeof(float) * chunk_floats * B;
    const int64_t o_plen = align_up(nb_x, 16), o_hyp = o_plen + 8 * (int64_t)B, o_slots = o_hyp + 16 * (int64_t)B, o_chunks = o_slots + 4 * (int64_t)B,
                  o_heads = o_chunks + 4 * (int64_t)B, o_ovf = align_up(o_heads + 4 * (int64_t)B, 16), in_bytes = o_ovf + 16;
    char* stage = static_cast<char*>(pinned_in(in_bytes));
    if (!from_fifo)
        for (int b = 0; b < B; b++) memcpy(stage + (size_t)b * chunk_floats * sizeof(float), chunks[b], sizeof(float) * chunk_floats);
    memcpy(stage + o_plen, plens, sizeof(long long) * B);
    memcpy(stage + o_hyp, hyps, sizeof(long long) * 2 * B);
    memcpy(stage + o_slots, slots, sizeof(int) * B);
    memcpy(stage + o_chunks, nchunks, sizeof(int) * B);
    if (from_fifo) memcpy(stage + o_heads, fifo_heads, sizeof(int) * B);
    else memset(stage + o_heads, 0, sizeof(int) * B);
    memset(stage + o_ovf, 0, 16);
    const int64_t nb_tok = (int64_t)B * Tp * 8, nb_ts = (int64_t)B * Tp * 4, nb_n = (int64_t)B * 4;
    run_sized([&](const Ctx& c) {
        Arena& ar = *c.arena;
        // one device block: the inputs, the overflow flag (uploaded as zero: the last 16 bytes of the input part), the outputs right
        // behind it -- [flag | tokens | timestamps | counts] is the layout finish_tokens downloads in one copy
        char* d_in = ar.take<char>(in_bytes + nb_tok + nb_ts + nb_n);
        char* d_out = d_in + in_bytes;
        d_tok = reinterpret_cast<long long*>(d_out);
        d_ts = reinterpret_cast<int*>(d_out + nb_tok);
        d_n = reinterpret_cast<int*>(d_out + nb_tok + nb_ts);
        d_ovf = reinterpret_cast<int*>(d_in + o_ovf);
        float* d_x = from_fifo ? ar.take<float>((int64_t)B * T * cf.feat) : reinterpret_cast<float*>(d_in);
        long long* d_plen = reinterpret_cast<long long*>(d_in + o_plen);
        long long* d_hyp = reinterpret_cast<long long*>(d_in + o_hyp);
        int* d_slots = reinterpret_cast<int*>(d_in + o_slots);
        int* d_chunks = reinterpret_cast<int*>(d_in + o_chunks);
        int* d_heads = reinterpret_cast<int*>(d_in + o_heads);
        if (!c.dry) {
            K2_HIP(hipEventRecord(ev_[0], c.stream));
            K2_HIP(hipMemcpyAsync(d_in, stage, (size_t)in_bytes, hipMemcpyHostToDevice, c.stream));
        }
        // the tick's search: rounds of joiner GEMMs + a per-stream step kernel, or the persistent kernel where its rounds go through
        // the f16 screen (measured per model: greedy_loop_screens)
        const bool persistent_search = tunables().search_rounds == 0 || (tunables().search_rounds < 0 && !cf.ctc && greedy_loop_screens(decjoin(), B, true, c.one_part));
        const bool ev_ok = !c.dry;
        // online PadSequence (PadHelper.cs:9-13,58): the floor of genuine zeros runs inside the gather when the chunks come from the FIFO
        if (from_fifo) fifo_gather(c, online_fifo_, kFifoFrames, cf.feat, d_slots, d_heads, d_x, B, T);
        else logfloor_inplace(c, d_x, (long long)B * T * cf.feat);
        if (cf.lstm || cf.conformer || cf.zip1) {
            int tc = Tp;
            float* enc = cf.lstm ? lstm_chunk(c, d_x, d_slots, B) : cf.zip1 ? zip1_chunk(c, d_x, d_slots, B, &tc) : conformer_chunk(c, d_x, d_slots, d_plen, B, &tc);
            K2_REQUIRE(tc == Tp, "internal: chunk yields %d frames, expected %d", tc, Tp);
            if (ev_ok) K2_HIP(hipEventRecord(ev_[3], c.stream));
            GreedyArgs a;
            a.enc = enc; a.B = B; a.Tp = Tp; a.t0 = nullptr; a.skip1 = 1; a.max_sym = INT_MAX;
            a.tokens = d_tok; a.timestamps = d_ts; a.n_tokens = d_n; a.max_tokens = Tp; a.overflow = d_ovf; a.init_ctx = d_hyp;
            if (persistent_search) greedy_loop(c, decjoin(), a);
            else greedy_rounds(c, decjoin(), model_->w("joiner.output_linear.weight"), a);
            if (ev_ok) K2_HIP(hipEventRecord(ev_[4], c.stream));
            return;
        }
        float* enc = online_encoder_zip2(c, d_x, d_slots, d_plen, d_chunks, B);
        if (ev_ok) K2_HIP(hipEventRecord(ev_[3], c.stream));
        if (cf.ctc) {
            // OnlineRecognizer.ForwardBatchGreedySearchCTC (:220-313): per-chunk CTC collapse, prev_id reset per chunk
            ctc_device(c, enc, B, Tp, d_tok, d_ts, d_n, Tp, d_ovf);
            if (ev_ok) K2_HIP(hipEventRecord(ev_[4], c.stream));
            return;
        }
        // OnlineRecognizer.cs:135-202: decoder on the streams' hyps, T' joiner steps, skip {blank, unk, 1}
        GreedyArgs a;
        a.enc = enc; a.B = B; a.Tp = Tp; a.t0 = nullptr; a.skip1 = 1; a.max_sym = INT_MAX;
        a.tokens = d_tok; a.timestamps = d_ts; a.n_tokens = d_n; a.max_tokens = Tp; a.overflow = d_ovf; a.init_ctx = d_hyp;
        if (persistent_search) greedy_loop(c, decjoin(), a);
        else greedy_rounds(c, decjoin(), model_->w("joiner.output_linear.weight"), a);
        if (ev_ok) K2_HIP(hipEventRecord(ev_[4], c.stream));
    });
    finish_tokens(d_tok, d_ts, d_n, d_ovf, B, Tp, tokens, ts, n_tokens);
    auto el = [&](int a, int b) { float ms = 0; (void)hipEventElapsedTime(&ms, ev_[a], ev_[b]); return ms; };
    timing_.fbank_ms = 0;
    timing_.pad_ms = 0;
    timing_.encoder_ms = el(0, 3);
    timing_.greedy_ms = el(3, 4);
    timing_.d2h_ms = el(4, 5);
    timing_.total_ms = el(0, 5);
}

// IOnlineProj.EncoderProj (OnlineProjOfZipformer2.cs:491-618) as an operator: feats [B, chunk_T, feat] host, the B slots' caches
// advance in place, encoder_out [B, T', joiner_dim] to host.  The search is the caller's (DecoderProj / JoinerProj).
void Engine::online_encoder(const int* slots, const float* feats, const long long* plens, const int* nchunks, int B, float* enc_out) {
    online_ensure_pool();
    K2_REQUIRE(B > 0, "online_encoder: empty batch");
    const Config& cf = model_->cfg();
    K2_REQUIRE(cf.streaming && !cf.lstm && !cf.conformer && !cf.zip1 && !cf.ctc, "online_encoder: streaming Zipformer2 transducer models only");
    const int T = cf.chunk_T, Tp = online_frames_per_chunk();
    K2_HIP(hipSetDevice(device_));
    const size_t in_bytes = sizeof(float) * (size_t)B * T * cf.feat;
    float* stage = static_cast<float*>(pinned_in((int64_t)in_bytes));
    memcpy(stage, feats, in_bytes);
    float* d_enc = nullptr;
    run_sized([&](const Ctx& c) {
        Arena& ar = *c.arena;
        float* d_x = ar.take<float>((int64_t)B * T * cf.feat);
        int* d_slots = ar.take<int>(B);
        long long* d_plen = ar.take<long long>(B);
        int* d_chunks = ar.take<int>(B);
        if (!c.dry) {
            K2_HIP(hipMemcpyAsync(d_x, stage, in_bytes, hipMemcpyHostToDevice, c.stream));
            K2_HIP(hipMemcpyAsync(d_slots, slots, sizeof(int) * B, hipMemcpyHostToDevice, c.stream));
            K2_HIP(hipMemcpyAsync(d_chunks, nchunks, sizeof(int) * B, hipMemcpyHostToDevice, c.stream));
            K2_HIP(hipMemcpyAsync(d_plen, plens, sizeof(long long) * B, hipMemcpyHostToDevice, c.stream));
        }
        logfloor_inplace(c, d_x, (long long)B * T * cf.feat);
        d_enc = online_encoder_zip2(c, d_x, d_slots, d_plen, d_chunks, B);
    });
    K2_HIP(hipMemcpyAsync(enc_out, d_enc, sizeof(float) * (size_t)B * Tp * cf.enc_dim(), hipMemcpyDeviceToHost, stream_));
    K2_HIP(hipStreamSynchronize(stream_));
}

}  // namespace k2hip
