// Streaming Zipformer (v1) encoder chunk on the gfx950 kernels: Model_type "zipformer" -> OnlineProjOfZipformer
// (OnlineRecognizer.cs:28-30).  Replaces, per tick, stack_states (OnlineProjOfZipformer.cs:133-300) +
// EncoderProj (:407-516: x [B,39,80] + 7 state tensors per stack -> encoder_out [B,8,512] + new states) + unstack_states:
// the states (cached_len / cached_avg / cached_key / cached_val / cached_val2 / cached_conv1 / cached_conv2, :56-111) stay in
// each stream's slot of the device pool and the kernels index them by slot.
//
// The graph is icefall's pruned_transducer_stateless7_streaming Zipformer.streaming_forward (not in the reference):
// Conv2dSubsampling ((T-7)//2 frames); per stack [skip SimpleCombiner] -> layers, behind AttentionDownsample / SimpleUpsample /
// out_combiner when the stack runs at a lower rate; AttentionDownsample by 2; joiner.encoder_proj (applied by the ONNX encoder
// wrapper).  Every Linear / pointwise conv is the fp32 MFMA GEMM with its bias, DoubleSwish and residual fused in the epilogue.
#include <cmath>

#include "engine.h"

namespace k2hip {

// RelPositionalEncoding.forward(x, left_context_len): row n <-> relative position (Tc + left - 1) - n, width D
const float* Engine::sinus_pos_emb(int Tc, int left, int D) {
    const auto key = std::make_tuple(Tc, left, D);
    auto it = sinus_cache_.find(key);
    if (it != sinus_cache_.end()) return it->second;
    if (sinus_cache_.size() > 256) {  // many distinct utterance lengths: start over
        synchronize();
        for (auto& kv : sinus_cache_) (void)hipFree(kv.second);
        sinus_cache_.clear();
    }
    const int n2 = left + 2 * Tc - 1;
    std::vector<float> pe((size_t)n2 * D);
    const float cc = -(logf(10000.0f) / (float)D);
    for (int n = 0; n < n2; n++) {
        const float r = (float)(left + Tc - 1 - n);
        for (int k = 0; k < D / 2; k++) {
            const float div = expf((float)(2 * k) * cc);
            pe[(size_t)n * D + 2 * k] = sinf(r * div);
            pe[(size_t)n * D + 2 * k + 1] = cosf(r * div);
        }
    }
    float* d = nullptr;
    K2_HIP(hipMalloc(&d, pe.size() * sizeof(float)));
    K2_HIP(copy_blocking(d, pe.data(), pe.size() * sizeof(float), hipMemcpyHostToDevice));
    sinus_cache_[key] = d;
    return d;
}

// Conv2dSubsampling (v1), NHWC: x [B,T,80] -> [B*Tc, D0], Tc = (T-7)//2
float* Engine::zip1_embed(const Ctx& c, const float* x, int B, int T, int* Tc_out) {
    const Model& m = *model_;
    const int F0 = 80, T1 = T - 2, T2 = (T1 - 3) / 2 + 1, F2 = (F0 - 3) / 2 + 1, T3 = T2 - 2, F3 = (F2 - 3) / 2 + 1;
    const int D0 = m.cfg().dim[0];
    K2_REQUIRE(T >= 9 && T3 > 0 && F3 == 19, "zipformer embed: chunk of %d frames unsupported", T);
    Arena& ar = *c.arena;
    float* out = ar.take<float>((int64_t)B * T3 * D0);
    int64_t mark = ar.mark();
    float* a1 = ar.take<float>((int64_t)B * T1 * F0 * 8);
    conv0_nopad_dswish(c, x, m.w("encoder.encoder_embed.conv.0.weight"), m.w("encoder.encoder_embed.conv.0.bias"), a1, B, T, F0);
    c.add_flops(0, 2.0 * B * T1 * (double)F0 * 8 * 9, 0);
    float* a2 = ar.take<float>((int64_t)B * T2 * F2 * 32);
    {
        GemmArgs g;
        g.A = a1; g.W = m.w("encoder.encoder_embed.conv.3.weight#ohwi"); g.ldw = 72; g.bias = m.w("encoder.encoder_embed.conv.3.bias");
        g.C = a2; g.ldc = 32; g.M = B * T2 * F2; g.N = 32; g.K = 72; g.act = ACT_DOUBLE_SWISH;
        g.cv_Fout = F2; g.cv_Tout = T2; g.cv_Tin = T1; g.cv_Fin = F0; g.cv_C = 8; g.cv_st = 2; g.cv_sf = 2;
        g.seg_len = 24; g.seg_stride = F0 * 8;
        gemm(c, g);
    }
    float* a3 = ar.take<float>((int64_t)B * T3 * F3 * 128);
    {
        GemmArgs g;
        g.A = a2; g.W = m.w("encoder.encoder_embed.conv.6.weight#ohwi"); g.ldw = 288; g.bias = m.w("encoder.encoder_embed.conv.6.bias");
        g.C = a3; g.ldc = 128; g.M = B * T3 * F3; g.N = 128; g.K = 288; g.act = ACT_DOUBLE_SWISH;
        g.cv_Fout = F3; g.cv_Tout = T3; g.cv_Tin = T2; g.cv_Fin = F2; g.cv_C = 32; g.cv_st = 1; g.cv_sf = 2;
        g.seg_len = 96; g.seg_stride = F2 * 32;
        gemm(c, g);
    }
    linear(c, a3, F3 * 128, m.w("encoder.encoder_embed.out.weight#fc"), m.w("encoder.encoder_embed.out.bias"), out, D0, B * T3, F3 * 128, D0);
    ar.rewind(mark);
    *Tc_out = T3;
    return out;
}

// ZipformerEncoderLayer.streaming_forward, in place on x [B*Tc, D]; l = global layer index (slot layout), pp = linear_pos(pos_emb)
void Engine::zip1_layer(const Ctx& c, int si, const std::string& pfx, int l, float* x, const float* pp, const int* d_slots, int B, int Tc,
                        int L) {
    const Model& m = *model_;
    const Config& cf = m.cfg();
    const int D = cf.dim[si], A = cf.att[si], H = cf.heads[si], F = cf.ff[si], K = cf.kern[si];
    const int M = B * Tc, KL = L + Tc, KLp = (KL + 3) & ~3, A2 = A / 2, vd = A2 / H, inproj = 2 * A + A2 + 4 * H;
    auto w = [&](const char* suffix) { return m.w(pfx + suffix); };
    Arena& ar = *c.arena;
    int64_t mark = ar.mark();
    const long long SS = lay_.floats_per_stream;
    float* src = ar.take<float>((int64_t)M * D);
    float* hid = ar.take<float>((int64_t)M * std::max({F, 2 * D, inproj}));
    float* tmp = ar.take<float>((int64_t)M * std::max(D, A2));
    float* kcat = ar.take<float>((int64_t)B * KL * A);
    float* vcat = ar.take<float>((int64_t)B * KL * A2);
    float* aw = ar.take<float>((int64_t)H * B * Tc * KLp);

    auto feed_forward = [&](int k, const float* in, float* out) {
        const std::string n = "feed_forward" + std::to_string(k);
        linear(c, in, D, w((n + ".in_proj.weight").c_str()), w((n + ".in_proj.bias").c_str()), hid, F, M, D, F, ACT_DOUBLE_SWISH);
        linear(c, hid, F, w((n + ".out_proj.weight").c_str()), w((n + ".out_proj.bias").c_str()), out, D, M, F, D, ACT_NONE, in, D);
    };
    // tmp[b, :, h*vd : (h+1)*vd] = aw[h][b] (Tc x KL) . vcat[b] (KL x A/2)[:, h*vd : (h+1)*vd]
    auto attn_apply = [&]() {
        GemmArgs g;
        g.A = aw; g.lda = KLp; g.sA0 = (long long)Tc * KLp; g.sA1 = (long long)B * Tc * KLp;
        g.W = vcat; g.w_kn = 1; g.ldw = A2; g.sW0 = (long long)KL * A2; g.sW1 = vd;
        g.C = tmp; g.ldc = A2; g.sC0 = (long long)Tc * A2; g.sC1 = vd;
        g.M = Tc; g.N = vd; g.K = KL; g.nb0 = B; g.nb1 = H;
        gemm(c, g);
    };
    auto conv_module = [&](int k, long long cache_off) {
        const std::string n = "conv_module" + std::to_string(k);
        linear(c, src, D, w((n + ".pointwise_conv1.weight").c_str()), w((n + ".pointwise_conv1.bias").c_str()), hid, 2 * D, M, D, 2 * D);
        z1_glu_conv(c, hid, online_pool_, SS, cache_off, d_slots, w((n + ".depthwise_conv.weight").c_str()),
                    w((n + ".depthwise_conv.bias").c_str()), tmp, B, Tc, D, K);
        linear(c, tmp, D, w((n + ".pointwise_conv2.weight").c_str()), w((n + ".pointwise_conv2.bias").c_str()), src, D, M, D, D, ACT_NONE, src, D);
    };

    feed_forward(1, x, src);
    z1_pool(c, src, online_pool_, SS, lay_.nonlin[l], lay_.clen[l], d_slots, tmp, B, Tc, D);
    linear(c, tmp, D, w("pooling.proj.weight"), nullptr, src, D, M, D, D, ACT_NONE, src, D);
    // self_attn.streaming_forward: in_proj -> q | k | v | p; keys and values behind their caches; weights kept for the second use
    linear(c, src, D, w("self_attn.in_proj.weight"), w("self_attn.in_proj.bias"), hid, inproj, M, D, inproj);
    cat_shift(c, online_pool_, SS, lay_.key[l], d_slots, hid + A, inproj, kcat, B, L, Tc, A);
    cat_shift(c, online_pool_, SS, lay_.val1[l], d_slots, hid + 2 * A, inproj, vcat, B, L, Tc, A2);
    z1_attn(c, hid, inproj, kcat, pp, aw, B, Tc, L, KLp, H, A);
    if (!attn_av_out(c, aw, vcat, w("self_attn.out_proj.weight"), w("self_attn.out_proj.bias"), src, B, Tc, KL, KLp, H, vd, D)) {
        attn_apply();
        linear(c, tmp, A2, w("self_attn.out_proj.weight"), w("self_attn.out_proj.bias"), src, D, M, A2, D, ACT_NONE, src, D);
    }
    conv_module(1, lay_.conv1[l]);
    feed_forward(2, src, src);
    // self_attn.streaming_forward2
    linear(c, src, D, w("self_attn.in_proj2.weight"), nullptr, hid, A2, M, D, A2);
    cat_shift(c, online_pool_, SS, lay_.val2[l], d_slots, hid, A2, vcat, B, L, Tc, A2);
    if (!attn_av_out(c, aw, vcat, w("self_attn.out_proj2.weight"), w("self_attn.out_proj2.bias"), src, B, Tc, KL, KLp, H, vd, D)) {
        attn_apply();
        linear(c, tmp, A2, w("self_attn.out_proj2.weight"), w("self_attn.out_proj2.bias"), src, D, M, A2, D, ACT_NONE, src, D);
    }
    conv_module(2, lay_.conv2[l]);
    feed_forward(3, src, src);
    z1_norm_bypass(c, src, x, w("norm_final.eps"), w("bypass_scale"), x, M, D);
    ar.rewind(mark);
}

// one chunk for B streams: x [B, T, 80] (log-floored) -> enc [B, T', J]
float* Engine::zip1_chunk(const Ctx& c, const float* x, const int* d_slots, int B, int* Tp_out) {
    const Model& m = *model_;
    const Config& cf = m.cfg();
    Arena& ar = *c.arena;
    int Tc = 0;
    float* cur = zip1_embed(c, x, B, cf.chunk_T, &Tc);
    const int M = B * Tc;
    float* outputs[kMaxStacks] = {nullptr};
    int Dcur = cf.dim[0], l = 0;
    auto skip_layer = [&](int i) {  // Zipformer._init_skip_modules
        if (i <= 1 || cf.ds[i - 1] <= cf.ds[i]) return -1;
        for (int j = i - 2; j >= 0; j--)
            if (cf.ds[j] <= cf.ds[i] || j == 0) return j;
        return -1;
    };
    for (int si = 0; si < cf.ns; si++) {
        const int D = cf.dim[si], ds = cf.ds[si], L = cf.left[si], H = cf.heads[si];
        const int k = skip_layer(si);
        if (k >= 0) {
            float* y = ar.take<float>((int64_t)M * Dcur);
            z1_combine(c, outputs[k], cf.dim[k], cur, Dcur, m.wf("encoder.skip_modules.%d.weight1", si), nullptr, 1, B, Tc, Tc, y);
            cur = y;
        }
        char pfx[96];
        if (ds == 1) {
            K2_REQUIRE(D == Dcur, "zipformer: stack %d has downsampling 1 but changes width %d -> %d", si, Dcur, D);
            float* xi = ar.take<float>((int64_t)M * D);  // the layers run in place; earlier outputs stay intact for skip connections
            if (!c.dry) K2_HIP(hipMemcpyAsync(xi, cur, sizeof(float) * (size_t)M * D, hipMemcpyDeviceToDevice, c.stream));
            for (int li = 0; li < cf.nlayer[si]; li++, l++) {
                snprintf(pfx, sizeof pfx, "encoder.encoders.%d.layers.%d.", si, li);
                const float* pp = c.dry ? nullptr : pos_proj_cached(c, 2000 + l, sinus_pos_emb(Tc, L, D), D, m.w(std::string(pfx) + "self_attn.linear_pos.weight"), 2 * Tc - 1 + L, H * 4);
                zip1_layer(c, si, pfx, l, xi, pp, d_slots, B, Tc, L);
            }
            cur = xi;
        } else {
            K2_REQUIRE(D >= Dcur, "zipformer: stack %d narrows %d -> %d (unsupported)", si, Dcur, D);
            const int Td = (Tc + ds - 1) / ds;
            float* y = ar.take<float>((int64_t)M * D);
            int64_t mark = ar.mark();
            float* xd = ar.take<float>((int64_t)B * Td * D);
            z1_attn_downsample(c, cur, m.wf("encoder.encoders.%d.downsample.query", si), xd, B, Tc, Dcur, D, ds);
            if (D > Dcur) {  // extra channels: extra_proj over the ds frames of a group side by side (= a reshape when ds | Tc)
                K2_REQUIRE(Tc % ds == 0, "zipformer: %d frames per chunk not divisible by downsampling %d with widening stacks", Tc, ds);
                linear(c, cur, ds * Dcur, m.wf("encoder.encoders.%d.downsample.extra_proj.weight", si), nullptr, xd + Dcur, D, B * Td,
                       ds * Dcur, D - Dcur);
            }
            for (int li = 0; li < cf.nlayer[si]; li++, l++) {
                snprintf(pfx, sizeof pfx, "encoder.encoders.%d.encoder.layers.%d.", si, li);
                const float* pp = c.dry ? nullptr : pos_proj_cached(c, 2000 + l, sinus_pos_emb(Td, L, D), D, m.w(std::string(pfx) + "self_attn.linear_pos.weight"), 2 * Td - 1 + L, H * 4);
                zip1_layer(c, si, pfx, l, xd, pp, d_slots, B, Td, L);
            }
            z1_combine(c, cur, Dcur, xd, D, m.wf("encoder.encoders.%d.out_combiner.weight1", si), m.wf("encoder.encoders.%d.upsample.bias", si),
                       ds, B, Tc, Td, y);
            ar.rewind(mark);
            cur = y;
            Dcur = D;
        }
        outputs[si] = cur;
    }
    const int Tp = (Tc + 1) / 2;
    float* dsd = ar.take<float>((int64_t)B * Tp * Dcur);
    z1_attn_downsample(c, cur, m.w("encoder.downsample_output.query"), dsd, B, Tc, Dcur, Dcur, 2);
    float* enc = ar.take<float>((int64_t)B * Tp * cf.J);
    linear(c, dsd, Dcur, m.w("joiner.encoder_proj.weight"), m.w("joiner.encoder_proj.bias"), enc, cf.J, B * Tp, Dcur, cf.J);
    *Tp_out = Tp;
    return enc;
}

// ------------------------------------------------------------------------------------------------------------------------
// Offline Zipformer v1: Model_type "zipformer" in OfflineRecognizer's switch (OfflineRecognizer.cs:40-44 -> OfflineProjOfTransducer,
// x [B,T,80] with x_lens = T, :48-92).  icefall's pruned_transducer_stateless7 Zipformer.forward: the modules of the streaming path
// with mean pooling over the utterance, attention over the whole utterance and centred depthwise convolutions -- i.e. the offline
// Zipformer2 kernels (32-row attention strips on the MFMA, fused attention-apply + out_proj, LDS-tiled GLU + depthwise conv) with
// v1's row layout (q | k | v | p, head size attention_dim / heads).
// ------------------------------------------------------------------------------------------------------------------------
void Engine::zip1_layer_offline(const Ctx& c, int si, const std::string& pfx, int l, float* x, const float* pe, int B, int T) {
    const Model& m = *model_;
    const Config& cf = m.cfg();
    const int D = cf.dim[si], A = cf.att[si], H = cf.heads[si], F = cf.ff[si], K = cf.kern[si];
    const int M = B * T, Tp = (T + 3) & ~3, A2 = A / 2, vd = A2 / H, hd = A / H, inproj = 2 * A + A2 + 4 * H;
    auto w = [&](const char* suffix) { return m.w(pfx + suffix); };
    Arena& ar = *c.arena;
    int64_t mark = ar.mark();
    float* src = ar.take<float>((int64_t)M * D);
    float* hid = ar.take<float>((int64_t)M * std::max({F, 2 * D, inproj}));
    float* tmp = ar.take<float>((int64_t)M * std::max(D, A2));
    float* aw = ar.take<float>((int64_t)H * B * T * Tp);
    float* pool = ar.take<float>((int64_t)2 * B * D);

    auto feed_forward = [&](int k, const float* in, float* out) {
        const std::string n = "feed_forward" + std::to_string(k);
        linear(c, in, D, w((n + ".in_proj.weight").c_str()), w((n + ".in_proj.bias").c_str()), hid, F, M, D, F, ACT_DOUBLE_SWISH);
        linear(c, hid, F, w((n + ".out_proj.weight").c_str()), w((n + ".out_proj.bias").c_str()), out, D, M, F, D, ACT_NONE, in, D);
    };
    auto attn_apply = [&](const float* v, int ldv, const char* ow, const char* ob) {  // src += out_proj(concat_h(aw_h . v_h)) + b
        if (ldv == A2 && attn_av_out(c, aw, v, w(ow), w(ob), src, B, T, T, Tp, H, vd, D)) return;
        GemmArgs g;
        g.A = aw; g.lda = Tp; g.sA0 = (long long)T * Tp; g.sA1 = (long long)B * T * Tp;
        g.W = v; g.w_kn = 1; g.ldw = ldv; g.sW0 = (long long)T * ldv; g.sW1 = vd;
        g.C = tmp; g.ldc = A2; g.sC0 = (long long)T * A2; g.sC1 = vd;
        g.M = T; g.N = vd; g.K = T; g.nb0 = B; g.nb1 = H;
        gemm(c, g);
        linear(c, tmp, A2, w(ow), w(ob), src, D, M, A2, D, ACT_NONE, src, D);
    };
    auto conv_module = [&](int k) {
        const std::string n = "conv_module" + std::to_string(k);
        linear(c, src, D, w((n + ".pointwise_conv1.weight").c_str()), w((n + ".pointwise_conv1.bias").c_str()), hid, 2 * D, M, D, 2 * D);
        glu_dwconv1d_dswish(c, hid, w((n + ".depthwise_conv.weight#kd").c_str()), w((n + ".depthwise_conv.bias").c_str()), tmp, B, T, D, K);
        linear(c, tmp, D, w((n + ".pointwise_conv2.weight").c_str()), w((n + ".pointwise_conv2.bias").c_str()), src, D, M, D, D, ACT_NONE, src, D);
    };

    feed_forward(1, x, src);
    z1_mean(c, src, pool, B, T, D);                                               // pooling: the utterance mean, projected, added to every frame
    linear(c, pool, D, w("pooling.proj.weight"), nullptr, pool + (long long)B * D, D, B, D, D);
    z1_add_bcast(c, src, pool + (long long)B * D, B, T, D);
    linear(c, src, D, w("self_attn.in_proj.weight"), w("self_attn.in_proj.bias"), hid, inproj, M, D, inproj);
    const float* pp = pos_proj_cached(c, 5000 + l, pe, D, w("self_attn.linear_pos.weight"), 2 * T - 1, H * 4);
    attn_scores_softmax(c, hid, inproj, pp, aw, B, T, Tp, H, hd, A, 2 * A + A2);
    {   // values: columns [2A, 2A + A/2) of the projected rows -> contiguous [M, A/2] (the fused kernel wants them dense)
        float* v = ar.take<float>((int64_t)M * A2);
        copy_cols(c, hid, inproj, 2 * A, v, A2, 0, M, A2);
        attn_apply(v, A2, "self_attn.out_proj.weight", "self_attn.out_proj.bias");
    }
    conv_module(1);
    feed_forward(2, src, src);
    {
        float* v2 = ar.take<float>((int64_t)M * A2);
        linear(c, src, D, w("self_attn.in_proj2.weight"), nullptr, v2, A2, M, D, A2);
        attn_apply(v2, A2, "self_attn.out_proj2.weight", "self_attn.out_proj2.bias");
    }
    conv_module(2);
    feed_forward(3, src, src);
    z1_norm_bypass(c, src, x, w("norm_final.eps"), w("bypass_scale"), x, M, D);
    ar.rewind(mark);
}

// taps: 0 = embed output; 1+i = output of stack i
float* Engine::zip1_forward(const Ctx& c, const float* x, int B, int T, int* Tp_out, int tap, float** tap_ptr, int* tap_rows, int* tap_dim) {
    const Model& m = *model_;
    const Config& cf = m.cfg();
    Arena& ar = *c.arena;
    const int Tc0 = (T - 7) / 2;
    K2_REQUIRE(T >= 9 && Tc0 > 0, "encoder: %d input frames are too few", T);
    const int Tp = (Tc0 + 1) / 2;
    float* enc = ar.take<float>((int64_t)B * Tp * cf.J);
    int Tc = 0;
    float* cur = zip1_embed(c, x, B, T, &Tc);
    if (tap_rows) *tap_rows = B * Tc;
    *Tp_out = Tp;
    if (tap == 0) { *tap_ptr = cur; *tap_dim = cf.dim[0]; return nullptr; }
    const int M = B * Tc;
    float* outputs[kMaxStacks] = {nullptr};
    int Dcur = cf.dim[0], l = 0;
    auto skip_layer = [&](int i) {
        if (i <= 1 || cf.ds[i - 1] <= cf.ds[i]) return -1;
        for (int j = i - 2; j >= 0; j--)
            if (cf.ds[j] <= cf.ds[i] || j == 0) return j;
        return -1;
    };
    for (int si = 0; si < cf.ns; si++) {
        const int D = cf.dim[si], ds = cf.ds[si];
        const int k = skip_layer(si);
        if (k >= 0) {
            float* y = ar.take<float>((int64_t)M * Dcur);
            z1_combine(c, outputs[k], cf.dim[k], cur, Dcur, m.wf("encoder.skip_modules.%d.weight1", si), nullptr, 1, B, Tc, Tc, y);
            cur = y;
        }
        char pfx[96];
        if (ds == 1) {
            K2_REQUIRE(D == Dcur, "zipformer: stack %d has downsampling 1 but changes width %d -> %d", si, Dcur, D);
            float* xi = ar.take<float>((int64_t)M * D);
            if (!c.dry) K2_HIP(hipMemcpyAsync(xi, cur, sizeof(float) * (size_t)M * D, hipMemcpyDeviceToDevice, c.stream));
            const float* pe = c.dry ? nullptr : sinus_pos_emb(Tc, 0, D);
            for (int li = 0; li < cf.nlayer[si]; li++, l++) {
                snprintf(pfx, sizeof pfx, "encoder.encoders.%d.layers.%d.", si, li);
                zip1_layer_offline(c, si, pfx, l, xi, pe, B, Tc);
            }
            cur = xi;
        } else {
            K2_REQUIRE(D >= Dcur, "zipformer: stack %d narrows %d -> %d (unsupported)", si, Dcur, D);
            const int Td = (Tc + ds - 1) / ds;
            float* y = ar.take<float>((int64_t)M * D);
            int64_t mark = ar.mark();
            float* xd = ar.take<float>((int64_t)B * Td * D);
            z1_attn_downsample(c, cur, m.wf("encoder.encoders.%d.downsample.query", si), xd, B, Tc, Dcur, D, ds);
            if (D > Dcur) {  // extra channels: extra_proj over the group's ds frames side by side
                const float* grp = cur;
                if (Tc % ds != 0) {
                    float* gb = ar.take<float>((int64_t)B * Td * ds * Dcur);
                    z1_group_rows(c, cur, gb, B, Tc, Dcur, ds);
                    grp = gb;
                }
                linear(c, grp, ds * Dcur, m.wf("encoder.encoders.%d.downsample.extra_proj.weight", si), nullptr, xd + Dcur, D, B * Td, ds * Dcur,
                       D - Dcur);
            }
            const float* pe = c.dry ? nullptr : sinus_pos_emb(Td, 0, D);
            for (int li = 0; li < cf.nlayer[si]; li++, l++) {
                snprintf(pfx, sizeof pfx, "encoder.encoders.%d.encoder.layers.%d.", si, li);
                zip1_layer_offline(c, si, pfx, l, xd, pe, B, Td);
            }
            z1_combine(c, cur, Dcur, xd, D, m.wf("encoder.encoders.%d.out_combiner.weight1", si), m.wf("encoder.encoders.%d.upsample.bias", si),
                       ds, B, Tc, Td, y);
            ar.rewind(mark);
            cur = y;
            Dcur = D;
        }
        outputs[si] = cur;
        if (tap == si + 1) { *tap_ptr = cur; *tap_dim = Dcur; return nullptr; }
    }
    float* dsd = ar.take<float>((int64_t)B * Tp * Dcur);
    z1_attn_downsample(c, cur, m.w("encoder.downsample_output.query"), dsd, B, Tc, Dcur, Dcur, 2);
    linear(c, dsd, Dcur, m.w("joiner.encoder_proj.weight"), m.w("joiner.encoder_proj.bias"), enc, cf.J, B * Tp, Dcur, cf.J);
    return enc;
}

}  // namespace k2hip
