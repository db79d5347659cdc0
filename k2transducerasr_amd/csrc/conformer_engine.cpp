// Offline Conformer encoder on the gfx950 kernels (BASELINE.json configs[4]).
//
// Reference side: Model_type "conformer" selects OfflineProjOfTransducer (OfflineRecognizer.cs:38-53);
// EncoderProj (OfflineProjOfTransducer.cs:48-92) passes x [B,T,80] with x_lens = T for every row and
// reads encoder_out [B,T',512].  The graph in between is icefall's pruned_transducer_stateless2
// Conformer (the published structure; DESIGN.md section 2 says how it is cross-checked):
//   Conv2dSubsampling(x4) -> 12 x { x += ff_macaron(x); x += rel-pos MHSA(x); x += conv_module(x);
//   x += ff(x); x = BasicNorm(x) } -> joiner.encoder_proj.
// Activations are batch-major [B*T', D]; every Linear / pointwise conv / implicit 3x3 conv / score and
// value product is a gemm_f32_mfma launch with bias + DoubleSwish + residual fused in its epilogue.
#include <cmath>

#include "engine.h"

namespace k2hip {

int Engine::conformer_out_frames(int T) const { return T < 7 ? 0 : ((T - 1) / 2 - 1) / 2; }

// RelPositionalEncoding.extend_pe in float32 arithmetic (as torch evaluates it): row n <-> relative
// position T-1-n, pe[n,2k] = sin(r * div_k), pe[n,2k+1] = cos(r * div_k), div_k = exp(2k * -(ln 1e4 / D))
const float* Engine::conformer_pos_emb(int T) {
    auto it = pe_cache_.find(-T);  // negative keys: conformer tables (the Zipformer cache uses +T)
    if (it != pe_cache_.end()) return it->second;
    const int D = model_->cfg().dim[0], n2 = 2 * T - 1;
    std::vector<float> pe((size_t)n2 * D);
    const float cc = -(logf(10000.0f) / (float)D);
    for (int n = 0; n < n2; n++) {
        const float r = (float)(T - 1 - n);
        for (int k = 0; k < D / 2; k++) {
            const float div = expf((float)(2 * k) * cc);
            pe[(size_t)n * D + 2 * k] = sinf(r * div);
            pe[(size_t)n * D + 2 * k + 1] = cosf(r * div);
        }
    }
    float* d = nullptr;
    K2_HIP(hipMalloc(&d, pe.size() * sizeof(float)));
    K2_HIP(copy_blocking(d, pe.data(), pe.size() * sizeof(float), hipMemcpyHostToDevice));
    pe_cache_[-T] = d;
    return d;
}

// Conv2dSubsampling, NHWC: x [B,T,80] -> [B,T3,D]
float* Engine::conformer_embed(const Ctx& c, const float* x, int B, int T, int* T_out) {
    const Model& m = *model_;
    const int F0 = 80, T2 = (T - 1) / 2, F2 = (F0 - 1) / 2, T3 = (T2 - 1) / 2, F3 = (F2 - 1) / 2, D = m.cfg().dim[0];
    K2_REQUIRE(T3 > 0, "encoder: %d input frames are too few (need >= 7)", T);
    Arena& ar = *c.arena;
    float* out = ar.take<float>((int64_t)B * T3 * D);
    int64_t mark = ar.mark();
    float* a1 = ar.take<float>((int64_t)B * T * F0 * 8);
    conv0_pad1_dswish(c, x, m.w("encoder.encoder_embed.conv.0.weight"), m.w("encoder.encoder_embed.conv.0.bias"), a1, B, T, F0);
    c.add_flops(0, 2.0 * B * T * (double)F0 * 8 * 9, 0);
    float* a2 = ar.take<float>((int64_t)B * T2 * F2 * 32);
    {
        GemmArgs g;
        g.A = a1; g.W = m.w("encoder.encoder_embed.conv.3.weight#ohwi"); g.ldw = 72; g.bias = m.w("encoder.encoder_embed.conv.3.bias");
        g.C = a2; g.ldc = 32; g.M = B * T2 * F2; g.N = 32; g.K = 72; g.act = ACT_DOUBLE_SWISH;
        g.cv_Fout = F2; g.cv_Tout = T2; g.cv_Tin = T; g.cv_Fin = F0; g.cv_C = 8; g.cv_st = 2; g.cv_sf = 2;
        g.seg_len = 24; g.seg_stride = F0 * 8;
        gemm(c, g);
    }
    float* a3 = ar.take<float>((int64_t)B * T3 * F3 * 128);
    {
        GemmArgs g;
        g.A = a2; g.W = m.w("encoder.encoder_embed.conv.6.weight#ohwi"); g.ldw = 288; g.bias = m.w("encoder.encoder_embed.conv.6.bias");
        g.C = a3; g.ldc = 128; g.M = B * T3 * F3; g.N = 128; g.K = 288; g.act = ACT_DOUBLE_SWISH;
        g.cv_Fout = F3; g.cv_Tout = T3; g.cv_Tin = T2; g.cv_Fin = F2; g.cv_C = 32; g.cv_st = 2; g.cv_sf = 2;
        g.seg_len = 96; g.seg_stride = F2 * 32;
        gemm(c, g);
    }
    // (b,t,f,c) flatten == [B*T3, F3*128] with the repacked out.weight
    float* lin = ar.take<float>((int64_t)B * T3 * D);
    linear(c, a3, F3 * 128, m.w("encoder.encoder_embed.out.weight#fc"), m.w("encoder.encoder_embed.out.bias"), lin, D, B * T3, F3 * 128, D);
    basicnorm(c, lin, m.w("encoder.encoder_embed.out_norm.eps"), out, B * T3, D);
    ar.rewind(mark);
    *T_out = T3;
    return out;
}

// ConformerEncoderLayer.forward (eval), in place on x [B*T, D]
void Engine::conformer_layer(const Ctx& c, int li, float* x, const float* pe, int B, int T) {
    const Model& m = *model_;
    const Config& cf = m.cfg();
    const int D = cf.dim[0], F = cf.ff[0], H = cf.heads[0], K = cf.kern[0], dk = D / H;
    const int M = B * T, Tp = (T + 3) & ~3, NP = 2 * T - 1, NPp = (NP + 3) & ~3;
    char p[96];
    snprintf(p, sizeof p, "encoder.encoder.layers.%d.", li);
    auto w = [&](const char* suffix) { return m.w(std::string(p) + suffix); };
    Arena& ar = *c.arena;
    int64_t mark = ar.mark();
    float* hid = ar.take<float>((int64_t)M * std::max(F, 3 * D));

    auto feed_forward = [&](const char* name) {
        std::string n(name);
        linear(c, x, D, w((n + ".0.weight").c_str()), w((n + ".0.bias").c_str()), hid, F, M, D, F, ACT_DOUBLE_SWISH);
        linear(c, hid, F, w((n + ".4.weight").c_str()), w((n + ".4.bias").c_str()), x, D, M, F, D, ACT_NONE, x, D);
    };

    feed_forward("feed_forward_macaron");
    {   // x += out_proj(softmax((q+u) k^T + rel_shift((q+v) p^T)) v)
        float* qkv = hid;
        linear(c, x, D, w("self_attn.in_proj.weight"), w("self_attn.in_proj.bias"), qkv, 3 * D, M, D, 3 * D);
        const float* pp = pos_proj_cached(c, 3000 + li, pe, D, w("self_attn.linear_pos.weight"), NP, D);
        float* qu = ar.take<float>((int64_t)M * D);
        float* qv = ar.take<float>((int64_t)M * D);
        float* ac = ar.take<float>((int64_t)B * H * T * Tp);
        float* bd = ar.take<float>((int64_t)B * H * T * NPp);
        // scores + softmax in one kernel (no [T,T] / [T,2T-1] score tensors), the two query operands q * scaling + pos_bias_u / _v formed
        // on its way in (round 5: k_conformer_qprep was a launch of its own per layer); the GEMM form below is the fallback for shapes it
        // does not cover (very long utterances, unusual head sizes)
        const float scaling = 1.0f / sqrtf((float)dk);
        if (!conformer_scores_softmax(c, qkv, nullptr, qkv + D, 3 * D, pp, ac, B, H, T, Tp, D, 3 * D, w("self_attn.pos_bias_u"), w("self_attn.pos_bias_v"),
                                      scaling)) {
            conformer_qprep(c, qkv, w("self_attn.pos_bias_u"), w("self_attn.pos_bias_v"), qu, qv, M, D, scaling);
            {   // ac[b,h] = qu[b,:,h] . k[b,:,h]^T      z = h + H*b
                GemmArgs g;
                g.A = qu; g.lda = D; g.sA0 = dk; g.sA1 = (long long)T * D;
                g.W = qkv + D; g.ldw = 3 * D; g.sW0 = dk; g.sW1 = (long long)T * 3 * D;
                g.C = ac; g.ldc = Tp; g.sC0 = (long long)T * Tp; g.sC1 = (long long)H * T * Tp;
                g.M = T; g.N = T; g.K = dk; g.nb0 = H; g.nb1 = B;
                gemm(c, g);
            }
            {   // bd[b,h] = qv[b,:,h] . p[:,h]^T
                GemmArgs g;
                g.A = qv; g.lda = D; g.sA0 = dk; g.sA1 = (long long)T * D;
                g.W = pp; g.ldw = D; g.sW0 = dk; g.sW1 = 0;
                g.C = bd; g.ldc = NPp; g.sC0 = (long long)T * NPp; g.sC1 = (long long)H * T * NPp;
                g.M = T; g.N = NP; g.K = dk; g.nb0 = H; g.nb1 = B;
                gemm(c, g);
            }
            conformer_softmax_shift(c, ac, bd, B * H, T, Tp, NPp);
        }
        float* ctxv = qu;  // [M, D], column block h
        {   // ctxv[b,:,h] = w[b,h] . v[b,:,h]
            GemmArgs g;
            g.A = ac; g.lda = Tp; g.sA0 = (long long)T * Tp; g.sA1 = (long long)H * T * Tp;
            g.W = qkv + 2 * D; g.w_kn = 1; g.ldw = 3 * D; g.sW0 = dk; g.sW1 = (long long)T * 3 * D;
            g.C = ctxv; g.ldc = D; g.sC0 = dk; g.sC1 = (long long)T * D;
            g.M = T; g.N = dk; g.K = T; g.nb0 = H; g.nb1 = B;
            gemm(c, g);
        }
        linear(c, ctxv, D, w("self_attn.out_proj.weight"), w("self_attn.out_proj.bias"), x, D, M, D, D, ACT_NONE, x, D);
    }
    {   // x += pointwise_conv2(DoubleSwish(depthwise(GLU(pointwise_conv1(x)))))
        float* tmp = ar.take<float>((int64_t)M * D);
        linear(c, x, D, w("conv_module.pointwise_conv1.weight"), w("conv_module.pointwise_conv1.bias"), hid, 2 * D, M, D, 2 * D);
        glu_dwconv1d_dswish(c, hid, w("conv_module.depthwise_conv.weight#kd"), w("conv_module.depthwise_conv.bias"), tmp, B, T, D, K);
        linear(c, tmp, D, w("conv_module.pointwise_conv2.weight"), w("conv_module.pointwise_conv2.bias"), x, D, M, D, D, ACT_NONE, x, D);
    }
    feed_forward("feed_forward");
    basicnorm(c, x, w("norm_final.eps"), x, M, D);
    ar.rewind(mark);
}

// taps: 0 = encoder_embed output; 1+i = output of layer i
float* Engine::conformer_forward(const Ctx& c, const float* x, int B, int T, int* Tp, int tap, float** tap_ptr, int* tap_rows,
                                 int* tap_dim) {
    const Model& m = *model_;
    const Config& cf = m.cfg();
    Arena& ar = *c.arena;
    const int Tpp = conformer_out_frames(T);
    K2_REQUIRE(Tpp > 0, "encoder: %d input frames are too few", T);
    float* enc_out = ar.take<float>((int64_t)B * Tpp * cf.J);
    int T3 = 0;
    float* xe = conformer_embed(c, x, B, T, &T3);
    if (tap_rows) *tap_rows = B * T3;
    const float* pe = c.dry ? nullptr : conformer_pos_emb(T3);
    for (int li = 0; li <= cf.nlayer[0]; li++) {
        if (tap == li) {
            *tap_ptr = xe;
            *tap_dim = cf.dim[0];
            return nullptr;
        }
        if (li < cf.nlayer[0]) conformer_layer(c, li, xe, pe, B, T3);
    }
    linear(c, xe, cf.dim[0], m.w("joiner.encoder_proj.weight"), m.w("joiner.encoder_proj.bias"), enc_out, cf.J, B * T3, cf.dim[0], cf.J);
    *Tp = T3;
    return enc_out;
}

// ---------------------------------------------------------------------------------------------------------------------
// Streaming (OnlineProjOfConformer): Conformer.streaming_forward / chunk_forward with causal convolutions and
// right_context R >= 0 (R > 0: see conformer_chunk).  Per stream slot the pool holds cached_attn [L][left][D] (the layers' attention INPUT; keys and values are
// re-projected every chunk) followed by cached_conv [L][K-1][D] (GLU outputs feeding the causal depthwise conv), the shapes of
// OnlineProjOfConformer.GetEncoderInitStates (:55-82).
// ---------------------------------------------------------------------------------------------------------------------
const float* Engine::conformer_pos_emb_left(int Tc, int left) {
    const int key = -(50000000 + Tc * 10000 + left);
    auto it = pe_cache_.find(key);
    if (it != pe_cache_.end()) return it->second;
    const int D = model_->cfg().dim[0], n2 = left + 2 * Tc - 1;
    std::vector<float> pe((size_t)n2 * D);
    const float cc = -(logf(10000.0f) / (float)D);
    for (int n = 0; n < n2; n++) {
        const float r = (float)(left + Tc - 1 - n);  // RelPositionalEncoding.forward(x, left_context)
        for (int k = 0; k < D / 2; k++) {
            const float div = expf((float)(2 * k) * cc);
            pe[(size_t)n * D + 2 * k] = sinf(r * div);
            pe[(size_t)n * D + 2 * k + 1] = cosf(r * div);
        }
    }
    float* d = nullptr;
    K2_HIP(hipMalloc(&d, pe.size() * sizeof(float)));
    K2_HIP(copy_blocking(d, pe.data(), pe.size() * sizeof(float), hipMemcpyHostToDevice));
    pe_cache_[key] = d;
    return d;
}

// one chunk for B streams: x [B, T, 80] -> enc [B, chunk, J]
float* Engine::conformer_chunk(const Ctx& c, const float* x, const int* d_slots, const long long* d_plen, int B, int* Tc_out) {
    const Model& m = *model_;
    const Config& cf = m.cfg();
    Arena& ar = *c.arena;
    const int D = cf.dim[0], F = cf.ff[0], H = cf.heads[0], K = cf.kern[0], dk = D / H, L = cf.nlayer[0], left = cf.left[0];
    // Tc frames go through the layers: chunk_size + right_context (the look-ahead frames are seen by this step's attention and
    // convolution, stay out of both caches and are cut from the output: they come again as the next chunk's first frames)
    const int R = cf.right;
    const int T3 = conformer_out_frames(cf.chunk_T), Tc = T3 - 2, KL = left + Tc, KLp = (KL + 3) & ~3, NP = left + 2 * Tc - 1,
              NPp = (NP + 3) & ~3, M = B * Tc, Tout = Tc - R;
    float* enc = ar.take<float>((int64_t)B * Tout * cf.J);
    int t3 = 0;
    float* e = conformer_embed(c, x, B, cf.chunk_T, &t3);
    K2_REQUIRE(t3 == T3 && Tout > 0, "internal: conformer chunk yields %d frames", t3);
    float* xs = ar.take<float>((int64_t)M * D);
    slice_rows(c, e, xs, B, T3, 1, Tc, D);  // embed[:, 1:-1]: the edge frames saw the conv padding
    const float* pe = c.dry ? nullptr : conformer_pos_emb_left(Tc, left);
    float* hid = ar.take<float>((int64_t)M * std::max(F, 2 * D));
    float* cat = ar.take<float>((int64_t)B * KL * D);
    float* kv = ar.take<float>((int64_t)B * KL * 2 * D);
    float* q = ar.take<float>((int64_t)M * D);
    float* qu = ar.take<float>((int64_t)M * D);
    float* qv = ar.take<float>((int64_t)M * D);
    float* ac = ar.take<float>((int64_t)B * H * Tc * KLp);
    float* bd = ar.take<float>((int64_t)B * H * Tc * NPp);
    float* ccat = ar.take<float>((int64_t)B * (K - 1 + Tc) * D);
    float* g = ar.take<float>((int64_t)M * D);
    const long long stride = lay_.floats_per_stream;
    for (int li = 0; li < L; li++) {
        char p[96];
        snprintf(p, sizeof p, "encoder.encoder.layers.%d.", li);
        auto w = [&](const char* suffix) { return m.w(std::string(p) + suffix); };
        auto feed_forward = [&](const char* name) {
            std::string n(name);
            linear(c, xs, D, w((n + ".0.weight").c_str()), w((n + ".0.bias").c_str()), hid, F, M, D, F, ACT_DOUBLE_SWISH);
            linear(c, hid, F, w((n + ".4.weight").c_str()), w((n + ".4.bias").c_str()), xs, D, M, F, D, ACT_NONE, xs, D);
        };
        feed_forward("feed_forward_macaron");
        {
            // key = [cached_attn ; chunk]; cached_attn <- key[-left:]  (right_context R: key[-(left + R) : -R])
            if (R == 0) cat_shift(c, online_pool_, stride, (long long)li * left * D, d_slots, xs, D, cat, B, left, Tc, D);
            else cat_keep(c, online_pool_, stride, (long long)li * left * D, d_slots, xs, D, cat, B, left, Tc, D, R);
            const float* Win = w("self_attn.in_proj.weight");
            const float* bin = w("self_attn.in_proj.bias");
            linear(c, xs, D, Win, bin, q, D, M, D, D);                                    // q from the chunk
            linear(c, cat, D, Win + (long long)D * D, bin + D, kv, 2 * D, B * KL, D, 2 * D);  // k | v from the keys
            const float* pp = pos_proj_cached(c, 4000 + li, pe, D, w("self_attn.linear_pos.weight"), NP, D);
            conformer_qprep(c, q, w("self_attn.pos_bias_u"), w("self_attn.pos_bias_v"), qu, qv, M, D, 1.0f / sqrtf((float)dk), D);
            {
                GemmArgs a;
                a.A = qu; a.lda = D; a.sA0 = dk; a.sA1 = (long long)Tc * D;
                a.W = kv; a.ldw = 2 * D; a.sW0 = dk; a.sW1 = (long long)KL * 2 * D;
                a.C = ac; a.ldc = KLp; a.sC0 = (long long)Tc * KLp; a.sC1 = (long long)H * Tc * KLp;
                a.M = Tc; a.N = KL; a.K = dk; a.nb0 = H; a.nb1 = B;
                gemm(c, a);
            }
            {
                GemmArgs a;
                a.A = qv; a.lda = D; a.sA0 = dk; a.sA1 = (long long)Tc * D;
                a.W = pp; a.ldw = D; a.sW0 = dk; a.sW1 = 0;
                a.C = bd; a.ldc = NPp; a.sC0 = (long long)Tc * NPp; a.sC1 = (long long)H * Tc * NPp;
                a.M = Tc; a.N = NP; a.K = dk; a.nb0 = H; a.nb1 = B;
                gemm(c, a);
            }
            conformer_softmax_shift_stream(c, ac, bd, d_plen, B, H, Tc, left, KLp, NPp);
            float* ctxv = qu;
            {
                GemmArgs a;
                a.A = ac; a.lda = KLp; a.sA0 = (long long)Tc * KLp; a.sA1 = (long long)H * Tc * KLp;
                a.W = kv + D; a.w_kn = 1; a.ldw = 2 * D; a.sW0 = dk; a.sW1 = (long long)KL * 2 * D;
                a.C = ctxv; a.ldc = D; a.sC0 = dk; a.sC1 = (long long)Tc * D;
                a.M = Tc; a.N = dk; a.K = KL; a.nb0 = H; a.nb1 = B;
                gemm(c, a);
            }
            linear(c, ctxv, D, w("self_attn.out_proj.weight"), w("self_attn.out_proj.bias"), xs, D, M, D, D, ACT_NONE, xs, D);
        }
        {   // causal ConvolutionModule: cache holds the K-1 latest GLU outputs
            linear(c, xs, D, w("conv_module.pointwise_conv1.weight"), w("conv_module.pointwise_conv1.bias"), hid, 2 * D, M, D, 2 * D);
            glu_sigmoid(c, hid, g, M, D);
            if (R == 0) cat_shift(c, online_pool_, stride, (long long)L * left * D + (long long)li * (K - 1) * D, d_slots, g, D, ccat, B, K - 1, Tc, D);
            else cat_keep(c, online_pool_, stride, (long long)L * left * D + (long long)li * (K - 1) * D, d_slots, g, D, ccat, B, K - 1, Tc, D, R);
            dwconv_valid_dswish(c, ccat, w("conv_module.depthwise_conv.weight#kd"), w("conv_module.depthwise_conv.bias"), g, B, Tc, D, K);
            linear(c, g, D, w("conv_module.pointwise_conv2.weight"), w("conv_module.pointwise_conv2.bias"), xs, D, M, D, D, ACT_NONE, xs, D);
        }
        feed_forward("feed_forward");
        basicnorm(c, xs, w("norm_final.eps"), xs, M, D);
    }
    const float* xo = xs;
    if (R > 0) {   // x = x[:-right_context]
        float* cut = ar.take<float>((int64_t)B * Tout * D);
        slice_rows(c, xs, cut, B, Tc, 0, Tout, D);
        xo = cut;
    }
    linear(c, xo, D, m.w("joiner.encoder_proj.weight"), m.w("joiner.encoder_proj.bias"), enc, cf.J, B * Tout, D, cf.J);
    *Tc_out = Tout;
    return enc;
}

}  // namespace k2hip
