/*
 * k2_oracle_conformer.c -- CPU restatement of the offline Conformer encoder (included by
 * k2_oracle.c; TEST INFRASTRUCTURE ONLY, see k2_oracle.h).  PARITY UNPINNED.
 *
 * Reference side: Model_type "conformer" selects OfflineProjOfTransducer
 * (K2TransducerAsr/OfflineRecognizer.cs:38-53), whose EncoderProj (:48-92) feeds x [B,T,80] and
 * x_lens = T for every row and reads encoder_out [B,T',512]; everything between is the ONNX graph.
 * That graph is not in the reference; this file restates the published icefall
 * pruned_transducer_stateless2 Conformer inference graph (the recipe the reference's conformer-zh
 * model zoo entry was exported from, README.EN.md:8-35), with Scaled* modules folded:
 *
 *   Conv2dSubsampling: Conv2d(1,8,3,pad 1) DoubleSwish; Conv2d(8,32,3,stride 2) DoubleSwish;
 *     Conv2d(32,128,3,stride 2) DoubleSwish; Linear(128*19 -> D) over (c,f); BasicNorm
 *     T' = ((T-1)//2 - 1)//2
 *   RelPositionalEncoding: sinusoids for relative positions T'-1 .. -(T'-1)
 *   ConformerEncoderLayer (eval): x += ff_macaron(x); x += self_attn(x); x += conv_module(x);
 *     x += ff(x); x = BasicNorm(x)
 *   RelPositionMultiheadAttention: q scaled by d_k^-0.5, (q+u).k^T + rel_shift((q+v).p^T), softmax
 *   ConvolutionModule: pointwise(D->2D) GLU depthwise(K, zero pad) DoubleSwish pointwise(D->D)
 * x_lens = T for all rows (OfflineProjOfTransducer.cs:66-70), so no key-padding mask is ever set.
 */

static inline float double_swish(float x) { return x / (1.0f + expf(1.0f - x)); } /* x * sigmoid(x - 1) */

static void basic_norm(float* y, const float* x, float log_eps, int M, int D) {
    const float eps = expf(log_eps);
#pragma omp parallel for schedule(static)
    for (int r = 0; r < M; r++) {
        const float* xr = x + (size_t)r * D;
        float ss = 0.f;
        for (int d = 0; d < D; d++) ss += xr[d] * xr[d];
        float sc = 1.0f / sqrtf(ss / (float)D + eps);
        for (int d = 0; d < D; d++) y[(size_t)r * D + d] = xr[d] * sc;
    }
}

static int conformer_out_frames(int T) {
    if (T < 7) return 0;
    return ((T - 1) / 2 - 1) / 2;
}

/* Conv2dSubsampling: x [B,T,80] -> [B,T3,D] */
static float* conformer_embed(const k2o_model* m, const float* x, int B, int T, int* T_out) {
    const int F0 = m->feat, D = m->dim[0];
    const int T2 = (T - 1) / 2, F2 = (F0 - 1) / 2, T3 = (T2 - 1) / 2, F3 = (F2 - 1) / 2;
    if (T3 <= 0) return NULL;
    const float* w0 = W(m, "encoder.encoder_embed.conv.0.weight"); /* [8,1,3,3] */
    const float* b0 = W(m, "encoder.encoder_embed.conv.0.bias");
    const float* w1 = W(m, "encoder.encoder_embed.conv.3.weight"); /* [32,8,3,3] */
    const float* b1 = W(m, "encoder.encoder_embed.conv.3.bias");
    const float* w2 = W(m, "encoder.encoder_embed.conv.6.weight"); /* [128,32,3,3] */
    const float* b2 = W(m, "encoder.encoder_embed.conv.6.bias");
    /* channel-first planes, as torch holds them */
    float* a1 = falloc((size_t)B * 8 * T * F0);
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; b++)
        for (int co = 0; co < 8; co++)
            for (int t = 0; t < T; t++)
                for (int f = 0; f < F0; f++) {
                    float s = b0[co];
                    for (int kt = 0; kt < 3; kt++) {
                        int ti = t + kt - 1;
                        if (ti < 0 || ti >= T) continue;
                        for (int kf = 0; kf < 3; kf++) {
                            int fi = f + kf - 1;
                            if (fi < 0 || fi >= F0) continue;
                            s += w0[(co * 3 + kt) * 3 + kf] * x[((size_t)b * T + ti) * F0 + fi];
                        }
                    }
                    a1[(((size_t)b * 8 + co) * T + t) * F0 + f] = double_swish(s);
                }
    float* a2 = falloc((size_t)B * 32 * T2 * F2);
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; b++)
        for (int co = 0; co < 32; co++)
            for (int t = 0; t < T2; t++)
                for (int f = 0; f < F2; f++) {
                    float s = b1[co];
                    for (int ci = 0; ci < 8; ci++)
                        for (int kt = 0; kt < 3; kt++)
                            for (int kf = 0; kf < 3; kf++)
                                s += w1[((co * 8 + ci) * 3 + kt) * 3 + kf] *
                                     a1[(((size_t)b * 8 + ci) * T + 2 * t + kt) * F0 + 2 * f + kf];
                    a2[(((size_t)b * 32 + co) * T2 + t) * F2 + f] = double_swish(s);
                }
    free(a1);
    /* a3 stored [B,T3,128*F3] in torch's (c,f) flatten order: column c*F3 + f */
    float* a3 = falloc((size_t)B * T3 * 128 * F3);
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; b++)
        for (int co = 0; co < 128; co++)
            for (int t = 0; t < T3; t++)
                for (int f = 0; f < F3; f++) {
                    float s = b2[co];
                    for (int ci = 0; ci < 32; ci++)
                        for (int kt = 0; kt < 3; kt++)
                            for (int kf = 0; kf < 3; kf++)
                                s += w2[((co * 32 + ci) * 3 + kt) * 3 + kf] *
                                     a2[(((size_t)b * 32 + ci) * T2 + 2 * t + kt) * F2 + 2 * f + kf];
                    a3[((size_t)b * T3 + t) * (128 * F3) + co * F3 + f] = double_swish(s);
                }
    free(a2);
    float* lin = falloc((size_t)B * T3 * D);
    linear(lin, D, a3, 128 * F3, WT(m, D, 128 * F3, "encoder.encoder_embed.out.weight"), W(m, "encoder.encoder_embed.out.bias"),
           B * T3, 128 * F3, D);
    free(a3);
    float* out = falloc((size_t)B * T3 * D);
    basic_norm(out, lin, W(m, "encoder.encoder_embed.out_norm.eps")[0], B * T3, D);
    free(lin);
    *T_out = T3;
    return out;
}

/* RelPositionalEncoding.extend_pe in torch's float32 arithmetic: row n <-> relative position T-1-n,
 * pe[n,2k] = sin(r * div_k), pe[n,2k+1] = cos(r * div_k), div_k = exp(2k * -(ln 10000 / D)) */
static float* conformer_pos_emb(int T, int D) {
    float* pe = falloc((size_t)(2 * T - 1) * D);
    const float c = -(logf(10000.0f) / (float)D);
    for (int n = 0; n < 2 * T - 1; n++) {
        float r = (float)(T - 1 - n);
        for (int k = 0; k < D / 2; k++) {
            float div = expf((float)(2 * k) * c);
            pe[(size_t)n * D + 2 * k] = sinf(r * div);
            pe[(size_t)n * D + 2 * k + 1] = cosf(r * div);
        }
    }
    return pe;
}

static void conformer_ff(const k2o_model* m, const char* pfx, const char* name, float* src, int M, int D, int F) {
    float* hid = falloc((size_t)M * F);
    linear(hid, F, src, D, WT(m, F, D, "%s%s.0.weight", pfx, name), W(m, "%s%s.0.bias", pfx, name), M, D, F);
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < (size_t)M * F; i++) hid[i] = double_swish(hid[i]);
    float* out = falloc((size_t)M * D);
    linear(out, D, hid, F, WT(m, D, F, "%s%s.4.weight", pfx, name), W(m, "%s%s.4.bias", pfx, name), M, F, D);
    add_inplace(src, out, (size_t)M * D);
    free(hid);
    free(out);
}

static void conformer_self_attn(const k2o_model* m, const char* pfx, float* src, const float* pe, int B, int T, int D, int H) {
    const int M = B * T, dk = D / H, NP = 2 * T - 1;
    const float scaling = 1.0f / sqrtf((float)dk);
    float* qkv = falloc((size_t)M * 3 * D);
    linear(qkv, 3 * D, src, D, WT(m, 3 * D, D, "%sself_attn.in_proj.weight", pfx), W(m, "%sself_attn.in_proj.bias", pfx), M, D, 3 * D);
    float* p = falloc((size_t)NP * D);
    linear(p, D, pe, D, WT(m, D, D, "%sself_attn.linear_pos.weight", pfx), NULL, NP, D, D);
    const float* bu = W(m, "%sself_attn.pos_bias_u", pfx);
    const float* bv = W(m, "%sself_attn.pos_bias_v", pfx);
    float* ctxv = falloc((size_t)M * D);
#pragma omp parallel for collapse(2) schedule(dynamic)
    for (int b = 0; b < B; b++)
        for (int h = 0; h < H; h++) {
            float* sc = (float*)malloc(sizeof(float) * T);
            float qu[256], qv[256];
            for (int i = 0; i < T; i++) {
                const float* q = qkv + ((size_t)b * T + i) * 3 * D + h * dk;
                for (int d = 0; d < dk; d++) {
                    float qs = q[d] * scaling;
                    qu[d] = qs + bu[h * dk + d];
                    qv[d] = qs + bv[h * dk + d];
                }
                float mx = -INFINITY;
                for (int j = 0; j < T; j++) {
                    const float* k = qkv + ((size_t)b * T + j) * 3 * D + D + h * dk;
                    const float* pr = p + (size_t)(T - 1 - i + j) * D + h * dk; /* rel_shift: column T-1-i+j */
                    float ac = 0.f, bd = 0.f;
                    for (int d = 0; d < dk; d++) ac += qu[d] * k[d];
                    for (int d = 0; d < dk; d++) bd += qv[d] * pr[d];
                    sc[j] = ac + bd;
                    if (sc[j] > mx) mx = sc[j];
                }
                float sum = 0.f;
                for (int j = 0; j < T; j++) {
                    sc[j] = expf(sc[j] - mx);
                    sum += sc[j];
                }
                float inv = 1.0f / sum;
                float* o = ctxv + ((size_t)b * T + i) * D + h * dk;
                for (int d = 0; d < dk; d++) o[d] = 0.f;
                for (int j = 0; j < T; j++) {
                    const float* v = qkv + ((size_t)b * T + j) * 3 * D + 2 * D + h * dk;
                    float a = sc[j] * inv;
                    for (int d = 0; d < dk; d++) o[d] += a * v[d];
                }
            }
            free(sc);
        }
    float* out = falloc((size_t)M * D);
    linear(out, D, ctxv, D, WT(m, D, D, "%sself_attn.out_proj.weight", pfx), W(m, "%sself_attn.out_proj.bias", pfx), M, D, D);
    add_inplace(src, out, (size_t)M * D);
    free(out);
    free(ctxv);
    free(p);
    free(qkv);
}

static void conformer_conv_module(const k2o_model* m, const char* pfx, float* src, int B, int T, int D, int K) {
    const int M = B * T, pad = (K - 1) / 2;
    float* x2 = falloc((size_t)M * 2 * D);
    linear(x2, 2 * D, src, D, WT(m, 2 * D, D, "%sconv_module.pointwise_conv1.weight", pfx), W(m, "%sconv_module.pointwise_conv1.bias", pfx),
           M, D, 2 * D);
    float* g = falloc((size_t)M * D);
#pragma omp parallel for schedule(static)
    for (int r = 0; r < M; r++)
        for (int d = 0; d < D; d++) {
            float a = x2[(size_t)r * 2 * D + d], gate = x2[(size_t)r * 2 * D + D + d];
            g[(size_t)r * D + d] = a * (1.0f / (1.0f + expf(-gate)));
        }
    free(x2);
    const float* dw = W(m, "%sconv_module.depthwise_conv.weight", pfx); /* [D,1,K] */
    const float* db = W(m, "%sconv_module.depthwise_conv.bias", pfx);
    float* y = falloc((size_t)M * D);
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; b++)
        for (int t = 0; t < T; t++)
            for (int d = 0; d < D; d++) {
                float s = db[d];
                for (int k = 0; k < K; k++) {
                    int ti = t + k - pad;
                    if (ti < 0 || ti >= T) continue;
                    s += dw[d * K + k] * g[((size_t)b * T + ti) * D + d];
                }
                y[((size_t)b * T + t) * D + d] = double_swish(s);
            }
    free(g);
    float* out = falloc((size_t)M * D);
    linear(out, D, y, D, WT(m, D, D, "%sconv_module.pointwise_conv2.weight", pfx), W(m, "%sconv_module.pointwise_conv2.bias", pfx), M, D, D);
    add_inplace(src, out, (size_t)M * D);
    free(out);
    free(y);
}

/* taps: 0 = encoder_embed output [B,T',D]; 1+i = output of layer i; -1 = encoder_out */
static int conformer_forward(const k2o_model* m, const float* xin, int B, int T, float* enc_out, int tap, float* tap_out,
                             int64_t tap_cap, int64_t* tap_n) {
    int Tp;
    float* x = conformer_embed(m, xin, B, T, &Tp);
    if (!x) return fail("conformer encoder: T=%d too short", T);
    const int D = m->dim[0], M = B * Tp, L = m->nlayer[0];
    int rc = 0;
    for (int li = 0; li <= L; li++) {
        if (tap == li) {
            int64_t n = (int64_t)M * D;
            if (n > tap_cap) rc = fail("tap buffer too small");
            else { memcpy(tap_out, x, sizeof(float) * n); *tap_n = n; }
            free(x);
            return rc;
        }
        if (li == L) break;
        char pfx[64];
        snprintf(pfx, sizeof pfx, "encoder.encoder.layers.%d.", li);
        float* pe = conformer_pos_emb(Tp, D);
        conformer_ff(m, pfx, "feed_forward_macaron", x, M, D, m->ff[0]);
        conformer_self_attn(m, pfx, x, pe, B, Tp, D, m->heads[0]);
        conformer_conv_module(m, pfx, x, B, Tp, D, m->kern[0]);
        conformer_ff(m, pfx, "feed_forward", x, M, D, m->ff[0]);
        basic_norm(x, x, W(m, "%snorm_final.eps", pfx)[0], M, D);
        free(pe);
    }
    linear(enc_out, m->J, x, D, WT(m, m->J, D, "joiner.encoder_proj.weight"), W(m, "joiner.encoder_proj.bias"), M, D, m->J);
    free(x);
    return 0;
}

/* ------------------------------------------------------------------------------------------------------------------------
 * Streaming Conformer (OnlineProjOfConformer): Conformer.streaming_forward / ConformerEncoderLayer.chunk_forward of the same
 * recipe with causal convolutions.  Per stream and layer the states are cached_attn [left_context, D] -- the layer INPUT to the
 * attention (keys / values are re-projected every chunk) -- and cached_conv [K-1, D] -- the GLU output feeding the causal
 * depthwise conv (OnlineProjOfConformer.cs:55-82 gives the shapes).  One chunk = T input frames -> embed -> drop one frame on
 * each side -> chunk_size frames; keys = [cached_attn ; chunk]; rel-pos table for left + chunk; the oldest
 * (left - processed_lens) cache slots are masked with -inf.  right_context R > 0 (OnlineModel.cs:161-165 reads the key): the chunk
 * carries R more encoder frames, which this step's attention and convolution see; the caches keep the rows in FRONT of them
 * (states[0] = key[-(left + R) : -R], the conv cache likewise) and the R frames are cut from the output -- they come again as
 * the next chunk's first frames.
 * ---------------------------------------------------------------------------------------------------------------------- */
static float* conformer_pos_emb_left(int Tc, int left, int D) {
    const int n2 = left + 2 * Tc - 1;
    float* pe = falloc((size_t)n2 * D);
    const float c = -(logf(10000.0f) / (float)D);
    for (int n = 0; n < n2; n++) {
        float r = (float)(left + Tc - 1 - n);
        for (int k = 0; k < D / 2; k++) {
            float div = expf((float)(2 * k) * c);
            pe[(size_t)n * D + 2 * k] = sinf(r * div);
            pe[(size_t)n * D + 2 * k + 1] = cosf(r * div);
        }
    }
    return pe;
}

/* x [Tc, D] in place; attn_cache [left, D], conv_cache [K-1, D] updated */
static void conformer_layer_stream(const k2o_model* m, int li, float* x, const float* pe, int Tc, int left, int right, int64_t processed,
                                   float* attn_cache, float* conv_cache) {
    const int D = m->dim[0], H = m->heads[0], K = m->kern[0], dk = D / H, KL = left + Tc;
    char pfx[64];
    snprintf(pfx, sizeof pfx, "encoder.encoder.layers.%d.", li);
    conformer_ff(m, pfx, "feed_forward_macaron", x, Tc, D, m->ff[0]);
    {   /* attention over key = [cache ; x] */
        float* key = falloc((size_t)KL * D);
        memcpy(key, attn_cache, sizeof(float) * (size_t)left * D);
        memcpy(key + (size_t)left * D, x, sizeof(float) * (size_t)Tc * D);
        memcpy(attn_cache, key + (size_t)(Tc - right) * D, sizeof(float) * (size_t)left * D);   /* states[0] = key[-(left + right) : -right] (key[-left:] for right = 0) */
        const float* wt = WT(m, 3 * D, D, "%sself_attn.in_proj.weight", pfx);           /* [D, 3D] */
        const float* bias = W(m, "%sself_attn.in_proj.bias", pfx);
        float* q = falloc((size_t)Tc * 3 * D);   /* only the first D columns are used */
        float* kv = falloc((size_t)KL * 3 * D);
        linear(q, 3 * D, x, D, wt, bias, Tc, D, 3 * D);
        linear(kv, 3 * D, key, D, wt, bias, KL, D, 3 * D);
        const int NP = left + 2 * Tc - 1;
        float* p = falloc((size_t)NP * D);
        linear(p, D, pe, D, WT(m, D, D, "%sself_attn.linear_pos.weight", pfx), NULL, NP, D, D);
        const float* bu = W(m, "%sself_attn.pos_bias_u", pfx);
        const float* bv = W(m, "%sself_attn.pos_bias_v", pfx);
        const float scaling = 1.0f / sqrtf((float)dk);
        float* ctxv = falloc((size_t)Tc * D);
        float* sc = falloc((size_t)KL);
        for (int h = 0; h < H; h++)
            for (int i = 0; i < Tc; i++) {
                float qu[256], qv[256];
                for (int d = 0; d < dk; d++) {
                    float qs = q[(size_t)i * 3 * D + h * dk + d] * scaling;
                    qu[d] = qs + bu[h * dk + d];
                    qv[d] = qs + bv[h * dk + d];
                }
                float mx = -INFINITY;
                for (int j = 0; j < KL; j++) {
                    const float* kk = kv + (size_t)j * 3 * D + D + h * dk;
                    const float* pr = p + (size_t)(Tc - 1 - i + j) * D + h * dk;   /* rel_shift(x, left_context) */
                    float ac = 0.f, bd = 0.f;
                    for (int d = 0; d < dk; d++) ac += qu[d] * kk[d];
                    for (int d = 0; d < dk; d++) bd += qv[d] * pr[d];
                    float s = ac + bd;
                    if (j < left && processed <= (int64_t)(left - 1 - j)) s = -INFINITY;   /* key_padding_mask */
                    sc[j] = s;
                    if (s > mx) mx = s;
                }
                float sum = 0.f;
                for (int j = 0; j < KL; j++) { sc[j] = expf(sc[j] - mx); sum += sc[j]; }
                float inv = 1.0f / sum;
                float* o = ctxv + (size_t)i * D + h * dk;
                for (int d = 0; d < dk; d++) o[d] = 0.f;
                for (int j = 0; j < KL; j++) {
                    const float* v = kv + (size_t)j * 3 * D + 2 * D + h * dk;
                    float a = sc[j] * inv;
                    for (int d = 0; d < dk; d++) o[d] += a * v[d];
                }
            }
        float* out = falloc((size_t)Tc * D);
        linear(out, D, ctxv, D, WT(m, D, D, "%sself_attn.out_proj.weight", pfx), W(m, "%sself_attn.out_proj.bias", pfx), Tc, D, D);
        add_inplace(x, out, (size_t)Tc * D);
        free(out); free(sc); free(ctxv); free(p); free(kv); free(q); free(key);
    }
    {   /* causal convolution module with its cache */
        const int lo = K - 1;
        float* x2 = falloc((size_t)Tc * 2 * D);
        linear(x2, 2 * D, x, D, WT(m, 2 * D, D, "%sconv_module.pointwise_conv1.weight", pfx), W(m, "%sconv_module.pointwise_conv1.bias", pfx),
               Tc, D, 2 * D);
        float* g = falloc((size_t)(lo + Tc) * D);
        memcpy(g, conv_cache, sizeof(float) * (size_t)lo * D);
        for (int t = 0; t < Tc; t++)
            for (int d = 0; d < D; d++)
                g[(size_t)(lo + t) * D + d] = x2[(size_t)t * 2 * D + d] * (1.0f / (1.0f + expf(-x2[(size_t)t * 2 * D + D + d])));
        memcpy(conv_cache, g + (size_t)(Tc - right) * D, sizeof(float) * (size_t)lo * D);   /* cache = x[-(lorder + right) : -right] (x[-lorder:] for right = 0) */
        const float* dw = W(m, "%sconv_module.depthwise_conv.weight", pfx);
        const float* db = W(m, "%sconv_module.depthwise_conv.bias", pfx);
        float* y = falloc((size_t)Tc * D);
        for (int t = 0; t < Tc; t++)
            for (int d = 0; d < D; d++) {
                float s = db[d];
                for (int k = 0; k < K; k++) s += dw[d * K + k] * g[(size_t)(t + k) * D + d];
                y[(size_t)t * D + d] = double_swish(s);
            }
        float* out = falloc((size_t)Tc * D);
        linear(out, D, y, D, WT(m, D, D, "%sconv_module.pointwise_conv2.weight", pfx), W(m, "%sconv_module.pointwise_conv2.bias", pfx), Tc, D, D);
        add_inplace(x, out, (size_t)Tc * D);
        free(out); free(y); free(g); free(x2);
    }
    conformer_ff(m, pfx, "feed_forward", x, Tc, D, m->ff[0]);
    basic_norm(x, x, W(m, "%snorm_final.eps", pfx)[0], Tc, D);
}

/* one chunk for one stream: xin [T, 80] -> enc_out [chunk_size, J]; caches [L][left][D], [L][K-1][D] */
static int conformer_stream_chunk(const k2o_model* m, const float* xin, int T, int left, int right, int64_t processed, float* attn_caches,
                                  float* conv_caches, float* enc_out) {
    int T3;
    float* e = conformer_embed(m, xin, 1, T, &T3);
    if (!e || T3 < 3 + right) { free(e); return fail("conformer chunk of %d frames is too short", T); }
    const int D = m->dim[0], Tc = T3 - 2, K = m->kern[0];
    float* x = falloc((size_t)Tc * D);
    memcpy(x, e + D, sizeof(float) * (size_t)Tc * D);   /* embed[:, 1:-1] */
    free(e);
    float* pe = conformer_pos_emb_left(Tc, left, D);
    for (int li = 0; li < m->nlayer[0]; li++)
        conformer_layer_stream(m, li, x, pe, Tc, left, right, processed, attn_caches + (size_t)li * left * D, conv_caches + (size_t)li * (K - 1) * D);
    free(pe);
    /* x = x[:-right_context]: the right-context frames are not output */
    linear(enc_out, m->J, x, D, WT(m, m->J, D, "joiner.encoder_proj.weight"), W(m, "joiner.encoder_proj.bias"), Tc - right, D, m->J);
    free(x);
    return Tc - right;
}
