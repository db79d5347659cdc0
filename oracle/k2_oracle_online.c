/*
 * k2_oracle_online.c -- CPU restatement of the streaming (OnlineRecognizer) hot path.
 * TEST INFRASTRUCTURE ONLY; PARITY UNPINNED (see k2_oracle.h).  Included at the end of
 * k2_oracle.c (shares its static helpers).
 *
 * From the reference (file:line in K2TransducerAsr/):
 *   - state inventory and shapes: OnlineProjOfZipformer2.GetEncoderInitStates :63-111
 *     (per layer cached_key[left,B,32H], cached_nonlin_attn[1,B,left,3D/4], cached_val1/2[left,B,12H],
 *      cached_conv1/2[B,D,K/2]; embed_states[B,128,3,19]; processed_lens[B])
 *   - chunking: ChunkLength = T, ShiftLength = decode_chunk_len (OnlineModel.cs:48-49)
 *   - online PadSequence (tail 0) still maps every 0.0 to the log floor (PadHelper.cs:9-13,58)
 *   - greedy loop: OnlineRecognizer.ForwardBatchGreedySearch :85-219 (emit filter skips 0, 2 AND 1, :181;
 *     hyp starts [blank, blank], OnlineStream.cs:44; timestamps are chunk-relative frame indices :184)
 *   - Q11: for B > 1 the reference's stack_states scrambles cached_nonlin_attn across streams
 *     (OnlineProjOfZipformer2.cs:254-262 vs :407-413); the oracle keeps every stream's state separate,
 *     i.e. reproduces the reference at B = 1 per stream.
 * From the published icefall streaming graph (zipformer.py *.streaming_forward, export-onnx-streaming.py):
 *   Conv2dSubsampling/ConvNeXt with a 3-frame left cache, processed_lens key mask over the left
 *   context, cached keys/values/nonlin inputs, ChunkCausalDepthwiseConv1d with its cache.
 */

struct k2o_online_stream {
    int nl;           /* total layers */
    float **key, **nonlin, **val1, **val2, **conv1, **conv2;
    size_t *nkey, *nnonlin, *nval, *nconv;
    float* embed;     /* [128][3][19] */
    int64_t processed_len;
    float *conf_attn, *conf_conv; /* streaming conformer: cached_attn [L][left][D], cached_conv [L][K-1][D] */
    float **avg; float* clen; size_t* navg; /* model_type zipformer (v1): cached_avg [D] and cached_len per layer; key / val1 / val2 /
                                               conv1 / conv2 above hold cached_key / cached_val / cached_val2 / cached_conv1 / cached_conv2 */
    float *lstm_h, *lstm_c; /* model_type lstm: h [layers][d_model], c [layers][rnn_hidden] (OnlineProjOfLstm.cs:55-75) */
    int64_t hyp[2];
    int64_t* tokens;  /* Tokens list, starts [blank, blank] (OnlineStream.cs:45) */
    int n_tokens, cap_tokens;
    int32_t* timestamps;
    int n_ts, cap_ts;
};

static int online_left(const k2o_model* m, int si) {
    int v[MAX_STACKS] = {0};
    parse_csv(k2o_meta(m, "left_context_len"), v, MAX_STACKS);
    return v[si];
}
int k2o_online_chunk_length(const k2o_model* m) { return meta_int(m, "T", 45); }
int k2o_online_shift_length(const k2o_model* m) { return meta_int(m, "decode_chunk_len", 32); }
int k2o_online_frames_per_chunk(const k2o_model* m) {
    if (m->zip1) return ((meta_int(m, "T", 39) - 7) / 2 + 1) / 2;
    if (m->lstm) return lstm_out_frames(meta_int(m, "T", 9));
    if (m->conformer) return meta_int(m, "chunk_size", 16);
    return (meta_int(m, "decode_chunk_len", 32) / 2 + 1) / 2;
}

k2o_online_stream* k2o_online_stream_create(const k2o_model* m) {
    const char* st = k2o_meta(m, "streaming");
    if (!m->lstm && (!st || strcmp(st, "1"))) {
        fail("model is not a streaming export (metadata streaming != 1)");
        return NULL;
    }
    k2o_online_stream* s = (k2o_online_stream*)calloc(1, sizeof *s);
    if (m->conformer) {
        /* OnlineProjOfConformer.GetEncoderInitStates (:55-82): zero caches and processed_lens[0] = 2 (sic) */
        const int left = meta_int(m, "left_context", 64), D = m->dim[0], K = m->kern[0];
        s->nl = m->nlayer[0];
        s->conf_attn = calloc((size_t)s->nl * left * D, sizeof(float));
        s->conf_conv = calloc((size_t)s->nl * (K - 1) * D, sizeof(float));
        s->processed_len = 2;
        s->cap_tokens = 64;
        s->tokens = malloc(sizeof(int64_t) * s->cap_tokens);
        s->tokens[0] = s->tokens[1] = 0;
        s->n_tokens = 2;
        s->cap_ts = 64;
        s->timestamps = malloc(sizeof(int32_t) * s->cap_ts);
        return s;
    }
    if (m->lstm) {  /* GetEncoderInitStates: zero h and c */
        s->nl = m->nlayer[0];
        s->lstm_h = calloc((size_t)s->nl * m->dim[0], sizeof(float));
        s->lstm_c = calloc((size_t)s->nl * m->rnn_hidden, sizeof(float));
        s->cap_tokens = 64;
        s->tokens = malloc(sizeof(int64_t) * s->cap_tokens);
        s->tokens[0] = s->tokens[1] = 0;
        s->n_tokens = 2;
        s->cap_ts = 64;
        s->timestamps = malloc(sizeof(int32_t) * s->cap_ts);
        return s;
    }
    for (int i = 0; i < m->ns; i++) s->nl += m->nlayer[i];
    if (m->zip1) {   /* OnlineProjOfZipformer.GetEncoderInitStates (:56-111), B = 1, every cache zero */
        s->key = calloc(s->nl, sizeof(float*)); s->val1 = calloc(s->nl, sizeof(float*)); s->val2 = calloc(s->nl, sizeof(float*));
        s->conv1 = calloc(s->nl, sizeof(float*)); s->conv2 = calloc(s->nl, sizeof(float*)); s->avg = calloc(s->nl, sizeof(float*));
        s->nonlin = calloc(s->nl, sizeof(float*));
        s->nkey = calloc(s->nl, sizeof(size_t)); s->nval = calloc(s->nl, sizeof(size_t)); s->nconv = calloc(s->nl, sizeof(size_t));
        s->navg = calloc(s->nl, sizeof(size_t)); s->nnonlin = calloc(s->nl, sizeof(size_t));
        s->clen = calloc(s->nl, sizeof(float));
        int l = 0;
        for (int si = 0; si < m->ns; si++)
            for (int li = 0; li < m->nlayer[si]; li++, l++) {
                int L = online_left(m, si);
                s->nkey[l] = (size_t)L * m->att[si];
                s->nval[l] = (size_t)L * (m->att[si] / 2);
                s->nconv[l] = (size_t)m->dim[si] * (m->kern[si] - 1);
                s->navg[l] = (size_t)m->dim[si];
                s->key[l] = calloc(s->nkey[l], sizeof(float));
                s->val1[l] = calloc(s->nval[l], sizeof(float));
                s->val2[l] = calloc(s->nval[l], sizeof(float));
                s->conv1[l] = calloc(s->nconv[l], sizeof(float));
                s->conv2[l] = calloc(s->nconv[l], sizeof(float));
                s->avg[l] = calloc(s->navg[l], sizeof(float));
            }
        s->cap_tokens = 64;
        s->tokens = malloc(sizeof(int64_t) * s->cap_tokens);
        s->tokens[0] = s->tokens[1] = 0;
        s->n_tokens = 2;
        s->cap_ts = 64;
        s->timestamps = malloc(sizeof(int32_t) * s->cap_ts);
        return s;
    }
    s->key = calloc(s->nl, sizeof(float*)); s->nonlin = calloc(s->nl, sizeof(float*));
    s->val1 = calloc(s->nl, sizeof(float*)); s->val2 = calloc(s->nl, sizeof(float*));
    s->conv1 = calloc(s->nl, sizeof(float*)); s->conv2 = calloc(s->nl, sizeof(float*));
    s->nkey = calloc(s->nl, sizeof(size_t)); s->nnonlin = calloc(s->nl, sizeof(size_t));
    s->nval = calloc(s->nl, sizeof(size_t)); s->nconv = calloc(s->nl, sizeof(size_t));
    int l = 0;
    for (int si = 0; si < m->ns; si++) {
        int L = online_left(m, si), D = m->dim[si], H = m->heads[si];
        for (int li = 0; li < m->nlayer[si]; li++, l++) {
            s->nkey[l] = (size_t)L * m->qhd[si] * H;
            s->nnonlin[l] = (size_t)L * (3 * D / 4);
            s->nval[l] = (size_t)L * m->vhd[si] * H;
            s->nconv[l] = (size_t)D * (m->kern[si] / 2);
            s->key[l] = calloc(s->nkey[l], sizeof(float));
            s->nonlin[l] = calloc(s->nnonlin[l], sizeof(float));
            s->val1[l] = calloc(s->nval[l], sizeof(float));
            s->val2[l] = calloc(s->nval[l], sizeof(float));
            s->conv1[l] = calloc(s->nconv[l], sizeof(float));
            s->conv2[l] = calloc(s->nconv[l], sizeof(float));
        }
    }
    s->embed = calloc(128 * 3 * 19, sizeof(float));
    s->hyp[0] = s->hyp[1] = 0;
    s->cap_tokens = 64;
    s->tokens = malloc(sizeof(int64_t) * s->cap_tokens);
    s->tokens[0] = s->tokens[1] = 0;
    s->n_tokens = 2;
    s->cap_ts = 64;
    s->timestamps = malloc(sizeof(int32_t) * s->cap_ts);
    return s;
}
void k2o_online_stream_free(k2o_online_stream* s) {
    if (!s) return;
    if (s->conf_attn) {
        free(s->conf_attn); free(s->conf_conv); free(s->tokens); free(s->timestamps);
        free(s);
        return;
    }
    if (s->lstm_h) {
        free(s->lstm_h); free(s->lstm_c); free(s->tokens); free(s->timestamps);
        free(s);
        return;
    }
    if (s->avg) {
        for (int l = 0; l < s->nl; l++) free(s->avg[l]);
        free(s->avg); free(s->navg); free(s->clen);
    }
    for (int l = 0; l < s->nl; l++) {
        free(s->key[l]); free(s->nonlin[l]); free(s->val1[l]); free(s->val2[l]); free(s->conv1[l]); free(s->conv2[l]);
    }
    free(s->key); free(s->nonlin); free(s->val1); free(s->val2); free(s->conv1); free(s->conv2);
    free(s->nkey); free(s->nnonlin); free(s->nval); free(s->nconv);
    free(s->embed); free(s->tokens); free(s->timestamps);
    free(s);
}
int k2o_online_stream_num_layers(const k2o_online_stream* s) { return s->nl; }
int64_t k2o_online_stream_processed_len(const k2o_online_stream* s) { return s->processed_len; }
/* kind: 0 key, 1 nonlin, 2 val1, 3 val2, 4 conv1, 5 conv2, 6 embed (layer ignored);
 * zipformer (v1) streams: 0 cached_key, 1 cached_avg, 2 cached_val, 3 cached_val2, 4 cached_conv1, 5 cached_conv2, 7 cached_len (1 float) */
int64_t k2o_online_stream_state(const k2o_online_stream* s, int layer, int kind, float* out, int64_t cap) {
    const float* p = NULL;
    size_t n = 0;
    if (s->lstm_h || s->conf_attn) {
        return fail("use k2o_online_stream_lstm_state for an lstm stream");
    }
    if (layer < 0 || layer >= s->nl) return fail("bad layer %d", layer);
    if (s->avg && (kind == 1 || kind == 6 || kind == 7)) {
        if (kind == 6) return fail("a zipformer (v1) stream has no embed state");
        p = kind == 1 ? s->avg[layer] : s->clen + layer;
        n = kind == 1 ? s->navg[layer] : 1;
        if (!out) return (int64_t)n;
        if ((int64_t)n > cap) return fail("state buffer too small");
        memcpy(out, p, sizeof(float) * n);
        return (int64_t)n;
    }
    switch (kind) {
        case 0: p = s->key[layer]; n = s->nkey[layer]; break;
        case 1: p = s->nonlin[layer]; n = s->nnonlin[layer]; break;
        case 2: p = s->val1[layer]; n = s->nval[layer]; break;
        case 3: p = s->val2[layer]; n = s->nval[layer]; break;
        case 4: p = s->conv1[layer]; n = s->nconv[layer]; break;
        case 5: p = s->conv2[layer]; n = s->nconv[layer]; break;
        case 6: p = s->embed; n = 128 * 3 * 19; break;
        default: return fail("bad state kind %d", kind);
    }
    if (!out) return (int64_t)n;
    if ((int64_t)n > cap) return fail("state buffer too small");
    memcpy(out, p, sizeof(float) * n);
    return (int64_t)n;
}
int k2o_online_stream_num_tokens(const k2o_online_stream* s) { return s->n_tokens; }
int k2o_online_stream_num_timestamps(const k2o_online_stream* s) { return s->n_ts; }
void k2o_online_stream_get_tokens(const k2o_online_stream* s, int64_t* out) { memcpy(out, s->tokens, sizeof(int64_t) * s->n_tokens); }
void k2o_online_stream_get_timestamps(const k2o_online_stream* s, int32_t* out) { memcpy(out, s->timestamps, sizeof(int32_t) * s->n_ts); }
void k2o_online_stream_get_hyp(const k2o_online_stream* s, int64_t* out) { out[0] = s->hyp[0]; out[1] = s->hyp[1]; }

/* CompactRelPositionalEncoding.forward(x, left_context_len): row n <-> relative position n - (Tc+L-1) */
static float* compact_rel_pos_stream(int Tc, int L, int pos_dim) {
    int n2 = 2 * Tc - 1 + L;
    float* pe = falloc((size_t)n2 * pos_dim);
    float cl = sqrtf((float)pos_dim), ls = (float)pos_dim / (2.0f * (float)M_PI), logcl = logf(cl);
    for (int n = 0; n < n2; n++) {
        float x = (float)(n - (Tc + L - 1));
        float sgn = (x > 0.f) - (x < 0.f);
        float xa = atanf(cl * sgn * (logf(fabsf(x) + cl) - logcl) / ls);
        for (int k = 0; k < pos_dim / 2; k++) {
            pe[(size_t)n * pos_dim + 2 * k] = cosf(xa * (float)(k + 1));
            pe[(size_t)n * pos_dim + 2 * k + 1] = sinf(xa * (float)(k + 1));
        }
        pe[(size_t)n * pos_dim + pos_dim - 1] = 1.0f;
    }
    return pe;
}

/* Conv2dSubsampling.streaming_forward for one stream: x [T,80] -> [Tc, D0]; cache [128][3][19] updated */
static float* online_embed(const k2o_model* m, const float* x, int T, float* cache, int* Tc_out) {
    int F0 = m->feat, T1 = T - 2, T2 = (T1 - 3) / 2 + 1, F2 = (F0 - 3) / 2 + 1, T3 = T2 - 2, F3 = (F2 - 3) / 2 + 1;
    int Tc = T3 - 3;
    const float *w0 = W(m, "encoder_embed.conv.0.weight"), *b0 = W(m, "encoder_embed.conv.0.bias");
    float* a1 = falloc((size_t)T1 * F0 * 8);
    for (int t = 0; t < T1; t++)
        for (int f = 0; f < F0; f++)
            for (int co = 0; co < 8; co++) {
                float s = b0[co];
                for (int kt = 0; kt < 3; kt++)
                    for (int kf = 0; kf < 3; kf++) {
                        int ff = f + kf - 1;
                        if (ff < 0 || ff >= F0) continue;
                        s += w0[(co * 3 + kt) * 3 + kf] * x[(size_t)(t + kt) * F0 + ff];
                    }
                a1[((size_t)t * F0 + f) * 8 + co] = swoosh_r(s);
            }
    const float *w4 = W(m, "encoder_embed.conv.4.weight"), *b4 = W(m, "encoder_embed.conv.4.bias");
    float* a2 = falloc((size_t)T2 * F2 * 32);
    for (int t = 0; t < T2; t++)
        for (int f = 0; f < F2; f++)
            for (int co = 0; co < 32; co++) {
                float s = b4[co];
                for (int kt = 0; kt < 3; kt++)
                    for (int kf = 0; kf < 3; kf++) {
                        const float* xi = a1 + ((size_t)(2 * t + kt) * F0 + 2 * f + kf) * 8;
                        for (int ci = 0; ci < 8; ci++) s += w4[((co * 8 + ci) * 3 + kt) * 3 + kf] * xi[ci];
                    }
                a2[((size_t)t * F2 + f) * 32 + co] = swoosh_r(s);
            }
    free(a1);
    const float *w7 = W(m, "encoder_embed.conv.7.weight"), *b7 = W(m, "encoder_embed.conv.7.bias");
    float* a3 = falloc((size_t)T3 * F3 * 128);
    for (int t = 0; t < T3; t++)
        for (int f = 0; f < F3; f++)
            for (int co = 0; co < 128; co++) {
                float s = b7[co];
                for (int kt = 0; kt < 3; kt++)
                    for (int kf = 0; kf < 3; kf++) {
                        const float* xi = a2 + ((size_t)(t + kt) * F2 + 2 * f + kf) * 32;
                        for (int ci = 0; ci < 32; ci++) s += w7[((co * 32 + ci) * 3 + kt) * 3 + kf] * xi[ci];
                    }
                a3[((size_t)t * F3 + f) * 128 + co] = swoosh_r(s);
            }
    free(a2);
    /* ConvNeXt.streaming_forward: cat = [cached_left_pad(3) ; x(T3)], valid conv over time, freq pad 3 */
    int Tcat = T3 + 3;
    float* cat = falloc((size_t)Tcat * F3 * 128);
    for (int t = 0; t < 3; t++)
        for (int f = 0; f < F3; f++)
            for (int c = 0; c < 128; c++) cat[((size_t)t * F3 + f) * 128 + c] = cache[(c * 3 + t) * F3 + f];
    memcpy(cat + (size_t)3 * F3 * 128, a3, sizeof(float) * (size_t)T3 * F3 * 128);
    for (int t = 0; t < 3; t++)   /* cached_left_pad = cat[:, :, Tc : Tc+3] */
        for (int f = 0; f < F3; f++)
            for (int c = 0; c < 128; c++) cache[(c * 3 + t) * F3 + f] = cat[((size_t)(Tc + t) * F3 + f) * 128 + c];
    const float *wd = W(m, "encoder_embed.convnext.depthwise_conv.weight"), *bd = W(m, "encoder_embed.convnext.depthwise_conv.bias");
    int npix = Tc * F3;
    float* dw = falloc((size_t)npix * 128);
    for (int t = 0; t < Tc; t++)
        for (int f = 0; f < F3; f++)
            for (int c = 0; c < 128; c++) {
                float s = bd[c];
                for (int kt = 0; kt < 7; kt++)
                    for (int kf = 0; kf < 7; kf++) {
                        int ff = f + kf - 3;
                        if (ff < 0 || ff >= F3) continue;
                        s += wd[(c * 7 + kt) * 7 + kf] * cat[((size_t)(t + kt) * F3 + ff) * 128 + c];
                    }
                dw[((size_t)t * F3 + f) * 128 + c] = s;
            }
    free(cat);
    float* h = falloc((size_t)npix * 384);
    linear(h, 384, dw, 128, WT(m, 384, 128, "encoder_embed.convnext.pointwise_conv1.weight"),
           W(m, "encoder_embed.convnext.pointwise_conv1.bias"), npix, 128, 384);
    apply_swoosh_l(h, (size_t)npix * 384);
    linear(dw, 128, h, 384, WT(m, 128, 384, "encoder_embed.convnext.pointwise_conv2.weight"),
           W(m, "encoder_embed.convnext.pointwise_conv2.bias"), npix, 384, 128);
    free(h);
    for (size_t i = 0; i < (size_t)npix * 128; i++) dw[i] += a3[i]; /* bypass = x[:, :, :Tc] */
    free(a3);
    int D0 = m->dim[0], KK = 128 * F3;
    float* flat = falloc((size_t)Tc * KK);
    for (int t = 0; t < Tc; t++)
        for (int f = 0; f < F3; f++)
            for (int c = 0; c < 128; c++) flat[(size_t)t * KK + c * F3 + f] = dw[((size_t)t * F3 + f) * 128 + c];
    free(dw);
    float* lin = falloc((size_t)Tc * D0);
    linear(lin, D0, flat, KK, WT(m, D0, KK, "encoder_embed.out.weight"), W(m, "encoder_embed.out.bias"), Tc, KK, D0);
    free(flat);
    float* out = falloc((size_t)Tc * D0);
    biasnorm(out, lin, W(m, "encoder_embed.out_norm.bias"), W(m, "encoder_embed.out_norm.log_scale")[0], Tc, D0);
    free(lin);
    *Tc_out = Tc;
    return out;
}

/* new_cache = concat(cache[L], x[Tc])[-L:]  (rows of width w) ; returns the concatenation */
static float* cat_and_shift(float* cache, const float* x, int L, int Tc, int w) {
    float* cat = falloc((size_t)(L + Tc) * w);
    memcpy(cat, cache, sizeof(float) * (size_t)L * w);
    memcpy(cat + (size_t)L * w, x, sizeof(float) * (size_t)Tc * w);
    memcpy(cache, cat + (size_t)Tc * w, sizeof(float) * (size_t)L * w);
    return cat;
}

/* out[i, c0+c] = sum_j aw[i,j] * v[j, c0+c] ; aw [Tc, KL], v rows have stride ldv */
static void attn_apply1(float* out, int ldo, const float* aw, const float* v, int ldv, int Tc, int KL, int c0, int nc) {
    for (int i = 0; i < Tc; i++)
        for (int c = 0; c < nc; c++) {
            float s = 0.f;
            for (int j = 0; j < KL; j++) s += aw[(size_t)i * KL + j] * v[(size_t)j * ldv + c0 + c];
            out[(size_t)i * ldo + c0 + c] = s;
        }
}

/* ConvolutionModule.streaming_forward; cache [D][K/2] */
static void online_conv_module(const k2o_model* m, int si, const char* pfx, int k, float* src, int Tc, float* cache) {
    int D = m->dim[si], K = m->kern[si], pad = K / 2, Kc = (K + 1) / 2;
    float* x = falloc((size_t)Tc * 2 * D);
    linear(x, 2 * D, src, D, WT(m, 2 * D, D, "%sconv_module%d.in_proj.weight", pfx, k), W(m, "%sconv_module%d.in_proj.bias", pfx, k), Tc, D, 2 * D);
    /* GLU, then [cache ; chunk] per channel */
    float* cat = falloc((size_t)(pad + Tc) * D); /* [t][d] */
    for (int t = 0; t < pad; t++)
        for (int d = 0; d < D; d++) cat[(size_t)t * D + d] = cache[(size_t)d * pad + t];
    for (int t = 0; t < Tc; t++)
        for (int d = 0; d < D; d++)
            cat[(size_t)(pad + t) * D + d] = x[(size_t)t * 2 * D + d] * (1.0f / (1.0f + expf(-x[(size_t)t * 2 * D + D + d])));
    for (int t = 0; t < pad; t++)   /* cache = cat[..., -left_pad:] */
        for (int d = 0; d < D; d++) cache[(size_t)d * pad + t] = cat[(size_t)(Tc + t) * D + d];
    const float* wc = W(m, "%sconv_module%d.depthwise_conv.causal_conv.weight", pfx, k);   /* [D,1,Kc] */
    const float* bc = W(m, "%sconv_module%d.depthwise_conv.causal_conv.bias", pfx, k);
    const float* ww = W(m, "%sconv_module%d.depthwise_conv.chunkwise_conv.weight", pfx, k); /* [D,1,K] */
    const float* bw = W(m, "%sconv_module%d.depthwise_conv.chunkwise_conv.bias", pfx, k);
    const float* sc = W(m, "%sconv_module%d.depthwise_conv.chunkwise_conv_scale", pfx, k);  /* [2,D,K] */
    float* c1 = falloc((size_t)Tc * D);
    for (int t = 0; t < Tc; t++)
        for (int d = 0; d < D; d++) {
            /* causal conv over [cache ; chunk], valid: output t uses cat[t .. t+Kc-1] */
            float xc = bc[d];
            for (int kk = 0; kk < Kc; kk++) xc += wc[d * Kc + kk] * cat[(size_t)(t + kk) * D + d];
            /* chunkwise conv: zero-padded K/2 each side WITHIN the chunk */
            float xw = bw[d];
            for (int kk = 0; kk < K; kk++) {
                int tt = t + kk - pad;
                if (tt < 0 || tt >= Tc) continue;
                xw += ww[d * K + kk] * cat[(size_t)(pad + tt) * D + d];
            }
            /* _get_chunk_scale */
            float le, re;
            if (Tc < K) {
                le = sc[(size_t)d * K + t];
                re = sc[(size_t)D * K + (size_t)d * K + (K - Tc) + t];
            } else {
                le = t < K ? sc[(size_t)d * K + t] : 0.f;
                re = t >= Tc - K ? sc[(size_t)D * K + (size_t)d * K + (t - (Tc - K))] : 0.f;
            }
            float v = xw * (1.0f + (le + re)) + xc;
            c1[(size_t)t * D + d] = swoosh_r(v);
        }
    float* o = falloc((size_t)Tc * D);
    linear(o, D, c1, D, WT(m, D, D, "%sconv_module%d.out_proj.weight", pfx, k), W(m, "%sconv_module%d.out_proj.bias", pfx, k), Tc, D, D);
    add_inplace(src, o, (size_t)Tc * D);
    free(x); free(cat); free(c1); free(o);
}

static void online_self_attn(const k2o_model* m, int si, const char* pfx, int k, float* src, const float* aw, int Tc, int L, float* cache) {
    int D = m->dim[si], H = m->heads[si], v = m->vhd[si], HV = H * v, KL = L + Tc;
    float* x = falloc((size_t)Tc * HV);
    linear(x, HV, src, D, WT(m, HV, D, "%sself_attn%d.in_proj.weight", pfx, k), W(m, "%sself_attn%d.in_proj.bias", pfx, k), Tc, D, HV);
    float* cat = cat_and_shift(cache, x, L, Tc, HV);
    float* a = falloc((size_t)Tc * HV);
    for (int h = 0; h < H; h++) attn_apply1(a, HV, aw + (size_t)h * Tc * KL, cat, HV, Tc, KL, h * v, v);
    float* o = falloc((size_t)Tc * D);
    linear(o, D, a, HV, WT(m, D, HV, "%sself_attn%d.out_proj.weight", pfx, k), W(m, "%sself_attn%d.out_proj.bias", pfx, k), Tc, HV, D);
    add_inplace(src, o, (size_t)Tc * D);
    free(x); free(cat); free(a); free(o);
}

/* Zipformer2EncoderLayer.streaming_forward, one stream, in place on src [Tc, D]; l = global layer index */
static void online_layer(const k2o_model* m, k2o_online_stream* s, int si, int li, int l, float* src, const float* pe, int Tc,
                         int L, const unsigned char* key_mask /* [L+Tc] */) {
    char pfx[128];
    snprintf(pfx, sizeof pfx, "encoder.encoders.%d.layers.%d.", si, li);
    int D = m->dim[si], F = m->ff[si], H = m->heads[si], q = m->qhd[si], p = m->phd[si];
    int inproj = (2 * q + p) * H, KL = L + Tc, n2 = 2 * Tc - 1 + L, Hc = 3 * D / 4;
    float* orig = falloc((size_t)Tc * D);
    memcpy(orig, src, sizeof(float) * (size_t)Tc * D);
    /* attention weights with cached keys */
    float* x = falloc((size_t)Tc * inproj);
    linear(x, inproj, src, D, WT(m, inproj, D, "%sself_attn_weights.in_proj.weight", pfx), W(m, "%sself_attn_weights.in_proj.bias", pfx), Tc, D, inproj);
    float* knew = falloc((size_t)Tc * q * H);
    for (int t = 0; t < Tc; t++) memcpy(knew + (size_t)t * q * H, x + (size_t)t * inproj + q * H, sizeof(float) * q * H);
    float* kcat = cat_and_shift(s->key[l], knew, L, Tc, q * H);
    free(knew);
    float* pp = falloc((size_t)n2 * p * H);
    linear(pp, p * H, pe, m->pos_dim, WT(m, p * H, m->pos_dim, "%sself_attn_weights.linear_pos.weight", pfx), NULL, n2, m->pos_dim, p * H);
    float* aw = falloc((size_t)H * Tc * KL);
    for (int h = 0; h < H; h++)
        for (int i = 0; i < Tc; i++) {
            const float* qi = x + (size_t)i * inproj + h * q;
            const float* pi = x + (size_t)i * inproj + 2 * q * H + h * p;
            float* row = aw + ((size_t)h * Tc + i) * KL;
            float mx = -INFINITY;
            for (int j = 0; j < KL; j++) {
                const float* kj = kcat + (size_t)j * q * H + h * q;
                float sc = 0.f;
                for (int d = 0; d < q; d++) sc += qi[d] * kj[d];
                const float* pr = pp + (size_t)(Tc - 1 - i + j) * (p * H) + h * p;
                float ps = 0.f;
                for (int c = 0; c < p; c++) ps += pi[c] * pr[c];
                sc += ps;
                if (key_mask[j]) sc = -1000.0f;  /* masked_fill(key_padding_mask, -1000) */
                row[j] = sc;
                if (sc > mx) mx = sc;
            }
            float sum = 0.f;
            for (int j = 0; j < KL; j++) { row[j] = expf(row[j] - mx); sum += row[j]; }
            float inv = 1.0f / sum;
            for (int j = 0; j < KL; j++) row[j] *= inv;
        }
    free(x); free(kcat); free(pp);

    feed_forward(m, pfx, 1, src, Tc, D, F * 3 / 4);
    {   /* NonlinAttention.streaming_forward */
        float* y = falloc((size_t)Tc * 3 * Hc);
        linear(y, 3 * Hc, src, D, WT(m, 3 * Hc, D, "%snonlin_attention.in_proj.weight", pfx), W(m, "%snonlin_attention.in_proj.bias", pfx), Tc, D, 3 * Hc);
        float* g = falloc((size_t)Tc * Hc);
        for (int t = 0; t < Tc; t++)
            for (int c = 0; c < Hc; c++) g[(size_t)t * Hc + c] = y[(size_t)t * 3 * Hc + Hc + c] * tanhf(y[(size_t)t * 3 * Hc + c]);
        float* cat = cat_and_shift(s->nonlin[l], g, L, Tc, Hc);
        float* a = falloc((size_t)Tc * Hc);
        attn_apply1(a, Hc, aw, cat, Hc, Tc, KL, 0, Hc);
        for (int t = 0; t < Tc; t++)
            for (int c = 0; c < Hc; c++) a[(size_t)t * Hc + c] *= y[(size_t)t * 3 * Hc + 2 * Hc + c];
        float* o = falloc((size_t)Tc * D);
        linear(o, D, a, Hc, WT(m, D, Hc, "%snonlin_attention.out_proj.weight", pfx), W(m, "%snonlin_attention.out_proj.bias", pfx), Tc, Hc, D);
        add_inplace(src, o, (size_t)Tc * D);
        free(y); free(g); free(cat); free(a); free(o);
    }
    online_self_attn(m, si, pfx, 1, src, aw, Tc, L, s->val1[l]);
    online_conv_module(m, si, pfx, 1, src, Tc, s->conv1[l]);
    feed_forward(m, pfx, 2, src, Tc, D, F);
    bypass(src, orig, src, W(m, "%sbypass_mid.bypass_scale", pfx), Tc, D);
    online_self_attn(m, si, pfx, 2, src, aw, Tc, L, s->val2[l]);
    online_conv_module(m, si, pfx, 2, src, Tc, s->conv2[l]);
    feed_forward(m, pfx, 3, src, Tc, D, F * 5 / 4);
    float* nm = falloc((size_t)Tc * D);
    biasnorm(nm, src, W(m, "%snorm.bias", pfx), W(m, "%snorm.log_scale", pfx)[0], Tc, D);
    bypass(src, orig, nm, W(m, "%sbypass.bypass_scale", pfx), Tc, D);
    free(nm); free(aw); free(orig);
}

#include "k2_oracle_zipformer1.c"

/* OnnxEncoder.forward of the streaming export for ONE stream: x [T,80] (already log-floored) -> enc_out [T'c, J] */
static int lstm_online_chunk(const k2o_model* m, k2o_online_stream* s, const float* x, float* enc_out);
int k2o_online_encoder_chunk(const k2o_model* m, k2o_online_stream* s, const float* x, float* enc_out) {
    if (m->lstm) return lstm_online_chunk(m, s, x, enc_out);
    if (m->zip1) return z1_stream_chunk(m, s, x, enc_out);
    if (m->conformer) {   /* model semantics: new processed_lens = processed_lens + chunk frames */
        const int left = meta_int(m, "left_context", 64);
        int tc = conformer_stream_chunk(m, x, k2o_online_chunk_length(m), left, meta_int(m, "right_context", 0), s->processed_len, s->conf_attn,
                                        s->conf_conv, enc_out);
        if (tc > 0) s->processed_len += tc;
        return tc;
    }
    int T = k2o_online_chunk_length(m);
    int Tc;
    float* cur = online_embed(m, x, T, s->embed, &Tc);
    int left50 = online_left(m, 0) * m->ds[0];
    /* processed_mask over the 50 Hz left context: cache slot j (0 = oldest) is masked iff processed_len <= left-1-j */
    unsigned char* mask50 = (unsigned char*)calloc(left50 + Tc, 1);
    for (int j = 0; j < left50; j++) mask50[j] = (s->processed_len <= (int64_t)(left50 - 1 - j));
    float* outputs[MAX_STACKS] = {0};
    int Dcur = m->dim[0], l = 0;
    for (int si = 0; si < m->ns; si++) {
        int D = m->dim[si], ds = m->ds[si], L = online_left(m, si);
        float* xi = convert_channels(cur, Tc, Dcur, D);
        free(cur);
        Dcur = D;
        /* src_key_padding_mask[..., ::ds] */
        int nm = (left50 + Tc + ds - 1) / ds;
        unsigned char* mask = (unsigned char*)malloc(nm);
        for (int j = 0; j < nm; j++) mask[j] = mask50[j * ds];
        if (ds == 1) {
            float* pe = compact_rel_pos_stream(Tc, L, m->pos_dim);
            for (int li = 0; li < m->nlayer[si]; li++, l++) online_layer(m, s, si, li, l, xi, pe, Tc, L, mask);
            free(pe);
            cur = xi;
        } else {
            int Td;
            float* xd = simple_downsample(xi, W(m, "encoder.encoders.%d.downsample.bias", si), ds, 1, Tc, D, &Td);
            if (L + Td != nm) { free(mask); free(mask50); return fail("internal: mask length %d != %d", nm, L + Td); }
            float* pe = compact_rel_pos_stream(Td, L, m->pos_dim);
            for (int li = 0; li < m->nlayer[si]; li++, l++) online_layer(m, s, si, li, l, xd, pe, Td, L, mask);
            free(pe);
            const float* sc = W(m, "encoder.encoders.%d.out_combiner.bypass_scale", si);
            float* y = falloc((size_t)Tc * D);
            for (int t = 0; t < Tc; t++)
                for (int d = 0; d < D; d++) {
                    float o = xi[(size_t)t * D + d], u = xd[(size_t)(t / ds) * D + d];
                    y[(size_t)t * D + d] = o + (u - o) * sc[d];
                }
            free(xd); free(xi);
            cur = y;
        }
        free(mask);
        outputs[si] = falloc((size_t)Tc * D);
        memcpy(outputs[si], cur, sizeof(float) * (size_t)Tc * D);
    }
    free(cur);
    free(mask50);
    int Dmax = m->dmax;
    float* full = falloc((size_t)Tc * Dmax);
    {
        int c = m->dim[m->ns - 1];
        for (int r = 0; r < Tc; r++) memcpy(full + (size_t)r * Dmax, outputs[m->ns - 1] + (size_t)r * c, sizeof(float) * c);
        for (int i = m->ns - 2; i >= 0; i--) {
            int d = m->dim[i];
            if (d > c) {
                for (int r = 0; r < Tc; r++) memcpy(full + (size_t)r * Dmax + c, outputs[i] + (size_t)r * d + c, sizeof(float) * (d - c));
                c = d;
            }
        }
    }
    for (int i = 0; i < m->ns; i++) free(outputs[i]);
    int Tp;
    float* dsd = simple_downsample(full, W(m, "encoder.downsample_output.bias"), 2, 1, Tc, Dmax, &Tp);
    free(full);
    if (m->ctc) ctc_head(m, dsd, Tp, enc_out);
    else linear(enc_out, m->J, dsd, Dmax, WT(m, m->J, Dmax, "joiner.encoder_proj.weight"), W(m, "joiner.encoder_proj.bias"), Tp, Dmax, m->J);
    free(dsd);
    s->processed_len += Tc; /* new_processed_lens = processed_lens + x_lens, x_lens = (T-7)//2 - 3 */
    return Tp;
}

/* lstm: one chunk x [T = 9, 80] -> enc_out [1, J]; h / c of every layer advance by one frame */
static int lstm_online_chunk(const k2o_model* m, k2o_online_stream* s, const float* x, float* enc_out) {
    int Tp;
    float* e = lstm_embed(m, x, 1, k2o_online_chunk_length(m), &Tp);
    if (!e) return fail("lstm chunk too short");
    const int D = m->dim[0];
    for (int li = 0; li < m->nlayer[0]; li++) lstm_layer(m, li, e, 1, Tp, s->lstm_h + (size_t)li * D, s->lstm_c + (size_t)li * m->rnn_hidden);
    linear(enc_out, m->J, e, D, WT(m, m->J, D, "joiner.encoder_proj.weight"), W(m, "joiner.encoder_proj.bias"), Tp, D, m->J);
    free(e);
    return Tp;
}
int64_t k2o_online_stream_lstm_state(const k2o_model* m, const k2o_online_stream* s, int layer, int kind, float* out, int64_t cap) {
    if (s->conf_attn) {   /* streaming conformer: kind 0 = cached_attn [left, D], 1 = cached_conv [K-1, D] of `layer` */
        const int left = meta_int(m, "left_context", 64), D = m->dim[0], K = m->kern[0];
        size_t n = kind == 0 ? (size_t)left * D : (size_t)(K - 1) * D;
        const float* p = kind == 0 ? s->conf_attn + (size_t)layer * n : s->conf_conv + (size_t)layer * n;
        if (!out) return (int64_t)n;
        if ((int64_t)n > cap) return fail("state buffer too small");
        memcpy(out, p, sizeof(float) * n);
        return (int64_t)n;
    }
    if (!s->lstm_h) return fail("not an lstm stream");
    size_t n = kind == 0 ? (size_t)m->dim[0] : (size_t)m->rnn_hidden;
    const float* p = kind == 0 ? s->lstm_h + (size_t)layer * n : s->lstm_c + (size_t)layer * n;
    if (!out) return (int64_t)n;
    if ((int64_t)n > cap) return fail("state buffer too small");
    memcpy(out, p, sizeof(float) * n);
    return (int64_t)n;
}

static void stream_push_token(k2o_online_stream* s, int64_t y, int32_t t) {
    if (s->n_tokens == s->cap_tokens) { s->cap_tokens *= 2; s->tokens = realloc(s->tokens, sizeof(int64_t) * s->cap_tokens); }
    s->tokens[s->n_tokens++] = y;
    if (s->n_ts == s->cap_ts) { s->cap_ts *= 2; s->timestamps = realloc(s->timestamps, sizeof(int32_t) * s->cap_ts); }
    s->timestamps[s->n_ts++] = t;
}

/* OnlineRecognizer.ForwardBatchGreedySearch (:85-219) for B streams that each have one full chunk
 * (chunks[b]: [T*80] raw features; the online PadSequence maps 0.0 to the log floor, PadHelper.cs:9-13,58). */
/* OnlineRecognizer.ForwardBatchGreedySearchCTC (:220-313): per chunk, CTC greedy over the chunk's log_probs with
 * prev_id reset to -1 (a symbol repeated across the chunk boundary is emitted again) and timestamp = t +
 * stream.FrameOffset -- which the reference never writes back (:298-301 update a local list only), so it stays 0
 * and timestamps restart at every chunk.  NumTrailingBlank is not written back either. */
static int online_step_ctc(const k2o_model* m, k2o_online_stream** streams, const float* const* chunks, int B, int32_t* n_new) {
    int T = k2o_online_chunk_length(m), V = m->V;
    int Tp = k2o_online_frames_per_chunk(m);
    float* lp = falloc((size_t)Tp * V);
    float* xf = falloc((size_t)T * m->feat);
    int rc = 0;
    for (int b = 0; b < B && !rc; b++) {
        for (int i = 0; i < T * m->feat; i++) xf[i] = chunks[b][i] == 0.0f ? -23.025850929940457F : chunks[b][i];
        int tp = k2o_online_encoder_chunk(m, streams[b], xf, lp);
        if (tp != Tp) { rc = tp < 0 ? tp : fail("chunk gave %d frames, expected %d", tp, Tp); break; }
        int64_t tok[64];
        int32_t ts[64], n = 0;
        rc = k2o_ctc_greedy(lp, 1, Tp, V, NULL, tok, ts, &n, 64, NULL);
        for (int i = 0; i < n && !rc; i++) stream_push_token(streams[b], tok[i], ts[i]);
        n_new[b] = n;
    }
    free(lp); free(xf);
    return rc;
}

int k2o_online_step(const k2o_model* m, k2o_online_stream** streams, const float* const* chunks, int B, int32_t* n_new) {
    if (m->ctc) return online_step_ctc(m, streams, chunks, B, n_new);
    int T = k2o_online_chunk_length(m), J = m->J, V = m->V, ctx = m->ctx;
    const int blank = 0, unk = 2;
    int Tp = k2o_online_frames_per_chunk(m);
    float* enc = falloc((size_t)B * Tp * J);
    float* xf = falloc((size_t)T * m->feat);
    for (int b = 0; b < B; b++) {
        for (int i = 0; i < T * m->feat; i++) xf[i] = chunks[b][i] == 0.0f ? -23.025850929940457F : chunks[b][i];
        int tp = k2o_online_encoder_chunk(m, streams[b], xf, enc + (size_t)b * Tp * J);
        if (tp != Tp) { free(enc); free(xf); return tp < 0 ? tp : fail("chunk gave %d frames, expected %d", tp, Tp); }
        n_new[b] = 0;
    }
    free(xf);
    int64_t* hyps = (int64_t*)malloc(sizeof(int64_t) * ctx * B);
    for (int b = 0; b < B; b++) { hyps[b * ctx] = streams[b]->hyp[0]; hyps[b * ctx + 1] = streams[b]->hyp[1]; }
    float* dec = falloc((size_t)B * J);
    int rc = k2o_decoder(m, hyps, B, dec);
    float* cur = falloc((size_t)B * J);
    float* logits = falloc((size_t)B * V);
    for (int t = 0; t < Tp && !rc; t++) {
        for (int b = 0; b < B; b++) memcpy(cur + (size_t)b * J, enc + ((size_t)b * Tp + t) * J, sizeof(float) * J);
        k2o_joiner(m, cur, dec, B, logits);
        int emitted = 0;
        for (int b = 0; b < B; b++) {
            int y = k2o_argmax_ref(logits + (size_t)b * V, V);
            if (y != blank && y != unk && y != 1) {   /* OnlineRecognizer.cs:181 */
                stream_push_token(streams[b], y, t);
                n_new[b]++;
                emitted = 1;
            }
        }
        if (emitted) {
            for (int b = 0; b < B; b++)
                for (int k = 0; k < ctx; k++) hyps[b * ctx + k] = streams[b]->tokens[streams[b]->n_tokens - ctx + k];
            rc = k2o_decoder(m, hyps, B, dec);
        }
    }
    if (m->conformer)   /* OnlineProjOfConformer.unstack_states (:229): processed_lens is overwritten with the BATCH SIZE, not the model output */
        for (int b = 0; b < B; b++) streams[b]->processed_len = B;
    for (int b = 0; b < B; b++) {   /* :208 stream.Hyp = last ctx tokens */
        streams[b]->hyp[0] = streams[b]->tokens[streams[b]->n_tokens - 2];
        streams[b]->hyp[1] = streams[b]->tokens[streams[b]->n_tokens - 1];
    }
    free(enc); free(hyps); free(dec); free(cur); free(logits);
    return rc;
}
