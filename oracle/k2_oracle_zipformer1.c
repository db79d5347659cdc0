/*
 * k2_oracle_zipformer1.c -- CPU restatement of the streaming Zipformer (v1) encoder chunk (included by
 * k2_oracle_online.c; TEST INFRASTRUCTURE ONLY, see k2_oracle.h).  PARITY UNPINNED.
 *
 * Reference side: Model_type "zipformer" -> OnlineProjOfZipformer (OnlineRecognizer.cs:28-30).  Its per-stack states
 * (GetEncoderInitStates, OnlineProjOfZipformer.cs:56-111; tensor shapes at :430-487) are
 *   cached_len [L,B] (int64 in the graph, kept as float on the host :437,:500), cached_avg [L,B,D],
 *   cached_key [L,left,B,att], cached_val / cached_val2 [L,left,B,att/2], cached_conv1 / cached_conv2 [L,B,D,K-1];
 * chunk = T = 39 frames, shift = decode_chunk_len = 32 (OnlineProjOfZipformer.cs:24-25, OnlineModel.cs:48-49).
 * The graph is not in the reference; this restates icefall's pruned_transducer_stateless7_streaming
 * Zipformer.streaming_forward with Scaled* modules folded and balancers / whiteners dropped (identity at inference):
 *   Conv2dSubsampling: Conv2d(1,8,3,pad (0,1)) DoubleSwish; Conv2d(8,32,3,stride 2) DoubleSwish;
 *     Conv2d(32,128,3,stride (1,2)) DoubleSwish; Linear(128*19 -> D0) over (c,f).   (T-7)//2 frames.
 *   per stack i: [skip SimpleCombiner] -> (ds = 1) layers | (ds > 1) AttentionDownsample -> layers -> SimpleUpsample ->
 *     truncate -> out_combiner SimpleCombiner(orig, up)
 *   ZipformerEncoderLayer.streaming_forward: x += ff1(x); x += pooling(x); x += attn(x) [weights kept];
 *     x += conv1(x); x += ff2(x); x += attn2(x, weights); x += conv2(x); x += ff3(x); x = BasicNorm(x);
 *     x = orig + (x - orig) * bypass_scale
 *   PoolingModule.streaming_forward: running mean over all frames seen so far (cached_len, cached_avg), then proj
 *   RelPositionMultiheadAttention.streaming: in_proj -> q | k | v (att/2) | p (pos_dim per head); keys / values are
 *     [cache ; chunk]; scores = q.k + p.pos[rel]; NO mask over the zero-initialised left context (v1 has none);
 *     softmax; out_proj(weights.v); second use: in_proj2 -> v2 with its own cache, out_proj2
 *   ConvolutionModule.streaming_forward: pointwise(D->2D) GLU, [cache(K-1) ; chunk], depthwise K (valid), DoubleSwish,
 *     pointwise(D->D)
 *   downsample_output: AttentionDownsample by 2; then joiner.encoder_proj as the ONNX encoder wrapper applies it.
 */

/* RelPositionalEncoding.forward(x, left_context_len): row n <-> relative position (Tc+L-1) - n */
static float* z1_rel_pos(int Tc, int L, int D) {
    const int n2 = 2 * Tc - 1 + L;
    float* pe = falloc((size_t)n2 * D);
    const float c = -(logf(10000.0f) / (float)D);
    for (int n = 0; n < n2; n++) {
        float r = (float)(L + Tc - 1 - n);
        for (int k = 0; k < D / 2; k++) {
            float div = expf((float)(2 * k) * c);
            pe[(size_t)n * D + 2 * k] = sinf(r * div);
            pe[(size_t)n * D + 2 * k + 1] = cosf(r * div);
        }
    }
    return pe;
}

/* Conv2dSubsampling (v1) for one chunk: x [T,80] -> [(T-7)//2, D0] */
static float* z1_embed(const k2o_model* m, const float* x, int T, int* Tc_out) {
    const int F0 = m->feat, T1 = T - 2, T2 = (T1 - 3) / 2 + 1, F2 = (F0 - 3) / 2 + 1, T3 = T2 - 2, F3 = (F2 - 3) / 2 + 1, D0 = m->dim[0];
    const float *w0 = W(m, "encoder.encoder_embed.conv.0.weight"), *b0 = W(m, "encoder.encoder_embed.conv.0.bias");
    const float *w1 = W(m, "encoder.encoder_embed.conv.3.weight"), *b1 = W(m, "encoder.encoder_embed.conv.3.bias");
    const float *w2 = W(m, "encoder.encoder_embed.conv.6.weight"), *b2 = W(m, "encoder.encoder_embed.conv.6.bias");
    float* a1 = falloc((size_t)8 * T1 * F0);
    for (int co = 0; co < 8; co++)
        for (int t = 0; t < T1; t++)
            for (int f = 0; f < F0; f++) {
                float s = b0[co];
                for (int kt = 0; kt < 3; kt++)
                    for (int kf = 0; kf < 3; kf++) {
                        int fi = f + kf - 1;
                        if (fi < 0 || fi >= F0) continue;
                        s += w0[(co * 3 + kt) * 3 + kf] * x[(size_t)(t + kt) * F0 + fi];
                    }
                a1[((size_t)co * T1 + t) * F0 + f] = double_swish(s);
            }
    float* a2 = falloc((size_t)32 * T2 * F2);
    for (int co = 0; co < 32; co++)
        for (int t = 0; t < T2; t++)
            for (int f = 0; f < F2; f++) {
                float s = b1[co];
                for (int ci = 0; ci < 8; ci++)
                    for (int kt = 0; kt < 3; kt++)
                        for (int kf = 0; kf < 3; kf++)
                            s += w1[((co * 8 + ci) * 3 + kt) * 3 + kf] * a1[((size_t)ci * T1 + 2 * t + kt) * F0 + 2 * f + kf];
                a2[((size_t)co * T2 + t) * F2 + f] = double_swish(s);
            }
    free(a1);
    float* a3 = falloc((size_t)T3 * 128 * F3);  /* [t][c*F3 + f] */
    for (int co = 0; co < 128; co++)
        for (int t = 0; t < T3; t++)
            for (int f = 0; f < F3; f++) {
                float s = b2[co];
                for (int ci = 0; ci < 32; ci++)
                    for (int kt = 0; kt < 3; kt++)
                        for (int kf = 0; kf < 3; kf++)
                            s += w2[((co * 32 + ci) * 3 + kt) * 3 + kf] * a2[((size_t)ci * T2 + t + kt) * F2 + 2 * f + kf];
                a3[(size_t)t * (128 * F3) + co * F3 + f] = double_swish(s);
            }
    free(a2);
    float* out = falloc((size_t)T3 * D0);
    linear(out, D0, a3, 128 * F3, WT(m, D0, 128 * F3, "encoder.encoder_embed.out.weight"), W(m, "encoder.encoder_embed.out.bias"), T3,
           128 * F3, D0);
    free(a3);
    *Tc_out = T3;
    return out;
}

/* SimpleCombiner.forward(src1 [T,d1], src2 [T,d2]) -> [T,d2]: src1*w + src2*(1-w), src1 zero-padded / truncated to d2 */
static float* z1_combine(const float* s1, int d1, const float* s2, int d2, float w1, int T) {
    float* y = falloc((size_t)T * d2);
    for (int t = 0; t < T; t++)
        for (int d = 0; d < d2; d++) {
            float a = d < d1 ? s1[(size_t)t * d1 + d] * w1 : 0.f;
            y[(size_t)t * d2 + d] = a + s2[(size_t)t * d2 + d] * (1.0f - w1);
        }
    return y;
}

/* AttentionDownsample.forward: src [T,Din] -> [ceil(T/ds), Dout] (Dout >= Din; extra channels = extra_proj of the ds frames) */
static float* z1_attn_downsample(const k2o_model* m, const char* pfx, const float* src, int T, int Din, int Dout, int ds, int* Td_out) {
    const int Td = (T + ds - 1) / ds;
    const float* q = W(m, "%squery", pfx);
    float* y = falloc((size_t)Td * Dout);
    float* grp = falloc((size_t)Td * ds * Din); /* [Td][ds*Din]: the group's frames side by side (last frame repeated as padding) */
    for (int t = 0; t < Td; t++) {
        float sc[16], mx = -INFINITY, sum = 0.f;
        for (int k = 0; k < ds; k++) {
            int tt = t * ds + k;
            if (tt >= T) tt = T - 1;
            const float* r = src + (size_t)tt * Din;
            memcpy(grp + ((size_t)t * ds + k) * Din, r, sizeof(float) * Din);
            float s = 0.f;
            for (int d = 0; d < Din; d++) s += r[d] * q[d];
            sc[k] = s;
            if (s > mx) mx = s;
        }
        for (int k = 0; k < ds; k++) { sc[k] = expf(sc[k] - mx); sum += sc[k]; }
        for (int d = 0; d < Din; d++) {
            float a = 0.f;
            for (int k = 0; k < ds; k++) a += grp[((size_t)t * ds + k) * Din + d] * (sc[k] / sum);
            y[(size_t)t * Dout + d] = a;
        }
    }
    if (Dout > Din) {
        float* ex = falloc((size_t)Td * (Dout - Din));
        linear(ex, Dout - Din, grp, ds * Din, WT(m, Dout - Din, ds * Din, "%sextra_proj.weight", pfx), NULL, Td, ds * Din, Dout - Din);
        for (int t = 0; t < Td; t++) memcpy(y + (size_t)t * Dout + Din, ex + (size_t)t * (Dout - Din), sizeof(float) * (Dout - Din));
        free(ex);
    }
    free(grp);
    *Td_out = Td;
    return y;
}

static void z1_feed_forward(const k2o_model* m, const char* pfx, int k, float* src, int M, int D, int F) {
    float* hid = falloc((size_t)M * F);
    linear(hid, F, src, D, WT(m, F, D, "%sfeed_forward%d.in_proj.weight", pfx, k), W(m, "%sfeed_forward%d.in_proj.bias", pfx, k), M, D, F);
    for (size_t i = 0; i < (size_t)M * F; i++) hid[i] = double_swish(hid[i]);
    float* o = falloc((size_t)M * D);
    linear(o, D, hid, F, WT(m, D, F, "%sfeed_forward%d.out_proj.weight", pfx, k), W(m, "%sfeed_forward%d.out_proj.bias", pfx, k), M, F, D);
    add_inplace(src, o, (size_t)M * D);
    free(hid); free(o);
}

/* ConvolutionModule.streaming_forward; cache [D][K-1] (channel-major, as the reference's [B,D,K-1] tensor) */
static void z1_conv_module(const k2o_model* m, const char* pfx, int k, float* src, int Tc, int D, int K, float* cache) {
    const int lo = K - 1;
    float* x = falloc((size_t)Tc * 2 * D);
    linear(x, 2 * D, src, D, WT(m, 2 * D, D, "%sconv_module%d.pointwise_conv1.weight", pfx, k), W(m, "%sconv_module%d.pointwise_conv1.bias", pfx, k), Tc, D, 2 * D);
    float* cat = falloc((size_t)(lo + Tc) * D);
    for (int t = 0; t < lo; t++)
        for (int d = 0; d < D; d++) cat[(size_t)t * D + d] = cache[(size_t)d * lo + t];
    for (int t = 0; t < Tc; t++)
        for (int d = 0; d < D; d++)
            cat[(size_t)(lo + t) * D + d] = x[(size_t)t * 2 * D + d] * (1.0f / (1.0f + expf(-x[(size_t)t * 2 * D + D + d])));
    for (int t = 0; t < lo; t++)
        for (int d = 0; d < D; d++) cache[(size_t)d * lo + t] = cat[(size_t)(Tc + t) * D + d];
    const float* wd = W(m, "%sconv_module%d.depthwise_conv.weight", pfx, k); /* [D,1,K] */
    const float* bd = W(m, "%sconv_module%d.depthwise_conv.bias", pfx, k);
    float* c1 = falloc((size_t)Tc * D);
    for (int t = 0; t < Tc; t++)
        for (int d = 0; d < D; d++) {
            float s = bd[d];
            for (int kk = 0; kk < K; kk++) s += wd[d * K + kk] * cat[(size_t)(t + kk) * D + d];
            c1[(size_t)t * D + d] = double_swish(s);
        }
    float* o = falloc((size_t)Tc * D);
    linear(o, D, c1, D, WT(m, D, D, "%sconv_module%d.pointwise_conv2.weight", pfx, k), W(m, "%sconv_module%d.pointwise_conv2.bias", pfx, k), Tc, D, D);
    add_inplace(src, o, (size_t)Tc * D);
    free(x); free(cat); free(c1); free(o);
}

/* ZipformerEncoderLayer.streaming_forward, one stream, in place on src [Tc, D]; l = global layer index */
static void z1_layer(const k2o_model* m, k2o_online_stream* s, int si, const char* pfx, int l, float* src, const float* pe, int Tc, int L) {
    const int D = m->dim[si], A = m->att[si], H = m->heads[si], F = m->ff[si], K = m->kern[si], P = m->pos_dim;
    const int hd = A / H, vd = A / 2 / H, KL = L + Tc, n2 = 2 * Tc - 1 + L, inproj = 2 * A + A / 2 + P * H;
    float* orig = falloc((size_t)Tc * D);
    memcpy(orig, src, sizeof(float) * (size_t)Tc * D);
    z1_feed_forward(m, pfx, 1, src, Tc, D, F);
    {   /* pooling: mean over every frame seen so far */
        float* pool = falloc((size_t)Tc * D);
        float len = s->clen[l];
        float* avg = s->avg[l];
        for (int d = 0; d < D; d++) {
            float cum = 0.f, base = avg[d] * len;   /* x.cumsum(0) + cached_avg * cached_len, then * 1 / (t + 1 + cached_len) */
            for (int t = 0; t < Tc; t++) {
                cum += src[(size_t)t * D + d];
                pool[(size_t)t * D + d] = (cum + base) * (1.0f / ((float)(t + 1) + len));
            }
            avg[d] = pool[(size_t)(Tc - 1) * D + d];
        }
        s->clen[l] = len + (float)Tc;
        float* o = falloc((size_t)Tc * D);
        linear(o, D, pool, D, WT(m, D, D, "%spooling.proj.weight", pfx), NULL, Tc, D, D);
        add_inplace(src, o, (size_t)Tc * D);
        free(pool); free(o);
    }
    float* aw = falloc((size_t)H * Tc * KL);
    {   /* self_attn.streaming_forward */
        float* x = falloc((size_t)Tc * inproj);
        linear(x, inproj, src, D, WT(m, inproj, D, "%sself_attn.in_proj.weight", pfx), W(m, "%sself_attn.in_proj.bias", pfx), Tc, D, inproj);
        float* knew = falloc((size_t)Tc * A);
        float* vnew = falloc((size_t)Tc * (A / 2));
        for (int t = 0; t < Tc; t++) {
            memcpy(knew + (size_t)t * A, x + (size_t)t * inproj + A, sizeof(float) * A);
            memcpy(vnew + (size_t)t * (A / 2), x + (size_t)t * inproj + 2 * A, sizeof(float) * (A / 2));
        }
        float* kcat = cat_and_shift(s->key[l], knew, L, Tc, A);
        float* vcat = cat_and_shift(s->val1[l], vnew, L, Tc, A / 2);
        free(knew); free(vnew);
        float* pp = falloc((size_t)n2 * P * H);
        linear(pp, P * H, pe, D, WT(m, P * H, D, "%sself_attn.linear_pos.weight", pfx), NULL, n2, D, P * H);
        for (int h = 0; h < H; h++)
            for (int i = 0; i < Tc; i++) {
                const float* qi = x + (size_t)i * inproj + h * hd;
                const float* pi = x + (size_t)i * inproj + 2 * A + A / 2 + h * P;
                float* row = aw + ((size_t)h * Tc + i) * KL;
                float mx = -INFINITY;
                for (int j = 0; j < KL; j++) {
                    const float* kj = kcat + (size_t)j * A + h * hd;
                    float sc = 0.f;
                    for (int d = 0; d < hd; d++) sc += qi[d] * kj[d];
                    const float* pr = pp + (size_t)(Tc - 1 - i + j) * (P * H) + h * P; /* as_strided: rel index = (Tc-1) - i + j */
                    float ps = 0.f;
                    for (int c = 0; c < P; c++) ps += pi[c] * pr[c];
                    row[j] = sc + ps;
                    if (row[j] > mx) mx = row[j];
                }
                float sum = 0.f;
                for (int j = 0; j < KL; j++) { row[j] = expf(row[j] - mx); sum += row[j]; }
                float inv = 1.0f / sum;
                for (int j = 0; j < KL; j++) row[j] *= inv;
            }
        float* a = falloc((size_t)Tc * (A / 2));
        for (int h = 0; h < H; h++) attn_apply1(a, A / 2, aw + (size_t)h * Tc * KL, vcat, A / 2, Tc, KL, h * vd, vd);
        float* o = falloc((size_t)Tc * D);
        linear(o, D, a, A / 2, WT(m, D, A / 2, "%sself_attn.out_proj.weight", pfx), W(m, "%sself_attn.out_proj.bias", pfx), Tc, A / 2, D);
        add_inplace(src, o, (size_t)Tc * D);
        free(x); free(kcat); free(vcat); free(pp); free(a); free(o);
    }
    z1_conv_module(m, pfx, 1, src, Tc, D, K, s->conv1[l]);
    z1_feed_forward(m, pfx, 2, src, Tc, D, F);
    {   /* self_attn.streaming_forward2: the same weights over a second value projection */
        float* v = falloc((size_t)Tc * (A / 2));
        linear(v, A / 2, src, D, WT(m, A / 2, D, "%sself_attn.in_proj2.weight", pfx), NULL, Tc, D, A / 2);
        float* vcat = cat_and_shift(s->val2[l], v, L, Tc, A / 2);
        float* a = falloc((size_t)Tc * (A / 2));
        for (int h = 0; h < H; h++) attn_apply1(a, A / 2, aw + (size_t)h * Tc * KL, vcat, A / 2, Tc, KL, h * vd, vd);
        float* o = falloc((size_t)Tc * D);
        linear(o, D, a, A / 2, WT(m, D, A / 2, "%sself_attn.out_proj2.weight", pfx), W(m, "%sself_attn.out_proj2.bias", pfx), Tc, A / 2, D);
        add_inplace(src, o, (size_t)Tc * D);
        free(v); free(vcat); free(a); free(o);
    }
    free(aw);
    z1_conv_module(m, pfx, 2, src, Tc, D, K, s->conv2[l]);
    z1_feed_forward(m, pfx, 3, src, Tc, D, F);
    basic_norm(src, src, W(m, "%snorm_final.eps", pfx)[0], Tc, D);
    const float bs = W(m, "%sbypass_scale", pfx)[0];
    for (size_t i = 0; i < (size_t)Tc * D; i++) src[i] = orig[i] + (src[i] - orig[i]) * bs;
    free(orig);
}

/* which earlier stack's output is mixed into stack i's input (Zipformer._init_skip_modules); -1 = none */
static int z1_skip_layer(const k2o_model* m, int i) {
    if (i <= 1 || m->ds[i - 1] <= m->ds[i]) return -1;
    for (int j = i - 2; j >= 0; j--)
        if (m->ds[j] <= m->ds[i] || j == 0) return j;
    return -1;
}

/* one chunk of one stream: x [T,80] (log-floored) -> enc_out [T', J]; returns T' */
static int z1_stream_chunk(const k2o_model* m, k2o_online_stream* s, const float* x, float* enc_out) {
    int Tc;
    float* cur = z1_embed(m, x, k2o_online_chunk_length(m), &Tc);
    float* outputs[MAX_STACKS] = {0};
    int Dcur = m->dim[0], l = 0;
    for (int si = 0; si < m->ns; si++) {
        const int D = m->dim[si], ds = m->ds[si], L = online_left(m, si);
        int k = z1_skip_layer(m, si);
        if (k >= 0) {
            float* y = z1_combine(outputs[k], m->dim[k], cur, Dcur, W(m, "encoder.skip_modules.%d.weight1", si)[0], Tc);
            free(cur);
            cur = y;
        }
        char pfx[128];
        if (ds == 1) {
            if (D != Dcur) { free(cur); return fail("zipformer: stack %d has downsampling 1 but changes width %d -> %d", si, Dcur, D); }
            float* pe = z1_rel_pos(Tc, L, D);
            for (int li = 0; li < m->nlayer[si]; li++, l++) {
                snprintf(pfx, sizeof pfx, "encoder.encoders.%d.layers.%d.", si, li);
                z1_layer(m, s, si, pfx, l, cur, pe, Tc, L);
            }
            free(pe);
        } else {
            if (D < Dcur) { free(cur); return fail("zipformer: stack %d narrows %d -> %d (unsupported)", si, Dcur, D); }
            int Td;
            snprintf(pfx, sizeof pfx, "encoder.encoders.%d.downsample.", si);
            float* xd = z1_attn_downsample(m, pfx, cur, Tc, Dcur, D, ds, &Td);
            float* pe = z1_rel_pos(Td, L, D);
            for (int li = 0; li < m->nlayer[si]; li++, l++) {
                snprintf(pfx, sizeof pfx, "encoder.encoders.%d.encoder.layers.%d.", si, li);
                z1_layer(m, s, si, pfx, l, xd, pe, Td, L);
            }
            free(pe);
            /* SimpleUpsample (+ bias[k]) truncated to Tc frames, then out_combiner(src_orig, src) */
            const float* ub = W(m, "encoder.encoders.%d.upsample.bias", si);
            float* up = falloc((size_t)Tc * D);
            for (int t = 0; t < Tc; t++)
                for (int d = 0; d < D; d++) up[(size_t)t * D + d] = xd[(size_t)(t / ds) * D + d] + ub[(size_t)(t % ds) * D + d];
            float* y = z1_combine(cur, Dcur, up, D, W(m, "encoder.encoders.%d.out_combiner.weight1", si)[0], Tc);
            free(up); free(xd); free(cur);
            cur = y;
            Dcur = D;
        }
        outputs[si] = falloc((size_t)Tc * Dcur);
        memcpy(outputs[si], cur, sizeof(float) * (size_t)Tc * Dcur);
    }
    for (int i = 0; i < m->ns; i++) free(outputs[i]);
    int Tp;
    float* dsd = z1_attn_downsample(m, "encoder.downsample_output.", cur, Tc, Dcur, Dcur, 2, &Tp);
    free(cur);
    linear(enc_out, m->J, dsd, Dcur, WT(m, m->J, Dcur, "joiner.encoder_proj.weight"), W(m, "joiner.encoder_proj.bias"), Tp, Dcur, m->J);
    free(dsd);
    return Tp;
}

/* ------------------------------------------------------------------------------------------------------------------------
 * Offline Zipformer v1: Model_type "zipformer" in OfflineRecognizer's switch (OfflineRecognizer.cs:40-44 -> OfflineProjOfTransducer,
 * x [B,T,80], x_lens = T for every row, :48-92).  The graph is icefall's pruned_transducer_stateless7 Zipformer.forward (the
 * non-streaming recipe), inference: the same modules as above with
 *   PoolingModule.forward: the mean over ALL frames of the utterance (the key-padding mask is all False, so every frame gets
 *     weight 1/T), projected and added to every frame;
 *   attention over the whole utterance (rel-pos table for positions T-1 .. -(T-1)), weights used twice;
 *   ConvolutionModule.forward: depthwise conv with zero padding K//2 on both sides (not causal).
 * Utterances are independent (no padding mask can couple them: x_lens = T), so the batch is a loop.
 * ------------------------------------------------------------------------------------------------------------------------ */
static void z1_conv_module_offline(const k2o_model* m, const char* pfx, int k, float* src, int T, int D, int K) {
    const int pad = K / 2;
    float* x = falloc((size_t)T * 2 * D);
    linear(x, 2 * D, src, D, WT(m, 2 * D, D, "%sconv_module%d.pointwise_conv1.weight", pfx, k), W(m, "%sconv_module%d.pointwise_conv1.bias", pfx, k), T, D, 2 * D);
    float* g = falloc((size_t)T * D);
    for (int t = 0; t < T; t++)
        for (int d = 0; d < D; d++) g[(size_t)t * D + d] = x[(size_t)t * 2 * D + d] * (1.0f / (1.0f + expf(-x[(size_t)t * 2 * D + D + d])));
    const float* wd = W(m, "%sconv_module%d.depthwise_conv.weight", pfx, k);
    const float* bd = W(m, "%sconv_module%d.depthwise_conv.bias", pfx, k);
    float* c1 = falloc((size_t)T * D);
    for (int t = 0; t < T; t++)
        for (int d = 0; d < D; d++) {
            float s = bd[d];
            for (int kk = 0; kk < K; kk++) {
                int tt = t + kk - pad;
                if (tt >= 0 && tt < T) s += wd[d * K + kk] * g[(size_t)tt * D + d];
            }
            c1[(size_t)t * D + d] = double_swish(s);
        }
    float* o = falloc((size_t)T * D);
    linear(o, D, c1, D, WT(m, D, D, "%sconv_module%d.pointwise_conv2.weight", pfx, k), W(m, "%sconv_module%d.pointwise_conv2.bias", pfx, k), T, D, D);
    add_inplace(src, o, (size_t)T * D);
    free(x); free(g); free(c1); free(o);
}

static void z1_layer_offline(const k2o_model* m, int si, const char* pfx, float* src, const float* pe, int T) {
    const int D = m->dim[si], A = m->att[si], H = m->heads[si], F = m->ff[si], K = m->kern[si], P = m->pos_dim;
    const int hd = A / H, vd = A / 2 / H, n2 = 2 * T - 1, inproj = 2 * A + A / 2 + P * H;
    float* orig = falloc((size_t)T * D);
    memcpy(orig, src, sizeof(float) * (size_t)T * D);
    z1_feed_forward(m, pfx, 1, src, T, D, F);
    {   /* pooling: mean over the utterance's frames, (x * 1/T).sum(0) */
        float* mean = falloc(D);
        const float wgt = 1.0f / (float)T;
        for (int d = 0; d < D; d++) {
            float a = 0.f;
            for (int t = 0; t < T; t++) a += src[(size_t)t * D + d] * wgt;
            mean[d] = a;
        }
        float* o = falloc(D);
        linear(o, D, mean, D, WT(m, D, D, "%spooling.proj.weight", pfx), NULL, 1, D, D);
        for (int t = 0; t < T; t++)
            for (int d = 0; d < D; d++) src[(size_t)t * D + d] += o[d];
        free(mean); free(o);
    }
    float* aw = falloc((size_t)H * T * T);
    {
        float* x = falloc((size_t)T * inproj);
        linear(x, inproj, src, D, WT(m, inproj, D, "%sself_attn.in_proj.weight", pfx), W(m, "%sself_attn.in_proj.bias", pfx), T, D, inproj);
        float* pp = falloc((size_t)n2 * P * H);
        linear(pp, P * H, pe, D, WT(m, P * H, D, "%sself_attn.linear_pos.weight", pfx), NULL, n2, D, P * H);
#pragma omp parallel for collapse(2) schedule(static)
        for (int h = 0; h < H; h++)
            for (int i = 0; i < T; i++) {
                const float* qi = x + (size_t)i * inproj + h * hd;
                const float* pi = x + (size_t)i * inproj + 2 * A + A / 2 + h * P;
                float* row = aw + ((size_t)h * T + i) * T;
                float mx = -INFINITY;
                for (int j = 0; j < T; j++) {
                    const float* kj = x + (size_t)j * inproj + A + h * hd;
                    float sc = 0.f;
                    for (int d = 0; d < hd; d++) sc += qi[d] * kj[d];
                    const float* pr = pp + (size_t)(T - 1 - i + j) * (P * H) + h * P;
                    float ps = 0.f;
                    for (int c = 0; c < P; c++) ps += pi[c] * pr[c];
                    row[j] = sc + ps;
                    if (row[j] > mx) mx = row[j];
                }
                float sum = 0.f;
                for (int j = 0; j < T; j++) { row[j] = expf(row[j] - mx); sum += row[j]; }
                float inv = 1.0f / sum;
                for (int j = 0; j < T; j++) row[j] *= inv;
            }
        float* v = falloc((size_t)T * (A / 2));
        for (int t = 0; t < T; t++) memcpy(v + (size_t)t * (A / 2), x + (size_t)t * inproj + 2 * A, sizeof(float) * (A / 2));
        float* a = falloc((size_t)T * (A / 2));
        for (int h = 0; h < H; h++) attn_apply1(a, A / 2, aw + (size_t)h * T * T, v, A / 2, T, T, h * vd, vd);
        float* o = falloc((size_t)T * D);
        linear(o, D, a, A / 2, WT(m, D, A / 2, "%sself_attn.out_proj.weight", pfx), W(m, "%sself_attn.out_proj.bias", pfx), T, A / 2, D);
        add_inplace(src, o, (size_t)T * D);
        free(x); free(pp); free(v); free(a); free(o);
    }
    z1_conv_module_offline(m, pfx, 1, src, T, D, K);
    z1_feed_forward(m, pfx, 2, src, T, D, F);
    {
        float* v = falloc((size_t)T * (A / 2));
        linear(v, A / 2, src, D, WT(m, A / 2, D, "%sself_attn.in_proj2.weight", pfx), NULL, T, D, A / 2);
        float* a = falloc((size_t)T * (A / 2));
        for (int h = 0; h < H; h++) attn_apply1(a, A / 2, aw + (size_t)h * T * T, v, A / 2, T, T, h * vd, vd);
        float* o = falloc((size_t)T * D);
        linear(o, D, a, A / 2, WT(m, D, A / 2, "%sself_attn.out_proj2.weight", pfx), W(m, "%sself_attn.out_proj2.bias", pfx), T, A / 2, D);
        add_inplace(src, o, (size_t)T * D);
        free(v); free(a); free(o);
    }
    free(aw);
    z1_conv_module_offline(m, pfx, 2, src, T, D, K);
    z1_feed_forward(m, pfx, 3, src, T, D, F);
    basic_norm(src, src, W(m, "%snorm_final.eps", pfx)[0], T, D);
    const float bs = W(m, "%sbypass_scale", pfx)[0];
    for (size_t i = 0; i < (size_t)T * D; i++) src[i] = orig[i] + (src[i] - orig[i]) * bs;
    free(orig);
}

/* taps: 0 = embed output [B*T50, D0]; 1+i = output of stack i [B*T50, dim[i]]; -1 = encoder_out [B, T', J] */
static int zip1_offline_forward(const k2o_model* m, const float* xin, int B, int T, float* enc_out, int tap, float* tap_out, int64_t tap_cap,
                                int64_t* tap_n) {
    if (T < 9) return fail("encoder: T=%d too short", T);
    int Tc0 = (T - 7) / 2, Tp0 = (Tc0 + 1) / 2;
    if (tap >= 0) {
        if (tap > m->ns) return fail("tap %d out of range", tap);
        int64_t n = (int64_t)B * Tc0 * (tap == 0 ? m->dim[0] : m->dim[tap - 1]);
        if (n > tap_cap) return fail("tap buffer too small");
        *tap_n = n;
    }
    for (int b = 0; b < B; b++) {
        int Tc;
        float* cur = z1_embed(m, xin + (size_t)b * T * m->feat, T, &Tc);
        float* outputs[MAX_STACKS] = {0};
        int Dcur = m->dim[0], rc = 0, done = 0;
        if (tap == 0) { memcpy(tap_out + (size_t)b * Tc * Dcur, cur, sizeof(float) * (size_t)Tc * Dcur); done = 1; }
        for (int si = 0; si < m->ns && !done && !rc; si++) {
            const int D = m->dim[si], ds = m->ds[si];
            int k = z1_skip_layer(m, si);
            if (k >= 0) {
                float* y = z1_combine(outputs[k], m->dim[k], cur, Dcur, W(m, "encoder.skip_modules.%d.weight1", si)[0], Tc);
                free(cur);
                cur = y;
            }
            char pfx[128];
            if (ds == 1) {
                if (D != Dcur) { rc = fail("zipformer: stack %d has downsampling 1 but changes width", si); break; }
                float* pe = z1_rel_pos(Tc, 0, D);
                for (int li = 0; li < m->nlayer[si]; li++) {
                    snprintf(pfx, sizeof pfx, "encoder.encoders.%d.layers.%d.", si, li);
                    z1_layer_offline(m, si, pfx, cur, pe, Tc);
                }
                free(pe);
            } else {
                if (D < Dcur) { rc = fail("zipformer: stack %d narrows %d -> %d (unsupported)", si, Dcur, D); break; }
                int Td;
                snprintf(pfx, sizeof pfx, "encoder.encoders.%d.downsample.", si);
                float* xd = z1_attn_downsample(m, pfx, cur, Tc, Dcur, D, ds, &Td);
                float* pe = z1_rel_pos(Td, 0, D);
                for (int li = 0; li < m->nlayer[si]; li++) {
                    snprintf(pfx, sizeof pfx, "encoder.encoders.%d.encoder.layers.%d.", si, li);
                    z1_layer_offline(m, si, pfx, xd, pe, Td);
                }
                free(pe);
                const float* ub = W(m, "encoder.encoders.%d.upsample.bias", si);
                float* up = falloc((size_t)Tc * D);
                for (int t = 0; t < Tc; t++)
                    for (int d = 0; d < D; d++) up[(size_t)t * D + d] = xd[(size_t)(t / ds) * D + d] + ub[(size_t)(t % ds) * D + d];
                float* y = z1_combine(cur, Dcur, up, D, W(m, "encoder.encoders.%d.out_combiner.weight1", si)[0], Tc);
                free(up); free(xd); free(cur);
                cur = y;
                Dcur = D;
            }
            outputs[si] = falloc((size_t)Tc * Dcur);
            memcpy(outputs[si], cur, sizeof(float) * (size_t)Tc * Dcur);
            if (tap == si + 1) { memcpy(tap_out + (size_t)b * Tc * Dcur, cur, sizeof(float) * (size_t)Tc * Dcur); done = 1; }
        }
        for (int i = 0; i < m->ns; i++) free(outputs[i]);
        if (rc) { free(cur); return rc; }
        if (!done) {
            int Tp;
            float* dsd = z1_attn_downsample(m, "encoder.downsample_output.", cur, Tc, Dcur, Dcur, 2, &Tp);
            linear(enc_out + (size_t)b * Tp0 * m->J, m->J, dsd, Dcur, WT(m, m->J, Dcur, "joiner.encoder_proj.weight"), W(m, "joiner.encoder_proj.bias"), Tp, Dcur, m->J);
            free(dsd);
        }
        free(cur);
    }
    return 0;
}
