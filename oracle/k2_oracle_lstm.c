/*
 * k2_oracle_lstm.c -- CPU restatement of the LSTM transducer encoder, offline and streaming (included by k2_oracle.c;
 * TEST INFRASTRUCTURE ONLY, see k2_oracle.h).  PARITY UNPINNED.
 *
 * Reference side: Model_type "lstm" -> OfflineProjOfTransducer offline (OfflineRecognizer.cs:38-53) and OnlineProjOfLstm
 * streaming, whose per-stream states are state0 = h [num_layers, B, d_model] and state1 = c [num_layers, B, rnn_hidden_size]
 * (OnlineProjOfLstm.cs:55-75), chunk = T frames, shift = decode_chunk_len (OnlineModel.cs:48-49).  The graph itself is
 * icefall's lstm_transducer_stateless2 (not in the reference):
 *   Conv2dSubsampling: Conv2d(1,8,3,pad (0,1)) DoubleSwish; Conv2d(8,32,3,stride 2) DoubleSwish; Conv2d(32,128,3,stride 2)
 *     DoubleSwish; Linear(128*19 -> d_model) over (c,f); BasicNorm.   T' = ((T-3)//2 - 1)//2, no padding in time, so a
 *     chunk of T = 9 frames advanced by 4 gives exactly the offline frames one at a time.
 *   RNNEncoderLayer: y, (h, c) = LSTM(x, (h, c)) with projection (torch.nn.LSTM, gates i,f,g,o; h = W_hr (o * tanh(c)));
 *     x = x + y; x = x + feed_forward(x) (Linear, DoubleSwish, Linear); x = BasicNorm(x).
 */

static int lstm_out_frames(int T) { return T < 9 ? 0 : ((T - 3) / 2 - 1) / 2; }

static float* lstm_embed(const k2o_model* m, const float* x, int B, int T, int* T_out) {
    const int F0 = m->feat, D = m->dim[0];
    const int T1 = T - 2, T2 = (T1 - 3) / 2 + 1, F2 = (F0 - 3) / 2 + 1, T3 = (T2 - 3) / 2 + 1, F3 = (F2 - 3) / 2 + 1;
    if (T < 9 || T3 <= 0) return NULL;
    const float *w0 = W(m, "encoder.encoder_embed.conv.0.weight"), *b0 = W(m, "encoder.encoder_embed.conv.0.bias");
    const float *w1 = W(m, "encoder.encoder_embed.conv.3.weight"), *b1 = W(m, "encoder.encoder_embed.conv.3.bias");
    const float *w2 = W(m, "encoder.encoder_embed.conv.6.weight"), *b2 = W(m, "encoder.encoder_embed.conv.6.bias");
    float* a1 = falloc((size_t)B * 8 * T1 * F0);
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; b++)
        for (int co = 0; co < 8; co++)
            for (int t = 0; t < T1; t++)
                for (int f = 0; f < F0; f++) {
                    float s = b0[co];
                    for (int kt = 0; kt < 3; kt++)
                        for (int kf = 0; kf < 3; kf++) {
                            int fi = f + kf - 1;
                            if (fi < 0 || fi >= F0) continue;
                            s += w0[(co * 3 + kt) * 3 + kf] * x[((size_t)b * T + t + kt) * F0 + fi];
                        }
                    a1[(((size_t)b * 8 + co) * T1 + t) * F0 + f] = double_swish(s);
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
                                s += w1[((co * 8 + ci) * 3 + kt) * 3 + kf] * a1[(((size_t)b * 8 + ci) * T1 + 2 * t + kt) * F0 + 2 * f + kf];
                    a2[(((size_t)b * 32 + co) * T2 + t) * F2 + f] = double_swish(s);
                }
    free(a1);
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
                                s += w2[((co * 32 + ci) * 3 + kt) * 3 + kf] * a2[(((size_t)b * 32 + ci) * T2 + 2 * t + kt) * F2 + 2 * f + kf];
                    a3[((size_t)b * T3 + t) * (128 * F3) + co * F3 + f] = double_swish(s);
                }
    free(a2);
    float* lin = falloc((size_t)B * T3 * D);
    linear(lin, D, a3, 128 * F3, WT(m, D, 128 * F3, "encoder.encoder_embed.out.weight"), W(m, "encoder.encoder_embed.out.bias"), B * T3,
           128 * F3, D);
    free(a3);
    float* out = falloc((size_t)B * T3 * D);
    basic_norm(out, lin, W(m, "encoder.encoder_embed.out_norm.eps")[0], B * T3, D);
    free(lin);
    *T_out = T3;
    return out;
}

static inline float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

/* one RNNEncoderLayer over x [B, T, D] in place; h [B, D] and c [B, Hh] are read and updated */
static void lstm_layer(const k2o_model* m, int li, float* x, int B, int T, float* h, float* c) {
    const int D = m->dim[0], Hh = m->rnn_hidden, F = m->ff[0], G = 4 * Hh, M = B * T;
    char p[64];
    snprintf(p, sizeof p, "encoder.encoder.layers.%d.", li);
    const float* bih = W(m, "%slstm.bias_ih_l0", p);
    const float* bhh = W(m, "%slstm.bias_hh_l0", p);
    float* gx = falloc((size_t)M * G);
    linear(gx, G, x, D, WT(m, G, D, "%slstm.weight_ih_l0", p), bih, M, D, G);
    const float* whh_t = WT(m, G, D, "%slstm.weight_hh_l0", p);
    const float* whr_t = WT(m, D, Hh, "%slstm.weight_hr_l0", p);
    float* y = falloc((size_t)M * D);
    float* g = falloc((size_t)B * G);
    float* hf = falloc((size_t)B * Hh);
    for (int t = 0; t < T; t++) {
        linear(g, G, h, D, whh_t, bhh, B, D, G);
#pragma omp parallel for schedule(static)
        for (int b = 0; b < B; b++) {
            const float* gxr = gx + ((size_t)b * T + t) * G;
            float* gr = g + (size_t)b * G;
            for (int j = 0; j < Hh; j++) {
                float ig = sigmoidf_(gxr[j] + gr[j]);
                float fg = sigmoidf_(gxr[Hh + j] + gr[Hh + j]);
                float gg = tanhf(gxr[2 * Hh + j] + gr[2 * Hh + j]);
                float og = sigmoidf_(gxr[3 * Hh + j] + gr[3 * Hh + j]);
                float cn = fg * c[(size_t)b * Hh + j] + ig * gg;
                c[(size_t)b * Hh + j] = cn;
                hf[(size_t)b * Hh + j] = og * tanhf(cn);
            }
        }
        linear(h, D, hf, Hh, whr_t, NULL, B, Hh, D);
        for (int b = 0; b < B; b++) memcpy(y + ((size_t)b * T + t) * D, h + (size_t)b * D, sizeof(float) * D);
    }
    add_inplace(x, y, (size_t)M * D);  /* src = lstm(src) + src */
    free(gx); free(y); free(g); free(hf);
    float* hid = falloc((size_t)M * F);
    linear(hid, F, x, D, WT(m, F, D, "%sfeed_forward.0.weight", p), W(m, "%sfeed_forward.0.bias", p), M, D, F);
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < (size_t)M * F; i++) hid[i] = double_swish(hid[i]);
    float* o = falloc((size_t)M * D);
    linear(o, D, hid, F, WT(m, D, F, "%sfeed_forward.4.weight", p), W(m, "%sfeed_forward.4.bias", p), M, F, D);
    add_inplace(x, o, (size_t)M * D);
    free(hid); free(o);
    basic_norm(x, x, W(m, "%snorm_final.eps", p)[0], M, D);
}

/* offline: zero initial states.  taps: 0 = embed output, 1+i = after layer i, -1 = encoder_out */
static int lstm_forward(const k2o_model* m, const float* xin, int B, int T, float* enc_out, int tap, float* tap_out, int64_t tap_cap,
                        int64_t* tap_n) {
    int Tp;
    float* x = lstm_embed(m, xin, B, T, &Tp);
    if (!x) return fail("lstm encoder: T=%d too short (need >= 9)", T);
    const int D = m->dim[0], L = m->nlayer[0], M = B * Tp;
    float* h = falloc((size_t)B * D);
    float* c = falloc((size_t)B * m->rnn_hidden);
    int rc = 0;
    for (int li = 0; li <= L; li++) {
        if (tap == li) {
            int64_t n = (int64_t)M * D;
            if (n > tap_cap) rc = fail("tap buffer too small");
            else { memcpy(tap_out, x, sizeof(float) * n); *tap_n = n; }
            free(x); free(h); free(c);
            return rc;
        }
        if (li == L) break;
        memset(h, 0, sizeof(float) * (size_t)B * D);
        memset(c, 0, sizeof(float) * (size_t)B * m->rnn_hidden);
        lstm_layer(m, li, x, B, Tp, h, c);
    }
    linear(enc_out, m->J, x, D, WT(m, m->J, D, "joiner.encoder_proj.weight"), W(m, "joiner.encoder_proj.bias"), M, D, m->J);
    free(x); free(h); free(c);
    return 0;
}
