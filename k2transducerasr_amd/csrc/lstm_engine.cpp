// LSTM transducer encoder on the gfx950 kernels: Model_type "lstm", offline through OfflineProjOfTransducer
// (OfflineRecognizer.cs:38-53) and streaming through OnlineProjOfLstm, whose per-stream states are
// h [layers, d_model] and c [layers, rnn_hidden_size] (OnlineProjOfLstm.cs:55-75).  The graph is icefall's
// lstm_transducer_stateless2: Conv2dSubsampling (no padding in time, two stride-2 convs: T' = ((T-3)//2 - 1)//2), then layers
// of { LSTM with projection; x += y; x += feed_forward(x); BasicNorm }.
//
// Per layer the input products of ALL frames are one MFMA GEMM ([B*T', D] x [D, 4H]); the recurrence is, per frame, the
// recurrent product h_{t-1}.W_hh^T (small-problem GEMM), the cell kernel and the projection hf.W_hr^T, which writes h_t straight
// into row t of the layer's output so that it is also the next frame's operand -- no copies.  (Launch-bound: three small
// launches per frame and layer; a persistent wavefront-over-layers kernel is the obvious next step.)
#include <cmath>

#include "engine.h"

namespace k2hip {

int Engine::lstm_out_frames(int T) const { return T < 9 ? 0 : ((T - 3) / 2 - 1) / 2; }

float* Engine::lstm_embed(const Ctx& c, const float* x, int B, int T, int* T_out) {
    const Model& m = *model_;
    const int F0 = 80, T1 = T - 2, T2 = (T1 - 3) / 2 + 1, F2 = (F0 - 3) / 2 + 1, T3 = (T2 - 3) / 2 + 1, F3 = (F2 - 3) / 2 + 1;
    const int D = m.cfg().dim[0];
    K2_REQUIRE(T >= 9 && T3 > 0, "encoder: %d input frames are too few (need >= 9)", T);
    Arena& ar = *c.arena;
    float* out = ar.take<float>((int64_t)B * T3 * D);
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
        g.cv_Fout = F3; g.cv_Tout = T3; g.cv_Tin = T2; g.cv_Fin = F2; g.cv_C = 32; g.cv_st = 2; g.cv_sf = 2;
        g.seg_len = 96; g.seg_stride = F2 * 32;
        gemm(c, g);
    }
    float* lin = ar.take<float>((int64_t)B * T3 * D);
    linear(c, a3, F3 * 128, m.w("encoder.encoder_embed.out.weight#fc"), m.w("encoder.encoder_embed.out.bias"), lin, D, B * T3, F3 * 128, D);
    basicnorm(c, lin, m.w("encoder.encoder_embed.out_norm.eps"), out, B * T3, D);
    ar.rewind(mark);
    *T_out = T3;
    return out;
}

// one RNNEncoderLayer in place on x [B*T, D]; h0 [B, D] (row stride ldh0) is the incoming hidden state, cst [B, Hh] the cell
// state (updated in place); *h_last receives the pointer / stride of the outgoing hidden state (row T-1 of the layer's y)
void Engine::lstm_layer(const Ctx& c, int li, float* x, const float* h0, int ldh0, float* cst, int B, int T, float* y) {
    const Model& m = *model_;
    const Config& cf = m.cfg();
    const int D = cf.dim[0], Hh = cf.rnn_hidden, F = cf.ff[0], G = 4 * Hh, M = B * T;
    char p[96];
    snprintf(p, sizeof p, "encoder.encoder.layers.%d.", li);
    auto w = [&](const char* suffix) { return m.w(std::string(p) + suffix); };
    Arena& ar = *c.arena;
    int64_t mark = ar.mark();
    float* gx = ar.take<float>((int64_t)M * G);
    linear(c, x, D, w("lstm.weight_ih_l0"), w("lstm.bias_ih_l0"), gx, G, M, D, G);
    float* gh = ar.take<float>((int64_t)B * G);
    float* hf = ar.take<float>((int64_t)B * Hh);
    const long long ldy = (long long)T * D;
    for (int t = 0; t < T; t++) {
        const float* hprev = t == 0 ? h0 : y + (long long)(t - 1) * D;
        const int ldh = t == 0 ? ldh0 : (int)ldy;
        linear(c, hprev, ldh, w("lstm.weight_hh_l0"), w("lstm.bias_hh_l0"), gh, G, B, D, G);
        lstm_cell(c, gx + (long long)t * G, (long long)T * G, gh, G, cst, hf, B, Hh);
        linear(c, hf, Hh, w("lstm.weight_hr_l0"), nullptr, y + (long long)t * D, (int)ldy, B, Hh, D);
    }
    add_inplace(c, x, y, (long long)M * D);  // src = lstm(src) + src
    float* hid = ar.take<float>((int64_t)M * F);
    linear(c, x, D, w("feed_forward.0.weight"), w("feed_forward.0.bias"), hid, F, M, D, F, ACT_DOUBLE_SWISH);
    linear(c, hid, F, w("feed_forward.4.weight"), w("feed_forward.4.bias"), x, D, M, F, D, ACT_NONE, x, D);
    basicnorm(c, x, w("norm_final.eps"), x, M, D);
    ar.rewind(mark);
}

// offline, layer by layer (three launches per frame and layer): the fallback when the layers' tensors are not evenly spaced
float* Engine::lstm_forward_seq(const Ctx& c, float* xe, int B, int T3, float* enc_out, int tap, float** tap_ptr, int* tap_dim) {
    const Model& m = *model_;
    const Config& cf = m.cfg();
    Arena& ar = *c.arena;
    const int D = cf.dim[0], Hh = cf.rnn_hidden;
    float* h0 = ar.take<float>((int64_t)B * D);
    float* cst = ar.take<float>((int64_t)B * Hh);
    float* y = ar.take<float>((int64_t)B * T3 * D);
    for (int li = 0; li <= cf.nlayer[0]; li++) {
        if (tap == li) {
            *tap_ptr = xe;
            *tap_dim = D;
            return nullptr;
        }
        if (li == cf.nlayer[0]) break;
        zero_floats(c, h0, (long long)B * D);
        zero_floats(c, cst, (long long)B * Hh);
        lstm_layer(c, li, xe, h0, D, cst, B, T3, y);
    }
    linear(c, xe, D, m.w("joiner.encoder_proj.weight"), m.w("joiner.encoder_proj.bias"), enc_out, cf.J, B * T3, D, cf.J);
    return enc_out;
}

// offline: zero initial states.  taps: 0 = embed output; 1+i = after layer i.
//
// Layer wavefront: layer l can work on frame t as soon as layer l-1 has finished frame t and layer l itself frame t-1, so at
// step s every layer l with 0 <= s-l < T' advances by one frame (t = s-l) -- T'+L-1 steps instead of T'.L.  The layers' weights
// sit in the model blob at one constant stride (same tensors, same order per layer), so each product of a step is ONE batched GEMM
// over the active layers (blockIdx.z = layer; operands, bias and output addressed by batch strides): 8 launches per step
// (x.W_ih, + h.W_hh, cell, projection, residual, feed-forward in / out, BasicNorm) against 3 per frame AND layer before.
float* Engine::lstm_forward(const Ctx& c, const float* x, int B, int T, int* Tp, int tap, float** tap_ptr, int* tap_rows, int* tap_dim) {
    const Model& m = *model_;
    const Config& cf = m.cfg();
    Arena& ar = *c.arena;
    const int Tpp = lstm_out_frames(T);
    K2_REQUIRE(Tpp > 0, "encoder: %d input frames are too few", T);
    float* enc_out = ar.take<float>((int64_t)B * Tpp * cf.J);
    int T3 = 0;
    float* xe = lstm_embed(c, x, B, T, &T3);
    if (tap_rows) *tap_rows = B * T3;
    *Tp = T3;
    const int D = cf.dim[0], Hh = cf.rnn_hidden, F = cf.ff[0], G = 4 * Hh, L = cf.nlayer[0];
    // constant layer stride of every per-layer tensor?
    static const char* kNames[] = {"lstm.weight_ih_l0", "lstm.weight_hh_l0", "lstm.bias_ih_l0", "lstm.bias_hh_l0", "lstm.weight_hr_l0",
                                   "feed_forward.0.weight", "feed_forward.0.bias", "feed_forward.4.weight", "feed_forward.4.bias",
                                   "norm_final.eps"};
    const float* base[10];
    long long LS = 0;
    bool even = !tunables().lstm_seq;
    for (int k = 0; k < 10 && even; k++) {
        base[k] = m.wf("encoder.encoder.layers.0.%s", kNames[k]);
        for (int l = 1; l < L && even; l++) {
            const long long d = m.wf("encoder.encoder.layers.%d.%s", l, kNames[k]) - m.wf("encoder.encoder.layers.%d.%s", l - 1, kNames[k]);
            if (LS == 0) LS = d;
            even = d == LS && d > 0;
        }
    }
    if (!even) return lstm_forward_seq(c, xe, B, T3, enc_out, tap, tap_ptr, tap_dim);

    const long long SY = (long long)B * T3 * D;
    float* Y = ar.take<float>((int64_t)(L + 1) * SY);        // Y[0] = embed output, Y[l+1] = output of layer l
    float* gates = ar.take<float>((int64_t)L * B * G);
    float* cst = ar.take<float>((int64_t)L * B * Hh);
    float* hf = ar.take<float>((int64_t)L * B * Hh);
    float* h = ar.take<float>((int64_t)L * B * D);
    float* x1 = ar.take<float>((int64_t)L * B * D);
    // the two N = d_model products (projection, feed_forward.4) have few output tiles but long K: split K four ways over
    // blockIdx.z as well (partials summed, in a fixed order, by the small kernel that consumes them)
    const int SP = (Hh % 128 == 0 && Hh >= 256) ? 4 : 1, SF = (F % 128 == 0 && F >= 256) ? 4 : 1;
    const long long PST = (long long)L * B * D;
    float* hp = ar.take<float>((int64_t)SP * PST);
    float* fp = ar.take<float>((int64_t)SF * PST);
    float* ffh = ar.take<float>((int64_t)L * B * F);
    if (!c.dry) K2_HIP(hipMemcpyAsync(Y, xe, sizeof(float) * (size_t)SY, hipMemcpyDeviceToDevice, c.stream));
    zero_floats(c, cst, (long long)L * B * Hh);
    zero_floats(c, h, (long long)L * B * D);
    const long long ldy = (long long)T3 * D;
    for (int s = 0; s < T3 + L - 1; s++) {
        const int lo = std::max(0, s - T3 + 1), hi = std::min(L - 1, s), n = hi - lo + 1;
        auto layers = [&](GemmArgs& g) {
            g.M = B; g.nb0 = n; g.nb1 = 1;
            g.W += lo * LS; g.sW0 = LS;
            if (g.bias) { g.bias += lo * LS; g.sBias0 = LS; }
        };
        {   // gates = x_t . W_ih^T + b_ih
            GemmArgs g;
            g.A = Y + lo * SY + (long long)(s - lo) * D; g.lda = (int)ldy; g.sA0 = SY - D;
            g.W = base[0]; g.ldw = D; g.bias = base[2];
            g.C = gates + (long long)lo * B * G; g.ldc = G; g.sC0 = (long long)B * G; g.N = G; g.K = D;
            layers(g);
            gemm(c, g);
        }
        {   // gates += h_{t-1} . W_hh^T + b_hh
            GemmArgs g;
            g.A = h + (long long)lo * B * D; g.lda = D; g.sA0 = (long long)B * D;
            g.W = base[1]; g.ldw = D; g.bias = base[3];
            g.C = gates + (long long)lo * B * G; g.ldc = G; g.sC0 = (long long)B * G; g.N = G; g.K = D;
            g.res = g.C; g.ldr = G; g.sR0 = g.sC0;
            layers(g);
            gemm(c, g);
        }
        lstm_cell_rows(c, gates + (long long)lo * B * G, cst + (long long)lo * B * Hh, hf + (long long)lo * B * Hh, n * B, Hh);
        {   // h_t = hf . W_hr^T (split-K partials)
            GemmArgs g;
            g.A = hf + (long long)lo * B * Hh; g.lda = Hh; g.sA0 = (long long)B * Hh;
            g.W = base[4]; g.ldw = Hh;
            g.C = hp + (long long)lo * B * D; g.ldc = D; g.sC0 = (long long)B * D; g.N = D; g.K = Hh / SP;
            layers(g);
            g.nb1 = SP; g.sA1 = Hh / SP; g.sW1 = Hh / SP; g.sC1 = PST;
            gemm(c, g);
        }
        // h_t = sum of the partials; x1 = x_t + h_t
        lstm_add_frame(c, Y, SY, hp + (long long)lo * B * D, PST, SP, h + (long long)lo * B * D, x1 + (long long)lo * B * D, n, B, T3, D, lo, s);
        {   // feed-forward
            GemmArgs g;
            g.A = x1 + (long long)lo * B * D; g.lda = D; g.sA0 = (long long)B * D;
            g.W = base[5]; g.ldw = D; g.bias = base[6]; g.act = ACT_DOUBLE_SWISH;
            g.C = ffh + (long long)lo * B * F; g.ldc = F; g.sC0 = (long long)B * F; g.N = F; g.K = D;
            layers(g);
            gemm(c, g);
        }
        {   // feed_forward.4 (split-K partials; bias, residual and BasicNorm in the consumer)
            GemmArgs g;
            g.A = ffh + (long long)lo * B * F; g.lda = F; g.sA0 = (long long)B * F;
            g.W = base[7]; g.ldw = F;
            g.C = fp + (long long)lo * B * D; g.ldc = D; g.sC0 = (long long)B * D; g.N = D; g.K = F / SF;
            layers(g);
            g.nb1 = SF; g.sA1 = F / SF; g.sW1 = F / SF; g.sC1 = PST;
            gemm(c, g);
        }
        lstm_norm_frame(c, x1 + (long long)lo * B * D, fp + (long long)lo * B * D, PST, SF, base[8], base[9], LS, Y, SY, n, B, T3, D, lo, s);
    }
    if (tap >= 0 && tap <= L) {
        *tap_ptr = Y + (long long)tap * SY;
        *tap_dim = D;
        return nullptr;
    }
    linear(c, Y + (long long)L * SY, D, m.w("joiner.encoder_proj.weight"), m.w("joiner.encoder_proj.bias"), enc_out, cf.J, B * T3, D, cf.J);
    return enc_out;
}

// streaming: one chunk (9 frames -> 1 encoder frame) for B streams whose h / c live in the pool; returns enc [B, 1, J]
float* Engine::lstm_chunk(const Ctx& c, const float* x, const int* d_slots, int B) {
    const Model& m = *model_;
    const Config& cf = m.cfg();
    Arena& ar = *c.arena;
    const int D = cf.dim[0], Hh = cf.rnn_hidden, L = cf.nlayer[0];
    float* enc = ar.take<float>((int64_t)B * cf.J);
    int T3 = 0;
    float* xe = lstm_embed(c, x, B, cf.chunk_T, &T3);
    K2_REQUIRE(T3 == 1, "internal: lstm chunk yields %d frames", T3);
    float* h0 = ar.take<float>((int64_t)B * D);
    float* cst = ar.take<float>((int64_t)B * Hh);
    float* y = ar.take<float>((int64_t)B * D);
    const long long stride = lay_.floats_per_stream;
    for (int li = 0; li < L; li++) {
        const long long off_h = (long long)li * D, off_c = (long long)L * D + (long long)li * Hh;
        gather_rows(c, online_pool_, stride, off_h, d_slots, h0, B, D);
        gather_rows(c, online_pool_, stride, off_c, d_slots, cst, B, Hh);
        lstm_layer(c, li, xe, h0, D, cst, B, 1, y);
        scatter_rows(c, online_pool_, stride, off_h, d_slots, y, D, B, D);
        scatter_rows(c, online_pool_, stride, off_c, d_slots, cst, Hh, B, Hh);
    }
    linear(c, xe, D, m.w("joiner.encoder_proj.weight"), m.w("joiner.encoder_proj.bias"), enc, cf.J, B, D, cf.J);
    return enc;
}

}  // namespace k2hip
