// Launchers for the gfx950 kernels.  Every launcher enqueues on ctx.stream and
// returns immediately; in a dry run (ctx.dry) nothing is launched -- the dry
// pass only sizes the arena.
#pragma once
#include "common.h"

namespace k2hip {

// Switches of the library (tunables.cpp): the K2HIP_* environment read once at the first model creation.  The shipped build keeps
// ONE path per operation plus the cross-check pairs the parity tests compare (an alternative kernel for the same math, a forced
// fallback, a size limit); the tuning probes exist only in a -DK2HIP_DEV build (make DEV=1), where K2HIP_DEV_SWITCH fields become
// ordinary members -- here they are compile-time constants and the code behind them folds away.
#ifdef K2HIP_DEV
#define K2HIP_DEV_SWITCH(name, def) int name = def
#else
#define K2HIP_DEV_SWITCH(name, def) static constexpr int name = def
#endif
struct Tunables {
    // ---- cross-check pairs and forced fallbacks (tests/)
    int no_glu_epilogue = 0;      // K2HIP_NO_GLU_EPILOGUE: conv modules' GLU in the depthwise kernel instead of the in_proj GEMM's epilogue
    int attn_long = 0;            // K2HIP_ATTN_LONG: two-pass attention scores for every length (the long-utterance form)
    int no_fused_av = 0;          // K2HIP_NO_FUSED_AV: attention apply + out_proj as two GEMMs
    int fused_vproj_min_t = 4;    // K2HIP_FUSED_VPROJ_MIN_T: the streaming self-attention's value projection inside k_attn_av_out from this many
                                  // chunk rows per stream on (128 streams, device time per tick 3.61 ms from 4 rows on, 3.62 from 8, 3.68 from 2)
    int no_fused_vproj = 0;       // K2HIP_NO_FUSED_VPROJ: ... as a GEMM launch of its own
    int no_fused_conv = 0;        // K2HIP_NO_FUSED_CONV: the streaming conv modules' GLU + chunk-causal depthwise conv as a launch of its own
                                  // instead of inside the in_proj GEMM's epilogue (round 5)
    int conformer_gemm_scores = 0;  // K2HIP_CONFORMER_GEMM_SCORES: two batched GEMMs + gather/softmax (the long-utterance form)
    int dw7_tiled = 0;            // K2HIP_DW7_TILED: the one-shot LDS-tiled 7x7 depthwise conv also for long inputs (default: the sliding LDS-DMA form)
    int lstm_seq = 0;             // K2HIP_LSTM_SEQ: layer-by-layer LSTM instead of the layer wavefront
    int greedy_one_part = 0;      // K2HIP_GREEDY_ONE_PART: one workgroup per stream in the search
    int greedy_parts = 0;         // K2HIP_GREEDY_PARTS: vocabulary slabs per stream (0 = automatic)
    int beam_launches = 0;        // K2HIP_BEAM_LAUNCHES: the modified beam search as 4 launches per frame (the large-vocabulary form) even when the one-kernel form applies
    int beam_parts = 0;           // K2HIP_BEAM_PARTS: 1 = one workgroup per stream in the one-kernel beam search even where two column slabs apply
    int beam_hyp_global = 0;      // K2HIP_BEAM_HYP_GLOBAL: the one-kernel beam search keeps its hypotheses in device memory even when they fit in LDS (the long-utterance form)
    int beam_trace = 0;           // K2HIP_BEAM_TRACE: the modified beam search records its per-frame selection (k2hip_debug.h: k2hip_debug_beam_trace)
    int search_rounds = -1;       // K2HIP_SEARCH_ROUNDS: 1 = every multi-stream search as rounds of joiner GEMMs (greedy_rounds),
                                  // 0 = always the persistent kernel (k_greedy), -1 = measured default
    int test_greedy_timeout = 0;  // K2HIP_TEST_GREEDY_TIMEOUT: test hook -- every parts > 1 search reports an exchange timeout, so the one-part retry runs
                                  // (1: without arming the engine's back-off, so that the next search is parted again; 2: as a real one)
    // ---- limits
    int decoder_table_mb = 1024;  // K2HIP_DECODER_TABLE_MB: build the all-contexts decoder table when it fits this many MiB (0 = never)
    int screen_min_v = 1024;      // K2HIP_SCREEN_MIN_V: vocabularies of at least this size get the f16 screening pass in the greedy search
                                  // (greedy.hip screen_round: exact tokens, ~2.5x fewer bytes per round); 0 = never
    int max_streams = 0;          // K2HIP_MAX_STREAMS: slots of the streaming state pool (0 = 256)
    // ---- tuning probes (-DK2HIP_DEV builds only)
    K2HIP_DEV_SWITCH(gemm_cfg, -1);         // K2HIP_GEMM_CFG: force one tile configuration
    K2HIP_DEV_SWITCH(xcd_panels, 0);        // K2HIP_XCD_PANELS: 1 = every GEMM's tiles as bands of M per XCD (rounds 1 - 3)
    K2HIP_DEV_SWITCH(dw1d_tt, 0);           // K2HIP_DW1D_TT: outputs per thread of the depthwise Conv1d (8 / 4 / 2; 0 = by grid size)
    K2HIP_DEV_SWITCH(greedy_stamps, 0);     // K2HIP_GREEDY_STAMPS: the persistent search reports where a round's time goes (stderr, synchronous)
    K2HIP_DEV_SWITCH(conformer_stamps, 0);  // K2HIP_CONFORMER_STAMPS: the Conformer scores kernel reports its phases (stderr, synchronous)
};
void tunables_init_from_env();            // idempotent; called by k2hip_model_create
const Tunables& tunables();
bool tunables_set(const char* env_name, int value);  // test hook (k2hip_debug_set_switch)

struct GemmStats {
    double flops = 0;       // algorithmic, 2*M*N*K per launch
    double total_flops = 0; // + non-GEMM matrix work (attention scores, convs)
    int launches = 0;
    float ms = 0;           // summed event time (instrumented runs only)
};

struct GemmLaunchRec {  // one row per GEMM launch of an instrumented call
    int M, N, K, batch, act, res, kind;  // kind: 0 plain, 1 conv gather, 2 [K,N] operand; +16 = LDS-DMA kernel
    float us;
};

struct GreedyLaunch;
struct Ctx {
    std::vector<GemmLaunchRec>* gemm_log = nullptr;  // instrumented runs only
    GreedyLaunch* greedy_rec = nullptr;  // greedy_loop notes its launch here, so that the engine can repeat it with one part per stream
    hipStream_t stream = nullptr;
    Arena* arena = nullptr;
    bool dry = false;
    bool one_part = false;    // the engine's searches recently timed out waiting for their other column slabs (a GPU shared with other
                              // handles or processes): one workgroup per stream until the back-off runs out (Engine::note_search)
    bool instrument = false;  // bracket each GEMM launch with events
    GemmStats* stats = nullptr;
    // instrumented runs: each GEMM launch records a (start, stop) pair from this pool
    // WITHOUT synchronising; the engine reads the pairs after the call has drained,
    // so the launches run back to back exactly as in an un-instrumented call.
    std::vector<hipEvent_t>* evpool = nullptr;
    int* evused = nullptr;
    hipEvent_t next_event() const {
        if ((size_t)*evused == evpool->size()) {
            hipEvent_t e;
            K2_HIP(hipEventCreate(&e));
            evpool->push_back(e);
        }
        return (*evpool)[(*evused)++];
    }
    // algorithmic work is tallied once, in the dry (sizing) pass
    void add_flops(double gemm_fl, double other_fl, int launches) const {
        if (stats && dry) {  // stats may be null (tuning hook)
            stats->flops += gemm_fl;
            stats->total_flops += gemm_fl + other_fl;
            stats->launches += launches;
        }
    }
};

enum Act : int { ACT_NONE = 0, ACT_SWOOSH_L = 1, ACT_SWOOSH_R = 2, ACT_TANH = 3, ACT_SIGMOID = 4, ACT_RELU = 5, ACT_DOUBLE_SWISH = 6 };

// C[M,N] = act(A[M,K] . W^T + bias) (+ residual)     (fp32 MFMA, exact-f32 products)
//   A: row-major, K contiguous, lda % 4 == 0; rows may be gathered (implicit conv)
//   W: [N,K] row-major (torch Linear layout), or [K,N] when w_kn
struct GemmArgs {
    const float* A = nullptr;
    const float* W = nullptr;
    const float* bias = nullptr;
    const float* res = nullptr;
    float* C = nullptr;
    int M = 0, N = 0, K = 0;
    int lda = 0, ldw = 0, ldc = 0, ldr = 0;
    int act = ACT_NONE;
    int xcd_panels = 0;  // tile order: 0 = N panels per XCD chosen by PN M + (8 / PN) N (gemm.hip xcd_tile); 1 = a band of M per XCD (K2HIP_XCD_PANELS)
    int act_cols = 0;  // > 0: the activation applies to output columns < act_cols only (two Linears of one input fused in one launch)
    int w_kn = 0;
    // batching over blockIdx.z = z0 + nb0 * z1
    int nb0 = 1, nb1 = 1;
    // device flag: the launch does nothing when *skip_if_zero == 0 (rounds of the batched search after every stream has finished)
    const int* skip_if_zero = nullptr;
    const int* wz_map = nullptr;  // [K,N] form only: batch z0 reads W + wz_map[z0] * sW0 (stream slots of the state pool)
    long long sA0 = 0, sA1 = 0, sW0 = 0, sW1 = 0, sC0 = 0, sC1 = 0, sR0 = 0, sR1 = 0;
    long long sBias0 = 0;  // bias + z0 * sBias0 (batched launches over layers, each with its own bias)
    // implicit-conv gather of A over an NHWC tensor [B, Tin, Fin, C]:
    //   row r = (b*Tout + t)*Fout + f  ->  base = ((b*Tin + t*st)*Fin + f*sf)*C
    //   k -> (k / seg_len) * seg_stride + k % seg_len
    int cv_Fout = 0, cv_Tout = 0, cv_Tin = 0, cv_Fin = 0, cv_C = 0, cv_st = 1, cv_sf = 1;
    int seg_len = 0, seg_stride = 0;
    // epilogue extras (after bias, activation and residual):
    //   mul:  v *= mul[row*ldm + col]                      (batched like C: + z0*sM0 + z1*sM1)   -- NonlinAttention's output gate
    //   byp:  v = o + (v - o) * byp_scale[col], o = byp_orig[row*ld_orig + col]                 -- Zipformer bypass module
    //   res_div > 1: residual row = row / res_div (K hypothesis rows share their stream's encoder frame)
    //   act_after_res: v = act(acc + bias + res) instead of act(acc + bias) + res          -- the joiner's tanh(enc + dec)
    int res_div = 1;
    int act_after_res = 0;
    // glu: the N output columns are blocks of 32 = 16 values | their 16 gates (weights interleaved at load, "#glu"); the epilogue
    // writes value * sigmoid(gate) to N / 2 columns of C (ldc counts those).  32-column C/D layouts only (not the skinny kernel).
    int glu = 0;       // 1: value * sigmoid(gate) (conv modules), 2: value * tanh(gate) (NonlinAttention)
    int glu_cols = 0;  // > 0: only the first glu_cols GEMM columns are (value | gate) blocks; the rest pass through behind them
    const float* mul = nullptr;
    int ldm = 0;
    long long sM0 = 0, sM1 = 0;
    const float* byp_orig = nullptr;
    const float* byp_scale = nullptr;
    int ld_orig = 0;
    // Streaming conv module (round 5, gemm_glu_causal_conv below): the in_proj GEMM over "#glu"-interleaved weights finishes with the
    // GLU AND the chunk-causal depthwise convolution of online.hip's k_glu_causal_conv_reg -- the Tc rows of a stream sit in one tile, a
    // 32-column block holds 16 channels' values and gates, so everything the convolution of those channels needs is in the workgroup
    // (+ the stream's K / 2 cached frames from the state pool, which it also advances).  C = y [M, N / 2].
    float* cf_pool = nullptr;          // != nullptr: the fused form
    long long cf_stride = 0, cf_off = 0;
    const int* cf_slots = nullptr;
    const float *cf_wc = nullptr, *cf_bc = nullptr, *cf_ww = nullptr, *cf_bw = nullptr, *cf_sc = nullptr;
    int cf_Tc = 0, cf_K = 0;
    unsigned long long* dbg = nullptr;  // tuning only: in-kernel s_memtime stamps of the ring kernel, [workgroup][wave][64]
    int ablate = 0;  // tuning only: 1 = skip in-loop global loads, 2 = skip MFMAs, 4 = skip epilogue stores
};
void gemm(const Ctx& ctx, const GemmArgs& a);
// Conv module of a streaming Zipformer2 layer, first two thirds in ONE launch: y [M, D] = SwooshR(chunk-causal depthwise conv(GLU(x Wg^T + bg)))
// with the streams' conv caches (pool slot + off, [D][K / 2]) read and advanced in place; M = B Tc rows, stream-major.  wg / bg: the
// "#glu" row-interleaved in_proj [2 D, D]; wc / bc / ww / bw / sc: causal_conv, chunkwise_conv and chunkwise_conv_scale as
// glu_causal_conv takes them.  Returns false (nothing launched, nothing tallied) when the shape has no fused form -- the caller then
// runs linear + glu_causal_conv.  The convolution's sums are formed in k_glu_causal_conv_reg's order; the in_proj product in the order of
// this launch's tile form (the dispatcher's own choice for most of these shapes): equal to the two launches up to float rounding.
bool gemm_glu_causal_conv(const Ctx& ctx, const float* x, const float* wg, const float* bg, float* pool, long long slot_stride, long long off,
                          const int* slots, const float* wc, const float* bc, const float* ww, const float* bw, const float* sc, float* y, int B,
                          int Tc, int D, int K);
void debug_force_gemm_cfg(int cfg);  // tuning hook: -1 = automatic
void debug_pipe_shape(int cfg, int M, int N, int* n_wg, int* waves);  // grid and waves of pipe cfg (>= 2000) on a shape
void debug_ring_shape(int idx, int* bm, int* bn, int* waves);  // tile and waves of ring table entry idx
// convenience: plain Linear  C = act(A W^T + b) (+res)
void linear(const Ctx& ctx, const float* A, int lda, const float* W, const float* bias, float* C, int ldc, int M, int K,
            int N, int act = ACT_NONE, const float* res = nullptr, int ldr = 0);

// A per-stream cache kept as a RING of KL = L + Tc rows inside the stream's slot of the state pool (streaming Zipformer2): the
// chunk's Tc new rows overwrite the Tc oldest ones in place, nothing is rolled.  Chunk n of a stream writes its row r to ring row
// (head + r) % KL, head = (n * Tc) % KL; in the order of the reference's [cache ; chunk] concatenation (j = 0 oldest .. KL - 1
// newest) row j is at (head + Tc + j) % KL.
struct RingRef {
    float* pool = nullptr;         // state pool
    long long slot_stride = 0;     // floats per stream
    long long off = 0;             // float offset of this ring inside a slot
    const int* slots = nullptr;    // [B] slot of each stream of the step
    const int* chunks = nullptr;   // [B] chunks each stream has decoded before this step
};

// ---- attention ------------------------------------------------------------
// qkp: [B*T, ld] rows = (q[H*32] | k[H*32] | p[H*4]); pp: [2T-1, H*4];
// aw out: [H][B][T][Tp] (Tp = T rounded up to 4, pad columns zeroed)
// qh: query / key head size (16, 24 or 32); koff0 / poff0: float offset of head 0's key / positional query in a row (default: the
// Zipformer2 row above; Zipformer v1: q [A] | k [A] | v [A/2] | p [H*4])
void attn_scores_softmax(const Ctx& ctx, const float* qkp, int ld, const float* pp, float* aw, int B, int T, int Tp, int H, int qh = 32,
                         int koff0 = -1, int poff0 = -1);

// fused x += out_proj(concat_h(aw_h . v_h)) + bias  (SelfAttention after its value projection); aw [H][B][T][Tp] over KL keys,
// v [B*KL, H*vh], wout [D, H*vh];
// returns false (nothing launched, nothing tallied) when the shape does not fit the kernel -- the caller then takes the GEMM path
bool attn_av_out(const Ctx& ctx, const float* aw, const float* v, const float* wout, const float* bias, float* x, int B, int T, int KL, int Tp,
                 int H, int vh, int D);
// streaming ring form: values of the left context come from the ring (rows in ring order, like aw's columns); the chunk's value
// rows (newrows [B*T, H*vh]) are written into the ring by the kernel itself before it reads them.  T <= 16 (one row strip).
bool attn_av_out_ring(const Ctx& ctx, const float* aw, const RingRef& vals, const float* newrows, const float* wout, const float* bias, float* x,
                      int B, int T, int KL, int Tp, int H, int vh, int D);
void attn_proj_av_out_ring(const Ctx& ctx, const float* aw, const RingRef& vals, const float* xin, const float* win, const float* bin,
                           const float* wout, const float* bias, float* xout, int B, int T, int KL, int Tp, int H, int vh, int D);

// NonlinAttention.streaming_forward after its in_proj: hid [B*T, ldh] rows = s | x | y (Hc columns each); x * tanh(s) of the chunk's
// rows goes into the cache ring, then ctx = (aw_head0 . ring) * y.  wout != nullptr: x[B*T, D] += out_proj(ctx) in the same launch
// (every stream's workgroup pulls the whole out_proj matrix through its CU: only worth it for narrow stacks); wout == nullptr:
// x [B*T, Hc] = ctx, and the caller runs out_proj as a GEMM over all streams' rows.  T <= 16.
void nonlin_av_out_ring(const Ctx& ctx, const float* aw, const RingRef& cache, const float* hid, int ldh, const float* wout, const float* bias,
                        float* x, int B, int T, int KL, int Tp, int Hc, int D);

// ---- elementwise / small -----------------------------------------------------
// packed: all streams' features back to back; d_off/d_len: per-stream start and float count (device)
void pad_logfloor(const Ctx& ctx, const float* packed, const long long* d_off, const long long* d_len, float* out, int B,
                  long long L);
void pad_logfloor_dense(const Ctx& ctx, const float* feats, long long n_each, float* out, int B, long long L);
void conv0_swoosh(const Ctx& ctx, const float* x, const float* w, const float* b, float* y, int B, int T, int F);
// Conformer Conv2dSubsampling conv.0: 1->8 ch, 3x3, padding 1 in time and frequency, DoubleSwish; y: NHWC [B,T,F,8]
void conv0_pad1_dswish(const Ctx& ctx, const float* x, const float* w, const float* b, float* y, int B, int T, int F);
// depthwise 7x7 over [B,Tin,F,C] -> [B,Tout,F,C]; time taps read in[t + kt - tpad] (zero outside [0,Tin)).
// offline ConvNeXt: Tin = Tout, tpad = 3; streaming: Tin = Tout + 6, tpad = 0 (left cache + right context supply the taps)
void dwconv7x7(const Ctx& ctx, const float* x, const float* w_kc, const float* b, float* y, int B, int Tin, int Tout, int tpad,
               int F, int C);
void biasnorm(const Ctx& ctx, const float* x, const float* bias, const float* log_scale, float* y, int M, int D);
// y = orig + (biasnorm(x) - orig) * scale     (layer tail: norm + bypass)
void biasnorm_bypass(const Ctx& ctx, const float* x, const float* orig, const float* nbias, const float* log_scale,
                     const float* scale, float* y, int M, int D);
// the same for the last layer of an input-rate stack in front of a downsampled one, and that stack's SimpleDownsample of the result
// (xd2 [B * ceil(T / ds2), D2]) in the same launch
void biasnorm_bypass_downsample(const Ctx& ctx, const float* x, const float* orig, const float* nbias, const float* log_scale, const float* scale,
                                float* y, const float* bias2, float* xd2, int B, int T, int D, int ds2, int D2);
// what a layer's last launch also has to produce when it is that layer (null bias2: nothing)
struct LayerTail {
    const float* bias2 = nullptr;
    float* xd2 = nullptr;
    int ds2 = 1, D2 = 0;
};
void bypass(const Ctx& ctx, const float* orig, const float* x, const float* scale, float* y, int M, int D);
void glu_sigmoid(const Ctx& ctx, const float* x, float* y, int M, int D);          // y = x[:, :D] * sigmoid(x[:, D:])
void tanh_gate(const Ctx& ctx, const float* x, float* y, int M, int Hc);           // y = x[:, Hc:2Hc] * tanh(x[:, :Hc])
void mul_cols(const Ctx& ctx, float* a, const float* x, int ldx, int col0, int M, int N);  // a[m,n] *= x[m, col0+n]
// y = SwooshR(dwconv1d(glu(x2)) + b);  x2: [B,T,2D] value|gate
// the same on an input whose GLU has already run (x: [B,T,D]): GemmArgs.glu
void dwconv1d_swoosh(const Ctx& ctx, const float* x, const float* w_kd, const float* b, float* y, int B, int T, int D, int K);
void glu_dwconv1d_swoosh(const Ctx& ctx, const float* x2, const float* w_kd, const float* b, float* y, int B, int T, int D,
                          int K);
// same with DoubleSwish (Conformer ConvolutionModule)
void glu_dwconv1d_dswish(const Ctx& ctx, const float* x2, const float* w_kd, const float* b, float* y, int B, int T, int D,
                          int K);
// LSTM transducer (lstm.hip)
void conv0_nopad_dswish(const Ctx& ctx, const float* x, const float* w, const float* b, float* y, int B, int T, int F);
// gates i,f,g,o = gx[b] + gh[b] ([4*Hh] each, row strides ldgx / ldgh): c = f*c + i*g; hf = o * tanh(c)
void lstm_cell(const Ctx& ctx, const float* gx, long long ldgx, const float* gh, int ldgh, float* c, float* hf, int B, int Hh);
void add_inplace(const Ctx& ctx, float* a, const float* b, long long n);
// layer-wavefront form (lstm_engine.cpp): rows r = z*B + b over the n active layers z (layer lo + z works on frame s - lo - z)
//   gates [n*B, 4Hh] (both biases already added by the GEMMs) -> c [n*B, Hh] updated, hf [n*B, Hh]
void lstm_cell_rows(const Ctx& ctx, const float* gates, float* c, float* hf, int rows, int Hh);
//   h[z][b] = sum_q hp[q][z][b] (the S split-K partials of the projection, `pstride` floats apart); x1[z][b] = Y[lo+z][b*T + t_z] + h
//   (Y: [L+1][B*T][D], layer stride SY)
void lstm_add_frame(const Ctx& ctx, const float* Y, long long SY, const float* hp, long long pstride, int S, float* h, float* x1, int n, int B,
                    int T, int D, int lo, int s);
//   x2 = x1 + b2 + sum_q fp[q] (split-K partials of feed_forward.4); Y[lo+z+1][b*T + t_z] = BasicNorm(x2); b2 / eps of layer l at
//   b2_0 + l*lstride / eps0 + l*lstride
void lstm_norm_frame(const Ctx& ctx, const float* x1, const float* fp, long long pstride, int S, const float* b2_0, const float* eps0,
                     long long lstride, float* Y, long long SY, int n, int B, int T, int D, int lo, int s);
// rows of `width` floats between a [B, width] work buffer and the per-stream pool slots (pool + slot*stride + off)
void gather_rows(const Ctx& ctx, const float* pool, long long slot_stride, long long off, const int* slots, float* out, int B, int width);
void scatter_rows(const Ctx& ctx, float* pool, long long slot_stride, long long off, const int* slots, const float* in, int ldin, int B, int width);
// BasicNorm: y = x * (mean(x^2) + exp(log_eps))^-0.5   (in place allowed)
void basicnorm(const Ctx& ctx, const float* x, const float* log_eps, float* y, int M, int D);
// Conformer rel-pos attention helpers (conformer.hip)
//   qkv [M, 3D] -> qu = q*dk^-0.5 + pos_bias_u, qv = q*dk^-0.5 + pos_bias_v   ([M, D] each)
void conformer_qprep(const Ctx& ctx, const float* qkv, const float* bias_u, const float* bias_v, float* qu, float* qv, int M,
                     int D, float scaling, int ldq = 0 /* row stride of q; 0 = 3*D */);
// streaming (OnlineProjOfConformer): keys = [left cache ; chunk]; w[i,j] = softmax_j(ac[i,j] + bd[i, Tc-1-i+j]) over KL = left + Tc keys,
// left slot j masked (-inf) while plen[b] <= left-1-j; rows of stream b are z = b*H + h
void conformer_softmax_shift_stream(const Ctx& ctx, float* ac, const float* bd, const long long* plen, int B, int H, int Tc, int left,
                                    int KLp, int NPp);
void slice_rows(const Ctx& ctx, const float* in, float* out, int B, int Tin, int row0, int Tout, int D);  // out[b] = in[b][row0 : row0+Tout]
// y[b,t,:] = DoubleSwish(bias + sum_k w_kd[k] * cat[b, t+k, :])  (valid depthwise conv over [K-1 cached ; chunk] frames)
void dwconv_valid_dswish(const Ctx& ctx, const float* cat, const float* w_kd, const float* bias, float* y, int B, int Tc, int D, int K);
//   ac[z][i][j] (ld Tp) <- softmax_j(ac[z][i][j] + bd[z][i][T-1-i+j]) (rel_shift in gather form); pad columns zeroed
void conformer_softmax_shift(const Ctx& ctx, float* ac, const float* bd, int Z, int T, int Tp, int NPp);
// fused form: aw[b*H+h][i][:] = softmax_j((q+u)_i.k_j + (q+v)_i.p[T-1-i+j]) straight from qu / qv [B*T, D], k rows (row stride ldk) and the
// projected positional table pp [2T-1, D]; false = shape not covered (caller takes the GEMM + conformer_softmax_shift form)
bool conformer_scores_softmax(const Ctx& ctx, const float* qu, const float* qv, const float* kmat, int ldk, const float* pp, float* aw, int B, int H,
                              int T, int Tp, int D, int ldq = 0, const float* bias_u = nullptr, const float* bias_v = nullptr, float scaling = 1.0f);
// Din / Dorig > 0: the input rows are that wide and are zero-extended / truncated to D on the fly (convert_channels folded in)
void downsample(const Ctx& ctx, const float* x, const float* bias, float* y, int B, int T, int D, int ds, int Din = 0);
void upsample_combine(const Ctx& ctx, const float* orig, const float* xd, const float* scale, float* y, int B, int T,
                      int Td, int D, int ds, int Dorig = 0);
// the same and the NEXT stack's downsample of its result in one launch: y as above, xd2[b, t2, :D2] = sum_k softmax(bias2)_k y[b, ds2 t2 + k, :]
void upsample_combine_downsample(const Ctx& ctx, const float* orig, const float* xd, const float* scale, float* y, const float* bias2,
                                 float* xd2, int B, int T, int Td, int D, int ds, int Dorig, int D2, int ds2);
void convert_channels(const Ctx& ctx, const float* x, float* y, int M, int Din, int Dout);
// Zipformer2._get_full_dim_output without the concatenated tensor: columns [col1[i-1], col1[i]) of the full-width row live in src[i]
// (row width ld[i]); downsample_full = that gather + SimpleDownsample(ds) in one launch
struct FullDimSegs {
    const float* src[8];
    int ld[8], col1[8], n = 0;
    // segment 0 (the last stack's output) formed on the fly instead of read, when that stack is downsampled: its out_combiner
    // orig + (upsample(xd) - orig) * scale (k_upsample_combine's expression) -- the tensor itself is never written
    const float* lz_orig = nullptr;   // [B*T, lz_Do] (zero-extended / truncated to the stack's width)
    const float* lz_xd = nullptr;     // [B*lz_Td, ld[0]]
    const float* lz_scale = nullptr;
    int lz_Td = 0, lz_ds = 1, lz_Do = 0;
};
void downsample_full(const Ctx& ctx, const FullDimSegs& segs, const float* bias, float* y, int B, int T, int D, int ds);
void copy_cols(const Ctx& ctx, const float* x, int ldx, int xcol0, float* y, int ldy, int ycol0, int M, int n);

// ---- fbank -----------------------------------------------------------------------
struct FbankArgs {
    const float* samples;  // device
    long long n_samples;   // per utterance
    long long utt_stride;  // samples between utterances
    int n_utts;
    long long n_frames;    // per utterance
    float* feats;          // [n_utts, n_frames, 80]
    const float* window;
    const float* melw;
    int frame_len, frame_shift;
    float preemph, input_scale;
    int remove_dc;
    const float* melrange = nullptr;  // [num_bins][2] extent of each mel filter's non-zero weights (model table)
};
void fbank(const Ctx& ctx, const FbankArgs& a);
// dst[b][0 .. nmax) = src[b][0 .. n[b]) followed by zeros; src[b] may be pinned host memory (read in place over PCIe), 16 B aligned
void gather_samples(const Ctx& ctx, const float* const* src, const long long* n, float* dst, int B, long long nmax);

// ---- decoder / joiner / greedy ----------------------------------------------------
struct DecJoinW {
    const float* emb;      // [V, DD]
    const float* conv;     // [DD, cpg, ctx] (cpg <= 4) or k-major [cpg*ctx][DD] (cpg > 4)
    int cpg;               // decoder conv input channels per group
    // groups = 1 only: per-token conv contributions P[tap][v][co] = sum_ci conv[co][ci][tap] * emb[v][ci]  ([2][V][DD]);
    // the conv of a context is then relu(P[0][y0] + P[1][y1]) -- no GEMV inside the search loops
    const float* ptab = nullptr;
    const float* dproj_kn; // [DD, J]
    const float* dproj_b;  // [J]
    const float* out_kn;   // [J, Vp]
    const float* out_b;    // [V]
    // small vocabularies only: decoder(y0, y1) of EVERY context, row (y0 + 1) V + y1 of [(V + 1) V][J], y0 = -1 .. V-1 (decoder_table);
    // the search kernels then read a row where they would run the decoder
    const float* dec_table = nullptr;
    // large vocabularies only (model.cpp add_repacks): output_linear as f16 in MFMA fragment order + the per-column bound on
    // |f16 product - f32 logit| -- the greedy search's screening pass (greedy.hip screen_round); null = every round sweeps in f32
    const void* out_h16 = nullptr;
    const float* out_eps = nullptr;
    const float* out_vj = nullptr;   // output_linear in its torch layout [V][J]: a candidate column's weights in one 2 KB piece (the re-check)
    int V, Vp, DD, J, ctx;
};
void decoder(const Ctx& ctx, const DecJoinW& w, const long long* y, int N, float* dec_out);
// out[2][J] = the search kernel's own decoder routine on [-1, blank] and [blank, blank] (bit-identical to what k_greedy computes itself)
void decoder_start_contexts(const Ctx& ctx, const DecJoinW& w, float* out);
// table[(V + 1) V][J]: the search kernels' decoder routine on every context (bit-identical to what they compute themselves)
void decoder_table(const Ctx& ctx, const DecJoinW& w, float* table);
// out[N][J] = that routine on the contexts y[N][2], computed (the table is not consulted): what the table's rows are checked against
void decoder_rows_wide(const Ctx& ctx, const DecJoinW& w, const long long* y, int N, float* out);
void tanh_add(const Ctx& ctx, const float* enc, const float* dec, int dec_stride, float* y, int N, int J);
// row argmax with the reference tie-break (later index wins) -> emit flag (token not in {0,2} [,1])
void argmax_rows(const Ctx& ctx, const float* logits, int ld, int N, int V, int* tok);
// ---- CTC (zipformer2ctc models) ------------------------------------------------------------------
void log_softmax_rows(const Ctx& ctx, float* x, int M, int V);  // in place
// Array.IndexOf(row, row.Max()): the FIRST index of the maximum (OfflineRecognizer.cs:388)
void argmax_first_rows(const Ctx& ctx, const float* logits, int ld, int N, int V, int* tok);
// per stream: drop blanks and repeats (prev_id = -1), timestamp = t + frame_off[b]; trail[b] = blank frames at the end,
// any[b] = 1 if some frame was non-blank (OfflineRecognizer.cs:383-408)
void ctc_collapse(const Ctx& ctx, const int* tok, int B, int Tp, const int* frame_off, long long* tokens, int* timestamps,
                  int* n_tokens, int max_tokens, int* trail, int* any, int* overflow);
// t0 = first frame at which any stream emits under the initial context
void first_emit_frame(const Ctx& ctx, const int* tok, int B, int Tp, int skip1, int* t0);
struct GreedyArgs {
    const float* enc;   // [B, Tp, J]
    int B, Tp;
    const int* t0;      // device scalar (batch quirk) or nullptr (single path)
    int skip1;          // online filter also skips id 1
    int max_sym;        // 1000 for the single path, else INT_MAX
    long long* tokens;  // [B, max_tokens]
    int* timestamps;    // [B, max_tokens]
    int* n_tokens;      // [B]
    int max_tokens;
    int* overflow;      // device flag
    // online loop: per-stream starting context (stream.Hyp, OnlineRecognizer.cs:109,122-126); null = offline
    const long long* init_ctx = nullptr;  // [B][2]
    // vocabulary-parallel form (set by greedy_loop): `parts` workgroups per stream, each sweeping a slab of the joiner
    // matrix; per round they exchange their per-frame (max, argmax) through tagged 8-byte granules
    int parts = 1;
    unsigned long long* gran = nullptr;  // [B][2 (round parity)][parts][GF][2], zeroed per call
    unsigned long long* gran2 = nullptr;  // [B][2 (emission parity)][J]: the parts' slices of a decoder update, zeroed per call
    // decoder outputs of the two start contexts [-1, blank] and [blank, blank] ([2][J], from decoder_start_contexts): constants of
    // the model, so every workgroup of every batch loads them instead of running the decoder twice (null: computed in the kernel)
    const float* dec_init = nullptr;
    // tuning only (K2HIP_GREEDY_STAMPS): workgroup 0 accumulates s_memrealtime (100 MHz) differences per phase of a round here --
    // [0] rounds, [1] activations, [2] screen tiles, [3] candidate scan, [4] re-check, [5] sweep passes, [6] publish + exchange,
    // [7] decision + decoder update
    unsigned long long* stamps = nullptr;
};
void greedy_loop(const Ctx& ctx, const DecJoinW& w, const GreedyArgs& a);
bool greedy_loop_screens(const DecJoinW& w, int B, bool streaming, bool one_part);  // would its rounds use the f16 screen (large vocabulary, slab fits)?
// The vocabulary-parallel search waits on its sibling workgroups (bounded spins; a timeout raises *overflow = 2).  All B x parts
// workgroups must be resident together for that, which a GPU shared with other processes or models does not promise.  The engine
// therefore keeps the launch (inputs are read-only, outputs are rewritten from scratch) and, on a timeout, runs it again with ONE
// workgroup per stream -- no inter-workgroup wait, same tokens -- instead of failing the call.
// arguments of the one-kernel modified beam search (beam.hip k_beam_loop); here because a repeatable launch is kept in GreedyLaunch
struct BeamLoopArgs {
    const float* enc;   // [B, Tp, J]
    int Tp, K, cap;
    long long* tokens;
    int* timestamps;
    int* n_tokens;
    float* scores;
    int max_tokens;
    int* overflow;
    // long utterances: the hypotheses' token / timestamp arrays do not fit in LDS beside the logits and live in device memory,
    // [B][2][K][cap] each (a workgroup reads back only what it wrote itself: one CU, one L1); null = in LDS
    int* ys_g;
    int* ts_g;
    // two column slabs per stream (beam <= 4, 256 < V <= 512): workgroup 2 b + p sweeps chunk p and the two exchange their 4 x 256
    // logits per frame as tagged granules xg[b][frame parity][slab][4][256]; null = one workgroup per stream
    unsigned long long* xg;
    int* trace = nullptr;   // debug tap [B][Tp][2 K + 1] (beam_step_body) or null
};
struct GreedyLaunch {
    bool valid = false;  // a launch with inter-workgroup waits (parts > 1 / two beam slabs) that can be repeated without them
    DecJoinW w;
    GreedyArgs a;        // (beam: only B and overflow are meaningful)
    bool beam = false;   // the launch was k_beam_loop with two slabs per stream: ba, beam_lds
    BeamLoopArgs ba;
    size_t beam_lds = 0;
};
void beam_relaunch_one_slab(hipStream_t stream, const GreedyLaunch& rec);
void greedy_relaunch_one_part(hipStream_t stream, const GreedyLaunch& rec);
// The same search as batched ROUNDS instead of one persistent workgroup pair per stream: every round evaluates the next S frames
// of every stream under the stream's current context with ONE joiner GEMM over all B x S rows, then a per-stream step accepts
// frames up to and including the first emission, updates the context, runs the decoder if it changed and forms the next
// window's tanh(enc + dec) rows.  Exactly the frame-by-frame loop (frames behind an emission are re-evaluated in the next round);
// max(Tp) rounds are enqueued up front, rounds after the last stream has finished return at their first instruction.
// out_w: joiner.output_linear.weight [V, J] (torch layout, for the GEMM).  Batch / online semantics only (no max_sym cap).
void greedy_rounds(const Ctx& ctx, const DecJoinW& w, const float* out_w, const GreedyArgs& a);

// ---- modified beam search (beam.hip) ------------------------------------------------------------
constexpr int kMaxBeam = 8;
struct BeamState {       // device arrays; hypotheses double-buffered by frame parity
    int K, cap;
    int* ys;             // [2][B][K][cap] tokens without the ctx-blank prefix
    int* ts;             // [2][B][K][cap]
    int* n;              // [2][B][K]
    float *lp, *lp_next; // [B][K] hypothesis log-probs (-inf = empty slot)
    long long *ctx, *ctx_next;  // [B][K][2] decoder inputs
    int *nhyp, *nhyp_next;      // [B]
};
struct BeamArgs {
    const float* enc;    // [B, Tp, J]
    const float* out_w;  // joiner.output_linear.weight [V, J] (torch layout, for the GEMM)
    const float* dproj_w;  // joiner.decoder_proj.weight [J, DD] (torch layout)
    int B, Tp, beam;
    long long* tokens;   // [B, max_tokens] best hypothesis
    int* timestamps;
    int* n_tokens;
    float* scores;       // [B] log_prob of the best hypothesis (may be null)
    int max_tokens;
    int* overflow;
    int* trace = nullptr;  // debug tap [B][Tp][2 beam + 1] or null (k2hip_debug.h: k2hip_debug_beam_trace)
};
void beam_search(const Ctx& ctx, const DecJoinW& w, const BeamArgs& a);

// ---- streaming (online.hip): device-resident per-stream caches indexed by slot ------------------
// ConvNeXt.streaming_forward's data movement in one launch: cat[b] = [cached_left_pad ; x], the cache advanced to x's frames Tc-3 .. Tc-1,
// byp[b] = x[b, :Tc] (the bypass operand)
void convnext_cat(const Ctx& ctx, const float* a3, float* pool, long long slot_stride, long long embed_off, const int* slots, float* cat,
                  float* byp, int B, int T3, int Tc, int F, int C);
// tanh_gated: the new rows are formed as x[width + c] * tanh(x[c]) from rows of >= 2*width floats (NonlinAttention's gated input)
void cat_shift(const Ctx& ctx, float* pool, long long slot_stride, long long off, const int* slots, const float* newrows,
               int ldn, float* cat, int B, int L, int Tc, int width, bool tanh_gated = false);
// ring form: the keys of the left context are read from the ring, the chunk's own key rows (columns [32 H, 64 H) of qkp) are
// written into it first; aw columns are in RING order (column p = the key stored in ring row p), which is also the order the
// value rings are read in -- a weighted sum does not care.
// the same with `keep_back` newest rows kept OUT of the cache (streaming Conformer with right_context: states = key[-(L + R) : -R]):
// cat = [cache ; new rows] first, then cache <- cat[Tc - keep_back .. Tc - keep_back + L) (two launches: the cache is read before it is written)
void cat_keep(const Ctx& ctx, float* pool, long long slot_stride, long long off, const int* slots, const float* newrows, int ldn, float* cat,
              int B, int L, int Tc, int width, int keep_back);
void attn_stream_ring(const Ctx& ctx, const float* qkp, int ld, const RingRef& keys, const float* pp, const long long* plen,
                      float* aw, int B, int Tc, int L, int KLp, int H, int ds, int left50);
void attn_stream(const Ctx& ctx, const float* qkp, int ld, const float* kcat, const float* pp, const long long* plen,
                 float* aw, int B, int Tc, int L, int KLp, int H, int ds, int left50);
void glu_causal_conv(const Ctx& ctx, const float* x2, float* pool, long long slot_stride, long long off, const int* slots,
                     const float* wc, const float* bc, const float* ww, const float* bw, const float* sc, float* y, int B,
                     int Tc, int D, int K);
// Device mirror of the streams' feature FIFOs (OnlineStream's Speech): fifo [slots][cap][feat] rings.
//   fifo_append: frames [g][0 .. nf) of `src` go to ring rows (pos[g] + i) % cap of slot slots[g] (pos[g] < 0: skipped)
//   fifo_gather: x[b][t] = ring row (head[b] + t) % cap of slot slots[b], t < T  -- the chunk input of a step, no host round trip
void fifo_append(const Ctx& ctx, float* fifo, int cap, int feat, const float* src, const int* slots, const int* pos, int G, int nf);
// (the gathered frames pass the online PadSequence's floor of genuine zeros on the way: PadHelper.cs:9-13,58)
void fifo_gather(const Ctx& ctx, const float* fifo, int cap, int feat, const int* slots, const int* head, float* x, int B, int T);
void zero_floats(const Ctx& ctx, float* p, long long n);
// ---- streaming Zipformer v1 (zipformer1.hip; OnlineProjOfZipformer) ------------------------------
// running mean over every frame seen so far: out [B*Tc, D]; cached_avg (avg_off, [D]) and cached_len (len_off, 1 float) per slot
void z1_pool(const Ctx& ctx, const float* x, float* pool, long long slot_stride, long long avg_off, long long len_off, const int* slots,
             float* out, int B, int Tc, int D);
// qkvp rows (ld): q [A] | k [A] | v [A/2] | p [H*4]; kcat [B, L+Tc, A]; pp [2Tc-1+L, H*4]; aw out [H][B][Tc][KLp]
void z1_attn(const Ctx& ctx, const float* qkvp, int ld, const float* kcat, const float* pp, float* aw, int B, int Tc, int L, int KLp,
             int H, int A);
// y = DoubleSwish(dwconv_K([cache ; glu(x2)]) + bias); cache [D][K-1] per slot at `off`, updated
void z1_glu_conv(const Ctx& ctx, const float* x2, float* pool, long long slot_stride, long long off, const int* slots, const float* w,
                 const float* bias, float* y, int B, int Tc, int D, int K);
// y = orig + (BasicNorm(x) - orig) * bypass_scale (scalar)
void z1_norm_bypass(const Ctx& ctx, const float* x, const float* orig, const float* log_eps, const float* bscale, float* y, int M, int D);
// AttentionDownsample, first Din channels: x [B,T,Din] -> y [B,ceil(T/ds),ldy]
void z1_attn_downsample(const Ctx& ctx, const float* x, const float* query, float* y, int B, int T, int Din, int ldy, int ds);
// offline graph: mean over the utterance's frames [B,D]; x[b,t,:] += v[b,:]; the ds frames of a downsampling group side by side
void z1_mean(const Ctx& ctx, const float* x, float* mean, int B, int T, int D);
void z1_add_bcast(const Ctx& ctx, float* x, const float* v, int B, int T, int D);
void z1_group_rows(const Ctx& ctx, const float* x, float* grp, int B, int T, int Din, int ds);
// SimpleCombiner(src1 [B*T,d1], src2) -> y [B*T,d2]; ub != null: src2 = SimpleUpsample(xd [B,Td,d2], ub [ds,d2])[:T]
void z1_combine(const Ctx& ctx, const float* s1, int d1, const float* s2, int d2, const float* w1, const float* ub, int ds, int B, int T,
                int Td, float* y);
void logfloor_inplace(const Ctx& ctx, float* x, long long n);

}  // namespace k2hip
