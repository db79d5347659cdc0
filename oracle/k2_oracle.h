/*
 * k2_oracle.h -- CPU restatement ("oracle") of the K2TransducerAsr RNN-T decode
 * hot path.  TEST INFRASTRUCTURE ONLY: nothing under oracle/ is linked, imported
 * or executed by the product (k2transducerasr_amd/, libk2hip.so).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it, and only
 * as the checker / the CPU baseline.
 *
 * PARITY UNPINNED: the reference (C#) has no tests, golden vectors or fixtures
 * for this path, cannot be built here (no dotnet/mono), and
 * the arithmetic it runs lives in un-vendored third-party packages
 * (Microsoft.ML.OnnxRuntime 1.22.1, ManySpeech.SpeechFeatures 1.1.6,
 * K2TransducerAsr/K2TransducerAsr.csproj:12-14) over model files that are not in
 * the tree.  What this oracle restates:
 *   - from the reference itself (file:line cited at each function): padding
 *     quirks, tensor layouts at the IOfflineProj boundary, the greedy-search
 *     control flow, argmax tie-break, emit filters, context seeding;
 *   - from the published algorithms the third-party pieces implement: kaldi
 *     fbank, icefall Zipformer2 / stateless decoder / joiner inference graphs.
 * It is pinned by hand-computed known-answer tests and by an independent torch
 * restatement (tests/torch_twin.py) -- not by reference outputs.
 */
#ifndef K2_ORACLE_H
#define K2_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct k2o_model k2o_model;

/* ---- model ---------------------------------------------------------------- */
k2o_model* k2o_model_load(const char* k2w_path);
void k2o_model_free(k2o_model* m);
const char* k2o_last_error(void);
/* metadata lookup (NULL if absent) -- mirrors CustomMetadataMap[key] */
const char* k2o_meta(const k2o_model* m, const char* key);
int k2o_vocab_size(const k2o_model* m);
int k2o_joiner_dim(const k2o_model* m);
/* width of the offline / online encoder output: joiner_dim, or vocab_size for a zipformer2ctc model (log_probs) */
int k2o_encoder_out_dim(const k2o_model* m);
int k2o_context_size(const k2o_model* m);
int k2o_feature_dim(const k2o_model* m);

/* ---- F1: fbank (WavFrontend.GetFbank, WavFrontend.cs:32-36) ----------------- */
/* number of frames for n samples (snip_edges) */
int64_t k2o_fbank_num_frames(const k2o_model* m, int64_t n_samples);
/* feats: [n_frames, feature_dim] frame-major; returns n_frames or <0 */
int64_t k2o_fbank(const k2o_model* m, const float* samples, int64_t n_samples, float* feats, int64_t cap_frames);

/* ---- F3: PadHelper.PadSequence (PadHelper.cs:14-60) ------------------------- */
/* returns padded per-utterance float count (max_len + 80*tail_frames); out is
 * [B, padded] flattened; out may be NULL to query the size. */
int64_t k2o_pad_sequence(const float* const* speech, const int64_t* n_floats, int B, int tail_frames, float* out);

/* ---- F4: offline encoder (OfflineProjOfTransducer.EncoderProj :48-92) ------- */
/* T' for T input frames */
int k2o_encoder_out_frames(const k2o_model* m, int T);
/* x: [B,T,feat]; enc_out: [B,T',joiner_dim]. */
int k2o_offline_encoder(const k2o_model* m, const float* x, int B, int T, float* enc_out);
/* debug taps: tap 0 = encoder_embed output [B,T50,D0]; tap 1+i = output of
 * stack i at 50 Hz [B,T50,D_i]; tap 100 = full-dim output before final
 * downsample [B,T50,Dmax]. Returns number of floats written or <0. */
int64_t k2o_offline_encoder_tap(const k2o_model* m, const float* x, int B, int T, int tap, float* out, int64_t cap);

/* ---- F5/F6: decoder, joiner -------------------------------------------------- */
/* y: [N, ctx] int64 (negative id -> zero embedding); dec_out: [N, joiner_dim] */
int k2o_decoder(const k2o_model* m, const int64_t* y, int N, float* dec_out);
/* logits: [N, vocab] = output_linear(tanh(enc + dec)) */
int k2o_joiner(const k2o_model* m, const float* enc, const float* dec, int N, float* logits);

/* ---- F7: greedy search -------------------------------------------------------- */
/* Reference argmax (OfflineRecognizer.cs:151-154): later index wins ties/NaN. */
int k2o_argmax_ref(const float* logits, int V);

/* ForwardBatchGreedySearch (OfflineRecognizer.cs:189-303) on a precomputed
 * encoder_out [B,T',J].  tokens/timestamps: [B, max_tokens] (real emitted
 * tokens only, WITHOUT the 2*B blank prefix the reference seeds, Q10);
 * n_tokens: [B].  margins (optional, [B*T']): top1-top2 logit gap per frame. */
int k2o_greedy_batch(const k2o_model* m, const float* enc_out, int B, int Tp,
                     int64_t* tokens, int32_t* timestamps, int32_t* n_tokens, int max_tokens,
                     float* margins);
/* ForwardGreedySearch (OfflineRecognizer.cs:93-187), B = 1. */
int k2o_greedy_single(const k2o_model* m, const float* enc_out, int Tp,
                      int64_t* tokens, int32_t* timestamps, int32_t* n_tokens, int max_tokens,
                      float* margins);

/* Modified beam search (icefall beam_search.py modified_beam_search; NOT in the reference, which only has
 * greedy_search -- BASELINE.json configs[2] asks for it): see k2_oracle_beam.c.  enc_out [B,T',J]; tokens /
 * timestamps [B,max_tokens] of the best hypothesis (length-normalised); scores [B] its log_prob (optional);
 * margins [B,T'+1] (optional): per-frame beam-boundary score gap, final best-vs-second gap. */
int k2o_modified_beam_search(const k2o_model* m, const float* enc_out, int B, int Tp, int beam, int64_t* tokens,
                             int32_t* timestamps, int32_t* n_tokens, int max_tokens, float* scores, float* margins);
/* the same with a per-frame tap: trace [B][T'][4*beam + 1] int32 words = the frame's 2*beam best candidates (flat index = slot * V +
 * token, then their scores as float bits) and the number of surviving hypotheses; see k2_oracle_beam.c */
int k2o_modified_beam_search_trace(const k2o_model* m, const float* enc_out, int B, int Tp, int beam, int64_t* tokens,
                                   int32_t* timestamps, int32_t* n_tokens, int max_tokens, float* scores, float* margins,
                                   int32_t* trace);

/* CTC greedy search (OfflineRecognizer.cs:305-424, OnlineRecognizer.cs:220-313) over log_probs [B,T',V]:
 * first-index argmax, drop blanks and repeats; frame_offsets / num_trailing_blank may be NULL */
int k2o_ctc_greedy(const float* log_probs, int B, int Tp, int V, const int32_t* frame_offsets, int64_t* tokens,
                   int32_t* timestamps, int32_t* n_tokens, int max_tokens, int32_t* num_trailing_blank);

/* End to end: features -> pad -> encoder -> batch greedy (GetResults). */
int k2o_offline_recognize_batch(const k2o_model* m, const float* const* feats, const int64_t* n_floats, int B,
                                int64_t* tokens, int32_t* timestamps, int32_t* n_tokens, int max_tokens);

/* ---- streaming path (OnlineRecognizer / OnlineProjOfZipformer2): see k2_oracle_online.c ---- */
typedef struct k2o_online_stream k2o_online_stream;
int k2o_online_chunk_length(const k2o_model* m);      /* ChunkLength = T          (OnlineModel.cs:48) */
int k2o_online_shift_length(const k2o_model* m);      /* ShiftLength = decode_chunk_len (:49)         */
int k2o_online_frames_per_chunk(const k2o_model* m);  /* encoder_out frames per chunk                 */
k2o_online_stream* k2o_online_stream_create(const k2o_model* m); /* GetEncoderInitStates + Hyp/Tokens = [blank, blank] */
void k2o_online_stream_free(k2o_online_stream* s);
int k2o_online_stream_num_layers(const k2o_online_stream* s);
int64_t k2o_online_stream_processed_len(const k2o_online_stream* s);
/* kind: 0 cached_key [left, 32H], 1 cached_nonlin_attn [left, 3D/4], 2/3 cached_val1/2 [left, 12H],
 * 4/5 cached_conv1/2 [D, K/2], 6 embed_states [128,3,19]; out == NULL returns the size */
int64_t k2o_online_stream_state(const k2o_online_stream* s, int layer, int kind, float* out, int64_t cap);
int k2o_online_stream_num_tokens(const k2o_online_stream* s);
int k2o_online_stream_num_timestamps(const k2o_online_stream* s);
void k2o_online_stream_get_tokens(const k2o_online_stream* s, int64_t* out);
void k2o_online_stream_get_timestamps(const k2o_online_stream* s, int32_t* out);
void k2o_online_stream_get_hyp(const k2o_online_stream* s, int64_t* out);
/* lstm streams (OnlineProjOfLstm.cs:55-75): kind 0 = h of `layer` [d_model], 1 = c of `layer` [rnn_hidden_size] */
int64_t k2o_online_stream_lstm_state(const k2o_model* m, const k2o_online_stream* s, int layer, int kind, float* out, int64_t cap);
/* one encoder chunk for one stream: x [T,80] (log-floored) -> enc_out [T'c, J]; returns T'c */
int k2o_online_encoder_chunk(const k2o_model* m, k2o_online_stream* s, const float* x, float* enc_out);
/* OnlineRecognizer.ForwardBatchGreedySearch (:85-219) over B streams with one full chunk each */
int k2o_online_step(const k2o_model* m, k2o_online_stream** streams, const float* const* chunks, int B, int32_t* n_new);

#ifdef __cplusplus
}
#endif
#endif
