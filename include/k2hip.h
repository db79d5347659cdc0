/*
 * k2hip.h -- C ABI of libk2hip.so, the MI355X (gfx950) engine behind
 * K2TransducerAsr's IOfflineProj / IOnlineProj operator boundary.
 *
 * Every entry point replaces one reference interface (file:line relative to the
 * reference repo manyeyes/K2TransducerAsr @ 2025-09-19).  The C# side binds them
 * with [DllImport("k2hip")] (csharp/OfflineProjOfHip.cs).
 *
 * Conventions
 *   - every function returns int32 status: 0 = ok, < 0 = error; the message is
 *     in k2hip_last_error() (thread-local).  Nothing throws across the ABI.
 *     (replaces `throw new Exception("EncoderProj failed", ex)`,
 *      OfflineProjOfTransducer.cs:87-90; "Offline recognition failed",
 *      OfflineRecognizer.cs:183-186,299-302)
 *   - all buffers are caller-owned, row-major, f32 / i64 exactly as the reference
 *     marshals them to ONNXRuntime; pointers are host pointers unless the name
 *     says `_dev`.
 *   - one k2hip_model_t = one GPU + one HIP stream; calls on the same handle are
 *     serialised by an internal mutex, different handles run concurrently.
 *   - there is NO CPU fallback: every compute entry point needs a gfx950 device
 *     and fails with K2HIP_ERR_NO_DEVICE otherwise.
 *   - memory: a handle holds, on its device, the model's weights (the .k2w file's size), a grow-only arena per pipeline slot
 *     (~1.1 GB each for 32 x 10 s of zipformer2-large), the streaming state pool (1.85 MB per stream for the medium model) and --
 *     for vocabularies up to ~700 -- the decoder's output for EVERY two-token context, (V + 1) * V * joiner_dim floats (0.5 GB at
 *     V = 500, J = 512), built inside k2hip_model_create (~3 ms of GPU time).  N handles on one GPU hold N copies.  The environment
 *     variable K2HIP_DECODER_TABLE_MB caps the table (default 1024; 0 = never build it: the searches then run the decoder after
 *     every emission, ~1.7x slower per search, same tokens).
 */
#ifndef K2HIP_H
#define K2HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
#if defined(__GNUC__)
#pragma GCC visibility push(default) /* the library is built with -fvisibility=hidden */
#endif

#define K2HIP_OK 0
#define K2HIP_ERR_INVALID (-1)   /* bad argument / shape */
#define K2HIP_ERR_IO (-2)        /* weight file missing or malformed */
#define K2HIP_ERR_NO_DEVICE (-3) /* no usable HIP device */
#define K2HIP_ERR_HIP (-4)       /* a HIP runtime call failed.  Also: the greedy search's vocabulary-parallel exchange timed out TWICE -- its
                                  * workgroups wait for each other and need to be resident together; on a timeout (a GPU shared with other
                                  * processes / models) the library repeats the search with one workgroup per stream, which waits for nobody,
                                  * so a call under load gets slower instead of failing; only if that repeat also reports a timeout (it cannot,
                                  * short of a device fault) does the call return this code */
#define K2HIP_ERR_CAPACITY (-5)  /* caller buffer too small */
#define K2HIP_ERR_UNSUPPORTED (-6)

typedef struct k2hip_model k2hip_model_t;
typedef struct k2hip_offline_stream k2hip_offline_stream_t;
typedef struct k2hip_online_stream k2hip_online_stream_t;
typedef struct k2hip_tokens k2hip_tokens_t;

/* Fixed ids of the reference: OfflineModel.cs:18-20. */
#define K2HIP_BLANK_ID 0
#define K2HIP_SOS_EOS_ID 1
#define K2HIP_UNK_ID 2

typedef struct k2hip_model_info {
    int32_t vocab_size;   /* decoder metadata "vocab_size"   OfflineModel.cs:36-38 */
    int32_t context_size; /* decoder metadata "context_size" OfflineModel.cs:33-35 */
    int32_t joiner_dim;   /* joiner metadata "joiner_dim"    OfflineModel.cs:43-45 */
    int32_t feature_dim;  /* OfflineModel.FeatureDim         OfflineModel.cs:21    */
    int32_t sample_rate;
    int32_t num_stacks;
    int32_t device;
    int32_t reserved; /* encoder output width: joiner_dim, or vocab_size for a zipformer2ctc model (log_probs) */
} k2hip_model_info;

/* Per-call stage timings of the last fused call, measured with HIP events on the
 * model's stream (replaces the reference's DateTime.Now.Ticks RTF print,
 * K2TransducerAsr.Examples/OfflineRecognizer.cs:144,184-189). */
typedef struct k2hip_timing {
    float total_ms;
    float fbank_ms;
    float pad_ms;
    float encoder_ms; /* embed + stacks + proj */
    float greedy_ms;
    float d2h_ms;
    /* GEMM accounting, filled only when instrumentation is enabled with
     * k2hip_set_instrument(model, 1): summed HIP-event duration of every launch
     * of the fp32 MFMA GEMM kernel in the last call, their count and their
     * algorithmic FLOPs (2*M*N*K per launch). */
    float gemm_ms;
    int32_t gemm_launches;
    double gemm_flops;
    double total_flops; /* all matrix work incl. attention/conv (2 FLOP per MAC) */
} k2hip_timing;

/* ---- library ---------------------------------------------------------------- */
const char* k2hip_version(void);
const char* k2hip_last_error(void);
/* number of visible HIP devices (0 if none); never fails */
int32_t k2hip_device_count(void);

/* ---- model: replaces OfflineModel ctor + 3x initModel (OfflineModel.cs:23-73,
 * 84-118).  `weights_path` is a .k2w container holding the ONNX custom-metadata
 * map and all initializers; `overrides` is NULL or "key=value;key=value" applied on
 * top of that map (same keys the reference reads: OfflineModel.cs:31-72,
 * OnlineModel.cs:38-166). A missing file is an error here (the reference returns
 * a null session and fails later with a NullReferenceException, :86-89). */
int32_t k2hip_model_create(const char* weights_path, const char* overrides, int32_t device, k2hip_model_t** out);
int32_t k2hip_model_destroy(k2hip_model_t* model);
/* Device selection through the reference's UNCHANGED constructors (OfflineRecognizer.cs:27-28, OnlineRecognizer.cs:18-19 take file
 * paths and no device): a model spec is "path.k2w" (device 0) or "path.k2w@N" (device N, decimal; the suffix is only split off when
 * what follows the LAST '@' is all digits).  path gets the spec without the suffix (NUL terminated, cap bytes), device the number.
 * Pure string work: no file access, no GPU.  csharp/K2Hip.cs (SplitSpec) implements the same rule in managed code -- it has to
 * answer for ONNX paths on installations without this library -- and tests/native/multi_handle_host.c names its handles this way. */
int32_t k2hip_parse_model_spec(const char* spec, char* path, int32_t cap, int32_t* device);
/* k2hip_model_create on a spec: the container of `spec` on the device it names */
int32_t k2hip_model_create_spec(const char* spec, const char* overrides, k2hip_model_t** out);
int32_t k2hip_model_get_info(const k2hip_model_t* model, k2hip_model_info* info);
/* CustomMetadataMap[key] -> buf (NUL terminated); K2HIP_ERR_INVALID if absent */
int32_t k2hip_model_meta(const k2hip_model_t* model, const char* key, char* buf, int32_t cap);
int32_t k2hip_set_instrument(k2hip_model_t* model, int32_t on);
/* per-launch table of the last instrumented call: rows of 8 floats (M, N, K, batch, act, has_residual, kind, microseconds);
 * kind & 3: 0 plain, 1 conv gather, 2 [K,N] operand; +16 LDS-DMA kernel, +32 skinny kernel, +64 ring kernel, +128 pipelined kernel
 * with its tile in the bits above (BM / 32 in bits 8-11, BN / 32 in bits 12-15; + 2^20: its 16x16x4-tile form).  rows == NULL only queries n_rows. */
int32_t k2hip_get_gemm_profile(k2hip_model_t* model, float* rows, int32_t cap_rows, int32_t* n_rows);
int32_t k2hip_get_timing(const k2hip_model_t* model, k2hip_timing* timing);

/* ---- F1: WavFrontend.GetFbank (WavFrontend.cs:32-36 -> SpeechFeatures.OnlineFbank)
 * samples: f32 in [-1,1]; feats: [n_frames, feature_dim] frame-major.  One-shot
 * (snip_edges) framing of exactly these samples. */
int64_t k2hip_fbank_num_frames(const k2hip_model_t* model, int64_t n_samples);
int32_t k2hip_fbank(k2hip_model_t* model, const float* samples, int64_t n_samples, float* feats, int64_t cap_frames,
                    int64_t* n_frames);

/* ---- F3: PadHelper.PadSequence(List<OfflineInputEntity>) (PadHelper.cs:14-60):
 * right-pad to max+80*tail_frames floats, then every 0.0 -> -23.025850929940457f.
 * out: [B, padded_len]. Runs on the device (the fused entries never call this
 * separately; it is exported so the quirk can be parity-tested on its own). */
int32_t k2hip_pad_sequence(k2hip_model_t* model, const float* const* speech, const int64_t* n_floats, int32_t B,
                           int32_t tail_frames, float* out, int64_t cap_floats, int64_t* padded_len);

/* ---- F4: IOfflineProj.EncoderProj body after padding
 * (OfflineProjOfTransducer.cs:52-85): x [B,T,feature_dim] f32, x_lens [B] i64 (the
 * reference always passes T for every row, :66-70; values are ignored exactly as
 * the reference's loop ignores encoder_out_lens) -> enc_out [B,T',joiner_dim],
 * enc_out_lens [B] (may be NULL), *Tprime. */
int32_t k2hip_encoder_out_frames(const k2hip_model_t* model, int32_t T);
int32_t k2hip_offline_encoder(k2hip_model_t* model, const float* x, const int64_t* x_lens, int32_t B, int32_t T,
                              float* enc_out, int64_t cap_floats, int64_t* enc_out_lens, int32_t* Tprime);
/* debug tap for layer-level parity: tap 0 = encoder_embed out [B,T50,D0];
 * 1+i = stack i out [B,T50,D_i]; 100 = full-dim out [B,T50,Dmax]. */
int32_t k2hip_offline_encoder_tap(k2hip_model_t* model, const float* x, int32_t B, int32_t T, int32_t tap, float* out,
                                  int64_t cap_floats, int64_t* n_floats);

/* ---- F5: IOfflineProj.DecoderProj (OfflineProjOfTransducer.cs:93-123):
 * y [N, context_size] i64 (id < 0 -> zero embedding, used at utterance start,
 * OfflineRecognizer.cs:105) -> dec_out [N, joiner_dim].  y == NULL means
 * [-1, blank] x N (:97-110). */
int32_t k2hip_decoder(k2hip_model_t* model, const int64_t* y, int32_t N, float* dec_out);

/* ---- F6: IOfflineProj.JoinerProj (OfflineProjOfTransducer.cs:125-152):
 * enc [N,J], dec [N,J] -> logits [N, vocab]. */
int32_t k2hip_joiner(k2hip_model_t* model, const float* enc, const float* dec, int32_t N, float* logits);

/* ---- F7 on a precomputed encoder_out: the greedy loops alone.
 * k2hip_greedy_batch  = OfflineRecognizer.ForwardBatchGreedySearch :202-296
 * k2hip_greedy_single = OfflineRecognizer.ForwardGreedySearch      :103-181
 * enc_out [B,T',J] (B = 1 for single).  tokens/timestamps [B, max_tokens] hold the
 * emitted symbols only (the reference's `Tokens` additionally starts with 2*B
 * blanks in the batch path, :250-258, or [-1, blank] in the single path,
 * :115-117; the C# shim re-adds that prefix).  timestamps are 25 Hz frame indices. */
int32_t k2hip_greedy_batch(k2hip_model_t* model, const float* enc_out, int32_t B, int32_t Tprime, int64_t* tokens,
                           int32_t* timestamps, int32_t* n_tokens, int32_t max_tokens);
int32_t k2hip_greedy_single(k2hip_model_t* model, const float* enc_out, int32_t Tprime, int64_t* tokens,
                            int32_t* timestamps, int32_t* n_tokens, int32_t max_tokens);

/* ---- fused hot path: the body of the ForwardBatchOffline delegate
 * (OfflineRecognizer.cs:11,58,189-303): pad (F3) + encoder (F4) + greedy (F7) in
 * one call, one host->device and one device->host crossing.
 * feats[b]: [n_floats[b]] = per-stream OfflineInputEntity.Speech. */
int32_t k2hip_offline_greedy(k2hip_model_t* model, const float* const* feats, const int64_t* n_floats, int32_t B,
                             int64_t* tokens, int32_t* timestamps, int32_t* n_tokens, int32_t max_tokens);
/* ... the ForwardOffline delegate (single stream, :10,57,93-187) */
int32_t k2hip_offline_greedy_single(k2hip_model_t* model, const float* feats, int64_t n_floats, int64_t* tokens,
                                    int32_t* timestamps, int32_t* n_tokens, int32_t max_tokens);
/* ... plus F1: raw samples in (OfflineStream.AddSamples + GetResults). */
int32_t k2hip_offline_greedy_from_samples(k2hip_model_t* model, const float* const* samples, const int64_t* n_samples,
                                          int32_t B, int64_t* tokens, int32_t* timestamps, int32_t* n_tokens,
                                          int32_t max_tokens);
/* Same, with the samples already resident in this model's GPU memory
 * (samples_dev: [B, n_samples_each] contiguous device buffer).  This is the
 * benchmark entry: the timed region starts with inputs in HBM. */
int32_t k2hip_offline_greedy_from_samples_dev(k2hip_model_t* model, const float* samples_dev, int64_t n_samples_each,
                                              int32_t B, int64_t* tokens, int32_t* timestamps, int32_t* n_tokens,
                                              int32_t max_tokens);

/* Pipelined form of the same call, for throughput serving: submit() enqueues the batch and
 * returns a ticket, wait() blocks until THAT batch's tokens are in host memory.  Up to
 * K2HIP_MAX_BATCHES_IN_FLIGHT batches may be in flight: with two, the greedy loop of batch i
 * (latency-bound, a few dozen workgroups) overlaps the encoder of batch i+1 on a second HIP stream; the
 * third is for the modified beam search (k2hip_set_beam), whose per-frame launches take longer
 * than an encoder pass once they share the GPU with one -- the searches of batches i and i+1
 * then run beside the encoder of batch i+2.  Results are identical to the
 * synchronous call (the reference is synchronous: GetResults returns when the batch is done,
 * OfflineRecognizer.cs:85-91; a host that wants that simply calls wait right after submit). */
#define K2HIP_MAX_BATCHES_IN_FLIGHT 3
int32_t k2hip_offline_submit_samples_dev(k2hip_model_t* model, const float* samples_dev, int64_t n_samples_each,
                                         int32_t B, int32_t max_tokens, int32_t* ticket);
int32_t k2hip_offline_wait(k2hip_model_t* model, int32_t ticket, int64_t* tokens, int32_t* timestamps,
                           int32_t* n_tokens);
/* The same with the samples still in HOST memory ([B, n_samples_each] f32), as the reference's caller holds them
 * (K2TransducerAsr.Examples/OfflineRecognizer.cs:164-171): the host-to-device copy is part of the pipeline -- it runs
 * on its own HIP stream under the previous batch's encoder.  The buffer must stay valid and unchanged until
 * k2hip_offline_wait returns for this ticket.  For the copy to be asynchronous it must be page-locked:
 * k2hip_host_alloc / k2hip_host_free hand out such memory (a managed host pins its float[] and registers it, or copies
 * into one of these). */
int32_t k2hip_offline_submit_samples(k2hip_model_t* model, const float* samples_host, int64_t n_samples_each,
                                     int32_t B, int32_t max_tokens, int32_t* ticket);
int32_t k2hip_host_alloc(k2hip_model_t* model, int64_t bytes, void** host_ptr);
int32_t k2hip_host_free(k2hip_model_t* model, void* host_ptr);

/* Development switches (INTEGRATION.md lists them): K2HIP_* environment variables select alternative kernels for the same
 * math or tuning variants.  They are read ONCE, when the first model of the process is created; this call flips one afterwards
 * (the parity tests compare both paths inside one process with it).  Not part of the reference's surface. */
int32_t k2hip_debug_set_switch(const char* env_name, int32_t value);

/* device memory helpers for the benchmark / host runtimes without a HIP binding */
int32_t k2hip_device_alloc(k2hip_model_t* model, int64_t bytes, void** dev_ptr);
int32_t k2hip_device_free(k2hip_model_t* model, void* dev_ptr);
int32_t k2hip_device_upload(k2hip_model_t* model, void* dev_dst, const void* host_src, int64_t bytes);
int32_t k2hip_synchronize(k2hip_model_t* model);

/* ---- OfflineStream (OfflineStream.cs:7-99): per-utterance feature buffer.
 * accept_samples = AddSamples (:43-57).  What a caller can observe is the reference's: speech_length counts the frames of
 * the streaming fbank over everything accepted so far (left-over samples shorter than a frame shift are carried to the
 * next call, as an OnlineFbank does; InputFinished is never called by the reference), get_speech returns them.
 * WHEN they are computed is not: the call only copies the samples into the stream's queue (no fbank launch, no lock on the
 * model); k2hip_offline_recognizer_get_results computes the whole batch's frames in one launch on the device, get_speech
 * computes this stream's.  The queue lives in PINNED host memory (~4 bytes per sample; the buffers of destroyed streams are
 * kept per model, at most 128 of them, and handed to the next streams), which the device reads in place. */
int32_t k2hip_offline_stream_create(k2hip_model_t* model, k2hip_offline_stream_t** out);
int32_t k2hip_offline_stream_destroy(k2hip_offline_stream_t* s);
int32_t k2hip_offline_stream_accept_samples(k2hip_offline_stream_t* s, const float* samples, int64_t n);
/* SpeechLength (float count) */
int64_t k2hip_offline_stream_speech_length(const k2hip_offline_stream_t* s);
/* copies Speech to out (cap floats) */
int32_t k2hip_offline_stream_get_speech(const k2hip_offline_stream_t* s, float* out, int64_t cap);
/* ---- token ids -> text (host only; no GPU involved) ------------------------------------------------------------------
 * OfflineRecognizer.DecodeMulti + CheckText + HexToStr (OfflineRecognizer.cs:432-565; online :321-352) and
 * Utils/ByteDataHelper.SmartByteDecode (:352-397): tokens.txt lookup (first space-separated field), stop at id 2, skip id -1
 * (offline), drop <blk> / <sos/eos> / <unk>, U+2581 -> ' ', <0x..> byte runs -> UTF-8, otherwise spaces removed + byte-BPE
 * decode, lower-case.  Text is UTF-8, NUL terminated; out == NULL only queries len (bytes without the NUL). */
int32_t k2hip_tokens_load(const char* tokens_path, k2hip_tokens_t** out);   /* File.ReadAllLines(tokensFilePath), :36 */
int32_t k2hip_tokens_destroy(k2hip_tokens_t* t);
int32_t k2hip_tokens_size(const k2hip_tokens_t* t);                          /* _tokens.Length (the CTC vocab_size, :325) */
int32_t k2hip_decode_text(const k2hip_tokens_t* t, const int64_t* ids, int32_t n, int32_t online, char* out, int32_t cap,
                          int32_t* len);
/* The byte-BPE alphabet the decode above uses: BYTE_TO_BCHAR[byte] as a code point (ByteDataHelper.cs:27-285,295-299;
 * -1 outside 0..255) and its inverse BCHAR_TO_BYTE (:300-304, BPE_UNK 8263 -> 32; -1 for a char outside the alphabet). */
int32_t k2hip_bbpe_char(int32_t byte);
int32_t k2hip_bbpe_byte(int32_t code_point);

/* ---- CTC models (Model_type "zipformer2ctc": OfflineProjOfZipformer2ctc / OnlineProjOfZipformer2ctc) -------------------
 * The encoder entry points (k2hip_offline_encoder, the online step) return log_probs [B,T',V] for such a model.
 * k2hip_ctc_greedy replaces the loop of ForwardBatchGreedySearchCTC (OfflineRecognizer.cs:383-408): y = first index of the
 * frame maximum (Array.IndexOf), emitted when y != blank and y != previous frame's y (prev_id = -1 at the start of every
 * call); timestamps get frame_offsets[b] added; num_trailing_blank[b] (in/out) follows :392-397.  The fused batch entries
 * (k2hip_offline_greedy*, get_results, k2hip_online_step) run this search automatically for a CTC model. */
int32_t k2hip_ctc_greedy(k2hip_model_t* model, const float* log_probs, int32_t B, int32_t Tprime, const int32_t* frame_offsets,
                         int64_t* tokens, int32_t* timestamps, int32_t* n_tokens, int32_t max_tokens, int32_t* num_trailing_blank);
/* OfflineStream.FrameOffset / NumTrailingBlank (OfflineStream.cs:39-40) */
int32_t k2hip_offline_stream_get_ctc_state(const k2hip_offline_stream_t* s, int32_t* frame_offset, int32_t* num_trailing_blank);

/* ---- decoding method --------------------------------------------------------------------------
 * The reference picks the search by the recognizer's `decodingMethod` string (OfflineRecognizer.cs:54-68) and only
 * knows "greedy_search".  BASELINE.json configs[2] adds "modified_beam_search" (icefall semantics, restated in
 * DESIGN.md): per stream at most `beam` hypotheses, log-softmax over the vocabulary, top-`beam` over
 * beam x V, equal token sequences merged by logaddexp, best hypothesis by length-normalised log-prob.
 * The setting is per model handle and applies to every BATCH entry point (k2hip_offline_greedy*,
 * k2hip_offline_recognizer_get_results, submit/wait); the single-stream path stays greedy. */
int32_t k2hip_set_decoding_method(k2hip_model_t* model, const char* method /* "greedy_search" | "modified_beam_search" */,
                                  int32_t beam /* 1..8, ignored for greedy_search */);
/* operator level: modified beam search over a host encoder_out [B,T',J]; scores [B] (optional) = log-prob of the
 * returned hypothesis */
int32_t k2hip_beam_search(k2hip_model_t* model, const float* enc_out, int32_t B, int32_t Tprime, int32_t beam, int64_t* tokens,
                          int32_t* timestamps, int32_t* n_tokens, int32_t max_tokens, float* scores);
/* log-probs of the hypotheses returned by the last synchronous batch call made under modified_beam_search */
int32_t k2hip_last_scores(k2hip_model_t* model, float* scores, int32_t B);

/* OfflineRecognizer.GetResults (:85-91) minus DecodeMulti: runs the fused batch
 * path on the streams' feature buffers, stores Tokens/Timestamps in each stream
 * (including the reference's 2*B-blank prefix) and calls RemoveSamples (:294). */
int32_t k2hip_offline_recognizer_get_results(k2hip_model_t* model, k2hip_offline_stream_t* const* streams, int32_t B);
/* OfflineRecognizer.GetResult (:77-83): single-stream path, Tokens = [-1, blank, ...] */
int32_t k2hip_offline_recognizer_get_result(k2hip_model_t* model, k2hip_offline_stream_t* stream);
/* stream.Tokens / stream.Timestamps exactly as the reference would hold them */
int32_t k2hip_offline_stream_num_tokens(const k2hip_offline_stream_t* s);
int32_t k2hip_offline_stream_num_timestamps(const k2hip_offline_stream_t* s);
int32_t k2hip_offline_stream_get_tokens(const k2hip_offline_stream_t* s, int64_t* tokens, int32_t cap);
int32_t k2hip_offline_stream_get_timestamps(const k2hip_offline_stream_t* s, int32_t* timestamps, int32_t cap);

/* ======================= streaming path: OnlineRecognizer / IOnlineProj ==========================
 * The reference's IOnlineProj (IOnlineProj.cs:65-71) exposes GetEncoderInitStates / stack_states /
 * unstack_states / EncoderProj(x, states): per tick it copies every stream's ~1.85 MB of caches into
 * batch-major ONNX inputs and back on the host (OnlineProjOfZipformer2.cs:144-489).  Here a stream's
 * caches live in a slot of a device-resident pool for the stream's whole life, so the drop-in unit is
 * the stream handle + one step call; stack/unstack have no counterpart (nothing is copied).
 * NB: for B > 1 the reference's stack_states mis-strides cached_nonlin_attn (:254-262 vs :407-413) and
 * mixes streams; this engine keeps streams independent, i.e. it reproduces the reference at B = 1. */

/* OnlineStream ctor (OnlineStream.cs:22-47): GetEncoderInitStates (zeroed caches), Hyp = Tokens =
 * [blank, blank]. */
int32_t k2hip_online_stream_create(k2hip_model_t* model, k2hip_online_stream_t** out);
int32_t k2hip_online_stream_destroy(k2hip_online_stream_t* s);
/* the stream as if freshly created (caches zeroed in its slot; FIFO, tokens, timestamps, Hyp cleared).  No reference counterpart: a
 * host would create a new OnlineStream (OnlineRecognizer.cs:60-64); SURVEY 8b lists it as a convenience of the C surface. */
int32_t k2hip_online_stream_reset(k2hip_online_stream_t* s);
/* ChunkLength = T, ShiftLength = decode_chunk_len (OnlineModel.cs:48-49); frames of encoder_out per chunk */
int32_t k2hip_online_chunk_info(const k2hip_model_t* model, int32_t* chunk_length, int32_t* shift_length,
                                int32_t* frames_per_chunk);
/* OnlineStream.AddSamples (:57-79): streaming fbank on the new samples, frames appended to the FIFO */
int32_t k2hip_online_stream_accept_samples(k2hip_online_stream_t* s, const float* samples, int64_t n);
/* AddSamples for many streams at once (one fbank launch for the whole round when every stream is at the
 * same position, e.g. a server that feeds all connections on a common tick; falls back to per-stream
 * launches otherwise).  Semantically B independent AddSamples calls. */
int32_t k2hip_online_accept_samples_batch(k2hip_model_t* model, k2hip_online_stream_t* const* streams, int32_t B,
                                          const float* const* samples, const int64_t* n);
/* ... or push ready-made feature frames ([n_frames, feature_dim]) */
/* the same over the rows of one [B, n] sample matrix (row_stride in floats) */
int32_t k2hip_online_accept_samples_matrix(k2hip_model_t* model, k2hip_online_stream_t* const* streams, int32_t B, const float* samples,
                                           int64_t row_stride, int64_t n);
int32_t k2hip_online_stream_accept_features(k2hip_online_stream_t* s, const float* feats, int64_t n_frames);
/* OnlineInputEntity.SpeechLength (floats buffered) */
int64_t k2hip_online_stream_speech_length(const k2hip_online_stream_t* s);
/* OnlineStream.IsFinished(isEndpoint) (:124-161), including its side effect of feeding 400 zero
 * samples when less than a chunk is buffered */
int32_t k2hip_online_stream_is_finished(k2hip_online_stream_t* s, int32_t is_endpoint, int32_t* finished);
/* OnlineRecognizer.GetResults -> ForwardBatchGreedySearch (OnlineRecognizer.cs:76-219), minus DecodeMulti:
 * every stream with a full chunk buffered (GetDecodeChunk :82-100) is decoded for that chunk
 * (RemoveChunk :102-117 drops ShiftLength frames), its Hyp / Tokens / Timestamps / caches are updated;
 * decoded[i] = 1 for those, 0 for streams that had no chunk (the reference removes them from the
 * caller's list, :117-120); n_new_tokens[i] = symbols emitted in this chunk.
 * On failure no stream's host-side state has moved (no chunk removed, no token appended), but the DEVICE caches of the streams
 * that were being decoded may have advanced in place; those streams are marked and every later k2hip_online_step that names one
 * returns K2HIP_ERR_INVALID until k2hip_online_stream_reset -- the same chunk is never fed into caches that already moved. */
int32_t k2hip_online_step(k2hip_model_t* model, k2hip_online_stream_t* const* streams, int32_t B, int32_t* decoded,
                          int32_t* n_new_tokens);
/* the stream's processed_lens state as the reference holds it between steps (Zipformer2: frames consumed; Conformer: 2 at
 * creation, then the batch size of its last step -- OnlineProjOfConformer.cs:77,229) */
/* ---- operator level of the streaming path: IOnlineProj (IOnlineProj.cs:65-71) ----------------------------------------
 * For a host that keeps the reference's OnlineRecognizer loop unchanged and swaps only the operator (csharp/OnlineProjOfHip.cs).
 * A k2hip_online_state_t is ONE stream's encoder caches: a slot of the device state pool plus processed_lens.  It replaces the
 * List<List<float[]>> of GetEncoderInitStates (OnlineProjOfZipformer2.cs:144-238); stack_states / unstack_states (:240-489)
 * are the identity on handles because EncoderProj advances the caches in place.  Streaming Zipformer2 transducer models only. */
typedef struct k2hip_online_state k2hip_online_state_t;
int32_t k2hip_online_state_create(k2hip_model_t* model, k2hip_online_state_t** out);
int32_t k2hip_online_state_destroy(k2hip_online_state_t* state);
int64_t k2hip_online_state_processed_len(const k2hip_online_state_t* state);
/* EncoderProj (OnlineProjOfZipformer2.cs:491-618): feats [B, ChunkLength, FeatureDim] raw fbank frames (OnlineInputEntity.Speech of
 * each stream's GetDecodeChunk), encoder_out [B, T', joiner_dim] (T' = k2hip_online_chunk_info's frames_per_chunk).  Decoder and
 * joiner are k2hip_decoder / k2hip_joiner.  K2HIP_ERR_CAPACITY if cap_floats is too small (checked before any work: nothing
 * advances).  If the device work itself fails, processed_lens does not advance but the states' caches may have been updated in
 * place: those states are marked, later calls that name one return K2HIP_ERR_INVALID; destroy and re-create them. */
int32_t k2hip_online_encoder(k2hip_model_t* model, k2hip_online_state_t* const* states, int32_t B, const float* feats, float* encoder_out,
                             int64_t cap_floats);

int64_t k2hip_online_stream_processed_len(const k2hip_online_stream_t* s);
int32_t k2hip_online_stream_num_tokens(const k2hip_online_stream_t* s);
int32_t k2hip_online_stream_num_timestamps(const k2hip_online_stream_t* s);
int32_t k2hip_online_stream_get_tokens(const k2hip_online_stream_t* s, int64_t* tokens, int32_t cap);
int32_t k2hip_online_stream_get_timestamps(const k2hip_online_stream_t* s, int32_t* timestamps, int32_t cap);
int32_t k2hip_online_stream_get_hyp(const k2hip_online_stream_t* s, int64_t* hyp2);
/* copy one cache out of the stream's device slot (parity tests / debugging):
 * kind 0 cached_key [left,32H], 1 cached_nonlin_attn [left,3D/4], 2/3 cached_val1/2 [left,12H],
 * 4/5 cached_conv1/2 [D,K/2], 6 embed_states [128,3,19] (layer ignored); out == NULL queries n.
 * For an "lstm" model (OnlineProjOfLstm.cs:55-75): kind 0 = h of `layer` [d_model], kind 1 = c of `layer` [rnn_hidden_size];
 * for a streaming "conformer" (OnlineProjOfConformer.cs:55-82): kind 0 = cached_attn of `layer` [left_context, D], kind 1 =
 * cached_conv of `layer` [K-1, D] */
int32_t k2hip_online_stream_state(k2hip_online_stream_t* s, int32_t layer, int32_t kind, float* out, int64_t cap, int64_t* n);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* K2HIP_H */
