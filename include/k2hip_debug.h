/*
 * k2hip_debug.h -- test, tuning and monitoring hooks of libk2hip.so.
 *
 * NOT part of the drop-in boundary (include/k2hip.h): nothing here has a counterpart in
 * K2TransducerAsr's IOfflineProj / IOnlineProj, and a host that replaces the reference's
 * operators never calls these.  They exist so that the parity tests (tests/), the tuning
 * tools (tools/) and a service's monitoring can look inside the engine through the same
 * C ABI instead of through private symbols; every exported `k2hip_debug_*` symbol of the
 * library is declared here (tests/test_abi.py holds the export table to the two headers).
 * Same conventions as k2hip.h: int32 status, message in k2hip_last_error(), host pointers.
 */
#ifndef K2HIP_DEBUG_H
#define K2HIP_DEBUG_H
#include "k2hip.h"

#ifdef __cplusplus
extern "C" {
#endif
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

/* (k2hip_debug_set_switch is declared in k2hip.h: INTEGRATION.md documents the switches.) */

/* ---- search ------------------------------------------------------------------------------------ */
/* one-part repeats of the vocabulary-parallel search since the model was created (K2HIP_ERR_HIP in k2hip.h explains when the
 * engine repeats a search) */
int32_t k2hip_debug_search_retries(k2hip_model_t* model, int32_t* n);
/* the all-contexts decoder table against the decoder routine itself on `n_samples` sampled contexts (plus the start contexts):
 * rows = rows compared, mismatched = rows that are not bit-equal */
int32_t k2hip_debug_decoder_table_check(k2hip_model_t* model, int32_t n_samples, uint32_t seed, int64_t* rows, int64_t* mismatched);
/* Per-frame tap of the modified beam search.  With the switch K2HIP_BEAM_TRACE on, every SYNCHRONOUS batch call under
 * modified_beam_search (and k2hip_beam_search) records, per stream and frame, 2 * beam + 1 int32 words: the `beam` selected
 * candidates in rank order (score desc, flat index asc) as flat indexes `hypothesis slot * vocab_size + token` (-1 where the frame
 * had fewer candidates), their scores as float bits, and the number of hypotheses that survive the frame's merges.
 * trace: [B][Tprime][2 * beam + 1] (NULL only queries the three sizes).  tests/parity.py compares it with the oracle's tap
 * (oracle/k2_oracle_beam.c) to find the first frame at which the two searches part. */
int32_t k2hip_debug_beam_trace(k2hip_model_t* model, int32_t* trace, int64_t cap_words, int32_t* B, int32_t* Tprime, int32_t* beam);

/* ---- streaming --------------------------------------------------------------------------------- */
/* mark a stream as if a chunk step over it had failed on the device (k2hip_online_step's poisoning rule, k2hip.h) */
int32_t k2hip_debug_poison_stream(k2hip_online_stream_t* s);
/* does the stream's device mirror of its feature FIFO hold the FIFO right now? */
int32_t k2hip_debug_stream_mirrored(const k2hip_online_stream_t* s, int32_t* ok);

/* ---- GEMM kernels ------------------------------------------------------------------------------ */
/* ONE launch of tile configuration `cfg` (-1: the dispatcher's own choice for the shape; the numbering is csrc/gemm.hip's) on the
 * caller's operands: C = act(A W^T + bias) (+ res).  A [M,K], W [N,K], bias [N] or NULL, res [M, ldo] or NULL, C [M, ldo] row-major
 * f32 in host memory; act = 0 none, 1 SwooshL, 2 SwooshR, 3 tanh, 4 sigmoid, 5 ReLU, 6 DoubleSwish.  glu = 0: ldo = N.
 * glu = 1 / 2: the first `glu_cols` (0 = all N) columns are blocks of 32 = 16 values | their 16 gates and the epilogue writes
 * value * sigmoid(gate) / value * tanh(gate): ldo = glu_cols / 2 + (N - glu_cols).  C is pre-filled with NaNs, so an element the
 * kernel does not write shows.  tests/test_gemm_gpu.py compares the result with a float64 product computed on the host. */
int32_t k2hip_debug_gemm_run(k2hip_model_t* model, const float* A, const float* W, const float* bias, const float* res, float* C,
                             int32_t M, int32_t N, int32_t K, int32_t act, int32_t glu, int32_t glu_cols, int32_t cfg);
/* time `iters` launches of a shape / configuration on pseudo-random operands (tools/gemm_lab.py, gemm_tune.py) */
int32_t k2hip_debug_gemm(k2hip_model_t* model, int32_t M, int32_t N, int32_t K, int32_t act, int32_t with_res, int32_t cfg,
                         int32_t iters, float* ms);
/* the same, and max_err = largest |difference| from the register-staged kernel on the same operands (a GPU-vs-GPU figure for
 * the tuning tools; the parity test is k2hip_debug_gemm_run against the host) */
int32_t k2hip_debug_gemm_check(k2hip_model_t* model, int32_t M, int32_t N, int32_t K, int32_t act, int32_t with_res, int32_t cfg,
                               int32_t iters, float* ms, float* max_err);
/* ONE launch with in-kernel s_memtime stamps: out [n_wg][n_waves][64] (tools/gemm_dma_trace.py) */
int32_t k2hip_debug_gemm_trace(k2hip_model_t* model, int32_t M, int32_t N, int32_t K, int32_t act, int32_t with_res, int32_t cfg,
                               unsigned long long* out, int64_t cap, int32_t* n_wg, int32_t* n_waves);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* K2HIP_DEBUG_H */
