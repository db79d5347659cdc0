"""Model configuration contract for the k2hip engine.

The reference's real configuration is the ONNX custom-metadata map
(K2TransducerAsr/OfflineModel.cs:31-72: ``context_size``, ``vocab_size``,
``joiner_dim``, ``model_type``, ``feature``; K2TransducerAsr/OnlineModel.cs:38-166
adds ``encoder_dims``, ``num_encoder_layers``, ``cnn_module_kernels``,
``left_context_len``, ``query_head_dims``, ``value_head_dims``, ``num_heads``,
``decode_chunk_len``, ``T``).  All values are *strings*; lists are
comma-separated integers, exactly as the reference parses them
(OnlineModel.cs:60-110).  The ``.k2w`` weight file carries the same map, plus
the per-stack hyper-parameters that an ONNX export bakes into the graph
(``feedforward_dims``, ``downsampling_factors``, ``pos_dim``,
``pos_head_dims``) and the fbank options that live inside the third-party
SpeechFeatures package.
"""
from __future__ import annotations

import copy

# kaldi-native-fbank defaults, with the window the reference passes
# (Model/FrontendConfEntity.cs:8 -> "hamming"; WavFrontend.cs:22-29).
FBANK_DEFAULTS = {
    "feature": "fbank",
    "sample_rate": "16000",
    "feature_dim": "80",
    "frame_length_ms": "25",
    "frame_shift_ms": "10",
    "window_type": "hamming",
    "dither": "0",
    "snip_edges": "1",
    "preemph_coeff": "0.97",
    "remove_dc_offset": "1",
    "low_freq": "20",
    "high_freq": "0",
    "input_scale": "1.0",
}


def _csv(xs):
    return ",".join(str(int(x)) for x in xs)


def make_zipformer2_meta(
    *,
    encoder_dims,
    num_encoder_layers,
    feedforward_dims,
    num_heads,
    cnn_module_kernels,
    downsampling_factors,
    query_head_dim=32,
    value_head_dim=12,
    pos_head_dim=4,
    pos_dim=48,
    joiner_dim=512,
    decoder_dim=512,
    vocab_size=500,
    context_size=2,
    comment="",
    streaming=False,
    chunk_size=16,
    left_context_frames=128,
    ctc=False,
):
    n = len(encoder_dims)
    assert all(
        len(x) == n
        for x in (num_encoder_layers, feedforward_dims, num_heads, cnn_module_kernels, downsampling_factors)
    )
    meta = dict(FBANK_DEFAULTS)
    meta.update(
        {
            "model_type": "zipformer2",
            "version": "1",
            "model_author": "k2hip-synthetic",
            "comment": comment,
            "encoder_dims": _csv(encoder_dims),
            "num_encoder_layers": _csv(num_encoder_layers),
            "feedforward_dims": _csv(feedforward_dims),
            "num_heads": _csv(num_heads),
            "cnn_module_kernels": _csv(cnn_module_kernels),
            "downsampling_factors": _csv(downsampling_factors),
            "query_head_dims": _csv([query_head_dim] * n),
            "value_head_dims": _csv([value_head_dim] * n),
            "pos_head_dims": _csv([pos_head_dim] * n),
            "pos_dim": str(pos_dim),
            "joiner_dim": str(joiner_dim),
            "decoder_dim": str(decoder_dim),
            "vocab_size": str(vocab_size),
            "context_size": str(context_size),
        }
    )
    if ctc:
        # Zipformer2 encoder + CTC head, no decoder / joiner: Model_type "zipformer2ctc" -> OfflineProjOfZipformer2ctc /
        # OnlineProjOfZipformer2ctc and decodingMethod "greedy_search_ctc" (OfflineRecognizer.cs:44-47, OnlineRecognizer.cs:33-36)
        meta["model_type"] = "zipformer2ctc"
    if streaming:
        # keys of a streaming export, as OnlineModel.cs:38-110 reads them: T = ChunkLength,
        # decode_chunk_len = ShiftLength (OnlineModel.cs:48-49); left_context_len is already divided
        # by each stack's downsampling factor (OnlineProjOfZipformer2.cs:72,78 use it with ds = 1)
        meta.update(
            {
                "streaming": "1",
                "decode_chunk_len": str(2 * chunk_size),
                "T": str(2 * chunk_size + 13),
                "left_context_len": _csv([left_context_frames // d for d in downsampling_factors]),
            }
        )
    return meta


def make_conformer_meta(
    *,
    encoder_dim=512,
    num_encoder_layers=12,
    feedforward_dim=2048,
    num_heads=8,
    cnn_module_kernel=31,
    joiner_dim=512,
    decoder_dim=512,
    vocab_size=5537,
    context_size=2,
    comment="",
    streaming=False,
    chunk_size=16,
    left_context=64,
    right_context=0,
):
    """Offline Conformer transducer (reference: Model_type "conformer" -> OfflineProjOfTransducer,
    OfflineRecognizer.cs:38-53; BASELINE.json configs[4]).  The graph is icefall's
    pruned_transducer_stateless2 Conformer (Conv2dSubsampling x4, rel-pos MHSA with pos_bias_u/v,
    macaron feed-forward with DoubleSwish, conv module with GLU + depthwise conv, BasicNorm); the
    csv keys reuse the Zipformer names with one entry."""
    assert encoder_dim % num_heads == 0
    meta = dict(FBANK_DEFAULTS)
    meta.update(
        {
            "model_type": "conformer",
            "version": "1",
            "model_author": "k2hip-synthetic",
            "comment": comment,
            "encoder_dims": _csv([encoder_dim]),
            "num_encoder_layers": _csv([num_encoder_layers]),
            "feedforward_dims": _csv([feedforward_dim]),
            "num_heads": _csv([num_heads]),
            "cnn_module_kernels": _csv([cnn_module_kernel]),
            "joiner_dim": str(joiner_dim),
            "decoder_dim": str(decoder_dim),
            "vocab_size": str(vocab_size),
            "context_size": str(context_size),
        }
    )
    if streaming:
        # streaming export (causal convolutions, chunk_forward): the keys OnlineModel.cs:131-166 reads for OnlineProjOfConformer.
        # T = (chunk_size + 2 + right_context) * 4 + 3 input frames per chunk (one embed frame is cut on each side),
        # decode_chunk_len = chunk_size * 4
        assert right_context >= 0
        meta.update(
            {
                "streaming": "1",
                "encoder_dim": str(encoder_dim),
                "cnn_module_kernel": str(cnn_module_kernel),
                "left_context": str(left_context),
                "right_context": str(right_context),
                "chunk_size": str(chunk_size),
                "pad_length": str((2 + right_context) * 4 + 3),
                "decode_chunk_len": str(chunk_size * 4),
                "T": str((chunk_size + 2 + right_context) * 4 + 3),
            }
        )
    return meta


def make_lstm_meta(
    *,
    d_model=512,
    rnn_hidden_size=1024,
    dim_feedforward=2048,
    num_encoder_layers=12,
    joiner_dim=512,
    decoder_dim=512,
    vocab_size=500,
    context_size=2,
    comment="",
):
    """LSTM transducer (Model_type "lstm": offline through OfflineProjOfTransducer, OfflineRecognizer.cs:38-53; streaming
    through OnlineProjOfLstm, whose states are h [layers, B, d_model] and c [layers, B, rnn_hidden_size],
    OnlineProjOfLstm.cs:55-75).  The graph is icefall's lstm_transducer_stateless2: Conv2dSubsampling (no padding in time, two
    stride-2 convs) + layers of {LSTM with projection, feed-forward with DoubleSwish, BasicNorm}.  The streaming export consumes
    T = 9 frames and advances by decode_chunk_len = 4 (one encoder frame per chunk); the same file serves both paths."""
    meta = dict(FBANK_DEFAULTS)
    meta.update(
        {
            "model_type": "lstm",
            "version": "1",
            "model_author": "k2hip-synthetic",
            "comment": comment,
            "num_encoder_layers": _csv([num_encoder_layers]),
            "encoder_dims": _csv([d_model]),
            "feedforward_dims": _csv([dim_feedforward]),
            "d_model": str(d_model),
            "rnn_hidden_size": str(rnn_hidden_size),
            "decode_chunk_len": "4",
            "T": "9",
            "joiner_dim": str(joiner_dim),
            "decoder_dim": str(decoder_dim),
            "vocab_size": str(vocab_size),
            "context_size": str(context_size),
        }
    )
    return meta


def make_zipformer_meta(
    *,
    encoder_dims,
    attention_dims,
    num_encoder_layers,
    feedforward_dims,
    num_heads,
    cnn_module_kernels,
    downsampling_factors,
    pos_dim=4,
    decode_chunk_size=16,
    num_left_chunks=4,
    joiner_dim=512,
    decoder_dim=512,
    vocab_size=500,
    context_size=2,
    comment="",
    streaming=True,
):
    """Streaming Zipformer (v1) transducer: Model_type "zipformer" -> OnlineProjOfZipformer (OnlineRecognizer.cs:28-30), whose
    per-stack states are cached_len [L,B], cached_avg [L,B,D], cached_key [L,left,B,att], cached_val / cached_val2
    [L,left,B,att/2], cached_conv1 / cached_conv2 [L,B,D,K-1] (OnlineProjOfZipformer.cs:56-111).  The graph is icefall's
    pruned_transducer_stateless7_streaming Zipformer: Conv2dSubsampling ((T-7)//2 frames), stacks of ZipformerEncoderLayer
    (three feed-forwards, cumulative-mean pooling, rel-pos attention whose weights are used twice, two causal convolution
    modules, BasicNorm, scalar bypass) behind AttentionDownsample / SimpleUpsample / SimpleCombiner, and a final
    AttentionDownsample by 2.  The reference reads encoder_dims, attention_dims, num_encoder_layers, cnn_module_kernels,
    left_context_len, T and decode_chunk_len (OnlineModel.cs:38-72); num_heads, feedforward_dims, downsampling_factors and
    pos_dim are baked into an ONNX graph and are carried as extra keys here."""
    n = len(encoder_dims)
    assert all(len(x) == n for x in (attention_dims, num_encoder_layers, feedforward_dims, num_heads, cnn_module_kernels, downsampling_factors))
    meta = dict(FBANK_DEFAULTS)
    meta.update(
        {
            "model_type": "zipformer",
            "version": "1",
            "model_author": "k2hip-synthetic",
            "comment": comment,
            "encoder_dims": _csv(encoder_dims),
            "attention_dims": _csv(attention_dims),
            "num_encoder_layers": _csv(num_encoder_layers),
            "feedforward_dims": _csv(feedforward_dims),
            "num_heads": _csv(num_heads),
            "cnn_module_kernels": _csv(cnn_module_kernels),
            "downsampling_factors": _csv(downsampling_factors),
            "pos_dim": str(pos_dim),
            "joiner_dim": str(joiner_dim),
            "decoder_dim": str(decoder_dim),
            "vocab_size": str(vocab_size),
            "context_size": str(context_size),
        }
    )
    if streaming:
        meta.update(
            {
                "left_context_len": _csv([num_left_chunks * decode_chunk_size // d for d in downsampling_factors]),
                "streaming": "1",
                "decode_chunk_len": str(2 * decode_chunk_size),
                "T": str(2 * decode_chunk_size + 7),
            }
        )
    # streaming=False: the offline graph of the non-streaming recipe (pruned_transducer_stateless7): Model_type "zipformer" in
    # OfflineRecognizer's switch (OfflineRecognizer.cs:40-44) -- mean pooling over the utterance, centred depthwise convolutions,
    # attention over the whole utterance; same state-dict names
    return meta


ZIPFORMER1_PRESETS = {
    # the icefall pruned_transducer_stateless7_streaming recipe (the streaming-zipformer-en / bilingual-zh-en exports the
    # reference's README lists): 5 stacks of 384, chunk 32 frames in (T = 39), 4 left chunks
    "zipformer-streaming-en": dict(
        encoder_dims=[384] * 5, attention_dims=[192] * 5, num_encoder_layers=[2, 4, 3, 2, 4], feedforward_dims=[1024, 1024, 2048, 2048, 1024],
        num_heads=[8] * 5, cnn_module_kernels=[31] * 5, downsampling_factors=[1, 2, 4, 8, 2], vocab_size=500,
    ),
    # the offline recipe (icefall pruned_transducer_stateless7, 70 M parameters)
    "zipformer-en": dict(
        encoder_dims=[384] * 5, attention_dims=[192] * 5, num_encoder_layers=[2, 4, 3, 2, 4], feedforward_dims=[1024, 1024, 2048, 2048, 1024],
        num_heads=[8] * 5, cnn_module_kernels=[31] * 5, downsampling_factors=[1, 2, 4, 8, 2], vocab_size=500, streaming=False,
    ),
    "zipformer-tiny-test": dict(
        encoder_dims=[64, 64, 96, 96], attention_dims=[32, 32, 96, 96], num_encoder_layers=[1, 2, 1, 1], feedforward_dims=[128, 128, 160, 160],
        num_heads=[2, 2, 4, 4], cnn_module_kernels=[7, 7, 5, 7], downsampling_factors=[1, 2, 4, 2],
        joiner_dim=512, decoder_dim=64, vocab_size=37, streaming=False,
    ),
    # parity-test model: growing stack widths (AttentionDownsample.extra_proj + SimpleCombiner padding), a skip connection
    # (stack 3 takes stack 1's output), head sizes 16 and 24, short left context
    "zipformer-streaming-tiny-test": dict(
        encoder_dims=[64, 64, 96, 96], attention_dims=[32, 32, 96, 96], num_encoder_layers=[1, 2, 1, 1], feedforward_dims=[128, 128, 160, 160],
        num_heads=[2, 2, 4, 4], cnn_module_kernels=[7, 7, 5, 7], downsampling_factors=[1, 2, 4, 2], num_left_chunks=2,
        joiner_dim=512, decoder_dim=64, vocab_size=37,
    ),
}


LSTM_PRESETS = {
    "lstm-en": dict(),
    "lstm-tiny-test": dict(d_model=64, rnn_hidden_size=96, dim_feedforward=160, num_encoder_layers=3, joiner_dim=512, decoder_dim=64,
                           vocab_size=41),
    # wide enough for the split-K form of the projection / feed-forward products of the layer wavefront, 5 layers deep
    "lstm-tiny-split-test": dict(d_model=64, rnn_hidden_size=256, dim_feedforward=384, num_encoder_layers=5, joiner_dim=512, decoder_dim=64,
                                 vocab_size=41),
}


CONFORMER_PRESETS = {
    # BASELINE.json configs[4]: conformer-zh (wenetspeech char model), 12 x (512, 2048, 8 heads, k=31)
    "conformer-zh": dict(vocab_size=5537),
    # the same architecture as a streaming export (OnlineProjOfConformer): chunks of 16 frames at 25 Hz, 64 frames of left context
    "conformer-streaming-zh": dict(vocab_size=5537, streaming=True, chunk_size=16, left_context=64),
    # parity-test model: odd head size, small kernel, decoder conv with groups = 1
    "conformer-streaming-tiny-test": dict(encoder_dim=64, num_encoder_layers=2, feedforward_dim=160, num_heads=4, cnn_module_kernel=7,
                                          joiner_dim=512, decoder_dim=64, vocab_size=41, streaming=True, chunk_size=8, left_context=16),
    # the same with two frames of right context (OnlineModel.cs:161-165 reads the key): a chunk is chunk_size + 2 encoder frames, the last
    # two are seen by the attention and the convolution of this step and come again, as the first two, in the next one
    "conformer-streaming-rc-tiny-test": dict(encoder_dim=64, num_encoder_layers=2, feedforward_dim=160, num_heads=4, cnn_module_kernel=7,
                                             joiner_dim=512, decoder_dim=64, vocab_size=41, streaming=True, chunk_size=8, left_context=16,
                                             right_context=2),
    "conformer-tiny-test": dict(encoder_dim=64, num_encoder_layers=2, feedforward_dim=160, num_heads=4, cnn_module_kernel=7,
                                joiner_dim=512, decoder_dim=64, vocab_size=41),
}


# Architecture presets.  Dimensions are those of the public icefall recipes the
# reference's model zoo was exported from (README.EN.md:8-35 lists the model
# names; the dims themselves are external to the reference).
PRESETS = {
    # BASELINE.json configs[1]: the model the headline metric is quoted on.
    "zipformer2-large-en": dict(
        encoder_dims=[192, 256, 512, 768, 512, 256],
        num_encoder_layers=[2, 2, 4, 5, 4, 2],
        feedforward_dims=[512, 768, 1536, 2048, 1536, 768],
        num_heads=[4, 4, 4, 8, 4, 4],
        cnn_module_kernels=[31, 31, 15, 15, 15, 31],
        downsampling_factors=[1, 2, 4, 8, 4, 2],
        vocab_size=500,
    ),
    # BASELINE.json configs[0].
    "zipformer2-small-en": dict(
        encoder_dims=[192, 256, 256, 256, 256, 256],
        num_encoder_layers=[2, 2, 2, 2, 2, 2],
        feedforward_dims=[512, 768, 768, 768, 768, 768],
        num_heads=[4, 4, 4, 8, 4, 4],
        cnn_module_kernels=[31, 31, 15, 15, 15, 31],
        downsampling_factors=[1, 2, 4, 8, 4, 2],
        vocab_size=500,
    ),
    # BASELINE.json configs[3]: streaming multi-zh-hans (OnlineProjOfZipformer2), chunk 32 frames.
    # 16 layers (OnlineProjOfZipformer2.cs:125 confirms 16 x 6 caches); dims of the icefall
    # streaming recipe.
    "zipformer2-streaming-zh": dict(
        encoder_dims=[192, 256, 384, 512, 384, 256],
        num_encoder_layers=[2, 2, 3, 4, 3, 2],
        feedforward_dims=[512, 768, 1024, 1536, 1024, 768],
        num_heads=[4, 4, 4, 8, 4, 4],
        cnn_module_kernels=[31, 31, 15, 15, 15, 31],
        downsampling_factors=[1, 2, 4, 8, 4, 2],
        vocab_size=2000,
        streaming=True,
        chunk_size=16,
        left_context_frames=128,
    ),
    "zipformer2-streaming-tiny-test": dict(
        encoder_dims=[64, 96, 128, 64],
        num_encoder_layers=[1, 2, 1, 1],
        feedforward_dims=[128, 192, 256, 128],
        num_heads=[2, 2, 4, 2],
        cnn_module_kernels=[15, 7, 7, 15],
        downsampling_factors=[1, 2, 4, 2],
        vocab_size=37,
        joiner_dim=512,
        decoder_dim=64,
        streaming=True,
        chunk_size=16,
        left_context_frames=32,
    ),
    "zipformer2-ctc-tiny-test": dict(
        encoder_dims=[64, 96, 128, 64],
        num_encoder_layers=[1, 2, 1, 1],
        feedforward_dims=[128, 192, 256, 128],
        num_heads=[2, 2, 4, 2],
        cnn_module_kernels=[15, 7, 7, 15],
        downsampling_factors=[1, 2, 4, 2],
        vocab_size=37,
        ctc=True,
    ),
    "zipformer2-ctc-streaming-tiny-test": dict(
        encoder_dims=[64, 96, 128, 64],
        num_encoder_layers=[1, 2, 1, 1],
        feedforward_dims=[128, 192, 256, 128],
        num_heads=[2, 2, 4, 2],
        cnn_module_kernels=[15, 7, 7, 15],
        downsampling_factors=[1, 2, 4, 2],
        vocab_size=37,
        ctc=True,
        streaming=True,
        chunk_size=16,
        left_context_frames=32,
    ),
    # Parity-test model: every structural feature of the big one (unequal
    # stack dims -> channel pad/truncate + full-dim concat, three
    # downsampling factors, odd lengths) at sizes the CPU oracle runs in
    # well under a second.
    "zipformer2-tiny-test": dict(
        encoder_dims=[64, 96, 128, 64],
        num_encoder_layers=[1, 2, 1, 1],
        feedforward_dims=[128, 192, 256, 128],
        num_heads=[2, 2, 4, 2],
        cnn_module_kernels=[15, 7, 7, 15],
        downsampling_factors=[1, 2, 4, 2],
        vocab_size=37,
        # the offline loops hard-code 512 (OfflineRecognizer.cs:103,136,201,219)
        joiner_dim=512,
        decoder_dim=64,
    ),
}


def preset(name: str) -> dict:
    if name in CONFORMER_PRESETS:
        return make_conformer_meta(comment=name, **copy.deepcopy(CONFORMER_PRESETS[name]))
    if name in LSTM_PRESETS:
        return make_lstm_meta(comment=name, **copy.deepcopy(LSTM_PRESETS[name]))
    if name in ZIPFORMER1_PRESETS:
        return make_zipformer_meta(comment=name, **copy.deepcopy(ZIPFORMER1_PRESETS[name]))
    if name not in PRESETS:
        raise KeyError(f"unknown model preset {name!r}; have {sorted(PRESETS) + sorted(CONFORMER_PRESETS)}")
    return make_zipformer2_meta(comment=name, **copy.deepcopy(PRESETS[name]))


def ints(meta: dict, key: str):
    return [int(x) for x in meta[key].split(",") if x != ""]
