"""Seeded synthetic model weights and audio (there are no real checkpoints or
wavs in this environment; SURVEY.md section 8d defines the recipe).

Weights are N(0, 1/sqrt(fan_in)) f32 under icefall state-dict names.  The same
``.k2w`` file is read by the CPU oracle and by the HIP engine, so parity never
depends on the generator.
"""
from __future__ import annotations

import zlib
from typing import Dict, List, Tuple

import numpy as np

from .config import ints, preset
from .k2w import write_k2w

AUDIO_SEED_BASE = 0x2A50000

# Joiner blank-bias per preset, frozen from tools/calibrate_blank_bias.py so that
# roughly a quarter to a tenth of the frames emit a symbol on synth_utterance().
BLANK_BIAS = {
    "zipformer2-large-en": 3.436,     # tools/calibrate_blank_bias.py ... 10 0.2 batch 32: 20 % of the frames of the benchmark batch emit
    "zipformer2-small-en": 2.849,
    "zipformer2-tiny-test": 1.033,
    "zipformer2-streaming-zh": 2.261,  # ~20 % emission under the online greedy loop
    "zipformer2-streaming-tiny-test": 1.0,
    "conformer-zh": 2.615,
    "conformer-streaming-zh": 2.615,
    "lstm-en": 3.0,
    "lstm-tiny-test": 2.0,
    "lstm-tiny-split-test": 2.0,
    "zipformer2-ctc-tiny-test": 1.5,
    "zipformer2-ctc-streaming-tiny-test": 1.5,
    "conformer-tiny-test": 2.179,
    "conformer-streaming-tiny-test": 2.0,
    "zipformer-streaming-en": 3.0,
    "zipformer-en": 3.0,
    "zipformer-tiny-test": 1.5,
    "zipformer-streaming-tiny-test": 1.5,
}


def _rng(seed: int, name: str) -> np.random.Generator:
    return np.random.default_rng([seed, zlib.crc32(name.encode())])


def zipformer2_tensor_specs(meta: Dict[str, str]) -> List[Tuple[str, tuple, str]]:
    """(name, shape, kind) for every tensor of an offline Zipformer2 transducer.

    kind: "w" weight (fan_in = prod(shape[1:])), "b" bias, "bypass", "logscale",
    "dsbias", "emb".
    """
    dims = ints(meta, "encoder_dims")
    layers = ints(meta, "num_encoder_layers")
    ffs = ints(meta, "feedforward_dims")
    heads = ints(meta, "num_heads")
    kernels = ints(meta, "cnn_module_kernels")
    dss = ints(meta, "downsampling_factors")
    qhd = ints(meta, "query_head_dims")
    vhd = ints(meta, "value_head_dims")
    phd = ints(meta, "pos_head_dims")
    pos_dim = int(meta["pos_dim"])
    J = int(meta["joiner_dim"])
    DD = int(meta["decoder_dim"])
    V = int(meta["vocab_size"])
    ctx = int(meta["context_size"])
    fdim = int(meta["feature_dim"])
    out_width = (((fdim - 1) // 2) - 1) // 2

    s: List[Tuple[str, tuple, str]] = []

    def lin(prefix, out_f, in_f, bias=True):
        s.append((prefix + ".weight", (out_f, in_f), "w"))
        if bias:
            s.append((prefix + ".bias", (out_f,), "b"))

    e = "encoder_embed."
    s += [
        (e + "conv.0.weight", (8, 1, 3, 3), "w"),
        (e + "conv.0.bias", (8,), "b"),
        (e + "conv.4.weight", (32, 8, 3, 3), "w"),
        (e + "conv.4.bias", (32,), "b"),
        (e + "conv.7.weight", (128, 32, 3, 3), "w"),
        (e + "conv.7.bias", (128,), "b"),
        (e + "convnext.depthwise_conv.weight", (128, 1, 7, 7), "w"),
        (e + "convnext.depthwise_conv.bias", (128,), "b"),
        (e + "convnext.pointwise_conv1.weight", (384, 128, 1, 1), "w"),
        (e + "convnext.pointwise_conv1.bias", (384,), "b"),
        (e + "convnext.pointwise_conv2.weight", (128, 384, 1, 1), "w"),
        (e + "convnext.pointwise_conv2.bias", (128,), "b"),
    ]
    lin(e + "out", dims[0], 128 * out_width)
    s += [(e + "out_norm.log_scale", (1,), "logscale"), (e + "out_norm.bias", (dims[0],), "b")]

    for i, (D, L, F, H, K, ds) in enumerate(zip(dims, layers, ffs, heads, kernels, dss)):
        st = f"encoder.encoders.{i}."
        if ds > 1:
            s.append((st + "downsample.bias", (ds,), "dsbias"))
            s.append((st + "out_combiner.bypass_scale", (D,), "bypass"))
        for j in range(L):
            p = st + f"layers.{j}."
            lin(p + "self_attn_weights.in_proj", (2 * qhd[i] + phd[i]) * H, D)
            lin(p + "self_attn_weights.linear_pos", phd[i] * H, pos_dim, bias=False)
            for k, Fk in ((1, F * 3 // 4), (2, F), (3, F * 5 // 4)):
                lin(p + f"feed_forward{k}.in_proj", Fk, D)
                lin(p + f"feed_forward{k}.out_proj", D, Fk)
            Hc = 3 * D // 4
            lin(p + "nonlin_attention.in_proj", 3 * Hc, D)
            lin(p + "nonlin_attention.out_proj", D, Hc)
            for k in (1, 2):
                lin(p + f"self_attn{k}.in_proj", vhd[i] * H, D)
                lin(p + f"self_attn{k}.out_proj", D, vhd[i] * H)
            for k in (1, 2):
                lin(p + f"conv_module{k}.in_proj", 2 * D, D)
                if meta.get("streaming") == "1":
                    # ChunkCausalDepthwiseConv1d (icefall zipformer.py): causal half-kernel conv +
                    # chunk-wise full-kernel conv scaled near the chunk edges
                    dc = p + f"conv_module{k}.depthwise_conv."
                    s.append((dc + "causal_conv.weight", (D, 1, (K + 1) // 2), "w"))
                    s.append((dc + "causal_conv.bias", (D,), "b"))
                    s.append((dc + "chunkwise_conv.weight", (D, 1, K), "w"))
                    s.append((dc + "chunkwise_conv.bias", (D,), "b"))
                    s.append((dc + "chunkwise_conv_scale", (2, D, K), "b"))
                else:
                    s.append((p + f"conv_module{k}.depthwise_conv.weight", (D, 1, K), "w"))
                    s.append((p + f"conv_module{k}.depthwise_conv.bias", (D,), "b"))
                lin(p + f"conv_module{k}.out_proj", D, D)
            s.append((p + "norm.log_scale", (1,), "logscale"))
            s.append((p + "norm.bias", (D,), "b"))
            s.append((p + "bypass.bypass_scale", (D,), "bypass"))
            s.append((p + "bypass_mid.bypass_scale", (D,), "bypass"))
    s.append(("encoder.downsample_output.bias", (2,), "dsbias"))
    if meta["model_type"] == "zipformer2ctc":
        lin("ctc_output.1", V, max(dims))  # icefall: ctc_output = Sequential(Dropout, Linear, LogSoftmax)
        return s
    lin("joiner.encoder_proj", J, max(dims))
    lin("joiner.decoder_proj", J, DD)
    lin("joiner.output_linear", V, J)
    s.append(("decoder.embedding.weight", (V, DD), "emb"))
    s.append(("decoder.conv.weight", (DD, 4, ctx), "w"))
    return s


def conformer_tensor_specs(meta: Dict[str, str]) -> List[Tuple[str, tuple, str]]:
    """Tensors of an offline Conformer transducer under the state-dict names of icefall's
    pruned_transducer_stateless2 (Scaled* modules folded to plain weights, as an export does)."""
    D = ints(meta, "encoder_dims")[0]
    L = ints(meta, "num_encoder_layers")[0]
    F = ints(meta, "feedforward_dims")[0]
    H = ints(meta, "num_heads")[0]
    K = ints(meta, "cnn_module_kernels")[0]
    J = int(meta["joiner_dim"])
    DD = int(meta["decoder_dim"])
    V = int(meta["vocab_size"])
    ctx = int(meta["context_size"])
    fdim = int(meta["feature_dim"])
    out_width = (((fdim - 1) // 2) - 1) // 2
    s: List[Tuple[str, tuple, str]] = []

    def lin(prefix, out_f, in_f, bias=True):
        s.append((prefix + ".weight", (out_f, in_f), "w"))
        if bias:
            s.append((prefix + ".bias", (out_f,), "b"))

    e = "encoder.encoder_embed."
    s += [
        (e + "conv.0.weight", (8, 1, 3, 3), "w"),
        (e + "conv.0.bias", (8,), "b"),
        (e + "conv.3.weight", (32, 8, 3, 3), "w"),
        (e + "conv.3.bias", (32,), "b"),
        (e + "conv.6.weight", (128, 32, 3, 3), "w"),
        (e + "conv.6.bias", (128,), "b"),
    ]
    lin(e + "out", D, 128 * out_width)
    s.append((e + "out_norm.eps", (1,), "logeps"))
    for j in range(L):
        p = f"encoder.encoder.layers.{j}."
        for ff in ("feed_forward_macaron", "feed_forward"):
            lin(p + ff + ".0", F, D)
            lin(p + ff + ".4", D, F)
        lin(p + "self_attn.in_proj", 3 * D, D)
        lin(p + "self_attn.out_proj", D, D)
        lin(p + "self_attn.linear_pos", D, D, bias=False)
        s.append((p + "self_attn.pos_bias_u", (H, D // H), "b"))
        s.append((p + "self_attn.pos_bias_v", (H, D // H), "b"))
        s.append((p + "conv_module.pointwise_conv1.weight", (2 * D, D, 1), "w"))
        s.append((p + "conv_module.pointwise_conv1.bias", (2 * D,), "b"))
        s.append((p + "conv_module.depthwise_conv.weight", (D, 1, K), "w"))
        s.append((p + "conv_module.depthwise_conv.bias", (D,), "b"))
        s.append((p + "conv_module.pointwise_conv2.weight", (D, D, 1), "w"))
        s.append((p + "conv_module.pointwise_conv2.bias", (D,), "b"))
        s.append((p + "norm_final.eps", (1,), "logeps"))
    lin("joiner.encoder_proj", J, D)
    lin("joiner.decoder_proj", J, DD)
    lin("joiner.output_linear", V, J)
    s.append(("decoder.embedding.weight", (V, DD), "emb"))
    s.append(("decoder.conv.weight", (DD, DD, ctx), "w"))  # stateless2 decoder: groups = 1
    return s


def lstm_tensor_specs(meta: Dict[str, str]) -> List[Tuple[str, tuple, str]]:
    """icefall lstm_transducer_stateless2 state dict (Scaled* folded); torch.nn.LSTM parameter names and gate order i,f,g,o."""
    D = int(meta["d_model"])
    Hh = int(meta["rnn_hidden_size"])
    F = ints(meta, "feedforward_dims")[0]
    L = ints(meta, "num_encoder_layers")[0]
    J = int(meta["joiner_dim"])
    DD = int(meta["decoder_dim"])
    V = int(meta["vocab_size"])
    ctx = int(meta["context_size"])
    fdim = int(meta["feature_dim"])
    out_width = (((fdim - 1) // 2) - 1) // 2
    s: List[Tuple[str, tuple, str]] = []

    def lin(prefix, out_f, in_f, bias=True):
        s.append((prefix + ".weight", (out_f, in_f), "w"))
        if bias:
            s.append((prefix + ".bias", (out_f,), "b"))

    e = "encoder.encoder_embed."
    s += [
        (e + "conv.0.weight", (8, 1, 3, 3), "w"), (e + "conv.0.bias", (8,), "b"),
        (e + "conv.3.weight", (32, 8, 3, 3), "w"), (e + "conv.3.bias", (32,), "b"),
        (e + "conv.6.weight", (128, 32, 3, 3), "w"), (e + "conv.6.bias", (128,), "b"),
    ]
    lin(e + "out", D, 128 * out_width)
    s.append((e + "out_norm.eps", (1,), "logeps"))
    for j in range(L):
        p = f"encoder.encoder.layers.{j}."
        s += [
            (p + "lstm.weight_ih_l0", (4 * Hh, D), "w"),
            (p + "lstm.weight_hh_l0", (4 * Hh, D), "w"),
            (p + "lstm.bias_ih_l0", (4 * Hh,), "b"),
            (p + "lstm.bias_hh_l0", (4 * Hh,), "b"),
            (p + "lstm.weight_hr_l0", (D, Hh), "w"),
        ]
        lin(p + "feed_forward.0", F, D)
        lin(p + "feed_forward.4", D, F)
        s.append((p + "norm_final.eps", (1,), "logeps"))
    lin("joiner.encoder_proj", J, D)
    lin("joiner.decoder_proj", J, DD)
    lin("joiner.output_linear", V, J)
    s.append(("decoder.embedding.weight", (V, DD), "emb"))
    s.append(("decoder.conv.weight", (DD, DD, ctx), "w"))
    return s


def zipformer1_tensor_specs(meta: Dict[str, str]) -> List[Tuple[str, tuple, str]]:
    """icefall pruned_transducer_stateless7_streaming state dict (Scaled* folded): Zipformer v1."""
    dims = ints(meta, "encoder_dims")
    atts = ints(meta, "attention_dims")
    layers = ints(meta, "num_encoder_layers")
    ffs = ints(meta, "feedforward_dims")
    heads = ints(meta, "num_heads")
    kernels = ints(meta, "cnn_module_kernels")
    dss = ints(meta, "downsampling_factors")
    pos_dim = int(meta["pos_dim"])
    J = int(meta["joiner_dim"])
    DD = int(meta["decoder_dim"])
    V = int(meta["vocab_size"])
    ctx = int(meta["context_size"])
    fdim = int(meta["feature_dim"])
    out_width = (((fdim - 1) // 2) - 1) // 2
    s: List[Tuple[str, tuple, str]] = []

    def lin(prefix, out_f, in_f, bias=True):
        s.append((prefix + ".weight", (out_f, in_f), "w"))
        if bias:
            s.append((prefix + ".bias", (out_f,), "b"))

    e = "encoder.encoder_embed."
    s += [
        (e + "conv.0.weight", (8, 1, 3, 3), "w"), (e + "conv.0.bias", (8,), "b"),
        (e + "conv.3.weight", (32, 8, 3, 3), "w"), (e + "conv.3.bias", (32,), "b"),
        (e + "conv.6.weight", (128, 32, 3, 3), "w"), (e + "conv.6.bias", (128,), "b"),
    ]
    lin(e + "out", dims[0], 128 * out_width)
    for i, (D, A, L, F, H, K, ds) in enumerate(zip(dims, atts, layers, ffs, heads, kernels, dss)):
        st = f"encoder.encoders.{i}."
        if ds > 1:
            Din = dims[i - 1] if i > 0 else dims[0]
            s.append((st + "downsample.query", (Din,), "w1"))
            if D > Din:
                s.append((st + "downsample.extra_proj.weight", (D - Din, Din * ds), "w"))
            s.append((st + "upsample.bias", (ds, D), "b"))
            s.append((st + "out_combiner.weight1", (1,), "mix"))
            st += "encoder."
        if i >= 2 and dss[i - 1] > ds:
            s.append((f"encoder.skip_modules.{i}.weight1", (1,), "mix"))
        for j in range(L):
            p = st + f"layers.{j}."
            for k in (1, 2, 3):
                lin(p + f"feed_forward{k}.in_proj", F, D)
                lin(p + f"feed_forward{k}.out_proj", D, F)
            lin(p + "pooling.proj", D, D, bias=False)
            lin(p + "self_attn.in_proj", 2 * A + A // 2 + pos_dim * H, D)
            lin(p + "self_attn.linear_pos", pos_dim * H, D, bias=False)
            lin(p + "self_attn.out_proj", D, A // 2)
            lin(p + "self_attn.in_proj2", A // 2, D, bias=False)
            lin(p + "self_attn.out_proj2", D, A // 2)
            for k in (1, 2):
                s.append((p + f"conv_module{k}.pointwise_conv1.weight", (2 * D, D, 1), "w"))
                s.append((p + f"conv_module{k}.pointwise_conv1.bias", (2 * D,), "b"))
                s.append((p + f"conv_module{k}.depthwise_conv.weight", (D, 1, K), "w"))
                s.append((p + f"conv_module{k}.depthwise_conv.bias", (D,), "b"))
                s.append((p + f"conv_module{k}.pointwise_conv2.weight", (D, D, 1), "w"))
                s.append((p + f"conv_module{k}.pointwise_conv2.bias", (D,), "b"))
            s.append((p + "norm_final.eps", (1,), "logeps"))
            s.append((p + "bypass_scale", (1,), "bypass"))
    s.append(("encoder.downsample_output.query", (dims[-1],), "w1"))
    lin("joiner.encoder_proj", J, dims[-1])
    lin("joiner.decoder_proj", J, DD)
    lin("joiner.output_linear", V, J)
    s.append(("decoder.embedding.weight", (V, DD), "emb"))
    s.append(("decoder.conv.weight", (DD, 4, ctx), "w"))
    return s


def tensor_specs(meta: Dict[str, str]) -> List[Tuple[str, tuple, str]]:
    if meta["model_type"] == "zipformer":
        return zipformer1_tensor_specs(meta)
    if meta["model_type"] == "conformer":
        return conformer_tensor_specs(meta)
    if meta["model_type"] == "lstm":
        return lstm_tensor_specs(meta)
    return zipformer2_tensor_specs(meta)


def _init(name: str, shape: tuple, kind: str, seed: int) -> np.ndarray:
    g = _rng(seed, name)
    if kind == "w":
        fan_in = int(np.prod(shape[1:]))
        return (g.standard_normal(shape) / np.sqrt(fan_in)).astype(np.float32)
    if kind == "b":
        return (0.1 * g.standard_normal(shape)).astype(np.float32)
    if kind == "bypass":
        # small per-layer bypass scales keep a random-weight stack close to the
        # identity, so the input's temporal structure survives 19 layers
        lo, hi = (0.3, 0.9) if "out_combiner" in name else (0.05, 0.25)
        return g.uniform(lo, hi, shape).astype(np.float32)
    if kind == "w1":  # a vector used as a dot-product weight (AttentionDownsample.query)
        return (g.standard_normal(shape) / np.sqrt(shape[0])).astype(np.float32)
    if kind == "mix":  # SimpleCombiner.weight1
        return g.uniform(0.2, 0.8, shape).astype(np.float32)
    if kind == "logscale":
        return g.uniform(-0.2, 0.2, shape).astype(np.float32)
    if kind == "dsbias":
        return (0.5 * g.standard_normal(shape)).astype(np.float32)
    if kind == "emb":
        return g.standard_normal(shape).astype(np.float32)
    if kind == "logeps":
        return np.log(g.uniform(0.1, 0.5, shape)).astype(np.float32)
    raise ValueError(kind)


def write_synthetic_model(path: str, preset_name: str, seed: int = 20231212, blank_bias: float | None = None,
                          meta_overrides: Dict[str, str] | None = None) -> Dict[str, str]:
    """Write a seeded random-weight Zipformer2 transducer.

    ``blank_bias`` is added to the joiner's output bias at id 0 so that blank
    wins most frames, as it does on speech; it is recorded in
    the metadata for traceability only.
    """
    meta = preset(preset_name)
    if blank_bias is None:
        blank_bias = BLANK_BIAS.get(preset_name, 1.0)
    if meta_overrides:
        meta.update(meta_overrides)
    meta["synthetic_seed"] = str(seed)
    meta["synthetic_blank_bias"] = repr(float(blank_bias))

    conformer = meta["model_type"] in ("conformer", "lstm", "zipformer")  # residual branches without a per-channel bypass

    def gen():
        for name, shape, kind in tensor_specs(meta):
            a = _init(name, shape, kind, seed)
            if name in ("joiner.output_linear.bias", "ctc_output.1.bias"):
                a[0] += np.float32(blank_bias)
            if name == "ctc_output.1.weight":
                a *= np.float32(4.0)
            if name == "joiner.encoder_proj.weight":
                a *= np.float32(4.0)  # let the (small) temporal variation reach the logits
            if conformer and (name.endswith(".4.weight") or name.endswith("out_proj.weight") or name.endswith("pointwise_conv2.weight")
                              or name.endswith("out_proj2.weight") or name.endswith("pooling.proj.weight")):
                a *= np.float32(0.25)  # the residual branches' output layers (icefall initial_scale = 0.25)
            if name.endswith("encoder_embed.conv.0.weight"):
                # zero-sum along time: a constant-in-time input (the large DC of
                # log-mel features) maps to 0, so that random-weight activations
                # keep their temporal variation instead of collapsing to a bias.
                a -= a.mean(axis=2, keepdims=True)
                a *= np.float32(4.0)
            yield name, a

    write_k2w(path, meta, gen())
    return meta


def synth_utterance(u: int, seconds: float, sample_rate: int = 16000) -> np.ndarray:
    """Deterministic speech-like test signal, float32 in [-0.99, 0.99], no exact zeros."""
    g = np.random.default_rng(AUDIO_SEED_BASE + int(u))
    n = int(round(seconds * sample_rate))
    t = np.arange(n, dtype=np.float64) / sample_rate
    x = np.zeros(n, dtype=np.float64)
    for _ in range(8):
        f = g.uniform(80.0, 7600.0)
        a = g.uniform(0.01, 0.1)
        ph = g.uniform(0.0, 2 * np.pi)
        x += a * np.sin(2 * np.pi * f * t + ph)
    env = 0.5 - 0.5 * np.cos(2 * np.pi * 3.0 * t + g.uniform(0.0, 2 * np.pi))
    x = x * (0.15 + 0.85 * env) + 0.01 * g.standard_normal(n)
    x = np.clip(x, -0.99, 0.99).astype(np.float32)
    x[x == 0.0] = np.float32(1e-7)
    return x
