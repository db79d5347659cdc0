"""A decoder/joiner whose outputs can be computed by hand (used by the greedy KATs).

V = 8, decoder_dim = 4, joiner_dim = 8:
  emb[0] = [0.5,0,0,0], emb[v] = [v,0,0,0]            (blank has a NON-zero embedding)
  conv: h0 = relu(e_prev[0] + e_cur[0]), h1..3 = 0       (one group of 4 channels, k = 2)
  decoder_proj: dec = [0,0,0,0.1*h0,0,0,0,0]
  output_linear = I, bias 0            ->  logits = tanh(enc + dec)
So only token 3's logit depends on the context:  ctx [-1,0] -> +0.05, [0,0] -> +0.1,
[0,y] -> 0.05+0.1y, [y1,y2] -> 0.1(y1+y2).
"""
import numpy as np

from k2transducerasr_amd.config import make_zipformer2_meta
from k2transducerasr_amd.k2w import write_k2w

V, DD, J = 8, 4, 8


def write_kat_model(path, bias=None):
    """bias: joiner.output_linear.bias (default zeros) -- the NaN-rule tests put NaNs there (a NaN logit at a chosen index)"""
    meta = make_zipformer2_meta(encoder_dims=[16], num_encoder_layers=[1], feedforward_dims=[16], num_heads=[1],
                                cnn_module_kernels=[3], downsampling_factors=[1], joiner_dim=J, decoder_dim=DD,
                                vocab_size=V, comment="greedy-kat")
    emb = np.zeros((V, DD), np.float32)
    emb[:, 0] = np.arange(V)
    emb[0, 0] = 0.5
    conv = np.zeros((DD, 4, 2), np.float32)
    conv[0, 0, 0] = 1.0
    conv[0, 0, 1] = 1.0
    dproj = np.zeros((J, DD), np.float32)
    dproj[3, 0] = 0.1
    tensors = [
        ("decoder.embedding.weight", emb),
        ("decoder.conv.weight", conv),
        ("joiner.decoder_proj.weight", dproj),
        ("joiner.decoder_proj.bias", np.zeros(J, np.float32)),
        ("joiner.output_linear.weight", np.eye(V, J, dtype=np.float32)),
        ("joiner.output_linear.bias", np.zeros(V, np.float32) if bias is None else np.asarray(bias, np.float32)),
    ]
    write_k2w(path, meta, tensors)
    return meta


def write_wide_model(path, vocab, joiner_dim=64, decoder_dim=8, nan_at=(), seed=3):
    """A decoder / joiner with random weights and a vocabulary wide enough for the search kernels' column slabs and passes
    (no encoder tensors: only the operator-level search entries are used); output bias NaN at the indexes `nan_at`."""
    rng = np.random.default_rng(seed)
    meta = make_zipformer2_meta(encoder_dims=[16], num_encoder_layers=[1], feedforward_dims=[16], num_heads=[1],
                                cnn_module_kernels=[3], downsampling_factors=[1], joiner_dim=joiner_dim, decoder_dim=decoder_dim,
                                vocab_size=vocab, comment="wide-kat")
    bias = (rng.standard_normal(vocab) * 0.5).astype(np.float32)
    bias[0] = 5.2           # blank wins a good share of the frames
    for p in nan_at:
        bias[p] = np.nan
    tensors = [
        ("decoder.embedding.weight", rng.standard_normal((vocab, decoder_dim)).astype(np.float32)),
        ("decoder.conv.weight", (rng.standard_normal((decoder_dim, 4, 2)) * 0.5).astype(np.float32)),
        ("joiner.decoder_proj.weight", (rng.standard_normal((joiner_dim, decoder_dim)) * 0.3).astype(np.float32)),
        ("joiner.decoder_proj.bias", np.zeros(joiner_dim, np.float32)),
        ("joiner.output_linear.weight", (rng.standard_normal((vocab, joiner_dim)) * 0.3).astype(np.float32)),
        ("joiner.output_linear.bias", bias),
    ]
    write_k2w(path, meta, tensors)
    return meta


def frames(rows):
    """rows: list of {token: value}; returns enc_out [T', J] with those pre-tanh inputs."""
    e = np.full((len(rows), J), -3.0, np.float32)
    for t, r in enumerate(rows):
        for k, v in r.items():
            e[t, k] = v
    return e


# ---- hand-derived cases ----------------------------------------------------------
# Every unspecified logit is tanh(-3 [+ boost]) < -0.98, i.e. never the maximum.
# Token 3's boost: ctx [-1,0] -> 0.05, [0,0] -> 0.1, [0,y] -> 0.05 + 0.1 y, [y1,y2] -> 0.1 (y1+y2).
CASES = {
    # blank (0) and unk (2) are never emitted (OfflineRecognizer.cs:268); 5 is.
    "skip_blank_unk": dict(
        streams=[frames([{0: 1.0}, {2: 1.0}, {5: 1.0}, {0: 1.0}])],
        batch=[([5], [2])], single=[([5], [2])]),
    # two equal maxima: the later index wins (OfflineRecognizer.cs:239).
    #   t0: 4 and 6 tie -> 6 ; t1: 0 and 1 tie -> 1, emitted (id 1 is only skipped by the ONLINE loop,
    #   OnlineRecognizer.cs:181) ; t2: 0 and 2 tie -> 2 = unk -> skipped
    "tie_later_index": dict(
        streams=[frames([{4: 0.7, 6: 0.7}, {0: 0.7, 1: 0.7}, {0: 0.3, 2: 0.3}])],
        batch=[([6, 1], [0, 1])], single=[([6, 1], [0, 1])]),
    # the decoder context feeds back: the same frame {3: 0.0, 4: 0.2} three times
    #   t0 ctx [-1,0]: tanh(0.05) < tanh(0.2) -> 4 ; t1 ctx [0,4]: tanh(0.45) > tanh(0.2) -> 3 ;
    #   t2 ctx [4,3]: tanh(0.7) -> 3 ; t3: blank
    "context_feedback": dict(
        streams=[frames([{3: 0.0, 4: 0.2}, {3: 0.0, 4: 0.2}, {3: 0.0, 4: 0.2}, {0: 5.0}])],
        batch=[([4, 3, 3], [0, 1, 2])], single=[([4, 3, 3], [0, 1, 2])]),
    # the batch quirk (OfflineRecognizer.cs:250-258, :278-286): A emits at t=0, which re-runs the decoder
    # for B on its seeded [blank, blank]; B's boost goes 0.05 -> 0.1 from t=1 on.
    #   B, t1 = {3: 0.0, 0: 0.07}: alone tanh(0.05) < tanh(0.07) -> blank, nothing emitted;
    #   in the batch tanh(0.1) > tanh(0.07) -> emits 3 at t=1.
    "batch_context_switch": dict(
        streams=[frames([{5: 1.0}, {0: 1.0}, {0: 1.0}]),
                 frames([{0: 1.0}, {3: 0.0, 0: 0.07}, {0: 1.0}])],
        batch=[([5], [0]), ([3], [1])], single=[([5], [0]), ([], [])]),
}
