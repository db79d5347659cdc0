"""GPU parity for the offline Conformer path (BASELINE.json configs[4]): libk2hip.so through the C
ABI against oracle/k2_oracle_conformer.c on the same seeded inputs."""
import numpy as np
import pytest

from parity import ACT_TOL, LOGIT_TOL, assert_tokens_match

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def feats(oracle_conformer, utts):
    return [oracle_conformer.fbank(u) for u in utts]


def test_conformer_out_frames(hip_conformer, oracle_conformer):
    for T in (7, 8, 9, 100, 101, 103, 1017, 3017):
        assert hip_conformer.encoder_out_frames(T) == oracle_conformer.encoder_out_frames(T)
    assert hip_conformer.encoder_out_frames(3017) == 753


@pytest.mark.parametrize("tap", [0, 1, 2])
def test_conformer_encoder_taps(hip_conformer, oracle_conformer, feats, tap):
    x = oracle_conformer.pad_sequence(feats[:3]).reshape(3, -1, 80)
    a = hip_conformer.encoder_tap(x, tap)
    b = oracle_conformer.encoder_tap(x, tap)
    assert a.shape == b.shape
    np.testing.assert_allclose(a, b, atol=ACT_TOL, rtol=0)


def test_conformer_encoder_proj(hip_conformer, oracle_conformer, feats):
    x = oracle_conformer.pad_sequence(feats).reshape(len(feats), -1, 80)
    a = hip_conformer.encoder_proj(x)
    b = oracle_conformer.encoder(x)
    assert a.shape == b.shape
    np.testing.assert_allclose(a, b, atol=ACT_TOL, rtol=0)


def test_conformer_odd_lengths(hip_conformer, oracle_conformer):
    # every residue of T mod 4 through the two stride-2 convs, T' not a multiple of 4 (padded score rows)
    rng = np.random.default_rng(5)
    for T in (23, 24, 25, 26, 61):
        x = rng.standard_normal((2, T, 80)).astype(np.float32)
        np.testing.assert_allclose(hip_conformer.encoder_proj(x), oracle_conformer.encoder(x), atol=ACT_TOL, rtol=0)


def test_conformer_decoder_groups_one(hip_conformer, oracle_conformer):
    y = np.array([[-1, 0], [0, 0], [5, 7], [40, 1], [-1, -1], [3, 3]], np.int64)
    np.testing.assert_allclose(hip_conformer.decoder_proj(y), oracle_conformer.decoder(y), atol=2e-5, rtol=0)


def test_conformer_greedy_on_oracle_encoder_out(hip_conformer, oracle_conformer, feats):
    x = oracle_conformer.pad_sequence(feats).reshape(len(feats), -1, 80)
    enc = oracle_conformer.encoder(x)
    want, mg = oracle_conformer.greedy_batch(enc, want_margins=True)
    assert sum(len(w[0]) for w in want) > 0
    assert_tokens_match(hip_conformer.greedy_batch(enc), want, mg, what="conformer greedy_batch")
    for b in range(len(feats)):
        want1, mg1 = oracle_conformer.greedy_single(enc[b], want_margins=True)
        assert_tokens_match([hip_conformer.greedy_single(enc[b])], [want1], mg1, what=f"conformer greedy_single[{b}]")


def test_conformer_fused_offline_greedy(hip_conformer, oracle_conformer, feats, utts):
    want = oracle_conformer.recognize_batch(feats)
    x = oracle_conformer.pad_sequence(feats).reshape(len(feats), -1, 80)
    _, mg = oracle_conformer.greedy_batch(oracle_conformer.encoder(x), want_margins=True)
    assert_tokens_match(hip_conformer.offline_greedy(feats), want, mg, what="conformer offline_greedy")
    assert_tokens_match(hip_conformer.offline_greedy_from_samples(utts), want, mg, what="conformer from_samples")


# ---------------------------------------------------------------- full architecture (conformer-zh, random weights)
@pytest.fixture(scope="module")
def zh_path(tmp_path_factory):
    from k2transducerasr_amd.synth import write_synthetic_model
    p = str(tmp_path_factory.mktemp("conformer_zh") / "conformer_zh.k2w")
    write_synthetic_model(p, "conformer-zh")
    return p


@pytest.fixture(scope="module")
def hip_zh(zh_path):
    from k2transducerasr_amd import Model
    return Model(zh_path, 0)


@pytest.fixture(scope="module")
def oracle_zh(zh_path):
    from oracle import Oracle
    return Oracle(zh_path)


def test_conformer_zh_matches_oracle(hip_zh, oracle_zh):
    """12 x (512, 2048, 8 heads, k = 31), V = 5537, decoder conv groups = 1, at a size the oracle finishes in seconds."""
    from k2transducerasr_amd.synth import synth_utterance
    utts = [synth_utterance(300 + u, s) for u, s in enumerate([5.0, 3.7])]
    feats = [oracle_zh.fbank(u) for u in utts]
    x = oracle_zh.pad_sequence(feats).reshape(len(utts), -1, 80)
    enc_o = oracle_zh.encoder(x)
    enc_h = hip_zh.encoder_proj(x)
    assert enc_h.shape == enc_o.shape
    np.testing.assert_allclose(enc_h, enc_o, atol=5e-4, rtol=0)
    dec = oracle_zh.decoder(np.array([[-1, 0]], np.int64))
    lo = oracle_zh.joiner(enc_o[0], np.repeat(dec, enc_o.shape[1], 0))
    lh = hip_zh.joiner_proj(enc_h[0], np.repeat(dec, enc_o.shape[1], 0))
    assert float(np.abs(lo - lh).max()) < LOGIT_TOL
    want, mg = oracle_zh.greedy_batch(enc_o, want_margins=True)
    assert sum(len(w[0]) for w in want) > 0
    assert_tokens_match(hip_zh.offline_greedy_from_samples(utts), want, mg, what="conformer-zh e2e")


def test_conformer_zh_full_size_properties(hip_zh):
    """BASELINE configs[4] per-GPU shard: 8 x 30 s -> T = 3017, T' = 753.  Size-independent properties: equal-length rows
    do not depend on their batch mates (x_lens = T for all, no masks); determinism; geometry."""
    from k2transducerasr_amd.synth import synth_utterance
    B = 8
    s = np.stack([synth_utterance(400 + u, 30.0) for u in range(B)])
    feats = [hip_zh.fbank(s[b]) for b in range(B)]
    assert feats[0].shape == (2998, 80)
    x = hip_zh.pad_sequence(feats).reshape(B, -1, 80)
    assert x.shape[1] == 3017
    enc = hip_zh.encoder_proj(x)
    assert enc.shape == (B, 753, 512) and np.isfinite(enc).all()
    alone = hip_zh.encoder_proj(x[5:6])
    np.testing.assert_allclose(enc[5], alone[0], atol=5e-5, rtol=0)
    r1 = hip_zh.offline_greedy_from_samples(list(s))
    r2 = hip_zh.offline_greedy_from_samples(list(s))
    assert r1 == r2
    assert all(len(t) == len(ts) and all(0 <= q < 753 for q in ts) and ts == sorted(ts) for t, ts in r1)
    assert sum(len(t) for t, _ in r1) > 0


def test_conformer_zh_full_size_matches_oracle(hip_zh, oracle_zh):
    """BASELINE configs[4] per-GPU shard at its own size against the oracle, strictly: 8 x 30 s -> T = 3017, T' = 753, V = 5537
    (6 024 argmax decisions over 16 column slabs per stream).  Fused entry (synchronous and pipelined) against `recognize_batch`;
    two rows of the encoder output (5e-4) and their logits (1e-3)."""
    from k2transducerasr_amd.synth import synth_utterance
    B = 8
    s = np.stack([synth_utterance(400 + u, 30.0) for u in range(B)])
    feats = [oracle_zh.fbank(s[b]) for b in range(B)]
    x = oracle_zh.pad_sequence(feats).reshape(B, -1, 80)
    enc_o = oracle_zh.encoder(x)
    want, mg = oracle_zh.greedy_batch(enc_o, want_margins=True)
    assert sum(len(w[0]) for w in want) > 8 * 753 // 20
    assert_tokens_match(hip_zh.offline_greedy_from_samples(list(s)), want, mg, what="configs[4] shard full size, synchronous")
    ptr = hip_zh.device_alloc(s.nbytes)
    try:
        hip_zh.device_upload(ptr, s)
        ta = hip_zh.offline_submit_samples_dev(ptr, s.shape[1], B)
        tb = hip_zh.offline_submit_samples_dev(ptr, s.shape[1], B)
        assert_tokens_match(hip_zh.offline_wait(ta), want, mg, what="configs[4] shard full size, pipelined slot 0")
        assert_tokens_match(hip_zh.offline_wait(tb), want, mg, what="configs[4] shard full size, pipelined slot 1")
    finally:
        hip_zh.device_free(ptr)
    enc_h = hip_zh.encoder_proj(x)
    dec = oracle_zh.decoder(np.array([[-1, 0]], np.int64))
    for b in (0, 5):
        np.testing.assert_allclose(enc_h[b], enc_o[b], atol=5e-4, rtol=0)
        lo = oracle_zh.joiner(enc_o[b], np.repeat(dec, enc_o.shape[1], 0))
        lh = hip_zh.joiner_proj(enc_h[b], np.repeat(dec, enc_o.shape[1], 0))
        assert float(np.abs(lo - lh).max()) < LOGIT_TOL


# ---------------------------------------------------------------- streaming (OnlineProjOfConformer)
def test_streaming_conformer_matches_oracle(tmp_path_factory):
    """chunk_forward on the GPU against the oracle: tokens, timestamps, hyp, every cache, and the reference's processed_lens
    behaviour (2 at creation, then the batch size of the last step, OnlineProjOfConformer.cs:77,229), which decides how much of
    the left context is visible."""
    from k2transducerasr_amd import OnlineRecognizer
    from k2transducerasr_amd.synth import synth_utterance, write_synthetic_model
    from oracle.online import OnlineOracle
    p = str(tmp_path_factory.mktemp("csg") / "conformer_stream.k2w")
    write_synthetic_model(p, "conformer-streaming-tiny-test")
    oo = OnlineOracle(p)
    rec = OnlineRecognizer(p)
    assert (rec.chunk_length, rec.shift_length, rec.frames_per_chunk) == (43, 32, 8)
    utts = [synth_utterance(60 + u, 2.2) for u in range(3)]
    feats = [oo.fbank(u) for u in utts]
    hs = [rec.create_online_stream() for _ in utts]
    os_ = [oo.create_stream() for _ in utts]
    assert [h.processed_len for h in hs] == [2, 2, 2]
    for h, f in zip(hs, feats):
        h.add_features(f)
    pos, n = 0, 0
    while pos + 43 <= feats[0].shape[0]:
        if n == 3:   # one step with a single stream: processed_lens becomes 1 for it
            oo.step(os_[:1], [feats[0][pos : pos + 43]])
            rec.get_results(hs[:1])
            oo.step(os_[1:], [f[pos : pos + 43] for f in feats[1:]])
            rec.get_results(hs[1:])
        else:
            oo.step(os_, [f[pos : pos + 43] for f in feats])
            rec.get_results(hs)
        assert [h.processed_len for h in hs] == [o.processed_len for o in os_]
        pos += 32
        n += 1
    assert n >= 5 and sum(len(o.tokens) - 2 for o in os_) > 0
    for h, o in zip(hs, os_):
        assert h.tokens == o.tokens and h.timestamps == o.timestamps and h.hyp == o.hyp
        for l in range(2):
            np.testing.assert_allclose(h.state(l, "conf_attn"), o.lstm_state(l, "h"), atol=ACT_TOL, rtol=0)
            np.testing.assert_allclose(h.state(l, "conf_conv"), o.lstm_state(l, "c"), atol=ACT_TOL, rtol=0)


def test_conformer_shortest_inputs(hip_conformer, oracle_conformer):
    rng = np.random.default_rng(9)
    for T in (7, 8, 9, 11, 20):     # T' = 1, 1, 1, 2, 4
        x = rng.standard_normal((3, T, 80)).astype(np.float32)
        np.testing.assert_allclose(hip_conformer.encoder_proj(x), oracle_conformer.encoder(x), atol=ACT_TOL, rtol=0)
    from k2transducerasr_amd import K2HipError
    with pytest.raises(K2HipError):
        hip_conformer.encoder_proj(np.zeros((1, 6, 80), np.float32))


def test_fused_and_gemm_attention_scores_agree(tmp_path_factory, utts):
    """K2HIP_CONFORMER_GEMM_SCORES selects the two-GEMM + gather form of the attention scores; both forms against the oracle and
    against each other, for lengths that are / are not multiples of the 16-row strips (40 frames: one wave's run holds the
    strip's first AND last key tile; 263: runs of 5 / 5 / 5 / 2 tiles, the last one partial)."""
    from k2transducerasr_amd import Model, set_switch
    from k2transducerasr_amd.synth import write_synthetic_model
    from oracle import Oracle
    from parity import ACT_TOL
    p = str(tmp_path_factory.mktemp("confs") / "conf.k2w")
    write_synthetic_model(p, "conformer-tiny-test")
    hip, ora = Model(p, 0), Oracle(p)
    rng = np.random.default_rng(5)
    for T in (40, 135, 263):
        x = rng.standard_normal((3, T, 80)).astype(np.float32)
        want = ora.encoder(x)
        fused = hip.encoder_proj(x)
        set_switch("K2HIP_CONFORMER_GEMM_SCORES", 1)
        try:
            gemm = hip.encoder_proj(x)
        finally:
            set_switch("K2HIP_CONFORMER_GEMM_SCORES", 0)
        np.testing.assert_allclose(fused, want, atol=ACT_TOL, rtol=0, err_msg=f"fused T={T}")
        np.testing.assert_allclose(gemm, want, atol=ACT_TOL, rtol=0, err_msg=f"gemm T={T}")


def test_streaming_conformer_with_right_context_matches_oracle(tmp_path):
    """right_context = 2 (OnlineModel.cs:161-165 reads the key; round 3 refused such a model): 51-frame chunks, 8 output frames, a
    shift of 32 -- the two look-ahead encoder frames are seen by this step's attention and convolution, stay out of both caches
    (states = key[-(left + R) : -R]) and are cut from the output.  Tokens, timestamps, Hyp and every cache against the oracle chunk
    by chunk (whose chunk function is held to an independent torch twin on the CPU, tests/test_oracle_conformer.py), including a step
    with one stream alone (the processed_lens quirk)."""
    from k2transducerasr_amd import OnlineRecognizer
    from k2transducerasr_amd.synth import synth_utterance, write_synthetic_model
    from oracle.online import OnlineOracle
    p = str(tmp_path / "conformer_stream_rc.k2w")
    write_synthetic_model(p, "conformer-streaming-rc-tiny-test")
    oo = OnlineOracle(p)
    rec = OnlineRecognizer(p)
    assert (rec.chunk_length, rec.shift_length, rec.frames_per_chunk) == (51, 32, 8)
    utts = [synth_utterance(64 + u, 2.4) for u in range(3)]
    feats = [oo.fbank(u) for u in utts]
    hs = [rec.create_online_stream() for _ in utts]
    os_ = [oo.create_stream() for _ in utts]
    for h, f in zip(hs, feats):
        h.add_features(f)
    pos, n = 0, 0
    while pos + 51 <= feats[0].shape[0]:
        if n == 2:
            oo.step(os_[:1], [feats[0][pos: pos + 51]])
            rec.get_results(hs[:1])
            oo.step(os_[1:], [f[pos: pos + 51] for f in feats[1:]])
            rec.get_results(hs[1:])
        else:
            oo.step(os_, [f[pos: pos + 51] for f in feats])
            rec.get_results(hs)
        for h, o in zip(hs, os_):
            assert h.tokens == o.tokens and h.timestamps == o.timestamps and h.processed_len == o.processed_len, n
            for l in range(2):
                np.testing.assert_allclose(h.state(l, "conf_attn"), o.lstm_state(l, "h"), atol=2e-4, rtol=0, err_msg=f"chunk {n} layer {l} cached_attn")
                np.testing.assert_allclose(h.state(l, "conf_conv"), o.lstm_state(l, "c"), atol=2e-4, rtol=0, err_msg=f"chunk {n} layer {l} cached_conv")
        pos += 32
        n += 1
    assert n >= 4 and sum(len(h.tokens) for h in hs) > 6
