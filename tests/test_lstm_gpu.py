"""GPU parity for the LSTM transducer (Model_type "lstm": offline via OfflineProjOfTransducer, streaming via OnlineProjOfLstm,
SURVEY 8f N4): libk2hip.so through the C ABI against oracle/k2_oracle_lstm.c."""
import numpy as np
import pytest

from parity import ACT_TOL, assert_tokens_match

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lstm_path(tmp_path_factory):
    from k2transducerasr_amd.synth import write_synthetic_model
    p = str(tmp_path_factory.mktemp("lstmg") / "lstm_tiny.k2w")
    write_synthetic_model(p, "lstm-tiny-test")
    return p


@pytest.fixture(scope="module")
def hip_lstm(lstm_path):
    from k2transducerasr_amd import Model
    return Model(lstm_path, 0)


@pytest.fixture(scope="module")
def oracle_lstm(lstm_path):
    from oracle.online import OnlineOracle
    return OnlineOracle(lstm_path)


@pytest.fixture(scope="module")
def feats(oracle_lstm, utts):
    return [oracle_lstm.fbank(u) for u in utts]


def test_lstm_geometry(hip_lstm, oracle_lstm):
    for T in (9, 12, 13, 17, 100, 1017):
        assert hip_lstm.encoder_out_frames(T) == oracle_lstm.encoder_out_frames(T)


@pytest.mark.parametrize("tap", [0, 1, 3])
def test_lstm_taps(hip_lstm, oracle_lstm, feats, tap):
    x = oracle_lstm.pad_sequence(feats[:3]).reshape(3, -1, 80)
    np.testing.assert_allclose(hip_lstm.encoder_tap(x, tap), oracle_lstm.encoder_tap(x, tap), atol=ACT_TOL, rtol=0)


def test_lstm_offline_end_to_end(hip_lstm, oracle_lstm, feats, utts):
    x = oracle_lstm.pad_sequence(feats).reshape(len(feats), -1, 80)
    enc = oracle_lstm.encoder(x)
    np.testing.assert_allclose(hip_lstm.encoder_proj(x), enc, atol=ACT_TOL, rtol=0)
    want, mg = oracle_lstm.greedy_batch(enc, want_margins=True)
    assert sum(len(w[0]) for w in want) > 0
    assert_tokens_match(hip_lstm.offline_greedy_from_samples(utts), want, mg, what="lstm offline")


def test_lstm_streaming_matches_oracle(lstm_path, oracle_lstm):
    from k2transducerasr_amd import OnlineRecognizer
    from k2transducerasr_amd.synth import synth_utterance
    rec = OnlineRecognizer(lstm_path)
    utts = [synth_utterance(40 + u, 1.5) for u in range(3)]
    hs = [rec.create_online_stream() for _ in utts]
    os_ = [oracle_lstm.create_stream() for _ in utts]
    feats = [oracle_lstm.fbank(u) for u in utts]
    assert (rec.chunk_length, rec.shift_length) == (9, 4)
    for h, f in zip(hs, feats):
        h.add_features(f)
    pos = 0
    while pos + 9 <= feats[0].shape[0]:
        oracle_lstm.step(os_, [f[pos : pos + 9] for f in feats])
        rec.get_results(hs)
        pos += 4
    assert sum(len(o.tokens) - 2 for o in os_) > 0
    for h, o in zip(hs, os_):
        assert h.tokens == o.tokens and h.timestamps == o.timestamps and h.hyp == o.hyp
        for l in range(3):
            np.testing.assert_allclose(h.state(l, "lstm_h"), o.lstm_state(l, "h"), atol=ACT_TOL, rtol=0)
            np.testing.assert_allclose(h.state(l, "lstm_c"), o.lstm_state(l, "c"), atol=ACT_TOL, rtol=0)


def test_lstm_shortest_inputs(hip_lstm, oracle_lstm):
    rng = np.random.default_rng(9)
    for T in (9, 12, 13, 20):
        x = rng.standard_normal((2, T, 80)).astype(np.float32)
        np.testing.assert_allclose(hip_lstm.encoder_proj(x), oracle_lstm.encoder(x), atol=ACT_TOL, rtol=0)
    from k2transducerasr_amd import K2HipError
    with pytest.raises(K2HipError):
        hip_lstm.encoder_proj(np.zeros((1, 8, 80), np.float32))


def test_lstm_wavefront_split_k_and_sequential_fallback(tmp_path_factory, utts):
    """A model wide enough for the split-K products of the layer wavefront (projection and feed_forward.4 summed from four
    partials) and deeper than the utterances are short: encoder output and tokens against the oracle, and against the
    layer-by-layer fallback path (K2HIP_LSTM_SEQ) of the same library."""
    import os
    from k2transducerasr_amd import Model
    from k2transducerasr_amd.synth import write_synthetic_model
    from oracle.online import OnlineOracle
    p = str(tmp_path_factory.mktemp("lstms") / "lstm_split.k2w")
    write_synthetic_model(p, "lstm-tiny-split-test")
    hip, ora = Model(p, 0), OnlineOracle(p)
    feats = [ora.fbank(u) for u in utts]
    x = ora.pad_sequence(feats).reshape(len(feats), -1, 80)
    enc = ora.encoder(x)
    got = hip.encoder_proj(x)
    np.testing.assert_allclose(got, enc, atol=ACT_TOL, rtol=0)
    want, mg = ora.greedy_batch(enc, want_margins=True)
    assert_tokens_match(hip.offline_greedy_from_samples(utts), want, mg, what="lstm split-K")
    for T in (9, 13, 20):   # fewer frames than layers: the wavefront never has all layers active
        xs = np.random.default_rng(T).standard_normal((2, T, 80)).astype(np.float32)
        np.testing.assert_allclose(hip.encoder_proj(xs), ora.encoder(xs), atol=ACT_TOL, rtol=0)
    from k2transducerasr_amd import set_switch
    set_switch("K2HIP_LSTM_SEQ", 1)
    try:
        np.testing.assert_allclose(hip.encoder_proj(x), got, atol=ACT_TOL, rtol=0)
    finally:
        set_switch("K2HIP_LSTM_SEQ", 0)
