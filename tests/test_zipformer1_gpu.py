"""GPU parity of the streaming Zipformer v1 path (OnlineProjOfZipformer replacement): libk2hip.so against the CPU
oracle, chunk after chunk: tokens, timestamps, Hyp and every cached state (cached_len / avg / key / val / val2 / conv1 / conv2)
in the stream's device slot."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

KINDS = ["len", "avg", "key", "val", "val2", "conv1", "conv2"]


@pytest.fixture(scope="module")
def z1_path(tmp_path_factory):
    from k2transducerasr_amd.synth import write_synthetic_model
    p = str(tmp_path_factory.mktemp("z1models") / "z1tiny.k2w")
    write_synthetic_model(p, "zipformer-streaming-tiny-test")
    return p


@pytest.fixture(scope="module")
def rec(z1_path):
    from k2transducerasr_amd import OnlineRecognizer
    return OnlineRecognizer(z1_path)


@pytest.fixture(scope="module")
def ora(z1_path):
    from oracle.online import OnlineOracle
    return OnlineOracle(z1_path)


def test_chunk_info_and_init(rec, ora):
    # _chunkLength = T = 39, _shiftLength = decode_chunk_len = 32 (OnlineProjOfZipformer.cs:24-25,39-40)
    assert (rec.chunk_length, rec.shift_length, rec.frames_per_chunk) == (ora.chunk_length, ora.shift_length, ora.frames_per_chunk) == (39, 32, 8)
    s = rec.create_online_stream()
    o = ora.create_stream()
    assert s.tokens == [0, 0] and s.hyp == [0, 0] and s.timestamps == []
    for l in range(o.num_layers):
        for k in KINDS:
            a = s.state(l, k)
            assert a.size == o.state(l, k).size and not a.any()   # GetEncoderInitStates (:56-111): zeros


def test_streaming_matches_oracle_chunk_by_chunk(rec, ora):
    from k2transducerasr_amd.synth import synth_utterance
    B = 3
    feats = [ora.fbank(synth_utterance(70 + u, d)) for u, d in enumerate([2.4, 1.5, 2.4])]
    hs = [rec.create_online_stream() for _ in range(B)]
    os_ = [ora.create_stream() for _ in range(B)]
    T, S = rec.chunk_length, rec.shift_length
    pos = [0] * B
    for h, f in zip(hs, feats):
        h.add_features(f)
    steps = 0
    while True:
        ready = [b for b in range(B) if pos[b] + T <= feats[b].shape[0]]
        dec, n_new = rec.get_results(hs)
        assert [b for b in range(B) if dec[b]] == ready
        if not ready:
            break
        want_new = ora.step([os_[b] for b in ready], [feats[b][pos[b] : pos[b] + T] for b in ready])
        for b, wn in zip(ready, want_new):
            assert n_new[b] == wn
            pos[b] += S
        for b in range(B):
            assert hs[b].tokens == os_[b].tokens, (steps, b)
            assert hs[b].timestamps == os_[b].timestamps
            assert hs[b].hyp == os_[b].hyp
        if steps in (0, 1, 4):
            for b in ready:
                for l in range(os_[b].num_layers):
                    for k in KINDS:
                        np.testing.assert_allclose(hs[b].state(l, k), os_[b].state(l, k), atol=2e-4, rtol=0, err_msg=f"step {steps} stream {b} layer {l} {k}")
        steps += 1
    assert steps >= 6
    assert sum(len(o.tokens) - 2 for o in os_) > 0


def test_slots_are_recycled(rec):
    a = rec.create_online_stream()
    a.add_features(np.random.default_rng(0).standard_normal((39, 80)).astype(np.float32))
    rec.get_results([a])
    assert a.state(0, "key").any() and a.state(0, "len")[0] == 16
    a.close()
    b = rec.create_online_stream()
    assert not b.state(0, "key").any() and not b.state(0, "avg").any() and b.state(0, "len")[0] == 0


def test_streaming_en_model_matches_oracle(tmp_path_factory):
    """The published recipe's architecture (5 stacks of 384, 15 layers, left context 64 at 50 Hz), random weights, two streams, a
    few chunks: tokens exact, states within float tolerance."""
    from k2transducerasr_amd import OnlineRecognizer
    from k2transducerasr_amd.synth import synth_utterance, write_synthetic_model
    from oracle.online import OnlineOracle
    p = str(tmp_path_factory.mktemp("z1en") / "z1en.k2w")
    write_synthetic_model(p, "zipformer-streaming-en")
    rec, ora = OnlineRecognizer(p), OnlineOracle(p)
    feats = [ora.fbank(synth_utterance(80 + u, 1.7)) for u in range(2)]
    hs = [rec.create_online_stream() for _ in feats]
    os_ = [ora.create_stream() for _ in feats]
    for h, f in zip(hs, feats):
        h.add_features(f)
    T, S = rec.chunk_length, rec.shift_length
    for k in range((feats[0].shape[0] - T) // S + 1):
        dec, n_new = rec.get_results(hs)
        assert dec == [1, 1]
        want = ora.step(os_, [f[k * S : k * S + T] for f in feats])
        assert n_new == want
        for h, o in zip(hs, os_):
            assert h.tokens == o.tokens and h.timestamps == o.timestamps
    for h, o in zip(hs, os_):
        for l in (0, 5, 14):
            for kind in KINDS:
                np.testing.assert_allclose(h.state(l, kind), o.state(l, kind), atol=5e-4, rtol=0)


# ---- offline graph: Model_type "zipformer" in OfflineRecognizer's switch (OfflineRecognizer.cs:40-44) ----
@pytest.fixture(scope="module")
def z1off(tmp_path_factory):
    from k2transducerasr_amd import Model
    from k2transducerasr_amd.synth import write_synthetic_model
    from oracle import Oracle
    p = str(tmp_path_factory.mktemp("z1off") / "z1off.k2w")
    write_synthetic_model(p, "zipformer-tiny-test")
    return Model(p, 0), Oracle(p)


@pytest.mark.parametrize("tap", [0, 1, 2, 3, 4])
def test_offline_taps(z1off, utts, tap):
    from parity import ACT_TOL
    hip, ora = z1off
    x = ora.pad_sequence([ora.fbank(u) for u in utts[:3]]).reshape(3, -1, 80)
    np.testing.assert_allclose(hip.encoder_tap(x, tap), ora.encoder_tap(x, tap), atol=ACT_TOL, rtol=0)


def test_offline_end_to_end_and_lengths(z1off, utts):
    from parity import ACT_TOL, assert_tokens_match
    hip, ora = z1off
    x = ora.pad_sequence([ora.fbank(u) for u in utts]).reshape(len(utts), -1, 80)
    enc = ora.encoder(x)
    np.testing.assert_allclose(hip.encoder_proj(x), enc, atol=ACT_TOL, rtol=0)
    want, mg = ora.greedy_batch(enc, want_margins=True)
    assert sum(len(w[0]) for w in want) > 0
    assert_tokens_match(hip.offline_greedy_from_samples(utts), want, mg, what="zipformer v1 offline")
    rng = np.random.default_rng(11)
    for T in (9, 10, 23, 38, 61, 100):   # frame counts that do / do not divide by the downsampling factors
        xs = rng.standard_normal((2, T, 80)).astype(np.float32)
        assert hip.encoder_out_frames(T) == ora.encoder_out_frames(T)
        np.testing.assert_allclose(hip.encoder_proj(xs), ora.encoder(xs), atol=ACT_TOL, rtol=0, err_msg=f"T={T}")


def test_offline_en_architecture(tmp_path_factory):
    """The published offline recipe's architecture (5 x 384, 15 layers, head size 24), random weights, 2 x 3 s: encoder output
    against the oracle and exact tokens."""
    from k2transducerasr_amd import Model
    from k2transducerasr_amd.synth import synth_utterance, write_synthetic_model
    from oracle import Oracle
    from parity import ACT_TOL, assert_tokens_match
    p = str(tmp_path_factory.mktemp("z1en") / "z1en_off.k2w")
    write_synthetic_model(p, "zipformer-en")
    hip, ora = Model(p, 0), Oracle(p)
    us = [synth_utterance(120 + u, 3.0) for u in range(2)]
    x = ora.pad_sequence([ora.fbank(u) for u in us]).reshape(2, -1, 80)
    enc = ora.encoder(x)
    np.testing.assert_allclose(hip.encoder_proj(x), enc, atol=5 * ACT_TOL, rtol=0)
    want, mg = ora.greedy_batch(enc, want_margins=True)
    assert_tokens_match(hip.offline_greedy_from_samples(us), want, mg, what="zipformer-en offline")
