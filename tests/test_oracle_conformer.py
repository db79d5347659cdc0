"""CPU tests for the offline Conformer restatement (BASELINE.json configs[4]):
oracle/k2_oracle_conformer.c against the independent torch twin (tests/torch_twin_conformer.py),
tap by tap, plus shape / frame-count known answers."""
import numpy as np
import pytest

from k2transducerasr_amd.k2w import read_k2w


@pytest.fixture(scope="module")
def twin(conformer_tiny_path):
    from torch_twin_conformer import ConformerTwin
    meta, tensors = read_k2w(conformer_tiny_path)
    return ConformerTwin(meta, tensors)


@pytest.fixture(scope="module")
def feats(oracle_conformer, utts):
    f = [oracle_conformer.fbank(u) for u in utts[:3]]
    return oracle_conformer.pad_sequence(f).reshape(3, -1, 80)


def test_conformer_out_frames(oracle_conformer):
    # T' = ((T-1)//2 - 1)//2 (Conv2dSubsampling); C5: 2998 + 19 = 3017 -> 753
    for T, want in [(3017, 753), (1017, 253), (9, 1), (8, 1), (7, 1), (6, 0), (100, 24), (101, 24), (103, 25)]:
        assert oracle_conformer.encoder_out_frames(T) == want


@pytest.mark.parametrize("tap", [0, 1, 2, -1])
def test_conformer_oracle_matches_twin(oracle_conformer, twin, feats, tap):
    B, T, _ = feats.shape
    Tp = oracle_conformer.encoder_out_frames(T)
    if tap < 0:
        got = oracle_conformer.encoder(feats)
    else:
        got = oracle_conformer.encoder_tap(feats, tap).reshape(B, Tp, -1)
    want = twin.forward(feats, tap)
    assert got.shape == want.shape
    scale = float(np.abs(want).max())
    assert scale > 1e-2  # the activations carry signal
    np.testing.assert_allclose(got, want, atol=2e-5 * max(1.0, scale), rtol=0)


def test_conformer_batch_rows_are_independent(oracle_conformer, feats):
    # x_lens = T for every row and no masks (OfflineProjOfTransducer.cs:66-70): row b of a batch equals
    # the same utterance run alone at the same padded length
    full = oracle_conformer.encoder(feats)
    one = oracle_conformer.encoder(feats[1:2])
    np.testing.assert_allclose(full[1:2], one, atol=1e-6, rtol=0)


def test_conformer_decoder_groups_one(oracle_conformer, twin):
    # stateless2 decoder: Conv1d(groups=1); id < 0 -> zero embedding (OfflineRecognizer.cs:105)
    y = np.array([[-1, 0], [0, 0], [3, 40], [17, 5], [-1, -1]], np.int64)
    got = oracle_conformer.decoder(y)
    want = twin.decoder(y)
    np.testing.assert_allclose(got, want, atol=2e-6, rtol=0)


def test_conformer_greedy_emits(oracle_conformer, feats):
    res = oracle_conformer.recognize_batch([f.reshape(-1)[: 80 * 100] for f in feats])
    n = [len(t) for t, _ in res]
    assert sum(n) > 0 and max(n) < 30


# ------------------------------------------------------------------------------------------------ streaming (OnlineProjOfConformer)
@pytest.fixture(scope="module")
def cstream_path(tmp_path_factory):
    from k2transducerasr_amd.synth import write_synthetic_model
    p = str(tmp_path_factory.mktemp("cs") / "conformer_stream.k2w")
    write_synthetic_model(p, "conformer-streaming-tiny-test")
    return p


def test_streaming_conformer_matches_twin(cstream_path, utts):
    import torch
    from oracle.online import OnlineOracle
    from torch_twin_conformer import ConformerStreamTwin
    meta, tensors = read_k2w(cstream_path)
    tw = ConformerStreamTwin(meta, tensors)
    oo = OnlineOracle(cstream_path)
    assert (oo.chunk_length, oo.shift_length, oo.frames_per_chunk) == (43, 32, 8)   # T = (8 + 2) * 4 + 3, decode_chunk_len = 8 * 4
    f = oo.fbank(utts[0])
    s = oo.create_stream()
    assert s.processed_len == 2            # OnlineProjOfConformer.GetEncoderInitStates: processed_lens[0] = 2 (:77)
    st = tw.init_states(1)
    pl = torch.tensor([2])
    pos = n = 0
    while pos + 43 <= f.shape[0]:
        x = f[pos : pos + 43]
        a = oo.encoder_chunk(s, x)
        b, st, pl = tw.chunk(x[None], st, pl)
        np.testing.assert_allclose(a, b[0], atol=5e-5, rtol=0)
        for l in range(2):
            np.testing.assert_allclose(s.lstm_state(l, "h").reshape(16, 64), st[0][l, :, 0].numpy(), atol=5e-5, rtol=0)   # cached_attn
            np.testing.assert_allclose(s.lstm_state(l, "c").reshape(6, 64), st[1][l, :, 0].numpy(), atol=5e-5, rtol=0)    # cached_conv
        assert s.processed_len == int(pl[0]) == 2 + 8 * (n + 1)
        pos += 32
        n += 1
    assert n >= 3


def test_streaming_conformer_with_right_context_matches_twin(tmp_path, utts):
    """right_context = 2 (OnlineModel.cs:161-165 reads the key): T = (8 + 2 + 2) * 4 + 3 = 51 frames per chunk, still 8 output frames
    and a shift of 32; the two extra encoder frames are seen by this step's attention and convolution, kept OUT of both caches and cut
    from the output.  Oracle against the independent torch twin: outputs and every cache, chunk by chunk; and the overlap must
    matter -- the outputs differ from the right_context = 0 model's on the same audio."""
    import torch
    from k2transducerasr_amd.synth import write_synthetic_model
    from oracle.online import OnlineOracle
    from torch_twin_conformer import ConformerStreamTwin
    p = str(tmp_path / "rc.k2w")
    write_synthetic_model(p, "conformer-streaming-rc-tiny-test")
    meta, tensors = read_k2w(p)
    assert meta["right_context"] == "2"
    tw = ConformerStreamTwin(meta, tensors)
    oo = OnlineOracle(p)
    assert (oo.chunk_length, oo.shift_length, oo.frames_per_chunk) == (51, 32, 8)
    f = oo.fbank(utts[0])
    s = oo.create_stream()
    st = tw.init_states(1)
    pl = torch.tensor([2])
    pos = n = 0
    outs = []
    while pos + 51 <= f.shape[0]:
        x = f[pos: pos + 51]
        a = oo.encoder_chunk(s, x)
        b, st, pl = tw.chunk(x[None], st, pl)
        assert a.shape == (8, 512) and b.shape == (1, 8, 512)
        np.testing.assert_allclose(a, b[0], atol=5e-5, rtol=0)
        for l in range(2):
            np.testing.assert_allclose(s.lstm_state(l, "h").reshape(16, 64), st[0][l, :, 0].numpy(), atol=5e-5, rtol=0)   # cached_attn
            np.testing.assert_allclose(s.lstm_state(l, "c").reshape(6, 64), st[1][l, :, 0].numpy(), atol=5e-5, rtol=0)    # cached_conv
        assert s.processed_len == int(pl[0]) == 2 + 8 * (n + 1)
        outs.append(a)
        pos += 32
        n += 1
    assert n >= 3
    # the same weights without the right context see less: different outputs
    p0 = str(tmp_path / "rc0.k2w")
    write_synthetic_model(p0, "conformer-streaming-tiny-test")
    o0 = OnlineOracle(p0)
    s0 = o0.create_stream()
    a0 = o0.encoder_chunk(s0, f[:43])
    assert np.abs(a0 - outs[0]).max() > 1e-3


def test_streaming_conformer_step_reproduces_the_processed_lens_bug(cstream_path, utts):
    """OnlineProjOfConformer.unstack_states (:229) stores the BATCH SIZE as processed_lens instead of the model's output, so
    after a step with B streams only the newest B left-context frames stay unmasked."""
    from oracle.online import OnlineOracle
    oo = OnlineOracle(cstream_path)
    f = [oo.fbank(u) for u in utts[:3]]
    ss = [oo.create_stream() for _ in f]
    oo.step(ss, [x[:43] for x in f])
    assert [s.processed_len for s in ss] == [3, 3, 3]
    oo.step(ss[:1], [f[0][32:75]])
    assert ss[0].processed_len == 1
