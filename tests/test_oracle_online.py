"""CPU tests of the streaming oracle (k2_oracle_online.c): against the independent torch twin
(encoder output AND every cached state, chunk after chunk) and known answers for the online greedy
loop (OnlineRecognizer.cs:85-219)."""
import numpy as np
import pytest

LOG_FLOOR = np.float32(-23.025850929940457)


@pytest.fixture(scope="module")
def stream_model_path(tmp_path_factory):
    from k2transducerasr_amd.synth import write_synthetic_model
    p = str(tmp_path_factory.mktemp("smodels") / "stiny.k2w")
    write_synthetic_model(p, "zipformer2-streaming-tiny-test")
    return p


@pytest.fixture(scope="module")
def online_oracle(stream_model_path):
    from oracle.online import OnlineOracle
    return OnlineOracle(stream_model_path)


def test_chunk_geometry(online_oracle):
    # ChunkLength = T = 2*16+13, ShiftLength = decode_chunk_len = 32 (OnlineModel.cs:48-49), 8 frames out
    assert (online_oracle.chunk_length, online_oracle.shift_length, online_oracle.frames_per_chunk) == (45, 32, 8)


def test_init_states_match_reference_shapes(online_oracle):
    # GetEncoderInitStates (OnlineProjOfZipformer2.cs:63-111): sizes per layer, all zero
    s = online_oracle.create_stream()
    dims, layers, heads, kern, left = [64, 96, 128, 64], [1, 2, 1, 1], [2, 2, 4, 2], [15, 7, 7, 15], [32, 16, 8, 16]
    l = 0
    for si in range(4):
        for _ in range(layers[si]):
            assert s.state(l, "key").size == left[si] * 32 * heads[si]
            assert s.state(l, "nonlin").size == left[si] * (3 * dims[si] // 4)
            assert s.state(l, "val1").size == s.state(l, "val2").size == left[si] * 12 * heads[si]
            assert s.state(l, "conv1").size == s.state(l, "conv2").size == dims[si] * (kern[si] // 2)
            assert not s.state(l, "key").any()
            l += 1
    assert s.state(0, "embed").size == 128 * 3 * 19   # embed_states [B,128,3,19] (:60)
    assert s.processed_len == 0 and s.hyp == [0, 0] and s.tokens == [0, 0]  # OnlineStream.cs:44-45


def test_streaming_encoder_and_states_match_torch_twin(online_oracle, stream_model_path):
    import torch
    from k2transducerasr_amd.k2w import read_k2w
    from k2transducerasr_amd.synth import synth_utterance
    from torch_twin_online import OnlineTwin
    torch.set_num_threads(4)
    meta, tensors = read_k2w(stream_model_path)
    tw = OnlineTwin(meta, tensors)
    feats = online_oracle.fbank(synth_utterance(3, 2.0))
    s = online_oracle.create_stream()
    st = tw.init_states(1)
    T, S = online_oracle.chunk_length, online_oracle.shift_length
    pos = n = 0
    while pos + T <= feats.shape[0]:
        x = feats[pos : pos + T].copy()
        a = online_oracle.encoder_chunk(s, x)
        with torch.no_grad():
            b, st = tw.encoder_chunk(torch.from_numpy(x[None]), st)
        np.testing.assert_allclose(a, b[0].numpy(), atol=5e-5, rtol=0)
        for l in range(s.num_layers):
            for j, kind in enumerate(["key", "nonlin", "val1", "val2", "conv1", "conv2"]):
                ref = st[l * 6 + j].numpy()
                ref = ref[:, 0, :] if kind in ("key", "val1", "val2") else (ref[0, 0] if kind == "nonlin" else ref[0])
                np.testing.assert_allclose(s.state(l, kind), ref.reshape(-1), atol=5e-5, rtol=0)
        np.testing.assert_allclose(s.state(0, "embed"), st[-2].numpy().reshape(-1), atol=5e-5, rtol=0)
        assert s.processed_len == int(st[-1][0]) == 16 * (n + 1)
        pos += S
        n += 1
    assert n >= 4


def test_streams_are_independent(online_oracle):
    """Each stream owns its state (OnlineStream.cs:14,26): stepping two streams together equals stepping
    them one at a time (the reference only holds this at B = 1 because of Q11; that is what we pin)."""
    from k2transducerasr_amd.synth import synth_utterance
    fa, fb = online_oracle.fbank(synth_utterance(5, 1.5)), online_oracle.fbank(synth_utterance(6, 1.5))
    T, S = online_oracle.chunk_length, online_oracle.shift_length
    s1, s2, t1, t2 = (online_oracle.create_stream() for _ in range(4))
    for k in range(3):
        ca, cb = fa[k * S : k * S + T], fb[k * S : k * S + T]
        online_oracle.step([s1, s2], [ca, cb])
        online_oracle.step([t1], [ca])
        online_oracle.step([t2], [cb])
    assert s1.tokens == t1.tokens and s2.tokens == t2.tokens
    assert s1.timestamps == t1.timestamps and s2.timestamps == t2.timestamps
    np.testing.assert_array_equal(s2.state(1, "nonlin"), t2.state(1, "nonlin"))


def test_online_step_bookkeeping(online_oracle):
    from k2transducerasr_amd.synth import synth_utterance
    f = online_oracle.fbank(synth_utterance(7, 3.0))
    T, S = online_oracle.chunk_length, online_oracle.shift_length
    s = online_oracle.create_stream()
    total = 0
    for k in range((f.shape[0] - T) // S + 1):
        n_before = len(s.tokens)
        (n_new,) = online_oracle.step([s], [f[k * S : k * S + T]])
        assert len(s.tokens) == n_before + n_new
        assert s.hyp == s.tokens[-2:]                      # OnlineRecognizer.cs:208
        total += n_new
    assert total > 0, "test signal should emit something"
    assert len(s.timestamps) == total and all(0 <= t < 8 for t in s.timestamps)  # chunk-relative (:184)
    assert all(t not in (0, 1, 2) for t in s.tokens[2:])   # emit filter skips blank, unk AND id 1 (:181)


def test_online_zero_features_are_floored(online_oracle):
    # online PadSequence (tail 0) still maps 0.0 -> log floor (PadHelper.cs:9-13,58)
    T = online_oracle.chunk_length
    a, b = online_oracle.create_stream(), online_oracle.create_stream()
    online_oracle.step([a], [np.zeros((T, 80), np.float32)])
    online_oracle.step([b], [np.full((T, 80), LOG_FLOOR, np.float32)])
    np.testing.assert_array_equal(a.state(0, "key"), b.state(0, "key"))
