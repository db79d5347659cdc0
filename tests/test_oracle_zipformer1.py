"""CPU tests of the streaming Zipformer (v1) oracle (oracle/k2_oracle_zipformer1.c; OnlineProjOfZipformer): against
the independent torch twin -- encoder output AND every cached state, chunk after chunk -- plus the reference's state inventory."""
import numpy as np
import pytest

LOG_FLOOR = np.float32(-23.025850929940457)


@pytest.fixture(scope="module")
def z1_model_path(tmp_path_factory):
    from k2transducerasr_amd.synth import write_synthetic_model
    p = str(tmp_path_factory.mktemp("z1models") / "z1tiny.k2w")
    write_synthetic_model(p, "zipformer-streaming-tiny-test")
    return p


@pytest.fixture(scope="module")
def z1_oracle(z1_model_path):
    from oracle.online import OnlineOracle
    return OnlineOracle(z1_model_path)


def test_chunk_geometry(z1_oracle):
    # _chunkLength = T = 39, _shiftLength = decode_chunk_len = 32 (OnlineProjOfZipformer.cs:24-25,39-40); 16 frames at 50 Hz -> 8 out
    assert (z1_oracle.chunk_length, z1_oracle.shift_length, z1_oracle.frames_per_chunk) == (39, 32, 8)


def test_init_states_match_reference_shapes(z1_oracle):
    # GetEncoderInitStates (OnlineProjOfZipformer.cs:56-111) at batchSize 1, per layer
    s = z1_oracle.create_stream()
    dims, atts, layers, kern, left = [64, 64, 96, 96], [32, 32, 96, 96], [1, 2, 1, 1], [7, 7, 5, 7], [32, 16, 8, 16]
    l = 0
    for si in range(4):
        for _ in range(layers[si]):
            assert s.state(l, "len").size == 1 and s.state(l, "len")[0] == 0
            assert s.state(l, "avg").size == dims[si]
            assert s.state(l, "key").size == left[si] * atts[si]
            assert s.state(l, "val").size == s.state(l, "val2").size == left[si] * atts[si] // 2
            assert s.state(l, "conv1").size == s.state(l, "conv2").size == dims[si] * (kern[si] - 1)
            assert not s.state(l, "key").any() and not s.state(l, "conv2").any()
            l += 1
    assert s.num_layers == 5 and s.hyp == [0, 0] and s.tokens == [0, 0]


def test_streaming_encoder_and_states_match_torch_twin(z1_oracle, z1_model_path):
    import torch
    from k2transducerasr_amd.k2w import read_k2w
    from k2transducerasr_amd.synth import synth_utterance
    from torch_twin_zipformer1 import Zipformer1Twin
    torch.set_num_threads(4)
    meta, tensors = read_k2w(z1_model_path)
    tw = Zipformer1Twin(meta, tensors)
    feats = z1_oracle.fbank(synth_utterance(5, 2.4))
    s = z1_oracle.create_stream()
    st = tw.init_states()
    T, S = z1_oracle.chunk_length, z1_oracle.shift_length
    pos = n = 0
    with torch.no_grad():
        while pos + T <= feats.shape[0]:
            x = feats[pos : pos + T].copy()
            x[x == 0] = LOG_FLOOR
            got = z1_oracle.encoder_chunk(s, x)
            want = tw.chunk(torch.from_numpy(x), st).numpy()
            assert got.shape == want.shape == (8, 512)
            np.testing.assert_allclose(got, want, rtol=2e-4, atol=2e-4, err_msg=f"chunk {n}")
            l = 0
            for si in range(4):
                for li in range(tw.layers[si]):
                    for kind, twk in (("len", "len"), ("avg", "avg"), ("key", "key"), ("val", "val"), ("val2", "val2"), ("conv1", "conv1"),
                                      ("conv2", "conv2")):
                        a, b = s.state(l, kind), st[si][twk][li].numpy().reshape(-1)
                        np.testing.assert_allclose(a, b, rtol=2e-4, atol=2e-4, err_msg=f"chunk {n} layer {l} {kind}")
                    l += 1
            pos += S
            n += 1
    assert n >= 6  # the left context of every stack (<= 32 frames at 50 Hz) has wrapped at least once


def test_online_step_matches_per_stream_chunks(z1_oracle):
    """ForwardBatchGreedySearch over two streams == each stream alone (states are per stream), tokens skip {0,2,1} (:181)."""
    from k2transducerasr_amd.synth import synth_utterance
    T, S = z1_oracle.chunk_length, z1_oracle.shift_length
    fa, fb = z1_oracle.fbank(synth_utterance(1, 1.6)), z1_oracle.fbank(synth_utterance(2, 1.6))
    sa, sb, ta, tb = (z1_oracle.create_stream() for _ in range(4))
    pos = 0
    while pos + T <= fa.shape[0]:
        z1_oracle.step([sa, sb], [fa[pos : pos + T], fb[pos : pos + T]])
        z1_oracle.step([ta], [fa[pos : pos + T]])
        z1_oracle.step([tb], [fb[pos : pos + T]])
        pos += S
    assert sa.tokens == ta.tokens and sb.tokens == tb.tokens and sa.timestamps == ta.timestamps
    assert len(sa.tokens) > 2 and all(t not in (0, 1, 2) for t in sa.tokens[2:])


def test_offline_encoder_matches_torch_twin(tmp_path_factory):
    """The offline Zipformer v1 graph (Model_type "zipformer" at OfflineRecognizer.cs:40): oracle against the independent torch
    restatement -- the embed output, every stack's output and the projected encoder output, for odd and even lengths."""
    import torch
    from k2transducerasr_amd.k2w import read_k2w
    from k2transducerasr_amd.synth import synth_utterance, write_synthetic_model
    from oracle import Oracle
    from torch_twin_zipformer1 import Zipformer1Twin
    p = str(tmp_path_factory.mktemp("z1off") / "z1off.k2w")
    write_synthetic_model(p, "zipformer-tiny-test")
    ora = Oracle(p)
    meta, tensors = read_k2w(p)
    tw = Zipformer1Twin(meta, tensors)
    torch.set_num_threads(4)
    for secs in (0.83, 1.21):
        feats = [ora.fbank(synth_utterance(7 + u, secs)) for u in range(2)]
        x = ora.pad_sequence(feats).reshape(2, -1, 80)
        assert ora.encoder_out_frames(x.shape[1]) == ((x.shape[1] - 7) // 2 + 1) // 2
        with torch.no_grad():
            for tap in (0, 1, 2, 3, 4):
                want = tw.forward_offline(torch.from_numpy(x), tap).numpy()
                np.testing.assert_allclose(ora.encoder_tap(x, tap).reshape(want.shape), want, rtol=2e-4, atol=2e-4, err_msg=f"tap {tap}")
            want = tw.forward_offline(torch.from_numpy(x)).numpy()
        np.testing.assert_allclose(ora.encoder(x), want, rtol=2e-4, atol=2e-4)
