"""GPU parity for the CTC path: zipformer2ctc head + ForwardBatchGreedySearchCTC, offline and streaming."""
import ctypes as C

import numpy as np
import pytest

from parity import ACT_TOL

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctc_path(tmp_path_factory):
    from k2transducerasr_amd.synth import write_synthetic_model
    p = str(tmp_path_factory.mktemp("ctcg") / "ctc_tiny.k2w")
    write_synthetic_model(p, "zipformer2-ctc-tiny-test")
    return p


@pytest.fixture(scope="module")
def hip_ctc(ctc_path):
    from k2transducerasr_amd import Model
    return Model(ctc_path, 0)


@pytest.fixture(scope="module")
def oracle_ctc(ctc_path):
    from oracle import Oracle
    return Oracle(ctc_path)


def _lp(rows, V=37):
    x = np.full((1, len(rows), V), -5.0, np.float32)
    for t, r in enumerate(rows):
        if isinstance(r, dict):
            for k, v in r.items():
                x[0, t, k] = v
        else:
            x[0, t, r] = 0.0
    return x


def test_ctc_greedy_known_answers(hip_ctc):
    res, tb = hip_ctc.ctc_greedy(_lp([0, 3, 3, 0, 3, 4, 4, 4, 0, 0]))
    assert res[0] == ([3, 3, 4], [1, 4, 5]) and tb[0] == 2
    res, _ = hip_ctc.ctc_greedy(_lp([2, 1, 1, 2]))                      # unk / sos are not filtered on this path
    assert res[0] == ([2, 1, 2], [0, 1, 3])
    res, _ = hip_ctc.ctc_greedy(_lp([{2: 1.0, 4: 1.0}, {0: 1.0, 5: 1.0}]))  # Array.IndexOf: first maximum wins
    assert res[0] == ([2], [0])
    res, tb = hip_ctc.ctc_greedy(_lp([0, 0]), frame_offsets=[7], num_trailing_blank=[5])
    assert res[0] == ([], []) and tb[0] == 7
    res, tb = hip_ctc.ctc_greedy(_lp([0, 4]), frame_offsets=[7], num_trailing_blank=[5])
    assert res[0] == ([4], [8]) and tb[0] == 0


def test_ctc_log_probs_match_oracle(hip_ctc, oracle_ctc, utts):
    f = [oracle_ctc.fbank(u) for u in utts]
    x = oracle_ctc.pad_sequence(f).reshape(len(utts), -1, 80)
    a = hip_ctc.encoder_proj(x)
    b = oracle_ctc.encoder(x)
    assert a.shape == b.shape and a.shape[2] == 37
    np.testing.assert_allclose(a, b, atol=ACT_TOL, rtol=0)
    want, _ = oracle_ctc.ctc_greedy(b)
    got, _ = hip_ctc.ctc_greedy(b)
    assert got == want and sum(len(t) for t, _ in want) > 0


def test_ctc_fused_and_stream_bookkeeping(ctc_path, oracle_ctc, utts):
    """GetResults on a zipformer2ctc model (OfflineRecognizer.cs:44-47 forces greedy_search_ctc): new symbols are appended to
    the stream's own Tokens = [blank, blank] (no 2*B prefix), RemoveSamples runs, NumTrailingBlank is updated."""
    from k2transducerasr_amd import OfflineRecognizer
    rec = OfflineRecognizer(ctc_path)
    streams = []
    for u in utts[:3]:
        s = rec.create_offline_stream()
        s.add_samples(u)
        streams.append(s)
    feats = [oracle_ctc.fbank(u) for u in utts[:3]]
    lp = oracle_ctc.encoder(oracle_ctc.pad_sequence(feats).reshape(3, -1, 80))
    want, tb = oracle_ctc.ctc_greedy(lp)
    res = rec.get_results(streams)
    for b, s in enumerate(streams):
        assert res[b][0] == [0, 0] + want[b][0] and res[b][1] == want[b][1]
        assert s.speech_length == 0
        fo, ntb = C.c_int32(-1), C.c_int32(-1)
        rec.model._L.k2hip_offline_stream_get_ctc_state(C.c_void_p(s._h.value if hasattr(s._h, "value") else s._h), C.byref(fo), C.byref(ntb))
        assert fo.value == 0 and ntb.value == tb[b]
    # single path: Tokens = [-1, blank] + symbols (:318-320)
    s = rec.create_offline_stream()
    s.add_samples(utts[1])
    lp1 = oracle_ctc.encoder(oracle_ctc.pad_sequence([feats[1]]).reshape(1, -1, 80))
    want1, _ = oracle_ctc.ctc_greedy(lp1)
    tok, ts = rec.get_result(s)
    assert tok == [-1, 0] + want1[0][0] and ts == want1[0][1]


def test_online_ctc_matches_oracle(tmp_path_factory):
    from k2transducerasr_amd import OnlineRecognizer
    from k2transducerasr_amd.synth import synth_utterance, write_synthetic_model
    from oracle.online import OnlineOracle
    p = str(tmp_path_factory.mktemp("ctcsg") / "ctc_stream.k2w")
    write_synthetic_model(p, "zipformer2-ctc-streaming-tiny-test")
    oo = OnlineOracle(p)
    rec = OnlineRecognizer(p)
    utts = [synth_utterance(10 + u, 2.0) for u in range(3)]
    hs = [rec.create_online_stream() for _ in utts]
    os_ = [oo.create_stream() for _ in utts]
    feats = [oo.fbank(u) for u in utts]
    for h, f in zip(hs, feats):
        h.add_features(f)
    T, shift = oo.chunk_length, oo.shift_length
    pos = 0
    while pos + T <= feats[0].shape[0]:
        oo.step(os_, [f[pos : pos + T] for f in feats])
        rec.get_results(hs)
        pos += shift
    assert sum(len(o.tokens) - 2 for o in os_) > 0
    for h, o in zip(hs, os_):
        assert h.tokens == o.tokens and h.timestamps == o.timestamps
        assert h.hyp == [0, 0]                      # the CTC delegate never touches Hyp
