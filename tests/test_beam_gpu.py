"""GPU parity for modified beam search (BASELINE.json configs[2]): libk2hip.so through the C ABI against
oracle/k2_oracle_beam.c (icefall semantics; the reference itself has no beam search)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from parity import assert_beam_match


def _check(got, want, margins, what):
    assert_beam_match(got, want, margins, what=what)   # strict: tokens and timestamps identical


@pytest.fixture(scope="module")
def enc_tiny(oracle_tiny, utts):
    f = [oracle_tiny.fbank(u) for u in utts]
    return oracle_tiny.encoder(oracle_tiny.pad_sequence(f).reshape(len(utts), -1, 80))


@pytest.mark.parametrize("beam", [1, 2, 4, 8])
def test_beam_search_matches_oracle(hip_tiny, oracle_tiny, enc_tiny, beam):
    want, mg, sc = oracle_tiny.modified_beam_search(enc_tiny, beam, want_margins=True, want_scores=True)
    got, gsc = hip_tiny.beam_search(enc_tiny, beam, want_scores=True)
    assert sum(len(t) for t, _ in want) > 0
    _check(got, want, mg, f"beam={beam}")
    np.testing.assert_allclose(gsc, sc, atol=2e-3, rtol=0)


def test_beam_search_stream_independence(hip_tiny, enc_tiny):
    # x_lens = T' for all streams: each stream's search is independent of its batch mates
    full = hip_tiny.beam_search(enc_tiny, 4)
    for b in range(enc_tiny.shape[0]):
        assert hip_tiny.beam_search(enc_tiny[b : b + 1], 4)[0] == full[b]


def test_beam_known_answers_and_merge(kat_hip_beam):
    from kat_model import frames
    hip, ora = kat_hip_beam
    enc = frames([{5: 9.0}, {0: 9.0}, {2: 9.0}, {7: 9.0}, {0: 9.0}])
    assert hip.beam_search(enc[None], 4)[0] == ([5, 7], [0, 3])  # blank and unk never enter ys
    assert hip.beam_search(frames([{0: 9.0}] * 3)[None], 4)[0] == ([], [])
    # two paths to the same ys merge by logaddexp; the first-inserted hypothesis keeps its timestamps
    enc = frames([{5: 1.0, 0: 1.0}, {5: 1.0, 0: 1.0}])
    want, sc = ora.modified_beam_search(enc[None], 4, want_scores=True)
    got, gsc = hip.beam_search(enc[None], 4, want_scores=True)
    assert got == want
    np.testing.assert_allclose(gsc, sc, atol=1e-5)


@pytest.fixture(scope="module")
def kat_hip_beam(tmp_path_factory):
    from kat_model import write_kat_model
    from k2transducerasr_amd import Model
    from oracle import Oracle
    p = str(tmp_path_factory.mktemp("katbg") / "kat.k2w")
    write_kat_model(p)
    return Model(p, 0), Oracle(p)


def test_decoding_method_switch_on_fused_path(tiny_model_path, oracle_tiny, utts):
    """OfflineRecognizer(decodingMethod=...) (OfflineRecognizer.cs:54-68): the fused batch entry follows the model's method."""
    from k2transducerasr_amd import K2HipError, Model
    m = Model(tiny_model_path, 0)
    feats = [oracle_tiny.fbank(u) for u in utts]
    enc = oracle_tiny.encoder(oracle_tiny.pad_sequence(feats).reshape(len(utts), -1, 80))
    greedy = m.offline_greedy(feats)
    m.set_decoding_method("modified_beam_search", 4)
    want, mg = oracle_tiny.modified_beam_search(enc, 4, want_margins=True)
    got = m.offline_greedy(feats)
    _check(got, want, mg, "fused beam")
    assert m.last_scores(len(utts)).shape == (len(utts),)
    m.set_decoding_method("greedy_search")
    assert m.offline_greedy(feats) == greedy
    with pytest.raises(K2HipError):
        m.set_decoding_method("fast_beam_search")
    with pytest.raises(K2HipError):
        m.set_decoding_method("modified_beam_search", 9)


def test_beam_search_conformer_large_vocab(hip_conformer, oracle_conformer, utts):
    f = [oracle_conformer.fbank(u) for u in utts[:3]]
    enc = oracle_conformer.encoder(oracle_conformer.pad_sequence(f).reshape(3, -1, 80))
    want, mg = oracle_conformer.modified_beam_search(enc, 4, want_margins=True)
    _check(hip_conformer.beam_search(enc, 4), want, mg, "conformer beam")
