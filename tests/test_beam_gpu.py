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


@pytest.mark.parametrize("beam", [1, 4, 8])
def test_beam_search_launch_form_equals_the_one_kernel_form(hip_tiny, oracle_tiny, enc_tiny, beam):
    """A model with the all-contexts decoder table runs the whole search as one kernel per batch (k_beam_loop); without it (large
    vocabularies) the search is four launches per frame.  K2HIP_BEAM_LAUNCHES forces the launch form on a small model: both must
    equal the oracle, and each other in tokens and timestamps (their scores differ by float rounding only: the decoder projection
    is the search loops' own routine in one and a GEMM in the other)."""
    import k2transducerasr_amd as pkg
    want, mg, sc = oracle_tiny.modified_beam_search(enc_tiny, beam, want_margins=True, want_scores=True)
    loop, lsc = hip_tiny.beam_search(enc_tiny, beam, want_scores=True)
    pkg.set_switch("K2HIP_BEAM_LAUNCHES", 1)
    try:
        launches, csc = hip_tiny.beam_search(enc_tiny, beam, want_scores=True)
    finally:
        pkg.set_switch("K2HIP_BEAM_LAUNCHES", 0)
    _check(launches, want, mg, f"launch form, beam={beam}")
    assert launches == loop
    # long utterances: the one-kernel form with its hypotheses in device memory instead of LDS -- the same arithmetic, the same bits
    pkg.set_switch("K2HIP_BEAM_HYP_GLOBAL", 1)
    try:
        far, fsc = hip_tiny.beam_search(enc_tiny, beam, want_scores=True)
    finally:
        pkg.set_switch("K2HIP_BEAM_HYP_GLOBAL", 0)
    assert far == loop and np.array_equal(fsc, lsc)
    np.testing.assert_allclose(csc, sc, atol=2e-3, rtol=0)
    np.testing.assert_allclose(csc, lsc, atol=1e-4, rtol=0)


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


def test_three_batches_in_flight_with_the_beam_search(tiny_model_path):
    """k2hip.h K2HIP_MAX_BATCHES_IN_FLIGHT: with the modified beam search every pipeline slot's search runs on the slot's own
    stream, so three batches may be outstanding (the searches of batches i and i+1 beside the encoder of i+2); greedy search
    keeps two.  Tokens equal the synchronous call's, tickets may be waited for in any order, a further submit is refused, and
    switching the method back restores the two-deep pipeline."""
    from k2transducerasr_amd import K2HipError, Model
    from k2transducerasr_amd.synth import synth_utterance
    m = Model(tiny_model_path, 0)
    B, n = 3, 16000 * 2
    batches = [np.stack([synth_utterance(900 + 10 * k + u, 2.0)[:n] for u in range(B)]) for k in range(4)]
    ptrs = []
    try:
        for s in batches:
            p = m.device_alloc(s.nbytes)
            m.device_upload(p, s)
            ptrs.append(p)
        m.set_decoding_method("modified_beam_search", 4)
        want = [m.offline_greedy_from_samples_dev(p, n, B) for p in ptrs]
        assert any(len(tok) for r in want for tok, _ in r)
        t = [m.offline_submit_samples_dev(ptrs[k], n, B) for k in range(3)]
        with pytest.raises(K2HipError):
            m.offline_submit_samples_dev(ptrs[3], n, B)
        assert m.offline_wait(t[1]) == want[1]
        t3 = m.offline_submit_samples_dev(ptrs[3], n, B)  # the freed slot is found wherever it is
        assert m.offline_wait(t[0]) == want[0]
        assert m.offline_wait(t3) == want[3]
        assert m.offline_wait(t[2]) == want[2]
        # from host memory too (the copies go on the idle shared stream)
        th = [m.offline_submit_samples(batches[k], None) for k in range(3)]
        assert [m.offline_wait(x) for x in th] == want[:3]
        m.set_decoding_method("greedy_search")
        g = [m.offline_greedy_from_samples_dev(p, n, B) for p in ptrs[:2]]
        ta, tb = m.offline_submit_samples_dev(ptrs[0], n, B), m.offline_submit_samples_dev(ptrs[1], n, B)
        with pytest.raises(K2HipError):
            m.offline_submit_samples_dev(ptrs[2], n, B)
        assert m.offline_wait(ta) == g[0] and m.offline_wait(tb) == g[1]
    finally:
        for p in ptrs:
            m.device_free(p)

@pytest.mark.parametrize("vocab", [400, 600, 1029])
def test_one_kernel_beam_search_beyond_256_columns(tmp_path_factory, utts, vocab):
    """Vocabularies that still get the decoder table and span more than one 256-column chunk (V = 400: two column slabs per stream
    at beam <= 4, exchanging their logits every frame; V = 600: three passes; V = 1029: an odd size, padded columns): one-kernel form against the oracle and the launch
    form, greedy search of the same model against the oracle (its slabs then span several chunks too)."""
    import k2transducerasr_amd as pkg
    from k2transducerasr_amd import Model
    from k2transducerasr_amd.synth import write_synthetic_model
    from oracle import Oracle
    from parity import assert_tokens_match
    p = str(tmp_path_factory.mktemp(f"v{vocab}") / "m.k2w")
    write_synthetic_model(p, "zipformer2-tiny-test", meta_overrides={"vocab_size": str(vocab)})
    hip, ora = Model(p, 0), Oracle(p)
    feats = [ora.fbank(u) for u in utts]
    enc = ora.encoder(ora.pad_sequence(feats).reshape(len(utts), -1, 80))
    for beam in (4, 8):
        want, mg = ora.modified_beam_search(enc, beam, want_margins=True)
        loop = hip.beam_search(enc, beam)
        _check(loop, want, mg, f"V={vocab} beam={beam}")
        pkg.set_switch("K2HIP_BEAM_LAUNCHES", 1)
        try:
            assert hip.beam_search(enc, beam) == loop
        finally:
            pkg.set_switch("K2HIP_BEAM_LAUNCHES", 0)
    want = ora.recognize_batch(feats)
    _, mg = ora.greedy_batch(enc, want_margins=True)
    assert_tokens_match(hip.offline_greedy(feats), want, mg, what=f"V={vocab} greedy")
