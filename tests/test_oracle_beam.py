"""CPU tests for the modified-beam-search restatement (BASELINE.json configs[2]; the reference itself has no beam search,
SURVEY.md section 0): oracle/k2_oracle_beam.c against a literal transcription of icefall's modified_beam_search built on
torch ops (log_softmax, topk, logaddexp) and the oracle's own decoder / joiner operators, plus hand-derived known answers."""
import numpy as np
import pytest
import torch

from kat_model import frames, write_kat_model


def icefall_modified_beam_search(oracle, enc_out, beam=4):
    """beam_search.py modified_beam_search for one stream (x_lens = T'), Hypothesis / HypothesisList as dict keyed by ys."""
    blank_id, unk_id, context_size = 0, 2, oracle.context_size
    B = {}
    ys0 = [blank_id] * context_size
    B[tuple(ys0)] = dict(ys=ys0, log_prob=torch.zeros(1, dtype=torch.float32), timestamp=[])
    T = enc_out.shape[0]
    for t in range(T):
        A = list(B.values())
        B = {}
        ys_log_probs = torch.cat([h["log_prob"].reshape(1, 1) for h in A])
        decoder_input = np.array([h["ys"][-context_size:] for h in A], np.int64)
        decoder_out = oracle.decoder(decoder_input)
        cur = np.repeat(enc_out[t : t + 1], len(A), 0)
        logits = torch.from_numpy(oracle.joiner(cur, decoder_out))
        log_probs = logits.log_softmax(dim=-1)
        log_probs.add_(ys_log_probs)
        vocab_size = log_probs.size(-1)
        log_probs = log_probs.reshape(-1)
        topk_log_probs, topk_indexes = log_probs.topk(min(beam, log_probs.numel()))
        topk_hyp_indexes = (topk_indexes // vocab_size).tolist()
        topk_token_indexes = (topk_indexes % vocab_size).tolist()
        for k in range(len(topk_hyp_indexes)):
            hyp = A[topk_hyp_indexes[k]]
            new_ys = hyp["ys"][:]
            new_token = topk_token_indexes[k]
            new_timestamp = hyp["timestamp"][:]
            if new_token not in (blank_id, unk_id):
                new_ys.append(new_token)
                new_timestamp.append(t)
            new_log_prob = topk_log_probs[k].reshape(1)
            key = tuple(new_ys)
            if key in B:
                B[key]["log_prob"] = torch.logaddexp(B[key]["log_prob"], new_log_prob)
            else:
                B[key] = dict(ys=new_ys, log_prob=new_log_prob, timestamp=new_timestamp)
    best = max(B.values(), key=lambda h: h["log_prob"] / len(h["ys"]))
    return best["ys"][context_size:], best["timestamp"], float(best["log_prob"])


@pytest.fixture(scope="module")
def enc_tiny(oracle_tiny, utts):
    f = [oracle_tiny.fbank(u) for u in utts]
    x = oracle_tiny.pad_sequence(f).reshape(len(utts), -1, 80)
    return oracle_tiny.encoder(x)


@pytest.mark.parametrize("beam", [1, 2, 4, 8])
def test_beam_oracle_matches_icefall_transcription(oracle_tiny, enc_tiny, beam):
    res, mg, sc = oracle_tiny.modified_beam_search(enc_tiny, beam, want_margins=True, want_scores=True)
    assert sum(len(t) for t, _ in res) > 0
    for b in range(enc_tiny.shape[0]):
        ys, ts, lp = icefall_modified_beam_search(oracle_tiny, enc_tiny[b], beam)
        if (ys, ts) != res[b]:
            assert float(mg[b].min()) < 1e-4, f"stream {b}: {res[b]} vs {(ys, ts)} with margin {mg[b].min()}"
            continue
        assert abs(lp - float(sc[b])) < 1e-3


def test_beam_one_without_merging_is_greedy_like(oracle_tiny, enc_tiny):
    # beam = 1 keeps the single best continuation per frame = greedy from ctx [blank, blank] with the {blank, unk} skip
    res = oracle_tiny.modified_beam_search(enc_tiny[:1], 1)
    ys, ts, _ = icefall_modified_beam_search(oracle_tiny, enc_tiny[0], 1)
    assert res[0] == (ys, ts)


@pytest.fixture(scope="module")
def kat_oracle(tmp_path_factory):
    from oracle import Oracle
    p = str(tmp_path_factory.mktemp("katb") / "kat.k2w")
    write_kat_model(p)
    return Oracle(p)


def test_beam_known_answers(kat_oracle):
    # KAT joiner: logits = enc[:V] + context boost (tests/kat_model.py).  Strongly peaked frames: beam search returns the
    # same tokens as greedy; unk (2) and blank (0) never enter ys (new_token not in (blank_id, unk_id)).
    enc = frames([{5: 9.0}, {0: 9.0}, {2: 9.0}, {7: 9.0}, {0: 9.0}])
    res, mg = kat_oracle.modified_beam_search(enc[None], 4, want_margins=True)
    assert res[0] == ([5, 7], [0, 3])
    # empty input / all blank
    res = kat_oracle.modified_beam_search(frames([{0: 9.0}] * 3)[None], 4)
    assert res[0] == ([], [])


def test_beam_merges_equal_sequences(kat_oracle):
    """Two paths to the same ys -- 'emit 5 at frame 0, blank at 1' and 'blank at 0, emit 5 at 1' -- must merge by logaddexp and
    keep the first-inserted hypothesis' timestamps (HypothesisList.add)."""
    enc = frames([{5: 1.0, 0: 1.0}, {5: 1.0, 0: 1.0}])
    res, mg, sc = kat_oracle.modified_beam_search(enc[None], 4, want_margins=True, want_scores=True)
    ys, ts, lp = icefall_modified_beam_search(kat_oracle, enc, 4)
    assert ys == [5]
    assert res[0] == (ys, ts)
    assert abs(lp - float(sc[0])) < 1e-5


def _as_engine_tap(tr):
    """an oracle tap cut down to what the engine's tap holds (the `beam` selected candidates)"""
    K = tr["beam"]
    return dict(idx=tr["idx"][:, :, :K], val=tr["val"][:, :, :K], n=tr["n"], beam=K)


def test_trace_tap_is_consistent_with_the_search(oracle_tiny, enc_tiny):
    """The per-frame tap the divergence localiser relies on: ranked (score desc, flat index asc), the first `beam` entries are the
    frame's selection, the margin is the gap between entries beam-1 and beam, survivors <= selection size; tracing changes nothing."""
    beam = 4
    res, mg, tr = oracle_tiny.modified_beam_search(enc_tiny, beam, want_margins=True, want_trace=True)
    assert res == oracle_tiny.modified_beam_search(enc_tiny, beam)
    idx, val, n = tr["idx"], tr["val"], tr["n"]
    B, Tp, E = idx.shape
    assert E == 2 * beam and n.shape == (B, Tp)
    V = oracle_tiny.vocab_size
    for b in range(B):
        for t in range(Tp):
            live = idx[b, t] >= 0
            v = val[b, t][live]
            assert (np.diff(v) <= 0).all()
            ties = np.nonzero(np.diff(v) == 0)[0]
            assert all(idx[b, t, i] < idx[b, t, i + 1] for i in ties)
            assert 1 <= n[b, t] <= min(beam, live.sum())
            assert (idx[b, t][live] // V < (1 if t == 0 else n[b, t - 1])).all()     # slots of the previous frame's survivors
            if live.sum() > beam:
                assert mg[b, t] == pytest.approx(val[b, t, beam - 1] - val[b, t, beam], abs=0)
    assert (n[:, 0] >= 1).all() and (idx[:, 0, :beam] < V).all()       # frame 0 expands the single start hypothesis


def test_divergence_localiser(oracle_tiny, enc_tiny):
    """tests/parity.py localise_beam / assert_beam_match on two oracle runs: equal inputs -> no divergence; a perturbed encoder_out
    plays the engine -- every stream whose result changes must be localised to a frame at which the unperturbed oracle's own scores
    of the candidates in question are closer than the perturbation can move them, and a gross perturbation must be REJECTED."""
    import parity
    beam = 4
    want, mg, tr_w = oracle_tiny.modified_beam_search(enc_tiny, beam, want_margins=True, want_trace=True)
    same = _as_engine_tap(tr_w)
    for b in range(len(want)):
        assert parity.localise_beam(same, tr_w, b) == (None, None)
    rng = np.random.default_rng(5)
    found = 0
    for trial in range(40):
        noisy = enc_tiny + rng.standard_normal(enc_tiny.shape).astype(np.float32) * 2e-3
        got, tr_g = oracle_tiny.modified_beam_search(noisy, beam, want_trace=True)
        tap = _as_engine_tap(tr_g)
        for b in range(len(want)):
            t, gap = parity.localise_beam(tap, tr_w, b)
            if t is None:
                if got[b] != want[b]:            # the searches agree on every frame: only the final pick can differ
                    assert mg[b, -1] < 0.1
                continue
            found += 1
            assert 0 <= t < enc_tiny.shape[1] and gap < 0.1, (trial, b, t, gap)    # 2e-3 of input noise moves scores by far less than 0.1
            # the localised frame is the FIRST differing one
            assert (tap["idx"][b, :t] == tr_w["idx"][b, :t, :beam]).all()
    assert found > 0, "the perturbation never moved a selection: the test does not exercise the localiser"
    # a gross perturbation: streams differ, and the localiser must refuse to call it a near-tie
    gross = enc_tiny[::-1].copy()
    got, tr_g = oracle_tiny.modified_beam_search(gross, beam, want_trace=True)
    if got != want:
        n_before = len(parity.NEAR_TIES)
        with pytest.raises(AssertionError):
            parity.assert_beam_match(got, want, mg, tol=1e-3, what="gross", allow_tie=True, trace_got=_as_engine_tap(tr_g), trace_want=tr_w)
        del parity.NEAR_TIES[n_before:]
        parity.COMPARED[0] -= len(want)
    with pytest.raises(AssertionError):          # no taps, no excuse
        parity.assert_beam_match([([1], [0])], [([2], [0])], np.zeros((1, 3)), allow_tie=True)
    parity.COMPARED[0] -= 1
