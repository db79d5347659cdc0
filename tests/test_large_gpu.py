"""GPU parity at the benchmark's architecture (zipformer2-large-en, random weights):
a batch small enough for the oracle to finish in seconds, plus size-independent
properties at the full B=32 x 10 s configuration."""
import numpy as np
import pytest

from parity import LOGIT_TOL, assert_tokens_match

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def large_path(tmp_path_factory):
    from k2transducerasr_amd.synth import write_synthetic_model
    p = str(tmp_path_factory.mktemp("large") / "large.k2w")
    write_synthetic_model(p, "zipformer2-large-en")
    return p


@pytest.fixture(scope="module")
def hip_large(large_path):
    from k2transducerasr_amd import Model
    return Model(large_path, 0)


@pytest.fixture(scope="module")
def oracle_large(large_path):
    from oracle import Oracle
    return Oracle(large_path)


def test_large_encoder_and_tokens_match_oracle(hip_large, oracle_large):
    from k2transducerasr_amd.synth import synth_utterance
    utts = [synth_utterance(200 + u, s) for u, s in enumerate([4.0, 3.1, 4.0])]
    feats = [oracle_large.fbank(u) for u in utts]
    x = oracle_large.pad_sequence(feats).reshape(len(utts), -1, 80)
    enc_o = oracle_large.encoder(x)
    enc_h = hip_large.encoder_proj(x)
    assert enc_h.shape == enc_o.shape
    # 19 layers of fp32 in a different summation order: activations O(1), agreement ~1e-5
    np.testing.assert_allclose(enc_h, enc_o, atol=5e-4, rtol=0)
    # logits through the joiner stay inside the north-star tolerance
    dec = oracle_large.decoder(np.array([[-1, 0]], np.int64))
    lo = oracle_large.joiner(enc_o[0], np.repeat(dec, enc_o.shape[1], 0))
    lh = hip_large.joiner_proj(enc_h[0], np.repeat(dec, enc_o.shape[1], 0))
    assert float(np.abs(lo - lh).max()) < LOGIT_TOL
    want, mg = oracle_large.greedy_batch(enc_o, want_margins=True)
    assert sum(len(w[0]) for w in want) > 0
    got = hip_large.offline_greedy_from_samples(utts)
    assert_tokens_match(got, want, mg, what="large e2e")
    # the conv modules' GLU runs in the in_proj GEMM's epilogue (weights interleaved at load) when a stack has >= 256 rows -- here
    # the 50 / 25 Hz stacks do, the others do not; K2HIP_NO_GLU_EPILOGUE puts it back into the depthwise kernel everywhere.
    # The same math; the compiler is free to form value * sigmoid(gate) differently in the two kernels (reciprocal / division), so
    # the encoder outputs agree to rounding, not bit for bit.
    from k2transducerasr_amd import set_switch
    set_switch("K2HIP_NO_GLU_EPILOGUE", 1)
    try:
        enc_g = hip_large.encoder_proj(x)
    finally:
        set_switch("K2HIP_NO_GLU_EPILOGUE", 0)
    np.testing.assert_allclose(enc_g, enc_h, atol=2e-5, rtol=0)
    np.testing.assert_allclose(enc_g, enc_o, atol=5e-4, rtol=0)
    # the 7x7 depthwise convolution of the embed: the sliding LDS-DMA kernel (inputs of >= 48 frames) against the one-shot tiled
    # kernel -- the same sums in the same order, so the whole encoder output is bit-identical
    set_switch("K2HIP_DW7_TILED", 1)
    try:
        enc_t = hip_large.encoder_proj(x)
    finally:
        set_switch("K2HIP_DW7_TILED", 0)
    np.testing.assert_array_equal(enc_t, enc_h)


def test_full_size_batch_properties(hip_large):
    """B=32 x 10 s (BASELINE configs[1]): (1) Q3 -- equal-length rows do not depend on
    their batch mates, so any row of the batched encoder equals the same utterance run
    alone; (2) determinism; (3) output geometry T=1017 -> T'=253."""
    from k2transducerasr_amd.synth import synth_utterance
    B = 32
    s = np.stack([synth_utterance(u, 10.0) for u in range(B)])
    feats = [hip_large.fbank(s[b]) for b in range(B)]
    assert feats[0].shape == (998, 80)
    x = hip_large.pad_sequence(feats).reshape(B, -1, 80)
    assert x.shape[1] == 1017
    enc = hip_large.encoder_proj(x)
    assert enc.shape == (B, 253, 512) and np.isfinite(enc).all()
    for b in (0, 17, 31):
        alone = hip_large.encoder_proj(x[b : b + 1])
        np.testing.assert_allclose(enc[b], alone[0], atol=2e-5, rtol=0)
    ptr = hip_large.device_alloc(s.nbytes)
    try:
        hip_large.device_upload(ptr, s)
        r1 = hip_large.offline_greedy_from_samples_dev(ptr, s.shape[1], B)
        r2 = hip_large.offline_greedy_from_samples_dev(ptr, s.shape[1], B)
    finally:
        hip_large.device_free(ptr)
    assert r1 == r2
    # pipelined submit/wait gives the same tokens, in submission order, with two batches in flight (greedy search: two slots)
    ptr2 = hip_large.device_alloc(s.nbytes)
    try:
        hip_large.device_upload(ptr2, s[::-1].copy())
        ptr3 = hip_large.device_alloc(s.nbytes)
        hip_large.device_upload(ptr3, s)
        ta = hip_large.offline_submit_samples_dev(ptr3, s.shape[1], B)
        tb = hip_large.offline_submit_samples_dev(ptr2, s.shape[1], B)
        from k2transducerasr_amd import K2HipError
        with pytest.raises(K2HipError):
            hip_large.offline_submit_samples_dev(ptr3, s.shape[1], B)  # only two slots
        ra = hip_large.offline_wait(ta)
        rb = hip_large.offline_wait(tb)
        assert ra == r1
        assert rb == hip_large.offline_greedy_from_samples_dev(ptr2, s.shape[1], B)
        hip_large.device_free(ptr3)
    finally:
        hip_large.device_free(ptr2)
    assert r1 == hip_large.greedy_batch(enc)
    for tok, ts in r1:
        assert len(tok) == len(ts) and all(0 <= t < 253 for t in ts) and ts == sorted(ts)
        assert len(set(ts)) == len(ts)  # at most one symbol per frame (Q5)
        assert all(t not in (0, 2) and 0 < t < 500 for t in tok)


def test_full_size_matches_oracle(hip_large, oracle_large):
    """BASELINE configs[1] at its own size against the oracle, strictly: B = 32 x 10 s (8 096 argmax decisions, ~1 600 emissions per
    batch; OfflineRecognizer.cs:189-303).  The large grids select kernel paths the short-utterance tests never reach (128 x 128
    tiles, GLU epilogue on >= 256-row stacks, two-slab search, pipeline slots), so the property tests alone cannot see a
    wrong-but-deterministic path.  Synchronous entry AND pipelined submit / wait (device-resident and host-memory samples) against
    `recognize_batch`; three rows of the encoder output (5e-4) and their joiner logits (north-star 1e-3)."""
    from k2transducerasr_amd.synth import synth_utterance
    B = 32
    s = np.stack([synth_utterance(u, 10.0) for u in range(B)])
    feats = [oracle_large.fbank(s[b]) for b in range(B)]
    want = oracle_large.recognize_batch(feats)
    assert sum(len(w[0]) for w in want) > 32 * 253 // 10          # the calibrated ~20 % emission rate: a real token load
    ptr = hip_large.device_alloc(s.nbytes)
    try:
        hip_large.device_upload(ptr, s)
        got = hip_large.offline_greedy_from_samples_dev(ptr, s.shape[1], B)
        assert_tokens_match(got, want, what="configs[1] full size, synchronous")
        ta = hip_large.offline_submit_samples_dev(ptr, s.shape[1], B)
        tb = hip_large.offline_submit_samples_dev(ptr, s.shape[1], B)
        assert_tokens_match(hip_large.offline_wait(ta), want, what="configs[1] full size, pipelined slot 0")
        assert_tokens_match(hip_large.offline_wait(tb), want, what="configs[1] full size, pipelined slot 1")
    finally:
        hip_large.device_free(ptr)
    h = hip_large.host_alloc(s.shape)
    try:
        h[:] = s
        assert_tokens_match(hip_large.offline_wait(hip_large.offline_submit_samples(h)), want, what="configs[1] full size, from host memory")
    finally:
        hip_large.host_free(h)
    # encoder output of the whole batch through the full-size grids, three rows against the oracle
    x = oracle_large.pad_sequence(feats).reshape(B, -1, 80)
    assert x.shape[1] == 1017
    enc_h = hip_large.encoder_proj(x)
    rows = (0, 13, 31)
    enc_o = oracle_large.encoder(x[list(rows)])   # rows are independent (Q3), so the oracle needs only these three
    dec = oracle_large.decoder(np.array([[-1, 0]], np.int64))
    for i, b in enumerate(rows):
        np.testing.assert_allclose(enc_h[b], enc_o[i], atol=5e-4, rtol=0)
        lo = oracle_large.joiner(enc_o[i], np.repeat(dec, enc_o.shape[1], 0))
        lh = hip_large.joiner_proj(enc_h[b], np.repeat(dec, enc_o.shape[1], 0))
        assert float(np.abs(lo - lh).max()) < LOGIT_TOL


def test_long_utterances_beyond_the_lds_strip(hip_tiny, oracle_tiny, hip_conformer, oracle_conformer):
    """Attention for sequences longer than the in-LDS score strip (Zipformer2: > ~1120 frames at 50 Hz, i.e. > 22 s; Conformer:
    > 2048 frames at 25 Hz) takes the two-pass kernels; same numbers, no length limit."""
    from k2transducerasr_amd.synth import synth_utterance
    u = synth_utterance(77, 31.0)                      # T = 3117 -> T50 = 1555
    f = oracle_tiny.fbank(u)
    x = oracle_tiny.pad_sequence([f]).reshape(1, -1, 80)
    assert (x.shape[1] - 7) // 2 > 1275
    np.testing.assert_allclose(hip_tiny.encoder_proj(x), oracle_tiny.encoder(x), atol=5e-4, rtol=0)
    u2 = synth_utterance(78, 84.0)                     # T' = 2104 > 2048
    f2 = oracle_conformer.fbank(u2)
    x2 = oracle_conformer.pad_sequence([f2]).reshape(1, -1, 80)
    assert oracle_conformer.encoder_out_frames(x2.shape[1]) > 2048
    np.testing.assert_allclose(hip_conformer.encoder_proj(x2), oracle_conformer.encoder(x2), atol=5e-4, rtol=0)


# ---- BASELINE configs[2]: modified beam search, beam = 4, on zipformer2-large-en ---------------------------
def test_large_beam4_matches_oracle(hip_large, oracle_large):
    """Three short utterances on the benchmark architecture: tokens, timestamps and scores of the device search equal
    oracle/k2_oracle_beam.c (icefall modified_beam_search; the reference itself has no beam search,
    OfflineRecognizer.cs:54-68), both at the operator level (same encoder_out) and through the fused samples -> tokens
    entry selected by set_decoding_method."""
    from parity import assert_beam_match
    from k2transducerasr_amd.synth import synth_utterance
    utts = [synth_utterance(400 + u, s) for u, s in enumerate([2.4, 1.9, 2.4])]
    feats = [oracle_large.fbank(u) for u in utts]
    x = oracle_large.pad_sequence(feats).reshape(len(utts), -1, 80)
    enc_o = oracle_large.encoder(x)
    want, mg, sc = oracle_large.modified_beam_search(enc_o, 4, want_margins=True, want_scores=True)
    assert sum(len(w[0]) for w in want) > 0
    got, gsc = hip_large.beam_search(enc_o, 4, want_scores=True)
    assert_beam_match(got, want, mg, what="large beam-4 (oracle encoder_out)")
    np.testing.assert_allclose(gsc, sc, atol=2e-3, rtol=0)
    hip_large.set_decoding_method("modified_beam_search", 4)
    try:
        got2 = hip_large.offline_greedy_from_samples(utts)
        assert_beam_match(got2, want, mg, what="large beam-4 (fused, from samples)")
        np.testing.assert_allclose(hip_large.last_scores(len(utts)), sc, atol=2e-3, rtol=0)
    finally:
        hip_large.set_decoding_method("greedy_search")


def test_full_size_beam4_properties(hip_large):
    """One GPU's shard of configs[2] (32 of the 256 utterances, 10 s each, beam 4): determinism, stream independence
    (x_lens = T' for every stream, so a stream searched alone gives the same hypothesis), pipelined == synchronous under
    set_decoding_method, and the result geometry."""
    from k2transducerasr_amd.synth import synth_utterance
    B = 32
    s = np.stack([synth_utterance(u, 10.0) for u in range(B)])
    ptr = hip_large.device_alloc(s.nbytes)
    hip_large.set_decoding_method("modified_beam_search", 4)
    try:
        hip_large.device_upload(ptr, s)
        r1 = hip_large.offline_greedy_from_samples_dev(ptr, s.shape[1], B)
        sc1 = hip_large.last_scores(B)
        r2 = hip_large.offline_greedy_from_samples_dev(ptr, s.shape[1], B)
        assert r1 == r2 and np.array_equal(sc1, hip_large.last_scores(B))
        ta = hip_large.offline_submit_samples_dev(ptr, s.shape[1], B)
        tb = hip_large.offline_submit_samples_dev(ptr, s.shape[1], B)
        assert hip_large.offline_wait(ta) == r1 and hip_large.offline_wait(tb) == r1
        feats = [hip_large.fbank(s[b]) for b in range(B)]
        enc = hip_large.encoder_proj(hip_large.pad_sequence(feats).reshape(B, -1, 80))
        full, fsc = hip_large.beam_search(enc, 4, want_scores=True)
        assert full == r1
        for b in (0, 13, 31):
            alone, asc = hip_large.beam_search(enc[b : b + 1], 4, want_scores=True)
            assert alone[0] == full[b] and abs(float(asc[0]) - float(fsc[b])) < 1e-4
    finally:
        hip_large.set_decoding_method("greedy_search")
        hip_large.device_free(ptr)
    assert sum(len(t) for t, _ in r1) > 0
    for tok, ts in r1:
        assert len(tok) == len(ts) and all(0 <= t < 253 for t in ts) and ts == sorted(ts)
        assert all(t not in (0, 2) and 0 < t < 500 for t in tok)      # blank / unk never enter ys
    assert np.isfinite(sc1).all() and (sc1 < 0).all()                 # log-probabilities


def test_full_size_beam4_matches_oracle(hip_large, oracle_large):
    """One GPU's shard of configs[2] at its own size against the oracle: 32 x 10 s, beam 4 -- tokens and timestamps exact, scores
    within 2e-3 (sums of <= 253 log-probabilities whose logits carry the 1e-3 tolerance), through the fused samples -> tokens
    entry and the three-slot pipeline."""
    from parity import assert_beam_match
    from k2transducerasr_amd.synth import synth_utterance
    B = 32
    s = np.stack([synth_utterance(u, 10.0) for u in range(B)])
    feats = [oracle_large.fbank(s[b]) for b in range(B)]
    enc_o = oracle_large.encoder(oracle_large.pad_sequence(feats).reshape(B, -1, 80))
    want, mg, sc = oracle_large.modified_beam_search(enc_o, 4, want_margins=True, want_scores=True)
    assert sum(len(w[0]) for w in want) > 32 * 253 // 10
    ptr = hip_large.device_alloc(s.nbytes)
    hip_large.set_decoding_method("modified_beam_search", 4)
    try:
        hip_large.device_upload(ptr, s)
        got = hip_large.offline_greedy_from_samples_dev(ptr, s.shape[1], B)
        assert_beam_match(got, want, mg, what="configs[2] shard full size, synchronous")
        np.testing.assert_allclose(hip_large.last_scores(B), sc, atol=2e-3, rtol=0)
        tk = [hip_large.offline_submit_samples_dev(ptr, s.shape[1], B) for _ in range(3)]
        for i, t in enumerate(tk):
            assert_beam_match(hip_large.offline_wait(t), want, mg, what=f"configs[2] shard full size, pipelined slot {i}")
    finally:
        hip_large.set_decoding_method("greedy_search")
        hip_large.device_free(ptr)


def _beam_two_levels(hip, ora, utts, beam, what, tol):
    """One batch through the modified beam search at two levels, every differing stream localised by the per-frame taps:
    (operator) the engine's search on the ORACLE's encoder_out -- only joiner + search differ; (fused) samples -> tokens, where
    the engine's own encoder (5e-4 from the oracle's) feeds the search -- in BOTH forms of the engine's search (one kernel per batch;
    four launches per frame, K2HIP_BEAM_LAUNCHES).  Returns (exact_operator, exact_fused, hidden, gaps, forms) with hidden = streams
    whose results agree although the searches parted on some frame, gaps = the oracle's own score gap at the localised frame of every
    tolerated stream (all three runs), forms = [(stream, gap of the one-kernel form or None, gap of the launch form or None)] for the
    streams the two forms decide differently."""
    import parity
    from k2transducerasr_amd import set_switch
    from parity import assert_beam_match, hidden_beam_divergences
    B = len(utts)
    feats = [ora.fbank(u) for u in utts]
    enc_o = ora.encoder(ora.pad_sequence(feats).reshape(B, -1, 80))
    want, mg, tr_w = ora.modified_beam_search(enc_o, beam, want_margins=True, want_trace=True)
    n0 = len(parity.NEAR_TIES)
    set_switch("K2HIP_BEAM_TRACE", 1)
    try:
        got_op = hip.beam_search(enc_o, beam)
        tr_op = hip.beam_trace()
        ex_op = assert_beam_match(got_op, want, mg, tol=tol, what=f"{what} (operator level)", allow_tie=True, trace_got=tr_op, trace_want=tr_w)
        hip.set_decoding_method("modified_beam_search", beam)
        try:
            got_f = hip.offline_greedy_from_samples(utts)
            tr_f = hip.beam_trace()
            set_switch("K2HIP_BEAM_LAUNCHES", 1)
            try:
                got_l = hip.offline_greedy_from_samples(utts)
                tr_l = hip.beam_trace()
            finally:
                set_switch("K2HIP_BEAM_LAUNCHES", 0)
        finally:
            hip.set_decoding_method("greedy_search")
        n1 = len(parity.NEAR_TIES)
        ex_f = assert_beam_match(got_f, want, mg, tol=tol, what=f"{what} (fused)", allow_tie=True, trace_got=tr_f, trace_want=tr_w)
        n2 = len(parity.NEAR_TIES)
        assert_beam_match(got_l, want, mg, tol=tol, what=f"{what} (fused, launch form)", allow_tie=True, trace_got=tr_l, trace_want=tr_w)
        hidden = hidden_beam_divergences(got_f, want, tr_f, tr_w)
        assert all(gap < tol for _, _, gap in hidden), hidden     # a search that parts from the oracle does so on a near-tie, whatever the result
    finally:
        set_switch("K2HIP_BEAM_TRACE", 0)
    ties = parity.NEAR_TIES
    gaps = [e[3] for e in ties[n0:]]
    gap_f = {e[1]: e[3] for e in ties[n1:n2]}          # stream -> gap where the one-kernel form left the oracle
    gap_l = {e[1]: e[3] for e in ties[n2:]}            # ... the launch form
    forms = [(b, gap_f.get(b), gap_l.get(b)) for b in range(B) if got_f[b] != got_l[b]]
    return ex_op, ex_f, hidden, gaps, forms


def test_full_size_beam4_fresh_audio_every_divergence_localised(hip_large, oracle_large):
    """configs[2] shard on audio NO earlier run has seen: the utterance seeds come from the clock (K2HIP_SOAK_SEED pins them; the seed is
    printed, so a failure can be replayed).  Beam search has no reference behaviour, so the oracle is the only truth; on fresh audio
    ~5 % of the streams meet a frame whose candidates are closer than the two encoders agree, and such a stream may end differently.
    This test accepts that ONLY where the two per-frame taps localise it: the first frame at which the selections differ, and the
    oracle's own scores of the candidates in question at THAT frame within 1e-3 (tests/parity.py localise_beam).  On the oracle's own
    encoder_out (operator level) only the joiner's summation order and the exp / log of the log-softmax differ; even there about one
    stream in forty parts from the oracle, always between candidates whose oracle scores agree to the last float32 bit or two (gap
    0.0 .. 1e-5 at |score| ~ 100) -- measured 5 of 192 in tools/soak_full_size.py, and unchanged when both sides accumulated the
    hypothesis scores in float64 (DESIGN.md "tried", round 4): those are ties of the algorithm, not of its bookkeeping.

    Round 5, the gate at what was MEASURED (profiles/r04_soak_full_size_10_batches.txt: operator level 311 / 320, fused 305 / 320, largest
    localised gap 1.2e-4): operator level >= 30 / 32 and fused >= 29 / 32 -- or, on a batch with more near-ties than that, every
    localised gap <= 2e-4; and the engine's two forms of the search (one kernel per batch, four launches per frame) must decide every
    stream alike unless the oracle's own candidates are a near-tie there, by the same 2e-4: the two forms sum the joiner's products in
    different orders (a sweep on the matrix pipe in the one-kernel form, a GEMM in the launch form), and at |score| ~ 100 a gap of
    6.1e-5 is eight float32 steps.  (The round's first version of this gate demanded an EXACT tie, gap 0.0; the round's own soak --
    tools/soak_full_size.py on two sets of seeds: 7 of 320 streams decided differently by the forms, 4 of them at 6.1e-5, 3 at 0.0;
    3 of 320 in profiles/r05_soak_full_size_10_batches.txt, one at 6.1e-5 --
    and this test's clock-seeded audio then showed that bar failing one run in three on rounding alone.)  A form-dependent result at a
    gap beyond the near-tie bound would be a search-kernel bug."""
    import os
    import time
    from k2transducerasr_amd.synth import synth_utterance
    seed = int(os.environ.get("K2HIP_SOAK_SEED", "0")) or (int(time.time()) % 1_000_000) * 64 + 100_000
    print(f"fresh-audio beam test: first utterance seed {seed} (K2HIP_SOAK_SEED={seed} replays it)")
    utts = [synth_utterance(seed + b, 10.0) for b in range(32)]
    ex_op, ex_f, hidden, gaps, forms = _beam_two_levels(hip_large, oracle_large, utts, 4, f"configs[2] fresh audio, seed {seed}", 1e-3)
    print(f"fresh-audio beam test: operator level {ex_op}/32 exact, fused {ex_f}/32 exact, {len(hidden)} equal results over parted searches; "
          f"largest localised gap {max(gaps, default=0.0):.3g}; {len(forms)} stream(s) decided differently by the launch form: {forms}")
    assert (ex_op >= 30 and ex_f >= 29) or max(gaps, default=0.0) <= 2e-4, \
        f"operator level {ex_op}/32, fused {ex_f}/32 with a localised gap of {max(gaps, default=0.0):.3g}: misses this frequent at gaps this wide are not near-ties"
    for b, gf, gl in forms:
        assert max(g for g in (gf, gl, 0.0) if g is not None) <= 2e-4, \
            f"stream {b}: the engine's two search forms disagree where the oracle's candidates are {gf} / {gl} apart (not a near-tie)"


def test_search_exchange_timeout_is_retried_with_one_part(hip_large):
    """The vocabulary-parallel search (two column-slab workgroups per stream here) spins on its sibling with a bounded wait; on a
    shared GPU the siblings may not be resident together.  A timeout is not an error any more: the engine repeats the search
    with one workgroup per stream (no waits) on the same encoder output.  K2HIP_TEST_GREEDY_TIMEOUT makes every two-part search
    report a timeout: synchronous and pipelined entries must return exactly the normal result, and count one retry each."""
    import ctypes as C
    from k2transducerasr_amd import load_library, set_switch
    from k2transducerasr_amd.synth import synth_utterance
    L = load_library()
    L.k2hip_debug_search_retries.argtypes = [C.c_void_p, C.POINTER(C.c_int32)]

    def retries():
        n = C.c_int32(-1)
        assert L.k2hip_debug_search_retries(hip_large.handle, C.byref(n)) == 0
        return n.value

    B = 8
    s = np.stack([synth_utterance(500 + u, 4.0) for u in range(B)])
    ptr = hip_large.device_alloc(s.nbytes)
    try:
        hip_large.device_upload(ptr, s)
        want = hip_large.offline_greedy_from_samples_dev(ptr, s.shape[1], B)
        assert sum(len(t) for t, _ in want) > 0
        r0 = retries()
        set_switch("K2HIP_TEST_GREEDY_TIMEOUT", 1)
        try:
            got = hip_large.offline_greedy_from_samples_dev(ptr, s.shape[1], B)
            assert retries() == r0 + 1
            ta = hip_large.offline_submit_samples_dev(ptr, s.shape[1], B)
            tb = hip_large.offline_submit_samples_dev(ptr, s.shape[1], B)
            ga, gb = hip_large.offline_wait(ta), hip_large.offline_wait(tb)
            assert retries() == r0 + 3
        finally:
            set_switch("K2HIP_TEST_GREEDY_TIMEOUT", 0)
        assert got == want and ga == want and gb == want
        assert hip_large.offline_greedy_from_samples_dev(ptr, s.shape[1], B) == want and retries() == r0 + 3
    finally:
        hip_large.device_free(ptr)

def test_beam_search_two_slabs_one_slab_and_the_retry(large_path):
    """The one-kernel beam search runs two column slabs per stream on this vocabulary (V = 500: 2 x 256 columns) -- two workgroups that
    wait for each other's logits every frame.  K2HIP_BEAM_PARTS=1 keeps one workgroup per stream: identical results, scores bit for bit
    (the same sums in the same order).  And as for the greedy search a timeout of the exchange is not an error: the engine repeats the
    search with one slab (K2HIP_TEST_GREEDY_TIMEOUT reports one for every two-slab launch), synchronous and pipelined."""
    import ctypes as C
    from k2transducerasr_amd import Model, load_library, set_switch
    from k2transducerasr_amd.synth import synth_utterance
    L = load_library()
    L.k2hip_debug_search_retries.argtypes = [C.c_void_p, C.POINTER(C.c_int32)]
    m = Model(large_path, 0)
    m.set_decoding_method("modified_beam_search", 4)

    def retries():
        n = C.c_int32(-1)
        assert L.k2hip_debug_search_retries(m.handle, C.byref(n)) == 0
        return n.value

    B = 6
    s = np.stack([synth_utterance(700 + u, 4.0) for u in range(B)])
    ptr = m.device_alloc(s.nbytes)
    try:
        m.device_upload(ptr, s)
        want = m.offline_greedy_from_samples_dev(ptr, s.shape[1], B)
        wsc = m.last_scores(B).copy()
        assert sum(len(t) for t, _ in want) > 0
        set_switch("K2HIP_BEAM_PARTS", 1)
        try:
            assert m.offline_greedy_from_samples_dev(ptr, s.shape[1], B) == want
            assert np.array_equal(m.last_scores(B), wsc)
        finally:
            set_switch("K2HIP_BEAM_PARTS", 0)
        r0 = retries()
        set_switch("K2HIP_TEST_GREEDY_TIMEOUT", 1)
        try:
            got = m.offline_greedy_from_samples_dev(ptr, s.shape[1], B)
            assert retries() == r0 + 1
            ta = m.offline_submit_samples_dev(ptr, s.shape[1], B)
            tb = m.offline_submit_samples_dev(ptr, s.shape[1], B)
            ga, gb = m.offline_wait(ta), m.offline_wait(tb)
            assert retries() == r0 + 3
        finally:
            set_switch("K2HIP_TEST_GREEDY_TIMEOUT", 0)
        assert got == want and ga == want and gb == want
    finally:
        m.device_free(ptr)
