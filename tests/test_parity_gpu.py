"""GPU parity: libk2hip.so (through the C ABI) against the CPU oracle, same seeded inputs."""
import numpy as np
import pytest

from parity import ACT_TOL, LOGIT_TOL, assert_tokens_match

pytestmark = pytest.mark.gpu


def _feats(oracle, utts):
    return [oracle.fbank(u) for u in utts]


def test_library_is_native(hip_tiny):
    from k2transducerasr_amd import library_path
    import os
    assert os.path.exists(library_path())
    assert hip_tiny.vocab_size == 37 and hip_tiny.joiner_dim == 512


def test_fbank(hip_tiny, oracle_tiny, utts):
    for u in utts:
        a = hip_tiny.fbank(u)
        b = oracle_tiny.fbank(u)
        assert a.shape == b.shape == (1 + (u.size - 400) // 160, 80)
        np.testing.assert_allclose(a, b, atol=2e-5, rtol=0)


def test_fbank_edge_cases(hip_tiny, oracle_tiny):
    assert hip_tiny.fbank(np.zeros(399, np.float32)).shape == (0, 80)
    z = hip_tiny.fbank(np.zeros(400, np.float32))
    assert z.shape == (1, 80)
    np.testing.assert_allclose(z, np.log(np.float32(np.finfo(np.float32).eps)), atol=1e-6)
    np.testing.assert_allclose(z, oracle_tiny.fbank(np.zeros(400, np.float32)), atol=1e-6)


def test_pad_sequence_bit_exact(hip_tiny, oracle_tiny, utts):
    feats = _feats(oracle_tiny, utts)
    feats[1] = feats[1].copy()
    feats[1][3, 5] = 0.0  # a genuine zero feature is floored too (PadHelper.cs:58)
    a = hip_tiny.pad_sequence(feats)
    b = oracle_tiny.pad_sequence(feats)
    assert a.shape == b.shape
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    assert a[1, 3 * 80 + 5] == np.float32(-23.025850929940457)


@pytest.mark.parametrize("tap", [0, 1, 2, 3, 4, 100])
def test_encoder_taps(hip_tiny, oracle_tiny, utts, tap):
    x = oracle_tiny.pad_sequence(_feats(oracle_tiny, utts[:3]))
    x = x.reshape(3, -1, 80)
    a = hip_tiny.encoder_tap(x, tap)
    b = oracle_tiny.encoder_tap(x, tap)
    assert a.shape == b.shape
    np.testing.assert_allclose(a, b, atol=ACT_TOL, rtol=0)


def test_encoder_proj(hip_tiny, oracle_tiny, utts):
    x = oracle_tiny.pad_sequence(_feats(oracle_tiny, utts)).reshape(len(utts), -1, 80)
    a = hip_tiny.encoder_proj(x)
    b = oracle_tiny.encoder(x)
    assert a.shape == b.shape
    np.testing.assert_allclose(a, b, atol=ACT_TOL, rtol=0)


def test_encoder_batch_of_one_equals_batch_rows(hip_tiny, oracle_tiny, utts):
    # Q3: no length masking, so with equal lengths a row does not depend on its batch mates
    f = _feats(oracle_tiny, [utts[0], utts[3]])
    x = oracle_tiny.pad_sequence(f).reshape(2, -1, 80)
    both = hip_tiny.encoder_proj(x)
    one = hip_tiny.encoder_proj(x[1:2])
    # not bit-identical: a different row count may select a different GEMM tiling, i.e. another f32 summation order
    np.testing.assert_allclose(both[1], one[0], atol=1e-5, rtol=0)


def test_decoder_proj(hip_tiny, oracle_tiny):
    y = np.array([[-1, 0], [0, 0], [5, 7], [36, 1], [-1, -1], [3, 3]], np.int64)
    np.testing.assert_allclose(hip_tiny.decoder_proj(y), oracle_tiny.decoder(y), atol=2e-5, rtol=0)
    # DecoderProj(null, B) -> [-1, blank] rows (OfflineProjOfTransducer.cs:97-110)
    np.testing.assert_allclose(hip_tiny.decoder_proj(None, 3), oracle_tiny.decoder(np.array([[-1, 0]] * 3)), atol=2e-5, rtol=0)


def test_decoder_rejects_out_of_vocab(hip_tiny):
    from k2transducerasr_amd import K2HipError
    with pytest.raises(K2HipError):
        hip_tiny.decoder_proj(np.array([[0, 37]], np.int64))


def test_joiner_proj_logits(hip_tiny, oracle_tiny):
    rng = np.random.default_rng(1)
    enc = rng.standard_normal((70, 512)).astype(np.float32)
    dec = rng.standard_normal((70, 512)).astype(np.float32)
    np.testing.assert_allclose(hip_tiny.joiner_proj(enc, dec), oracle_tiny.joiner(enc, dec), atol=LOGIT_TOL / 10, rtol=0)


def test_greedy_batch_on_oracle_encoder_out(hip_tiny, oracle_tiny, utts):
    x = oracle_tiny.pad_sequence(_feats(oracle_tiny, utts)).reshape(len(utts), -1, 80)
    enc = oracle_tiny.encoder(x)
    want, mg = oracle_tiny.greedy_batch(enc, want_margins=True)
    got = hip_tiny.greedy_batch(enc)
    assert sum(len(w[0]) for w in want) > 0
    assert_tokens_match(got, want, mg, what="greedy_batch")


def test_greedy_single_on_oracle_encoder_out(hip_tiny, oracle_tiny, utts):
    x = oracle_tiny.pad_sequence(_feats(oracle_tiny, utts)).reshape(len(utts), -1, 80)
    enc = oracle_tiny.encoder(x)
    for b in range(len(utts)):
        want, mg = oracle_tiny.greedy_single(enc[b], want_margins=True)
        got = hip_tiny.greedy_single(enc[b])
        assert_tokens_match([got], [want], mg, what=f"greedy_single[{b}]")


def test_greedy_batch_cross_stream_context_switch(hip_tiny, oracle_tiny, utts):
    """OfflineRecognizer.cs:278-286: the first emission of ANY stream re-runs the decoder
    for all streams on [blank, blank]; a stream decoded alone must therefore be able to
    differ from the same stream decoded in a batch -- and the engine must follow suit."""
    x = oracle_tiny.pad_sequence(_feats(oracle_tiny, utts)).reshape(len(utts), -1, 80)
    enc = oracle_tiny.encoder(x)
    batch = hip_tiny.greedy_batch(enc)
    alone = [hip_tiny.greedy_batch(enc[b : b + 1])[0] for b in range(len(utts))]
    want_batch = oracle_tiny.greedy_batch(enc)
    want_alone = [oracle_tiny.greedy_batch(enc[b : b + 1])[0] for b in range(len(utts))]
    assert batch == want_batch
    assert alone == want_alone


def test_greedy_ties_pick_later_index(hip_tiny):
    """Q6: all-equal logits (enc+dec = 0 -> tanh 0 -> logits = bias) cannot tie here, so
    build the tie through the API that exposes argmax: identical encoder frames give
    identical tokens, and the joiner's argmax of two equal maxima is the later one."""
    enc = np.zeros((1, 4, 512), np.float32)
    a = hip_tiny.greedy_batch(enc)
    b = hip_tiny.greedy_batch(np.concatenate([enc, enc]))
    assert b[0] == a[0] and b[1] == a[0]


def test_fused_offline_greedy_matches_oracle(hip_tiny, oracle_tiny, utts):
    feats = _feats(oracle_tiny, utts)
    want = oracle_tiny.recognize_batch(feats)
    x = oracle_tiny.pad_sequence(feats).reshape(len(utts), -1, 80)
    _, mg = oracle_tiny.greedy_batch(oracle_tiny.encoder(x), want_margins=True)
    got = hip_tiny.offline_greedy(feats)
    assert_tokens_match(got, want, mg, what="offline_greedy")


def test_fused_single_matches_oracle(hip_tiny, oracle_tiny, utts):
    f = oracle_tiny.fbank(utts[2])
    x = oracle_tiny.pad_sequence([f]).reshape(1, -1, 80)
    want, mg = oracle_tiny.greedy_single(oracle_tiny.encoder(x)[0], want_margins=True)
    got = hip_tiny.offline_greedy_single(f)
    assert_tokens_match([got], [want], mg, what="offline_greedy_single")


def test_fused_from_samples_ragged(hip_tiny, oracle_tiny, utts):
    feats = _feats(oracle_tiny, utts)
    want = oracle_tiny.recognize_batch(feats)
    x = oracle_tiny.pad_sequence(feats).reshape(len(utts), -1, 80)
    _, mg = oracle_tiny.greedy_batch(oracle_tiny.encoder(x), want_margins=True)
    got = hip_tiny.offline_greedy_from_samples(utts)
    assert_tokens_match(got, want, mg, what="offline_greedy_from_samples")


def test_fused_from_device_samples_equal_length(hip_tiny, oracle_tiny):
    from k2transducerasr_amd.synth import synth_utterance
    B, n = 4, 16000
    s = np.stack([synth_utterance(100 + u, 1.0) for u in range(B)])
    feats = [oracle_tiny.fbank(s[b]) for b in range(B)]
    want = oracle_tiny.recognize_batch(feats)
    x = oracle_tiny.pad_sequence(feats).reshape(B, -1, 80)
    _, mg = oracle_tiny.greedy_batch(oracle_tiny.encoder(x), want_margins=True)
    ptr = hip_tiny.device_alloc(s.nbytes)
    try:
        hip_tiny.device_upload(ptr, s)
        got = hip_tiny.offline_greedy_from_samples_dev(ptr, n, B)
        t = hip_tiny.timing()
        assert t["total_ms"] > 0 and t["encoder_ms"] > 0
    finally:
        hip_tiny.device_free(ptr)
    assert_tokens_match(got, want, mg, what="from_samples_dev")


def test_stream_api_mirrors_reference_bookkeeping(tiny_model_path, oracle_tiny, utts):
    from k2transducerasr_amd import OfflineRecognizer
    rec = OfflineRecognizer(tiny_model_path)
    streams = [rec.create_offline_stream() for _ in utts]
    assert streams[0].tokens == [0, 0]  # OfflineStream.cs:34
    for s, u in zip(streams, utts):
        # two AddSamples calls: streaming fbank must give the same frames as one call
        s.add_samples(u[:5000])
        s.add_samples(u[5000:])
    feats = _feats(oracle_tiny, utts)
    for s, f in zip(streams, feats):
        assert s.speech_length == f.size
        np.testing.assert_allclose(s.speech.reshape(-1, 80), f, atol=2e-5, rtol=0)
    want = oracle_tiny.recognize_batch([s.speech for s in streams])
    res = rec.get_results(streams)
    B = len(utts)
    for (tok, ts), (wt, wts), s in zip(res, want, streams):
        assert tok == [0] * (2 * B) + wt          # OfflineRecognizer.cs:250-258
        assert ts == [0] * (2 * B) + wts          # :259-267
        assert s.speech_length == 0               # RemoveSamples :294
    # single-stream path: Tokens = [-1, blank, ...] (:115-117,:180), samples kept
    s = rec.create_offline_stream()
    s.add_samples(utts[2])
    f = oracle_tiny.fbank(utts[2])
    x = oracle_tiny.pad_sequence([f]).reshape(1, -1, 80)
    wt, wts = oracle_tiny.greedy_single(oracle_tiny.encoder(x)[0])
    tok, ts = rec.get_result(s)
    assert tok == [-1, 0] + wt and ts == wts
    assert s.speech_length == f.size


def test_sample_queues_grow_and_are_reused_across_streams(tiny_model_path, oracle_tiny):
    """The native OfflineStreams' sample queues are pinned host buffers that GetResults reads in place: (a) a queue that outgrows its
    buffer (a short first piece: 64 K samples of capacity, then a long second piece) is re-allocated with its contents; (b) the buffers
    of destroyed streams go back to the model's pool and are handed to later streams of OTHER lengths -- a short utterance in a long
    stream's buffer, a longer one than anything pooled.  Every round's tokens and timestamps equal the oracle's on the same audio."""
    from k2transducerasr_amd import OfflineRecognizer
    from k2transducerasr_amd.synth import synth_utterance
    rec = OfflineRecognizer(tiny_model_path)
    rounds = [[5.5, 0.9, 4.6], [0.7, 5.0], [6.5, 1.2, 0.8, 5.9]]      # seconds; 64 K samples = 4.1 s
    for r, secs in enumerate(rounds):
        # (lengths of every residue mod 4, the longest included: the dense block's rows are padded to 16 bytes whatever they are)
        us = [synth_utterance(4200 + 10 * r + k, s)[: int(s * 16000) - (k + r + 1) % 4] for k, s in enumerate(secs)]
        feats = [oracle_tiny.fbank(u) for u in us]
        want = oracle_tiny.recognize_batch(feats)
        streams = [rec.create_offline_stream() for _ in us]
        for s, u, f in zip(streams, us, feats):
            s.add_samples(u[:1000])          # takes a pooled buffer (or a fresh 64 K one) ...
            s.add_samples(u[1000:])          # ... which a long utterance outgrows here
            assert s.speech_length == f.size
        res = rec.get_results(streams)
        B = len(us)
        for (tok, ts), (wt, wts) in zip(res, want):
            assert tok == [0] * (2 * B) + wt and ts == [0] * (2 * B) + wts, r
        for s in streams:
            s.close()


def test_get_results_from_queued_samples_equals_the_feature_route(tiny_model_path, oracle_tiny, utts):
    """Round 5: AddSamples on a native OfflineStream only queues the raw samples (what the C# shim's fused route calls: no fbank per call,
    SpeechLength counts the queued samples' frames at once); GetResults then runs fbank + pad + encoder + search for the whole RAGGED
    batch on the device in one pass (k2hip_offline_recognizer_get_results -> Engine::offline_greedy_samples).  Tokens, timestamps and the
    stream bookkeeping must equal the oracle's / the feature route's: (a) nobody reads Speech before GetResults -- the batch goes samples ->
    tokens; (b) one stream's Speech is read first -- that batch is decoded from features materialised at GetResults; (c) a second
    utterance on the same streams (samples left over behind the last whole frame shift stay queued, as an OnlineFbank keeps them)."""
    from k2transducerasr_amd import OfflineRecognizer
    rec = OfflineRecognizer(tiny_model_path)
    B = len(utts)
    feats = _feats(oracle_tiny, utts)
    want = oracle_tiny.recognize_batch(feats)
    for read_one in (False, True):
        streams = [rec.create_offline_stream() for _ in utts]
        for k, (s, u) in enumerate(zip(streams, utts)):
            cut = 3001 + 517 * k                      # an odd split: the second piece starts inside a frame
            s.add_samples(u[:cut])
            s.add_samples(u[cut:])
            assert s.speech_length == feats[k].size   # OfflineStream.cs:55, before anything was computed
        if read_one:
            np.testing.assert_allclose(streams[1].speech.reshape(-1, 80), feats[1], atol=2e-5, rtol=0)
        res = rec.get_results(streams)
        for (tok, ts), (wt, wts), s in zip(res, want, streams):
            assert tok == [0] * (2 * B) + wt and ts == [0] * (2 * B) + wts
            assert s.speech_length == 0               # RemoveSamples :294
        if not read_one:
            # (c) the next utterance on the same streams: what is decoded is fbank([left-over samples ; new samples])
            tails = [u[(f.shape[0] * 160):] for u, f in zip(utts, feats)]
            nxt = [np.concatenate([t, u[::-1].copy()]) for t, u in zip(tails, utts)]
            f2 = [oracle_tiny.fbank(x) for x in nxt]
            for s, u in zip(streams, utts):
                s.add_samples(u[::-1].copy())
            for s, f in zip(streams, f2):
                assert s.speech_length == f.size
            want2 = oracle_tiny.recognize_batch(f2)
            res2 = rec.get_results(streams)
            for (tok, ts), (wt, wts), (_, ts1) in zip(res2, want2, res):
                assert tok == [0] * (2 * B) + wt
                assert ts == ts1 + [0] * (2 * B) + wts    # Timestamps.AddRange (:293): the list keeps growing across calls
        for s in streams:
            s.close()


def test_errors_are_reported_not_thrown(hip_tiny):
    from k2transducerasr_amd import K2HipError, Model
    with pytest.raises(K2HipError) as e:
        Model("/nonexistent/model.k2w")
    assert e.value.code == -2
    with pytest.raises(K2HipError):
        hip_tiny.encoder_proj(np.zeros((1, 8, 80), np.float32))  # too few frames


# ---- hand-derived greedy known-answer tests, through the HIP engine -----------------
@pytest.fixture(scope="module")
def kat_hip(tmp_path_factory):
    from k2transducerasr_amd import Model
    from kat_model import write_kat_model
    p = str(tmp_path_factory.mktemp("kat") / "kat.k2w")
    write_kat_model(p)
    return Model(p, 0)


def test_kat_decoder_by_hand(kat_hip):
    d = kat_hip.decoder_proj(np.array([[-1, 0], [0, 0], [0, 4], [4, 3], [-1, -1]], np.int64))
    np.testing.assert_allclose(d[:, 3], [0.05, 0.1, 0.45, 0.7, 0.0], rtol=1e-6)  # Q9: id < 0 -> zero embedding
    assert (np.delete(d, 3, axis=1) == 0).all()


def test_greedy_known_answers(kat_hip):
    from kat_model import CASES
    for name, case in sorted(CASES.items()):
        enc = np.stack(case["streams"])
        assert kat_hip.greedy_batch(enc) == case["batch"], name
        for b, s in enumerate(case["streams"]):
            assert kat_hip.greedy_single(s) == case["single"][b], name


def test_greedy_single_max_symbols(kat_hip):
    from kat_model import frames
    enc = frames([{5: 1.0}] * 1003)
    tok, ts = kat_hip.greedy_single(enc)
    assert len(tok) == 1000 and ts == list(range(1000))  # max_sym_per_utt (OfflineRecognizer.cs:122)
    assert len(kat_hip.greedy_batch(enc[None])[0][0]) == 1003


def test_hip_matches_committed_golden(hip_tiny):
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "tiny_golden.npz"))
    np.testing.assert_allclose(hip_tiny.fbank(g["samples"]), g["fbank"], atol=2e-4, rtol=0)
    np.testing.assert_allclose(hip_tiny.encoder_proj(g["x"]), g["encoder_out"], atol=ACT_TOL, rtol=0)
    np.testing.assert_allclose(hip_tiny.decoder_proj(g["y"]), g["decoder_out"], atol=2e-5, rtol=0)
    np.testing.assert_allclose(hip_tiny.joiner_proj(g["encoder_out"][0, :6], g["decoder_out"][:6]), g["logits"], atol=1e-4, rtol=0)
    want = [(g[f"tok{b}"].tolist(), g[f"ts{b}"].tolist()) for b in range(g["x"].shape[0])]
    assert hip_tiny.greedy_batch(g["encoder_out"]) == want


def test_two_handles_from_two_threads(tiny_model_path, oracle_tiny, utts):
    """INTEGRATION.md threading contract: different model handles may be driven concurrently from different host threads
    (a C# host opens one per GPU); each handle serialises its own calls."""
    import threading
    from k2transducerasr_amd import Model
    feats = [oracle_tiny.fbank(u) for u in utts]
    want = oracle_tiny.recognize_batch(feats)
    models = [Model(tiny_model_path, 0) for _ in range(2)]
    out = [None, None]

    def work(i):
        for _ in range(5):
            out[i] = models[i].offline_greedy(feats)

    ts = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert out[0] == want and out[1] == want
    # one handle from two threads is serialised by its mutex
    res = [None, None]

    def work2(i):
        res[i] = models[0].offline_greedy_from_samples(utts)

    ts = [threading.Thread(target=work2, args=(i,)) for i in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert res[0] == res[1] == want


def test_random_ragged_batches_match_oracle(hip_tiny, oracle_tiny):
    """Seeded sweep over batch sizes and ragged lengths (the reference pads every stream to the longest + 19 frames and decodes
    all rows for all frames, Q1-Q3): fused samples -> tokens on the GPU against the oracle, including 1-stream batches, the
    shortest inputs the encoder accepts and lengths around the frame / subsampling boundaries."""
    from k2transducerasr_amd.synth import synth_utterance
    rng = np.random.default_rng(2024)
    lens_pool = [400, 401, 559, 560, 1999, 3200, 4801, 8000, 12345, 16000, 20001]
    for case in range(14):
        B = int(rng.integers(1, 7))
        ns = [int(rng.choice(lens_pool)) if rng.random() < 0.6 else int(rng.integers(400, 24000)) for _ in range(B)]
        utts = [synth_utterance(1000 + 10 * case + b, n / 16000.0)[:n] for b, n in enumerate(ns)]
        feats = [oracle_tiny.fbank(u) for u in utts]
        want = oracle_tiny.recognize_batch(feats)
        x = oracle_tiny.pad_sequence(feats).reshape(B, -1, 80)
        _, mg = oracle_tiny.greedy_batch(oracle_tiny.encoder(x), want_margins=True)
        got = hip_tiny.offline_greedy_from_samples(utts)
        assert_tokens_match(got, want, mg, what=f"ragged case {case} (B={B}, samples={ns})")


def test_alternative_kernel_paths_agree(hip_tiny, oracle_tiny, utts):
    """The switches of INTEGRATION.md select other kernels for the same math: attention apply + out_proj as two GEMMs
    (K2HIP_NO_FUSED_AV) and the two-pass attention-scores kernel (K2HIP_ATTN_LONG; test_large_gpu covers it at lengths that need it).  Same input, both paths, against each other and against the oracle."""
    import os
    x = oracle_tiny.pad_sequence([oracle_tiny.fbank(u) for u in utts]).reshape(len(utts), -1, 80)
    want = oracle_tiny.encoder(x)
    fused = hip_tiny.encoder_proj(x)
    from k2transducerasr_amd import set_switch
    set_switch("K2HIP_NO_FUSED_AV", 1)
    try:
        plain = hip_tiny.encoder_proj(x)
    finally:
        set_switch("K2HIP_NO_FUSED_AV", 0)
    np.testing.assert_allclose(fused, want, atol=ACT_TOL, rtol=0)
    np.testing.assert_allclose(plain, want, atol=ACT_TOL, rtol=0)
    np.testing.assert_allclose(fused, plain, atol=ACT_TOL, rtol=0)
    set_switch("K2HIP_ATTN_LONG", 1)     # two-pass attention scores at a length the in-LDS strip would also take
    try:
        np.testing.assert_allclose(hip_tiny.encoder_proj(x), want, atol=ACT_TOL, rtol=0)
    finally:
        set_switch("K2HIP_ATTN_LONG", 0)
    from k2transducerasr_amd import K2HipError
    with pytest.raises(K2HipError):
        set_switch("K2HIP_NO_SUCH_SWITCH", 1)


def test_search_as_rounds_equals_persistent_kernel(hip_tiny, oracle_tiny, utts, kat_hip):
    """The multi-stream search has two forms: one persistent kernel (k_greedy, the offline default) and rounds of joiner GEMMs +
    a per-stream step (greedy_rounds, the streaming default); K2HIP_SEARCH_ROUNDS forces either.  Same tokens and timestamps
    from both, on the tiny model (ragged batch, with the first-emission context switch, and a batch where no stream ever
    emits) and on the hand-derived cases (tie-break, skip set, negative ids, the switch itself)."""
    from k2transducerasr_amd import set_switch
    from kat_model import CASES
    feats = [oracle_tiny.fbank(u) for u in utts]
    x = oracle_tiny.pad_sequence(feats).reshape(len(utts), -1, 80)
    enc = oracle_tiny.encoder(x)
    want = oracle_tiny.greedy_batch(enc)
    silent = np.zeros((2, 4, enc.shape[2]), np.float32)

    def run():
        return (hip_tiny.greedy_batch(enc), hip_tiny.greedy_batch(enc[1:2]),  # batch of one: no t0 pre-pass
                hip_tiny.greedy_batch(silent),
                {name: kat_hip.greedy_batch(np.stack(case["streams"])) for name, case in CASES.items()})
    try:
        set_switch("K2HIP_SEARCH_ROUNDS", 1)
        rounds = run()
        set_switch("K2HIP_SEARCH_ROUNDS", 0)
        persistent = run()
    finally:
        set_switch("K2HIP_SEARCH_ROUNDS", -1)
    assert rounds[0] == persistent[0] == want
    assert rounds[1:3] == persistent[1:3]
    for name, case in sorted(CASES.items()):
        assert rounds[3][name] == case["batch"] == persistent[3][name], name


def test_large_vocabulary_screened_search_end_to_end(tmp_path):
    """A vocabulary of 3000 behind the tiny encoder: the persistent search runs its large-vocabulary form (k_greedy<true>: per round an
    f16 screening sweep with load-time error bounds, then the exact f32 re-check of the columns that can win; greedy.hip screen_round)
    over several column slabs per stream.  Ragged batches from samples against the oracle, the same batches with the screen switched
    off (every round sweeps in f32), and one slab per stream (2 .. 3 passes of 1536 columns would not fit the screen's LDS area:
    that form keeps the f32 passes) -- all three must give the oracle's tokens."""
    from k2transducerasr_amd import Model, set_switch
    from k2transducerasr_amd.synth import synth_utterance, write_synthetic_model
    from oracle import Oracle
    p = str(tmp_path / "wide.k2w")
    write_synthetic_model(p, "zipformer2-tiny-test", blank_bias=2.4, meta_overrides={"vocab_size": "3000"})
    hip, ora = Model(p, 0), Oracle(p)
    rng = np.random.default_rng(77)
    emitted = frames = 0
    for case in range(6):
        B = int(rng.integers(1, 7))
        utts = [synth_utterance(900 + 8 * case + b, float(rng.uniform(0.4, 2.2))) for b in range(B)]
        feats = [ora.fbank(u) for u in utts]
        want = ora.recognize_batch(feats)
        _, mg = ora.greedy_batch(ora.encoder(ora.pad_sequence(feats).reshape(B, -1, 80)), want_margins=True)
        assert_tokens_match(hip.offline_greedy_from_samples(utts), want, mg, what=f"screened search, case {case}")
        emitted += sum(len(t) for t, _ in want)
        frames += mg.size
        for switch, value in (("K2HIP_SCREEN_MIN_V", 0), ("K2HIP_GREEDY_ONE_PART", 1)):
            set_switch(switch, value)
            try:
                assert_tokens_match(hip.offline_greedy_from_samples(utts), want, mg, what=f"{switch}={value}, case {case}")
            finally:
                set_switch(switch, 1024 if switch == "K2HIP_SCREEN_MIN_V" else 0)
    assert 0.03 * frames < emitted < 0.7 * frames, (emitted, frames)     # blank wins many frames and loses some: both branches of the loop run
    # an exchange timeout between the column slabs (forced: the flag is raised behind the search and its outputs are wiped): the
    # repeat with one workgroup per stream -- whose 3000 columns exceed the screen's LDS area, so it sweeps in f32 -- gives the tokens
    import ctypes as C
    from k2transducerasr_amd import load_library
    L = load_library()
    L.k2hip_debug_search_retries.argtypes = [C.c_void_p, C.POINTER(C.c_int32)]
    n0, n1 = C.c_int32(0), C.c_int32(0)
    assert L.k2hip_debug_search_retries(hip.handle, C.byref(n0)) == 0
    set_switch("K2HIP_TEST_GREEDY_TIMEOUT", 1)
    try:
        assert_tokens_match(hip.offline_greedy_from_samples(utts), want, mg, what="screened search, forced retry")
    finally:
        set_switch("K2HIP_TEST_GREEDY_TIMEOUT", 0)
    assert L.k2hip_debug_search_retries(hip.handle, C.byref(n1)) == 0
    assert n1.value == n0.value + 1 or B == 1, (n0.value, n1.value)      # (a one-stream batch may run one slab: nothing to time out)
    hip.close()


def test_search_timeouts_back_off_to_one_workgroup_per_stream(tmp_path):
    """A parted search whose slabs are not co-resident (other handles or processes hold the CUs) times out and is repeated with one
    workgroup per stream; paying that timeout on every call made four streaming recognizers on one GPU three times slower than one.
    After a timeout the engine's next 4 searches go out with one part straight away (no timeout to wait for), then a parted search is
    tried again; a further timeout doubles the span, a parted search that comes through clears it (round 5: the first span was 64 --
    one timeout while a process warmed up kept a whole short run on the slower form).  Forced here with the test hook in its "as a real
    one" form."""
    import ctypes as C
    from k2transducerasr_amd import Model, load_library, set_switch
    from k2transducerasr_amd.synth import synth_utterance, write_synthetic_model
    from oracle import Oracle
    p = str(tmp_path / "wide.k2w")
    write_synthetic_model(p, "zipformer2-tiny-test", blank_bias=2.4, meta_overrides={"vocab_size": "3000"})
    hip, ora = Model(p, 0), Oracle(p)
    L = load_library()
    L.k2hip_debug_search_retries.argtypes = [C.c_void_p, C.POINTER(C.c_int32)]

    def retries():
        n = C.c_int32(0)
        assert L.k2hip_debug_search_retries(hip.handle, C.byref(n)) == 0
        return n.value
    utts = [synth_utterance(1200 + b, 0.8 + 0.3 * b) for b in range(3)]
    want = ora.recognize_batch([ora.fbank(u) for u in utts])
    n0 = retries()
    assert hip.offline_greedy_from_samples(utts) == want and retries() == n0          # parted, undisturbed
    try:
        set_switch("K2HIP_TEST_GREEDY_TIMEOUT", 2)
        assert hip.offline_greedy_from_samples(utts) == want and retries() == n0 + 1  # timed out, repeated with one part
        for _ in range(4):                                                           # backed off: one part, nothing to time out
            assert hip.offline_greedy_from_samples(utts) == want
        assert retries() == n0 + 1
        assert hip.offline_greedy_from_samples(utts) == want and retries() == n0 + 2  # the 5th is parted again -- and times out again
        for _ in range(8):                                                           # the span doubled: one part for 8 searches
            assert hip.offline_greedy_from_samples(utts) == want
        assert retries() == n0 + 2
        set_switch("K2HIP_TEST_GREEDY_TIMEOUT", 0)
        assert hip.offline_greedy_from_samples(utts) == want and retries() == n0 + 2  # parted, comes through: the span is cleared
        set_switch("K2HIP_TEST_GREEDY_TIMEOUT", 2)
        assert hip.offline_greedy_from_samples(utts) == want and retries() == n0 + 3  # so the next timeout starts from 4 again
        for _ in range(4):
            assert hip.offline_greedy_from_samples(utts) == want
        assert retries() == n0 + 3
        assert hip.offline_greedy_from_samples(utts) == want and retries() == n0 + 4
    finally:
        set_switch("K2HIP_TEST_GREEDY_TIMEOUT", 0)
    hip.close()
