"""GPU parity of the streaming path (OnlineRecognizer / OnlineProjOfZipformer2 replacement):
libk2hip.so against the streaming CPU oracle, chunk after chunk: tokens, timestamps, Hyp, and every
cached state tensor in the stream's device slot."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

KINDS = ["key", "nonlin", "val1", "val2", "conv1", "conv2"]


@pytest.fixture(scope="module")
def stream_model_path(tmp_path_factory):
    from k2transducerasr_amd.synth import write_synthetic_model
    p = str(tmp_path_factory.mktemp("smodels") / "stiny.k2w")
    write_synthetic_model(p, "zipformer2-streaming-tiny-test")
    return p


@pytest.fixture(scope="module")
def rec(stream_model_path):
    from k2transducerasr_amd import OnlineRecognizer
    return OnlineRecognizer(stream_model_path)


@pytest.fixture(scope="module")
def ora(stream_model_path):
    from oracle.online import OnlineOracle
    return OnlineOracle(stream_model_path)


def test_chunk_info_and_init(rec, ora):
    assert (rec.chunk_length, rec.shift_length, rec.frames_per_chunk) == (ora.chunk_length, ora.shift_length, ora.frames_per_chunk) == (45, 32, 8)
    s = rec.create_online_stream()
    o = ora.create_stream()
    assert s.tokens == [0, 0] and s.hyp == [0, 0] and s.timestamps == []   # OnlineStream.cs:44-45
    for l in range(o.num_layers):
        for k in KINDS:
            a = s.state(l, k)
            assert a.size == o.state(l, k).size and not a.any()           # GetEncoderInitStates: zeros
    assert s.state(0, "embed").size == 128 * 3 * 19


def test_streaming_matches_oracle_chunk_by_chunk(rec, ora):
    from k2transducerasr_amd.synth import synth_utterance
    B = 3
    feats = [ora.fbank(synth_utterance(40 + u, d)) for u, d in enumerate([2.2, 1.6, 2.2])]
    hs = [rec.create_online_stream() for _ in range(B)]
    os_ = [ora.create_stream() for _ in range(B)]
    T, S = rec.chunk_length, rec.shift_length
    pos = [0] * B
    for h, f in zip(hs, feats):
        h.add_features(f)                      # whole utterance buffered; chunks are consumed one per step
    steps = 0
    while True:
        ready = [b for b in range(B) if pos[b] + T <= feats[b].shape[0]]
        dec, n_new = rec.get_results(hs)
        assert [b for b in range(B) if dec[b]] == ready      # streams without a full chunk are skipped (:101-120)
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
        if steps in (0, 2):
            for b in ready:
                for l in range(os_[b].num_layers):
                    for k in KINDS:
                        np.testing.assert_allclose(hs[b].state(l, k), os_[b].state(l, k), atol=2e-4, rtol=0, err_msg=f"step {steps} stream {b} layer {l} {k}")
                np.testing.assert_allclose(hs[b].state(0, "embed"), os_[b].state(0, "embed"), atol=2e-4, rtol=0)
        steps += 1
    assert steps >= 5
    assert sum(len(o.tokens) - 2 for o in os_) > 0


def test_fifo_add_samples_and_is_finished(rec, ora):
    """AddSamples in 800-sample pushes (K2TransducerAsr.Examples/OnlineRecognizer.cs:135-139) produces the
    same frames as one-shot fbank; IsFinished mirrors OnlineStream.cs:124-161."""
    from k2transducerasr_amd.synth import synth_utterance
    u = synth_utterance(50, 1.0)
    s = rec.create_online_stream()
    for i in range(0, u.size, 800):
        s.add_samples(u[i : i + 800])
    f = ora.fbank(u)
    assert s.speech_length == f.size
    assert s.is_finished(False) is False                      # not an endpoint -> false (:157-160)
    assert s.is_finished(True) is False                       # data pending, more than a chunk buffered: no side effect
    assert s.speech_length == f.size
    n0 = 0
    while rec.get_results([s])[0][0]:
        n0 += 1
    assert n0 == (f.shape[0] - rec.chunk_length) // rec.shift_length + 1
    left = s.speech_length
    assert 0 < left <= rec.chunk_length * 80
    assert s.is_finished(True) is False                       # <= one chunk buffered: feeds 400 zero samples (:144-147)
    grown = s.speech_length - left
    assert grown > 0 and grown % 80 == 0                      # the 400 zero samples completed whole frames
    e = rec.create_online_stream()
    assert e.is_finished(True) is True                        # empty FIFO -> finished (:152-155)
    z = rec.create_online_stream()
    z.add_features(np.full((3, 80), 1.5, np.float32))
    assert z.is_finished(True) is True                        # all elements equal their average -> finished (:136-141)


def test_slots_are_recycled(rec):
    a = rec.create_online_stream()
    a.add_features(np.random.default_rng(0).standard_normal((45, 80)).astype(np.float32))
    rec.get_results([a])
    assert a.state(0, "key").any()
    a.close()
    b = rec.create_online_stream()                            # may reuse a's slot: must start from zeros
    assert not b.state(0, "key").any() and not b.state(0, "embed").any()


def test_streaming_zh_model_matches_oracle(tmp_path_factory):
    """The benchmark's streaming architecture (BASELINE configs[3]: 16 layers, left context 128, vocab 2000),
    random weights, two streams, a few chunks: tokens exact, states and nothing else drift."""
    from k2transducerasr_amd import OnlineRecognizer
    from k2transducerasr_amd.synth import synth_utterance, write_synthetic_model
    from oracle.online import OnlineOracle
    p = str(tmp_path_factory.mktemp("szh") / "szh.k2w")
    write_synthetic_model(p, "zipformer2-streaming-zh")
    rec, ora = OnlineRecognizer(p), OnlineOracle(p)
    feats = [ora.fbank(synth_utterance(60 + u, 1.7)) for u in range(2)]
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
        for l in (0, 7, 15):
            for kind in KINDS:
                np.testing.assert_allclose(h.state(l, kind), o.state(l, kind), atol=5e-4, rtol=0)


@pytest.mark.parametrize("preset", ["zipformer2-streaming-tiny-test", "zipformer-streaming-tiny-test", "conformer-streaming-tiny-test"])
def test_streaming_random_churn(tmp_path_factory, preset):
    """Streams that arrive at different times, are fed in uneven pushes, finish and are replaced (slots recycled), so every
    GetResults call sees a different ready subset at different positions: tokens / timestamps / Hyp of every stream must equal
    the oracle's, which decodes each stream on its own.  (The streaming conformer is left out of the mixed-batch comparison of
    processed_lens: the reference overwrites it with the batch size, OnlineProjOfConformer.cs:229, so there the oracle is stepped
    with exactly the same ready subsets.)"""
    from k2transducerasr_amd import OnlineRecognizer
    from k2transducerasr_amd.synth import synth_utterance, write_synthetic_model
    from oracle.online import OnlineOracle
    p = str(tmp_path_factory.mktemp("churn") / f"{preset}.k2w")
    write_synthetic_model(p, preset)
    rec, ora = OnlineRecognizer(p), OnlineOracle(p)
    T, S = rec.chunk_length, rec.shift_length
    rng = np.random.default_rng(20240607)
    NSLOT, TOTAL = 6, 14
    live = []          # dicts: h (hip stream), o (oracle stream), feats, fed (frames handed over), pos (frames consumed)
    started = finished = 0
    checked_tokens = 0
    for it in range(400):
        while len(live) < NSLOT and started < TOTAL and rng.random() < 0.5:
            f = ora.fbank(synth_utterance(300 + started, float(rng.uniform(0.6, 1.8))))
            live.append(dict(h=rec.create_online_stream(), o=ora.create_stream(), feats=f, fed=0, pos=0))
            started += 1
        for s in live:   # uneven pushes of whole frames
            if s["fed"] < s["feats"].shape[0] and rng.random() < 0.8:
                n = int(rng.integers(1, 40))
                s["h"].add_features(s["feats"][s["fed"] : s["fed"] + n])
                s["fed"] = min(s["fed"] + n, s["feats"].shape[0])
        if not live:
            if started == TOTAL:
                break
            continue
        ready = [i for i, s in enumerate(live) if s["pos"] + T <= s["fed"]]
        dec, n_new = rec.get_results([s["h"] for s in live])
        assert [i for i in range(len(live)) if dec[i]] == ready, it
        if ready:
            want = ora.step([live[i]["o"] for i in ready], [live[i]["feats"][live[i]["pos"] : live[i]["pos"] + T] for i in ready])
            for i, wn in zip(ready, want):
                assert n_new[i] == wn
                live[i]["pos"] += S
        for s in live:
            assert s["h"].tokens == s["o"].tokens and s["h"].timestamps == s["o"].timestamps and s["h"].hyp == s["o"].hyp, it
        keep = []
        for s in live:
            if s["fed"] == s["feats"].shape[0] and s["pos"] + T > s["fed"]:
                checked_tokens += len(s["o"].tokens) - 2
                s["h"].close()
                finished += 1
            else:
                keep.append(s)
        live = keep
    assert finished == TOTAL and checked_tokens > 0


def test_left_context_shorter_than_chunk(tmp_path_factory):
    """left_context_len 8 / 4 / 2 / 4 against chunks of 16 / 8 / 4 / 8 frames: every cache is refilled entirely from the newest rows
    of each chunk (the concat + roll kernel's chains are one element long and read only new rows)."""
    from k2transducerasr_amd import OnlineRecognizer
    from k2transducerasr_amd.synth import synth_utterance, write_synthetic_model
    from oracle.online import OnlineOracle
    p = str(tmp_path_factory.mktemp("shortleft") / "m.k2w")
    write_synthetic_model(p, "zipformer2-streaming-tiny-test", meta_overrides={"left_context_len": "8,4,2,4"})
    rec, ora = OnlineRecognizer(p), OnlineOracle(p)
    feats = [ora.fbank(synth_utterance(90 + u, 1.4)) for u in range(2)]
    hs = [rec.create_online_stream() for _ in feats]
    os_ = [ora.create_stream() for _ in feats]
    for h, f in zip(hs, feats):
        h.add_features(f)
    T, S = rec.chunk_length, rec.shift_length
    for k in range((feats[0].shape[0] - T) // S + 1):
        rec.get_results(hs)
        ora.step(os_, [f[k * S : k * S + T] for f in feats])
        for h, o in zip(hs, os_):
            assert h.tokens == o.tokens and h.timestamps == o.timestamps
    for h, o in zip(hs, os_):
        for l in range(o.num_layers):
            for kind in KINDS:
                np.testing.assert_allclose(h.state(l, kind), o.state(l, kind), atol=2e-4, rtol=0, err_msg=f"layer {l} {kind}")


def test_left_context_256_and_longer(tmp_path_factory):
    """left_context_len 256 on the first stack: keys = 256 + 16 = 272 per chunk, more than the 256 the fused NonlinAttention kernel
    (k_nonlin_av_out) takes in one trip of loads -- an exported model with such a left context must decode (trips of 4 x 64 keys),
    not be refused on its first chunk after earlier layers have already moved their caches.  The other stacks follow at their rates (128 / 64 / 128)."""
    from k2transducerasr_amd import OnlineRecognizer
    from k2transducerasr_amd.synth import synth_utterance, write_synthetic_model
    from oracle.online import OnlineOracle
    p = str(tmp_path_factory.mktemp("longleft") / "m.k2w")
    write_synthetic_model(p, "zipformer2-streaming-tiny-test", meta_overrides={"left_context_len": "256,128,64,128"})
    rec, ora = OnlineRecognizer(p), OnlineOracle(p)
    feats = [ora.fbank(synth_utterance(95 + u, 1.4)) for u in range(2)]
    hs = [rec.create_online_stream() for _ in feats]
    os_ = [ora.create_stream() for _ in feats]
    for h, f in zip(hs, feats):
        h.add_features(f)
    T, S = rec.chunk_length, rec.shift_length
    for k in range((feats[0].shape[0] - T) // S + 1):
        rec.get_results(hs)
        ora.step(os_, [f[k * S : k * S + T] for f in feats])
        for h, o in zip(hs, os_):
            assert h.tokens == o.tokens and h.timestamps == o.timestamps
    for h, o in zip(hs, os_):
        for l in range(o.num_layers):
            for kind in KINDS:
                np.testing.assert_allclose(h.state(l, kind), o.state(l, kind), atol=2e-4, rtol=0, err_msg=f"layer {l} {kind}")


def test_streaming_128_concurrent_slots_zh(tmp_path_factory):
    """BASELINE configs[3] at its own size: 128 concurrent streams of the zipformer-multi-zh-hans streaming architecture in one
    GetResults call per chunk (pool growth past its first allocation, 128-row launches, every slot live).  Streams 0 and 77 are
    checked against the oracle chunk by chunk; all 128 are checked for stream independence: five of them (first, last, and three
    in between, including an oracle-checked one) are decoded again ALONE on a fresh recognizer and must give the same tokens,
    timestamps and cached states; the rest repeat 8 distinct utterances, and copies of one utterance sitting in different slots
    of the same 128-row launches must agree with each other exactly."""
    from k2transducerasr_amd import OnlineRecognizer
    from k2transducerasr_amd.synth import synth_utterance, write_synthetic_model
    from oracle.online import OnlineOracle
    p = str(tmp_path_factory.mktemp("szh128") / "szh.k2w")
    write_synthetic_model(p, "zipformer2-streaming-zh")
    rec, ora = OnlineRecognizer(p), OnlineOracle(p)
    N, DISTINCT = 128, 8
    T, S = rec.chunk_length, rec.shift_length
    base = [ora.fbank(synth_utterance(800 + u, 1.7)) for u in range(DISTINCT)]
    nchunks = (base[0].shape[0] - T) // S + 1
    assert nchunks >= 4
    feats = [base[u % DISTINCT] for u in range(N)]
    hs = [rec.create_online_stream() for _ in range(N)]
    for h, f in zip(hs, feats):
        h.add_features(f)
    checked = {0: ora.create_stream(), 77: ora.create_stream()}
    group = rec.batch(hs)   # the handle array a native host would hold (no per-call scan of the 128 stream objects)
    for k in range(nchunks):
        dec, n_new = rec.get_results(group if k % 2 == 0 else hs)   # both ways of naming the same streams
        assert dec == [1] * N
        for u, o in checked.items():
            want = ora.step([o], [feats[u][k * S : k * S + T]])
            assert n_new[u] == want[0]
            assert hs[u].tokens == o.tokens and hs[u].timestamps == o.timestamps and hs[u].hyp == o.hyp, (k, u)
    assert rec.get_results(hs)[0] == [0] * N                  # nothing left to decode
    assert sum(len(h.tokens) - 2 for h in hs[:DISTINCT]) > 0
    for u, o in checked.items():
        for l in (0, 8, 15):
            for kind in KINDS:
                np.testing.assert_allclose(hs[u].state(l, kind), o.state(l, kind), atol=5e-4, rtol=0)
    # copies of one utterance in different slots of the same launches: bit-identical results
    for u in range(DISTINCT, N):
        r = u % DISTINCT
        assert hs[u].tokens == hs[r].tokens and hs[u].timestamps == hs[r].timestamps, u
    for u in (DISTINCT + 3, 127):
        for kind in KINDS:
            assert np.array_equal(hs[u].state(15, kind), hs[u % DISTINCT].state(15, kind))
    # a stream decoded alone (batch of one, fresh pool) gives what it gave among 127 others
    rec1 = OnlineRecognizer(p)
    for u in (0, 3, 5, 6, 7):
        a = rec1.create_online_stream()
        a.add_features(feats[u])
        for _ in range(nchunks):
            assert rec1.get_results([a])[0] == [1]
        assert a.tokens == hs[u].tokens and a.timestamps == hs[u].timestamps and a.hyp == hs[u].hyp, u
        for kind in KINDS:
            np.testing.assert_allclose(a.state(15, kind), hs[u].state(15, kind), atol=1e-5, rtol=0)
        a.close()


def test_streaming_128_slots_16_streams_against_oracle_zh(tmp_path_factory):
    """BASELINE configs[3] at its own size, more of it held to the oracle: 128 concurrent streams of the zh streaming architecture,
    16 DISTINCT utterances (slots 0..15) each followed chunk by chunk by its own oracle stream over 7 chunks -- tokens, timestamps,
    Hyp after every tick, every cache of three layers at the end -- while the other 112 slots carry copies that must agree exactly
    with their originals (OnlineRecognizer.cs:85-219, one GetResults call per tick)."""
    from k2transducerasr_amd import OnlineRecognizer
    from k2transducerasr_amd.synth import synth_utterance, write_synthetic_model
    from oracle.online import OnlineOracle
    p = str(tmp_path_factory.mktemp("szh128b") / "szh.k2w")
    write_synthetic_model(p, "zipformer2-streaming-zh")
    rec, ora = OnlineRecognizer(p), OnlineOracle(p)
    N, DISTINCT = 128, 16
    T, S = rec.chunk_length, rec.shift_length
    base = [ora.fbank(synth_utterance(900 + u, 2.5)) for u in range(DISTINCT)]
    nchunks = (base[0].shape[0] - T) // S + 1
    assert nchunks >= 6
    feats = [base[u % DISTINCT] for u in range(N)]
    hs = [rec.create_online_stream() for _ in range(N)]
    for h, f in zip(hs, feats):
        h.add_features(f)
    os_ = [ora.create_stream() for _ in range(DISTINCT)]
    group = rec.batch(hs)
    for k in range(nchunks):
        dec, n_new = rec.get_results(group)
        assert dec == [1] * N
        want = ora.step(os_, [base[u][k * S : k * S + T] for u in range(DISTINCT)])
        for u, o in enumerate(os_):
            assert n_new[u] == want[u], (k, u)
            assert hs[u].tokens == o.tokens and hs[u].timestamps == o.timestamps and hs[u].hyp == o.hyp, (k, u)
    import parity
    parity.COMPARED[0] += DISTINCT
    assert sum(len(h.tokens) - 2 for h in hs[:DISTINCT]) > DISTINCT
    for u in (0, 7, 15):
        for l in (0, 8, 15):
            for kind in KINDS:
                np.testing.assert_allclose(hs[u].state(l, kind), os_[u].state(l, kind), atol=5e-4, rtol=0)
    for u in range(DISTINCT, N):
        r = u % DISTINCT
        assert hs[u].tokens == hs[r].tokens and hs[u].timestamps == hs[r].timestamps, u


@pytest.mark.parametrize("vocab", [0, 3000])
def test_streaming_search_forms_agree(stream_model_path, ora, tmp_path, vocab):
    """The tick's search over the ready streams runs as rounds of joiner GEMMs (greedy_rounds; K2HIP_SEARCH_ROUNDS=1, the default for
    a vocabulary the f16 screen does not cover) or as one persistent kernel (=0; the default where its rounds go through the screen:
    the 3000-token case).  Both carry each stream's Hyp context across
    chunks; tokens, timestamps and Hyp must be the oracle's after every call from either form, and from the default's own choice."""
    from k2transducerasr_amd import OnlineRecognizer, set_switch
    from k2transducerasr_amd.synth import synth_utterance
    rc = None
    if vocab:
        from k2transducerasr_amd.synth import write_synthetic_model
        from oracle.online import OnlineOracle
        stream_model_path = str(tmp_path / "swide.k2w")
        write_synthetic_model(stream_model_path, "zipformer2-streaming-tiny-test", blank_bias=2.4, meta_overrides={"vocab_size": str(vocab)})
        ora = OnlineOracle(stream_model_path)
        rc = OnlineRecognizer(stream_model_path)
    ra, rb = OnlineRecognizer(stream_model_path), OnlineRecognizer(stream_model_path)
    N = 5
    feats = [ora.fbank(synth_utterance(700 + u, 1.0 + 0.3 * (u % 3))) for u in range(N)]
    sa = [ra.create_online_stream() for _ in range(N)]
    sb = [rb.create_online_stream() for _ in range(N)]
    sc = [rc.create_online_stream() for _ in range(N)] if rc else []
    so = [ora.create_stream() for _ in range(N)]
    for u, f in enumerate(feats):
        for ss in (sa, sb, sc):
            if ss:
                ss[u].add_features(f)
    T, S = ra.chunk_length, ra.shift_length
    pos = [0] * N
    calls = 0
    try:
        while True:
            ready = [u for u in range(N) if pos[u] + T <= feats[u].shape[0]]
            set_switch("K2HIP_SEARCH_ROUNDS", 1)
            ra.get_results(sa)
            set_switch("K2HIP_SEARCH_ROUNDS", 0)
            rb.get_results(sb)
            set_switch("K2HIP_SEARCH_ROUNDS", -1)
            if rc:
                rc.get_results(sc)
                for u in range(N):
                    assert sc[u].tokens == sa[u].tokens and sc[u].timestamps == sa[u].timestamps and sc[u].hyp == sa[u].hyp, (calls, u)
            if not ready:
                break
            ora.step([so[u] for u in ready], [feats[u][pos[u] : pos[u] + T] for u in ready])
            for u in ready:
                pos[u] += S
            for u in range(N):
                assert sa[u].tokens == sb[u].tokens == so[u].tokens, (calls, u)
                assert sa[u].timestamps == sb[u].timestamps == so[u].timestamps and sa[u].hyp == sb[u].hyp == so[u].hyp
            calls += 1
    finally:
        set_switch("K2HIP_SEARCH_ROUNDS", -1)
    assert calls >= 3 and sum(len(s.tokens) - 2 for s in so) > 0


def test_operator_level_online_proj_runs_the_reference_loop(stream_model_path, ora):
    """IOnlineProj as an operator (IOnlineProj.cs:65-71; csharp/OnlineProjOfHip.cs): the host keeps the reference's own
    ForwardBatchGreedySearch loop (OnlineRecognizer.cs:85-219, transcribed below: EncoderProj over the ready streams, DecoderProj on
    their Hyps, T' JoinerProj steps with the host-side argmax and skip set {0, 2, 1}) and only the three operators run on the GPU.
    Tokens, timestamps and Hyp must be the oracle's after every call; the states are handles and advance in place."""
    from k2transducerasr_amd import OnlineProj
    from k2transducerasr_amd.synth import synth_utterance
    proj = OnlineProj(stream_model_path)
    N = 4
    feats = [ora.fbank(synth_utterance(900 + u, 1.0 + 0.35 * (u % 3))) for u in range(N)]
    so = [ora.create_stream() for _ in range(N)]
    T, S, Tp = proj.chunk_length, proj.shift_length, proj.frames_per_chunk
    states = [proj.get_encoder_init_states() for _ in range(N)]
    hyp = [[0, 0] for _ in range(N)]
    tokens = [[0, 0] for _ in range(N)]
    stamps = [[] for _ in range(N)]
    pos = [0] * N
    calls = 0
    try:
        while True:
            ready = [u for u in range(N) if pos[u] + T <= feats[u].shape[0]]
            if not ready:
                break
            x = np.stack([feats[u][pos[u] : pos[u] + T] for u in ready])
            enc = proj.encoder_proj(x, [states[u] for u in ready])                      # :121-131
            dec = proj.decoder_proj(np.array([hyp[u] for u in ready], np.int64))        # :135-147
            for t in range(Tp):                                                         # :149-202
                logits = proj.joiner_proj(enc[:, t], dec)
                emitted = False
                for r, u in enumerate(ready):
                    row = logits[r]
                    y = int(len(row) - 1 - np.argmax(row[::-1]))                        # later index wins ties (`>=` scan)
                    if y not in (0, 2, 1):
                        tokens[u].append(y)
                        stamps[u].append(t)
                        emitted = True
                if emitted:
                    dec = proj.decoder_proj(np.array([tokens[u][-2:] for u in ready], np.int64))
            for u in ready:
                hyp[u] = tokens[u][-2:]                                                # :208
                pos[u] += S
            ora.step([so[u] for u in ready], [feats[u][pos[u] - S : pos[u] - S + T] for u in ready])
            for u in range(N):
                assert tokens[u] == so[u].tokens and stamps[u] == so[u].timestamps and hyp[u] == list(so[u].hyp), (calls, u)
            calls += 1
        assert calls >= 3 and sum(len(t) - 2 for t in tokens) > 0
        assert proj.processed_len(states[0]) == so[0].processed_len
        with pytest.raises(Exception):
            proj.encoder_proj(np.zeros((2, T, 80), np.float32), [states[0], states[0]])   # one state twice in a batch
    finally:
        for st in states:
            proj.free_states(st)


def test_feature_fifo_device_mirror_wraps_overflows_and_recovers(rec, ora):
    """The feature FIFO of a stream is mirrored in a 512-frame ring on the device so that a chunk step gathers its input there
    (no host copy of the features on the critical path).  Three streams in one batch: one whose FIFO stays small but whose ring
    position wraps (fed 40 frames at a time, ~1100 frames in total), one that is handed 700 frames at once (outgrows the ring:
    the step falls back to the host copy for the whole batch until that FIFO fits the ring again -- 700 - 512 frames later, not
    when it is empty, which a FIFO that is decoded chunk by chunk never is -- then what is left is re-uploaded and the mirror is
    valid again), and one
    fed with samples (frames produced on the device go into the ring directly).  Tokens / timestamps / Hyp against the oracle
    after every call."""
    from k2transducerasr_amd.synth import synth_utterance
    T, S = rec.chunk_length, rec.shift_length
    wav = [synth_utterance(950 + u, d) for u, d in enumerate([11.0, 7.0, 6.0])]
    feats = [ora.fbank(w) for w in wav]
    hs = [rec.create_online_stream() for _ in range(3)]
    so = [ora.create_stream() for _ in range(3)]
    fed = [0, 0, 0]          # frames handed to the HIP stream so far (stream 2: samples, counted in frames of 160)
    pos = [0, 0, 0]
    hs[1].add_features(feats[1][:700])   # outgrows the ring at once
    fed[1] = 700
    import ctypes as C
    from k2transducerasr_amd import load_library
    L = load_library()
    L.k2hip_debug_stream_mirrored.argtypes = [C.c_void_p, C.POINTER(C.c_int32)]

    def mirrored(h):
        ok = C.c_int32(-1)
        assert L.k2hip_debug_stream_mirrored(h._h, C.byref(ok)) == 0
        return ok.value

    assert mirrored(hs[1]) == 0 and mirrored(hs[0]) == 1
    recovered_at = None
    calls = 0
    while True:
        if fed[0] < feats[0].shape[0]:   # small pieces: the ring position wraps twice over the utterance
            n = min(40, feats[0].shape[0] - fed[0])
            hs[0].add_features(feats[0][fed[0] : fed[0] + n])
            fed[0] += n
        if fed[1] < feats[1].shape[0] and pos[1] + T > fed[1]:   # the rest only after the big block has drained
            hs[1].add_features(feats[1][fed[1] :])
            fed[1] = feats[1].shape[0]
        if fed[2] * 160 < wav[2].size:
            n = min(6400, wav[2].size - fed[2] * 160)
            hs[2].add_samples(wav[2][fed[2] * 160 : fed[2] * 160 + n])
            fed[2] += n // 160
        avail = [fed[0], fed[1], (fed[2] * 160 - 400) // 160 + 1 if fed[2] * 160 >= 400 else 0]
        ready = [u for u in range(3) if pos[u] + T <= min(avail[u], feats[u].shape[0])]
        dec, _ = rec.get_results(hs)
        assert [u for u in range(3) if dec[u]] == ready, (calls, avail, pos)
        if not ready and fed[0] >= feats[0].shape[0] and fed[2] * 160 >= wav[2].size:
            break
        if ready:
            ora.step([so[u] for u in ready], [feats[u][pos[u] : pos[u] + T] for u in ready])
            for u in ready:
                pos[u] += S
        for u in range(3):
            assert hs[u].tokens == so[u].tokens and hs[u].timestamps == so[u].timestamps and hs[u].hyp == so[u].hyp, (calls, u)
        if recovered_at is None and mirrored(hs[1]) == 1:
            recovered_at = pos[1]
        calls += 1
    assert calls >= 30 and pos[0] > 512 and sum(len(s.tokens) - 2 for s in so) > 0
    # the big block was back on the device path as soon as it fitted: after ceil((700 - 512) / 32) = 6 chunks, long before it drained
    assert recovered_at == 6 * S, recovered_at


def test_stream_reset_equals_a_fresh_stream(rec, ora):
    """k2hip_online_stream_reset: the same object decodes a second utterance exactly as a new stream would (caches re-zeroed in
    its slot, FIFO / remainder / tokens / timestamps / Hyp / processed_lens cleared), while another stream of the batch keeps going."""
    from k2transducerasr_amd.synth import synth_utterance
    T, S = rec.chunk_length, rec.shift_length
    f1, f2 = ora.fbank(synth_utterance(970, 1.6)), ora.fbank(synth_utterance(971, 1.4))
    a, other = rec.create_online_stream(), rec.create_online_stream()
    o_other = ora.create_stream()
    f_other = ora.fbank(synth_utterance(972, 3.4))
    other.add_features(f_other)
    pos_other = 0

    def decode_all(stream, feats):
        nonlocal pos_other
        o = ora.create_stream()
        stream.add_features(feats)
        pos = 0
        while pos + T <= feats.shape[0]:
            both = pos_other + T <= f_other.shape[0]
            rec.get_results([stream, other])
            ora.step([o] + ([o_other] if both else []), [feats[pos : pos + T]] + ([f_other[pos_other : pos_other + T]] if both else []))
            pos += S
            pos_other += S if both else 0
            assert stream.tokens == o.tokens and stream.timestamps == o.timestamps and stream.hyp == o.hyp
            assert other.tokens == o_other.tokens and other.hyp == o_other.hyp
        return o

    o1 = decode_all(a, f1)
    a.add_samples(np.zeros(123, np.float32))   # something left in the sample remainder too
    a.reset()
    assert a.tokens == [0, 0] and a.timestamps == [] and a.hyp == [0, 0] and a.processed_len == 0 and a.speech_length == 0
    o2 = decode_all(a, f2)
    assert len(o1.tokens) + len(o2.tokens) > 4
    for l in (0, 1):
        for k in KINDS:
            np.testing.assert_allclose(a.state(l, k), o2.state(l, k), atol=2e-4, rtol=0)


def test_failed_step_poisons_its_streams_until_reset(rec, ora):
    """A chunk step that fails on the device may already have advanced the conv / embed caches of its streams in place; feeding the
    same chunk again would corrupt the transcript silently.  Such streams refuse further steps until k2hip_online_stream_reset
    (IOnlineProj.cs:65-71 has no such failure mode because ONNXRuntime copies every state in and out)."""
    import ctypes as C
    from k2transducerasr_amd import K2HipError, load_library
    from k2transducerasr_amd.synth import synth_utterance
    L = load_library()
    L.k2hip_debug_poison_stream.argtypes = [C.c_void_p]
    f = ora.fbank(synth_utterance(31, 1.2))
    a, b = rec.create_online_stream(), rec.create_online_stream()
    a.add_features(f)
    b.add_features(f)
    assert rec.get_results([a, b])[0] == [1, 1]
    assert L.k2hip_debug_poison_stream(a._h) == 0
    tok_b = list(b.tokens)
    with pytest.raises(K2HipError, match="reset it first"):
        rec.get_results([a, b])
    assert b.tokens == tok_b                      # the refused call moved nothing
    assert rec.get_results([b])[0] == [1]         # the healthy stream goes on alone
    a.reset()
    a.add_features(f)
    o = ora.create_stream()
    T, S = rec.chunk_length, rec.shift_length
    for k in range((f.shape[0] - T) // S + 1):
        assert rec.get_results([a])[0] == [1]
        ora.step([o], [f[k * S : k * S + T]])
        assert a.tokens == o.tokens and a.timestamps == o.timestamps


def test_value_projection_inside_the_attention_kernel_equals_the_gemm_form(stream_model_path, ora):
    """The streaming self-attention modules run their value projection inside k_attn_av_out (one launch per module: projection of the
    chunk's rows, ring update, attention apply, out_proj, residual; input and output in different buffers) where a stream's chunk has at
    least K2HIP_FUSED_VPROJ_MIN_T rows, and as a GEMM launch of their own below that / with K2HIP_NO_FUSED_VPROJ.  Three recognizers -- every
    stack fused, none, the default -- step the same streams: tokens, Hyp and every value cache must follow the oracle in all three."""
    from k2transducerasr_amd import OnlineRecognizer, set_switch
    from k2transducerasr_amd.synth import synth_utterance
    recs = [OnlineRecognizer(stream_model_path) for _ in range(3)]
    forms = [("K2HIP_FUSED_VPROJ_MIN_T", 1, 4), ("K2HIP_NO_FUSED_VPROJ", 1, 0), (None, 0, 0)]
    N = 4
    feats = [ora.fbank(synth_utterance(820 + u, 1.2 + 0.25 * u)) for u in range(N)]
    hs = [[r.create_online_stream() for _ in range(N)] for r in recs]
    so = [ora.create_stream() for _ in range(N)]
    for u, f in enumerate(feats):
        for h in hs:
            h[u].add_features(f)
    T, S = recs[0].chunk_length, recs[0].shift_length
    pos = [0] * N
    calls = 0
    while True:
        ready = [u for u in range(N) if pos[u] + T <= feats[u].shape[0]]
        for r, h, (sw, on, off) in zip(recs, hs, forms):
            if sw:
                set_switch(sw, on)
            try:
                r.get_results(h)
            finally:
                if sw:
                    set_switch(sw, off)
        if not ready:
            break
        ora.step([so[u] for u in ready], [feats[u][pos[u] : pos[u] + T] for u in ready])
        for u in ready:
            pos[u] += S
        for u in range(N):
            for h in hs:
                assert h[u].tokens == so[u].tokens and h[u].timestamps == so[u].timestamps and h[u].hyp == so[u].hyp, (calls, u)
        calls += 1
    for u in range(N):
        for l in range(recs[0].num_layers):
            for kind in ("val1", "val2"):
                want = so[u].state(l, kind)
                for h in hs:
                    np.testing.assert_allclose(h[u].state(l, kind), want, atol=5e-4, rtol=0)
    assert calls >= 3


def test_streaming_search_timeout_backs_off(tmp_path):
    """The tick of a large-vocabulary model runs the parted persistent search.  A timeout of the slabs' exchange (forced in the hook's
    "as a real one" form) is noticed when the tick's results come down: the search is repeated with one workgroup per stream from the
    launch's record, and the engine backs off to the unparted form for the next searches (four after a first timeout, doubling while they repeat).  Tokens, timestamps and Hyp stay on the oracle's through all of it, and the retry
    counter moves exactly once."""
    import ctypes as C
    from k2transducerasr_amd import OnlineRecognizer, load_library, set_switch
    from k2transducerasr_amd.synth import synth_utterance, write_synthetic_model
    from oracle.online import OnlineOracle
    p = str(tmp_path / "swide.k2w")
    write_synthetic_model(p, "zipformer2-streaming-tiny-test", blank_bias=2.4, meta_overrides={"vocab_size": "3000"})
    rec, ora = OnlineRecognizer(p), OnlineOracle(p)
    L = load_library()
    L.k2hip_debug_search_retries.argtypes = [C.c_void_p, C.POINTER(C.c_int32)]

    def retries():
        n = C.c_int32(0)
        assert L.k2hip_debug_search_retries(rec.model.handle, C.byref(n)) == 0
        return n.value
    feats = [ora.fbank(synth_utterance(930 + u, 3.6)) for u in range(3)]
    hs = [rec.create_online_stream() for _ in feats]
    os_ = [ora.create_stream() for _ in feats]
    for h, f in zip(hs, feats):
        h.add_features(f)
    T, S = rec.chunk_length, rec.shift_length
    nchunks = (feats[0].shape[0] - T) // S + 1
    assert nchunks >= 9
    n0 = retries()
    try:
        for k in range(nchunks):
            set_switch("K2HIP_TEST_GREEDY_TIMEOUT", 2 if k == 4 else 0)   # tick 4 times out
            rec.get_results(hs)
            ora.step(os_, [f[k * S : k * S + T] for f in feats])
            for h, o in zip(hs, os_):
                assert h.tokens == o.tokens and h.timestamps == o.timestamps and h.hyp == o.hyp, k
            assert retries() == n0 + (1 if k >= 4 else 0), k
    finally:
        set_switch("K2HIP_TEST_GREEDY_TIMEOUT", 0)
    assert sum(len(o.tokens) - 2 for o in os_) > 0


@pytest.mark.parametrize("preset,n_streams,secs", [("zipformer2-streaming-tiny-test", 5, 2.6), ("zipformer2-streaming-zh", 70, 1.4)])
def test_conv_module_inside_the_in_proj_gemm_equals_the_two_launches(tmp_path, preset, n_streams, secs):
    """Round 5: a streaming conv module's in_proj GEMM finishes with the GLU and the chunk-causal depthwise convolution on its own tile
    (gemm_glu_causal_conv: conv_module 3 -> 2 launches), where the shape has the fused form; K2HIP_NO_FUSED_CONV keeps linear +
    k_glu_causal_conv_reg.  The convolution's sums are the separate kernel's, in the same order; the in_proj product itself is summed
    in the order of the fused launch's tile form, which is the dispatcher's own choice for most shapes but not all -- so the two forms
    agree to float rounding, not bit for bit.  Two recognizers, one per form, step the same streams (ragged lengths, so the number of
    ready streams -- the GEMM's M, its tile form and its partial last tile -- changes from tick to tick): every token, timestamp and Hyp
    equal, every convolution cache within 5e-4; the tiny model also follows the oracle.  The zh architecture at 70 streams walks through
    all four tile forms (32x32 / 64x32 with the K step split four ways, 32x64 / 64x64 split two ways) and both kernel sizes."""
    from k2transducerasr_amd import OnlineRecognizer, set_switch
    from k2transducerasr_amd.synth import synth_utterance, write_synthetic_model
    p = str(tmp_path / "m.k2w")
    write_synthetic_model(p, preset)
    tiny = "tiny" in preset
    fused, plain = OnlineRecognizer(p), OnlineRecognizer(p)
    waves = [synth_utterance(5100 + u, secs + 0.35 * (u % 4)) for u in range(n_streams)]
    hf = [fused.create_online_stream() for _ in waves]
    hp = [plain.create_online_stream() for _ in waves]
    for a, b, w in zip(hf, hp, waves):
        a.add_samples(w)
        b.add_samples(w)
    ticks = 0
    while True:
        df, _ = fused.get_results(hf)
        set_switch("K2HIP_NO_FUSED_CONV", 1)
        try:
            dp, _ = plain.get_results(hp)
        finally:
            set_switch("K2HIP_NO_FUSED_CONV", 0)
        assert df == dp
        if not any(df):
            break
        ticks += 1
        for u, (a, b) in enumerate(zip(hf, hp)):
            assert a.tokens == b.tokens and a.timestamps == b.timestamps and a.hyp == b.hyp, (ticks, u)
        for u in range(0, n_streams, max(1, n_streams // 6)):
            for layer in range(fused.num_layers):
                for kind in ("conv1", "conv2"):
                    np.testing.assert_allclose(hf[u].state(layer, kind), hp[u].state(layer, kind), atol=5e-4, rtol=0, err_msg=f"tick {ticks} stream {u} layer {layer} {kind}")
    assert ticks >= 3 and sum(len(a.tokens) - 2 for a in hf) > 0
    if tiny:
        from oracle.online import OnlineOracle
        ora = OnlineOracle(p)
        for u, w in enumerate(waves):
            f = ora.fbank(w)
            o = ora.create_stream()
            for k in range((f.shape[0] - fused.chunk_length) // fused.shift_length + 1):
                ora.step([o], [f[k * fused.shift_length : k * fused.shift_length + fused.chunk_length]])
            assert hf[u].tokens == o.tokens and hf[u].timestamps == o.timestamps, u
