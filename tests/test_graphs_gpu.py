"""hipGraph replay of the encoder passes (Engine::graphed): the first call with a shape is enqueued eagerly, the second is captured,
later ones replay the instance.  A replay must be the same computation -- same tokens, same encoder output bits -- as the eager
chain, for the offline batch entries (synchronous and pipelined) and for the streaming tick, and a changed shape / pool must not
hit a stale instance."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def streaming_graphs_on():
    """the streaming tick's graph replay is opt-in (K2HIP_GRAPH_STREAMING; off by default since it stopped paying): on for this file"""
    from k2transducerasr_amd import set_switch
    set_switch("K2HIP_GRAPH_STREAMING", 1)
    yield
    set_switch("K2HIP_GRAPH_STREAMING", 0)


def graph_launches(model):
    from k2transducerasr_amd import load_library
    L = load_library()
    L.k2hip_debug_graph_launches.argtypes = [C.c_void_p, C.POINTER(C.c_int32)]
    n = C.c_int32(0)
    assert L.k2hip_debug_graph_launches(model.handle, C.byref(n)) == 0
    return n.value


def test_offline_replays_equal_the_eager_chain(tiny_model_path, oracle_tiny):
    from k2transducerasr_amd import Model, set_switch
    from k2transducerasr_amd.synth import synth_utterance
    m = Model(tiny_model_path, 0)
    a = np.stack([synth_utterance(50 + u, 1.2) for u in range(4)])
    b = np.stack([synth_utterance(60 + u, 0.9) for u in range(3)])     # another shape in between
    pa, pb = m.device_alloc(a.nbytes), m.device_alloc(b.nbytes)
    m.device_upload(pa, a)
    m.device_upload(pb, b)
    want_a = m.offline_greedy_from_samples_dev(pa, a.shape[1], 4)      # (offline passes are eager unless K2HIP_GRAPH_OFFLINE is set)
    want_b = m.offline_greedy_from_samples_dev(pb, b.shape[1], 3)
    for _ in range(3):
        assert m.offline_greedy_from_samples_dev(pa, a.shape[1], 4) == want_a
    assert graph_launches(m) == 0
    set_switch("K2HIP_GRAPH_OFFLINE", 1)
    feats = [oracle_tiny.fbank(u) for u in a]
    assert want_a == oracle_tiny.recognize_batch(feats)
    n0 = graph_launches(m)
    for k in range(5):          # eager, captured + launched, replayed ...
        assert m.offline_greedy_from_samples_dev(pa, a.shape[1], 4) == want_a, k
        assert m.offline_greedy_from_samples_dev(pb, b.shape[1], 3) == want_b, k
    assert graph_launches(m) - n0 >= 6, "the encoder passes were not replayed from graphs"
    # the pipelined entries: each slot has its own arena, hence its own instance
    n1 = graph_launches(m)
    for k in range(4):
        t1 = m.offline_submit_samples_dev(pa, a.shape[1], 4)
        t2 = m.offline_submit_samples_dev(pa, a.shape[1], 4)
        assert m.offline_wait(t1) == want_a and m.offline_wait(t2) == want_a, k
    assert graph_launches(m) > n1
    set_switch("K2HIP_GRAPH_OFFLINE", 0)
    m.device_free(pa)
    m.device_free(pb)
    m.close()


def test_streaming_ticks_replay_and_follow_the_oracle(tmp_path):
    """ticks with the same number of ready streams replay one instance; the streams' slots, ring heads and contexts are data, so the
    instance serves whichever streams are ready -- tokens and caches stay on the oracle's, chunk by chunk"""
    from k2transducerasr_amd import OnlineRecognizer
    from k2transducerasr_amd.synth import synth_utterance, write_synthetic_model
    from oracle.online import OnlineOracle
    p = str(tmp_path / "s.k2w")
    write_synthetic_model(p, "zipformer2-streaming-tiny-test")
    rec, ora = OnlineRecognizer(p), OnlineOracle(p)
    feats = [ora.fbank(synth_utterance(70 + u, 2.6)) for u in range(3)]
    hs = [rec.create_online_stream() for _ in feats]
    os_ = [ora.create_stream() for _ in feats]
    for h, f in zip(hs, feats):
        h.add_features(f)
    T, S = rec.chunk_length, rec.shift_length
    nchunks = (feats[0].shape[0] - T) // S + 1
    assert nchunks >= 6
    from k2transducerasr_amd import set_switch
    for k in range(nchunks):
        set_switch("K2HIP_NO_GRAPHS", 1 if k == 4 else 0)     # one eager tick in the middle continues where the replays left the caches
        n_before = graph_launches(rec.model)
        rec.get_results(hs)
        if k == 4:
            assert graph_launches(rec.model) == n_before
        ora.step(os_, [f[k * S: k * S + T] for f in feats])
        for h, o in zip(hs, os_):
            assert h.tokens == o.tokens and h.timestamps == o.timestamps, k
    set_switch("K2HIP_NO_GRAPHS", 0)
    assert graph_launches(rec.model) >= nchunks - 3
    for h, o in zip(hs, os_):
        np.testing.assert_allclose(h.state(0, "key"), o.state(0, "key"), atol=2e-4, rtol=0)


def test_foreign_legacy_stream_traffic_while_a_tick_is_recorded(tmp_path):
    """Another thread of the HOST process (not this library) keeps issuing legacy-stream copies -- what a plain hipMemcpy or a
    framework on the default stream does.  While a tick is being recorded the runtime fails those copies and invalidates the
    recording; the tick must then run eagerly from the same arena position and give the same tokens (Engine::graphed: nothing was
    enqueued by the failed recording), and later ticks must keep working whether or not their recording survived.  (The copies that
    land inside a recording window are the ones the runtime refuses -- the count is printed; that direction is the host's to avoid:
    INTEGRATION.md "Threading".)

    The scenario runs in a child process (tests/foreign_legacy_child.py): hammering legacy copies against stream captures also races
    INSIDE the HIP runtime of ROCm 7.2 -- one run in about ten of this test died with a segmentation fault in the foreign thread's
    hipMemcpy, in libamdhip64, not in this library -- and a crash of the runtime must not take the test session with it.  A child that
    finishes must report equal tokens; a child killed by the runtime's own fault is reported as a skip that says so."""
    import subprocess
    import sys
    child = os.path.join(os.path.dirname(os.path.abspath(__file__)), "foreign_legacy_child.py")
    r = subprocess.run([sys.executable, child, str(tmp_path / "s.k2w")], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    print(r.stdout.strip())
    if r.returncode < 0 or r.returncode in (134, 139):
        pytest.skip(f"the HIP runtime itself crashed under concurrent legacy copies and stream captures (exit {r.returncode}); "
                    "hosts with such threads run with K2HIP_NO_GRAPHS=1 (INTEGRATION.md)")
    assert r.returncode == 0, (r.returncode, r.stdout[-2000:], r.stderr[-3000:])
    assert "tokens equal: True" in r.stdout


def test_a_flipped_switch_never_replays_the_old_chain(tmp_path):
    """Recorded graphs are keyed by the generation of the switches: one recognizer whose ticks alternate between the two forms of the
    self-attention value projection (a different chain of launches, different arena use) must follow the oracle through every tick --
    a replay of the chain recorded under the other form would not."""
    from k2transducerasr_amd import OnlineRecognizer, set_switch
    from k2transducerasr_amd.synth import synth_utterance, write_synthetic_model
    from oracle.online import OnlineOracle
    p = str(tmp_path / "s.k2w")
    write_synthetic_model(p, "zipformer2-streaming-tiny-test")
    rec, ora = OnlineRecognizer(p), OnlineOracle(p)
    feats = [ora.fbank(synth_utterance(170 + u, 3.4)) for u in range(3)]
    hs = [rec.create_online_stream() for _ in feats]
    os_ = [ora.create_stream() for _ in feats]
    for h, f in zip(hs, feats):
        h.add_features(f)
    T, S = rec.chunk_length, rec.shift_length
    nchunks = (feats[0].shape[0] - T) // S + 1
    assert nchunks >= 9
    try:
        for k in range(nchunks):
            set_switch("K2HIP_NO_FUSED_VPROJ", (k // 3) & 1)      # three ticks of one form (eager, recorded, replayed), then the other
            rec.get_results(hs)
            ora.step(os_, [f[k * S : k * S + T] for f in feats])
            for h, o in zip(hs, os_):
                assert h.tokens == o.tokens and h.timestamps == o.timestamps, k
    finally:
        set_switch("K2HIP_NO_FUSED_VPROJ", 0)
    for h, o in zip(hs, os_):
        for kind in ("val1", "val2", "conv1"):
            np.testing.assert_allclose(h.state(0, kind), o.state(0, kind), atol=5e-4, rtol=0)
    assert graph_launches(rec.model) >= 2
