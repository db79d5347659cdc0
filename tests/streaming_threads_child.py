"""Child process of tests/test_multi_handle_gpu.py::test_streaming_handles_on_host_threads (TEST
INFRASTRUCTURE): the scenario itself; exit code 0 iff every stream of every thread equals the one-handle decode.
usage: streaming_threads_child.py <model path to write>"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main(model_path):

    import threading

    from k2transducerasr_amd import OnlineRecognizer
    from k2transducerasr_amd.synth import synth_utterance, write_synthetic_model
    p = model_path
    write_synthetic_model(p, "zipformer2-streaming-tiny-test")
    NT, ROUNDS = 4, 5
    counts = [[3, 5, 2, 6, 4], [4, 2, 6, 3, 5], [5, 6, 3, 2, 4], [2, 4, 5, 6, 3]]
    ref = OnlineRecognizer(p)
    waves = {}

    def decode(rec, thread, rnd, with_extras):
        n = counts[thread][rnd]
        hs = [rec.create_online_stream() for _ in range(n)]
        for u, h in enumerate(hs):
            key = (thread, rnd, u)
            if key not in waves:
                waves[key] = synth_utterance(4000 + 97 * thread + 13 * rnd + u, 2.0 + 0.3 * (u % 3))
            w = waves[key]
            for pos in range(0, w.size, 1600):
                h.add_samples(w[pos: pos + 1600])
        ticks = 0
        while True:
            dec, _ = rec.get_results(hs)
            if not any(dec):
                break
            ticks += 1
            if with_extras and ticks % 2 == 0:
                hs[0].state(0, "key")              # a blocking device -> host read between ticks
                rec.model.synchronize()
        out = [(list(h.tokens), list(h.timestamps)) for h in hs]
        for h in hs:
            h.close()
        return out, ticks

    for t in range(NT):                  # the audio, made once (the threads then only decode)
        for r in range(ROUNDS):
            decode(ref, t, r, False)
    want = {(t, r): decode(ref, t, r, False)[0] for t in range(NT) for r in range(ROUNDS)}
    recs = [OnlineRecognizer(p) for _ in range(NT)]
    got, errors = {}, []
    bar = threading.Barrier(NT)

    def worker(t):
        try:
            bar.wait()
            for r in range(ROUNDS):
                got[(t, r)], ticks = decode(recs[t], t, r, True)
                assert ticks >= 3
        except Exception as e:  # noqa: BLE001 -- reported by the main thread
            errors.append((t, repr(e)))

    # ... and one more thread opens and closes handles of its own meanwhile (device allocations, uploads, the decoder table's kernel,
    # frees, stream destruction -- next to the others' recordings), decoding a round through each
    stop = threading.Event()
    lifecycle = {"models": 0}

    def opener():
        try:
            while not stop.is_set():
                r = OnlineRecognizer(p)
                out, _ = decode(r, 0, 0, False)
                assert out == want[(0, 0)]
                r.model.close()
                lifecycle["models"] += 1
        except Exception as e:  # noqa: BLE001
            errors.append(("opener", repr(e)))

    th = [threading.Thread(target=worker, args=(t,)) for t in range(NT)]
    op = threading.Thread(target=opener)
    op.start()
    for x in th:
        x.start()
    for x in th:
        x.join()
    stop.set()
    op.join()
    assert not errors, errors
    assert got == want
    assert lifecycle["models"] >= 1
    assert sum(len(tok) - 2 for v in want.values() for tok, _ in v) > 0
    print(f"threads ok: {sum(len(v) for v in want.values())} streams over {NT} threads x {ROUNDS} rounds, {lifecycle['models']} handles opened and closed meanwhile")
    sys.stdout.flush()
    os._exit(0)


if __name__ == "__main__":
    main(sys.argv[1])
