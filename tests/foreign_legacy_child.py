"""Child process of tests/test_multi_handle_gpu.py::test_foreign_legacy_stream_traffic_beside_the_ticks (TEST INFRASTRUCTURE).

One streaming recognizer decodes through changing stream counts while a second host thread -- playing a HOST application's own HIP
code -- issues legacy-stream hipMemcpy calls back to back.  The library never uses the legacy stream and (since round 5) never captures
a stream, so none of those copies may be refused and the tokens must equal those of an undisturbed recognizer; exit code 0 iff both
hold.  usage: foreign_legacy_child.py <model path to write>"""
import ctypes as C
import os
import sys
import threading

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from k2transducerasr_amd import OnlineRecognizer, load_library  # noqa: E402
from k2transducerasr_amd.synth import synth_utterance, write_synthetic_model  # noqa: E402


def main():
    p = sys.argv[1]
    hip = C.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    write_synthetic_model(p, "zipformer2-streaming-tiny-test")
    waves = [synth_utterance(900 + u, 2.4) for u in range(6)]

    def decode(rec, counts):
        out = []
        for n in counts:
            hs = [rec.create_online_stream() for _ in range(n)]
            for h, w in zip(hs, waves):
                h.add_samples(w)
            while any(rec.get_results(hs)[0]):
                pass
            out.append([(list(h.tokens), list(h.timestamps)) for h in hs])
            for h in hs:
                h.close()
        return out

    counts = [3, 5, 2, 6, 4, 3, 5, 2, 6, 4]
    quiet = OnlineRecognizer(p)
    want = decode(quiet, counts)
    stop, stats = threading.Event(), {"copies": 0, "refused": 0}

    def foreign():
        dev, host = C.c_void_p(), (C.c_char * 4096)()
        assert hip.hipMalloc(C.byref(dev), 4096) == 0
        while not stop.is_set():
            rc = hip.hipMemcpy(dev, host, 4096, 1)        # hipMemcpyHostToDevice on the legacy stream
            stats["copies"] += 1
            stats["refused"] += rc != 0
    th = threading.Thread(target=foreign)
    th.start()
    try:
        rec = OnlineRecognizer(p)
        got = decode(rec, counts)
    finally:
        stop.set()
        th.join()
    print(f"foreign legacy-stream copies: {stats['copies']}, refused by the runtime: {stats['refused']}")
    print(f"tokens equal: {got == want}")
    sys.stdout.flush()
    sys.exit(0 if got == want and stats["copies"] > 100 and stats["refused"] == 0 else 1)


if __name__ == "__main__":
    main()
