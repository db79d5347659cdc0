#!/usr/bin/env python3
"""Dev tool (GPU): a longer version of tests/test_online_gpu.py::test_streaming_random_churn over every streaming operator: streams
arrive at random times, are fed in uneven pushes, finish and are replaced (slots recycled); after every GetResults call every live
stream's tokens / timestamps / Hyp must equal the CPU oracle's (stepped with the same ready subsets).
usage: soak_streaming.py [total-streams-per-model] [seed]"""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import k2transducerasr_amd as pkg  # noqa: E402
from k2transducerasr_amd.synth import synth_utterance, write_synthetic_model  # noqa: E402
from oracle.online import OnlineOracle  # noqa: E402

TOTAL = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 3
presets = ["zipformer2-streaming-tiny-test", "zipformer-streaming-tiny-test", "conformer-streaming-tiny-test", "conformer-streaming-rc-tiny-test",
           "zipformer2-ctc-streaming-tiny-test",
           "lstm-tiny-test"]
tmp = tempfile.mkdtemp()
for preset in presets:
    p = os.path.join(tmp, preset + ".k2w")
    write_synthetic_model(p, preset)
    rec, ora = pkg.OnlineRecognizer(p), OnlineOracle(p)
    T, S = rec.chunk_length, rec.shift_length
    rng = np.random.default_rng(seed)
    NSLOT = 9
    live, started, finished, toks, calls = [], 0, 0, 0, 0
    for it in range(200000):
        while len(live) < NSLOT and started < TOTAL and rng.random() < 0.5:
            f = ora.fbank(synth_utterance(9000 + started, float(rng.uniform(0.4, 2.2))))
            live.append(dict(h=rec.create_online_stream(), o=ora.create_stream(), feats=f, fed=0, pos=0))
            started += 1
        for s in live:
            if s["fed"] < s["feats"].shape[0] and rng.random() < 0.8:
                n = int(rng.integers(1, 50))
                s["h"].add_features(s["feats"][s["fed"] : s["fed"] + n])
                s["fed"] = min(s["fed"] + n, s["feats"].shape[0])
        if not live:
            if started == TOTAL:
                break
            continue
        ready = [i for i, s in enumerate(live) if s["pos"] + T <= s["fed"]]
        dec, n_new = rec.get_results([s["h"] for s in live])
        calls += 1
        assert [i for i in range(len(live)) if dec[i]] == ready, (preset, it)
        if ready:
            want = ora.step([live[i]["o"] for i in ready], [live[i]["feats"][live[i]["pos"] : live[i]["pos"] + T] for i in ready])
            for i, wn in zip(ready, want):
                assert n_new[i] == wn, (preset, it, i)
                live[i]["pos"] += S
        for s in live:
            assert s["h"].tokens == s["o"].tokens and s["h"].timestamps == s["o"].timestamps and s["h"].hyp == s["o"].hyp, (preset, it)
        keep = []
        for s in live:
            if s["fed"] == s["feats"].shape[0] and s["pos"] + T > s["fed"]:
                toks += len(s["o"].tokens) - 2
                s["h"].close()
                finished += 1
            else:
                keep.append(s)
        live = keep
    assert finished == TOTAL
    print(f"{preset}: {TOTAL} streams through {NSLOT} slots, {calls} GetResults calls, {toks} tokens, all equal to the oracle", flush=True)
print("soak ok")
