#!/usr/bin/env python3
"""Streaming benchmark (BASELINE.json configs[3]): zipformer streaming (OnlineProjOfZipformer2 replacement),
chunk = 32 frames, N concurrent streams on one MI355X.  Driver protocol of the reference's example
(K2TransducerAsr.Examples/OnlineRecognizer.cs:135-139,184-238): every stream is fed 800-sample (50 ms) pushes,
then 30 x 400 zero samples of tail; one GetResults per push round over all streams.
RTFx = total audio seconds / wall seconds, host samples in -> tokens in host memory (PCIe included).
Not the headline metric (that is bench.py); prints one JSON line."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=128)
    ap.add_argument("--seconds", type=float, default=20.0)
    ap.add_argument("--preset", default="zipformer2-streaming-zh")
    ap.add_argument("--check", type=int, default=0, help="verify the first K streams against the CPU oracle")
    args = ap.parse_args()
    if args.check and "conformer" in args.preset:
        raise SystemExit("--check decodes each stream alone on the oracle; the streaming conformer's processed_lens quirk "
                         "(OnlineProjOfConformer.cs:229: it becomes the batch size) makes that a different computation -- see "
                         "tests/test_online_gpu.py::test_streaming_random_churn for the batched comparison")
    import k2transducerasr_amd as pkg
    from k2transducerasr_amd.synth import synth_utterance, write_synthetic_model

    weights = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"k2hip_bench_{args.preset}.k2w")
    if not os.path.exists(weights):
        write_synthetic_model(weights, args.preset)
    os.environ.setdefault("K2HIP_MAX_STREAMS", str(max(256, args.streams)))
    rec = pkg.OnlineRecognizer(weights)
    N = args.streams
    utts = [synth_utterance(1000 + u, args.seconds) for u in range(N)]
    n = utts[0].size
    wave = np.stack(utts)  # [N, n]: a push round hands over one strided view, as a native host would hand over its buffers

    prof = [0.0, 0.0, 0.0]  # seconds in AddSamples, in GetResults with a decodable chunk, in GetResults without one

    def run():
        prof[:] = [0.0, 0.0, 0.0]
        streams = [rec.create_online_stream() for _ in range(N)]
        steps = 0
        t0 = time.perf_counter()
        for pos in range(0, n, 800):
            ta = time.perf_counter()
            rec.add_samples_batch(streams, wave[:, pos : pos + 800])
            tb = time.perf_counter()
            prof[0] += tb - ta
            # GetResults until no stream has a whole chunk left (models whose shift is shorter than a push -- lstm: 40 ms -- decode
            # more than one chunk per push; the Zipformers decode at most one and the second call is the idle one)
            while True:
                tb = time.perf_counter()
                dec, _ = rec.get_results(streams)
                tc = time.perf_counter()
                prof[1 if any(dec) else 2] += tc - tb
                steps += any(dec)
                if not any(dec):
                    break
        zeros = np.zeros((N, 400), np.float32)
        for _ in range(30):
            rec.add_samples_batch(streams, zeros)
            while True:
                dec, _ = rec.get_results(streams)
                steps += any(dec)
                if not any(dec):
                    break
        rec.model.synchronize()
        dt = time.perf_counter() - t0
        return streams, dt, steps

    streams, _, _ = run()  # warm-up (arena sizing, code load)
    for s in streams:
        s.close()
    streams, dt, steps = run()
    audio = N * args.seconds
    t = rec.model.timing()
    out = {
        "metric": "RTFx (audio-sec/wall-sec) streaming Zipformer2 greedy",
        "value": round(audio / dt, 1),
        "unit": "audio-sec/wall-sec",
        "n_gpus": 1,
        "config": {"workload": f"{args.preset} streaming greedy, chunk=32 frames, {N} concurrent streams x {args.seconds:g} s, "
                               "800-sample pushes + 30 x 400 zero tail (BASELINE.json configs[3])"},
        "chunk_steps": steps,
        "ms_per_chunk_step": round(dt / max(steps, 1) * 1e3, 3),
        "last_step_ms": {k: round(t[k], 3) for k in ("total_ms", "encoder_ms", "greedy_ms")},
        "host_phases_ms": {"add_samples": round(prof[0] * 1e3, 1), "get_results_decoding": round(prof[1] * 1e3, 1), "get_results_idle": round(prof[2] * 1e3, 1)},
        "tokens": int(sum(len(s.tokens) - 2 for s in streams)),
        "dtype": "f32",
        "data": "synthetic",
    }
    if args.check:
        from oracle.online import OnlineOracle
        ora = OnlineOracle(weights)
        for u in range(args.check):
            wav = np.concatenate([utts[u], np.zeros(30 * 400, np.float32)])
            f = ora.fbank(wav)
            o = ora.create_stream()
            for k in range((f.shape[0] - rec.chunk_length) // rec.shift_length + 1):
                ora.step([o], [f[k * rec.shift_length : k * rec.shift_length + rec.chunk_length]])
            assert o.tokens == streams[u].tokens, f"stream {u} differs from the oracle"
        out["oracle_checked_streams"] = args.check
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
