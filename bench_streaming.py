#!/usr/bin/env python3
"""Streaming benchmark (BASELINE.json configs[3]): zipformer streaming (OnlineProjOfZipformer2 replacement),
chunk = 32 frames, N concurrent streams on one MI355X.  Driver protocol of the reference's example
(K2TransducerAsr.Examples/OnlineRecognizer.cs:135-139,184-238): every stream is fed 800-sample (50 ms) pushes,
then 30 x 400 zero samples of tail; one GetResults per push round over all streams.
RTFx = total audio seconds / wall seconds, host samples in -> tokens in host memory (PCIe included).
Not the headline metric (that is bench.py); prints one JSON line.

--gpus N ("streams pinned to a GPU at CreateOnlineStream time, slot = stream_id mod nGPU"): the N ranks -- started by this
program itself before anything touches a GPU, or by torch.distributed.run -- each open ONE recognizer on their GPU and own the streams
u with u mod N == rank; every rank runs the same push / GetResults protocol over its streams, the timed region is bracketed by barriers,
the slowest rank's time counts, `value` = all streams' audio / that time.  No data-path collective (a stream never leaves its GPU).
--dist-backend gloo rehearses N ranks on fewer GPUs (ranks share devices): tests/test_dist_gloo.py."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def cpu_baseline(weights, n_streams, seconds, ids=None):
    """The CPU restatement (oracle/k2_oracle_online.c, 'port') of the same streaming path on this box's host cores, on a
    bounded sample: the first n_streams streams of the workload, `seconds` s each (+ the 30 x 400 zero tail), one batched step per
    chunk.  Returns the record and each stream's token list (compared with the GPU's for the same streams by the caller)."""
    from k2transducerasr_amd.synth import synth_utterance
    from oracle.online import OnlineOracle

    from bench import cpu_model, host_cores
    cores = host_cores()   # every core this process may run on, count and CPU model stated in the record
    os.environ["OMP_NUM_THREADS"] = str(cores)
    ora = OnlineOracle(weights)
    T, S = ora.chunk_length, ora.shift_length
    t0 = time.time()
    ids = list(range(n_streams)) if ids is None else list(ids)[:n_streams]   # stream ids of the workload (stream u's audio: seed 1000 + u)
    feats = [ora.fbank(np.concatenate([synth_utterance(1000 + u, seconds), np.zeros(30 * 400, np.float32)])) for u in ids]
    ss = [ora.create_stream() for _ in range(n_streams)]
    k = 0
    while k * S + T <= feats[0].shape[0]:
        ora.step(ss, [f[k * S : k * S + T] for f in feats])
        k += 1
    dt = time.time() - t0
    return {
        "value": round(n_streams * seconds / dt, 2),
        "unit": "x real-time (audio-sec/wall-sec)",
        "cores": cores,
        "cpu_model": cpu_model(),
        "kind": "port",
        "sample": f"{n_streams} concurrent streams x {seconds:g} s of the same synthetic workload ({k} chunk steps) through oracle/ "
                  f"(C + OpenMP restatement of fbank + the streaming encoder + the online greedy loop; the reference's ONNXRuntime path "
                  f"cannot run here), {dt:.2f} s wall",
    }, [(o.tokens, o.timestamps) for o in ss]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=128)
    ap.add_argument("--seconds", type=float, default=20.0)
    ap.add_argument("--preset", default="zipformer2-streaming-zh")
    ap.add_argument("--check", type=int, default=0, help="verify the first K streams (tokens and timestamps) against the CPU oracle: the "
                    "CPU-baseline sample then IS those K streams at full length, and the line carries oracle_match")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-streams", type=int, default=8, help="streams of the CPU-baseline sample")
    ap.add_argument("--cpu-seconds", type=float, default=4.0, help="audio seconds per stream of the CPU-baseline sample")
    ap.add_argument("--gpus", type=int, default=1, help="ranks = GPUs; stream u runs on rank u mod N (the program starts its own ranks)")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, one rank per GPU) or gloo (rehearsal: ranks may share a GPU)")
    ap.add_argument("--dump-results", default="", help="rank 0 writes every stream's (tokens, timestamps), in stream order, here (JSON)")
    args = ap.parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:   # before anything touches the GPU: the ranks are separate programs
        from bench import launch_ranks
        sys.exit(launch_ranks(args.gpus, os.path.abspath(__file__)))
    rank, local_rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        print(f"[bench_streaming] WORLD_SIZE={world} but --gpus {args.gpus}: refusing to report a line for a different job size", file=sys.stderr)
        sys.exit(2)
    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist
        if args.dist_backend == "nccl":
            if local_rank >= torch.cuda.device_count():
                print(f"[bench_streaming] rank {rank}: no GPU {local_rank} here; --dist-backend gloo rehearses N ranks on fewer GPUs", file=sys.stderr)
                sys.exit(4)
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend="gloo")

    def barrier():
        if dist is not None:
            dist.barrier()
    if args.check and "conformer" in args.preset:
        raise SystemExit("--check decodes each stream alone on the oracle; the streaming conformer's processed_lens quirk "
                         "(OnlineProjOfConformer.cs:229: it becomes the batch size) makes that a different computation -- see "
                         "tests/test_online_gpu.py::test_streaming_random_churn for the batched comparison")
    import k2transducerasr_amd as pkg
    from k2transducerasr_amd.synth import synth_utterance, write_synthetic_model

    from k2transducerasr_amd.shard import gather_results, max_over_ranks
    weights = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"k2hip_bench_{args.preset}.k2w")
    if rank == 0 and not os.path.exists(weights):
        tmp = weights + f".tmp{os.getpid()}"
        write_synthetic_model(tmp, args.preset)
        os.replace(tmp, weights)
    barrier()
    os.environ.setdefault("K2HIP_MAX_STREAMS", str(max(256, args.streams)))
    n_dev = pkg.load_library().k2hip_device_count()
    device = local_rank if args.dist_backend == "nccl" else local_rank % max(n_dev, 1)
    rec = pkg.OnlineRecognizer(weights, device)
    mine = [u for u in range(args.streams) if u % world == rank]   # CreateOnlineStream pins stream u to GPU u mod N
    N = len(mine)
    utts = [synth_utterance(1000 + u, args.seconds) for u in mine]
    n = utts[0].size
    wave = np.stack(utts)  # [N, n]: a push round hands over one strided view, as a native host would hand over its buffers

    prof = [0.0, 0.0, 0.0]  # seconds in AddSamples, in GetResults with a decodable chunk, in GetResults without one

    def run():
        prof[:] = [0.0, 0.0, 0.0]
        streams = rec.batch([rec.create_online_stream() for _ in range(N)])  # the handle array a native host would hold
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
    barrier()
    streams, dt, steps = run()
    barrier()
    dt = max_over_ranks(dist, dt, device="cuda" if (dist is not None and args.dist_backend == "nccl") else None)
    audio = args.streams * args.seconds   # whole job: every rank's streams over the slowest rank's time
    mine_res = [(u, list(s.tokens), list(s.timestamps)) for u, s in zip(mine, streams)]
    all_res = sorted(gather_results(dist, mine_res, world, rank)) if world > 1 else mine_res
    if rank == 0 and args.dump_results:
        with open(args.dump_results, "w") as f:
            json.dump({"results": [[t, ts] for _, t, ts in all_res], "n_gpus": world}, f)
    if rank != 0:   # the roofline / oracle legs below are rank 0's (one GPU's tick); the other ranks are done
        for s in streams:
            s.close()
        if dist is not None:
            dist.destroy_process_group()
        return
    tokens_all = int(sum(len(t) - 2 for _, t, _ in all_res))
    t = rec.model.timing()
    tokens = tokens_all
    checked = [(streams[k].tokens, streams[k].timestamps) for k in range(min(args.check, N))]   # rank 0's first streams: ids 0, world, 2 world, ...
    args.check = len(checked)
    for s in streams:
        s.close()

    # roofline legs: one more full-width tick with HIP events around every GEMM launch (instrumented pass, not timed above).
    # All N streams get exactly one chunk, so this is the 128-row-per-frame tick the metric is made of.
    F32_MFMA_PEAK_TFLOPS, HBM_PEAK_GBS = 157.3, 8000.0   # MI355X_MICROARCH.md
    probe = [rec.create_online_stream() for _ in range(N)]
    need = (rec.chunk_length - 1) * 160 + 400             # samples that give exactly chunk_length frames
    rec.add_samples_batch(probe, wave[:, :need])
    rec.model.set_instrument(True)
    dec, _ = rec.get_results(probe)
    it = rec.model.timing()
    rec.model.set_instrument(False)
    assert all(dec), "instrumented tick did not decode every stream"
    state_floats = sum(probe[0].state(l, k).size for l in range(rec.num_layers) for k in rec.state_kinds) + rec.embed_state_floats
    for s in probe:
        s.close()
    import k2transducerasr_amd.k2w as k2w
    _, tensors = k2w.read_k2w(weights)
    weight_bytes = int(sum(a.nbytes for n_, a in tensors.items() if n_.startswith(("encoder", "joiner", "decoder"))))
    tick_ms = dt / max(steps, 1) * 1e3
    tick_bytes = 2 * 4 * state_floats * N + weight_bytes   # every cache read and rewritten once per tick + the weights once
    ach = it["gemm_flops"] / (it["gemm_ms"] * 1e-3) / 1e12 if it["gemm_ms"] > 0 else 0.0
    out = {
        "metric": "RTFx (audio-sec/wall-sec) streaming Zipformer2 greedy",
        "value": round(audio / dt, 1),
        "unit": "audio-sec/wall-sec",
        "n_gpus": world,
        "scaling": "strong",
        "higher_is_better": True,
        "vs_baseline": None,
        "config": {"workload": f"{args.preset} streaming greedy, chunk=32 frames, {args.streams} concurrent streams x {args.seconds:g} s"
                               f"{'' if world == 1 else ' (stream u on GPU u mod %d: %d on rank 0)' % (world, N)}, "
                               "800-sample pushes + 30 x 400 zero tail (BASELINE.json configs[3]); host samples in, tokens in host memory"},
        "chunk_steps": steps,
        "ms_per_chunk_step": round(tick_ms, 3),
        "last_step_ms": {k: round(t[k], 3) for k in ("total_ms", "encoder_ms", "greedy_ms")},
        "host_phases_ms": {"add_samples": round(prof[0] * 1e3, 1), "get_results_decoding": round(prof[1] * 1e3, 1), "get_results_idle": round(prof[2] * 1e3, 1)},
        "tokens": tokens,
        "emission_rate": round(tokens / max(steps * args.streams * rec.frames_per_chunk, 1), 4),
        "dtype": "f32",
        "data": "synthetic",
        "roofline": {
            "kernel": "gemm_f32_mfma* (every Linear / pointwise-conv / implicit-conv / attention-apply launch of one chunk step)",
            "bound": "mfma",
            "achieved": round(ach, 2),
            "peak": F32_MFMA_PEAK_TFLOPS,
            "unit": "TFLOP/s",
            "frac": round(ach / F32_MFMA_PEAK_TFLOPS, 4),
            "traffic": None,
            "launches_per_tick": it["gemm_launches"],
            "flops_per_tick": it["gemm_flops"],
            "avg_launch_us": round(it["gemm_ms"] * 1e3 / max(it["gemm_launches"], 1), 2),
            "gemm_ms_per_tick": round(it["gemm_ms"], 3),
            # the whole tick against both ceilings (the step is launch/latency bound: it sits far from either)
            "tick": {"all_matrix_flops": it["total_flops"], "tflops": round(it["total_flops"] / (tick_ms * 1e-3) / 1e12, 2),
                     "frac_of_mfma_peak": round(it["total_flops"] / (tick_ms * 1e-3) / 1e12 / F32_MFMA_PEAK_TFLOPS, 4),
                     "algorithmic_hbm_bytes": tick_bytes, "gbs": round(tick_bytes / (tick_ms * 1e-3) / 1e9, 1),
                     "frac_of_hbm_peak": round(tick_bytes / (tick_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                     "state_bytes_per_stream": 4 * state_floats, "weight_bytes": weight_bytes},
        },
    }
    bad = False
    if args.check and "conformer" not in args.preset and "lstm" not in args.preset:
        # Zipformer streams are independent of their batch mates: the oracle steps the K checked streams as one batch (that run is
        # also the CPU baseline), and every token and timestamp must equal what the GPU gave the same streams among the other N - K
        cb, want = cpu_baseline(weights, args.check, args.seconds, mine)
        exact = sum(1 for g, w in zip(checked, want) if g == w)
        out["oracle_match"] = {"streams": args.check, "exact": exact,
                               "what": "tokens and timestamps of the timed run's first streams == oracle/ on the same streams"}
        out["oracle_checked_streams"] = args.check
        bad = exact < args.check
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cb
    elif args.check:
        from oracle.online import OnlineOracle
        ora = OnlineOracle(weights)
        for u in range(args.check):
            wav = np.concatenate([utts[u], np.zeros(30 * 400, np.float32)])
            f = ora.fbank(wav)
            o = ora.create_stream()
            for k in range((f.shape[0] - rec.chunk_length) // rec.shift_length + 1):
                ora.step([o], [f[k * rec.shift_length : k * rec.shift_length + rec.chunk_length]])
            assert o.tokens == checked[u][0], f"stream {u} differs from the oracle"
        out["oracle_checked_streams"] = args.check
    if "cpu_baseline" not in out and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(weights, args.cpu_streams, args.cpu_seconds)[0]
    if dist is not None:
        out["config"]["parallelism"] = f"stream u on rank u mod {world} ({'one rank per GPU, RCCL barrier only' if args.dist_backend == 'nccl' else 'gloo rehearsal, ranks share GPUs'}); no data-path collective"
    print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()
    if bad:
        print("[bench_streaming] oracle_match failed", file=sys.stderr)
        sys.exit(5)


if __name__ == "__main__":
    main()
