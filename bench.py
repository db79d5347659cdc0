#!/usr/bin/env python3
"""Benchmark of the K2TransducerAsr offline greedy hot path on MI355X.

Workload (BASELINE.json configs[1]): zipformer-large-en offline greedy, batch = 32
synthetic 10 s utterances per GPU, 16 kHz f32 samples already resident in HBM when the
timed region starts.  One step = one pass of the whole path over one batch:
fbank -> PadSequence (+19 frames, log-floor) -> Zipformer2-large encoder -> on-device
greedy search -> token arrays back in host memory.

Metric: RTFx = audio seconds / wall seconds (inverse of the reference's printed
`rtf = elapsed_ms / total_duration_ms`, K2TransducerAsr.Examples/OfflineRecognizer.cs:185-189),
whole job over all ranks.  Multi-GPU: one process per GPU (torch.distributed.run), each
rank decodes its own shard of utterances with no collective on the data path (weak
scaling); RCCL is used only for the timing barrier and the max-over-ranks.

    python bench.py --gpus 1 --steps 10 --warmup 2
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PRESET = "zipformer2-large-en"
BATCH = 32
UTT_SECONDS = 10.0
F32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: Peak FP32 (matrix), dense


def pmc_traffic():
    """HBM bytes per GEMM launch from the committed rocprofv3 PMC passes (cannot be read live:
    counters need their own profiler run).  Valid only for the default workload they were taken on."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_gemm_traffic.json")))
    if not files:
        return None, "no PMC summary under profiles/"
    with open(files[-1]) as f:
        d = json.load(f)
    return d["hbm_bytes_per_launch"], (f"bytes per launch (fetch {d['fetch_bytes_per_launch']} + write {d['write_bytes_per_launch']}), "
                                       f"from {os.path.relpath(files[-1], ROOT)}: {d['method']}")


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def ensure_weights(path, preset, rank, barrier):
    from k2transducerasr_amd.synth import write_synthetic_model

    if rank == 0 and not os.path.exists(path):
        tmp = path + f".tmp{os.getpid()}"
        t = time.time()
        write_synthetic_model(tmp, preset)
        os.replace(tmp, path)
        log(f"[bench] wrote synthetic {preset} weights to {path} in {time.time() - t:.1f}s")
    barrier()


def cpu_baseline(weights, n_utts, seconds):
    """The CPU restatement (oracle/, 'port') of the same path on this box's host cores,
    on a bounded sample of the same workload."""
    from k2transducerasr_amd.synth import synth_utterance
    from oracle import Oracle

    cores = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(len(os.sched_getaffinity(0)), 16)
    os.environ["OMP_NUM_THREADS"] = str(cores)
    ora = Oracle(weights)
    utts = [synth_utterance(u, seconds) for u in range(n_utts)]

    def run():
        feats = [ora.fbank(u) for u in utts]
        return ora.recognize_batch(feats)

    run_one = [synth_utterance(0, 1.0)]
    ora.recognize_batch([ora.fbank(run_one[0])])  # builds the transposed-weight cache
    t = time.time()
    res = run()
    dt = time.time() - t
    return {
        "value": round(n_utts * seconds / dt, 2),
        "unit": "x real-time (audio-sec/wall-sec)",
        "cores": cores,
        "kind": "port",
        "sample": f"{n_utts} x {seconds:g} s utterances of the same synthetic workload as one batch through oracle/ "
                  f"(C + OpenMP restatement; the reference's ONNXRuntime path cannot run here), {dt:.2f} s wall",
    }, res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--preset", default=PRESET)
    ap.add_argument("--batch", type=int, default=BATCH)
    ap.add_argument("--seconds", type=float, default=UTT_SECONDS)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pipeline", action="store_true", help="one synchronous call per step (no batch overlap)")
    ap.add_argument("--cpu-utts", type=int, default=32)
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, one rank per GPU) or gloo (rehearsal of the N > 1 path on one GPU)")
    ap.add_argument("--beam", type=int, default=0, help="0 = greedy_search (headline metric); K = modified_beam_search with beam K (BASELINE configs[2])")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        log(f"[bench] WORLD_SIZE={world} but --gpus {args.gpus}; using WORLD_SIZE")
    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist

        if args.dist_backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=args.dist_backend)

    def barrier():
        if dist is not None:
            dist.barrier()

    import k2transducerasr_amd as pkg
    from k2transducerasr_amd.synth import synth_utterance

    weights = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"k2hip_bench_{args.preset}.k2w")
    ensure_weights(weights, args.preset, rank, barrier)

    n_dev = pkg.load_library().k2hip_device_count()
    device = local_rank if args.dist_backend == "nccl" else local_rank % max(n_dev, 1)  # gloo rehearsal: ranks may share a GPU
    model = pkg.Model(weights, device)  # no fallback: raises without a GPU / library
    if args.beam > 0:
        model.set_decoding_method("modified_beam_search", args.beam)
    B, secs = args.batch, args.seconds
    n_each = int(round(secs * 16000))
    # each rank owns a different shard of utterances (seeds offset by rank)
    samples = np.stack([synth_utterance(rank * B + u, secs) for u in range(B)])
    dptr = model.device_alloc(samples.nbytes)
    model.device_upload(dptr, samples)

    def step():
        return model.offline_greedy_from_samples_dev(dptr, n_each, B)

    def run_steps(n):
        """n passes over the batch, software-pipelined two deep: batch i+1 is submitted before
        batch i's tokens are collected, so its encoder overlaps batch i's greedy loop.  Every
        batch's tokens are back in host memory before this returns."""
        if args.no_pipeline or n == 0:
            out = None
            for _ in range(n):
                out = step()
            return out
        tk = model.offline_submit_samples_dev(dptr, n_each, B)
        for _ in range(n - 1):
            nxt = model.offline_submit_samples_dev(dptr, n_each, B)
            out = model.offline_wait(tk)
            tk = nxt
        return model.offline_wait(tk)

    res = run_steps(args.warmup)
    model.synchronize()
    barrier()
    t0 = time.perf_counter()
    res = run_steps(args.steps)
    model.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    from k2transducerasr_amd.shard import max_over_ranks

    elapsed = max_over_ranks(dist, elapsed, device="cuda" if (dist is not None and args.dist_backend == "nccl") else None)
    res_sync = step()  # one synchronous pass: per-stage HIP-event timings + pipelined == synchronous check
    stages = model.timing()
    assert res_sync == res, "pipelined and synchronous results differ"

    # roofline of the dominant kernel (fp32 MFMA GEMM): one extra instrumented pass
    # over the same batch, HIP events recorded around every GEMM launch on the
    # engine's own stream (no per-launch sync, launches stay back to back).
    model.set_instrument(True)
    step()
    it = model.timing()
    model.set_instrument(False)

    if rank == 0:
        audio = world * args.steps * B * secs
        value = audio / elapsed
        ach = it["gemm_flops"] / (it["gemm_ms"] * 1e-3) / 1e12 if it["gemm_ms"] > 0 else 0.0
        default_workload = args.preset == PRESET and B == BATCH and abs(secs - UTT_SECONDS) < 1e-9 and args.beam == 0
        traffic, traffic_note = pmc_traffic() if default_workload else (None, "PMC passes exist for the default workload only")
        out = {
            "metric": "RTFx (audio-sec/wall-sec) offline Zipformer greedy",
            "value": round(value, 1),
            "unit": "audio-sec/wall-sec",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"{args.preset} offline {'greedy' if args.beam == 0 else 'modified-beam-search beam=%d' % args.beam}, batch={B} synthetic {secs:g} s utterances per GPU "
                            "(BASELINE.json configs[1]); samples resident in HBM, tokens returned to host",
                "batch_per_gpu": B,
                "utt_seconds": secs,
                "parallelism": f"utterance-sharded x{world}, no data-path collective",
                "pipeline": "synchronous" if args.no_pipeline else "2 batches in flight (greedy of batch i overlaps encoder of i+1)",
                "weights": "seeded random init of the zipformer-large architecture (no checkpoints available)",
            },
            "roofline": {
                "kernel": "gemm_f32_mfma (all Linear / pointwise-conv / implicit-conv / attention-apply launches)",
                "bound": "mfma",
                "achieved": round(ach, 2),
                "peak": F32_MFMA_PEAK_TFLOPS,
                "unit": "TFLOP/s",
                "frac": round(ach / F32_MFMA_PEAK_TFLOPS, 4),
                "traffic": traffic,
                "traffic_note": traffic_note,
                "launches_per_step": it["gemm_launches"],
                "flops_per_step": it["gemm_flops"],
                "avg_launch_us": round(it["gemm_ms"] * 1e3 / max(it["gemm_launches"], 1), 2),
                "all_matrix_flops_per_step": it["total_flops"],
            },
            "stages_ms_one_synchronous_pass": {k: round(stages[k], 3) for k in ("total_ms", "fbank_ms", "pad_ms", "encoder_ms", "greedy_ms", "d2h_ms")},
            "tokens_emitted_last_step": int(sum(len(r[0]) for r in res)),
        }
        if world == 1 and not args.no_cpu_baseline:
            cb, _ = cpu_baseline(weights, args.cpu_utts, secs)
            out["cpu_baseline"] = cb
        print(json.dumps(out), flush=True)
    model.device_free(dptr)
    model.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
