#!/usr/bin/env python3
"""Benchmark of the K2TransducerAsr offline hot path on MI355X.

Default workload (BASELINE.json configs[1]): zipformer-large-en offline greedy, batch = 32
synthetic 10 s utterances per GPU.  One step = one pass of the whole path over the rank's shard
of utterances: fbank -> PadSequence (+19 frames, log-floor) -> Zipformer2-large encoder ->
on-device greedy search -> token arrays back in host memory.

Metric: RTFx = audio seconds / wall seconds (inverse of the reference's printed
`rtf = elapsed_ms / total_duration_ms`, K2TransducerAsr.Examples/OfflineRecognizer.cs:185-189),
whole job over all ranks.  `value` is timed with the samples already resident in HBM;
`value_from_host_memory` is the same K steps with the samples in page-locked host memory and
the H2D copy inside the pipeline (the reference's protocol: float[] in, text out).

Multi-GPU: one process per GPU, each rank decodes its own contiguous shard of the utterance list
exactly as the reference would decode that shard as its own GetResults batches; no collective on
the data path (RCCL / gloo carry the timing barrier, the max-over-ranks and the KB-sized gather
of the token lists).  `python bench.py --gpus N` with no WORLD_SIZE in the environment starts the
N ranks itself (before any GPU call); under torch.distributed.run it is one of the ranks.

    python bench.py --gpus 1 --steps 10 --warmup 2
    python bench.py --gpus 8 --total-utts 256 --beam 4                              # configs[2]
    python bench.py --gpus 8 --preset conformer-zh --total-utts 64 --batch 8 --seconds 30   # configs[4]
    python bench.py --gpus 2 --dist-backend gloo                                    # rehearsal on one GPU
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
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


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--preset", default=PRESET)
    ap.add_argument("--batch", type=int, default=BATCH, help="utterances per GetResults batch")
    ap.add_argument("--seconds", type=float, default=UTT_SECONDS)
    ap.add_argument("--total-utts", type=int, default=0,
                    help="utterances of the whole job, sharded contiguously over the ranks (strong scaling); "
                         "default = --batch per rank (weak scaling)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-leg", action="store_true", help="skip the second timing with samples in host memory")
    ap.add_argument("--no-pipeline", action="store_true", help="one synchronous call per batch (no batch overlap)")
    ap.add_argument("--cpu-utts", type=int, default=0, help="utterances of the CPU-baseline sample (default: one batch)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the bounded configs[2] / [3] / [4] legs that follow the headline run on one GPU")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, one rank per GPU) or gloo (rehearsal of the N > 1 path on one GPU)")
    ap.add_argument("--beam", type=int, default=0, help="0 = greedy_search (headline metric); K = modified_beam_search with beam K (BASELINE configs[2])")
    ap.add_argument("--depth", type=int, default=0, help="batches in flight (0: 2; 3 is allowed with --beam, and with K2HIP_PIPE_MODE=2 for the greedy search)")
    ap.add_argument("--dump-results", default="", help="rank 0 writes every utterance's (tokens, timestamps) of the last step here (JSON)")
    ap.add_argument("--launch-check", action="store_true", help="rendezvous + shard bookkeeping only; no GPU work (CPU test of the launcher)")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------
# launcher: `python bench.py --gpus N` starts its own ranks.  Nothing here may touch the GPU: the
# children are separate programs, started before any HIP call of this process.
# ------------------------------------------------------------------------------------------------
def launch_ranks(n: int) -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    alive = set(range(n))
    while alive:
        for r in sorted(alive):
            code = procs[r].poll()
            if code is None:
                continue
            alive.discard(r)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                log(f"[bench] rank {r} exited with {code}; stopping the other ranks")
                for o in alive:
                    procs[o].terminate()     # exactly the processes started above
        time.sleep(0.05)
    return rc


def ensure_weights(path, preset, rank, barrier):
    from k2transducerasr_amd.synth import write_synthetic_model

    if rank == 0 and not os.path.exists(path):
        tmp = path + f".tmp{os.getpid()}"
        t = time.time()
        write_synthetic_model(tmp, preset)
        os.replace(tmp, path)
        log(f"[bench] wrote synthetic {preset} weights to {path} in {time.time() - t:.1f}s")
    barrier()


def cpu_baseline(weights, n_utts, seconds, beam=0):
    """The CPU restatement (oracle/, 'port') of the same path on this box's host cores, on a bounded sample of the same workload:
    the first `n_utts` utterances as ONE GetResults batch.  Returns the baseline record and the oracle's (tokens, timestamps) per
    utterance, which the caller holds against what the GPU returned for the same batch (`oracle_match`)."""
    from k2transducerasr_amd.synth import synth_utterance
    from oracle import Oracle

    cores = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(len(os.sched_getaffinity(0)), 16)
    os.environ["OMP_NUM_THREADS"] = str(cores)
    ora = Oracle(weights)
    utts = [synth_utterance(u, seconds) for u in range(n_utts)]

    def run():
        feats = [ora.fbank(u) for u in utts]
        if beam > 0:   # icefall modified_beam_search over the oracle's own encoder output (oracle/k2_oracle_beam.c)
            x = ora.pad_sequence(feats).reshape(len(feats), -1, ora.feature_dim)
            return ora.modified_beam_search(ora.encoder(x), beam)
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
                  f"(C + OpenMP restatement{', modified beam search beam=%d' % beam if beam else ''}; the reference's ONNXRuntime path cannot run here), {dt:.2f} s wall",
    }, res


def pmc_mfma_busy():
    """MFMA-pipe utilisation of the GEMM kernels from the committed rocprofv3 --pmc pass (SQ_VALU_MFMA_BUSY_CYCLES against
    SQ_BUSY_CU_CYCLES; counters need their own profiler run, so this cannot be read live)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_gemm_mfma_busy.json")))
    if not files:
        return None, "no MFMA PMC summary under profiles/"
    with open(files[-1]) as f:
        d = json.load(f)
    return d["mfma_busy"], f"from {os.path.relpath(files[-1], ROOT)}: {d['method']}"


def run_secondary(timeout_s=240):
    """Bounded legs for BASELINE configs[2], [3], [4] (one GPU's share each), run as child programs AFTER the headline model has
    been closed, one at a time; each child prints its own JSON line (with its own cpu_baseline and oracle_match), condensed here.
    (16 timed steps each: the timed region ends with the last batch's search running alone, a 5 - 10 ms tail that 6 steps spread thinly.)"""
    here = os.path.dirname(os.path.abspath(__file__))
    legs = {
        "beam4_c2_shard": [sys.executable, os.path.join(here, "bench.py"), "--beam", "4", "--steps", "16", "--warmup", "2",
                           "--no-host-leg", "--no-secondary"],
        "conformer_zh_c4_shard": [sys.executable, os.path.join(here, "bench.py"), "--preset", "conformer-zh", "--batch", "8",
                                  "--seconds", "30", "--steps", "16", "--warmup", "2", "--no-host-leg", "--no-secondary"],
        "streaming_c3": [sys.executable, os.path.join(here, "bench_streaming.py"), "--check", "8"],
    }
    out = {}
    for name, cmd in legs.items():
        t = time.time()
        try:
            p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout_s, text=True)
            line = [l for l in p.stdout.splitlines() if l.startswith("{")]
            if not line:
                out[name] = {"error": f"exit {p.returncode}: {p.stderr[-400:]}"}
                continue
            d = json.loads(line[-1])
            out[name] = {
                "metric": d["metric"], "value": d["value"], "unit": d["unit"],
                "ms_per_step": d.get("ms_per_step", d.get("ms_per_chunk_step")),
                "workload": d["config"]["workload"],
                "roofline": {k: d["roofline"][k] for k in ("bound", "achieved", "peak", "unit", "frac") if k in d.get("roofline", {})},
                "cpu_baseline": d.get("cpu_baseline"),
                "oracle_match": d.get("oracle_match"),
                "emission_rate": d.get("emission_rate"),
                "exit_code": p.returncode,
                "wall_s": round(time.time() - t, 1),
            }
            if "tick" in d.get("roofline", {}):
                out[name]["roofline"]["tick_frac_of_mfma_peak"] = d["roofline"]["tick"]["frac_of_mfma_peak"]
        except subprocess.TimeoutExpired:
            out[name] = {"error": f"timed out after {timeout_s} s"}
    return out


def baseline_config_of(args, world, total):
    """which BASELINE.json config (if any) this invocation is"""
    if args.preset == PRESET and args.beam == 0 and args.batch == 32 and abs(args.seconds - 10.0) < 1e-9 and not args.total_utts:
        return "BASELINE.json configs[1]" + ("" if world == 1 else f" per GPU, x{world} GPUs")
    if args.preset == PRESET and args.beam == 4 and abs(args.seconds - 10.0) < 1e-9:
        if total == 256 and world == 8:
            return "BASELINE.json configs[2]"
        return f"BASELINE.json configs[2] at {total} utterances on {world} GPU(s) (the config is 256 on 8)"
    if args.preset == "conformer-zh" and args.beam == 0 and abs(args.seconds - 30.0) < 1e-9:
        if total == 64 and world == 8:
            return "BASELINE.json configs[4]"
        return f"BASELINE.json configs[4] at {total} utterances on {world} GPU(s) (the config is 64 on 8)"
    return "not a BASELINE.json config"


def main():
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        log(f"[bench] WORLD_SIZE={world} but --gpus {args.gpus}: refusing to report a line for a different job size")
        sys.exit(2)
    if os.environ.get("K2HIP_BENCH_FAIL_RANK") == str(rank):  # launcher test hook: this rank dies before the rendezvous
        sys.exit(3)
    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist

        if args.dist_backend == "nccl" and not args.launch_check:
            if local_rank >= torch.cuda.device_count():
                log(f"[bench] rank {rank}: no GPU {local_rank} on this node ({torch.cuda.device_count()} visible); "
                    "use --dist-backend gloo to rehearse N ranks on fewer GPUs")
                sys.exit(4)
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend="gloo")

    def barrier():
        if dist is not None:
            dist.barrier()

    from k2transducerasr_amd.shard import batches_of, gather_results, max_over_ranks, shard_range

    B, secs = args.batch, args.seconds
    total = args.total_utts or B * world
    lo, hi = shard_range(total, world, rank)
    my_batches = batches_of(lo, hi, B)          # [(first utterance id, count)]: the GetResults batches of this rank
    if args.launch_check:
        got = gather_results(dist, [(rank, lo, hi, my_batches)], world, rank)
        barrier()
        if rank == 0:
            print(json.dumps({"launch_check": True, "n_gpus": world, "total_utts": total, "shards": got}), flush=True)
        if dist is not None:
            dist.destroy_process_group()
        return

    import k2transducerasr_amd as pkg
    from k2transducerasr_amd.synth import synth_utterance

    weights = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"k2hip_bench_{args.preset}.k2w")
    ensure_weights(weights, args.preset, rank, barrier)

    n_dev = pkg.load_library().k2hip_device_count()
    device = local_rank if args.dist_backend == "nccl" else local_rank % max(n_dev, 1)  # gloo rehearsal: ranks may share a GPU
    model = pkg.Model(weights, device)  # no fallback: raises without a GPU / library
    if args.beam > 0:
        model.set_decoding_method("modified_beam_search", args.beam)
    # batches in flight (k2hip.h K2HIP_MAX_BATCHES_IN_FLIGHT = 3): two for the greedy search (it hides under the next encoder); three for
    # the beam search, whose per-frame launches take longer than an encoder pass when they share the GPU with one
    # (with the beam search as one kernel per batch a third batch in flight no longer pays: 14.90 - 14.93 ms at depth 2, 14.95 - 15.01 at 3)
    depth = args.depth if args.depth > 0 else 2
    n_each = int(round(secs * 16000))
    # utterance u is the same signal whichever rank decodes it (seed = u)
    host, dev = [], []
    for first, cnt in my_batches:
        h = model.host_alloc((cnt, n_each))
        for i in range(cnt):
            h[i] = synth_utterance(first + i, secs)
        d = model.device_alloc(h.nbytes)
        model.device_upload(d, h)
        host.append(h)
        dev.append(d)
    nb = len(my_batches)

    def submit(i, from_host):
        cnt = my_batches[i][1]
        return model.offline_submit_samples(host[i], None) if from_host else model.offline_submit_samples_dev(dev[i], n_each, cnt)

    def run_steps(n, from_host=False):
        """n passes over the rank's batches, software-pipelined `depth` deep: the next batch(es) are submitted before the oldest one's
        tokens are collected, so its encoder (and, from host memory, its H2D copy) overlaps that batch's search.  Every batch's
        tokens are back in host memory before this returns; the last pass's results are returned in utterance order."""
        last = [None] * nb
        if n == 0 or nb == 0:
            return last
        if args.no_pipeline:
            for _ in range(n):
                for i in range(nb):
                    last[i] = model.offline_wait(submit(i, from_host))
            return last
        seq = [i for _ in range(n) for i in range(nb)]
        pending = []  # (batch index, ticket), oldest first: `depth` batches in flight
        for k in range(len(seq)):
            pending.append((seq[k], submit(seq[k], from_host)))
            if len(pending) == depth:
                i0, tk = pending.pop(0)
                last[i0] = model.offline_wait(tk)
        for i0, tk in pending:
            last[i0] = model.offline_wait(tk)
        return last

    def timed(from_host):
        run_steps(args.warmup, from_host)
        model.synchronize()
        barrier()
        t0 = time.perf_counter()
        r = run_steps(args.steps, from_host)
        model.synchronize()
        barrier()
        el = time.perf_counter() - t0
        return r, max_over_ranks(dist, el, device="cuda" if (dist is not None and args.dist_backend == "nccl") else None)

    res, elapsed = timed(False)
    host_elapsed = None
    if not args.no_host_leg:
        res_h, host_elapsed = timed(True)
        assert res_h == res, "results from host memory and from device memory differ"
    local = [r for batch in res for r in batch]
    allres = gather_results(dist, local, world, rank)   # rank order == utterance order

    # one synchronous pass over the first batch: per-stage HIP-event timings + pipelined == synchronous check
    stages = it = None
    if nb:
        res_sync = model.offline_greedy_from_samples_dev(dev[0], n_each, my_batches[0][1])
        stages = model.timing()
        assert res_sync == res[0], "pipelined and synchronous results differ"
        # roofline of the dominant kernel (fp32 MFMA GEMM): one extra instrumented pass over the same batch, HIP events
        # recorded around every GEMM launch on the engine's own stream (no per-launch sync, launches stay back to back).
        model.set_instrument(True)
        model.offline_greedy_from_samples_dev(dev[0], n_each, my_batches[0][1])
        it = model.timing()
        model.set_instrument(False)

    bad_match = False
    if rank == 0:
        assert len(allres) == total, f"gathered {len(allres)} results for {total} utterances"
        audio = args.steps * total * secs
        value = audio / elapsed
        ach = it["gemm_flops"] / (it["gemm_ms"] * 1e-3) / 1e12 if it and it["gemm_ms"] > 0 else 0.0
        default_workload = args.preset == PRESET and B == BATCH and abs(secs - UTT_SECONDS) < 1e-9 and args.beam == 0 and not args.total_utts
        traffic, traffic_note = pmc_traffic() if default_workload else (None, "PMC passes exist for the default workload only")
        method = "greedy" if args.beam == 0 else f"modified-beam-search beam={args.beam}"
        tprime = max(1, model.encoder_out_frames(model.fbank_num_frames(n_each) + 19))
        n_tok = int(sum(len(r[0]) for r in allres))
        out = {
            "metric": "RTFx (audio-sec/wall-sec) offline Zipformer greedy" if default_workload else f"RTFx (audio-sec/wall-sec) offline {args.preset} {method}",
            "value": round(value, 1),
            "unit": "audio-sec/wall-sec",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "strong" if args.total_utts else "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"{args.preset} offline {method}, {total} synthetic {secs:g} s utterances per step in GetResults batches of "
                            f"{B} ({baseline_config_of(args, world, total)}); samples resident in HBM, tokens returned to host",
                "batch": B,
                "total_utts_per_step": total,
                "utt_seconds": secs,
                "parallelism": f"utterance-sharded x{world} ({'one rank per GPU, RCCL' if args.dist_backend == 'nccl' else 'gloo rehearsal, ranks share GPUs'}), "
                               "no data-path collective",
                "pipeline": "synchronous" if args.no_pipeline else ("2 batches in flight (search of batch i overlaps encoder of i+1)" if depth == 2 else
                             "3 batches in flight (the beam searches of batches i and i+1 overlap each other and the encoder of i+2)"),
                "weights": f"seeded random init of the {args.preset} architecture (no checkpoints available)",
            },
            "tokens_emitted_per_step": n_tok,
            "emission_rate": round(n_tok / (total * tprime), 4),
            "results_sha1": hashlib.sha1(json.dumps(allres).encode()).hexdigest()[:16],
        }
        if host_elapsed is not None:
            out["value_from_host_memory"] = round(audio / host_elapsed, 1)
            out["ms_per_step_from_host_memory"] = round(host_elapsed / args.steps * 1e3, 3)
        if it:
            out["roofline"] = {
                "kernel": "gemm_f32_mfma (all Linear / pointwise-conv / implicit-conv / attention-apply launches)",
                "bound": "mfma",
                "achieved": round(ach, 2),
                "peak": F32_MFMA_PEAK_TFLOPS,
                "unit": "TFLOP/s",
                "frac": round(ach / F32_MFMA_PEAK_TFLOPS, 4),
                "traffic": traffic,
                "traffic_note": traffic_note,
                "launches_per_batch": it["gemm_launches"],
                "flops_per_batch": it["gemm_flops"],
                "avg_launch_us": round(it["gemm_ms"] * 1e3 / max(it["gemm_launches"], 1), 2),
                "all_matrix_flops_per_batch": it["total_flops"],
            }
            out["stages_ms_one_synchronous_batch"] = {k: round(stages[k], 3) for k in ("total_ms", "fbank_ms", "pad_ms", "encoder_ms", "greedy_ms", "d2h_ms")}
            mb, mb_note = pmc_mfma_busy() if default_workload else (None, "PMC passes exist for the default workload only")
            out["roofline"]["mfma_busy"] = mb
            out["roofline"]["mfma_busy_note"] = mb_note
        if world == 1 and not args.no_cpu_baseline:
            n_cpu = args.cpu_utts or my_batches[0][1]
            cb, ores = cpu_baseline(weights, n_cpu, secs, args.beam)
            out["cpu_baseline"] = cb
            # the oracle's results for the sample against what the timed legs returned for the same batch (outside the timed region)
            if my_batches[0] == (0, n_cpu):
                exact = sum(1 for g, w in zip(allres[:n_cpu], ores) if [list(g[0]), list(g[1])] == [list(w[0]), list(w[1])])
                out["oracle_match"] = {"streams": n_cpu, "exact": exact,
                                       "what": "tokens and timestamps of the timed legs' first batch == oracle/ on the same batch"}
                bad_match = exact < n_cpu
            else:
                out["oracle_match"] = {"streams": 0, "exact": 0, "what": "the CPU sample is not one of the timed batches"}
        if args.dump_results:
            with open(args.dump_results, "w") as f:
                json.dump({"results": allres, "batches_per_rank": nb, "n_gpus": world}, f)
    for h, d in zip(host, dev):
        model.device_free(d)
        model.host_free(h)
    model.close()
    if rank == 0:
        if world == 1 and default_workload and not args.no_secondary:
            out["secondary"] = run_secondary()   # the GPU is free again: one child program per leg
            bad_match = bad_match or any(("error" in v) or v.get("exit_code") for v in out["secondary"].values())
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()
    if rank == 0 and bad_match:
        log("[bench] oracle_match failed (or a secondary leg failed): see the JSON line")
        sys.exit(5)


if __name__ == "__main__":
    main()
