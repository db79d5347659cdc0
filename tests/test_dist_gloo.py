"""world_size-2 CPU (gloo) test of the multi-GPU plumbing: contiguous utterance shards,
no data-path collective, rank-ordered gather of the token lists, max-over-ranks timing."""
import os
import socket
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_range_partitions_exactly():
    from k2transducerasr_amd.shard import shard_range
    for n in (0, 1, 7, 64, 256, 257):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    assert shard_range(256, 8, 3) == (96, 128)  # BASELINE config 3: 256 utterances, 32 per GPU
    assert shard_range(64, 8, 7) == (56, 64)    # config 5: 64 utterances, 8 per GPU
    with pytest.raises(ValueError):
        shard_range(4, 2, 2)


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from k2transducerasr_amd.shard import gather_results, max_over_ranks, shard_range
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n = 11
    lo, hi = shard_range(n, world, rank)
    # stand-in for "decode my shard": a deterministic function of the utterance id
    local = [([u, u + 1], [0, u]) for u in range(lo, hi)]
    allr = gather_results(dist, local, world, rank)
    t = max_over_ranks(dist, 1.0 + rank)
    dist.barrier()
    q.put((rank, allr, t))
    dist.destroy_process_group()


def test_two_ranks_gloo():
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = [([u, u + 1], [0, u]) for u in range(11)]
    for rank, allr, t in got:
        assert allr == want          # rank order == utterance order, nothing lost or duplicated
        assert t == 2.0              # max over ranks


# ---- bench.py's own launcher: `python bench.py --gpus N` with no WORLD_SIZE starts N ranks itself -------------------
def _bench(*argv, env=None, timeout=240):
    import subprocess
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], env=e, capture_output=True, text=True, timeout=timeout)


def test_bench_spawns_its_own_ranks():
    """No GPU needed: --launch-check stops after the rendezvous and the shard bookkeeping."""
    import json
    r = _bench("--gpus", "2", "--dist-backend", "gloo", "--launch-check", "--total-utts", "7", "--batch", "2")
    assert r.returncode == 0, r.stderr
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["total_utts"] == 7
    # rank order == utterance order; every utterance in exactly one batch
    assert line["shards"] == [[0, 0, 4, [[0, 2], [2, 2]]], [1, 4, 7, [[4, 2], [6, 1]]]]
    r = _bench("--gpus", "3", "--dist-backend", "gloo", "--launch-check")   # weak default: one batch of 32 per rank
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert r.returncode == 0 and line["n_gpus"] == 3 and [s[1:3] for s in line["shards"]] == [[0, 32], [32, 64], [64, 96]]


def test_bench_fails_when_a_rank_fails_or_the_world_is_wrong():
    r = _bench("--gpus", "2", "--dist-backend", "gloo", "--launch-check", env={"K2HIP_BENCH_FAIL_RANK": "1"}, timeout=120)
    assert r.returncode != 0 and r.stdout.strip() == ""           # no JSON line for a job that lost a rank
    # under an external launcher whose world differs from --gpus the bench refuses instead of mislabelling the line
    r = _bench("--gpus", "4", "--launch-check", env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode == 2 and r.stdout.strip() == ""


def test_batches_of():
    from k2transducerasr_amd.shard import batches_of, shard_range
    assert batches_of(0, 0, 4) == []
    assert batches_of(3, 10, 4) == [(3, 4), (7, 3)]
    for total, world, b in ((256, 8, 32), (64, 8, 8), (7, 2, 3)):
        ids = []
        for r in range(world):
            for first, cnt in batches_of(*shard_range(total, world, r), b):
                ids += list(range(first, first + cnt))
        assert ids == list(range(total))


@pytest.mark.gpu
def test_two_ranks_through_the_product_equal_one_process(tmp_path):
    """Two gloo ranks sharing device 0, each decoding its shard through libk2hip.so (bench.py's own launcher), against one
    process decoding the same batches: rank-order concatenation == the single-process result, for greedy and for beam 4."""
    import json
    common = ["--preset", "zipformer2-tiny-test", "--total-utts", "6", "--batch", "3", "--seconds", "1.2", "--steps", "1",
              "--warmup", "0", "--no-cpu-baseline"]
    for extra in ([], ["--beam", "4"]):
        one, two = str(tmp_path / "one.json"), str(tmp_path / "two.json")
        r1 = _bench("--gpus", "1", *common, *extra, "--dump-results", one)
        assert r1.returncode == 0, r1.stderr[-2000:]
        r2 = _bench("--gpus", "2", "--dist-backend", "gloo", *common, *extra, "--dump-results", two)
        assert r2.returncode == 0, r2.stderr[-2000:]
        l1, l2 = (json.loads(r.stdout.strip().splitlines()[-1]) for r in (r1, r2))
        assert l1["n_gpus"] == 1 and l2["n_gpus"] == 2 and l2["scaling"] == "strong"
        a, b = json.load(open(one)), json.load(open(two))
        assert a["batches_per_rank"] == 2 and b["batches_per_rank"] == 1
        assert len(a["results"]) == 6 and a["results"] == b["results"]
        assert l1["results_sha1"] == l2["results_sha1"] and l1["tokens_emitted_per_step"] > 0


def _bench_streaming(*argv, env=None, timeout=600):
    import subprocess
    e = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench_streaming.py"), *argv], env=e, capture_output=True, text=True, timeout=timeout)


def test_bench_streaming_refuses_a_world_that_differs_from_gpus():
    r = _bench_streaming("--gpus", "4", env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"}, timeout=120)
    assert r.returncode == 2 and "refusing" in r.stderr


@pytest.mark.gpu
def test_streaming_two_ranks_equal_one_process(tmp_path):
    """SURVEY 8(e), streaming: streams pinned to a GPU at CreateOnlineStream time, stream u on rank u mod N.  bench_streaming.py's own
    launcher with two gloo ranks sharing device 0 (five streams: three on rank 0, two on rank 1) against one process with all five:
    every stream's tokens and timestamps equal, in stream order; the oracle check of rank 0's first streams (ids 0 and 2) passes."""
    import json
    common = ["--preset", "zipformer2-streaming-tiny-test", "--streams", "5", "--seconds", "3", "--no-cpu-baseline"]
    one, two = str(tmp_path / "one.json"), str(tmp_path / "two.json")
    r1 = _bench_streaming("--gpus", "1", *common, "--dump-results", one)
    assert r1.returncode == 0, r1.stderr[-2000:]
    r2 = _bench_streaming("--gpus", "2", "--dist-backend", "gloo", *common, "--check", "2", "--dump-results", two)
    assert r2.returncode == 0, (r2.stdout[-1000:], r2.stderr[-2000:])
    l1, l2 = (json.loads(r.stdout.strip().splitlines()[-1]) for r in (r1, r2))
    assert l1["n_gpus"] == 1 and l2["n_gpus"] == 2 and l2["oracle_match"] == {**l2["oracle_match"], "streams": 2, "exact": 2}
    a, b = json.load(open(one)), json.load(open(two))
    assert len(a["results"]) == 5 and a["results"] == b["results"] and l1["tokens"] == l2["tokens"] > 0
