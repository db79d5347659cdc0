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
