"""world_size = 2 over gloo (CPU): chunk-range sharding + the single all-gather of per-chunk results
(SURVEY.md section 8e).  Checks that every rank reassembles exactly the unsharded tensors."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from pyannote_audio_amd import parallel
    rng = np.random.default_rng(0)
    seg = (rng.uniform(size=(total, 37, 3)) < 0.4).astype(np.float32)
    emb = rng.standard_normal((total, 3, 16)).astype(np.float32)
    emb[min(3, total - 1), 1] = np.nan
    shard = parallel.shard_from_env()
    assert (shard.rank, shard.world_size) == (rank, world)
    b, e = parallel.chunk_range(total, shard)
    s_all, e_all = parallel.all_gather_chunks(seg[b:e], emb[b:e], total, shard, torch.device("cpu"))
    ok = np.array_equal(s_all, seg) and np.array_equal(e_all, emb, equal_nan=True)
    q.put((rank, b, e, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("total", [21, 8, 3])
def test_all_gather_chunks_gloo(total):
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[3] for r in res)
    assert res[0][1] == 0 and res[0][2] == res[1][1] and res[1][2] == total


def test_chunk_range_partitions():
    from pyannote_audio_amd.parallel import Shard, chunk_range
    for total in (1, 7, 3591, 3592):
        for world in (1, 2, 4, 8):
            if world == 1:
                assert chunk_range(total, Shard()) is None
                continue
            ranges = [chunk_range(total, Shard(r, world)) for r in range(world)]
            assert ranges[0][0] == 0 and ranges[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
            sizes = [b - a for a, b in ranges]
            assert max(sizes) - min(sizes) <= 1
