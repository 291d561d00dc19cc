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


def _worker(rank, world, port, total, q, port2=None):
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
    s_all, e_all = parallel.all_gather_chunks(torch.from_numpy(seg[b:e]), torch.from_numpy(emb[b:e]), total,
                                              shard, torch.device("cpu"))
    assert isinstance(s_all, torch.Tensor) and s_all.dtype == torch.uint8 and e_all.dtype == torch.float32
    ok = np.array_equal(s_all.numpy(), seg.astype(np.uint8)) and \
        np.array_equal(e_all.numpy(), emb, equal_nan=True)
    # multi-file exchange (configs[4]): rank 0 contributes 2 files, rank 1 one, different lengths
    lens = [[5, 2], [7]]
    files = [(rng.uniform(size=(c, 37, 3)) < 0.4, rng.standard_normal((c, 3, 16)).astype(np.float32))
             for per in lens for c in per]
    first = sum(len(per) for per in lens[:rank])
    mine = files[first:first + len(lens[rank])]
    recs = [parallel.pack_records(torch.from_numpy(sg), torch.from_numpy(em)) for sg, em in mine]
    def check(everything, lens, files):
        good, k = True, 0
        for r, per in enumerate(everything):
            assert len(per) == len(lens[r])
            for rec in per:
                sg, em = parallel.unpack_records(rec, 37, 3, 16)
                good = good and np.array_equal(sg.numpy(), files[k][0].astype(np.uint8)) \
                    and np.array_equal(em.numpy(), files[k][1])
                k += 1
        return good

    # first exchange of the process group: no agreed capacity yet -> header round + data round
    before = parallel.collectives_issued
    ok = ok and check(parallel.all_gather_files(recs, shard, torch.device("cpu")), lens, files)
    assert parallel.collectives_issued - before == 2
    # steady state: ONE all-gather (the chunk counts travel in the header of the same buffer)
    before = parallel.collectives_issued
    ok = ok and check(parallel.all_gather_files(recs, shard, torch.device("cpu")), lens, files)
    assert parallel.collectives_issued - before == 1
    # a rank without files (record size given), still one
    before = parallel.collectives_issued
    got = parallel.all_gather_files(recs if rank == 0 else [], shard, torch.device("cpu"), record_bytes=37 * 3 + 4 * 3 * 16)
    assert parallel.collectives_issued - before == 1 and len(got[1]) == 0 and len(got[0]) == 2
    # one rank outgrows the agreed capacity (512 chunks): every rank sees the overflow flag and repeats once
    big = (rng.uniform(size=(700, 37, 3)) < 0.4, rng.standard_normal((700, 3, 16)).astype(np.float32))
    lens2, files2 = [[5, 2], [700]], files[:2] + [big]
    mine2 = files2[:2] if rank == 0 else [big]
    recs2 = [parallel.pack_records(torch.from_numpy(sg), torch.from_numpy(em)) for sg, em in mine2]
    before = parallel.collectives_issued
    ok = ok and check(parallel.all_gather_files(recs2, shard, torch.device("cpu")), lens2, files2)
    assert parallel.collectives_issued - before == 2
    before = parallel.collectives_issued
    ok = ok and check(parallel.all_gather_files(recs2, shard, torch.device("cpu")), lens2, files2)
    assert parallel.collectives_issued - before == 1
    # more files on one rank than the header holds (60): the header capacity is part of the agreement and grows
    # through the same flag round -- nobody raises before the collective, nobody is left waiting in it
    many = [(rng.uniform(size=(1 + (i % 3), 37, 3)) < 0.4, rng.standard_normal((1 + (i % 3), 3, 16)).astype(np.float32))
            for i in range(70)]
    lens3, files3 = [[5, 2], [m[0].shape[0] for m in many]], files[:2] + many
    mine3 = files3[:2] if rank == 0 else many
    recs3 = [parallel.pack_records(torch.from_numpy(sg), torch.from_numpy(em)) for sg, em in mine3]
    before = parallel.collectives_issued
    ok = ok and check(parallel.all_gather_files(recs3, shard, torch.device("cpu")), lens3, files3)
    assert parallel.collectives_issued - before == 2
    before = parallel.collectives_issued
    ok = ok and check(parallel.all_gather_files(recs3, shard, torch.device("cpu")), lens3, files3)
    assert parallel.collectives_issued - before == 1
    # a rank that has neither a file nor the record size contributes nothing (and stalls nobody)
    got = parallel.all_gather_files(recs if rank == 0 else [], shard, torch.device("cpu"))
    ok = ok and len(got[1]) == 0 and len(got[0]) == 2
    # ranks that announce different record sizes: EVERY rank raises, after the collective
    odd = [torch.zeros((2, 99 if rank == 0 else 98), dtype=torch.uint8)]
    try:
        parallel.all_gather_files(odd, shard, torch.device("cpu"))
        ok = False
    except ValueError as err:
        ok = ok and "different record sizes" in str(err)
    # the label broadcast of a joint job (pipelining.pipelined_owned): numpy arrays / None from one rank to all
    side = dist.new_group(list(range(world)), backend="gloo")
    for src in range(world):
        obj = (np.arange(12, dtype=np.int8).reshape(4, 3) * (src + 1), rng.standard_normal((2, 5))) if src == rank else None
        got = parallel.broadcast_object(obj, src, shard, side, torch.device("cpu"))
        ok = ok and got[0].dtype == np.int8 and np.array_equal(got[0], np.arange(12, dtype=np.int8).reshape(4, 3) * (src + 1)) \
            and got[1].shape == (2, 5)
    ok = ok and parallel.broadcast_object(None if rank == 1 else "x", 1, shard, side, torch.device("cpu")) is None
    # the choice between two collective schedules is agreed on: true only when every rank says so
    ok = ok and parallel.agree_all(True, shard, torch.device("cpu")) is True
    ok = ok and parallel.agree_all(rank == 0, shard, torch.device("cpu")) is False
    ok = ok and parallel.agree_all(False, shard, torch.device("cpu")) is False
    q.put((rank, b, e, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()
    if port2 is not None:
        # a NEW process group of the same process starts without an agreement, whatever id() it gets
        os.environ["MASTER_PORT"] = str(port2)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        shard = parallel.shard_from_env()
        before = parallel.collectives_issued
        good = check(parallel.all_gather_files(recs, shard, torch.device("cpu")), lens, files)
        assert parallel.collectives_issued - before == 2 and good
        assert len(parallel._agreed) == 1
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("total", [21, 8, 3])
def test_all_gather_chunks_gloo(total):
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port2 = _free_port() if total == 21 else None
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, q, port2)) for r in range(world)]
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
