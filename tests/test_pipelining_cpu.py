"""Host logic of the pipelined batch forms (pyannote_audio_amd/pipelining.py): order, the gate between the tail of
item i and the front of item i+1, the "alone" flag of the last item, and that nothing strands the worker thread."""
import threading
import time

import pytest

from pyannote_audio_amd.pipelining import pipelined


def test_results_in_order_and_tail_waits_for_the_release_of_the_next_front():
    log, lock = [], threading.Lock()

    def note(*event):
        with lock:
            log.append(event)

    def front(item, release):
        note("front-begin", item)
        time.sleep(0.02)              # "segmentation": the previous tail must not have started yet
        note("release", item)
        release()
        time.sleep(0.25)              # "embeddings": the previous tail runs beside this (long enough for a loaded machine)
        note("front-end", item)
        return item * 10

    def tail(state, alone):
        note("tail-begin", state // 10, alone)
        time.sleep(0.01)
        note("tail-end", state // 10)
        return state + 1

    out = list(pipelined(range(4), front, tail))
    assert out == [(0, 1), (1, 11), (2, 21), (3, 31)]
    pos = {e: i for i, e in enumerate(log)}
    for i in range(3):
        # tail i starts after front i+1 released, and (here) finishes before front i+1 does: they overlap
        assert pos[("release", i + 1)] < pos[("tail-begin", i, False)] < pos[("front-end", i + 1)]
    assert ("tail-begin", 3, True) in pos            # the last item: nothing beside its tail
    assert pos[("front-end", 3)] < pos[("tail-begin", 3, True)]


def test_a_front_that_never_releases_still_lets_the_previous_tail_run():
    started = []
    out = list(pipelined("ab", lambda item, release: item, lambda s, alone: started.append((s, alone)) or s.upper()))
    assert out == [("a", "A"), ("b", "B")] and started == [("a", False), ("b", True)]


def test_empty_and_single_item():
    assert list(pipelined([], lambda i, r: i, lambda s, a: s)) == []
    assert list(pipelined([7], lambda i, r: i, lambda s, alone: (s, alone))) == [(7, (7, True))]


def test_failing_front_does_not_strand_the_tail():
    done = []

    def front(item, release):
        if item == 1:
            raise ValueError("bad file")
        return item

    def tail(state, alone):
        done.append(state)
        return state

    gen = pipelined(range(3), front, tail, gate_timeout=30.0)
    t0 = time.perf_counter()
    with pytest.raises(ValueError, match="bad file"):
        list(gen)
    assert done == [0] and time.perf_counter() - t0 < 5.0       # the gate was opened, not timed out


def test_tail_exception_reaches_the_consumer_and_abandoned_generator_terminates():
    def tail(state, alone):
        if state == 1:
            raise RuntimeError("clustering failed")
        return state

    gen = pipelined(range(3), lambda i, r: i, tail)
    assert next(gen) == (0, 0)
    with pytest.raises(RuntimeError, match="clustering failed"):
        next(gen)
    gen2 = pipelined(range(5), lambda i, r: i, lambda s, a: s, gate_timeout=30.0)
    assert next(gen2) == (0, 0)
    t0 = time.perf_counter()
    gen2.close()                                                  # consumer walks away
    assert time.perf_counter() - t0 < 5.0


def test_a_gate_that_times_out_is_logged(caplog):
    """the safety net of the gate must not act silently (the tail then runs beside the next item's first stage)"""
    import logging
    import threading
    from pyannote_audio_amd.pipelining import pipelined
    hold = threading.Event()

    def front(item, release):
        if item == 1:
            hold.wait(timeout=2.0)   # the front of item 1 neither releases nor returns in time
        return item

    def tail(state, alone):
        if state == 0:
            hold.set()
        return state * 10

    with caplog.at_level(logging.WARNING, logger="pyannote_audio_amd.pipelining"):
        out = list(pipelined([0, 1], front, tail, gate_timeout=0.05))
    assert out == [(0, 0), (1, 10)]
    assert any("not released within" in r.getMessage() for r in caplog.records)


# ---------------------------------------------------------------------------------------------------------------
# pipelined_owned: jobs whose middle step is computed by ONE rank (job j -> rank j % world) and shared with the others
# ---------------------------------------------------------------------------------------------------------------
def _owned_worker(rank, world, port, jobs, front_s, solve_s, q):
    import os
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    side = dist.new_group(list(range(world)), backend="gloo")     # the tail thread's collectives (as in the pipeline)
    from pyannote_audio_amd.pipelining import pipelined_owned
    solved_here = []

    def front(item, release):
        time.sleep(front_s / 4)
        release()
        time.sleep(3 * front_s / 4)
        t = torch.tensor([item + 100 * rank], dtype=torch.int64)
        got = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(got, t)                                   # the main thread's collective, in job order
        return {"item": item, "seen": [int(g) for g in got]}

    def solve(state):                                             # the expensive step: owners only
        solved_here.append(state["item"])
        time.sleep(solve_s)
        return sum(state["seen"]) * 7

    def share(j, owner, state, solution):
        t = torch.tensor([solution if rank == owner else -1], dtype=torch.int64)
        dist.broadcast(t, src=owner, group=side)
        return int(t)

    def finish(state, solution):
        return state["item"], solution

    t0 = time.perf_counter()
    out = list(pipelined_owned(range(jobs), front, solve, share, finish, rank, world))
    dt = time.perf_counter() - t0
    q.put((rank, out, solved_here, dt))
    dist.barrier()
    dist.destroy_process_group()


def test_owned_solve_steps_are_shared_in_order_and_do_not_serialise():
    """3 ranks over gloo, 9 jobs, solve = 4 x front: every rank gets every job's solution (computed once, by rank
    j % 3), in order; the stream runs at about max(front, solve / world) per job -- far from the `solve` per job that a
    redundant solve step (or a tail thread that waits for job j - 1's result before starting its own solve) costs."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    world, jobs, front_s, solve_s = 3, 9, 0.10, 0.40
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_owned_worker, args=(r, world, port, jobs, front_s, solve_s, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = [(j, j, (3 * j + 100 * (0 + 1 + 2)) * 7) for j in range(jobs)]
    for rank, out, solved_here, dt in res:
        assert [(item, o[0], o[1]) for item, o in out] == want, rank
        assert solved_here == [j for j in range(jobs) if j % world == rank]
        # serialised it would take jobs * solve_s = 3.6 s; pipelined ~ jobs * max(front, solve / world) + one solve
        # = 1.6 s (the bound leaves a second for a loaded machine: three processes, gloo over loopback)
        assert dt < 0.75 * jobs * solve_s, (rank, dt)


def test_owned_pipeline_surfaces_errors_and_single_rank_is_plain():
    from pyannote_audio_amd.pipelining import pipelined_owned
    out = list(pipelined_owned(range(5), lambda it, rel: it, lambda st: st * 2, lambda j, o, st, sol: sol,
                               lambda st, sol: (st, sol), rank=0, world=1))
    assert out == [(i, (i, 2 * i)) for i in range(5)]

    def bad_solve(st):
        if st == 2:
            raise RuntimeError("solve failed")
        return st

    with pytest.raises(RuntimeError, match="solve failed"):
        list(pipelined_owned(range(4), lambda it, rel: it, bad_solve, lambda j, o, st, sol: sol,
                             lambda st, sol: sol, rank=0, world=1))


def test_owned_front_error_surfaces_while_a_tail_is_stuck_in_its_collective():
    """rank 1 of 2 (emulated: `share` blocks like a broadcast nobody answers): job 0's tail sits in `share` when the
    front of job 1 raises -- the exception reaches the caller at once instead of after the tail's collective gives up"""
    import threading
    from pyannote_audio_amd.pipelining import pipelined_owned
    stuck = threading.Event()

    def front(item, release):
        if item == 1:
            time.sleep(0.2)        # (job 0's tail is inside `share` by now)
            raise KeyError("bad file")
        return item

    def share(j, owner, state, solution):
        stuck.wait(timeout=30.0)   # the other rank never comes
        return solution

    t0 = time.perf_counter()
    with pytest.raises(KeyError, match="bad file"):
        list(pipelined_owned(range(3), front, lambda st: st, share, lambda st, sol: sol, rank=1, world=2))
    assert time.perf_counter() - t0 < 5.0
    stuck.set()                    # (let the worker thread go)


def _owned_failing_worker(rank, world, port, q):
    import os
    import pickle
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    side = dist.new_group(list(range(world)), backend="gloo")
    from pyannote_audio_amd.pipelining import pipelined_owned, SolveFailed

    def solve(state):
        if state == 3:                                            # job 3 belongs to rank 1
            raise ValueError("no clusters today")
        return state * 2

    def share(j, owner, state, solution):                         # a pickled object, as parallel.broadcast_object
        box = [solution if rank == owner else None]
        dist.broadcast_object_list(box, src=owner, group=side)
        return box[0]

    got, err = [], None
    try:
        for item, out in pipelined_owned(range(6), lambda it, rel: it, solve, share, lambda st, sol: sol, rank, world):
            got.append(out)
    except (ValueError, SolveFailed) as exc:
        err = (type(exc).__name__, str(exc))
    q.put((rank, got, err))
    dist.barrier()
    dist.destroy_process_group()


def test_owned_solve_failure_reaches_every_rank():
    """the owner of job 3 raises in its solve step: it re-raises its own exception, the other rank raises SolveFailed
    naming the owner and the job -- neither waits in the broadcast -- and jobs 0..2 were delivered on both"""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_owned_failing_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, got0, err0), (r1, got1, err1) = res
    assert got0 == got1 == [0, 2, 4]
    assert err1[0] == "ValueError" and "no clusters today" in err1[1]
    assert err0[0] == "SolveFailed" and "rank 1 failed to solve job 3" in err0[1] and "no clusters today" in err0[1]


def test_read_ahead_loads_one_item_ahead_in_order_and_raises_at_the_right_take():
    import threading
    from pyannote_audio_amd.pipelining import ReadAhead
    started, lock = [], threading.Lock()

    def load(item):
        with lock:
            started.append(item)
        if item == "bad":
            raise OSError("cannot read bad")
        time.sleep(0.02)
        return item.upper()

    ahead = ReadAhead(["a", "b", "bad", "d"], load)
    assert ahead.take(0) == "A"
    deadline = time.time() + 5.0
    while len(started) < 2 and time.time() < deadline:
        time.sleep(0.005)
    time.sleep(0.05)
    assert started == ["a", "b"]                  # item 1 was started by take(0); item 2 not yet
    assert ahead.take(1) == "B"
    with pytest.raises(OSError, match="cannot read bad"):
        ahead.take(2)                             # ... raised where ITS item is taken
    assert ahead.take(3) == "D"                   # (started by take(2))
    assert started == ["a", "b", "bad", "d"]
    ahead.close()
    # out-of-order / repeated takes load on demand
    again = ReadAhead(["x", "y"], load)
    assert again.take(1) == "Y" and again.take(0) == "X"
    again.close()


# ---------------------------------------------------------------------------------------------------------------
# run_ahead: the producer of results keeps going while the consumer works on the previous one
# ---------------------------------------------------------------------------------------------------------------
def test_run_ahead_keeps_order_and_overlaps_the_consumer():
    import threading
    from pyannote_audio_amd.pipelining import run_ahead
    made_in = []

    def make():
        for i in range(6):
            time.sleep(0.05)                      # "front end + tail of file i"
            made_in.append(threading.current_thread().name)
            yield i

    t0 = time.perf_counter()
    got = []
    for v in run_ahead(make, depth=1):
        time.sleep(0.05)                          # "serialize + write_rttm"
        got.append(v)
    dt = time.perf_counter() - t0
    assert got == list(range(6))
    assert set(made_in) == {"pa-run-ahead"}       # produced in the worker thread
    assert dt < 0.50, dt                          # 6 x (0.05 + 0.05) = 0.6 s in turns; ~0.35 s overlapped
    assert list(run_ahead(lambda: iter(()))) == []


def test_run_ahead_raises_at_the_right_position_and_stops_when_abandoned():
    import threading
    from pyannote_audio_amd.pipelining import run_ahead
    closed = threading.Event()
    produced = []

    def make():
        try:
            for i in range(100):
                if i == 3:
                    raise ValueError("file 3 is broken")
                produced.append(i)
                yield i
        finally:
            closed.set()

    got = []
    with pytest.raises(ValueError, match="file 3 is broken"):
        for v in run_ahead(make):
            got.append(v)
    assert got == [0, 1, 2] and closed.wait(2.0)

    closed.clear()
    produced.clear()

    def endless():
        try:
            i = 0
            while True:
                produced.append(i)
                yield i
                i += 1
        finally:
            closed.set()

    it = run_ahead(endless, depth=2)
    assert next(it) == 0 and next(it) == 1
    it.close()                                    # the consumer walks away
    assert closed.wait(2.0)                       # ... the inner iterator was closed, in the producer's thread
    assert len(produced) <= 6                     # and it was never more than depth + 1 results ahead
    # the context manager is entered in the producer thread
    where = []

    class Ctx:
        def __enter__(self):
            where.append(threading.current_thread().name)

        def __exit__(self, *a):
            where.append("exit")

    assert list(run_ahead(lambda: iter([1, 2]), context=Ctx)) == [1, 2]
    assert where == ["pa-run-ahead", "exit"]
