"""Software pipelines of the batch forms of `SpeakerDiarization` (`apply_batch`, `apply_joint_batches`).

    front(item i+1)   main thread / stream 0:   |-- segmentation --|------ embeddings ------|
    tail(item i)      worker thread / stream 1:                     |-- clustering, back end --|

`front(item, release)` runs in the calling thread, one item after the other.  `tail(state, alone)` of item i runs in
ONE worker thread and starts only when the front of item i+1 has called `release()` (the batch forms call it when the
segmentation stage has left the device: its one-round grids must not share CUs with the dendrogram merge) -- or has
returned, or does not exist; `alone` is True for the last item (nothing runs beside its tail).  Results come back in
input order.  Pure host logic, no torch: tests/test_pipelining_cpu.py."""
from __future__ import annotations

import logging
import threading
from concurrent.futures import ThreadPoolExecutor
from typing import Any, Callable, Iterable, Iterator, Sequence, Tuple


def pipelined(items: Iterable[Any], front: Callable[[Any, Callable[[], None]], Any],
              tail: Callable[[Any, bool], Any], gate_timeout: float = 5.0) -> Iterator[Tuple[Any, Any]]:
    """Yields (item, tail(front(item))) in order.  `gate_timeout`: longest wait (s) of a tail for its release -- a
    safety net only: every gate is opened in a `finally`, so neither a failing front nor an abandoned generator
    strands the worker."""
    no_next_item = threading.Event()

    def gated_tail(state, gate: threading.Event):
        if not gate.wait(timeout=gate_timeout):
            # never expected (the gate is opened in a `finally`): say so instead of silently running the tail beside
            # the next item's first stage, where it costs that stage up to a quarter of its time
            logging.getLogger(__name__).warning(
                "pipelined: the tail of an item was not released within %.1f s and runs ungated", gate_timeout)
        return tail(state, no_next_item.is_set())

    with ThreadPoolExecutor(max_workers=1) as pool:
        in_flight = gate = None
        try:
            for item in items:
                state = front(item, gate.set if gate is not None else _nothing)
                if gate is not None:
                    gate.set()                      # (a front that never released)
                if in_flight is not None:
                    yield in_flight[0], in_flight[1].result()
                gate = threading.Event()
                in_flight = (item, pool.submit(gated_tail, state, gate))
            if in_flight is not None:
                no_next_item.set()
                gate.set()
                yield in_flight[0], in_flight[1].result()
        finally:
            if gate is not None:
                gate.set()


def _nothing() -> None:
    pass


class SolveFailed(RuntimeError):
    """what the owner of a job shares instead of a solution when its solve step raised: every rank then raises (the
    owner its own exception, the others this one) instead of waiting for a result that will not come"""

    def __init__(self, owner: int, job: int, what: str):
        super().__init__(f"rank {owner} failed to solve job {job}: {what}")
        self.owner, self.job, self.what = owner, job, what

    def __reduce__(self):
        return SolveFailed, (self.owner, self.job, self.what)


def pipelined_owned(items: Iterable[Any], front: Callable[[Any, Callable[[], None]], Any],
                    solve: Callable[[Any], Any], share: Callable[[int, int, Any, Any], Any],
                    finish: Callable[[Any, Any], Any], rank: int, world: int, depth: int = 0,
                    gate_timeout: float = 5.0) -> Iterator[Tuple[Any, Any]]:
    """Jobs whose expensive middle step is needed by every rank but has to be COMPUTED only once (the joint
    clustering of BASELINE.json configs[4]: every rank needs the cluster labels, one rank can make them).

        main thread      front(job j, release)          in job order (its collectives stay in this thread)
        solver thread    solve(state j)                 only on the OWNER of job j = rank j % world
        tail thread      share(j, owner, state, solution or None) -> solution on every rank
                         finish(state, solution)        in job order (the collectives of `share` stay in this thread,
                                                        in the same order on every rank)

    With the solve step done redundantly on every rank a stream of jobs runs at max(front, solve) per job; with owners
    it runs at max(front, solve / world): rank r solves job r while the other ranks' fronts and solves go on, and its
    tail thread picks the results up in order.  Up to `depth` jobs (default world + 1) are in flight before the main
    thread waits for the oldest; results come back in input order.  `solve` of job j starts when the front of job j + 1
    has called `release()` -- or has returned, or does not exist (as `pipelined`: the solve step must not run beside
    the first stage of the next front).  A solve step that raises is SHARED as a `SolveFailed` marker: the owner re-raises
    its exception, every other rank raises the marker -- nobody is left waiting in `share`.  Pure host logic:
    tests/test_pipelining_cpu.py."""
    depth = depth or world + 1
    log = logging.getLogger(__name__)

    def gated_solve(state, gate: threading.Event):
        if not gate.wait(timeout=gate_timeout):
            log.warning("pipelined_owned: a solve step was not released within %.1f s and runs ungated", gate_timeout)
        return solve(state)

    def tail(j: int, state, solved):
        owner = j % world
        failure = None
        try:
            solution = solved.result() if solved is not None else None
        except BaseException as exc:    # the owner's solve step failed: the other ranks are waiting in `share`
            failure, solution = exc, SolveFailed(owner, j, f"{type(exc).__name__}: {exc}")
        shared = share(j, owner, state, solution)
        if failure is not None:
            raise failure
        if isinstance(shared, SolveFailed):
            raise shared
        return finish(state, shared)

    # (not `with` blocks: leaving one WAITS for the queued tails -- which sit in `share` collectives that the other ranks
    #  only answer if this rank keeps going -- so a local error in `front` or in the consumer would turn into a hang
    #  until the process group's timeout.  On the error path the executors are dropped without waiting and what is
    #  queued is cancelled: the local exception surfaces at once.  The job is lost on every rank then -- the caller has
    #  to abort the other ranks (they time out in `share` otherwise); a tail already inside a collective cannot be
    #  interrupted and keeps its thread until that collective fails.)
    solver, tails = ThreadPoolExecutor(max_workers=1), ThreadPoolExecutor(max_workers=1)
    in_flight = []          # (item, tail future), oldest first
    gate = None             # gate of the newest job's solve step
    clean = False
    try:
        for j, item in enumerate(items):
            state = front(item, gate.set if gate is not None else _nothing)
            if gate is not None:
                gate.set()
            gate = threading.Event()
            solved = solver.submit(gated_solve, state, gate) if j % world == rank else None
            in_flight.append((item, tails.submit(tail, j, state, solved)))
            while len(in_flight) > depth:
                done_item, fut = in_flight.pop(0)
                yield done_item, fut.result()
        if gate is not None:
            gate.set()
        while in_flight:
            done_item, fut = in_flight.pop(0)
            yield done_item, fut.result()
        clean = True
    finally:
        if gate is not None:
            gate.set()
        solver.shutdown(wait=clean, cancel_futures=not clean)
        tails.shutdown(wait=clean, cancel_futures=not clean)


def run_ahead(make_iter: Callable[[], Iterator[Any]], depth: int = 1, context: Callable[[], Any] = None) -> Iterator[Any]:
    """`make_iter()` iterated in a PRODUCER thread, up to `depth` finished results ahead of the consumer: what the
    consumer does with result i (the reference's benchmark loop serialises it and writes its RTTM,
    src/pyannote/audio/__main__.py:700-720) no longer suspends the generator that would be starting the front end of
    item i + 2 -- the GPU keeps working while the host writes.  Results in order; an exception of the producer is raised
    by the consumer at the position where it happened; a consumer that stops early (break / close / exception) stops
    the producer at its next result and closes the inner iterator there, in the producer's thread.  `context()`: a
    context manager entered in the producer thread (the CUDA device of the pipeline: it is a per-thread setting)."""
    import contextlib
    import queue
    results: "queue.Queue" = queue.Queue(maxsize=max(1, depth))
    stop = threading.Event()

    def offer(entry) -> bool:
        while not stop.is_set():
            try:
                results.put(entry, timeout=0.05)
                return True
            except queue.Full:
                continue
        return False

    def produce():
        it = None
        try:
            with (context() if context is not None else contextlib.nullcontext()):
                it = make_iter()
                for value in it:
                    if not offer(("value", value)):
                        break
        except BaseException as exc:      # handed to the consumer, raised there
            offer(("error", exc))
        finally:
            try:
                if it is not None and hasattr(it, "close"):
                    it.close()
            finally:
                offer(("done", None))

    worker = threading.Thread(target=produce, name="pa-run-ahead", daemon=True)
    worker.start()
    try:
        while True:
            kind, payload = results.get()
            if kind == "value":
                yield payload
            elif kind == "error":
                raise payload
            else:
                return
    finally:
        stop.set()
        while worker.is_alive():          # unblock a producer waiting for room, then let it close the inner iterator
            try:
                results.get_nowait()
            except queue.Empty:
                pass
            worker.join(timeout=0.05)


class ReadAhead:
    """`load(items[i + 1])` runs in ONE worker thread while the caller works on item i: `take(i)` hands out
    `load(items[i])` (started by `take(i - 1)`, or now) and starts the next one.  At most one result is held ahead of
    the caller; what `load` raises is raised by the `take` of its item.  (apply_batch: the next file is read from
    disk while the GPU runs the current one.)"""

    def __init__(self, items: Sequence[Any], load: Callable[[Any], Any]):
        self._items, self._load = items, load
        self._pool = ThreadPoolExecutor(max_workers=1)
        self._pending: dict = {}

    def take(self, i: int):
        if i not in self._pending:
            self._pending[i] = self._pool.submit(self._load, self._items[i])
        fut = self._pending.pop(i)
        if i + 1 < len(self._items) and i + 1 not in self._pending:
            self._pending[i + 1] = self._pool.submit(self._load, self._items[i + 1])
        return fut.result()

    def close(self) -> None:
        self._pending.clear()
        self._pool.shutdown(wait=False, cancel_futures=True)
