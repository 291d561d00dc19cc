"""Two-stage software pipeline of the batch forms of `SpeakerDiarization` (`apply_batch`, `apply_joint_batches`).

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
from typing import Any, Callable, Iterable, Iterator, Tuple


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
