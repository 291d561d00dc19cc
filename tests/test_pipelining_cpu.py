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
        time.sleep(0.05)              # "embeddings": the previous tail runs beside this
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
