"""Columnar `Annotation` (pipeline output) == incrementally built `Annotation` (pyannote.core semantics),
and the vectorised Binarize == the oracle's frame-by-frame state machine (utils/signal.py:254-318)."""
import numpy as np
from hypothesis import given, settings, strategies as st

import pyannote_audio_amd as pa
from pyannote_audio_amd.core import Annotation, Segment, SlidingWindow, SlidingWindowFeature
from pyannote_audio_amd.diarization import to_annotation


@settings(max_examples=60, deadline=None)
@given(st.lists(st.tuples(st.floats(0, 50, allow_nan=False), st.floats(0, 3, allow_nan=False),
                          st.sampled_from("ABC"), st.sampled_from([0, 1, 2, "x"])), max_size=40))
def test_columnar_equals_incremental(rows):
    starts = [r[0] for r in rows]
    ends = [r[0] + r[1] for r in rows]
    tracks = [r[2] for r in rows]
    labels = [r[3] for r in rows]
    col = Annotation.from_columns(starts, ends, tracks, labels, uri="u")
    inc = Annotation(uri="u")
    for a, b, t, l in zip(starts, ends, tracks, labels):
        inc[Segment(a, b), t] = l
    mapping = {0: "SPEAKER_00", 1: "SPEAKER_01"}
    assert col.labels() == inc.labels()
    assert len(col) == len(inc) and bool(col) == bool(inc)
    assert col.rename_labels(mapping=mapping).labels() == inc.rename_labels(mapping=mapping).labels()
    assert col.to_rttm() == inc.to_rttm()
    assert list(col.itertracks(yield_label=True)) == list(inc.itertracks(yield_label=True))
    assert col == inc
    # editing a columnar annotation materialises it first
    col2 = Annotation.from_columns(starts, ends, tracks, labels)
    col2[Segment(100.0, 101.0), "Z"] = "late"
    inc[Segment(100.0, 101.0), "Z"] = "late"
    assert list(col2.itertracks(yield_label=True)) == list(inc.itertracks(yield_label=True))


def test_columnar_duplicate_keys_last_write_wins():
    """the hypothesis counter-example of round 1: same (segment, track), different labels"""
    rows = [(0.0, 1.0, "A", 0), (0.0, 1.0, "A", 1)]
    col = Annotation.from_columns(*zip(*rows))
    inc = Annotation()
    for a, b, t, l in rows:
        inc[Segment(a, b), t] = l
    assert col.labels() == inc.labels() == [1]
    assert len(col) == len(inc) == 1
    assert col == inc


def test_binarize_with_pyannote_core_style_annotation(monkeypatch):
    """When the real pyannote.core is importable core.py re-exports ITS Annotation, which has no
    `from_columns`: Binarize must fall back to item assignment (utils/signal.py:283-305)."""
    import pyannote_audio_amd.diarization as dz

    class PlainAnnotation:
        def __init__(self, uri=None, modality=None):
            self.rows = {}

        def __setitem__(self, key, label):
            segment, track = key
            if segment:
                self.rows[(segment.start, segment.end, track)] = label

    d = np.zeros((50, 2), np.float32)
    d[3:10, 0] = 1
    d[20:50, 1] = 1
    frames = SlidingWindow(start=0.0, duration=0.0619375, step=0.016875)
    want = to_annotation(SlidingWindowFeature(d, frames))
    monkeypatch.setattr(dz, "Annotation", PlainAnnotation)
    got = dz.to_annotation(SlidingWindowFeature(d, frames))
    assert sorted(got.rows.items()) == sorted(
        ((s.start, s.end, t), l) for s, t, l in want.itertracks(yield_label=True))


@settings(max_examples=40, deadline=None)
@given(st.integers(1, 400), st.integers(1, 4), st.integers(0, 2 ** 31 - 1))
def test_binarize_matches_oracle(num_frames, K, seed):
    from oracle import pipeline as op
    rng = np.random.default_rng(seed)
    d = (rng.uniform(size=(num_frames, K)) < 0.5).astype(np.float32)
    # long runs as well as flicker
    for k in range(K):
        if rng.uniform() < 0.5:
            a = rng.integers(0, num_frames)
            d[a:a + rng.integers(1, num_frames + 1), k] = rng.integers(0, 2)
    frames = SlidingWindow(start=0.0, duration=0.0619375, step=0.016875)
    ann = to_annotation(SlidingWindowFeature(d, frames))
    got = [(s.start, s.end, t, l) for s, t, l in ann.itertracks(yield_label=True)]
    want = op.binarize(d, op.SW(0.0, 0.0619375, 0.016875))
    assert sorted(got, key=lambda r: (r[0], r[1], str(r[2]))) == sorted(want, key=lambda r: (r[0], r[1], str(r[2])))


@settings(max_examples=50, deadline=None)
@given(st.integers(1, 4000), st.sampled_from([(10.0, 1.0), (5.0, 0.5), (10.0, 2.5), (3.0, 3.0)]))
def test_frame_geometry_matches_oracle(num_chunks, geom):
    from oracle import pipeline as op
    from pyannote_audio_amd.frames import frame_geometry
    dur, step = geom
    chunks = SlidingWindow(start=0.0, duration=dur, step=step)
    frames = SlidingWindow(start=0.0, duration=0.0619375, step=0.016875)
    starts, T, _ = frame_geometry(chunks, frames, num_chunks)
    ochunks, oframes = op.SW(0.0, dur, step), op.SW(0.0, 0.0619375, 0.016875)
    want_T = oframes.closest_frame(ochunks.start + ochunks.duration + (num_chunks - 1) * ochunks.step
                                   + 0.5 * oframes.duration) + 1
    assert T == want_T
    idx = sorted({0, num_chunks - 1, num_chunks // 2, min(num_chunks - 1, 37)})
    for c in idx:
        s0, _ = ochunks.segment(c)
        assert starts[c] == oframes.closest_frame(s0 + 0.5 * oframes.duration)


def test_flat_rows_rttm_and_serialize_from_columns_equal_the_dictionary_path():
    """`flat_rows` (what `to_rttm` / `DiarizeOutput.serialize` read) lists a columnar annotation without building a
    Segment per turn: same rows, same order as `itertracks` -- also with empty segments, several tracks per segment,
    coarse (tied) times and shuffled input"""
    for seed in range(6):
        rng = np.random.default_rng(seed)
        n = 2000 if seed == 0 else 200
        digits = 6 if seed < 2 else 1
        starts = np.round(np.cumsum(rng.uniform(0.1, 1.0, n)), digits)
        ends = starts + np.round(rng.uniform(0.0, 2.0, n), digits)
        ends[::17] = starts[::17]                                  # empty segments: duration 0.000 in RTTM
        labels = [f"SPEAKER_{int(k):02d}" for k in rng.integers(0, 4, n)]
        tracks = list(rng.integers(0, 3, n))
        k = 30                                                      # more tracks on the first segments
        starts = np.concatenate([starts, starts[:k], starts[:k]])
        ends = np.concatenate([ends, ends[:k], ends[:k]])
        tracks = tracks + [7] * k + ["a"] * k
        labels = labels + ["X"] * k + ["B"] * k
        perm = rng.permutation(len(starts))
        starts, ends = starts[perm], ends[perm]
        tracks, labels = [tracks[i] for i in perm], [labels[i] for i in perm]
        fast = Annotation.from_columns(starts, ends, tracks, labels, uri="u")
        slow = Annotation.from_columns(starts, ends, tracks, labels, uri="u")
        assert slow._tracks is not None and slow._cols is None      # (materialised: the dictionary path)
        assert fast._cols is not None
        want = [(s.start, s.end, t, l) for s, t, l in slow.itertracks(yield_label=True)]
        assert fast.flat_rows() == want == slow.flat_rows()
        assert fast._cols is not None, "listing the rows must not materialise the dictionary"
        assert fast.to_rttm() == slow.to_rttm()
        one = pa.DiarizeOutput(speaker_diarization=fast, exclusive_speaker_diarization=fast, speaker_embeddings=None)
        two = pa.DiarizeOutput(speaker_diarization=slow, exclusive_speaker_diarization=slow, speaker_embeddings=None)
        assert one.serialize() == two.serialize()
    assert Annotation(uri="e").flat_rows() == [] and Annotation(uri="e").to_rttm() == ""
