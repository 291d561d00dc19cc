"""Counting / discretisation helpers of the diarization pipeline, vectorised
(mirrors pipelines/utils/diarization.py:34-268 and utils/signal.py:207-318).

The reference walks every frame in Python (hot loops #4 and #5 of SURVEY.md section 3.2:
diarization.py:264-266 and signal.py:276-305, ~213 k iterations per audio-hour each); here they are
array operations with the same tie-breaking and the same floating-point timestamps."""
from __future__ import annotations

from typing import Mapping, Optional, Tuple

import numpy as np

from .core import Annotation, Segment, SlidingWindow, SlidingWindowFeature, string_generator
from .inference import Inference


def set_num_speakers(num_speakers: Optional[int] = None, min_speakers: Optional[int] = None,
                     max_speakers: Optional[int] = None):
    """diarization.py:34-69"""
    min_speakers = num_speakers or min_speakers or 1
    max_speakers = num_speakers or max_speakers or np.inf
    if min_speakers > max_speakers:
        raise ValueError(f"min_speakers must be smaller than (or equal to) max_speakers "
                         f"(here: min_speakers={min_speakers:g} and max_speakers={max_speakers:g}).")
    if min_speakers == max_speakers:
        num_speakers = min_speakers
    return num_speakers, min_speakers, max_speakers


def speaker_count(binarized_segmentations: SlidingWindowFeature, frames: SlidingWindow,
                  warm_up: Tuple[float, float] = (0.1, 0.1)) -> SlidingWindowFeature:
    """diarization.py:150-185: overlap-add average of the per-chunk number of active speakers."""
    trimmed = Inference.trim(binarized_segmentations, warm_up=warm_up)
    summed = SlidingWindowFeature(np.sum(trimmed.data, axis=-1, keepdims=True), trimmed.sliding_window)
    count = Inference.aggregate(summed, frames, hamming=False, missing=0.0, skip_average=False)
    count.data = np.rint(count.data).astype(np.uint8)
    return count


def to_diarization(segmentations: SlidingWindowFeature, count: SlidingWindowFeature
                   ) -> SlidingWindowFeature:
    """diarization.py:221-268: overlap-add SUM of clustered activations, then keep the
    count[t] most active speakers per frame (ties: lowest index, as np.argsort on K < 16)."""
    activations = Inference.aggregate(segmentations, count.sliding_window, hamming=False, missing=0.0,
                                      skip_average=True)
    _, num_speakers = activations.data.shape
    max_speakers_per_frame = np.max(count.data)
    if num_speakers < max_speakers_per_frame:
        activations.data = np.pad(activations.data,
                                  ((0, 0), (0, max_speakers_per_frame - num_speakers)))
    extent = activations.extent & count.extent
    activations = activations.crop(extent, return_data=False)
    count = count.crop(extent, return_data=False)
    act = activations.data
    sorted_speakers = np.argsort(-act, axis=-1)
    K = act.shape[1]
    c = np.minimum(count.data.reshape(-1).astype(np.int64), K)
    keep = np.arange(K)[None, :] < c[:, None]          # the first count[t] entries of each sorted row
    binary = np.zeros_like(act)
    rows = np.nonzero(keep)[0]
    binary[rows, sorted_speakers[keep]] = 1.0
    return SlidingWindowFeature(binary, activations.sliding_window)


class Binarize:
    """utils/signal.py:207-318 (hysteresis thresholding -> Annotation), vectorised per class."""

    def __init__(self, onset: float = 0.5, offset: Optional[float] = None, min_duration_on: float = 0.0,
                 min_duration_off: float = 0.0, pad_onset: float = 0.0, pad_offset: float = 0.0):
        self.onset = onset
        self.offset = offset or onset
        self.pad_onset = pad_onset
        self.pad_offset = pad_offset
        self.min_duration_on = min_duration_on
        self.min_duration_off = min_duration_off

    def __call__(self, scores: SlidingWindowFeature) -> Annotation:
        num_frames, num_classes = scores.data.shape
        frames = scores.sliding_window
        i = np.arange(num_frames, dtype=np.float64)
        start = frames.start + i * frames.step
        timestamps = 0.5 * (start + (start + frames.duration))   # Segment.middle of frames[i]
        track_generator = string_generator()
        labels = getattr(scores, "labels", None)
        col_start, col_end, col_track, col_label = [], [], [], []
        for k in range(num_classes):
            y = scores.data[:, k]
            label = k if labels is None else labels[k]
            track = next(track_generator)
            if num_frames == 0:
                continue
            on, off = y > self.onset, y < self.offset
            # state[t] = 1 if last decisive event up to t was "on"; frame 0 initialises the state
            on0 = on.copy()
            off0 = off.copy()
            off0[0] = not on0[0]          # is_active = k_scores[0] > onset
            ev = np.where(on0, 1, np.where(off0, 0, -1))
            idx = np.where(ev >= 0, np.arange(num_frames), 0)
            np.maximum.accumulate(idx, out=idx)
            state = ev[idx].astype(bool)
            d = np.diff(state.astype(np.int8))
            ups = np.nonzero(d == 1)[0] + 1
            downs = np.nonzero(d == -1)[0] + 1
            if state[0]:
                ups = np.concatenate([[0], ups])
            if state[-1]:
                downs = np.concatenate([downs, [num_frames - 1]])
            m = min(len(ups), len(downs))
            col_start.append(timestamps[ups[:m]] - self.pad_onset)
            col_end.append(timestamps[downs[:m]] + self.pad_offset)
            col_track += [track] * m
            col_label += [label] * m
        if col_start and hasattr(Annotation, "from_columns"):
            active = Annotation.from_columns(np.concatenate(col_start), np.concatenate(col_end),
                                             col_track, col_label)
        else:
            # the real pyannote.core.Annotation (re-exported by core.py when it is importable) has no
            # columnar constructor: insert the rows the way utils/signal.py:283-305 does
            active = Annotation()
            if col_start:
                for a, b, t, l in zip(np.concatenate(col_start).tolist(),
                                      np.concatenate(col_end).tolist(), col_track, col_label):
                    active[Segment(a, b), t] = l
        if self.pad_offset > 0.0 or self.pad_onset > 0.0 or self.min_duration_off > 0.0:
            active = active.support(collar=self.min_duration_off)
        if self.min_duration_on > 0:
            for segment, track in list(active.itertracks()):
                if segment.duration < self.min_duration_on:
                    del active[segment, track]
        return active


def to_annotation(discrete_diarization: SlidingWindowFeature, min_duration_on: float = 0.0,
                  min_duration_off: float = 0.0) -> Annotation:
    """diarization.py:188-218"""
    return Binarize(onset=0.5, offset=0.5, min_duration_on=min_duration_on,
                    min_duration_off=min_duration_off)(discrete_diarization)


def cooccurrence(a: Annotation, b: Annotation, annotated=None) -> Tuple[list, list, np.ndarray]:
    """(labels of a, labels of b, seconds during which label i of `a` and label j of `b` are both on) --
    the co-occurrence matrix `a * b` of pyannote.core, optionally restricted to the `annotated` regions."""
    la, lb = a.labels(), b.labels()
    ia, ib = {l: i for i, l in enumerate(la)}, {l: j for j, l in enumerate(lb)}
    regions = None if annotated is None else [(r.start, r.end) for r in annotated]
    tb = [(s.start, s.end, ib[l]) for s, _, l in b.itertracks(yield_label=True)]
    out = np.zeros((len(la), len(lb)))
    for s, _, l in a.itertracks(yield_label=True):
        for start, end, j in tb:
            lo, hi = max(s.start, start), min(s.end, end)
            if hi <= lo:
                continue
            if regions is None:
                out[ia[l], j] += hi - lo
            else:
                out[ia[l], j] += sum(max(0.0, min(hi, r1) - max(lo, r0)) for r0, r1 in regions)
    return la, lb, out


def optimal_mapping(reference, hypothesis: Annotation, return_mapping: bool = False):
    """Hypothesis labels renamed to the reference labels they overlap most with, one-to-one (Hungarian on the
    co-occurrence durations; pairs that never overlap stay unmapped) -- pipelines/utils/diarization.py:104-148,
    which delegates to pyannote.metrics' DiarizationErrorRate().optimal_mapping (not installed: restated,
    unpinned).  `reference` may be the annotation or a file mapping with "annotation" [and "annotated"]."""
    from scipy.optimize import linear_sum_assignment
    annotated = None
    if isinstance(reference, Mapping):
        annotated = reference["annotated"] if "annotated" in reference else None
        reference = reference["annotation"]
    hyp_labels, ref_labels, together = cooccurrence(hypothesis, reference, annotated)
    mapping = {}
    if together.size:
        for i, j in zip(*linear_sum_assignment(-together)):
            if together[i, j] > 0:
                mapping[hyp_labels[i]] = ref_labels[j]
    mapped = hypothesis.rename_labels(mapping=mapping)
    return (mapped, mapping) if return_mapping else mapped
