"""Minimal `pyannote.core` data types used on the hot path.

The reference returns `pyannote.core.Annotation` / `SlidingWindowFeature` objects.  When the real
package is importable we re-export it (true drop-in); otherwise these restatements of its published
semantics are used (pyannote-core 6.0.1 is not installed in the build / GPU containers).  Only the
subset the diarization path touches is provided: Segment, SlidingWindow (closest_frame, indexing,
crop, range_to_segment), SlidingWindowFeature (iteration, extent, crop, numpy protocol), Annotation
(track insertion, itertracks, labels, rename_labels, support, RTTM)."""
from __future__ import annotations

import itertools
import string
from typing import Iterator, Optional

import numpy as np

try:  # pragma: no cover - exercised only where pyannote.core exists
    from pyannote.core import Annotation, Segment, SlidingWindow, SlidingWindowFeature  # type: ignore

    HAVE_PYANNOTE_CORE = True
except Exception:  # noqa: BLE001
    HAVE_PYANNOTE_CORE = False

SEGMENT_PRECISION = 1e-6


def string_generator() -> Iterator[str]:
    """pyannote.core.utils.generators.string_generator: A..Z, AA..ZZ, ..."""
    r = 1
    while True:
        for c in itertools.product(string.ascii_uppercase, repeat=r):
            yield "".join(c)
        r += 1


if not HAVE_PYANNOTE_CORE:

    class Segment:
        __slots__ = ("start", "end")

        def __init__(self, start: float = 0.0, end: float = 0.0):
            self.start = start
            self.end = end

        def __bool__(self):
            return bool((self.end - self.start) > SEGMENT_PRECISION)

        @property
        def duration(self) -> float:
            return self.end - self.start if self else 0.0

        @property
        def middle(self) -> float:
            return 0.5 * (self.start + self.end)

        def __and__(self, other: "Segment") -> "Segment":
            return Segment(max(self.start, other.start), min(self.end, other.end))

        def __or__(self, other: "Segment") -> "Segment":
            if not self:
                return other
            if not other:
                return self
            return Segment(min(self.start, other.start), max(self.end, other.end))

        def intersects(self, other: "Segment") -> bool:
            return (self.start < other.start and other.start < self.end - SEGMENT_PRECISION) or \
                   (self.start > other.start and self.start < other.end - SEGMENT_PRECISION) or \
                   (self.start == other.start)

        def __iter__(self):
            yield self.start
            yield self.end

        def _key(self):
            return (self.start, self.end)

        def __eq__(self, other):
            return isinstance(other, Segment) and self._key() == other._key()

        def __lt__(self, other):
            return self._key() < other._key()

        def __le__(self, other):
            return self._key() <= other._key()

        def __hash__(self):
            return hash(self._key())

        def __repr__(self):
            return f"<Segment({self.start:g}, {self.end:g})>"

        def __str__(self):
            return f"[{self.start:.3f} --> {self.end:.3f}]"

    class SlidingWindow:
        def __init__(self, duration: float = 0.030, step: float = 0.010, start: float = 0.000,
                     end: Optional[float] = None):
            if duration <= 0:
                raise ValueError("'duration' must be a float > 0.")
            if step <= 0:
                raise ValueError("'step' must be a float > 0.")
            self.duration = duration
            self.step = step
            self.start = start
            self.end = np.inf if end is None else end

        def closest_frame(self, t: float) -> int:
            return int(np.rint((t - self.start - 0.5 * self.duration) / self.step))

        def __getitem__(self, i: int) -> Segment:
            start = self.start + i * self.step
            if start >= self.end:
                return None
            return Segment(start=start, end=start + self.duration)

        def range_to_segment(self, i0: int, n: int) -> Segment:
            start = self.start + (i0 - 0.5) * self.step + 0.5 * self.duration
            end = start + n * self.step
            if i0 == 0:
                start = self.start
            return Segment(start, end)

        def crop(self, focus: Segment, mode: str = "loose", fixed: Optional[float] = None,
                 return_ranges: bool = False):
            if mode == "loose":
                i = int(np.ceil((focus.start - self.duration - self.start) / self.step))
                j = int(np.floor((focus.end - self.start) / self.step))
                rng = (i, j + 1)
            elif mode == "strict":
                i = int(np.ceil((focus.start - self.start) / self.step))
                j = int(np.floor((focus.end - self.duration - self.start) / self.step))
                rng = (i, j + 1)
            elif mode == "center":
                i = self.closest_frame(focus.start)
                if fixed is None:
                    j = self.closest_frame(focus.end)
                    rng = (i, j + 1)
                else:
                    n = int(np.rint(fixed / self.step))
                    rng = (i, i + n)
            else:
                raise ValueError("'mode' must be one of {'loose', 'strict', 'center'}.")
            if return_ranges:
                return [list(rng)]
            return np.array(range(*rng), dtype=np.int64)

        def __repr__(self):
            return f"<SlidingWindow(start={self.start:g}, duration={self.duration:g}, step={self.step:g})>"

    class SlidingWindowFeature:
        def __init__(self, data: np.ndarray, sliding_window: SlidingWindow, labels=None):
            self.sliding_window = sliding_window
            self.data = data
            self.labels = labels

        def __len__(self):
            return self.data.shape[0]

        @property
        def extent(self) -> Segment:
            return self.sliding_window.range_to_segment(0, len(self))

        @property
        def dimension(self):
            return self.data.shape[1]

        def __iter__(self):
            for i in range(len(self)):
                yield self.sliding_window[i], self.data[i]

        def __getitem__(self, i):
            return self.data[i]

        def crop(self, focus: Segment, mode: str = "loose", fixed: Optional[float] = None,
                 return_data: bool = True):
            ranges = self.sliding_window.crop(focus, mode=mode, fixed=fixed, return_ranges=True)
            n = self.data.shape[0]
            clipped = [[max(s, 0), min(e, n)] for s, e in ranges if not (e < 0 or s >= n)]
            if clipped:
                data = np.vstack([self.data[s:e] for s, e in clipped])
            else:
                data = np.empty((0,) + self.data.shape[1:], dtype=self.data.dtype)
            if return_data:
                return data
            sw = SlidingWindow(start=self.sliding_window[clipped[0][0]].start,
                               duration=self.sliding_window.duration, step=self.sliding_window.step)
            return SlidingWindowFeature(data, sw, labels=self.labels)

        # numpy protocol (np.sum(swf, axis=...) returns a SlidingWindowFeature in pyannote.core;
        # the hot path only relies on np.argsort(-swf) -> ndarray and np.sum(..., keepdims=True))
        def __array__(self, dtype=None, copy=None):
            return self.data if dtype is None else self.data.astype(dtype)

        def __neg__(self):
            return SlidingWindowFeature(-self.data, self.sliding_window, labels=self.labels)

    class Annotation:
        """Segment -> {track: label} container (pyannote.core.Annotation subset).

        Native storage is COLUMNAR when the annotation comes out of the pipeline
        (`Annotation.from_columns`: start / end / track / label arrays straight from the run
        boundaries of the frame-level decisions); `Segment` objects and the per-segment dicts are only
        built when a caller edits or iterates.  `labels()`, `rename_labels()`, `to_rttm()` and `len()`
        work on the columns."""

        def __init__(self, uri: Optional[str] = None, modality: Optional[str] = None):
            self.uri = uri
            self.modality = modality
            self._dict: dict = {}     # Segment -> {track: label}
            self._cols = None         # (starts f64, ends f64, tracks list, labels list) not yet in _dict
            self._sorted: Optional[list] = None

        @classmethod
        def from_columns(cls, starts, ends, tracks, labels, uri=None, modality=None) -> "Annotation":
            """rows (start[i], end[i], tracks[i], labels[i]); empty segments (duration <= precision)
            are dropped exactly like `annotation[segment, track] = label` drops them."""
            starts = np.asarray(starts, dtype=np.float64)
            ends = np.asarray(ends, dtype=np.float64)
            keep = (ends - starts) > SEGMENT_PRECISION
            out = cls(uri=uri, modality=modality)
            if not keep.all():
                idx = np.nonzero(keep)[0]
                starts, ends = starts[idx], ends[idx]
                tracks = [tracks[i] for i in idx]
                labels = [labels[i] for i in idx]
            tracks, labels = list(tracks), list(labels)
            # `annotation[segment, track] = label` is last-write-wins per (segment, track): keep only
            # the last row of every duplicated key so that the columnar answers (labels(), len(),
            # rename_labels()) equal those of the incrementally built container.
            last = {}
            for i, key in enumerate(zip(starts.tolist(), ends.tolist(), tracks)):
                last[key] = i
            if len(last) < len(tracks):
                idx = np.fromiter(last.values(), dtype=np.int64, count=len(last))
                idx.sort()
                starts, ends = starts[idx], ends[idx]
                tracks = [tracks[i] for i in idx]
                labels = [labels[i] for i in idx]
            out._cols = (starts, ends, tracks, labels)
            return out

        @property
        def _tracks(self) -> dict:
            if self._cols is not None:
                starts, ends, tracks, labels = self._cols
                self._cols = None
                d = self._dict
                for a, b, t, l in zip(starts.tolist(), ends.tolist(), tracks, labels):
                    d.setdefault(Segment(a, b), {})[t] = l
                self._sorted = None
            return self._dict

        @_tracks.setter
        def _tracks(self, value: dict):
            self._cols = None
            self._dict = value
            self._sorted = None

        # -- construction
        def __setitem__(self, key, label):
            if isinstance(key, Segment):
                key = (key, "_")
            segment, track = key
            if not segment:
                return
            self._tracks.setdefault(segment, {})[track] = label
            self._sorted = None

        def __getitem__(self, key):
            if isinstance(key, Segment):
                key = (key, "_")
            return self._tracks[key[0]][key[1]]

        def __delitem__(self, key):
            if isinstance(key, Segment):
                del self._tracks[key]
            else:
                segment, track = key
                del self._tracks[segment][track]
                if not self._tracks[segment]:
                    del self._tracks[segment]
            self._sorted = None

        def _segments(self):
            if self._sorted is None:
                self._sorted = sorted(self._tracks)
            return self._sorted

        def __len__(self):
            if self._cols is not None and not self._dict:
                return len(set(zip(self._cols[0].tolist(), self._cols[1].tolist())))
            return len(self._tracks)

        def __bool__(self):
            if self._cols is not None and len(self._cols[0]):
                return True
            return len(self._tracks) > 0

        def itersegments(self):
            return iter(self._segments())

        def itertracks(self, yield_label: bool = False):
            for segment in self._segments():
                for track, label in sorted(self._tracks[segment].items(),
                                           key=lambda tl: (str(tl[0]), str(tl[1]))):
                    if yield_label:
                        yield segment, track, label
                    else:
                        yield segment, track

        def labels(self) -> list:
            if self._cols is not None and not self._dict:
                return sorted(set(self._cols[3]), key=str)
            return sorted({l for t in self._tracks.values() for l in t.values()}, key=str)

        def get_timeline(self):
            return list(self._segments())

        def label_duration(self, label) -> float:
            return sum(s.duration for s, _, l in self.itertracks(yield_label=True) if l == label)

        def discretize(self, support: Optional[Segment] = None, resolution=0.01, labels: Optional[list] = None,
                       duration: Optional[float] = None) -> "SlidingWindowFeature":
            """(num_frames, num_labels) uint8 {0, 1}: label k is on in frame i when a segment of k, clipped to
            `support`, covers the frame in the "center" sense of SlidingWindow.crop.  Restates
            pyannote.core's Annotation.discretize (frames start at the support's start; `duration` fixes
            the number of frames to round(duration / step); labels default to those inside the support)."""
            segments = self._segments()
            if support is None:
                if not segments:
                    raise ValueError("cannot discretize an empty annotation without `support`")
                support = Segment(min(s.start for s in segments), max(s.end for s in segments))
            inside = []
            for segment, _, label in self.itertracks(yield_label=True):
                clipped = segment & support
                if clipped:
                    inside.append((clipped, label))
            if labels is None:
                labels = sorted({label for _, label in inside}, key=str)
            if isinstance(resolution, SlidingWindow):
                frames = SlidingWindow(start=support.start, step=resolution.step, duration=resolution.duration)
            else:
                frames = SlidingWindow(start=support.start, step=resolution, duration=resolution)
            if duration is None:
                num_frames = frames.closest_frame(support.end) - frames.closest_frame(support.start)
            else:
                num_frames = int(round(duration / frames.step))
            column = {label: k for k, label in enumerate(labels)}
            data = np.zeros((max(num_frames, 0), len(labels)), dtype=np.uint8)
            for clipped, label in inside:
                if label not in column:
                    continue
                (first, stop), = frames.crop(clipped, mode="center", return_ranges=True)
                data[max(0, first):min(stop, num_frames), column[label]] = 1
            return SlidingWindowFeature(data, frames, labels=labels)

        def rename_labels(self, mapping: Optional[dict] = None, generator="string", copy: bool = True):
            if mapping is None:
                gen = string_generator() if generator == "string" else itertools.count()
                mapping = {label: next(gen) for label in self.labels()}
            if self._cols is not None and not self._dict:
                starts, ends, tracks, labels = self._cols
                new_labels = [mapping.get(l, l) for l in labels]
                if copy:
                    out = Annotation(uri=self.uri, modality=self.modality)
                    out._cols = (starts, ends, tracks, new_labels)
                    return out
                self._cols = (starts, ends, tracks, new_labels)
                return self
            out = Annotation(uri=self.uri, modality=self.modality) if copy else self
            items = [(s, t, l) for s, t, l in self.itertracks(yield_label=True)]
            if not copy:
                self._tracks = {}
            for s, t, l in items:
                out[s, t] = mapping.get(l, l)
            return out

        def support(self, collar: float = 0.0):
            """merge same-label segments closer than `collar` (Annotation.support)."""
            gen = string_generator()
            out = Annotation(uri=self.uri, modality=self.modality)
            for label in self.labels():
                segs = sorted(s for s, _, l in self.itertracks(yield_label=True) if l == label)
                if not segs:
                    continue
                cur = segs[0]
                for seg in segs[1:]:
                    # Timeline.support: merge when the segments intersect or the gap is < collar
                    gap = Segment(min(cur.end, seg.end), max(cur.start, seg.start))
                    if (cur & seg) or gap.duration < collar:
                        cur = Segment(cur.start, max(cur.end, seg.end))
                    else:
                        out[cur, next(gen)] = label
                        cur = seg
                out[cur, next(gen)] = label
            return out

        # -- RTTM (sample/sample.rttm: SPEAKER uri 1 start dur <NA> <NA> label <NA> <NA>)
        def flat_rows(self) -> list:
            """[(start, end, track, label)] in the order of `itertracks` (segments by (start, end), the tracks of a
            segment by (str(track), str(label))).  An annotation that still lives in its columns (the pipeline's
            output: ~20 000 turns per audio-hour) is listed WITHOUT building a Segment and a dictionary entry per
            turn -- what writing its RTTM / serialising it used to spend most of its time on."""
            if self._cols is not None and not self._dict:
                starts, ends, tracks, labels = self._cols
                order = np.lexsort((ends, starts))
                a, b = starts[order], ends[order]
                tied = np.flatnonzero((a[1:] == a[:-1]) & (b[1:] == b[:-1]))
                order = order.tolist()
                if len(tied):                       # several tracks of ONE segment: by (str(track), str(label))
                    lo = 0
                    while lo < len(tied):
                        hi = lo
                        while hi + 1 < len(tied) and tied[hi + 1] == tied[hi] + 1:
                            hi += 1
                        i0, i1 = int(tied[lo]), int(tied[hi]) + 2
                        order[i0:i1] = sorted(order[i0:i1], key=lambda i: (str(tracks[i]), str(labels[i])))
                        lo = hi + 1
                a, b = starts.tolist(), ends.tolist()
                return [(a[i], b[i], tracks[i], labels[i]) for i in order]
            return [(s.start, s.end, t, l) for s, t, l in self.itertracks(yield_label=True)]

        def to_rttm(self) -> str:
            uri = self.uri if self.uri else "<NA>"
            lines = []
            for start, end, _, label in self.flat_rows():
                duration = end - start if (end - start) > SEGMENT_PRECISION else 0.0     # (Segment.duration)
                lines.append(f"SPEAKER {uri} 1 {start:.3f} {duration:.3f} <NA> <NA> {label} <NA> <NA>\n")
            return "".join(lines)

        def write_rttm(self, file):
            file.write(self.to_rttm())

        def __eq__(self, other):
            return isinstance(other, Annotation) and \
                list(self.itertracks(yield_label=True)) == list(other.itertracks(yield_label=True))

        def __str__(self):
            return "\n".join(f"{s} {t} {l}" for s, t, l in self.itertracks(yield_label=True))
