"""Reference annotation -> frame-level chunks: what an (imaginary) perfect segmentation model would output
for a file, on the chunk grid and frame resolution of the real one (pipelines/utils/oracle.py:31-115).
Needs `file["annotation"]`, i.e. evaluation / tuning set-ups: it is what `OracleClustering` clusters with."""
from __future__ import annotations

from typing import Optional, Union

import numpy as np

from .audio import Audio, AudioFile
from .core import SlidingWindow, SlidingWindowFeature


def oracle_segmentation(file: AudioFile, window: SlidingWindow, frames: Union[SlidingWindow, float],
                        num_speakers: Optional[int] = None) -> SlidingWindowFeature:
    """(num_chunks, num_frames, num_speakers) float32 {0, 1}.  Chunk c = [c step, c step + duration] of the
    reference annotation, discretised at the `frames` resolution with one column per reference label
    (sorted); only chunks that lie completely inside the file exist (`window(Segment(0, duration))` of
    pyannote.core yields nothing else -- so there can be one chunk fewer than `Inference.slide` produces,
    whose last chunk is zero padded).  `num_speakers` above the number of reference speakers pads with
    never-active columns; below it, every chunk keeps its most talkative speakers (what oracle.py:104-110
    documents; its `np.argsort(-np.sum(uint8 data))` negates an UNSIGNED sum, which puts silent speakers first --
    that quirk is not reproduced; `OracleClustering` never takes this branch)."""
    duration = file["duration"] if "duration" in file else Audio(mono="downmix").get_duration(file)
    reference = file["annotation"]
    if not isinstance(frames, SlidingWindow):
        frames = SlidingWindow(start=0.0, step=frames, duration=frames)
    labels = list(reference.labels())
    actual = len(labels)
    wanted = actual if num_speakers is None else num_speakers
    labels += [f"FakeSpeakerForOracleSegmentationInference{i:d}" for i in range(max(0, wanted - actual))]
    grid = SlidingWindow(start=0.0, duration=window.duration, step=window.step)
    num_frames = int(round(grid.duration / frames.step))
    chunks = []
    index = 0
    while grid[index].end <= duration:
        data = np.asarray(reference.discretize(grid[index], resolution=frames, labels=labels,
                                               duration=grid.duration).data)
        if wanted < actual:
            data = data[:, np.argsort(-np.sum(data, axis=0, dtype=np.int64), kind="stable")[:wanted]]
        chunks.append(data)
        index += 1
    stacked = np.stack(chunks) if chunks else np.zeros((0, num_frames, wanted))
    return SlidingWindowFeature(np.float32(stacked), grid)
