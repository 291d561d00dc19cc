"""OracleClustering (pipelines/clustering.py:672-756) and oracle_segmentation (pipelines/utils/oracle.py:31-115):
the reference annotation discretised on the model's chunk / frame grid, and the per-chunk permutation that maps
the model's local speakers onto the reference speakers.  Annotation.discretize of the local container restates
pyannote.core (not installed: unpinned); it is checked here against its definition frame by frame."""
import numpy as np
import pytest

import pyannote_audio_amd as pa
from pyannote_audio_amd.annotation_frames import oracle_segmentation
from pyannote_audio_amd.core import Segment, SlidingWindow, SlidingWindowFeature

FRAMES = SlidingWindow(start=0.0, duration=0.0619375, step=0.016875)     # PyanNet's receptive field
CHUNKS = SlidingWindow(start=0.0, duration=5.0, step=1.0)


def _reference():
    ann = pa.Annotation(uri="toy")
    ann[Segment(0.5, 4.0), "a"] = "alice"
    ann[Segment(3.0, 7.5), "b"] = "bob"
    ann[Segment(7.0, 9.0), "c"] = "alice"
    ann[Segment(8.5, 12.0), "d"] = "carol"
    return ann


def test_discretize_matches_its_definition():
    ann = _reference()
    support = Segment(2.0, 7.0)
    swf = ann.discretize(support, resolution=FRAMES, labels=["alice", "bob", "carol"], duration=5.0)
    num_frames = int(round(5.0 / FRAMES.step))
    assert swf.data.shape == (num_frames, 3) and swf.data.dtype == np.uint8
    assert swf.sliding_window.start == 2.0 and swf.sliding_window.step == FRAMES.step
    grid = SlidingWindow(start=2.0, duration=FRAMES.duration, step=FRAMES.step)
    want = np.zeros((num_frames, 3), dtype=np.uint8)
    for (a, b), k in (((2.0, 4.0), 0), ((3.0, 7.0), 1)):                 # clipped to the support; carol is outside
        i, j = grid.closest_frame(a), grid.closest_frame(b)
        want[max(i, 0):min(j + 1, num_frames), k] = 1
    assert np.array_equal(swf.data, want)
    assert swf.data[:, 2].sum() == 0 and 0 < swf.data[:, 0].sum() < swf.data[:, 1].sum()
    free = ann.discretize(support, resolution=0.01)                      # labels and length from the support
    assert free.data.shape[1] == 2 and abs(free.data.shape[0] - 500) <= 1


def test_oracle_segmentation_grid_and_speakers():
    file = {"uri": "toy", "annotation": _reference(), "duration": 12.0}
    seg = oracle_segmentation(file, CHUNKS, FRAMES)
    assert seg.data.shape == (8, int(round(5.0 / FRAMES.step)), 3) and seg.data.dtype == np.float32   # chunks 0..7
    assert seg.sliding_window.start == 0.0 and seg.sliding_window.duration == 5.0 and seg.sliding_window.step == 1.0
    assert seg.data[0, :, 2].sum() == 0 and seg.data[7, :, 2].sum() > 0       # carol only speaks late
    padded = oracle_segmentation(file, CHUNKS, FRAMES, num_speakers=5)
    assert padded.data.shape[2] == 5 and padded.data[..., 3:].sum() == 0
    fewer = oracle_segmentation(file, CHUNKS, FRAMES, num_speakers=1)
    assert fewer.data.shape[2] == 1
    assert np.array_equal(fewer.data[0, :, 0], seg.data[0, :, 0])             # alice talks most in chunk 0
    short = oracle_segmentation({"uri": "s", "annotation": _reference(), "duration": 3.0}, CHUNKS, FRAMES)
    assert short.data.shape[0] == 0


def test_oracle_clustering_recovers_the_permutation():
    file = {"uri": "toy", "annotation": _reference(), "duration": 12.0}
    oracle = oracle_segmentation(file, CHUNKS, FRAMES)
    num_chunks, num_frames, _ = oracle.data.shape
    rng = np.random.default_rng(0)
    perms = [rng.permutation(3) for _ in range(num_chunks)]
    # the model sees the same activity with its local speakers shuffled, one frame fewer, plus a padded last chunk
    model = np.stack([oracle.data[c][:, perms[c]] for c in range(num_chunks)] + [np.zeros((num_frames, 3), np.float32)])
    model = model[:, :num_frames - 1]
    segmentations = SlidingWindowFeature(model, CHUNKS)
    clustering = pa.OracleClustering()
    hard, soft, centroids = clustering(segmentations=segmentations, file=file, frames=FRAMES)
    assert centroids is None and hard.shape == (num_chunks + 1, 3) and soft.shape == (num_chunks + 1, 3, 3)
    for c in range(num_chunks):
        for local in range(3):
            if model[c, :, local].sum() > 0:                     # silent local speakers may go anywhere
                assert hard[c, local] == perms[c][local]
                assert soft[c, local, hard[c, local]] == 1.0
    assert np.all(hard[-1] == -2)                                # no reference chunk for the padded one
    assert "oracle_segmentations" in file
    # with embeddings: centroids = mean training embedding of every reference speaker
    emb = rng.standard_normal((num_chunks + 1, 3, 8)).astype(np.float32)
    hard2, _, centroids = clustering(embeddings=emb, segmentations=segmentations, file=file, frames=FRAMES)
    assert np.array_equal(hard, hard2) and centroids.shape == (3, 8)
    train, ci, si = clustering.filter_embeddings(emb, segmentations=SlidingWindowFeature(model, CHUNKS))
    for k in range(3):
        sel = hard[ci, si] == k
        assert np.allclose(centroids[k], train[sel].mean(axis=0))
    assert pa.OracleClustering.expects_num_clusters is True
