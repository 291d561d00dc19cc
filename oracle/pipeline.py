"""Oracle (CPU) restatement of the SpeakerDiarization pipeline, loop for loop.

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.  Follows, relative to
/root/reference/src/pyannote/audio:
  core/inference.py:217-373 (slide), :498-620 (aggregate), :622-667 (trim)
  pipelines/utils/diarization.py:150-185 (speaker_count), :221-268 (to_diarization)
  pipelines/speaker_diarization.py:332-478 (get_embeddings), :480-528 (reconstruct), :530-784 (apply)
  pipelines/clustering.py:77-125, :142-212, :214-289, :330-480
  utils/signal.py:254-318 (Binarize.__call__)
pyannote.core semantics (closest_frame, frame middles, labels() sorted by str) restated from
pyannote-core 6.0.1 (parity unpinned there: third-party, not vendored).

Everything is self-contained: plain numpy + torch CPU modules from oracle.models + scipy."""
from __future__ import annotations

import itertools
import math
import string
from dataclasses import dataclass
from typing import List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F
from scipy.cluster.hierarchy import fcluster, linkage
from scipy.spatial.distance import cdist

from .models import Powerset


@dataclass
class SW:
    """pyannote.core.SlidingWindow (start, duration, step)"""
    start: float
    duration: float
    step: float

    def closest_frame(self, t: float) -> int:
        return int(np.rint((t - self.start - 0.5 * self.duration) / self.step))

    def segment(self, i: int) -> Tuple[float, float]:
        s = self.start + i * self.step
        return s, s + self.duration

    def middle(self, i: int) -> float:
        s, e = self.segment(i)
        return 0.5 * (s + e)


def receptive_field(model, sample_rate: int = 16000) -> SW:
    """core/model.py:168-184 for SincNet kernels [251,3,5,3,5,3] / strides [10,3,1,3,1,3] (PyanNet), or the
    wav2vec feature extractor's convolutions (SSeRiouSS.py:217-287)."""
    if hasattr(model, "wav2vec"):
        layers = model.wav2vec.feature_extractor.conv_layers
        ks, ss = [l.kernel_size for l in layers], [l.stride for l in layers]
    else:
        ks, ss = [251, 3, 5, 3, 5, 3], [model.sincnet.stride, 3, 1, 3, 1, 3]

    def size(n):
        for k, s in reversed(list(zip(ks, ss))):
            n = 1 + (k - 1) + (n - 1) * s
        return n

    def center(f):
        for k, s in reversed(list(zip(ks, ss))):
            f = f * s + (k - 1) // 2
        return f

    rf_size = size(1)
    rf_step = size(2) - rf_size
    rf_start = center(0) - (rf_size - 1) / 2
    return SW(rf_start / sample_rate, rf_size / sample_rate, rf_step / sample_rate)


def slide(model, waveform: torch.Tensor, sample_rate: int, duration: float, step: float,
          batch_size: int = 32) -> np.ndarray:
    """Inference.slide with skip_aggregation=True -> (C, F, S) float32: {0,1} hard multilabel for a
    powerset model (inference.py:130-136, 210-215), the sigmoid scores themselves for a multi-label one."""
    window_size = round(duration * sample_rate)
    step_size = round(step * sample_rate)
    _, num_samples = waveform.shape
    conversion = Powerset(3, 2) if getattr(model, "powerset", True) else (lambda x: x)
    outputs = []
    if num_samples >= window_size:
        chunks = waveform.unfold(1, window_size, step_size).permute(1, 0, 2)
        num_chunks = chunks.shape[0]
    else:
        num_chunks = 0
    has_last_chunk = (num_samples < window_size) or (num_samples - window_size) % step_size > 0
    with torch.inference_mode():
        for c in np.arange(0, num_chunks, batch_size):
            batch = chunks[c: c + batch_size]
            outputs.append(conversion(model(batch)).cpu().numpy())
        if has_last_chunk:
            last_chunk = waveform[:, num_chunks * step_size:]
            last_chunk = F.pad(last_chunk, (0, window_size - last_chunk.shape[1]))
            outputs.append(conversion(model(last_chunk[None])).cpu().numpy())
    return np.vstack(outputs)


def aggregate(scores: np.ndarray, chunks: SW, frames: SW, warm_up=(0.0, 0.0), epsilon=1e-12,
              hamming=False, missing=np.nan, skip_average=False) -> Tuple[np.ndarray, SW]:
    num_chunks, num_frames_per_chunk, num_classes = scores.shape
    frames = SW(chunks.start, frames.duration, frames.step)
    hamming_window = np.hamming(num_frames_per_chunk).reshape(-1, 1) if hamming \
        else np.ones((num_frames_per_chunk, 1))
    warm_up_window = np.ones((num_frames_per_chunk, 1))
    warm_up_left = round(warm_up[0] / chunks.duration * num_frames_per_chunk)
    warm_up_window[:warm_up_left] = epsilon
    warm_up_right = round(warm_up[1] / chunks.duration * num_frames_per_chunk)
    warm_up_window[num_frames_per_chunk - warm_up_right:] = epsilon
    num_frames = frames.closest_frame(
        chunks.start + chunks.duration + (num_chunks - 1) * chunks.step + 0.5 * frames.duration) + 1
    aggregated_output = np.zeros((num_frames, num_classes), dtype=np.float32)
    overlapping_chunk_count = np.zeros((num_frames, num_classes), dtype=np.float32)
    aggregated_mask = np.zeros((num_frames, num_classes), dtype=np.float32)
    for c in range(num_chunks):
        score = np.array(scores[c])
        chunk_start, _ = chunks.segment(c)
        mask = 1 - np.isnan(score)
        np.nan_to_num(score, copy=False, nan=0.0)
        start_frame = frames.closest_frame(chunk_start + 0.5 * frames.duration)
        sl = slice(start_frame, start_frame + num_frames_per_chunk)
        aggregated_output[sl] += score * mask * hamming_window * warm_up_window
        overlapping_chunk_count[sl] += mask * hamming_window * warm_up_window
        aggregated_mask[sl] = np.maximum(aggregated_mask[sl], mask)
    if skip_average:
        average = aggregated_output
    else:
        average = aggregated_output / np.maximum(overlapping_chunk_count, epsilon)
    average[aggregated_mask == 0.0] = missing
    return average, frames


def hysteresis(scores: np.ndarray, onset: float = 0.5, offset: Optional[float] = None,
               initial_state=None) -> np.ndarray:
    """`binarize` for (C, F, K) scores (utils/signal.py:78-204): per (chunk, class) on where score > onset,
    off where score < offset, the previous state in between; NaN -> 0; `initial_state` None means
    scores[:, 0] >= (onset + offset) / 2.  Written as the state machine the reference's index arithmetic
    (`same_as` / `well_defined_idx`) evaluates."""
    offset = offset or onset
    C, F_, K = scores.shape
    data = np.nan_to_num(np.transpose(scores, (0, 2, 1)).reshape(C * K, F_))
    if initial_state is None:
        state = data[:, 0] >= 0.5 * (onset + offset)
    else:
        state = np.full(C * K, bool(initial_state))
    out = np.zeros((C * K, F_), dtype=bool)
    state = state.copy()
    for f in range(F_):
        col = data[:, f]
        state = np.where(col > onset, True, np.where(col < offset, False, state))
        out[:, f] = state
    return 1.0 * np.transpose(out.reshape(C, K, F_), (0, 2, 1))


def speaker_count(binarized: np.ndarray, chunks: SW, frames: SW) -> Tuple[np.ndarray, SW]:
    """warm_up=(0,0): trim is the identity (diarization.py:176)."""
    summed = np.sum(binarized, axis=-1, keepdims=True)
    count, fr = aggregate(summed, chunks, frames, hamming=False, missing=0.0, skip_average=False)
    return np.rint(count).astype(np.uint8), fr


def get_embeddings(emb_model, waveform: torch.Tensor, binarized: np.ndarray, chunks: SW,
                   sample_rate: int = 16000, exclude_overlap: bool = False, batch_size: int = 32,
                   min_num_samples: int = 400) -> np.ndarray:
    """one forward pass per (chunk, speaker), batched by `batch_size` like the reference."""
    duration = chunks.duration
    num_chunks, num_frames, num_speakers = binarized.shape
    if exclude_overlap:
        num_samples = duration * sample_rate
        min_num_frames = math.ceil(num_frames * min_num_samples / num_samples)
        clean_frames = 1.0 * (np.sum(binarized, axis=2, keepdims=True) < 2)
        clean = binarized * clean_frames
    else:
        min_num_frames = -1
        clean = binarized
    window = round(duration * sample_rate)

    def iter_waveform_and_mask():
        for c in range(num_chunks):
            start, _ = chunks.segment(c)
            s = math.floor(start * sample_rate)
            w = waveform[:, s: s + window]
            if w.shape[1] < window:
                w = F.pad(w, (0, window - w.shape[1]))
            masks = np.nan_to_num(binarized[c], nan=0.0).astype(np.float32)
            clean_masks = np.nan_to_num(clean[c], nan=0.0).astype(np.float32)
            for mask, clean_mask in zip(masks.T, clean_masks.T):
                used = clean_mask if np.sum(clean_mask) > min_num_frames else mask
                yield w[None], torch.from_numpy(used)[None]

    out = []
    it = iter_waveform_and_mask()
    with torch.inference_mode():
        while True:
            batch = list(itertools.islice(it, batch_size))
            if not batch:
                break
            waveforms, masks = zip(*batch)
            out.append(emb_model(torch.vstack(waveforms), weights=torch.vstack(masks)).cpu().numpy())
    emb = np.vstack(out)
    return emb.reshape(num_chunks, num_speakers, -1)


def filter_embeddings(embeddings, segmentations, min_active_ratio=0.2):
    _, num_frames, _ = segmentations.shape
    single_active_mask = (np.sum(segmentations, axis=2, keepdims=True) == 1)
    num_clean_frames = np.sum(segmentations * single_active_mask, axis=1)
    active = num_clean_frames >= min_active_ratio * num_frames
    valid = ~np.any(np.isnan(embeddings), axis=2)
    chunk_idx, speaker_idx = np.where(active * valid)
    return embeddings[chunk_idx, speaker_idx], chunk_idx, speaker_idx


def set_num_clusters(num_embeddings, num_clusters=None, min_clusters=None, max_clusters=None):
    min_clusters = num_clusters or min_clusters or 1
    min_clusters = max(1, min(num_embeddings, min_clusters))
    max_clusters = num_clusters or max_clusters or num_embeddings
    max_clusters = max(1, min(num_embeddings, max_clusters))
    if min_clusters > max_clusters:
        raise ValueError("min_clusters must be smaller than (or equal to) max_clusters")
    if min_clusters == max_clusters:
        num_clusters = min_clusters
    return num_clusters, min_clusters, max_clusters


def ahc_cluster(embeddings, min_clusters, max_clusters, num_clusters=None, *, method="centroid",
                threshold=0.7045654963945799, min_cluster_size=12, metric="cosine"):
    """AgglomerativeClustering.cluster (clustering.py:330-480)"""
    num_embeddings, _ = embeddings.shape
    min_cluster_size = min(min_cluster_size, max(1, round(0.1 * num_embeddings)))
    if num_embeddings == 1:
        return np.zeros((1,), dtype=np.uint8)
    if metric == "cosine" and method in ["centroid", "median", "ward"]:
        with np.errstate(divide="ignore", invalid="ignore"):
            embeddings /= np.linalg.norm(embeddings, axis=-1, keepdims=True)
        dendrogram = linkage(embeddings, method=method, metric="euclidean")
    else:
        dendrogram = linkage(embeddings, method=method, metric=metric)
    clusters = fcluster(dendrogram, threshold, criterion="distance") - 1
    cluster_unique, cluster_counts = np.unique(clusters, return_counts=True)
    large_clusters = cluster_unique[cluster_counts >= min_cluster_size]
    num_large_clusters = len(large_clusters)
    if num_large_clusters < min_clusters:
        num_clusters = min_clusters
    elif num_large_clusters > max_clusters:
        num_clusters = max_clusters
    if num_clusters is not None and num_large_clusters != num_clusters:
        _dendrogram = np.copy(dendrogram)
        _dendrogram[:, 2] = np.arange(num_embeddings - 1)
        best_iteration = num_embeddings - 1
        best_num_large_clusters = 1
        for iteration in np.argsort(np.abs(dendrogram[:, 2] - threshold)):
            new_cluster_size = _dendrogram[iteration, 3]
            if new_cluster_size < min_cluster_size:
                continue
            clusters = fcluster(_dendrogram, iteration, criterion="distance") - 1
            cluster_unique, cluster_counts = np.unique(clusters, return_counts=True)
            large_clusters = cluster_unique[cluster_counts >= min_cluster_size]
            num_large_clusters = len(large_clusters)
            if abs(num_large_clusters - num_clusters) < abs(best_num_large_clusters - num_clusters):
                best_iteration = iteration
                best_num_large_clusters = num_large_clusters
            if num_large_clusters == num_clusters:
                break
        if best_num_large_clusters != num_clusters:
            clusters = fcluster(_dendrogram, best_iteration, criterion="distance") - 1
            cluster_unique, cluster_counts = np.unique(clusters, return_counts=True)
            large_clusters = cluster_unique[cluster_counts >= min_cluster_size]
            num_large_clusters = len(large_clusters)
    if num_large_clusters == 0:
        clusters[:] = 0
        return clusters
    small_clusters = cluster_unique[cluster_counts < min_cluster_size]
    if len(small_clusters) == 0:
        return clusters
    large_centroids = np.vstack([np.mean(embeddings[clusters == k], axis=0) for k in large_clusters])
    small_centroids = np.vstack([np.mean(embeddings[clusters == k], axis=0) for k in small_clusters])
    centroids_cdist = cdist(large_centroids, small_centroids, metric=metric)
    for small_k, large_k in enumerate(np.argmin(centroids_cdist, axis=0)):
        clusters[clusters == small_clusters[small_k]] = large_clusters[large_k]
    _, clusters = np.unique(clusters, return_inverse=True)
    return clusters


def assign_embeddings(embeddings, train_chunk_idx, train_speaker_idx, train_clusters, metric="cosine"):
    num_clusters = np.max(train_clusters) + 1
    num_chunks, num_speakers, dimension = embeddings.shape
    train_embeddings = embeddings[train_chunk_idx, train_speaker_idx]
    centroids = np.vstack([np.mean(train_embeddings[train_clusters == k], axis=0)
                           for k in range(num_clusters)])
    e2k_distance = cdist(embeddings.reshape(-1, dimension), centroids, metric=metric).reshape(
        num_chunks, num_speakers, num_clusters)
    soft_clusters = 2 - e2k_distance
    hard_clusters = np.argmax(soft_clusters, axis=2)
    return hard_clusters, soft_clusters, centroids


def clustering(embeddings, segmentations, num_clusters=None, min_clusters=None, max_clusters=None,
               **hyper):
    """BaseClustering.__call__ (clustering.py:214-289)"""
    train_embeddings, train_chunk_idx, train_speaker_idx = filter_embeddings(embeddings, segmentations)
    num_embeddings, _ = train_embeddings.shape
    num_clusters, min_clusters, max_clusters = set_num_clusters(num_embeddings, num_clusters,
                                                                min_clusters, max_clusters)
    if max_clusters < 2:
        num_chunks, num_speakers, _ = embeddings.shape
        return (np.zeros((num_chunks, num_speakers), dtype=np.int8),
                np.ones((num_chunks, num_speakers, 1)), np.mean(train_embeddings, axis=0, keepdims=True))
    train_clusters = ahc_cluster(train_embeddings, min_clusters, max_clusters, num_clusters, **hyper)
    return assign_embeddings(embeddings, train_chunk_idx, train_speaker_idx, train_clusters)


def to_diarization(clustered: np.ndarray, chunks: SW, count: np.ndarray, frames: SW) -> np.ndarray:
    activations, _ = aggregate(clustered, chunks, frames, hamming=False, missing=0.0, skip_average=True)
    _, num_speakers = activations.shape
    max_speakers_per_frame = np.max(count)
    if num_speakers < max_speakers_per_frame:
        activations = np.pad(activations, ((0, 0), (0, max_speakers_per_frame - num_speakers)))
    n = min(len(activations), len(count))  # same sliding window: the common extent
    activations, count = activations[:n], count[:n]
    sorted_speakers = np.argsort(-activations, axis=-1)
    binary = np.zeros_like(activations)
    for t in range(n):
        for i in range(int(count[t, 0])):
            binary[t, sorted_speakers[t, i]] = 1.0
    return binary


def reconstruct(segmentations: np.ndarray, chunks: SW, hard_clusters: np.ndarray, count: np.ndarray,
                frames: SW) -> np.ndarray:
    num_chunks, num_frames, local_num_speakers = segmentations.shape
    num_clusters = np.max(hard_clusters) + 1
    clustered = np.nan * np.zeros((num_chunks, num_frames, num_clusters))
    for c, (cluster, segmentation) in enumerate(zip(hard_clusters, segmentations)):
        for k in np.unique(cluster):
            if k == -2:
                continue
            clustered[c, :, k] = np.max(segmentation[:, cluster == k], axis=1)
    return to_diarization(clustered, chunks, count, frames)


def _string_generator():
    r = 1
    while True:
        for c in itertools.product(string.ascii_uppercase, repeat=r):
            yield "".join(c)
        r += 1


def binarize(scores: np.ndarray, frames: SW, onset=0.5, offset=0.5) -> List[Tuple[float, float, str, int]]:
    """Binarize.__call__ (signal.py:254-318), min_duration_on/off = 0, no padding ->
    tracks (start, end, track, label) in pyannote.core iteration order."""
    num_frames, num_classes = scores.shape
    timestamps = [frames.middle(i) for i in range(num_frames)]
    tracks = []
    gen = _string_generator()
    for k, k_scores in enumerate(scores.T):
        track = next(gen)
        start = timestamps[0]
        is_active = k_scores[0] > onset
        t = timestamps[0]
        for t, y in zip(timestamps[1:], k_scores[1:]):
            if is_active:
                if y < offset:
                    tracks.append((start, t, track, k))
                    start = t
                    is_active = False
            else:
                if y > onset:
                    start = t
                    is_active = True
        if is_active:
            tracks.append((start, t, track, k))
    tracks = [tr for tr in tracks if (tr[1] - tr[0]) > 1e-6]  # Annotation ignores empty segments
    tracks.sort(key=lambda tr: (tr[0], tr[1], str(tr[2]), str(tr[3])))
    return tracks


@dataclass
class OracleOutput:
    diarization: List[Tuple[float, float, str]]
    exclusive_diarization: List[Tuple[float, float, str]]
    centroids: Optional[np.ndarray]
    segmentations: np.ndarray
    count: np.ndarray
    embeddings: Optional[np.ndarray]
    hard_clusters: Optional[np.ndarray]
    timings: dict
    raw_segmentations: Optional[np.ndarray] = None     # what Inference.slide returned (soft for non-powerset)


def diarize(seg_model, emb_model, waveform: torch.Tensor, sample_rate: int = 16000,
            duration: float = 10.0, segmentation_step: float = 0.1, exclude_overlap: bool = True,
            segmentation_batch_size: int = 32, embedding_batch_size: int = 32,
            num_speakers=None, min_speakers=None, max_speakers=None,
            method="centroid", threshold=0.7045654963945799, min_cluster_size=12,
            segmentation_threshold: float = 0.5, min_num_samples: int = 400, cluster=None) -> OracleOutput:
    """SpeakerDiarization.apply (speaker_diarization.py:530-784) for the 3.1 configuration.  `cluster`: another
    clustering step with the call contract of `self.clustering(...)` there (:660-668) -- cluster(embeddings,
    segmentations, num_clusters=, min_clusters=, max_clusters=) -> (hard, soft, centroids) -- e.g. oracle.vbx.vbx_clustering
    bound to a PLDA for the 4.x / community-1 configuration; default: agglomerative clustering with the arguments above."""
    import time
    timings = {}
    min_speakers_ = num_speakers or min_speakers or 1
    max_speakers_ = num_speakers or max_speakers or np.inf
    if min_speakers_ == max_speakers_:
        num_speakers = min_speakers_
    chunks = SW(0.0, duration, segmentation_step * duration)
    frames = receptive_field(seg_model, sample_rate)

    t0 = time.perf_counter()
    raw = slide(seg_model, waveform, sample_rate, duration, chunks.step, segmentation_batch_size)
    # non-powerset models: hysteresis thresholding (:599-606); the RAW scores are reconstructed from (:687)
    segmentations = raw if getattr(seg_model, "powerset", True) else \
        hysteresis(raw, onset=segmentation_threshold, initial_state=False).astype(np.float32)
    timings["segmentation"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    count, count_frames = speaker_count(segmentations, chunks, frames)
    timings["speaker_counting"] = time.perf_counter() - t0
    if np.nanmax(count) == 0.0:
        return OracleOutput([], [], np.zeros((0, 256)), segmentations, count, None, None, timings)
    t0 = time.perf_counter()
    embeddings = get_embeddings(emb_model, waveform, segmentations, chunks, sample_rate,
                                exclude_overlap=exclude_overlap, batch_size=embedding_batch_size,
                                min_num_samples=min_num_samples)
    timings["embeddings"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    if cluster is not None:
        hard_clusters, _, centroids = cluster(np.array(embeddings), segmentations, num_clusters=num_speakers,
                                              min_clusters=min_speakers_, max_clusters=max_speakers_)
    else:
        hard_clusters, _, centroids = clustering(
            np.array(embeddings), segmentations, num_clusters=num_speakers, min_clusters=min_speakers_,
            max_clusters=max_speakers_, method=method, threshold=threshold,
            min_cluster_size=min_cluster_size)
    timings["clustering"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    count = np.minimum(count, max_speakers_).astype(np.int8)
    inactive_speakers = np.sum(segmentations, axis=1) == 0
    hard_clusters[inactive_speakers] = -2
    discrete = reconstruct(raw, chunks, hard_clusters, count, count_frames)
    tracks = binarize(discrete, count_frames)
    count1 = np.minimum(count, 1).astype(np.int8)
    exclusive = reconstruct(raw, chunks, hard_clusters, count1, count_frames)
    ex_tracks = binarize(exclusive, count_frames)
    labels = sorted({tr[3] for tr in tracks}, key=str)
    mapping = {label: f"SPEAKER_{i:02d}" for i, label in enumerate(labels)}
    diar = [(s, e, mapping.get(l, l)) for s, e, _, l in tracks]
    exdiar = [(s, e, mapping.get(l, l)) for s, e, _, l in ex_tracks]
    if len(labels) > centroids.shape[0]:
        centroids = np.pad(centroids, ((0, len(labels) - centroids.shape[0]), (0, 0)))
    inverse_mapping = {label: index for index, label in mapping.items()}
    new_labels = sorted(mapping.values(), key=str)
    centroids = centroids[[inverse_mapping[label] for label in new_labels]]
    timings["reconstruction"] = time.perf_counter() - t0
    return OracleOutput(diar, exdiar, centroids, segmentations, count, embeddings, hard_clusters, timings, raw)
