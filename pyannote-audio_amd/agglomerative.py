"""Embedding clustering of the 3.1 pipeline (mirrors pipelines/clustering.py: BaseClustering :44-289,
AgglomerativeClustering :292-480, Clustering enum :759-763).

Arithmetic contract ("Clustering dtypes"): embeddings arrive as float32; the
training copy is L2-normalised in float32; pairwise Euclidean distances and the cosine distances to
centroids are float64 exactly as SciPy computes them (sequential-k summation), so the dendrogram, the
flat clusters and the arg-max assignment are bit-identical to the reference given identical embeddings.
The O(N^2 D) distance work runs on the GPU (`pa_pdist_f64`, `pa_cdist_cosine_f64`) when the inputs are
large enough and a device is available; the serial dendrogram merge stays in SciPy's `linkage`
(which accepts the condensed matrix: `linkage(pdist(X)) == linkage(X, metric="euclidean")`)."""
from __future__ import annotations

import time
from enum import Enum
from typing import Optional

import numpy as np
from scipy.cluster.hierarchy import fcluster, linkage
from scipy.spatial.distance import cdist

from . import distance
from .core import SlidingWindowFeature
from .pipeline import Categorical, Integer, Pipeline, Uniform


class BaseClustering(Pipeline):
    def __init__(self, metric: str = "cosine", constrained_assignment: bool = False):
        super().__init__()
        self.metric = metric
        self.constrained_assignment = constrained_assignment
        self.device = None
        object.__setattr__(self, "timings", {})   # wall seconds of the last call, per sub-step

    def to(self, device):
        self.device = device
        return self

    def set_num_clusters(self, num_embeddings: int, num_clusters: Optional[int] = None,
                         min_clusters: Optional[int] = None, max_clusters: Optional[int] = None):
        """clustering.py:54-75"""
        min_clusters = num_clusters or min_clusters or 1
        min_clusters = max(1, min(num_embeddings, min_clusters))
        max_clusters = num_clusters or max_clusters or num_embeddings
        max_clusters = max(1, min(num_embeddings, max_clusters))
        if min_clusters > max_clusters:
            raise ValueError(f"min_clusters must be smaller than (or equal to) max_clusters "
                             f"(here: min_clusters={min_clusters:g} and max_clusters={max_clusters:g}).")
        if min_clusters == max_clusters:
            num_clusters = min_clusters
        return num_clusters, min_clusters, max_clusters

    def filter_embeddings(self, embeddings: np.ndarray, segmentations: SlidingWindowFeature,
                          min_active_ratio: float = 0.2, num_clean_frames: Optional[np.ndarray] = None):
        """clustering.py:77-125: keep (chunk, speaker) pairs that speak alone for >= 20 % of the chunk
        and whose embedding is finite.  `num_clean_frames` (C, S): the per-(chunk, speaker) count of
        single-speaker frames when the caller already has it (pa_seg_chunk_stats on the GPU)."""
        seg = segmentations.data
        _, num_frames, _ = seg.shape
        if num_clean_frames is not None:
            num_clean = num_clean_frames
        else:
            single = np.sum(seg, axis=2, keepdims=True) == 1
            num_clean = np.sum(seg * single, axis=1)
        active = num_clean >= min_active_ratio * num_frames
        valid = ~np.any(np.isnan(embeddings), axis=2)
        chunk_idx, speaker_idx = np.where(active * valid)
        return embeddings[chunk_idx, speaker_idx], chunk_idx, speaker_idx

    def constrained_argmax(self, soft_clusters: np.ndarray) -> np.ndarray:
        from scipy.optimize import linear_sum_assignment
        soft_clusters = np.nan_to_num(soft_clusters, nan=np.nanmin(soft_clusters))
        num_chunks, num_speakers, _ = soft_clusters.shape
        hard = -2 * np.ones((num_chunks, num_speakers), dtype=np.int8)
        for c, cost in enumerate(soft_clusters):
            speakers, clusters = linear_sum_assignment(cost, maximize=True)
            hard[c, speakers] = clusters
        return hard

    def assign_embeddings(self, embeddings: np.ndarray, train_chunk_idx: np.ndarray,
                          train_speaker_idx: np.ndarray, train_clusters: np.ndarray,
                          constrained: bool = False):
        """clustering.py:142-212: centroids = float32 means of the (un-normalised) training
        embeddings; soft = 2 - cosine distance (float64); hard = arg-max (first maximum wins)."""
        num_clusters = np.max(train_clusters) + 1
        num_chunks, num_speakers, dimension = embeddings.shape
        train_embeddings = embeddings[train_chunk_idx, train_speaker_idx]
        centroids = np.vstack([np.mean(train_embeddings[train_clusters == k], axis=0)
                               for k in range(num_clusters)])
        flat = embeddings.reshape(num_chunks * num_speakers, dimension)
        e2k = distance.cdist(flat, centroids, metric=self.metric, device=self.device)
        soft = 2 - e2k.reshape(num_chunks, num_speakers, num_clusters)
        hard = self.constrained_argmax(soft) if constrained else np.argmax(soft, axis=2)
        return hard, soft, centroids

    def __call__(self, embeddings: np.ndarray, segmentations: Optional[SlidingWindowFeature] = None,
                 num_clusters: Optional[int] = None, min_clusters: Optional[int] = None,
                 max_clusters: Optional[int] = None, **kwargs):
        """clustering.py:214-289"""
        self.timings.clear()
        t0 = time.perf_counter()
        train_embeddings, train_chunk_idx, train_speaker_idx = self.filter_embeddings(
            embeddings, segmentations=segmentations,
            num_clean_frames=kwargs.get("num_clean_frames", None))
        self.timings["filter"] = time.perf_counter() - t0
        num_embeddings, _ = train_embeddings.shape
        num_clusters, min_clusters, max_clusters = self.set_num_clusters(
            num_embeddings, num_clusters=num_clusters, min_clusters=min_clusters,
            max_clusters=max_clusters)
        if max_clusters < 2:
            num_chunks, num_speakers, _ = embeddings.shape
            hard = np.zeros((num_chunks, num_speakers), dtype=np.int8)
            soft = np.ones((num_chunks, num_speakers, 1))
            centroids = np.mean(train_embeddings, axis=0, keepdims=True)
            return hard, soft, centroids
        t0 = time.perf_counter()
        train_clusters = self.cluster(train_embeddings, min_clusters=min_clusters,
                                      max_clusters=max_clusters, num_clusters=num_clusters)
        t1 = time.perf_counter()
        out = self.assign_embeddings(embeddings, train_chunk_idx, train_speaker_idx, train_clusters,
                                     constrained=self.constrained_assignment)
        self.timings.update(cluster=t1 - t0, assign=time.perf_counter() - t1)
        return out


class AgglomerativeClustering(BaseClustering):
    """clustering.py:292-480.  Hyper-parameters: method, threshold, min_cluster_size."""

    expects_num_clusters: bool = False

    def __init__(self, metric: str = "cosine", constrained_assignment: bool = False):
        super().__init__(metric=metric, constrained_assignment=constrained_assignment)
        self.threshold = Uniform(0.0, 2.0)
        self.method = Categorical(["average", "centroid", "complete", "median", "single", "ward",
                                   "weighted"])
        self.min_cluster_size = Integer(1, 20)

    def dendrogram(self, embeddings: np.ndarray) -> np.ndarray:
        """linkage step (:368-382).  NOTE: normalises `embeddings` in place, like the reference."""
        if self.metric == "cosine" and self.method in ["centroid", "median", "ward"]:
            with np.errstate(divide="ignore", invalid="ignore"):
                embeddings /= np.linalg.norm(embeddings, axis=-1, keepdims=True)
            t0 = time.perf_counter()
            if (self.method == "centroid" and len(embeddings) >= 2 and self.device is not None
                    and getattr(self.device, "type", None) == "cuda"):
                if not np.isfinite(embeddings).all():
                    # scipy.cluster.hierarchy.linkage(X, ...) validates its input the same way; without
                    # the check a zero-norm / inf embedding (NaN after the normalisation) would walk the
                    # merge kernel out of bounds
                    raise ValueError("The condensed distance matrix must contain only finite values.")
                Z = distance.linkage_centroid(embeddings, self.device)
                self.timings.update(pdist=0.0, linkage=time.perf_counter() - t0,
                                    num_embeddings=len(embeddings))
                return Z
            condensed = distance.pdist_euclidean(embeddings, device=self.device)
            t1 = time.perf_counter()
            Z = linkage(condensed, method=self.method)
            self.timings.update(pdist=t1 - t0, linkage=time.perf_counter() - t1,
                                num_embeddings=len(embeddings))
            return Z
        if self.metric == "euclidean":
            condensed = distance.pdist_euclidean(embeddings, device=self.device)
            return linkage(condensed, method=self.method)
        return linkage(embeddings, method=self.method, metric=self.metric)

    def cluster(self, embeddings: np.ndarray, min_clusters: Optional[int] = None,
                max_clusters: Optional[int] = None, num_clusters: Optional[int] = None):
        num_embeddings, _ = embeddings.shape
        min_cluster_size = min(self.min_cluster_size, max(1, round(0.1 * num_embeddings)))
        if num_embeddings == 1:
            return np.zeros((1,), dtype=np.uint8)
        dendrogram = self.dendrogram(embeddings)
        clusters = fcluster(dendrogram, self.threshold, criterion="distance") - 1

        def large(cl):
            unique, counts = np.unique(cl, return_counts=True)
            return unique, counts, unique[counts >= min_cluster_size]

        cluster_unique, cluster_counts, large_clusters = large(clusters)
        num_large_clusters = len(large_clusters)
        if num_large_clusters < min_clusters:
            num_clusters = min_clusters
        elif num_large_clusters > max_clusters:
            num_clusters = max_clusters

        if num_clusters is not None and num_large_clusters != num_clusters:
            # walk the dendrogram away from the threshold, by iteration index (:405-451)
            _dendrogram = np.copy(dendrogram)
            _dendrogram[:, 2] = np.arange(num_embeddings - 1)
            best_iteration = num_embeddings - 1
            best_num_large_clusters = 1
            for iteration in np.argsort(np.abs(dendrogram[:, 2] - self.threshold)):
                if _dendrogram[iteration, 3] < min_cluster_size:
                    continue
                clusters = fcluster(_dendrogram, iteration, criterion="distance") - 1
                cluster_unique, cluster_counts, large_clusters = large(clusters)
                num_large_clusters = len(large_clusters)
                if abs(num_large_clusters - num_clusters) < abs(best_num_large_clusters - num_clusters):
                    best_iteration = iteration
                    best_num_large_clusters = num_large_clusters
                if num_large_clusters == num_clusters:
                    break
            if best_num_large_clusters != num_clusters:
                clusters = fcluster(_dendrogram, best_iteration, criterion="distance") - 1
                cluster_unique, cluster_counts, large_clusters = large(clusters)
                num_large_clusters = len(large_clusters)
                print(f"Found only {num_large_clusters} clusters. Using a smaller value than "
                      f"{min_cluster_size} for `min_cluster_size` might help.")

        if num_large_clusters == 0:
            clusters[:] = 0
            return clusters
        small_clusters = cluster_unique[cluster_counts < min_cluster_size]
        if len(small_clusters) == 0:
            return clusters
        # merge every small cluster into the most similar large one (centroid cosine distance)
        large_centroids = np.vstack([np.mean(embeddings[clusters == k], axis=0) for k in large_clusters])
        small_centroids = np.vstack([np.mean(embeddings[clusters == k], axis=0) for k in small_clusters])
        centroids_cdist = cdist(large_centroids, small_centroids, metric=self.metric)
        for small_k, large_k in enumerate(np.argmin(centroids_cdist, axis=0)):
            clusters[clusters == small_clusters[small_k]] = large_clusters[large_k]
        _, clusters = np.unique(clusters, return_inverse=True)
        return clusters


def _not_built(name):
    class _Missing(BaseClustering):
        def __init__(self, *a, **k):
            raise NotImplementedError(
                f"{name} is outside the accelerated 3.1 hot path (SURVEY.md section 8f); only "
                "AgglomerativeClustering is built.")
    _Missing.__name__ = name
    return _Missing


class Clustering(Enum):
    """clustering.py:759-763"""
    AgglomerativeClustering = AgglomerativeClustering
    KMeansClustering = _not_built("KMeansClustering")
    VBxClustering = _not_built("VBxClustering")
    OracleClustering = _not_built("OracleClustering")
