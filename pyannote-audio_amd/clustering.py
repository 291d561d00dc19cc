"""Embedding clustering back ends of the diarization pipeline (the b4 interface of SURVEY.md section 8b:
`Clustering[name].value(metric=...)(embeddings, segmentations, num_clusters, min_clusters,
max_clusters) -> (hard (C,S), soft (C,S,K), centroids (K,D))`, pipelines/clustering.py:44-763).

What runs where
  GPU (libpyannote_amd.so)   float64 pdist + the whole centroid-linkage merge (`distance.linkage_centroid`,
                             bit-identical to SciPy incl. ties), cosine cdist of every embedding to the
                             centroids, the PLDA projection and the VB iterations of VBx
  host, SciPy / sklearn      `fcluster` on the (GPU-built) dendrogram, Hungarian assignment, KMeans --
                             the very library calls the reference makes, O(N) or tiny
  host, this file            everything between those calls, organised around array operations instead
                             of the reference's per-cluster / per-iteration Python loops:
                               * `segment_means`: all cluster centroids from one stable sort (contiguous
                                 slices instead of K boolean masks), bit-identical to
                                 `np.mean(X[labels == k], axis=0)`; on a GPU the centroids of the final
                                 assignment come from `distance.centroid_means` (`pa_centroid_means`: the
                                 same sums in the same order on the embeddings that are still in HBM)
                               * `Dendrogram.large_cluster_counts`: the number of large clusters after
                                 EVERY merge from one O(N) scan of the merge sizes, which turns the
                                 reference's forced-number search (one `fcluster` per candidate cut,
                                 clustering.py:405-451) into a lookup + ONE `fcluster`
Arithmetic contract: embeddings are float32; the AHC training copy is
L2-normalised in float32; distances are float64 with SciPy's summation order; centroids are float32 row
sums in row order divided in float32.  Cluster ids are bit-identical to the reference's given identical
embeddings (oracle: oracle/pipeline.py, oracle/vbx.py hold the loop-for-loop restatement)."""
from __future__ import annotations

import time
import warnings
from enum import Enum
from typing import Optional, Tuple

import numpy as np
import torch
from scipy.cluster.hierarchy import fcluster, linkage
from scipy.spatial.distance import cdist as scipy_cdist

from . import distance
from .core import SlidingWindowFeature
from .pipeline import Categorical, Integer, Pipeline, Uniform


# ---------------------------------------------------------------------------------------------------
# array helpers
# ---------------------------------------------------------------------------------------------------
def segment_means(X: np.ndarray, labels: np.ndarray, num_segments: int) -> np.ndarray:
    """row means of X per label 0..num_segments-1, bit-identical to
    `np.vstack([np.mean(X[labels == k], axis=0) for k in range(num_segments)])`: ONE stable sort groups
    the rows (original order kept inside a segment), then each segment is a contiguous slice reduced by
    `np.add.reduce(axis=0)` -- numpy adds the rows of a C-contiguous block top to bottom in X's dtype,
    which is what `np.mean` does (`np.add.reduceat` and `np.add.at` do NOT keep that order) -- and divided
    in X's dtype.  An empty segment yields NaN."""
    order = np.argsort(labels, kind="stable")
    counts = np.bincount(labels, minlength=num_segments)[:num_segments]
    ends = np.cumsum(counts)
    grouped = X[order]
    out = np.full((num_segments, X.shape[1]), np.nan, dtype=X.dtype)
    for k in np.nonzero(counts)[0]:
        out[k] = np.add.reduce(grouped[ends[k] - counts[k]:ends[k]], axis=0) / X.dtype.type(counts[k])
    return out


def clamp_cluster_bounds(num_items: int, num_clusters, min_clusters, max_clusters
                         ) -> Tuple[Optional[int], int, int]:
    """resolve (num, min, max) against the number of items (clustering.py:54-75): an explicit number
    pins both bounds, both are clipped to [1, num_items], and equal bounds become an explicit number."""
    lo = min(num_items, num_clusters or min_clusters or 1)
    hi = min(num_items, num_clusters or max_clusters or num_items)
    lo, hi = max(1, lo), max(1, hi)
    if lo > hi:
        raise ValueError(f"min_clusters must be smaller than (or equal to) max_clusters "
                         f"(here: min_clusters={lo:g} and max_clusters={hi:g}).")
    return (lo if lo == hi else num_clusters), lo, hi


class Dendrogram:
    """(N-1, 4) SciPy linkage matrix with the two queries the pipeline needs."""

    def __init__(self, Z: np.ndarray):
        self.Z = Z
        self.num_leaves = Z.shape[0] + 1

    def cut(self, height: float) -> np.ndarray:
        """flat clusters at `height`, numbered like fcluster(..., "distance") - 1"""
        return fcluster(self.Z, height, criterion="distance") - 1

    def cut_after_merge(self, index: int) -> np.ndarray:
        """flat clusters once merges 0..index have been applied (heights replaced by their rank, which
        also removes the inversions centroid linkage can produce: clustering.py:407-410)"""
        ranked = self.Z.copy()
        ranked[:, 2] = np.arange(self.num_leaves - 1)
        return fcluster(ranked, index, criterion="distance") - 1

    def large_cluster_counts(self, min_size: int) -> np.ndarray:
        """counts[i] = number of clusters with >= min_size members after merges 0..i."""
        Z, n = self.Z, self.num_leaves
        child_size = np.ones((n - 1, 2))
        for side in (0, 1):
            ids = Z[:, side].astype(np.int64)
            inner = ids >= n
            child_size[inner, side] = Z[ids[inner] - n, 3]
        delta = (Z[:, 3] >= min_size).astype(np.int64) - (child_size >= min_size).sum(axis=1)
        return (n if min_size <= 1 else 0) + np.cumsum(delta)


# ---------------------------------------------------------------------------------------------------
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
        return clamp_cluster_bounds(num_embeddings, num_clusters, min_clusters, max_clusters)

    def filter_embeddings(self, embeddings: np.ndarray, segmentations: SlidingWindowFeature,
                          min_active_ratio: float = 0.2, num_clean_frames: Optional[np.ndarray] = None):
        """training set = (chunk, speaker) pairs that speak ALONE for at least 20 % of the chunk and
        whose embedding has no NaN (clustering.py:77-125).  `num_clean_frames` (C, S): those frame
        counts when the caller already has them (pa_seg_chunk_stats on the GPU)."""
        num_frames = segmentations.data.shape[1]
        if num_clean_frames is None:
            seg = segmentations.data
            alone = seg.sum(axis=2, keepdims=True) == 1
            num_clean_frames = (seg * alone).sum(axis=1)
        keep = (num_clean_frames >= min_active_ratio * num_frames) & ~np.isnan(embeddings).any(axis=2)
        chunk_idx, speaker_idx = np.nonzero(keep)
        return embeddings[chunk_idx, speaker_idx], chunk_idx, speaker_idx

    def constrained_argmax(self, soft_clusters: np.ndarray) -> np.ndarray:
        """one cluster per local speaker and chunk, no cluster twice (Hungarian, clustering.py:127-140)"""
        from scipy.optimize import linear_sum_assignment
        scores = np.nan_to_num(soft_clusters, nan=np.nanmin(soft_clusters))
        hard = np.full(scores.shape[:2], -2, dtype=np.int8)
        for c, cost in enumerate(scores):
            speakers, clusters = linear_sum_assignment(cost, maximize=True)
            hard[c, speakers] = clusters
        return hard

    def _similarities(self, embeddings: np.ndarray, centroids: np.ndarray, device_embeddings=None) -> np.ndarray:
        """soft (C, S, K) = 2 - distance of every (chunk, speaker) embedding to every centroid.
        `device_embeddings`: the same (C, S, D) values as a device tensor, when the caller still has them there
        (saves the float64 conversion on the host and the upload)."""
        C, S, D = embeddings.shape
        A = embeddings.reshape(C * S, D)
        if device_embeddings is not None and tuple(device_embeddings.shape) == (C, S, D):
            A = device_embeddings.reshape(C * S, D)
        d = distance.cdist(A, centroids, metric=self.metric, device=self.device)
        return 2 - d.reshape(C, S, -1)

    def assign_embeddings(self, embeddings: np.ndarray, train_chunk_idx: np.ndarray,
                          train_speaker_idx: np.ndarray, train_clusters: np.ndarray,
                          constrained: bool = False, device_embeddings=None):
        """centroid of every cluster from the (un-normalised) training embeddings, then every
        (chunk, speaker) goes to its most similar centroid (clustering.py:142-212)"""
        K = int(np.max(train_clusters)) + 1
        C, S, D = embeddings.shape
        on_gpu = getattr(self.device, "type", None) == "cuda"
        if (on_gpu and device_embeddings is not None and tuple(device_embeddings.shape) == (C, S, D)
                and device_embeddings.dtype == torch.float32):
            # the embeddings are still in HBM: centroids there (pa_centroid_means), only (K, D) comes back
            centroids = distance.centroid_means(device_embeddings.reshape(C * S, D).contiguous(),
                                                train_chunk_idx * S + train_speaker_idx, train_clusters, K,
                                                self.device)
        else:
            centroids = segment_means(embeddings[train_chunk_idx, train_speaker_idx], train_clusters, K)
        soft = self._similarities(embeddings, centroids, device_embeddings)
        hard = self.constrained_argmax(soft) if constrained else np.argmax(soft, axis=2)
        return hard, soft, centroids

    def _single_cluster(self, embeddings: np.ndarray, train_embeddings: np.ndarray):
        C, S, _ = embeddings.shape
        return (np.zeros((C, S), dtype=np.int8), np.ones((C, S, 1)),
                np.mean(train_embeddings, axis=0, keepdims=True))

    def __call__(self, embeddings: np.ndarray, segmentations: Optional[SlidingWindowFeature] = None,
                 num_clusters: Optional[int] = None, min_clusters: Optional[int] = None,
                 max_clusters: Optional[int] = None, **kwargs):
        """clustering.py:214-289"""
        self.timings.clear()
        t0 = time.perf_counter()
        train, chunk_idx, speaker_idx = self.filter_embeddings(
            embeddings, segmentations=segmentations, num_clean_frames=kwargs.get("num_clean_frames"))
        self.timings["filter"] = time.perf_counter() - t0
        num_clusters, min_clusters, max_clusters = self.set_num_clusters(
            train.shape[0], num_clusters=num_clusters, min_clusters=min_clusters, max_clusters=max_clusters)
        if max_clusters < 2:
            return self._single_cluster(embeddings, train)
        t0 = time.perf_counter()
        labels = self.cluster(train, min_clusters=min_clusters, max_clusters=max_clusters,
                              num_clusters=num_clusters)
        t1 = time.perf_counter()
        out = self.assign_embeddings(embeddings, chunk_idx, speaker_idx, labels,
                                     constrained=self.constrained_assignment,
                                     device_embeddings=kwargs.get("device_embeddings"))
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
        """linkage matrix (:368-382).  Geometric methods on cosine embeddings are run as Euclidean
        linkage of the unit-normalised vectors -- normalised IN PLACE, like the reference, because the
        small-cluster centroids below are means of the normalised copy."""
        geometric = self.metric == "cosine" and self.method in ("centroid", "median", "ward")
        if geometric:
            with np.errstate(divide="ignore", invalid="ignore"):
                embeddings /= np.linalg.norm(embeddings, axis=-1, keepdims=True)
        t0 = time.perf_counter()
        on_gpu = self.device is not None and getattr(self.device, "type", None) == "cuda"
        if geometric and self.method == "centroid" and on_gpu and len(embeddings) >= 2:
            if not np.isfinite(embeddings).all():
                # scipy.cluster.hierarchy.linkage validates its input the same way; a zero-norm / inf
                # embedding (NaN after the normalisation) must not reach the merge kernel
                raise ValueError("The condensed distance matrix must contain only finite values.")
            Z = distance.linkage_centroid(embeddings, self.device)
            self.timings.update(pdist=0.0, linkage=time.perf_counter() - t0, num_embeddings=len(embeddings))
            return Z
        if geometric or self.metric == "euclidean":
            condensed = distance.pdist_euclidean(embeddings, device=self.device)
            t1 = time.perf_counter()
            Z = linkage(condensed, method=self.method)
            self.timings.update(pdist=t1 - t0, linkage=time.perf_counter() - t1,
                                num_embeddings=len(embeddings))
            return Z
        return linkage(embeddings, method=self.method, metric=self.metric)

    def _cut_for_target(self, tree: Dendrogram, target: int, min_size: int) -> Tuple[np.ndarray, bool]:
        """flat clusters with `target` large clusters, examining cuts by increasing |height - threshold|
        (clustering.py:405-451).  Returns (labels, exact)."""
        Z = tree.Z
        order = np.argsort(np.abs(Z[:, 2] - self.threshold))      # same call as the reference: ties
        order = order[Z[order, 3] >= min_size]                    # merges that create a large cluster
        large = tree.large_cluster_counts(min_size)[order]
        hits = np.nonzero(large == target)[0]
        if len(hits):
            return tree.cut_after_merge(int(order[hits[0]])), True
        # no cut gives the target: the first cut that comes closest, if it beats "everything in one"
        gap = np.abs(large - target)
        if len(gap) and gap.min() < abs(1 - target):
            return tree.cut_after_merge(int(order[int(np.argmin(gap))])), False
        return tree.cut_after_merge(tree.num_leaves - 1), False

    def cluster(self, embeddings: np.ndarray, min_clusters: Optional[int] = None,
                max_clusters: Optional[int] = None, num_clusters: Optional[int] = None):
        n = embeddings.shape[0]
        if n == 1:
            return np.zeros((1,), dtype=np.uint8)
        min_size = min(self.min_cluster_size, max(1, round(0.1 * n)))
        tree = Dendrogram(self.dendrogram(embeddings))
        labels = tree.cut(self.threshold)
        sizes = np.bincount(labels)
        num_large = int((sizes >= min_size).sum())
        # a bound that the threshold cut violates becomes the target (:394-403)
        if num_large < min_clusters:
            num_clusters = min_clusters
        elif num_large > max_clusters:
            num_clusters = max_clusters
        if num_clusters is not None and num_large != num_clusters:
            labels, exact = self._cut_for_target(tree, num_clusters, min_size)
            sizes = np.bincount(labels)
            num_large = int((sizes >= min_size).sum())
            if not exact:
                warnings.warn(f"hierarchical clustering yields {num_large} clusters of at least "
                              f"{min_size} embeddings where {num_clusters} were requested; a smaller "
                              "`min_cluster_size` may help")
        if num_large == 0:
            return np.zeros_like(labels)
        is_large = sizes >= min_size
        if is_large[np.unique(labels)].all():
            return labels
        # every small cluster joins the large cluster with the nearest centroid (:457-476); centroids
        # are means of the (normalised) training copy
        ids = np.nonzero(sizes > 0)[0]
        means = segment_means(embeddings, labels, len(sizes))
        large_ids, small_ids = ids[is_large[ids]], ids[~is_large[ids]]
        nearest = np.argmin(scipy_cdist(means[large_ids], means[small_ids], metric=self.metric), axis=0)
        relabel = np.arange(len(sizes))
        relabel[small_ids] = large_ids[nearest]
        return np.unique(relabel[labels], return_inverse=True)[1]


class KMeansClustering(BaseClustering):
    """clustering.py:483-547.  scikit-learn's KMeans on the host -- the reference's own call; not part
    of the accelerated 3.1 / community-1 paths (it needs an explicit number of speakers)."""

    expects_num_clusters: bool = True

    def __init__(self, metric: str = "cosine"):
        if metric not in ("cosine", "euclidean"):
            raise ValueError(f"Unsupported metric: {metric}. Must be 'cosine' or 'euclidean'.")
        super().__init__(metric=metric)

    def cluster(self, embeddings: np.ndarray, min_clusters: Optional[int] = None,
                max_clusters: Optional[int] = None, num_clusters: Optional[int] = None):
        if num_clusters is None:
            raise ValueError("`num_clusters` must be provided.")
        if embeddings.shape[0] < num_clusters:
            return np.arange(embeddings.shape[0], dtype=np.int32)
        if self.metric == "cosine":
            with np.errstate(divide="ignore", invalid="ignore"):
                embeddings /= np.linalg.norm(embeddings, axis=-1, keepdims=True)
        from sklearn.cluster import KMeans
        return KMeans(n_clusters=num_clusters, n_init=3, random_state=42, copy_x=False).fit_predict(embeddings)


class VBxClustering(BaseClustering):
    """clustering.py:550-669: AHC initialisation (GPU linkage) -> PLDA projection (GPU) -> VBx (GPU,
    csrc/vbx.hip) -> centroids of the surviving speakers -> constrained assignment.
    Hyper-parameters: threshold (AHC cut), Fa, Fb."""

    expects_num_clusters: bool = False
    max_iterations = 20          # cluster_vbx(maxIters=20), utils/vbx.py:143
    init_smoothing = 7.0         # softmax temperature of the one-hot AHC labels, utils/vbx.py:143-147
    elbo_epsilon = 1e-4          # VBx(epsilon=1e-4), utils/vbx.py:36
    prune_below = 1e-7           # speakers whose prior collapsed (clustering.py:621)

    def __init__(self, plda=None, metric: str = "cosine", constrained_assignment: bool = True):
        super().__init__(metric=metric, constrained_assignment=constrained_assignment)
        from .plda import get_plda
        self.plda = get_plda(plda)
        self.threshold = Uniform(0.5, 0.8)
        self.Fa = Uniform(0.01, 0.5)
        self.Fb = Uniform(0.01, 15.0)

    def to(self, device):
        super().to(device)
        if self.plda is not None:
            self.plda.to(device)
        return self

    def _ahc_labels(self, train: np.ndarray) -> np.ndarray:
        unit = train / np.linalg.norm(train, axis=1, keepdims=True)
        if not np.isfinite(unit).all():
            raise ValueError("The condensed distance matrix must contain only finite values.")
        Z = distance.linkage_centroid(unit, self.device)
        labels = fcluster(Z, self.threshold, criterion="distance") - 1
        return np.unique(labels, return_inverse=True)[1], unit

    def _vbx(self, fea, labels: np.ndarray):
        """responsibilities (N, S) and speaker priors (S,) after VB inference on the device"""
        import torch
        from scipy.special import softmax
        from . import ffi
        lib = ffi.load()
        device = self.device
        n, d = fea.shape
        s = int(labels.max()) + 1
        onehot = np.zeros((n, s))
        onehot[np.arange(n), labels] = 1.0
        q0 = onehot if self.init_smoothing < 0 else softmax(onehot * self.init_smoothing, axis=1)
        gamma = torch.from_numpy(q0).to(device)
        phi = torch.from_numpy(np.ascontiguousarray(self.plda.phi, dtype=np.float64)).to(device)
        elbo = torch.zeros(self.max_iterations, dtype=torch.float64, device=device)
        ws = torch.empty(lib.pa_vbx_workspace_bytes(n, s, d), dtype=torch.uint8, device=device)
        history = []
        with torch.cuda.device(device):
            for it in range(self.max_iterations):
                ffi.check(lib.pa_vbx_iteration(ffi.ptr(fea), ffi.ptr(phi), n, s, d, float(self.Fa),
                                               float(self.Fb), int(it == 0), ffi.ptr(gamma),
                                               ffi.ptr(elbo[it:]), ffi.ptr(ws), ws.numel(), ffi.stream()),
                          "pa_vbx_iteration")
                history.append(float(elbo[it].item()))     # the convergence test of utils/vbx.py:129-133
                if it > 0 and history[-1] - history[-2] < self.elbo_epsilon:
                    break
        q = gamma.cpu().numpy()
        pi = q.sum(axis=0)
        self.timings["vbx_iterations"] = len(history)
        return q, pi / pi.sum()

    def __call__(self, embeddings: np.ndarray, segmentations: Optional[SlidingWindowFeature] = None,
                 num_clusters: Optional[int] = None, min_clusters: Optional[int] = None,
                 max_clusters: Optional[int] = None, **kwargs):
        if self.plda is None:
            raise ValueError("VBxClustering needs a PLDA model (`plda=` of SpeakerDiarization)")
        if self.device is None or getattr(self.device, "type", None) != "cuda":
            raise RuntimeError("VBxClustering runs on the GPU: pipeline.to(torch.device('cuda')) first")
        self.timings.clear()
        min_clusters = 1 if min_clusters is None else min_clusters
        max_clusters = np.inf if max_clusters is None else max_clusters
        train, _, _ = self.filter_embeddings(embeddings, segmentations=segmentations,
                                             num_clean_frames=kwargs.get("num_clean_frames"))
        if train.shape[0] < 2:
            return self._single_cluster(embeddings, train)
        t0 = time.perf_counter()
        labels, unit = self._ahc_labels(train)
        t1 = time.perf_counter()
        fea = self.plda.transform_device(train, self.device)
        q, priors = self._vbx(fea, labels)
        t2 = time.perf_counter()
        W = q[:, priors > self.prune_below]
        centroids = W.T @ train / W.sum(axis=0, keepdims=True).T
        constrained = self.constrained_assignment
        found = centroids.shape[0]
        # a violated bound (or an explicit number that VBx did not find) falls back to KMeans on the
        # unit vectors, without the assignment constraint (clustering.py:624-645)
        if found < min_clusters:
            num_clusters = min_clusters
        elif found > max_clusters:
            num_clusters = max_clusters
        if num_clusters and num_clusters != found:
            from sklearn.cluster import KMeans
            constrained = False
            km = KMeans(n_clusters=num_clusters, n_init=3, random_state=42, copy_x=False).fit_predict(unit)
            centroids = segment_means(train, km, num_clusters)
        soft = self._similarities(embeddings, centroids)
        if constrained:
            # silent local speakers must never win a cluster in the assignment (:658-660)
            silent = (segmentations.data.sum(axis=1) == 0) if kwargs.get("active_frames") is None \
                else (kwargs["active_frames"] == 0)
            soft[silent] = soft.min() - 1.0
            hard = self.constrained_argmax(soft)
        else:
            hard = np.argmax(soft, axis=2)
        self.timings.update(ahc=t1 - t0, vbx=t2 - t1, assign=time.perf_counter() - t2,
                            num_embeddings=train.shape[0])
        return hard.reshape(embeddings.shape[:2]), soft, centroids


class OracleClustering(BaseClustering):
    """Clusters with the answer sheet (clustering.py:672-756): every chunk's local speakers are mapped onto
    the reference speakers by the cost-minimising permutation between the model's segmentation and the
    reference annotation discretised on the same chunk / frame grid.  Needs `file["annotation"]`; no
    embeddings are involved in the assignment (centroids are computed when they are given)."""

    expects_num_clusters = True

    def __call__(self, embeddings: Optional[np.ndarray] = None,
                 segmentations: Optional[SlidingWindowFeature] = None, file=None, frames=None, **kwargs):
        from .annotation_frames import oracle_segmentation
        from .permutation import permutate
        num_chunks, num_frames, num_speakers = segmentations.data.shape
        oracle = oracle_segmentation(file, segmentations.sliding_window, frames=frames)
        file["oracle_segmentations"] = oracle
        num_clusters = oracle.data.shape[2]
        common = min(num_frames, oracle.data.shape[1])
        seg, ref = segmentations.data[:, :common], oracle.data[:, :common]
        hard = np.full((num_chunks, num_speakers), -2, dtype=np.int8)
        soft = np.zeros((num_chunks, num_speakers, num_clusters))
        for c in range(min(num_chunks, ref.shape[0])):     # the zero-padded last chunk has no reference chunk
            _, (permutation,) = permutate(ref[c][np.newaxis], seg[c])
            for cluster, speaker in enumerate(permutation):
                if speaker is not None:
                    hard[c, speaker] = cluster
                    soft[c, speaker, cluster] = 1.0
        if embeddings is None:
            return hard, soft, None
        train, chunk_idx, speaker_idx = self.filter_embeddings(
            embeddings, segmentations=SlidingWindowFeature(seg, segmentations.sliding_window))
        train_clusters = hard[chunk_idx, speaker_idx]
        centroids = np.vstack([np.mean(train[train_clusters == k], axis=0) for k in range(num_clusters)])
        return hard, soft, centroids


class Clustering(Enum):
    """clustering.py:759-763"""
    AgglomerativeClustering = AgglomerativeClustering
    KMeansClustering = KMeansClustering
    VBxClustering = VBxClustering
    OracleClustering = OracleClustering
