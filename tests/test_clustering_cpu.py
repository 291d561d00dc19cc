"""Host side of the rewritten clustering module (pyannote_audio_amd/clustering.py) against the oracle's
loop-for-loop restatement of pipelines/clustering.py (oracle/pipeline.py) and against the primitives
it replaces: bit-exact labels / centroids / similarities for every bound configuration, including the
forced-number search (clustering.py:405-451) that the product answers from ONE scan of the merge sizes."""
import warnings

import numpy as np
import torch
import pytest
from scipy.cluster.hierarchy import fcluster, linkage

import pyannote_audio_amd as pa
from pyannote_audio_amd.clustering import Dendrogram, clamp_cluster_bounds, segment_means
from pyannote_audio_amd.core import SlidingWindow, SlidingWindowFeature
from oracle import pipeline as O

CHUNKS = SlidingWindow(start=0.0, duration=10.0, step=1.0)


def test_segment_means_equals_numpy_mean_bitwise():
    rng = np.random.default_rng(0)
    for n, d, k in ((1, 3, 1), (50, 7, 4), (4000, 256, 9), (300, 256, 40)):
        X = (rng.standard_normal((n, d)) * rng.uniform(0.1, 30)).astype(np.float32)
        labels = rng.integers(0, k, n)
        labels[:k] = np.arange(k) % k                       # every segment non-empty
        want = np.vstack([np.mean(X[labels == j], axis=0) for j in range(k)])
        got = segment_means(X, labels, k)
        assert got.dtype == np.float32 and np.array_equal(got, want)
    # float64 input stays float64; an empty segment yields NaN like np.mean of an empty slice
    X = rng.standard_normal((10, 4))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        got = segment_means(X, np.array([0, 0, 2, 2, 2, 0, 2, 0, 0, 2]), 3)
    assert got.dtype == np.float64 and np.isnan(got[1]).all()
    assert np.array_equal(got[0], np.mean(X[[0, 1, 5, 7, 8]], axis=0))


def test_large_cluster_counts_equal_fcluster_per_merge():
    rng = np.random.default_rng(1)
    for n, m in ((30, 1), (200, 5), (777, 12)):
        X = rng.standard_normal((n, 8))
        X[rng.integers(0, n, 5)] = X[rng.integers(0, n, 5)]     # exact ties
        Z = linkage(X, method="centroid")
        tree = Dendrogram(Z)
        counts = tree.large_cluster_counts(m)
        ranked = Z.copy()
        ranked[:, 2] = np.arange(n - 1)
        for i in list(range(0, n - 1, max(1, n // 40))) + [n - 2]:
            labels = fcluster(ranked, i, criterion="distance")
            assert counts[i] == int((np.bincount(labels)[1:] >= m).sum()), (n, m, i)
            assert np.array_equal(tree.cut_after_merge(i), labels - 1)


def test_clamp_cluster_bounds():
    assert clamp_cluster_bounds(10, None, None, None) == (None, 1, 10)
    assert clamp_cluster_bounds(10, 3, None, None) == (3, 3, 3)
    assert clamp_cluster_bounds(2, 5, None, None) == (2, 2, 2)
    assert clamp_cluster_bounds(10, None, 4, 4) == (4, 4, 4)
    assert clamp_cluster_bounds(10, None, 2, np.inf) == (None, 2, 10)
    with pytest.raises(ValueError):
        clamp_cluster_bounds(10, None, 5, 2)


def _data(C, K, noise, seed, D_=24, S=3, F=40):
    rng = np.random.default_rng(seed)
    centers = rng.standard_normal((K, D_))
    who = rng.integers(0, K, size=(C, S))
    emb = (centers[who] + noise * rng.standard_normal((C, S, D_))).astype(np.float32)
    emb[rng.integers(0, C), rng.integers(0, S)] = np.nan
    seg = (rng.uniform(size=(C, F, S)) < 0.45).astype(np.float32)
    seg[rng.integers(0, C, C // 6), :, rng.integers(0, S, C // 6)] = 0.0
    return emb, seg


@pytest.mark.parametrize("C,K,noise,seed", [(40, 3, 0.1, 0), (120, 5, 0.25, 1), (300, 8, 0.35, 2),
                                            (90, 2, 0.6, 3), (12, 2, 0.05, 4)])
@pytest.mark.parametrize("kw", [dict(), dict(num_clusters=1), dict(num_clusters=2), dict(num_clusters=7),
                                dict(min_clusters=9), dict(max_clusters=2), dict(min_clusters=3, max_clusters=4),
                                dict(num_clusters=50)])
@pytest.mark.parametrize("min_size,threshold", [(12, 0.7045654963945799), (3, 0.4), (1, 1.1)])
def test_agglomerative_matches_oracle(C, K, noise, seed, kw, min_size, threshold):
    emb, seg = _data(C, K, noise, seed)
    clu = pa.AgglomerativeClustering(metric="cosine").to(torch.device("cpu")).instantiate(
        {"method": "centroid", "min_cluster_size": min_size, "threshold": threshold})
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        hard, soft, cen = clu(embeddings=emb.copy(), segmentations=SlidingWindowFeature(seg, CHUNKS), **kw)
    rh, rs, rc = O.clustering(emb.copy(), seg, method="centroid", threshold=threshold,
                              min_cluster_size=min_size, **kw)
    assert np.array_equal(hard, rh)
    assert np.array_equal(cen, rc, equal_nan=True)
    assert np.array_equal(soft, rs, equal_nan=True)


def test_other_linkage_methods_and_metrics_match_oracle():
    emb, seg = _data(80, 4, 0.2, 9)
    for method in ("average", "ward", "complete"):
        clu = pa.AgglomerativeClustering(metric="cosine").to(torch.device("cpu")).instantiate(
            {"method": method, "min_cluster_size": 5, "threshold": 0.8})
        hard, _, cen = clu(embeddings=emb.copy(), segmentations=SlidingWindowFeature(seg, CHUNKS))
        rh, _, rc = O.clustering(emb.copy(), seg, method=method, threshold=0.8, min_cluster_size=5)
        assert np.array_equal(hard, rh) and np.array_equal(cen, rc, equal_nan=True), method


def test_kmeans_and_enum_members():
    assert set(pa.Clustering.__members__) == {"AgglomerativeClustering", "KMeansClustering",
                                              "VBxClustering", "OracleClustering"}
    emb, seg = _data(60, 3, 0.1, 5)
    km = pa.KMeansClustering().to(torch.device("cpu")).instantiate({})
    hard, soft, cen = km(embeddings=emb.copy(), segmentations=SlidingWindowFeature(seg, CHUNKS), num_clusters=3)
    assert hard.shape == (60, 3) and cen.shape == (3, 24) and set(np.unique(hard)) <= {0, 1, 2}
    with pytest.raises(ValueError):
        km.cluster(emb[:, 0].copy())
    assert pa.Clustering["OracleClustering"].value is pa.OracleClustering
    with pytest.raises(ValueError, match="PLDA"):
        pa.VBxClustering().instantiate({"threshold": 0.6, "Fa": 0.07, "Fb": 0.8})(
            embeddings=emb, segmentations=SlidingWindowFeature(seg, CHUNKS))


def test_unplaced_clustering_object_is_the_reference_stand_alone_use():
    """the reference's public API allows `AgglomerativeClustering().instantiate(...)(embeddings, ...)` without any
    placement (pipelines/clustering.py:214-289): an unplaced object computes on the host through SciPy, exactly like
    one that was explicitly put there (the PIPELINE itself refuses to run anywhere but on a GPU:
    tests/test_pipeline_cpu.py)."""
    rng = np.random.default_rng(0)
    centers = rng.standard_normal((2, 16))
    emb = (centers[rng.integers(0, 2, (30, 3))] + 0.1 * rng.standard_normal((30, 3, 16))).astype(np.float32)
    seg = SlidingWindowFeature(np.ones((30, 589, 3), dtype=np.float32), SlidingWindow(start=0.0, duration=10.0, step=1.0))
    seg.data[:, :, 1:] = 0
    params = {"method": "centroid", "min_cluster_size": 2, "threshold": 0.7}
    unplaced = pa.AgglomerativeClustering(metric="cosine").instantiate(params)
    assert unplaced.device is None
    on_host = pa.AgglomerativeClustering(metric="cosine").instantiate(params).to(torch.device("cpu"))
    a = unplaced(embeddings=emb.copy(), segmentations=seg, min_clusters=1, max_clusters=np.inf)
    b = on_host(embeddings=emb.copy(), segmentations=seg, min_clusters=1, max_clusters=np.inf)
    for u, v in zip(a, b):
        assert np.array_equal(u, v, equal_nan=True)


def test_merge_placement_hint_is_thread_local_and_nests():
    """`distance.device_to_ourselves()` (one file on its own / last file of a batch -> the dendrogram merge may use
    several workgroups) only marks the calling thread and restores the previous state on exit."""
    import threading
    from pyannote_audio_amd import distance

    def alone():
        return getattr(distance._hint, "alone", False)

    assert not alone()
    seen = {}
    with distance.device_to_ourselves():
        assert alone()
        t = threading.Thread(target=lambda: seen.setdefault("other", alone()))
        t.start()
        t.join()
        with distance.device_to_ourselves():
            assert alone()
        assert alone()
        with pytest.raises(RuntimeError):
            with distance.device_to_ourselves():
                raise RuntimeError("boom")
        assert alone()
    assert not alone() and seen == {"other": False}
