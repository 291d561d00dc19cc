"""Oracle (CPU) restatement of VBx clustering + PLDA.

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.  Follows, relative to
/root/reference/src/pyannote/audio:
  utils/vbx.py:27-140   VBx (Landini et al., "Bayesian HMM clustering of x-vector sequences (VBx)...";
                        equation numbers in comments refer to that paper, as in the reference)
  utils/vbx.py:143-157  cluster_vbx (AHC one-hot initialisation, softmax smoothing 7.0)
  utils/vbx.py:160-218  l2_norm, vbx_setup (x-vector transform + PLDA whitening via scipy eigh)
  core/plda.py:33-60    PLDA.__call__ / phi
  pipelines/clustering.py:550-669  VBxClustering.__call__
numpy / scipy / scikit-learn calls are the very calls the reference makes; nothing is pinned by a
reference test (tests/ has none for VBx) -> parity of this file is by construction only."""
from __future__ import annotations

import numpy as np
from scipy.cluster.hierarchy import fcluster, linkage
from scipy.linalg import eigh
from scipy.optimize import linear_sum_assignment
from scipy.spatial.distance import cdist
from scipy.special import logsumexp, softmax

from .pipeline import filter_embeddings


def vbx(X, Phi, Fa=1.0, Fb=1.0, pi=10, gamma=None, maxIters=10, epsilon=1e-4):
    """utils/vbx.py:27-140 (return_model / alpha / invL / ref / plot arguments unused by the pipeline)"""
    D = X.shape[1]
    if type(pi) is int:
        pi = np.ones(pi) / pi
    assert gamma.shape[1] == len(pi) and gamma.shape[0] == X.shape[0]
    G = -0.5 * (np.sum(X ** 2, axis=1, keepdims=True) + D * np.log(2 * np.pi))   # (23) constant term
    V = np.sqrt(Phi)
    rho = X * V                                                                   # (18)
    Li = []
    for ii in range(maxIters):
        invL = 1.0 / (1 + Fa / Fb * gamma.sum(axis=0, keepdims=True).T * Phi)     # (17)
        alpha = Fa / Fb * invL * gamma.T.dot(rho)                                 # (16)
        log_p_ = Fa * (rho.dot(alpha.T) - 0.5 * (invL + alpha ** 2).dot(Phi) + G)  # (23)
        eps = 1e-8
        lpi = np.log(pi + eps)
        log_p_x = logsumexp(log_p_ + lpi, axis=-1)
        log_pX_ = np.sum(log_p_x, axis=0)
        gamma = np.exp(log_p_ + lpi - log_p_x[:, None])
        pi = np.sum(gamma, axis=0)
        pi = pi / pi.sum()
        ELBO = log_pX_ + Fb * 0.5 * np.sum(np.log(invL) - invL - alpha ** 2 + 1)   # (25)
        Li.append([ELBO])
        if ii > 0 and ELBO - Li[-2][0] < epsilon:
            break
    return gamma, pi, Li


def cluster_vbx(ahc_init, fea, Phi, Fa, Fb, maxIters=20, init_smoothing=7.0):
    """utils/vbx.py:143-157"""
    qinit = np.zeros((len(ahc_init), ahc_init.max() + 1))
    qinit[range(len(ahc_init)), ahc_init.astype(int)] = 1.0
    qinit = qinit if init_smoothing < 0 else softmax(qinit * init_smoothing, axis=1)
    gamma, pi, Li = vbx(fea, Phi, Fa=Fa, Fb=Fb, pi=qinit.shape[1], gamma=qinit, maxIters=maxIters)
    return gamma, pi, Li


def l2_norm(m):
    return m / np.linalg.norm(m, axis=1, ord=2)[:, np.newaxis]


class PLDA:
    """core/plda.py:33-60 over utils/vbx.py:181-218"""

    def __init__(self, transform_npz, plda_npz, lda_dimension: int = 128):
        x = np.load(transform_npz)
        self.mean1, self.mean2, self.lda = x["mean1"], x["mean2"], x["lda"]
        p = np.load(plda_npz)
        self.plda_mu, plda_tr, plda_psi = p["mu"], p["tr"], p["psi"]
        W = np.linalg.inv(plda_tr.T.dot(plda_tr))
        B = np.linalg.inv((plda_tr.T / plda_psi).dot(plda_tr))
        acvar, wccn = eigh(B, W)
        self.plda_psi = acvar[::-1]
        self.plda_tr = wccn.T[::-1]
        self.lda_dimension = lda_dimension

    @property
    def phi(self):
        return self.plda_psi[: self.lda_dimension]

    def __call__(self, embeddings):
        lda = self.lda
        x0 = np.sqrt(lda.shape[1]) * l2_norm(
            lda.T.dot(np.sqrt(lda.shape[0]) * l2_norm(embeddings - self.mean1).T).T - self.mean2)
        return (x0 - self.plda_mu).dot(self.plda_tr.T)[:, : self.lda_dimension]


def constrained_argmax(soft_clusters):
    """pipelines/clustering.py:127-140"""
    soft_clusters = np.nan_to_num(soft_clusters, nan=np.nanmin(soft_clusters))
    num_chunks, num_speakers, _ = soft_clusters.shape
    hard = -2 * np.ones((num_chunks, num_speakers), dtype=np.int8)
    for c, cost in enumerate(soft_clusters):
        speakers, clusters = linear_sum_assignment(cost, maximize=True)
        for s, k in zip(speakers, clusters):
            hard[c, s] = k
    return hard


def vbx_clustering(embeddings, segmentations, plda: PLDA, threshold=0.6, Fa=0.07, Fb=0.8,
                   num_clusters=None, min_clusters=None, max_clusters=None,
                   constrained_assignment=True, metric="cosine"):
    """VBxClustering.__call__ (pipelines/clustering.py:576-669); min/max_clusters as set by
    SpeakerDiarization.apply (set_num_speakers: 1 / inf when not given)."""
    min_clusters = 1 if min_clusters is None else min_clusters
    max_clusters = np.inf if max_clusters is None else max_clusters
    train_embeddings, _, _ = filter_embeddings(embeddings, segmentations)
    if train_embeddings.shape[0] < 2:
        num_chunks, num_speakers, _ = embeddings.shape
        return (np.zeros((num_chunks, num_speakers), dtype=np.int8),
                np.ones((num_chunks, num_speakers, 1)), np.mean(train_embeddings, axis=0, keepdims=True))
    normed = train_embeddings / np.linalg.norm(train_embeddings, axis=1, keepdims=True)
    dendrogram = linkage(normed, method="centroid", metric="euclidean")
    ahc_clusters = fcluster(dendrogram, threshold, criterion="distance") - 1
    _, ahc_clusters = np.unique(ahc_clusters, return_inverse=True)
    fea = plda(train_embeddings)
    q, sp, _ = cluster_vbx(ahc_clusters, fea, plda.phi, Fa=Fa, Fb=Fb, maxIters=20)
    num_chunks, num_speakers, dimension = embeddings.shape
    W = q[:, sp > 1e-7]
    centroids = W.T @ train_embeddings.reshape(-1, dimension) / W.sum(0, keepdims=True).T
    auto_num_clusters, _ = centroids.shape
    if auto_num_clusters < min_clusters:
        num_clusters = min_clusters
    elif auto_num_clusters > max_clusters:
        num_clusters = max_clusters
    if num_clusters and num_clusters != auto_num_clusters:
        from sklearn.cluster import KMeans
        constrained_assignment = False
        km = KMeans(n_clusters=num_clusters, n_init=3, random_state=42, copy_x=False).fit_predict(normed)
        centroids = np.vstack([np.mean(train_embeddings[km == k], axis=0) for k in range(num_clusters)])
    e2k = cdist(embeddings.reshape(-1, dimension), centroids, metric=metric).reshape(
        num_chunks, num_speakers, -1)
    soft = 2 - e2k
    if constrained_assignment:
        const = soft.min() - 1.0
        soft[segmentations.sum(1) == 0] = const
        hard = constrained_argmax(soft)
    else:
        hard = np.argmax(soft, axis=2)
    return hard.reshape(num_chunks, num_speakers), soft, centroids


def synth_plda(directory, dim: int = 256, lda_dim: int = 128, seed: int = 7):
    """Synthetic `xvec_transform.npz` + `plda.npz` in the layout vbx_setup reads (no pretrained PLDA
    exists offline): random LDA projection, positive between-class spectrum."""
    import os
    rng = np.random.default_rng(seed)
    os.makedirs(directory, exist_ok=True)
    lda = rng.standard_normal((dim, lda_dim)) / np.sqrt(dim)
    np.savez(os.path.join(directory, "xvec_transform.npz"), mean1=0.05 * rng.standard_normal(dim),
             mean2=0.05 * rng.standard_normal(lda_dim), lda=lda)
    A = rng.standard_normal((lda_dim, lda_dim)) / np.sqrt(lda_dim) + np.eye(lda_dim)
    psi = np.sort(rng.uniform(0.05, 6.0, lda_dim))[::-1].copy()
    np.savez(os.path.join(directory, "plda.npz"), mu=0.1 * rng.standard_normal(lda_dim), tr=A, psi=psi)
    return directory
