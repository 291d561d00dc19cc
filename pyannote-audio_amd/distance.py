"""float64 distance kernels for clustering: GPU (`pa_pdist_f64`, `pa_cdist_cosine_f64`) with results
bit-identical to SciPy's `pdist(X, "euclidean")` / `cdist(A, B, "cosine")`.

SciPy computes Euclidean distances as a sequential-k sum of (u_k - v_k)^2 in double without FMA
contraction, then sqrt; cosine as 1 - <u,v> / (|u| |v|) with sequential dot products.  The kernels
reproduce that summation order with explicit `__dmul_rn/__dadd_rn` (see csrc/cluster.hip).  Small
problems stay in SciPy: the launch + copy overhead exceeds the work."""
from __future__ import annotations

import contextlib
import os
import threading
import time

import numpy as np
import torch
from scipy.spatial.distance import cdist as _scipy_cdist
from scipy.spatial.distance import pdist as _scipy_pdist

from . import ffi

def _use_gpu(device, n: int) -> bool:
    """GPU kernels whenever the clustering object lives on a GPU (`pipeline.to(cuda)`); a missing library on a GPU
    device raises -- no silent downgrade there.  A clustering object that was never placed (device None) or that
    its user put on the host is the reference's stand-alone use (`AgglomerativeClustering().instantiate(...)(
    embeddings, ...)`, pipelines/clustering.py:214-289) and calls SciPy, i.e. the reference's own implementation
    of these two functions; the PIPELINE never gets there: SpeakerDiarization._require_device refuses to run
    anywhere but on a GPU."""
    kind = getattr(device, "type", None)
    if kind == "cuda":
        ffi.require_gpu()
        return n >= 2
    return False


@ffi.on_device(lambda X, device=None: device)
def pdist_euclidean(X: np.ndarray, device=None) -> np.ndarray:
    """condensed float64 Euclidean distance matrix of the rows of X (any float dtype)."""
    n = X.shape[0]
    if not _use_gpu(device, n):
        return _scipy_pdist(X, metric="euclidean")
    lib = ffi.load()
    Xd = torch.from_numpy(np.ascontiguousarray(X, dtype=np.float64)).to(device)
    out = torch.empty(n * (n - 1) // 2, dtype=torch.float64, device=device)
    ffi.check(lib.pa_pdist_f64(ffi.ptr(Xd), n, X.shape[1], ffi.ptr(out), ffi.stream()), "pa_pdist_f64")
    return out.cpu().numpy()


@ffi.on_device(lambda X, rows, labels, num_segments, device: device)
def centroid_means(X: torch.Tensor, rows: np.ndarray, labels: np.ndarray, num_segments: int, device) -> np.ndarray:
    """means of the rows X[rows] per label 0..num_segments-1 on the GPU (`pa_centroid_means`), bit-identical to
    `np.vstack([np.mean(X[rows][labels == k], axis=0) for k in range(num_segments)])` (pipelines/clustering.py:
    182-187): ONE stable sort groups the row numbers (original order kept inside a cluster), the kernel adds the
    rows of a cluster top to bottom in float32 and divides in float32.  `X`: (., D) float32 tensor on `device`
    (the embeddings never leave HBM for this); an empty cluster yields NaN."""
    ffi.require_gpu()
    lib = ffi.load()
    if X.dtype != torch.float32 or not X.is_contiguous():
        raise ValueError("centroid_means: X must be a contiguous float32 tensor")
    order = np.argsort(labels, kind="stable")
    counts = np.bincount(labels, minlength=num_segments)[:num_segments]
    offsets = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    grouped = np.ascontiguousarray(np.asarray(rows)[order], dtype=np.int32)
    D = X.shape[1]
    rows_d = torch.from_numpy(grouped).to(device)
    offs_d = torch.from_numpy(offsets).to(device)
    out = torch.empty((num_segments, D), dtype=torch.float32, device=device)
    ffi.check(lib.pa_centroid_means(ffi.ptr(X), D, ffi.ptr(rows_d), ffi.ptr(offs_d), num_segments, ffi.ptr(out),
                                    ffi.stream()), "pa_centroid_means")
    return out.cpu().numpy()


_hint = threading.local()


@contextlib.contextmanager
def device_to_ourselves():
    """Inside this block the calling thread promises that nothing else keeps the GPU busy (one file applied on its
    own, the last file of a batch): `linkage_centroid` may then spread the merge over several workgroups."""
    previous = getattr(_hint, "alone", False)
    _hint.alone = True
    try:
        yield
    finally:
        _hint.alone = previous


@ffi.on_device(lambda X, device: device)
def linkage_centroid(X: np.ndarray, device) -> np.ndarray:
    """scipy.cluster.hierarchy.linkage(X, method="centroid", metric="euclidean") entirely on the GPU:
    float64 pdist (`pa_pdist_f64`) feeds the persistent merge kernels (`pa_linkage_centroid_f64`: the heap-free
    merge of csrc/linkage_fast.hip, then -- only if two rows ever tied for the smallest lower bound -- the exact
    heap replay of csrc/linkage.hip) without leaving HBM; only the (n-1, 4) dendrogram comes back.  Bit-identical
    to SciPy."""
    n = X.shape[0]
    ffi.require_gpu()
    lib = ffi.load()
    global last_linkage_stats, last_linkage_phases
    timed = os.environ.get("PA_LINKAGE_TIMING") == "1"   # development: where the wall time of a call goes
    marks = [("start", time.perf_counter())]

    host_only = os.environ.get("PA_LINKAGE_EVENTS") == "1"   # host clock at every step WITHOUT synchronising

    def mark(name):
        if timed:
            torch.cuda.synchronize(device)
        if timed or host_only:
            marks.append((name, time.perf_counter()))

    Xd = torch.from_numpy(np.ascontiguousarray(X, dtype=np.float64)).to(device)
    cond = torch.empty(n * (n - 1) // 2, dtype=torch.float64, device=device)
    mark("upload + allocate condensed")
    ffi.check(lib.pa_pdist_f64(ffi.ptr(Xd), n, X.shape[1], ffi.ptr(cond), ffi.stream()), "pa_pdist_f64")
    mark("pdist")
    Z = torch.empty((n - 1, 4), dtype=torch.float64, device=device)
    ws = torch.empty(lib.pa_linkage_workspace_bytes(n), dtype=torch.uint8, device=device)
    mark("allocate workspace")
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)] if os.environ.get("PA_LINKAGE_EVENTS") == "1" else None
    if ev:
        ev[0].record()
    ffi.check(lib.pa_linkage_centroid_f64_ex(ffi.ptr(cond), n, ffi.ptr(Z), ffi.ptr(ws), ws.numel(),
                                             1 if getattr(_hint, "alone", False) else 0, ffi.stream()),
              "pa_linkage_centroid_f64")
    if ev:
        ev[1].record()
    mark("merge kernels")
    if ev:   # development: device time of the merge kernels without changing what the host does around them
        ev[1].synchronize()
        last_linkage_phases = [("host: " + b[0], b[1] - a[1]) for a, b in zip(marks, marks[1:])] + [
            ("whole call until the events are done", time.perf_counter() - marks[0][1]),
            ("merge kernels (events)", ev[0].elapsed_time(ev[1]) / 1e3)]
    if timed:
        last_linkage_phases = [(b[0], b[1] - a[1]) for a, b in zip(marks, marks[1:])]
    # development counters: [0:8] heap kernel (csrc/linkage.hip; all zero when the heap-free merge completed the
    # dendrogram), [8:16] heap-free merge (csrc/linkage_fast.hip): status (0 = complete, 1 = tie -> heap, 2 =
    # degenerate input -> heap, 3 = a workgroup
    # never showed up -> heap), lower-bound repairs, cycles in pop / pass, n, workgroups, re-publishes, merges done
    last_linkage_stats = ws[-128:].view(torch.int64).cpu().numpy()
    return Z.cpu().numpy()


last_linkage_stats = None
last_linkage_phases = None


@ffi.on_device(lambda A, B, metric="cosine", device=None: device)
def cdist(A, B: np.ndarray, metric: str = "cosine", device=None) -> np.ndarray:
    """`A`: host array, or a tensor that already lives on `device` (converted to float64 there)."""
    on_device = isinstance(A, torch.Tensor)
    if metric != "cosine" or not _use_gpu(device, A.shape[0]):
        return _scipy_cdist(A.cpu().numpy() if on_device else A, B, metric=metric)
    lib = ffi.load()
    if on_device:
        Ad = A.to(device=device, dtype=torch.float64).contiguous()
    else:
        Ad = torch.from_numpy(np.ascontiguousarray(A, dtype=np.float64)).to(device)
    Bd = torch.from_numpy(np.ascontiguousarray(B, dtype=np.float64)).to(device)
    out = torch.empty((A.shape[0], B.shape[0]), dtype=torch.float64, device=device)
    norms = torch.empty(A.shape[0] + B.shape[0], dtype=torch.float64, device=device)
    ffi.check(lib.pa_cdist_cosine_f64(ffi.ptr(Ad), A.shape[0], ffi.ptr(Bd), B.shape[0], A.shape[1],
                                      ffi.ptr(out), ffi.ptr(norms), ffi.stream()), "pa_cdist_cosine_f64")
    return out.cpu().numpy()
