"""Frame-domain stages on the GPU (csrc/frames.hip): speaker counting, overlap-add reconstruction,
top-k discretisation, per-(chunk, speaker) statistics and embedding-mask selection.

Mirrors pipelines/utils/diarization.py:150-268, pipelines/speaker_diarization.py:375-427, 480-528 and
the overlap-add of core/inference.py:498-620 for the case the diarization pipeline uses (hamming=False,
warm_up=(0, 0), hard {0,1} scores).  Results are bit-identical to the reference's float32 arithmetic
(small-integer sums; see the kernel file).  There is no host implementation behind these calls."""
from __future__ import annotations

from typing import Tuple

import numpy as np
import torch

from . import ffi
from .core import SlidingWindow, SlidingWindowFeature


def frame_geometry(chunks: SlidingWindow, frames: SlidingWindow, num_chunks: int
                   ) -> Tuple[np.ndarray, int, SlidingWindow]:
    """start frame of every chunk and the total number of global frames, in the reference's float64
    arithmetic (core/inference.py:529-571, :596)."""
    out_frames = SlidingWindow(start=chunks.start, duration=frames.duration, step=frames.step)
    num_frames = out_frames.closest_frame(
        chunks.start + chunks.duration + (num_chunks - 1) * chunks.step + 0.5 * out_frames.duration) + 1
    c = np.arange(num_chunks, dtype=np.float64)
    t = (chunks.start + c * chunks.step) + 0.5 * out_frames.duration
    starts = np.rint((t - out_frames.start - 0.5 * out_frames.duration) / out_frames.step)
    return starts.astype(np.int32), int(num_frames), out_frames


def as_device_segmentation(seg, device: torch.device) -> torch.Tensor:
    """(C, F, S) uint8 device tensor from a device tensor or a host array of {0,1} (NaN -> 0)."""
    if isinstance(seg, torch.Tensor):
        return seg.to(device=device, dtype=torch.uint8).contiguous()
    return torch.from_numpy(np.nan_to_num(np.asarray(seg), nan=0.0).astype(np.uint8)).to(device)


@ffi.on_device(lambda seg, *a, **k: seg.device)
def chunk_stats(seg: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """-> active (C,S) int32 = #frames speaker s is on, clean (C,S) int32 = #frames it speaks alone."""
    C, F, S = seg.shape
    active = torch.empty((C, S), dtype=torch.int32, device=seg.device)
    clean = torch.empty((C, S), dtype=torch.int32, device=seg.device)
    ffi.check(ffi.load().pa_seg_chunk_stats(ffi.ptr(seg), C, F, S, ffi.ptr(active), ffi.ptr(clean),
                                            ffi.stream()), "pa_seg_chunk_stats")
    return active, clean


@ffi.on_device(lambda seg, *a, **k: seg.device)
def embedding_masks(seg: torch.Tensor, clean: torch.Tensor, exclude_overlap: bool,
                    min_num_frames: int) -> torch.Tensor:
    C, F, S = seg.shape
    masks = torch.empty((C, S, F), dtype=torch.float32, device=seg.device)
    ffi.check(ffi.load().pa_embedding_masks(ffi.ptr(seg), C, F, S, ffi.ptr(clean), int(exclude_overlap),
                                            int(min_num_frames), ffi.ptr(masks), ffi.stream()),
              "pa_embedding_masks")
    return masks


@ffi.on_device(lambda seg, *a, **k: seg.device)
def speaker_count(seg: torch.Tensor, chunks: SlidingWindow, frames: SlidingWindow
                  ) -> SlidingWindowFeature:
    """pipelines/utils/diarization.py:150-185 with warm_up=(0, 0)."""
    C, F, S = seg.shape
    starts, T, out_frames = frame_geometry(chunks, frames, C)
    dev = seg.device
    st = torch.from_numpy(starts).to(dev)
    count = torch.empty(T, dtype=torch.uint8, device=dev)
    scratch = torch.empty(2 * T, dtype=torch.int32, device=dev)
    ffi.check(ffi.load().pa_speaker_count(ffi.ptr(seg), C, F, S, ffi.ptr(st), T, ffi.ptr(count),
                                          ffi.ptr(scratch), ffi.stream()), "pa_speaker_count")
    return SlidingWindowFeature(count.cpu().numpy().reshape(T, 1), out_frames)


def aggregate(scores, chunks: SlidingWindow, frames: SlidingWindow, device: torch.device,
              warm_up: Tuple[float, float] = (0.0, 0.0), epsilon: float = 1e-12, hamming: bool = False,
              missing: float = np.nan, skip_average: bool = False) -> SlidingWindowFeature:
    """`Inference.aggregate` (core/inference.py:498-620) on the GPU: scores (C, F, K) host array or device
    tensor -> SlidingWindowFeature (T, K) float32, bit-identical to the reference's chunk loop."""
    x = torch.as_tensor(scores).to(device=device, dtype=torch.float32).contiguous()
    C, F, K = x.shape
    starts, T, out_frames = frame_geometry(chunks, frames, C)
    window = np.hamming(F) if hamming else np.ones(F)
    warm = np.ones(F)
    left = round(warm_up[0] / chunks.duration * F)
    right = round(warm_up[1] / chunks.duration * F)
    warm[:left] = epsilon
    warm[F - right:] = epsilon
    with torch.cuda.device(device):
        w = torch.from_numpy(np.ascontiguousarray(window, dtype=np.float64)).to(device)
        wu = torch.from_numpy(warm).to(device)
        st = torch.from_numpy(starts).to(device)
        out = torch.empty((T, K), dtype=torch.float32, device=device)
        ffi.check(ffi.load().pa_aggregate(ffi.ptr(x), C, F, K, ffi.ptr(st), T, ffi.ptr(w), ffi.ptr(wu), float(epsilon),
                                          float(missing), int(skip_average), ffi.ptr(out), ffi.stream()),
                  "pa_aggregate")
    return SlidingWindowFeature(out.cpu().numpy(), out_frames)


@ffi.on_device(lambda scores, *a, **k: scores.device)
def binarize(scores: torch.Tensor, onset: float = 0.5, offset: float | None = None,
             initial_state: bool | None = None) -> torch.Tensor:
    """`binarize` (utils/signal.py:78-204) on the device: (C, F, K) float32 scores -> (C, F, K) uint8 by
    hysteresis thresholding per (chunk, class); what SpeakerDiarization.apply does to the segmentations of a
    non-powerset model (pipelines/speaker_diarization.py:599-606)."""
    offset = offset or onset
    x = scores.to(torch.float32).contiguous()
    C, F, K = x.shape
    out = torch.empty((C, F, K), dtype=torch.uint8, device=x.device)
    init = -1 if initial_state is None else int(bool(initial_state))
    ffi.check(ffi.load().pa_binarize_hysteresis(ffi.ptr(x), C, F, K, float(onset), float(offset), init,
                                                ffi.ptr(out), ffi.stream()), "pa_binarize_hysteresis")
    return out


class Reconstructor:
    """speaker_diarization.py:480-528 + diarization.py:221-268: cluster activations are accumulated
    once, then discretised for any per-frame cap (regular and exclusive diarization share them)."""

    @ffi.on_device(lambda self, seg, *a, **k: seg.device)
    def __init__(self, seg: torch.Tensor, chunks: SlidingWindow, frames: SlidingWindow,
                 hard_clusters: np.ndarray, count: np.ndarray):
        """`seg`: (C, F, S) uint8 hard segmentations (powerset models: sums of small integers), or float32
        soft scores (non-powerset models: the reference reconstructs from the RAW segmentations,
        speaker_diarization.py:687-691, so activations are float32 overlap-add sums of sigmoid scores)."""
        C, F, S = seg.shape
        self.soft = seg.dtype != torch.uint8
        dev = seg.device
        starts, T, self.frames = frame_geometry(chunks, frames, C)
        count = np.ascontiguousarray(count.reshape(-1))
        if len(count) != T:
            raise ValueError(f"count has {len(count)} frames, the chunks cover {T}")
        num_clusters = int(np.max(hard_clusters)) + 1
        # to_diarization pads the cluster axis up to max(count) (diarization.py:250-256)
        self.K = max(num_clusters, int(np.max(count)) if len(count) else 0, 1)
        self.T = T
        self.count = torch.from_numpy(count.astype(np.uint8)).to(dev)
        hard = torch.from_numpy(np.ascontiguousarray(hard_clusters, dtype=np.int32)).to(dev)
        st = torch.from_numpy(starts).to(dev)
        if self.soft:
            lib = ffi.load()
            clustered = torch.empty((C, F, self.K), dtype=torch.float32, device=dev)
            ffi.check(lib.pa_cluster_max(ffi.ptr(seg.contiguous()), C, F, S, ffi.ptr(hard), self.K,
                                         ffi.ptr(clustered), ffi.stream()), "pa_cluster_max")
            ones = torch.ones(F, dtype=torch.float64, device=dev)
            self.act = torch.empty((T, self.K), dtype=torch.float32, device=dev)
            # aggregate(hamming=False, missing=0, skip_average=True) (diarization.py:243-249)
            ffi.check(lib.pa_aggregate(ffi.ptr(clustered), C, F, self.K, ffi.ptr(st), T, ffi.ptr(ones),
                                       ffi.ptr(ones), 1e-12, 0.0, 1, ffi.ptr(self.act), ffi.stream()),
                      "pa_aggregate")
            return
        self.act = torch.empty((T, self.K), dtype=torch.int32, device=dev)
        ffi.check(ffi.load().pa_cluster_activations(ffi.ptr(seg), C, F, S, ffi.ptr(st), ffi.ptr(hard),
                                                    self.K, T, ffi.ptr(self.act), ffi.stream()),
                  "pa_cluster_activations")

    @ffi.on_device(lambda self, *a, **k: self.act.device)
    def discretize(self, cap: int = 255) -> SlidingWindowFeature:
        """Top-min(count[t], cap) clusters per frame.  Frames whose selection boundary falls inside a
        group of EQUAL activations are re-decided with `np.argsort(-activations)` -- the reference's
        own call (diarization.py:261), whose order among equals depends on the host's numpy build
        (SIMD sorting networks) -- so the output equals the reference's on this host, not merely up
        to ties.  Every other frame is decided on the GPU."""
        dev = self.act.device
        out = torch.empty((self.T, self.K), dtype=torch.uint8, device=dev)
        tie = torch.empty(self.T, dtype=torch.uint8, device=dev)
        topk = ffi.load().pa_topk_binarize_f32 if self.soft else ffi.load().pa_topk_binarize
        ffi.check(topk(ffi.ptr(self.act), ffi.ptr(self.count), self.T, self.K, int(cap), ffi.ptr(out),
                       ffi.ptr(tie), ffi.stream()), "pa_topk_binarize")
        idx = torch.nonzero(tie).view(-1)
        binary = out.cpu().numpy().astype(np.float32)
        if idx.numel():
            rows = idx.cpu().numpy()
            act = self.act[idx].cpu().numpy().astype(np.float32)
            n = np.minimum(np.minimum(self.count[idx].cpu().numpy().astype(np.int64), cap), self.K)
            order = np.argsort(-act, axis=-1)
            keep = np.arange(self.K)[None, :] < n[:, None]
            fixed = np.zeros_like(act)
            fixed[np.nonzero(keep)[0], order[keep]] = 1.0
            binary[rows] = fixed
        self.num_tie_frames = int(idx.numel())
        return SlidingWindowFeature(binary, self.frames)
