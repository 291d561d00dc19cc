"""Sliding-window inference engine, device resident (mirrors core/inference.py: __init__ :78-167,
slide :217-373, aggregate :498-620, trim :622-667).

Differences in mechanism, not in results: the waveform is uploaded to HBM once and chunks are
gathered by the kernels through (offset, stride) arithmetic, instead of `unfold` + one H2D copy and
one D2H copy per batch (the reference's hot loop #1, inference.py:295-305)."""
from __future__ import annotations

import time
import warnings
from typing import Callable, Optional, Tuple

import numpy as np
import torch

from .audio import AudioFile, Audio
from .core import Segment, SlidingWindow, SlidingWindowFeature
from .model import Model, Resolution
from .pipeline import BaseInference


class Inference(BaseInference):
    """`Inference(model, window="sliding", duration=None, step=None, pre_aggregation_hook=None,
    skip_aggregation=False, skip_conversion=False, device=None, batch_size=32)` -- the constructor
    contract of core/inference.py:78-167."""

    def __init__(self, model: Model, window: str = "sliding", duration: Optional[float] = None,
                 step: Optional[float] = None, pre_aggregation_hook: Optional[Callable] = None,
                 skip_aggregation: bool = False, skip_conversion: bool = False,
                 device: Optional[torch.device] = None, batch_size: int = 32):
        if window not in ("sliding", "whole"):
            raise ValueError('`window` must be "sliding" or "whole".')
        self.model = model
        self.device = device if device is not None else model.device
        model.eval()
        model.to(self.device)
        spec = model.specifications
        self.window = window
        if window == "whole" and spec.resolution == Resolution.FRAME:
            warnings.warn('"whole"-window inference of a frame-level model processes the entire file as '
                          'one chunk (memory grows with the file, and the model never saw such lengths): '
                          'prefer window="sliding".')
        self.duration = duration or spec.duration
        if self.duration != spec.duration:
            warnings.warn(f"chunk duration {self.duration:g}s differs from the {spec.duration:g}s the "
                          f"model was trained with: expect degraded results.")
        self.warm_up = spec.warm_up
        # default hop: 10 % of a chunk, or the left warm-up when the model declares one
        self.step = step or (self.warm_up[0] if self.warm_up[0] != 0.0 else 0.1 * self.duration)
        if self.step > self.duration:
            raise ValueError(
                f"Step between consecutive chunks is set to {self.step:g}s, while chunks are only "
                f"{self.duration:g}s long, leading to gaps between consecutive chunks. Either decrease "
                f"step or increase duration.")
        self.skip_conversion = skip_conversion
        self.skip_aggregation = skip_aggregation
        self.pre_aggregation_hook = pre_aggregation_hook
        # kept for API compatibility: the engine sizes its own launch groups; the
        # value only sets the granularity at which `hook` is called.
        self.batch_size = batch_size
        # device-side copy of the last hard segmentation, for the embedding stage
        self.last_device_output: Optional[torch.Tensor] = None
        self.last_host_output: Optional[np.ndarray] = None   # its host image (identity-checked)
        self.last_enqueued: float = 0.0

    def to(self, device: torch.device) -> "Inference":
        if not isinstance(device, torch.device):
            raise TypeError(
                f"`device` must be an instance of `torch.device`, got `{type(device).__name__}`")
        self.model.to(device)
        self.device = device
        return self

    # ---------------------------------------------------------------------------------------
    def _forward(self, wav: torch.Tensor, stride: int, count: int, window: int, want_logp: bool,
                 want_multilabel: bool):
        """strided chunks of a flat device waveform through the segmentation engine; running out of
        device memory is reported the way core/inference.py:199-208 reports it (MemoryError + advice)"""
        try:
            return self.model.engine.forward_strided(wav, stride, count, window, want_logp=want_logp,
                                                     want_multilabel=want_multilabel)
        except (MemoryError, torch.OutOfMemoryError):
            raise MemoryError(f"batch_size ({self.batch_size: d}) is probably too large. "
                              f"Try with a smaller value until memory error disappears.") from None

    def infer(self, chunks: torch.Tensor) -> np.ndarray:
        """(batch, 1, samples) chunks -> (batch, frames, classes) numpy, converted from powerset to hard
        multilabel unless `skip_conversion` (core/inference.py:182-215)."""
        if chunks.dim() != 3 or chunks.shape[1] != 1:
            raise ValueError("`chunks` must be a (batch_size, 1, num_samples) tensor (mono models)")
        batch, _, num_samples = chunks.shape
        flat = chunks.to(self.model.device, torch.float32).contiguous().view(-1)
        convert = bool(self.model.specifications.powerset) and not self.skip_conversion
        logp, ml = self._forward(flat, num_samples, batch, num_samples, want_logp=not convert,
                                 want_multilabel=convert)
        return (ml.to(torch.float32) if convert else logp).cpu().numpy()

    def crop(self, file: AudioFile, chunk, hook: Optional[Callable] = None):
        """Inference on one excerpt, or on the smallest excerpt containing a list of them (sliding window),
        or on their concatenation (whole window) -- core/inference.py:420-496.  Sliding-window output
        frames start at the excerpt's start time."""
        audio = Audio(self.model.audio.sample_rate, mono="downmix", device=self.device)
        if self.window == "sliding":
            if not isinstance(chunk, Segment):
                chunk = Segment(min(c.start for c in chunk), max(c.end for c in chunk))
            waveform, sample_rate = audio.crop(file, chunk)
            output = self.slide(waveform, sample_rate, hook=hook)
            frames = output.sliding_window
            shifted = SlidingWindow(start=chunk.start, duration=frames.duration, step=frames.step)
            return SlidingWindowFeature(output.data, shifted)
        if isinstance(chunk, Segment):
            waveform, _ = audio.crop(file, chunk)
        else:
            waveform = torch.cat([audio.crop(file, c)[0] for c in chunk], dim=1)
        return self.infer(waveform[None])[0]

    def slide(self, waveform: torch.Tensor, sample_rate: int, hook: Optional[Callable] = None,
              chunk_range: Optional[Tuple[int, int]] = None) -> SlidingWindowFeature:
        """waveform: (1, num_samples), host or device.  `chunk_range` restricts processing to chunks
        [begin, end) (multi-GPU sharding); geometry is always that of the whole file."""
        specifications = self.model.specifications
        if specifications.resolution != Resolution.FRAME:
            raise NotImplementedError("the accelerated sliding window covers frame-level segmentation "
                                      "models (powerset, as on the 3.1 hot path, or multi-label)")
        window_size: int = self.model.audio.get_num_samples(self.duration)
        step_size: int = round(self.step * sample_rate)
        _, num_samples = waveform.shape
        num_chunks, has_last_chunk = self.num_chunks(num_samples, window_size, step_size)
        total = num_chunks + has_last_chunk
        begin, end = (0, total) if chunk_range is None else chunk_range

        wav = waveform.to(self.model.device, torch.float32).contiguous().view(-1)
        if hook is not None:
            hook(completed=0, total=total)
        # multi-label checkpoints have no powerset conversion: their sigmoid scores are the output and the
        # pipeline binarizes them itself (pipelines/speaker_diarization.py:599-606)
        want_logp = self.skip_conversion or not specifications.powerset
        sub = wav[begin * step_size:]
        logp, ml = self._forward(sub, step_size, end - begin, window_size,
                                 want_logp=want_logp, want_multilabel=not want_logp)
        self.last_device_output = ml if ml is not None else logp
        self.last_enqueued = time.perf_counter()     # host clock when the launch group was queued
        outputs = (logp if want_logp else ml.to(torch.float32)).cpu().numpy()
        self.last_host_output = None if want_logp else outputs
        if hook is not None:
            hook(completed=total, total=total)

        frames = self.model.receptive_field
        if (self.skip_aggregation or (specifications.permutation_invariant
                                      and self.pre_aggregation_hook is None)):
            chunks = SlidingWindow(start=begin * self.step, duration=self.duration, step=self.step) \
                if begin else SlidingWindow(start=0.0, duration=self.duration, step=self.step)
            return SlidingWindowFeature(outputs, chunks)

        if self.pre_aggregation_hook is not None:
            outputs = self.pre_aggregation_hook(outputs)
        # Hamming-weighted overlap-add on the GPU (pa_aggregate), bit-identical to `self.aggregate`
        from . import frames as frame_ops
        aggregated = frame_ops.aggregate(
            outputs, SlidingWindow(start=0.0, duration=self.duration, step=self.step), frames,
            self.model.device, warm_up=self.warm_up, hamming=True, missing=0.0)
        if has_last_chunk:
            aggregated.data = aggregated.crop(Segment(0.0, num_samples / sample_rate), mode="loose")
        return aggregated

    @staticmethod
    def num_chunks(num_samples: int, window_size: int, step_size: int) -> Tuple[int, bool]:
        """inference.py:258-278: complete chunks from `unfold` + one zero-padded orphan chunk."""
        n = (num_samples - window_size) // step_size + 1 if num_samples >= window_size else 0
        has_last = (num_samples < window_size) or (num_samples - window_size) % step_size > 0
        return n, bool(has_last)

    def __call__(self, file: AudioFile, hook: Optional[Callable] = None):
        waveform, sample_rate = Audio(self.model.audio.sample_rate, mono="downmix", device=self.device)(file)
        if self.window == "sliding":
            return self.slide(waveform, sample_rate, hook=hook)
        return self.infer(waveform[None])[0]            # core/inference.py:412-418

    # ---------------------------------------------------------------------------------------
    @staticmethod
    def aggregate(scores: SlidingWindowFeature, frames: SlidingWindow,
                  warm_up: Tuple[float, float] = (0.0, 0.0), epsilon: float = 1e-12,
                  hamming: bool = False, missing: float = np.nan,
                  skip_average: bool = False) -> SlidingWindowFeature:
        """Overlap-add aggregation (inference.py:498-620).  Same arithmetic, in the same order
        (contributions are added chunk by chunk, in float64, into float32 accumulators)."""
        data = scores.data if isinstance(scores, SlidingWindowFeature) else np.asarray(scores)
        num_chunks, F, K = data.shape
        chunks = scores.sliding_window
        frames = SlidingWindow(start=chunks.start, duration=frames.duration, step=frames.step)
        w = np.hamming(F).reshape(-1, 1) if hamming else np.ones((F, 1))
        warm = np.ones((F, 1))
        left = round(warm_up[0] / chunks.duration * F)
        warm[:left] = epsilon
        right = round(warm_up[1] / chunks.duration * F)
        warm[F - right:] = epsilon
        num_frames = frames.closest_frame(
            chunks.start + chunks.duration + (num_chunks - 1) * chunks.step + 0.5 * frames.duration) + 1
        agg = np.zeros((num_frames, K), dtype=np.float32)
        cnt = np.zeros((num_frames, K), dtype=np.float32)
        msk = np.zeros((num_frames, K), dtype=np.float32)
        starts = aggregate_start_frames(chunks, frames, num_chunks)
        nan = np.isnan(data)
        has_nan = bool(nan.any())
        ww = w * warm
        for c in range(num_chunks):
            s = starts[c]
            if has_nan:
                m = 1 - nan[c]
                sc = np.where(nan[c], 0.0, data[c])
                agg[s:s + F] += sc * m * w * warm
                cnt[s:s + F] += m * w * warm
                np.maximum(msk[s:s + F], m, out=msk[s:s + F])
            else:
                agg[s:s + F] += data[c] * w * warm
                cnt[s:s + F] += ww
        if not has_nan:
            if num_chunks:
                msk[starts[0]:starts[-1] + F] = 1.0
                # frames between two non-adjacent chunks (step > duration never happens) stay 0
        average = agg if skip_average else agg / np.maximum(cnt, epsilon)
        average[msk == 0.0] = missing
        return SlidingWindowFeature(average, frames)

    @staticmethod
    def trim(scores: SlidingWindowFeature, warm_up: Tuple[float, float] = (0.1, 0.1)
             ) -> SlidingWindowFeature:
        """drop the warm-up frames at both ends of every chunk and shrink the chunk grid accordingly
        (core/inference.py:622-667); `warm_up` are fractions of a chunk."""
        if scores.data.ndim != 3:
            raise ValueError("trim expects (chunks, frames, classes) scores")
        num_frames = scores.data.shape[1]
        grid = scores.sliding_window
        lead, tail = warm_up
        first, last = round(num_frames * lead), num_frames - round(num_frames * tail)
        if last - first < round(num_frames * grid.step / grid.duration):
            warnings.warn(f"a warm-up of {100 * (lead + tail):g}% leaves less than one hop "
                          f"({grid.step:g}s) of every chunk: consecutive chunks no longer overlap")
        trimmed = SlidingWindow(start=grid.start + lead * grid.duration, step=grid.step,
                                duration=(1 - lead - tail) * grid.duration)
        return SlidingWindowFeature(scores.data[:, first:last], trimmed)


def aggregate_start_frames(chunks: SlidingWindow, frames: SlidingWindow, num_chunks: int) -> np.ndarray:
    """start frame of every chunk: frames.closest_frame(chunk.start + 0.5 * frames.duration)
    (inference.py:596), evaluated for all chunks at once with identical float64 operations."""
    c = np.arange(num_chunks, dtype=np.float64)
    chunk_start = chunks.start + c * chunks.step
    t = chunk_start + 0.5 * frames.duration
    return np.rint((t - frames.start - 0.5 * frames.duration) / frames.step).astype(np.int64)
