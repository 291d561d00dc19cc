"""Speaker-embedding extractor as the diarization pipeline sees it: the b3 interface of SURVEY.md
section 8b (pipelines/speaker_verification.py:622-778; model resolution pipelines/utils/getter.py:74-136).

Contract kept: `extractor(waveforms (B,1,N), masks (B,F) | None) -> (B, D) float32 ndarray`, `.to(device)`
(TypeError for anything but a torch.device), read-only `sample_rate`, `dimension`, `metric`,
`min_num_samples`.  The diarization pipeline itself does not go through `__call__`: it hands the
device-resident waveform and all masks of a chunk to the engine at once (`model_.engine`, one backbone
pass per chunk instead of one per (chunk, speaker))."""
from __future__ import annotations

from pathlib import Path
from typing import Callable, Mapping, Optional, Union

import numpy as np
import torch

from . import ffi
from .model import Model
from .pipeline import BaseInference

PipelineModel = Union[Model, str, Mapping]


def get_model(model: PipelineModel, token=None, cache_dir=None) -> Model:
    """anything the config may hold for a model -> a `Model` in eval mode: an instance, a checkpoint
    path / directory, or the keyword arguments of `Model.from_pretrained` (what `$model/<subfolder>`
    expands to)."""
    if isinstance(model, Mapping):
        options = {"token": token, "cache_dir": cache_dir, **model}
        model = Model.from_pretrained(**options)
    elif isinstance(model, (str, Path)):
        model = Model.from_pretrained(model, token=token, cache_dir=cache_dir, strict=False)
    elif not isinstance(model, Model):
        raise TypeError(f"Unsupported type ({type(model)}) for loading model: expected `str` or `dict`.")
    model.eval()
    return model


def first_true(predicate: Callable[[int], bool], lo: int, hi: int) -> int:
    """smallest n in (lo, hi] with predicate(n), for a monotone predicate with predicate(lo) false and
    predicate(hi) true"""
    while hi - lo > 1:
        mid = (lo + hi) // 2
        if predicate(mid):
            hi = mid
        else:
            lo = mid
    return hi


class PyannoteAudioPretrainedSpeakerEmbedding(BaseInference):
    """wrapper around a pyannote-format embedding checkpoint (WeSpeaker ResNet here)"""

    metric = "cosine"

    def __init__(self, embedding: PipelineModel, device: Optional[torch.device] = None, token=None,
                 cache_dir=None):
        super().__init__()
        self.embedding = embedding
        self.model_: Model = get_model(embedding, token=token, cache_dir=cache_dir)
        self.device = device or torch.device("cpu")
        self.model_.to(self.device)
        self._min_num_samples: Optional[int] = None

    def to(self, device: torch.device):
        if not isinstance(device, torch.device):
            raise TypeError(
                f"`device` must be an instance of `torch.device`, got `{type(device).__name__}`")
        self.model_.to(device)
        self.device = device
        return self

    @property
    def sample_rate(self) -> int:
        return self.model_.audio.sample_rate

    @property
    def dimension(self) -> int:
        return self.model_.dimension

    @property
    def min_num_samples(self) -> int:
        """shortest waveform the model can embed.  The reference finds it by running the model on
        shorter and shorter random inputs until it raises (:688-702); here the only failure mode is
        "shorter than one fbank frame", which the C ABI answers directly (`pa_emb_num_fbank_frames`)."""
        if self._min_num_samples is None:
            if hasattr(self.model_, "_TDNN"):     # x-vector: SincNet + TDNN must leave at least one frame
                enough = lambda n: self.model_.num_frames(n) > 0   # noqa: E731
            else:
                frames = ffi.load().pa_emb_num_fbank_frames
                enough = lambda n: frames(n) > 0                   # noqa: E731
            self._min_num_samples = first_true(enough, 2, round(0.5 * self.sample_rate))
        return self._min_num_samples

    def __call__(self, waveforms: torch.Tensor, masks: Optional[torch.Tensor] = None) -> np.ndarray:
        return self.model_(waveforms, weights=masks).cpu().numpy()


def PretrainedSpeakerEmbedding(embedding: PipelineModel, device: Optional[torch.device] = None,
                               token=None, cache_dir=None):
    """factory of :719-778: SpeechBrain / NeMo / ONNX names select third-party runtimes that are not part
    of this build; everything else is a pyannote-format checkpoint."""
    third_party = isinstance(embedding, str) and not Path(embedding).exists() and \
        any(vendor in embedding for vendor in ("speechbrain", "nvidia"))
    if third_party:
        raise NotImplementedError(f"{embedding}: SpeechBrain/NeMo embedding back-ends are out of scope")
    return PyannoteAudioPretrainedSpeakerEmbedding(embedding, device=device, token=token,
                                                   cache_dir=cache_dir)
