"""Embedding-extractor wrapper: the b3 interface of SURVEY.md section 8b
(mirrors pipelines/speaker_verification.py:622-778 and pipelines/utils/getter.py:74-136)."""
from __future__ import annotations

from functools import cached_property
from pathlib import Path
from typing import Mapping, Optional, Union

import numpy as np
import torch

from . import ffi
from .model import Model
from .pipeline import BaseInference

PipelineModel = Union[Model, str, Mapping]


def get_model(model: PipelineModel, token=None, cache_dir=None) -> Model:
    """getter.py:74-136: Model instance, checkpoint path, or kwargs of Model.from_pretrained."""
    if isinstance(model, Model):
        pass
    elif isinstance(model, (str, Path)):
        model = Model.from_pretrained(model, token=token, cache_dir=cache_dir, strict=False)
    elif isinstance(model, Mapping):
        model = dict(model)
        model.setdefault("token", token)
        model.setdefault("cache_dir", cache_dir)
        model = Model.from_pretrained(**model)
    else:
        raise TypeError(f"Unsupported type ({type(model)}) for loading model: "
                        f"expected `str` or `dict`.")
    model.eval()
    return model


class PyannoteAudioPretrainedSpeakerEmbedding(BaseInference):
    def __init__(self, embedding: PipelineModel, device: Optional[torch.device] = None, token=None,
                 cache_dir=None):
        super().__init__()
        self.embedding = embedding
        self.device = device or torch.device("cpu")
        self.model_: Model = get_model(self.embedding, token=token, cache_dir=cache_dir)
        self.model_.eval()
        self.model_.to(self.device)

    def to(self, device: torch.device):
        if not isinstance(device, torch.device):
            raise TypeError(
                f"`device` must be an instance of `torch.device`, got `{type(device).__name__}`")
        self.model_.to(device)
        self.device = device
        return self

    @cached_property
    def sample_rate(self) -> int:
        return self.model_.audio.sample_rate

    @cached_property
    def dimension(self) -> int:
        return self.model_.dimension

    @cached_property
    def metric(self) -> str:
        return "cosine"

    @cached_property
    def min_num_samples(self) -> int:
        """speaker_verification.py:688-702: smallest input the model accepts, by bisection.  The
        reference probes by running the model and catching exceptions; the only failure mode of this
        architecture is "shorter than one fbank frame", which the C ABI exposes directly."""
        lib = ffi.load()
        lower, upper = 2, round(0.5 * self.sample_rate)
        middle = (lower + upper) // 2
        while lower + 1 < upper:
            if lib.pa_emb_num_fbank_frames(middle) > 0:
                upper = middle
            else:
                lower = middle
            middle = (lower + upper) // 2
        return upper

    def __call__(self, waveforms: torch.Tensor, masks: Optional[torch.Tensor] = None) -> np.ndarray:
        """(B,1,N) [, (B,F)] -> (B,D) float32 ndarray"""
        emb = self.model_(waveforms, weights=masks)
        return emb.cpu().numpy()


def PretrainedSpeakerEmbedding(embedding: PipelineModel, device: Optional[torch.device] = None,
                               token=None, cache_dir=None):
    """speaker_verification.py:719-778.  SpeechBrain / NeMo / ONNX back-ends are third-party runtimes
    outside the accelerated path; everything else is a pyannote-format checkpoint."""
    if isinstance(embedding, str) and any(k in embedding for k in ("speechbrain", "nvidia")) \
            and not Path(embedding).exists():
        raise NotImplementedError(f"{embedding}: SpeechBrain/NeMo embedding back-ends are out of scope")
    return PyannoteAudioPretrainedSpeakerEmbedding(embedding, device=device, token=token,
                                                   cache_dir=cache_dir)
