"""Voice activity detection pipeline on the accelerated segmentation path
(mirrors pipelines/voice_activity_detection.py:66-218).

Same segmentation kernels as the diarization pipeline, different epilogue: the per-chunk speaker
activations collapse to "somebody speaks" (max over speakers), are aggregated over overlapping chunks
with a Hamming window on the GPU (`pa_aggregate`, bit-identical to `Inference.aggregate`,
core/inference.py:498-620) and binarised with hysteresis (`Binarize`, utils/signal.py:207-318)."""
from __future__ import annotations

from typing import Callable, Optional

import numpy as np

from .audio import AudioFile
from .core import Annotation, SlidingWindowFeature
from .diarization import Binarize
from .inference import Inference
from .pipeline import Pipeline, Uniform
from .speaker_verification import PipelineModel, get_model


def any_speaker(scores: np.ndarray) -> np.ndarray:
    """(chunks, frames, speakers) -> (chunks, frames, 1): the most active speaker of every frame"""
    return np.max(scores, axis=-1, keepdims=True)


class VoiceActivityDetection(Pipeline):
    """Hyper-parameters: onset / offset (fixed to 0.5 for powerset models, whose outputs are already
    hard decisions), min_duration_on, min_duration_off."""

    def __init__(self, segmentation: PipelineModel = None, fscore: bool = False, token=None, cache_dir=None,
                 **inference_kwargs):
        super().__init__()
        if segmentation is None:
            raise ValueError("`segmentation` must be a local checkpoint (or a Model instance): Hugging Face "
                             "defaults cannot be downloaded in this build.")
        self.segmentation = segmentation
        self.fscore = fscore
        model = get_model(segmentation, token=token, cache_dir=cache_dir)
        self._segmentation = Inference(model, **{**inference_kwargs, "pre_aggregation_hook": any_speaker})
        if model.specifications.powerset:
            self.onset = self.offset = 0.5
        else:
            self.onset = Uniform(0.0, 1.0)
            self.offset = Uniform(0.0, 1.0)
        self.min_duration_on = Uniform(0.0, 1.0)
        self.min_duration_off = Uniform(0.0, 1.0)

    def default_parameters(self):
        # segmentation-3.0 style (powerset) models: no smoothing by default (:143-147)
        if self._segmentation.model.specifications.powerset:
            return {"min_duration_on": 0.0, "min_duration_off": 0.0}
        raise NotImplementedError()

    def classes(self):
        return ["SPEECH"]

    def initialize(self):
        self._binarize = Binarize(onset=self.onset, offset=self.offset,
                                  min_duration_on=self.min_duration_on,
                                  min_duration_off=self.min_duration_off)

    def apply(self, file: AudioFile, hook: Optional[Callable] = None) -> Annotation:
        """-> speech regions, every track labelled "SPEECH" (:161-203)"""
        hook = self.setup_hook(file, hook=hook)
        progress = (lambda **kw: hook("segmentation", None, **kw))
        speech_scores: SlidingWindowFeature = self._segmentation(file, hook=progress)
        hook("segmentation", speech_scores)
        if not hasattr(self, "_binarize"):
            self.initialize()
        speech = self._binarize(speech_scores)
        speech.uri = file["uri"]
        return speech.rename_labels({label: "SPEECH" for label in speech.labels()})
