"""Host logic of Inference.infer / crop / __call__(window="whole") (core/inference.py:182-215, 396-496)
with the segmentation engine replaced by a recording stub: which engine call is made, how its outputs are
converted, how the frames of a cropped excerpt are shifted, how device OOM is reported.  The engine calls
themselves are covered on the GPU (tests/test_seg_gpu.py)."""
import numpy as np
import pytest
import torch

import pyannote_audio_amd as pa
from pyannote_audio_amd.core import Segment, SlidingWindow, SlidingWindowFeature
from pyannote_audio_amd.inference import Inference
from pyannote_audio_amd.model import Resolution


class _Engine:
    def __init__(self, fail=None):
        self.calls = []
        self.fail = fail

    def forward_strided(self, wav, stride, count, window, want_logp=True, want_multilabel=True):
        if self.fail is not None:
            raise self.fail
        self.calls.append((wav.numel(), stride, count, window, want_logp, want_multilabel))
        frames = 5
        logp = torch.full((count, frames, 7), -1.0) if want_logp else None
        ml = torch.ones((count, frames, 3), dtype=torch.uint8) if want_multilabel else None
        return logp, ml


class _Spec:
    duration, warm_up, powerset, permutation_invariant = 10.0, (0.0, 0.0), True, True
    resolution = Resolution.FRAME


class _Audio:
    sample_rate = 16000

    def get_num_samples(self, duration, sample_rate=None):
        return round(duration * (sample_rate or 16000))


class _Model:
    def __init__(self, engine):
        self.engine, self.specifications, self.audio = engine, _Spec(), _Audio()
        self.device = torch.device("cpu")
        self.receptive_field = SlidingWindow(start=0.0, duration=0.0619375, step=0.016875)

    def eval(self):
        return self

    def to(self, device):
        return self


def test_infer_converts_powerset_unless_told_not_to():
    engine = _Engine()
    chunks = torch.zeros(4, 1, 160000)
    out = Inference(_Model(engine)).infer(chunks)
    assert out.shape == (4, 5, 3) and out.dtype == np.float32 and engine.calls[-1] == (640000, 160000, 4, 160000, False, True)
    out = Inference(_Model(engine), skip_conversion=True).infer(chunks)
    assert out.shape == (4, 5, 7) and engine.calls[-1][4:] == (True, False)
    with pytest.raises(ValueError):
        Inference(_Model(engine)).infer(torch.zeros(4, 2, 16000))


def test_out_of_memory_is_reported_like_the_reference():
    for error in (MemoryError("rc 2"), torch.OutOfMemoryError("HIP out of memory")):
        inference = Inference(_Model(_Engine(fail=error)), batch_size=32)
        with pytest.raises(MemoryError, match=r"batch_size \( 32\) is probably too large"):
            inference.infer(torch.zeros(1, 1, 16000))


def test_whole_window_goes_through_infer():
    engine = _Engine()
    with pytest.warns(UserWarning):                                  # frame-level model on a whole file
        inference = Inference(_Model(engine), window="whole")
    out = inference({"waveform": torch.zeros(1, 48000), "sample_rate": 16000})
    assert out.shape == (5, 3) and engine.calls[-1] == (48000, 48000, 1, 48000, False, True)
    out = inference.crop({"waveform": torch.zeros(1, 160000), "sample_rate": 16000},
                         [Segment(1.0, 2.0), Segment(4.0, 4.5)])
    assert out.shape == (5, 3) and engine.calls[-1][0] == 16000 + 8000   # the excerpts are concatenated


def test_sliding_crop_shifts_the_frames(monkeypatch):
    inference = Inference(_Model(_Engine()), skip_aggregation=True)
    seen = {}

    def fake_slide(waveform, sample_rate, hook=None, chunk_range=None):
        seen["shape"], seen["rate"] = tuple(waveform.shape), sample_rate
        return SlidingWindowFeature(np.zeros((3, 5, 3), dtype=np.float32),
                                    SlidingWindow(start=0.0, duration=10.0, step=1.0))

    monkeypatch.setattr(inference, "slide", fake_slide)
    file = {"waveform": torch.zeros(1, 40 * 16000), "sample_rate": 16000}
    out = inference.crop(file, Segment(12.0, 24.0))
    assert seen == {"shape": (1, 12 * 16000), "rate": 16000}
    assert out.sliding_window.start == 12.0 and out.sliding_window.duration == 10.0 and out.sliding_window.step == 1.0
    out = inference.crop(file, [Segment(20.0, 22.0), Segment(5.0, 6.0)])       # smallest excerpt containing all
    assert seen["shape"] == (1, 17 * 16000) and out.sliding_window.start == 5.0
