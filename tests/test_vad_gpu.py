"""GPU: aggregated `Inference` + `VoiceActivityDetection` (SURVEY.md section 8f-4;
core/inference.py:349-369, 498-620; pipelines/voice_activity_detection.py:66-218)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("hamming,warm_up,skip_average,with_nan", [
    (True, (0.0, 0.0), False, False), (False, (0.0, 0.0), True, True), (True, (0.1, 0.1), False, True),
    (False, (0.05, 0.2), False, False)])
def test_aggregate_kernel_bit_identical_to_chunk_loop(gpu_device, hamming, warm_up, skip_average, with_nan):
    from pyannote_audio_amd import frames as fo
    from pyannote_audio_amd.core import SlidingWindow, SlidingWindowFeature
    from pyannote_audio_amd.inference import Inference
    rng = np.random.default_rng(3)
    C, F, K = 57, 293, 4
    scores = rng.uniform(size=(C, F, K)).astype(np.float32)
    if with_nan:
        scores[rng.uniform(size=scores.shape) < 0.2] = np.nan
        scores[:3] = np.nan                                    # frames nobody voted on -> `missing`
    chunks = SlidingWindow(start=0.0, duration=5.0, step=0.5)
    frames = SlidingWindow(start=0.0, duration=0.0619375, step=0.016875)
    kw = dict(warm_up=warm_up, hamming=hamming, missing=0.0, skip_average=skip_average)
    want = Inference.aggregate(SlidingWindowFeature(scores.copy(), chunks), frames, **kw)
    got = fo.aggregate(scores, chunks, frames, gpu_device, **kw)
    assert got.data.shape == want.data.shape and got.data.dtype == np.float32
    assert np.array_equal(got.data, want.data)
    assert (got.sliding_window.start, got.sliding_window.step) == (want.sliding_window.start,
                                                                    want.sliding_window.step)


@pytest.mark.parametrize("seconds,seed", [(33.0, 5), (10.5, 8)])
def test_voice_activity_detection_matches_oracle(pipeline_dir, synthetic_models, gpu_device, seconds, seed):
    import pyannote_audio_amd as pa
    from oracle import pipeline as op
    from oracle.synthetic import synth_conversation
    seg_o, _ = synthetic_models
    wav, _ = synth_conversation(seconds, seed=seed)
    vad = pa.VoiceActivityDetection(segmentation=os.path.join(pipeline_dir, "segmentation"))
    vad.instantiate({"min_duration_on": 0.0, "min_duration_off": 0.0}).to(gpu_device)
    art = {}
    speech = vad({"waveform": wav, "sample_rate": 16000, "uri": "vad"},
                 hook=lambda step, a, **kw: art.__setitem__(step, a) if (a is not None and kw.get("total") is None) else None)
    assert isinstance(speech, pa.Annotation) and speech.uri == "vad"
    assert speech.labels() in ([], ["SPEECH"])
    # oracle: slide -> hard multilabel -> max over speakers -> Hamming overlap-add -> crop -> binarize
    chunks, frames = op.SW(0.0, 10.0, 1.0), op.SW(0.0, 0.0619375, 0.016875)
    seg = op.slide(seg_o, wav, 16000, 10.0, 1.0, 32)
    hooked = np.max(seg, axis=-1, keepdims=True)
    agg, _ = op.aggregate(hooked, chunks, frames, hamming=True, missing=0.0)
    n = wav.shape[1]
    if n < 160000 or (n - 160000) % 16000 > 0:               # "remove padding that was added to last chunk"
        keep = int(np.floor((n / 16000.0 - frames.start) / frames.step)) + 1
        agg = agg[:keep]
    scores = art["segmentation"]
    assert scores.data.shape == agg.shape
    assert np.array_equal(scores.data, agg.astype(np.float32))
    want = op.binarize(agg, frames, onset=0.5, offset=0.5)
    got = [(s.start, s.end) for s, _ in speech.itertracks()]
    assert got == [(a, b) for a, b, _, _ in want]
