"""pipelines/utils/hook.py:37-240: ArtifactHook / TimingHook / ProgressHook / Hooks driven with the call
sequence SpeakerDiarization.apply emits (progress reports carry total / completed and no artifact, results
carry the artifact), bound to the file by Pipeline.setup_hook as core/pipeline.py does."""
import time

import numpy as np
import torch

import pyannote_audio_amd as pa
from pyannote_audio_amd.hook import ArtifactHook, Hooks, ProgressHook, TimingHook


def _play(hook, file):
    """what one apply() reports, in order"""
    bound = pa.Pipeline.setup_hook(file, hook=hook)
    bound("segmentation", None, total=21, completed=0)
    time.sleep(0.02)
    bound("segmentation", None, total=21, completed=21)
    bound("segmentation", np.zeros((21, 589, 3), dtype=np.float32))
    bound("speaker_counting", np.ones((100, 1), dtype=np.uint8))
    bound("embeddings", None, total=2, completed=0)
    time.sleep(0.01)
    bound("embeddings", torch.ones(2, 3), total=2, completed=2)
    bound("discrete_diarization", np.eye(3))


def test_artifact_hook_keeps_copies():
    file = {"uri": "x"}
    with ArtifactHook() as hook:
        _play(hook, file)
    kept = file["artifact"]
    assert set(kept) == {"segmentation", "speaker_counting", "embeddings", "discrete_diarization"}
    assert isinstance(kept["embeddings"], np.ndarray) and kept["embeddings"].shape == (2, 3)   # tensors -> numpy
    file2 = {"uri": "y"}
    with ArtifactHook("embeddings", file_key="kept") as hook:
        _play(hook, file2)
    assert set(file2["kept"]) == {"embeddings"}


def test_timing_hook_times_steps_that_report_progress():
    file = {"uri": "x"}
    with TimingHook() as hook:
        _play(hook, file)
    t = file["timing"]
    assert set(t) == {"total", "segmentation", "embeddings"}        # only steps with completed / total reports
    assert 0.015 < t["segmentation"] <= t["total"] and 0.005 < t["embeddings"] <= t["total"]
    file2 = {"uri": "y"}
    with TimingHook(file_key="clock") as hook:
        _play(hook, file2)
    assert "clock" in file2


def test_progress_hook_hidden_and_visible():
    with ProgressHook(hidden=True) as hook:
        _play(hook, {"uri": "x"})
        assert not hasattr(hook, "progress")
    with ProgressHook(transient=True) as hook:
        _play(hook, {"uri": "x"})
        names = [task.description for task in hook.progress.tasks]
    assert names == ["segmentation", "speaker_counting", "embeddings", "discrete_diarization"]
    assert all(task.finished for task in hook.progress.tasks)


def test_hooks_fan_out():
    file = {"uri": "x"}
    calls = []
    with Hooks(ArtifactHook("speaker_counting"), TimingHook(),
               lambda step, artifact, file=None, total=None, completed=None: calls.append(step)) as hook:
        _play(hook, file)
    assert "artifact" in file and "timing" in file and len(calls) == 7
