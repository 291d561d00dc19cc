"""Golden vectors (tests/golden/oracle_v1.npz, made by tests/golden/make_golden.py from the CPU oracle on
seeded inputs).  CPU: the oracle still reproduces them (pins the checker against drift).  GPU: the HIP
path reproduces them without the oracle in the loop."""
import os

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_v1.npz")


def _inputs():
    g = torch.Generator().manual_seed(7)
    wav = (0.1 * torch.randn(2, 1, 160000, generator=g)).clamp(-1, 1)
    masks = (torch.rand(2, 589, generator=g) < 0.7).float()
    return wav, masks


def test_oracle_reproduces_golden():
    from oracle import kaldi_fbank, seeded_pyannet, seeded_wespeaker
    gold = np.load(GOLD)
    wav, masks = _inputs()
    with torch.inference_mode():
        logp = seeded_pyannet(seed=1234, num_layers=4)(wav).numpy()[:, ::19]
        fb = kaldi_fbank(wav[0, :, :48000] * 32768.0).numpy()[::23]
        e = seeded_wespeaker(seed=4321)(wav[:, :, :48000], weights=masks).numpy()
    # (thread count / SIMD width of the host may re-associate fp32 sums: tolerance, not bit equality)
    assert np.allclose(logp, gold["seg_logp"], rtol=1e-4, atol=1e-5)
    assert np.allclose(fb, gold["fbank"], rtol=1e-4, atol=2e-4)
    assert np.abs(e - gold["embeddings"]).max() <= 1e-4 * np.abs(gold["embeddings"]).max()


def test_oracle_pipeline_reproduces_golden():
    from oracle.pipeline import diarize
    from oracle.synthetic import calibrated_pyannet, calibrated_wespeaker, synth_conversation
    gold = np.load(GOLD)
    conv, _ = synth_conversation(24.0, seed=3)
    out = diarize(calibrated_pyannet(calib_seconds=40.0), calibrated_wespeaker(calib_seconds=12.0), conv,
                  exclude_overlap=True)
    assert np.array_equal(out.count.reshape(-1)[::5].astype(np.uint8), gold["pipeline_count"])
    assert np.array_equal(out.hard_clusters.astype(np.int8), gold["pipeline_hard_clusters"])
    turns = np.array([(s, t) for s, t, _ in out.diarization], dtype=np.float64)
    assert turns.shape == gold["pipeline_turns"].shape and np.array_equal(turns, gold["pipeline_turns"])


@pytest.mark.gpu
def test_hip_path_reproduces_golden(gpu_device):
    from oracle import seeded_pyannet, seeded_wespeaker   # weights only; nothing is run on the CPU
    from pyannote_audio_amd.embedding import EmbeddingEngine
    from pyannote_audio_amd.segmentation import SegmentationEngine
    from pyannote_audio_amd.weights import EmbeddingPack, SegmentationPack
    gold = np.load(GOLD)
    wav, masks = _inputs()
    seg = SegmentationEngine(SegmentationPack(seeded_pyannet(seed=1234, num_layers=4).state_dict(),
                                              {"lstm": {"num_layers": 4}}, 7, 3, 2, gpu_device))
    emb = EmbeddingEngine(EmbeddingPack(seeded_wespeaker(seed=4321).state_dict(), gpu_device))
    logp = seg.forward(wav.to(gpu_device)).cpu().numpy()[:, ::19]
    e = emb.forward(wav[:, :, :48000].to(gpu_device), weights=masks.to(gpu_device)).cpu().numpy()
    assert np.allclose(logp, gold["seg_logp"], rtol=1e-4, atol=1e-5)
    assert np.abs(e - gold["embeddings"]).max() <= 1e-4 * np.abs(gold["embeddings"]).max()
