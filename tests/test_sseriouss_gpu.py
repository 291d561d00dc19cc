"""GPU parity of SSeRiouSS (models/segmentation/SSeRiouSS.py:42-328): the wav2vec
2.0 / WavLM encoder (strided convolutions as MFMA GEMMs, GroupNorm / LayerNorm, grouped positional convolution,
attention with WavLM's gated relative position bias, GELU MLPs), the layer mix, then the PyanNet LSTM / head
kernels -- against oracle.models.SSeRiouSS.  The reference's own SSeRiouSS.py is pinned bit for bit to that
oracle class by tests/test_reference_pipeline.py; the encoder underneath (torchaudio, absent offline) is the
restatement in oracle/wav2vec2.py, pinned layer by layer to HuggingFace transformers' Wav2Vec2Model / WavLMModel
by tests/test_oracle_wav2vec2_pin.py.  Tolerance: rtol 1e-4 / atol 1e-5 on log-probs."""
import os

import numpy as np
import pytest
import torch

from conftest import north_star_ratio

pytestmark = pytest.mark.gpu


def _engine(model, hparams, gpu_device, powerset=True):
    from pyannote_audio_amd.segmentation import SSeRiouSSEngine
    from pyannote_audio_amd.weights import SSeRiouSSPack
    args = (7, 3, 2) if powerset else (3, 3, None)
    return SSeRiouSSEngine(SSeRiouSSPack(model.state_dict(), hparams, *args, gpu_device))


def _hard_check(name, ml, ref):
    from oracle import Powerset
    top2 = ref.topk(2, dim=-1).values
    safe = (top2[..., 0] - top2[..., 1]) > 1e-4
    ref_ml = Powerset(3, 2)(ref).to(torch.uint8)
    mism = (ml.cpu() != ref_ml).any(dim=-1)
    with open("gpurun_out/parity.log", "a") as fp:
        fp.write(f"{name}: hard-decision mismatches {int(mism.sum())} of {mism.numel()} frames, "
                 f"{int((mism & safe).sum())} outside the 1e-4 top-2 gap\n")
    assert torch.equal(ml.cpu()[safe], ref_ml[safe])


def test_wavlm_base_sseriouss(gpu_device):
    """the default architecture: WAVLM_BASE (12 layers, 768 wide, 95 M parameters) + 4-layer bi-LSTM"""
    from oracle import seeded_sseriouss
    model = seeded_sseriouss(num_layers=4)
    eng = _engine(model, {"wav2vec": "WAVLM_BASE", "wav2vec_layer": -1, "lstm": {"num_layers": 4}}, gpu_device)
    assert eng.frames_of(160000) == 499 and eng.frames_of(399) == 0
    B, N, stride = 3, 160000, 16000
    total = stride * (B - 1) + N - 7000                      # the last chunk is zero padded by 7000 samples
    g = torch.Generator().manual_seed(11)
    wav = (0.1 * torch.randn(total, generator=g)).clamp(-1, 1)
    chunks = torch.zeros(B, 1, N)
    for b in range(B):
        seg = wav[b * stride: b * stride + N]
        chunks[b, 0, :seg.numel()] = seg
    with torch.inference_mode():
        ref = model(chunks)
    logp, ml = eng.forward_strided(wav.to(gpu_device), stride, B, N)
    assert logp.shape == (B, 499, 7) and ml.shape == (B, 499, 3)
    assert north_star_ratio("sseriouss_wavlm_base", logp, ref) <= 1.0
    _hard_check("sseriouss_wavlm_base", ml, ref)
    out = eng.forward(chunks[:1, :, :80000].to(gpu_device))
    with torch.inference_mode():
        assert north_star_ratio("sseriouss_wavlm_base_5s", out, model(chunks[:1, :, :80000])) <= 1.0


@pytest.mark.parametrize("layer", [-1, 2])
def test_wav2vec2_dict_config(gpu_device, layer):
    """an explicit wav2vec2_model configuration (SSeRiouSS.py:120-123): layer_norm feature extractor with
    convolution biases, pre-LN encoder, plain attention; all-layer mix and a single layer (wav2vec_layer)."""
    from oracle import seeded_sseriouss
    from oracle.models import TINY_WAV2VEC2
    model = seeded_sseriouss(wav2vec=dict(TINY_WAV2VEC2), num_layers=2, wav2vec_layer=layer)
    eng = _engine(model, {"wav2vec": dict(TINY_WAV2VEC2), "wav2vec_layer": layer, "lstm": {"num_layers": 2}},
                  gpu_device)
    g = torch.Generator().manual_seed(5)
    wav = (0.1 * torch.randn(19, 1, 16000, generator=g)).clamp(-1, 1)       # 19 chunks: two 16-chunk LSTM tiles
    with torch.inference_mode():
        ref = model(wav)
    got = eng.forward(wav.to(gpu_device))
    assert got.shape == ref.shape == (19, eng.frames_of(16000), 7)
    assert north_star_ratio(f"sseriouss_tiny_layer{layer}", got, ref) <= 1.0


def test_sseriouss_model_and_inference(gpu_device, tmp_path):
    """the product `SSeRiouSS` model class from a reference-format checkpoint, and `Inference` sliding over a
    file with it (10 s / 1 s chunks, 499 frames of 20 ms per chunk) vs the oracle's slide."""
    import pyannote_audio_amd as pa
    from oracle import pipeline as op
    from oracle import seeded_sseriouss
    from pyannote_audio_amd.model import save_checkpoint, segmentation_specifications
    model = seeded_sseriouss(num_layers=2)
    hparams = {"wav2vec": "WAVLM_BASE", "wav2vec_frozen": False, "wav2vec_layer": -1,
               "lstm": {"hidden_size": 128, "num_layers": 2, "bidirectional": True, "monolithic": True, "dropout": 0.0},
               "linear": {"hidden_size": 128, "num_layers": 2}, "sample_rate": 16000, "num_channels": 1}
    path = os.path.join(str(tmp_path), "pytorch_model.bin")
    save_checkpoint(path, model.state_dict(), hparams, pa.SSeRiouSS.ARCHITECTURE, segmentation_specifications(10.0))
    product = pa.Model.from_pretrained(path).to(gpu_device)
    assert isinstance(product, pa.SSeRiouSS) and product.num_frames(160000) == 499
    rf = product.receptive_field
    assert (round(rf.duration * 16000), round(rf.step * 16000)) == (400, 320)
    g = torch.Generator().manual_seed(2)
    wav = (0.1 * torch.randn(1, 16000 * 12 + 3333, generator=g)).clamp(-1, 1)
    inference = pa.Inference(product, skip_aggregation=True, batch_size=4)
    swf = inference({"waveform": wav, "sample_rate": 16000})
    want = op.slide(model, wav, 16000, 10.0, 1.0, 4)
    assert swf.data.shape == want.shape == (4, 499, 3)
    chunks = torch.stack([torch.nn.functional.pad(wav[0, c * 16000: c * 16000 + 160000],
                                                  (0, max(0, 160000 - (wav.shape[1] - c * 16000))))
                          for c in range(4)])[:, None]
    with torch.inference_mode():
        ref = model(chunks)
    top2 = ref.topk(2, dim=-1).values
    safe = ((top2[..., 0] - top2[..., 1]) > 1e-4).numpy()
    assert np.array_equal(swf.data[safe], want[safe])
