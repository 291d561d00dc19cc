"""GPU: `pa_resample_poly` through the audio front door == the oracle's torchaudio restatement
(core/io.py:258-262) for the rates real files come in; then a 48 kHz stereo file through the pipeline."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("orig,n", [(8000, 40000), (44100, 3 * 44100 + 17), (48000, 5 * 48000 + 1),
                                    (22050, 30000), (32000, 64000)])
def test_resample_matches_oracle(gpu_device, orig, n):
    from oracle.audio import resample
    from pyannote_audio_amd.audio import Audio
    x = 0.3 * torch.randn(2, n, generator=torch.Generator().manual_seed(orig))
    audio = Audio(sample_rate=16000, mono=None, device=gpu_device)
    got, sr = audio({"waveform": x, "sample_rate": orig})
    want = resample(x, orig, 16000)
    assert sr == 16000 and tuple(got.shape) == tuple(want.shape)
    err = (got.cpu() - want).abs().max().item()
    with open("gpurun_out/parity.log", "a") as fp:
        fp.write(f"resample {orig}->16000: max|d| = {err:.2e} (signal max {want.abs().max().item():.2f})\n")
    assert torch.allclose(got.cpu(), want, rtol=1e-5, atol=2e-6)


def test_pipeline_accepts_48k_stereo(pipeline_dir, synthetic_models, gpu_device):
    """48 kHz stereo in -> downmix + GPU resample -> identical to feeding the oracle-resampled mono"""
    import pyannote_audio_amd as pa
    from oracle.audio import resample
    from oracle.synthetic import synth_conversation
    wav16, _ = synth_conversation(21.0, seed=9)
    # a 48 kHz stereo rendition: upsample with the oracle, two slightly different channels
    up = resample(wav16, 16000, 48000)
    stereo = torch.cat([up * 0.9, up * 1.1], dim=0)
    pipeline = pa.Pipeline.from_pretrained(pipeline_dir).to(gpu_device)
    out = pipeline({"waveform": stereo, "sample_rate": 48000, "uri": "st"})
    mono16 = resample(stereo.mean(dim=0, keepdim=True), 48000, 16000)
    ref = pipeline({"waveform": mono16, "sample_rate": 16000, "uri": "st"})
    a = [(round(s.start, 3), round(s.end, 3), l) for s, _, l in out.speaker_diarization.itertracks(yield_label=True)]
    b = [(round(s.start, 3), round(s.end, 3), l) for s, _, l in ref.speaker_diarization.itertracks(yield_label=True)]
    assert a == b and len(a) > 0
