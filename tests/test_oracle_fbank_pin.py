"""Pin the oracle's restatement of `torchaudio.compliance.kaldi.fbank` (oracle/models.py, used at
models/embedding/wespeaker/__init__.py:88-99,135) against an INDEPENDENT Kaldi-compatible
implementation that is installed in this image: `transformers.audio_utils` (spectrogram with
`remove_dc_offset`, `preemphasis`, povey/hamming windows and `mel_scale="kaldi"`,
`triangularize_in_mel_space=True`).  torchaudio itself is not installed.

What is compared and why the tolerances are what they are:
* the mel filter bank: torchaudio builds it in float32 (mel(f) ~ 2840 at 8 kHz, so the triangle slopes
  carry ~1e-5 of rounding), transformers in float64 -> agreement to 2e-5 abs on weights in [0, 1];
* the fbank itself, oracle run in float64 on the same samples vs transformers (float64): only the
  float32 filter bank separates them -> 3e-4 abs on log energies up to 26 (1e-5 relative);
* the float32 oracle (what the kernels are checked against): in the ENERGY domain, relative to the
  larger of the value and 1e-3 x the peak, <= 1e-4 (a log-domain bound would be dominated by float32
  cancellation in bins whose energy is 1e-4 of the peak -- 5e-3 observed on one bin of the 10 s case).
Inputs: seeded 3 s / 10 s noise and the reference's own 30 s fixture
`sample.wav` (src/pyannote/audio/sample/, copied to tests/golden/)."""
import os

import numpy as np
import pytest
import torch

from oracle.models import kaldi_fbank, kaldi_mel_banks

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
audio_utils = pytest.importorskip("transformers.audio_utils")


def _independent_fbank(wave64: np.ndarray) -> np.ndarray:
    mel = audio_utils.mel_filter_bank(num_frequency_bins=257, num_mel_filters=80, min_frequency=20,
                                      max_frequency=8000, sampling_rate=16000, norm=None,
                                      mel_scale="kaldi", triangularize_in_mel_space=True)
    win = audio_utils.window_function(400, "hamming", periodic=False)
    return audio_utils.spectrogram(wave64, win, frame_length=400, hop_length=160, fft_length=512,
                                   power=2.0, center=False, preemphasis=0.97, mel_filters=mel,
                                   log_mel="log", mel_floor=1.192092955078125e-07,
                                   remove_dc_offset=True).T


def _inputs():
    g = torch.Generator().manual_seed(0)
    yield "seeded-3s", (0.1 * torch.randn(48000, generator=g)).clamp(-1, 1)
    yield "seeded-10s", (0.1 * torch.randn(160000, generator=g)).clamp(-1, 1)
    from scipy.io import wavfile
    sr, x = wavfile.read(os.path.join(GOLDEN, "sample.wav"))
    assert sr == 16000 and x.dtype == np.int16 and x.shape == (480000,)
    yield "sample.wav", torch.from_numpy(x.astype(np.float32) / 32768.0)


def test_mel_banks_match_independent_implementation():
    mel = audio_utils.mel_filter_bank(num_frequency_bins=257, num_mel_filters=80, min_frequency=20,
                                      max_frequency=8000, sampling_rate=16000, norm=None,
                                      mel_scale="kaldi", triangularize_in_mel_space=True)
    ours = kaldi_mel_banks(80, 512, 16000.0).numpy()          # (80, 256): no Nyquist column
    assert mel.shape == (257, 80) and ours.shape == (80, 256)
    assert np.abs(mel[:256].T - ours).max() < 2e-5
    assert np.abs(mel[256]).max() == 0.0                      # the column kaldi pads with zeros


@pytest.mark.parametrize("name,wave", list(_inputs()), ids=[n for n, _ in _inputs()])
def test_oracle_fbank_pinned_by_transformers(name, wave):
    scaled = wave * 32768.0                                    # wespeaker/__init__.py:128
    want = _independent_fbank(scaled.numpy().astype(np.float64))
    got64 = kaldi_fbank(scaled.double().unsqueeze(0)).numpy()
    got32 = kaldi_fbank(scaled.unsqueeze(0)).numpy()
    assert got64.shape == want.shape == (1 + (wave.numel() - 400) // 160, 80)
    assert np.abs(got64 - want).max() < 3e-4, name
    e_want, e_got = np.exp(want), np.exp(got32.astype(np.float64))
    rel = np.abs(e_got - e_want) / np.maximum(e_want, 1e-3 * e_want.max())
    assert rel.max() < 1e-4, name
