"""Oracle (CPU, torch fp32) restatement of the two neural models of the hot path.

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.

State-dict key layout is identical to the reference's so that the same checkpoints load
into both the oracle and the product.

Reference files restated (relative to /root/reference/src/pyannote/audio):
  models/blocks/sincnet.py:40-184          -> SincNet
  models/segmentation/PyanNet.py:38-240    -> PyanNet
  models/blocks/pooling.py:30-130          -> _pool / StatsPool
  models/embedding/wespeaker/resnet.py:37-145, 215-452 -> TSTP / BasicBlock / ResNet
  models/embedding/wespeaker/__init__.py:113-157, 324-372 -> compute_fbank / forward
  utils/powerset.py:37-140                 -> Powerset
Third-party algorithms restated from their published source (PARITY UNPINNED):
  asteroid_filterbanks 0.4.0 ParamSincFB / Encoder
  torchaudio 2.10.0 compliance.kaldi.fbank
"""

from __future__ import annotations

import math
from itertools import combinations
from typing import Optional

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


# --------------------------------------------------------------------------------------
# asteroid_filterbanks.ParamSincFB (restated; parity unpinned)
# --------------------------------------------------------------------------------------
class ParamSincFB(nn.Module):
    """80 learned band-pass FIRs: 40 'cos' (even) + 40 'sin' (odd) filters.

    Follows asteroid_filterbanks/param_sinc_fb.py (v0.4.0): mel-spaced init between 30 Hz and
    sr/2 - (min_low_hz + min_band_hz); low = min_low + |low_hz_|; high = clamp(low + min_band +
    |band_hz_|, min_low, sr/2); half Hamming window; filters / (2 * band).
    """

    def __init__(self, n_filters=80, kernel_size=251, stride=10, sample_rate=16000.0,
                 min_low_hz=50, min_band_hz=50):
        super().__init__()
        if kernel_size % 2 == 0:
            kernel_size += 1
        self.n_filters = n_filters
        self.kernel_size = kernel_size
        self.stride = stride
        self.sample_rate = float(sample_rate)
        self.min_low_hz, self.min_band_hz = min_low_hz, min_band_hz
        self.half_kernel = self.kernel_size // 2
        self.cutoff = self.n_filters // 2

        low_hz = 30
        high_hz = self.sample_rate / 2 - (self.min_low_hz + self.min_band_hz)
        mel = np.linspace(self.to_mel(low_hz), self.to_mel(high_hz), self.n_filters // 2 + 1,
                          dtype="float32")
        hz = self.to_hz(mel)
        self.low_hz_ = nn.Parameter(torch.from_numpy(hz[:-1]).view(-1, 1))
        self.band_hz_ = nn.Parameter(torch.from_numpy(np.diff(hz)).view(-1, 1))

        window_ = np.hamming(self.kernel_size)[: self.half_kernel]
        n_ = 2 * np.pi * (torch.arange(-self.half_kernel, 0.0).view(1, -1) / self.sample_rate)
        self.register_buffer("window_", torch.from_numpy(window_).float())
        self.register_buffer("n_", n_)

    @staticmethod
    def to_mel(hz):
        return 2595 * np.log10(1 + hz / 700)

    @staticmethod
    def to_hz(mel):
        return 700 * (10 ** (mel / 2595) - 1)

    def make_filters(self, low, high, filt_type="cos"):
        band = (high - low)[:, 0]
        ft_low = torch.matmul(low, self.n_)
        ft_high = torch.matmul(high, self.n_)
        if filt_type == "cos":
            bp_left = ((torch.sin(ft_high) - torch.sin(ft_low)) / (self.n_ / 2)) * self.window_
            bp_center = 2 * band.view(-1, 1)
            bp_right = torch.flip(bp_left, dims=[1])
        else:
            bp_left = ((torch.cos(ft_low) - torch.cos(ft_high)) / (self.n_ / 2)) * self.window_
            bp_center = torch.zeros_like(band.view(-1, 1))
            bp_right = -torch.flip(bp_left, dims=[1])
        band_pass = torch.cat([bp_left, bp_center, bp_right], dim=1)
        band_pass = band_pass / (2 * band[:, None])
        return band_pass.view(self.cutoff, 1, self.kernel_size)

    def filters(self):
        low = self.min_low_hz + torch.abs(self.low_hz_)
        high = torch.clamp(low + self.min_band_hz + torch.abs(self.band_hz_),
                           self.min_low_hz, self.sample_rate / 2)
        cos_filters = self.make_filters(low, high, filt_type="cos")
        sin_filters = self.make_filters(low, high, filt_type="sin")
        return torch.cat([cos_filters, sin_filters], dim=0)


class Encoder(nn.Module):
    """asteroid_filterbanks.Encoder restricted to the 3-D input case: plain strided conv1d."""

    def __init__(self, filterbank: ParamSincFB):
        super().__init__()
        self.filterbank = filterbank

    def forward(self, waveform):
        return F.conv1d(waveform, self.filterbank.filters(), stride=self.filterbank.stride,
                        padding=0)


# --------------------------------------------------------------------------------------
# SincNet / PyanNet  (models/blocks/sincnet.py:40-184, models/segmentation/PyanNet.py:38-240)
# --------------------------------------------------------------------------------------
class SincNet(nn.Module):
    def __init__(self, sample_rate=16000, stride=10):
        super().__init__()
        self.stride = stride
        self.wav_norm1d = nn.InstanceNorm1d(1, affine=True)
        self.conv1d = nn.ModuleList()
        self.pool1d = nn.ModuleList()
        self.norm1d = nn.ModuleList()
        self.conv1d.append(Encoder(ParamSincFB(80, 251, stride=stride, sample_rate=sample_rate,
                                               min_low_hz=50, min_band_hz=50)))
        self.pool1d.append(nn.MaxPool1d(3, stride=3, padding=0, dilation=1))
        self.norm1d.append(nn.InstanceNorm1d(80, affine=True))
        self.conv1d.append(nn.Conv1d(80, 60, 5, stride=1))
        self.pool1d.append(nn.MaxPool1d(3, stride=3, padding=0, dilation=1))
        self.norm1d.append(nn.InstanceNorm1d(60, affine=True))
        self.conv1d.append(nn.Conv1d(60, 60, 5, stride=1))
        self.pool1d.append(nn.MaxPool1d(3, stride=3, padding=0, dilation=1))
        self.norm1d.append(nn.InstanceNorm1d(60, affine=True))

    def num_frames(self, num_samples: int) -> int:
        n = num_samples
        for k, s in zip([251, 3, 5, 3, 5, 3], [self.stride, 3, 1, 3, 1, 3]):
            n = 1 + (n - (k - 1) - 1) // s
        return n

    def forward(self, waveforms):
        outputs = self.wav_norm1d(waveforms)
        for c, (conv1d, pool1d, norm1d) in enumerate(zip(self.conv1d, self.pool1d, self.norm1d)):
            outputs = conv1d(outputs)
            if c == 0:
                outputs = torch.abs(outputs)
            outputs = F.leaky_relu(norm1d(pool1d(outputs)))
        return outputs


class Powerset(nn.Module):
    """utils/powerset.py:37-140 (hard / soft conversion to multilabel)."""

    def __init__(self, num_classes: int, max_set_size: int):
        super().__init__()
        self.num_classes = num_classes
        self.max_set_size = max_set_size
        sets = []
        for set_size in range(0, max_set_size + 1):
            for current_set in combinations(range(num_classes), set_size):
                sets.append(current_set)
        self.num_powerset_classes = len(sets)
        mapping = torch.zeros(len(sets), num_classes)
        for k, current_set in enumerate(sets):
            mapping[k, current_set] = 1
        self.register_buffer("mapping", mapping, persistent=False)

    def to_multilabel(self, powerset, soft=False):
        if soft:
            probs = torch.exp(powerset)
        else:
            probs = F.one_hot(torch.argmax(powerset, dim=-1), self.num_powerset_classes).float()
        return torch.matmul(probs, self.mapping)

    def to_powerset(self, multilabel):
        return F.one_hot(torch.argmax(torch.matmul(multilabel, self.mapping.T), dim=-1),
                         self.num_powerset_classes)

    forward = to_multilabel


class PyanNet(nn.Module):
    """SincNet > LSTM > feed-forward > classifier > log-softmax  (PyanNet.py:211-240)."""

    def __init__(self, num_classes: int = 7, sincnet: Optional[dict] = None,
                 lstm: Optional[dict] = None, linear: Optional[dict] = None,
                 sample_rate: int = 16000, powerset: bool = True):
        super().__init__()
        self.powerset = powerset
        self.hp_sincnet = {"stride": 10, **(sincnet or {})}
        self.hp_lstm = {"hidden_size": 128, "num_layers": 2, "bidirectional": True,
                        "monolithic": True, "dropout": 0.0, **(lstm or {})}
        self.hp_linear = {"hidden_size": 128, "num_layers": 2, **(linear or {})}
        self.sincnet = SincNet(sample_rate=sample_rate, stride=self.hp_sincnet["stride"])
        H, L = self.hp_lstm["hidden_size"], self.hp_lstm["num_layers"]
        bi = self.hp_lstm["bidirectional"]
        if self.hp_lstm["monolithic"]:
            self.lstm = nn.LSTM(60, hidden_size=H, num_layers=L, bidirectional=bi,
                                dropout=self.hp_lstm["dropout"], batch_first=True)
        else:
            self.lstm = nn.ModuleList([
                nn.LSTM(60 if i == 0 else H * (2 if bi else 1), hidden_size=H, num_layers=1,
                        bidirectional=bi, batch_first=True) for i in range(L)])
        out = H * (2 if bi else 1)
        dims = [out] + [self.hp_linear["hidden_size"]] * self.hp_linear["num_layers"]
        if self.hp_linear["num_layers"] > 0:
            self.linear = nn.ModuleList([nn.Linear(a, b) for a, b in zip(dims[:-1], dims[1:])])
        self.classifier = nn.Linear(dims[-1], num_classes)
        # default_activation (core/model.py:271-299): log-softmax for the (mono-label) powerset problem,
        # sigmoid for multi-label checkpoints (one score per speaker)
        self.activation = nn.LogSoftmax(dim=-1) if powerset else nn.Sigmoid()

    def num_frames(self, num_samples: int) -> int:
        return self.sincnet.num_frames(num_samples)

    def forward(self, waveforms):
        outputs = self.sincnet(waveforms)
        outputs = outputs.transpose(1, 2)  # batch feature frame -> batch frame feature
        if self.hp_lstm["monolithic"]:
            outputs, _ = self.lstm(outputs)
        else:
            for lstm in self.lstm:
                outputs, _ = lstm(outputs)
        if self.hp_linear["num_layers"] > 0:
            for linear in self.linear:
                outputs = F.leaky_relu(linear(outputs))
        return self.activation(self.classifier(outputs))


# --------------------------------------------------------------------------------------
# torchaudio.compliance.kaldi.fbank (restated; parity unpinned)
# --------------------------------------------------------------------------------------
def _mel_scale(freq):
    return 1127.0 * (1.0 + freq / 700.0).log()


def _mel_scale_scalar(freq: float) -> float:
    return 1127.0 * math.log(1.0 + freq / 700.0)


def kaldi_mel_banks(num_bins: int, window_length_padded: int, sample_freq: float,
                    low_freq: float = 20.0, high_freq: float = 0.0) -> torch.Tensor:
    """get_mel_banks(...) with vtln_warp == 1.0 -> (num_bins, padded // 2) fp32."""
    num_fft_bins = window_length_padded / 2
    nyquist = 0.5 * sample_freq
    if high_freq <= 0.0:
        high_freq += nyquist
    fft_bin_width = sample_freq / window_length_padded
    mel_low_freq = _mel_scale_scalar(low_freq)
    mel_high_freq = _mel_scale_scalar(high_freq)
    mel_freq_delta = (mel_high_freq - mel_low_freq) / (num_bins + 1)
    bin = torch.arange(num_bins).unsqueeze(1)
    left_mel = mel_low_freq + bin * mel_freq_delta
    center_mel = mel_low_freq + (bin + 1.0) * mel_freq_delta
    right_mel = mel_low_freq + (bin + 2.0) * mel_freq_delta
    mel = _mel_scale(fft_bin_width * torch.arange(num_fft_bins)).unsqueeze(0)
    up_slope = (mel - left_mel) / (center_mel - left_mel)
    down_slope = (right_mel - mel) / (right_mel - center_mel)
    return torch.max(torch.zeros(1), torch.min(up_slope, down_slope))


def kaldi_fbank(waveform: torch.Tensor, num_mel_bins: int = 80, frame_length: float = 25.0,
                frame_shift: float = 10.0, sample_frequency: float = 16000.0,
                preemphasis_coefficient: float = 0.97, low_freq: float = 20.0,
                high_freq: float = 0.0) -> torch.Tensor:
    """kaldi.fbank(waveform (1, n)) with the WeSpeaker settings: snip_edges, dither 0,
    remove_dc_offset, hamming window, round_to_power_of_two, use_power, use_log_fbank,
    use_energy False  ->  (m, num_mel_bins)."""
    waveform = waveform[0]
    window_shift = int(sample_frequency * frame_shift * 0.001)
    window_size = int(sample_frequency * frame_length * 0.001)
    padded_window_size = 1 << (window_size - 1).bit_length()
    epsilon = torch.tensor(torch.finfo(waveform.dtype).eps, dtype=waveform.dtype)
    num_samples = waveform.size(0)
    if num_samples < window_size:
        return torch.empty((0, num_mel_bins), dtype=waveform.dtype)
    m = 1 + (num_samples - window_size) // window_shift
    strided = waveform.as_strided((m, window_size), (window_shift, 1))
    # remove dc offset
    strided = strided - torch.mean(strided, dim=1).unsqueeze(1)
    # pre-emphasis: x[j] -= coeff * x[max(0, j-1)]
    offset = F.pad(strided.unsqueeze(0), (1, 0), mode="replicate").squeeze(0)
    strided = strided - preemphasis_coefficient * offset[:, :-1]
    window = torch.hamming_window(window_size, periodic=False, alpha=0.54, beta=0.46,
                                  dtype=waveform.dtype).unsqueeze(0)
    strided = strided * window
    strided = F.pad(strided.unsqueeze(0), (0, padded_window_size - window_size),
                    mode="constant", value=0).squeeze(0)
    spectrum = torch.fft.rfft(strided).abs().pow(2.0)
    mel = kaldi_mel_banks(num_mel_bins, padded_window_size, sample_frequency, low_freq,
                          high_freq).to(waveform.dtype)
    mel = F.pad(mel, (0, 1), mode="constant", value=0)
    mel_energies = torch.mm(spectrum, mel.T)
    return torch.max(mel_energies, epsilon).log()


# --------------------------------------------------------------------------------------
# StatsPool / TSTP / ResNet34 (pooling.py, wespeaker/resnet.py)
# --------------------------------------------------------------------------------------
def _pool(sequences: torch.Tensor, weights: torch.Tensor) -> torch.Tensor:
    weights = weights.unsqueeze(dim=1)
    v1 = weights.sum(dim=2) + 1e-8
    mean = torch.sum(sequences * weights, dim=2) / v1
    dx2 = torch.square(sequences - mean.unsqueeze(2))
    v2 = torch.square(weights).sum(dim=2)
    var = torch.sum(dx2 * weights, dim=2) / (v1 - v2 / v1 + 1e-8)
    std = torch.sqrt(var)
    return torch.cat([mean, std], dim=1)


class StatsPool(nn.Module):
    def forward(self, sequences, weights=None):
        if weights is None:
            mean = sequences.mean(dim=-1)
            std = sequences.std(dim=-1, correction=1)
            return torch.cat([mean, std], dim=-1)
        if weights.dim() == 2:
            has_speaker_dimension = False
            weights = weights.unsqueeze(dim=1)
        else:
            has_speaker_dimension = True
        _, _, num_frames = sequences.size()
        _, num_speakers, num_weights = weights.size()
        if num_frames != num_weights:
            weights = F.interpolate(weights, size=num_frames, mode="nearest")
        output = torch.stack([_pool(sequences, weights[:, s, :]) for s in range(num_speakers)],
                             dim=1)
        if not has_speaker_dimension:
            return output.squeeze(dim=1)
        return output


class TSTP(nn.Module):
    def __init__(self):
        super().__init__()
        self.stats_pool = StatsPool()

    def forward(self, features, weights=None):
        b, d, c, t = features.shape
        return self.stats_pool(features.reshape(b, d * c, t), weights=weights)


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, in_planes, planes, stride=1):
        super().__init__()
        self.conv1 = nn.Conv2d(in_planes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=1, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.shortcut = nn.Sequential()
        if stride != 1 or in_planes != planes:
            self.shortcut = nn.Sequential(
                nn.Conv2d(in_planes, planes, 1, stride=stride, bias=False),
                nn.BatchNorm2d(planes))

    def forward(self, x):
        out = F.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        out = out + self.shortcut(x)
        return F.relu(out)


class Bottleneck(nn.Module):
    """wespeaker/resnet.py:148-212 (ResNet50/101/152/221/293)"""
    expansion = 4

    def __init__(self, in_planes, planes, stride=1):
        super().__init__()
        self.conv1 = nn.Conv2d(in_planes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, self.expansion * planes, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(self.expansion * planes)
        self.shortcut = nn.Sequential()
        if stride != 1 or in_planes != self.expansion * planes:
            self.shortcut = nn.Sequential(
                nn.Conv2d(in_planes, self.expansion * planes, 1, stride=stride, bias=False),
                nn.BatchNorm2d(self.expansion * planes))

    def forward(self, x):
        out = F.relu(self.bn1(self.conv1(x)))
        out = F.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        out = out + self.shortcut(x)
        return F.relu(out)


class ResNet(nn.Module):
    def __init__(self, num_blocks=(3, 4, 6, 3), m_channels=32, feat_dim=80, embed_dim=256, block=None):
        super().__init__()
        self.block = block or BasicBlock
        self.in_planes = m_channels
        self.stats_dim = int(feat_dim / 8) * m_channels * 8
        self.conv1 = nn.Conv2d(1, m_channels, 3, stride=1, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(m_channels)
        self.layer1 = self._make_layer(m_channels, num_blocks[0], 1)
        self.layer2 = self._make_layer(m_channels * 2, num_blocks[1], 2)
        self.layer3 = self._make_layer(m_channels * 4, num_blocks[2], 2)
        self.layer4 = self._make_layer(m_channels * 8, num_blocks[3], 2)
        self.pool = TSTP()
        self.seg_1 = nn.Linear(self.stats_dim * self.block.expansion * 2, embed_dim)

    def _make_layer(self, planes, n, stride):
        layers = []
        for s in [stride] + [1] * (n - 1):
            layers.append(self.block(self.in_planes, planes, s))
            self.in_planes = planes * self.block.expansion
        return nn.Sequential(*layers)

    def forward_frames(self, fbank):
        x = fbank.permute(0, 2, 1).unsqueeze(1)  # (B,T,F) -> (B,1,F,T)
        out = F.relu(self.bn1(self.conv1(x)))
        out = self.layer1(out)
        out = self.layer2(out)
        out = self.layer3(out)
        out = self.layer4(out)
        return out

    def forward(self, fbank, weights=None):
        out = self.forward_frames(fbank)
        stats = self.pool(out, weights=weights)
        return self.seg_1(stats)  # embed_a (two_emb_layer=False)


class WeSpeakerResNet34(nn.Module):
    """fbank -> ResNet34 -> TSTP -> Linear(5120, 256)  (wespeaker/__init__.py:324-372)."""

    def __init__(self, sample_rate=16000, num_mel_bins=80, frame_length=25, frame_shift=10,
                 num_blocks=(3, 4, 6, 3), block=None, fbank_centering_span=None):
        super().__init__()
        self.sample_rate = sample_rate
        self.num_mel_bins = num_mel_bins
        self.frame_length = frame_length
        self.frame_shift = frame_shift
        self.fbank_centering_span = fbank_centering_span     # seconds, or None (wespeaker/__init__.py:56-71)
        # block=Bottleneck with (3, 8, 36, 3) / (6, 16, 48, 3) / (10, 20, 64, 3) = WeSpeakerResNet152 /
        # 221 / 293 (wespeaker/__init__.py:375-470, resnet.py:477-507)
        self.resnet = ResNet(num_blocks, 32, num_mel_bins, 256, block=block)

    def compute_fbank(self, waveforms):
        waveforms = waveforms * (1 << 15)
        feats = torch.stack([
            kaldi_fbank(w, num_mel_bins=self.num_mel_bins, frame_length=self.frame_length,
                        frame_shift=self.frame_shift, sample_frequency=self.sample_rate)
            for w in waveforms])
        if self.fbank_centering_span is None:                        # wespeaker/__init__.py:137-139
            return feats - torch.mean(feats, dim=1, keepdim=True)
        # running average over `fbank_centering_span` seconds (wespeaker/__init__.py:141-157): the span in frames
        # (conv1d_num_frames of the fbank framing, utils/receptive_field.py:26-53), made odd, as an average pooling
        # that does not count the padding
        window_size = int(self.sample_rate * self.frame_length * 0.001)
        step_size = int(self.sample_rate * self.frame_shift * 0.001)
        kernel_size = 1 + (int(self.fbank_centering_span * self.sample_rate) - (window_size - 1) - 1) // step_size
        return feats - F.avg_pool1d(feats.transpose(1, 2), kernel_size=2 * (kernel_size // 2) + 1, stride=1,
                                    padding=kernel_size // 2, count_include_pad=False).transpose(1, 2)

    def forward(self, waveforms, weights=None):
        return self.resnet(self.compute_fbank(waveforms), weights=weights)


class SSeRiouSS(nn.Module):
    """wav2vec > LSTM > feed-forward > classifier (models/segmentation/SSeRiouSS.py:42-328; pinned bit for bit
    to that class -- both on oracle/wav2vec2.py, the unpinned restatement of torchaudio's encoder -- by
    tests/test_reference_pipeline.py).  `wav2vec`: a bundle name ("WAVLM_BASE") or the keyword arguments of
    torchaudio.models.wav2vec2_model."""

    def __init__(self, num_classes: int = 7, wav2vec="WAVLM_BASE", wav2vec_layer: int = -1,
                 lstm: Optional[dict] = None, linear: Optional[dict] = None, powerset: bool = True):
        super().__init__()
        from . import wav2vec2 as w2v
        self.powerset = powerset
        self.wav2vec_layer = wav2vec_layer
        if isinstance(wav2vec, str):
            bundle = w2v.PIPELINES[wav2vec]
            dim, layers = bundle._params["encoder_embed_dim"], bundle._params["encoder_num_layers"]
            self.wav2vec = bundle.get_model()
        else:
            self.wav2vec = w2v.wav2vec2_model(**wav2vec)
            dim, layers = wav2vec["encoder_embed_dim"], wav2vec["encoder_num_layers"]
        if wav2vec_layer < 0:
            self.wav2vec_weights = nn.Parameter(torch.ones(layers))
        self.hp_lstm = {"hidden_size": 128, "num_layers": 4, "bidirectional": True, "monolithic": True,
                        "dropout": 0.0, **(lstm or {})}
        self.hp_linear = {"hidden_size": 128, "num_layers": 2, **(linear or {})}
        H, L, bi = self.hp_lstm["hidden_size"], self.hp_lstm["num_layers"], self.hp_lstm["bidirectional"]
        if self.hp_lstm["monolithic"]:
            self.lstm = nn.LSTM(dim, hidden_size=H, num_layers=L, bidirectional=bi, batch_first=True,
                                dropout=self.hp_lstm["dropout"])
        else:
            self.lstm = nn.ModuleList([nn.LSTM(dim if i == 0 else H * (2 if bi else 1), hidden_size=H,
                                               num_layers=1, bidirectional=bi, batch_first=True)
                                       for i in range(L)])
        dims = [H * (2 if bi else 1)] + [self.hp_linear["hidden_size"]] * self.hp_linear["num_layers"]
        if self.hp_linear["num_layers"] > 0:
            self.linear = nn.ModuleList([nn.Linear(a, b) for a, b in zip(dims[:-1], dims[1:])])
        self.classifier = nn.Linear(dims[-1], num_classes)
        self.activation = nn.LogSoftmax(dim=-1) if powerset else nn.Sigmoid()

    def forward(self, waveforms):
        num_layers = None if self.wav2vec_layer < 0 else self.wav2vec_layer
        outputs, _ = self.wav2vec.extract_features(waveforms.squeeze(1), num_layers=num_layers)
        if num_layers is None:
            outputs = torch.stack(outputs, dim=-1) @ F.softmax(self.wav2vec_weights, dim=0)
        else:
            outputs = outputs[-1]
        if self.hp_lstm["monolithic"]:
            outputs, _ = self.lstm(outputs)
        else:
            for lstm in self.lstm:
                outputs, _ = lstm(outputs)
        if self.hp_linear["num_layers"] > 0:
            for linear in self.linear:
                outputs = F.leaky_relu(linear(outputs))
        return self.activation(self.classifier(outputs))


#: a small wav2vec 2.0 configuration for tests (layer_norm extractor, pre-LN encoder: the "large" recipe)
TINY_WAV2VEC2 = dict(
    extractor_mode="layer_norm", extractor_conv_layer_config=[(64, 10, 5), (64, 3, 2), (64, 3, 2), (64, 2, 2)],
    extractor_conv_bias=True, encoder_embed_dim=128, encoder_projection_dropout=0.0, encoder_pos_conv_kernel=32,
    encoder_pos_conv_groups=4, encoder_num_layers=3, encoder_num_heads=4, encoder_attention_dropout=0.0,
    encoder_ff_interm_features=256, encoder_ff_interm_dropout=0.0, encoder_dropout=0.0,
    encoder_layer_norm_first=True, encoder_layer_drop=0.0, aux_num_out=None)


def seeded_sseriouss(seed: int = 1357, wav2vec="WAVLM_BASE", num_layers: int = 2, wav2vec_layer: int = -1,
                     classifier_gain: float = 8.0) -> SSeRiouSS:
    """Default-initialised SSeRiouSS; norms, layer weights, the gate constants and the relative position
    table are perturbed so that every parameter takes part."""
    g = torch.Generator().manual_seed(seed)
    torch.manual_seed(seed)
    model = SSeRiouSS(wav2vec=wav2vec, lstm={"num_layers": num_layers}, wav2vec_layer=wav2vec_layer)
    with torch.no_grad():
        for m in model.modules():
            if isinstance(m, (nn.LayerNorm, nn.GroupNorm)):
                m.weight.copy_(1.0 + 0.1 * torch.randn(m.weight.shape, generator=g))
                m.bias.copy_(0.1 * torch.randn(m.bias.shape, generator=g))
        for name, p in model.named_parameters():
            if name.endswith("gru_rel_pos_const"):
                p.copy_(1.0 + 0.2 * torch.randn(p.shape, generator=g))
            if name.endswith("in_proj_bias") or name.endswith("out_proj.bias"):
                p.copy_(0.05 * torch.randn(p.shape, generator=g))
        if wav2vec_layer < 0:
            model.wav2vec_weights.copy_(torch.randn(model.wav2vec_weights.shape, generator=g))
        model.classifier.weight.mul_(classifier_gain)
    return model.eval()


class XVectorSincNet(nn.Module):
    """SincNet -> 5 x (Conv1d + LeakyReLU + BatchNorm1d) -> StatsPool -> Linear
    (models/embedding/xvector.py:205-349; pinned bit for bit to that class by
    tests/test_reference_pipeline.py)."""

    def __init__(self, sample_rate: int = 16000, dimension: int = 512, stride: int = 10):
        super().__init__()
        self.sincnet = SincNet(sample_rate=sample_rate, stride=stride)
        self.tdnns = nn.ModuleList()
        in_channel = 60
        for out_channel, kernel_size, dilation in zip([512, 512, 512, 512, 1500], [5, 3, 3, 1, 1],
                                                      [1, 2, 3, 1, 1]):
            self.tdnns.extend([nn.Conv1d(in_channel, out_channel, kernel_size, dilation=dilation),
                               nn.LeakyReLU(), nn.BatchNorm1d(out_channel)])
            in_channel = out_channel
        self.stats_pool = StatsPool()
        self.embedding = nn.Linear(in_channel * 2, dimension)

    def forward(self, waveforms, weights=None):
        outputs = self.sincnet(waveforms).squeeze(dim=1)
        for tdnn in self.tdnns:
            outputs = tdnn(outputs)
        return self.embedding(self.stats_pool(outputs, weights=weights))


def seeded_xvector(seed: int = 2468, dimension: int = 512) -> XVectorSincNet:
    """Default-initialised XVectorSincNet with randomised BatchNorm statistics / affine and SincNet norms."""
    g = torch.Generator().manual_seed(seed)
    torch.manual_seed(seed)
    model = XVectorSincNet(dimension=dimension)
    with torch.no_grad():
        sn = model.sincnet
        sn.wav_norm1d.weight.copy_(1.0 + 0.1 * torch.randn(1, generator=g))
        sn.wav_norm1d.bias.copy_(0.1 * torch.randn(1, generator=g))
        for n in sn.norm1d:
            n.weight.copy_(1.0 + 0.2 * torch.randn(n.weight.shape, generator=g))
            n.bias.copy_(0.2 * torch.randn(n.bias.shape, generator=g))
        for m in model.modules():
            if isinstance(m, nn.BatchNorm1d):
                m.running_mean.copy_(0.1 * torch.randn(m.running_mean.shape, generator=g))
                m.running_var.copy_(0.5 + torch.rand(m.running_var.shape, generator=g))
                m.weight.copy_(1.0 + 0.1 * torch.randn(m.weight.shape, generator=g))
                m.bias.copy_(0.1 * torch.randn(m.bias.shape, generator=g))
    return model.eval()


# --------------------------------------------------------------------------------------
# seeded synthetic weights (no pretrained checkpoints are available)
# --------------------------------------------------------------------------------------
def seeded_pyannet(seed: int = 1234, num_layers: int = 4, classifier_gain: float = 8.0,
                   **kw) -> PyanNet:
    """Default-initialised PyanNet (reference layout).  Instance-norm affine parameters and
    the classifier are perturbed so that every code path (gamma/beta, bias) is exercised and
    the arg-max class actually varies over time with random weights."""
    g = torch.Generator().manual_seed(seed)
    torch.manual_seed(seed)
    model = PyanNet(lstm={"num_layers": num_layers}, **kw)
    with torch.no_grad():
        sn = model.sincnet
        sn.wav_norm1d.weight.copy_(1.0 + 0.1 * torch.randn(1, generator=g))
        sn.wav_norm1d.bias.copy_(0.1 * torch.randn(1, generator=g))
        for n in sn.norm1d:
            n.weight.copy_(1.0 + 0.2 * torch.randn(n.weight.shape, generator=g))
            n.bias.copy_(0.2 * torch.randn(n.bias.shape, generator=g))
        model.classifier.weight.mul_(classifier_gain)
    return model.eval()


def seeded_wespeaker(seed: int = 4321) -> WeSpeakerResNet34:
    """Default-initialised ResNet34 with randomised BatchNorm statistics/affine so that BN
    folding is exercised."""
    g = torch.Generator().manual_seed(seed)
    torch.manual_seed(seed)
    model = WeSpeakerResNet34()
    with torch.no_grad():
        for m in model.modules():
            if isinstance(m, nn.BatchNorm2d):
                m.running_mean.copy_(0.1 * torch.randn(m.running_mean.shape, generator=g))
                m.running_var.copy_(0.5 + torch.rand(m.running_var.shape, generator=g))
                m.weight.copy_(1.0 + 0.1 * torch.randn(m.weight.shape, generator=g))
                m.bias.copy_(0.1 * torch.randn(m.bias.shape, generator=g))
    return model.eval()
