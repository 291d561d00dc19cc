"""The audio front door against the behaviours the reference's own
tests/io_test.py checks: channel selection, in-memory waveforms, fixed-size crops, paths and binary
file objects (rewound after reading, core/io.py:348-349).  The reference decodes with torchcodec; here
the fixture is a PCM WAV written by scipy (the decoder this build has).  Resampling runs on the GPU only
(tests/test_resample_gpu.py): without one it must fail loudly, never fall back to a CPU resampler."""
import io

import numpy as np
import pytest
import torch
from scipy.io import wavfile

from pyannote_audio_amd.audio import Audio
from pyannote_audio_amd.core import Segment


@pytest.fixture
def wav_file(tmp_path):
    sr = 16000
    rng = np.random.default_rng(5)
    data = (rng.uniform(-0.5, 0.5, size=(2 * sr, 2)) * 32767).astype(np.int16)   # 2 s, stereo
    path = tmp_path / "dev00.wav"
    wavfile.write(path, sr, data)
    return path, sr, data


def test_basic_load_with_defaults(wav_file):                      # io_test.py:21-25
    path, sr, data = wav_file
    wav, rate = Audio(mono="downmix")(path)
    assert isinstance(wav, torch.Tensor) and rate == sr and wav.shape == (1, data.shape[0])
    expected = (data.astype(np.float32) / 32768.0).mean(axis=1)
    assert torch.allclose(wav[0], torch.from_numpy(expected), atol=1e-7)


def test_correct_audio_channel():                                  # io_test.py:28-35
    waveform = torch.rand(2, 16000 * 2)
    wav, sr = Audio(mono="downmix")({"waveform": waveform, "sample_rate": 16000, "channel": 1})
    assert torch.equal(wav, waveform[1:2]) and sr == 16000


def test_can_load_with_waveform():                                 # io_test.py:38-45
    waveform = torch.rand(2, 16000 * 2)
    wav, sr = Audio(mono="downmix")({"waveform": waveform, "sample_rate": 16000})
    assert isinstance(wav, torch.Tensor) and sr == 16000 and wav.shape == (1, 32000)
    assert torch.allclose(wav, waveform.mean(dim=0, keepdim=True))


def test_can_crop(wav_file):                                       # io_test.py:48-54
    path, _, _ = wav_file
    wav, sr = Audio(mono="downmix").crop(path, Segment(0.2, 0.7))
    assert wav.shape[1] / sr == 0.5


def test_can_crop_waveform():                                      # io_test.py:57-65
    waveform = torch.rand(1, 16000 * 2)
    wav, sr = Audio(mono="downmix").crop({"waveform": waveform, "sample_rate": 16000}, Segment(0.2, 0.7))
    assert isinstance(wav, torch.Tensor) and sr == 16000
    assert torch.equal(wav, waveform[:, 3200:11200])


def test_can_load_from_file_like(wav_file):                        # io_test.py:69-77
    path, sr, data = wav_file
    loader = Audio(mono="downmix")
    with open(path, "rb") as f:
        wav, rate = loader(f)
        assert f.tell() == 0                                       # rewound for the next reader
        again, _ = loader({"audio": f, "uri": "stream"})
    assert isinstance(wav, torch.Tensor) and rate == sr and torch.equal(wav, again)
    assert torch.equal(wav, loader(path)[0])


def test_can_crop_from_file_like(wav_file):                        # io_test.py:80-90
    path, sr, _ = wav_file
    loader = Audio(mono="downmix")
    with open(path, "rb") as f:
        wav, rate = loader.crop(f, Segment(0.2, 0.7))
    assert isinstance(wav, torch.Tensor) and rate == sr and wav.shape[1] == 0.5 * 16000
    in_memory = io.BytesIO(open(path, "rb").read())
    assert torch.equal(loader.crop(in_memory, Segment(0.2, 0.7))[0], wav)


def test_validate_file_contract(tmp_path):                         # core/io.py:153-216
    assert Audio.validate_file(io.BytesIO(b""))["uri"] == "stream"
    with pytest.raises(ValueError):
        Audio.validate_file({"waveform": torch.zeros(16000, 1), "sample_rate": 16000})   # (time, channel)
    with pytest.raises(ValueError):
        Audio.validate_file({"waveform": torch.zeros(1, 16000)})                          # no sample rate
    with pytest.raises(ValueError):
        Audio.validate_file(tmp_path / "missing.wav")
    with pytest.raises(ValueError):
        Audio.validate_file(3.14)


def test_resampling_never_falls_back_to_the_cpu(wav_file):         # io_test.py:10-18 needs a GPU here
    path, sr, _ = wav_file
    if torch.cuda.is_available():
        pytest.skip("covered on the GPU by tests/test_resample_gpu.py")
    with pytest.raises(RuntimeError, match="no CPU path"):
        Audio(sample_rate=sr // 2, mono="downmix")(path)


def test_crop_bounds_follow_the_reference():                       # core/io.py:384-413 (in-memory), :433-455 (files)
    sr = 16000
    waveform = torch.arange(2 * sr, dtype=torch.float32)[None]
    loader = Audio(mono="downmix")
    file = {"waveform": waveform, "sample_rate": sr}
    with pytest.raises(ValueError, match="negative start"):
        loader.crop(file, Segment(-0.5, 0.5))
    with pytest.raises(ValueError, match="greater than"):
        loader.crop(file, Segment(1.5, 2.0))                       # an end AT the last sample raises (>=)
    wav, _ = loader.crop(file, Segment(1.5, 2.0 - 1.0 / sr))
    assert wav.shape[1] == 7999 and wav[0, 0] == 24000
    wav, _ = loader.crop(file, Segment(-0.5, 0.5), mode="pad")
    assert wav.shape[1] == sr and torch.all(wav[0, :8000] == 0) and wav[0, 8000] == 0 and wav[0, 8001] == 1
    wav, _ = loader.crop(file, Segment(1.5, 2.5), mode="pad")
    assert wav.shape[1] == sr and wav[0, 0] == 24000 and torch.all(wav[0, 8000:] == 0)


def test_file_crop_bounds(wav_file):
    path, sr, data = wav_file
    loader = Audio(mono="downmix")
    mono = torch.from_numpy((data.astype(np.float32) / 32768.0).mean(axis=1))
    wav, _ = loader.crop(path, Segment(1.5, 2.0))                  # files: the end of the file is a valid end
    assert wav.shape[1] == 8000 and torch.allclose(wav[0], mono[24000:], atol=1e-7)
    with pytest.raises(ValueError, match="greater than"):
        loader.crop(path, Segment(1.5, 2.5))
    wav, _ = loader.crop(path, Segment(1.5, 2.5), mode="pad")
    assert wav.shape[1] == sr and torch.all(wav[0, 8000:] == 0)
    wav, _ = loader.crop({"audio": str(path), "channel": 1}, Segment(0.0, 0.25))
    assert torch.allclose(wav[0], torch.from_numpy(data[:4000, 1].astype(np.float32) / 32768.0), atol=1e-7)


def test_power_normalize_and_duration(wav_file):                   # core/io.py:134-151, 266-290
    path, sr, data = wav_file
    x = torch.randn(3, 2, 8000) * 7.0
    y = Audio.power_normalize(x)
    assert torch.allclose(y.square().mean(dim=-1), torch.ones(3, 2), atol=1e-4)
    assert Audio().get_duration(path) == data.shape[0] / sr
    assert Audio().get_duration({"waveform": torch.zeros(1, 24000), "sample_rate": 16000}) == 1.5
    with open(path, "rb") as f:
        assert Audio().get_duration(f) == 2.0 and f.tell() == 0


def test_device_side_sample_conversion_equals_the_host_one(tmp_path):
    """`Audio.load_on_device` copies the stored samples and scales them on the target device (here: the CPU, the
    same torch operations); every stored format gives the bits of `Audio.__call__` -- power-of-two scalings -- and a
    file with more than two channels takes the host path"""
    from scipy.io import wavfile
    rng = np.random.default_rng(0)
    n = 16000
    cases = {"i16": (rng.standard_normal(n) * 8000).astype(np.int16),
             "i16_stereo": (rng.standard_normal((n, 2)) * 8000).astype(np.int16),
             "i16_three": (rng.standard_normal((n, 3)) * 8000).astype(np.int16),
             "i32": (rng.standard_normal(n) * 2e8).astype(np.int32),
             "u8": rng.integers(0, 256, n).astype(np.uint8),
             "f32": (rng.standard_normal(n) * 0.1).astype(np.float32)}
    audio = Audio(sample_rate=16000, mono="downmix")
    for name, data in cases.items():
        path = tmp_path / f"{name}.wav"
        wavfile.write(str(path), 16000, data)
        want, sr = audio(str(path))
        got, sr2 = audio.load_on_device(str(path), torch.device("cpu"))
        assert sr == sr2 == 16000 and got.dtype == torch.float32 and torch.equal(got, want), name
        raw = Audio._read_raw(str(path))                               # what apply_batch reads one file ahead
        assert torch.equal(audio.load_on_device({"audio": str(path)}, torch.device("cpu"), raw=raw)[0], want)
        assert torch.equal(audio.load_on_device({"audio": str(path), "channel": 0}, torch.device("cpu"))[0],
                           audio({"audio": str(path), "channel": 0})[0])
    # the historical host conversion (numpy: astype + divide) gives the same bits as the torch one used now
    assert np.array_equal(cases["i16"].astype(np.float32) / 32768.0, audio(str(tmp_path / "i16.wav"))[0][0].numpy())
    assert np.array_equal((cases["u8"].astype(np.float32) - 128.0) / 128.0, audio(str(tmp_path / "u8.wav"))[0][0].numpy())
    assert np.array_equal(cases["i32"].astype(np.float32) / 2147483648.0, audio(str(tmp_path / "i32.wav"))[0][0].numpy())
    # a waveform mapping passes through
    wav = torch.randn(1, 100)
    assert torch.equal(audio.load_on_device({"waveform": wav, "sample_rate": 16000}, torch.device("cpu"))[0], wav)


def test_what_apply_batch_reads_ahead(tmp_path):
    """only a file on DISK is read ahead (worker thread, host only): resident waveforms and file objects are left to
    the front end itself"""
    import io
    from scipy.io import wavfile
    import pyannote_audio_amd as pa
    data = (np.random.default_rng(1).standard_normal(800) * 5000).astype(np.int16)
    path = tmp_path / "a.wav"
    wavfile.write(str(path), 8000, data)
    read_ahead = pa.SpeakerDiarization._read_ahead
    raw, rate = read_ahead({"audio": str(path), "uri": "a"})
    assert rate == 8000 and raw.dtype == np.int16 and np.array_equal(raw, data)
    raw2, _ = read_ahead({"audio": path, "uri": "a"})                 # a pathlib.Path
    assert np.array_equal(raw2, data)
    assert read_ahead({"waveform": torch.zeros(1, 10), "sample_rate": 8000, "audio": str(path)}) is None
    with open(path, "rb") as fp:
        assert read_ahead({"audio": fp, "uri": "stream"}) is None
    assert read_ahead({"audio": io.BytesIO(path.read_bytes()), "uri": "stream"}) is None
