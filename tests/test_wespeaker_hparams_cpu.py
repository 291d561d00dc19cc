"""The fbank hyper-parameters of a WeSpeaker checkpoint at load time ("What's weak" 1; reference:
wespeaker/__init__.py:56-99, 137-157, 346-372): a value the HIP front end is not built for is REFUSED; a key the
reference's class does not take is dropped with a warning, as the reference's loader drops it; the running-mean span
is available to the counterpart of a user class that forwards it."""
import os
import warnings

import pytest
import torch

import pyannote_audio_amd as pa
from pyannote_audio_amd import model as pm
from conftest import WESPEAKER_HPARAMS


@pytest.fixture(scope="module")
def small_state_dict():
    """a ResNet34-shaped state dict is not needed for loading: the loader only looks at hyper-parameters here"""
    from oracle.models import seeded_wespeaker
    return seeded_wespeaker().state_dict()


def _write(tmp_path, sd, hparams, arch=pm.WeSpeakerResNet34.ARCHITECTURE, name="m.bin"):
    path = str(tmp_path / name)
    pm.save_checkpoint(path, sd, hparams, arch, pm.embedding_specifications())
    return path


def test_defaults_load_quietly(tmp_path, small_state_dict):
    full = dict(WESPEAKER_HPARAMS, round_to_power_of_two=True, snip_edges=True, fbank_centering_span=None)
    for hp in (WESPEAKER_HPARAMS, full, {"sample_rate": 16000}):
        with warnings.catch_warnings():
            warnings.simplefilter("error")
            m = pa.Model.from_pretrained(_write(tmp_path, small_state_dict, hp))
        assert type(m).__name__ == "WeSpeakerResNet34" and m.fbank_center_kernel == 0


@pytest.mark.parametrize("key,value", [
    ("sample_rate", 8000), ("num_channels", 2), ("num_mel_bins", 64), ("num_mel_bins", 40), ("frame_length", 20.0),
    ("frame_length", 30), ("frame_shift", 5), ("frame_shift", 12.5), ("dither", 1.0), ("dither", 1e-5),
    ("window_type", "povey"), ("window_type", "hanning"), ("use_energy", True),
])
def test_every_unsupported_value_is_refused(tmp_path, small_state_dict, key, value):
    path = _write(tmp_path, small_state_dict, dict(WESPEAKER_HPARAMS, **{key: value}))
    with pytest.raises(NotImplementedError) as err:
        pa.Model.from_pretrained(path)
    assert key in str(err.value) and repr(value) in str(err.value)


@pytest.mark.parametrize("key,value,default", [
    ("round_to_power_of_two", False, True), ("snip_edges", False, True), ("fbank_centering_span", 3.0, None),
])
def test_keys_the_reference_class_does_not_take_are_dropped_with_a_warning(tmp_path, small_state_dict, key, value,
                                                                           default):
    path = _write(tmp_path, small_state_dict, dict(WESPEAKER_HPARAMS, **{key: value}))
    with pytest.warns(UserWarning, match=f"{key} = {value!r}.*as in the reference"):
        m = pa.Model.from_pretrained(path)
    assert key not in m.hparams and m.fbank_center_kernel == 0


def test_a_registered_user_class_that_forwards_the_span(tmp_path, small_state_dict):
    @pm.register_architecture
    class CentredResNet34(pm.WeSpeakerResNet34):
        ARCHITECTURE = ("my_project.models", "CentredResNet34")
        INIT_KEYS = pm.WeSpeakerResNet34.INIT_KEYS + ("fbank_centering_span",)

    try:
        for span, kernel in ((None, 0), (3.0, 299), (0.4, 39), (0.03, 1), (0.02, 1), (30.0, 2999)):
            hp = dict(WESPEAKER_HPARAMS, fbank_centering_span=span)
            m = pa.Model.from_pretrained(_write(tmp_path, small_state_dict, hp, CentredResNet34.ARCHITECTURE))
            assert type(m) is CentredResNet34 and m.fbank_center_kernel == kernel
            # the reference's arithmetic (wespeaker/__init__.py:141-157 with utils/receptive_field.py:26-53)
            if span is not None:
                k = 1 + (int(span * 16000) - 400) // 160
                assert kernel == 2 * (max(k, 0) // 2) + 1
        bad = _write(tmp_path, small_state_dict, dict(WESPEAKER_HPARAMS, fbank_centering_span=-1.0),
                     CentredResNet34.ARCHITECTURE)
        with pytest.raises(ValueError, match="fbank_centering_span"):
            pa.Model.from_pretrained(bad)
        # the stock class with the same file: dropped, global mean
        with pytest.warns(UserWarning, match="fbank_centering_span"):
            stock = pa.Model.from_pretrained(_write(tmp_path, small_state_dict,
                                                    dict(WESPEAKER_HPARAMS, fbank_centering_span=3.0)))
        assert stock.fbank_center_kernel == 0
    finally:
        pm._USER_ARCHITECTURES.pop(CentredResNet34.ARCHITECTURE, None)
    with pytest.raises(NotImplementedError, match="outside the accelerated hot path"):
        pa.Model.from_pretrained(_write(tmp_path, small_state_dict, WESPEAKER_HPARAMS, CentredResNet34.ARCHITECTURE))


def test_bottleneck_classes_follow_the_same_rules(tmp_path, small_state_dict):
    for klass in (pm.WeSpeakerResNet152, pm.WeSpeakerResNet221, pm.WeSpeakerResNet293):
        path = _write(tmp_path, small_state_dict, dict(WESPEAKER_HPARAMS, window_type="povey"), klass.ARCHITECTURE)
        with pytest.raises(NotImplementedError, match="window_type"):
            pa.Model.from_pretrained(path)
