"""Loads modules of the REAL reference (pyannote.audio 4.0.7 under /root/reference/src) so that tests
can execute the reference's own code beside the oracle.  TEST INFRASTRUCTURE ONLY.

`import pyannote.audio` itself fails in this image: lightning, pyannote.core / .pipeline / .database /
.metrics, torchaudio, asteroid_filterbanks, torchcodec and torch_audiomentations are not installed and
there is no network.  But most of the hot path lives in files whose OWN code only needs torch / numpy /
scipy / einops.  So:

  * `pyannote`, `pyannote.audio` and its sub-packages are registered as EMPTY package shells whose
    `__path__` points into /root/reference/src -- their `__init__.py` (which import everything) are
    not executed, but `import pyannote.audio.utils.vbx` etc. resolve to the reference's files;
  * the third-party packages that are absent get STAND-INS (section "third-party stand-ins" below):
    pyannote.core = the product's restatement (pyannote_audio_amd.core), lightning = a LightningModule
    that is an nn.Module with `save_hyperparameters`, asteroid_filterbanks / torchaudio.compliance.kaldi
    / torchaudio.functional.resample = the oracle's restatements.  Those stay PARITY UNPINNED (they are
    ours on both sides of the comparison); everything else that runs is the reference's own code.

The reference is read where it lies; nothing is copied.  /root/reference does not exist on the GPU
box: every user of this module skips when `available()` is False, and the vectors the GPU tests need
are committed under tests/golden/ (made by tests/golden/make_reference_golden.py).

    with reference_modules() as ref:
        vbx = ref.load("pyannote.audio.utils.vbx")
"""
from __future__ import annotations

import contextlib
import importlib
import importlib.abc
import importlib.machinery
import os
import sys
import types

REF_SRC = "/root/reference/src"
REF_PKG = os.path.join(REF_SRC, "pyannote", "audio")

_SHELLS = [
    "pyannote", "pyannote.audio", "pyannote.audio.core", "pyannote.audio.utils", "pyannote.audio.models",
    "pyannote.audio.models.blocks", "pyannote.audio.models.embedding", "pyannote.audio.models.segmentation",
    "pyannote.audio.pipelines", "pyannote.audio.pipelines.utils",
]
_PREFIXES = ("pyannote", "lightning", "asteroid_filterbanks", "torchaudio", "torchcodec",
             "torch_audiomentations", "torchmetrics")


def available() -> bool:
    return os.path.isfile(os.path.join(REF_PKG, "utils", "vbx.py"))


def _shell(name: str) -> types.ModuleType:
    mod = types.ModuleType(name)
    rel = name.split(".")
    mod.__path__ = [os.path.join(REF_SRC, *rel)]
    mod.__package__ = name
    return mod


def _module(name: str, **attrs) -> types.ModuleType:
    mod = types.ModuleType(name)
    mod.__dict__.update(attrs)
    return mod


# ----------------------------------------------------------------------------- third-party stand-ins
def _lightning_standins() -> dict:
    """lightning 2.6.1 is absent.  `LightningModule` = nn.Module + `save_hyperparameters` / `hparams`
    (what core/model.py:94 and PyanNet.py:90 use); `is_oom_error`; `pl_load` = torch.load."""
    import inspect

    import torch
    import torch.nn as nn

    class _HParams(dict):
        __getattr__ = dict.__getitem__
        __setattr__ = dict.__setitem__

    class LightningModule(nn.Module):
        def save_hyperparameters(self, *names, **_):
            frame = inspect.currentframe().f_back
            if not hasattr(self, "_hparams"):
                object.__setattr__(self, "_hparams", _HParams())
            for n in names:
                self._hparams[n] = frame.f_locals[n]

        @property
        def hparams(self):
            if not hasattr(self, "_hparams"):
                object.__setattr__(self, "_hparams", _HParams())
            return self._hparams

        @property
        def device(self):
            p = next(self.parameters(), None)
            return p.device if p is not None else torch.device("cpu")

        @classmethod
        def load_from_checkpoint(cls, checkpoint_path, map_location=None, hparams_file=None, strict=True,
                                 weights_only=None, **kwargs):
            """lightning's published loading sequence: cls(**hyper_parameters, **kwargs) ->
            on_load_checkpoint(checkpoint) -> load_state_dict(state_dict, strict) -> eval()."""
            ckpt = torch.load(checkpoint_path, map_location=map_location, weights_only=False)
            hparams = dict(ckpt.get("hyper_parameters", {}))
            hparams.update(kwargs)
            # lightning 2.x core/saving.py::_load_state: "filter kwargs according to class init unless it allows
            # unspecified arguments via kwargs" -- a saved hyper-parameter that cls.__init__ does not take is dropped
            # (the eleven fbank hyper-parameters BaseWeSpeakerResNet saves vs the eight WeSpeakerResNet34 takes)
            import inspect
            params = inspect.signature(cls.__init__).parameters
            if not any(p.kind is inspect.Parameter.VAR_KEYWORD for p in params.values()):
                hparams = {k: v for k, v in hparams.items() if k in params}
            model = cls(**hparams)
            model.on_load_checkpoint(ckpt)
            model.load_state_dict(ckpt["state_dict"], strict=strict)
            return model.eval()

        def on_load_checkpoint(self, checkpoint):
            pass

    def is_oom_error(exception) -> bool:
        return isinstance(exception, RuntimeError) and "out of memory" in str(exception)

    def _load(path, map_location=None, weights_only=None):
        return torch.load(path, map_location=map_location, weights_only=bool(weights_only))

    class ModelSummary:
        def __init__(self, *a, **k):
            pass

    class Trainer:
        pass

    class Callback:
        pass

    lightning = _module("lightning", LightningModule=LightningModule, LightningDataModule=object,
                        Trainer=Trainer, Callback=Callback)
    lightning.__path__ = []
    mods = {"lightning": lightning}
    for name, attrs in [
        ("lightning.pytorch", {"LightningModule": LightningModule, "Trainer": Trainer, "Callback": Callback}),
        ("lightning.pytorch.utilities", {}),
        ("lightning.pytorch.utilities.memory", {"is_oom_error": is_oom_error}),
        ("lightning.pytorch.utilities.model_summary", {}),
        ("lightning.pytorch.utilities.model_summary.model_summary", {"ModelSummary": ModelSummary}),
        ("lightning.fabric", {}),
        ("lightning.fabric.utilities", {}),
        ("lightning.fabric.utilities.cloud_io", {"_load": _load}),
    ]:
        m = _module(name, **attrs)
        m.__path__ = []
        mods[name] = m
    return mods


def _pyannote_core_standins() -> dict:
    """pyannote-core 6.0.1 is absent: the product's restatement of the subset the path touches."""
    import itertools

    import pyannote_audio_amd.core as pcore

    def pairwise(iterable):
        a, b = itertools.tee(iterable)
        next(b, None)
        return zip(a, b)

    class Timeline(list):
        def __init__(self, segments=None, uri=None):
            super().__init__(segments or [])
            self.uri = uri

    import numpy as np

    class SlidingWindowFeature(pcore.SlidingWindowFeature, np.lib.mixins.NDArrayOperatorsMixin):
        """+ pyannote.core's numpy protocol: a ufunc (or its reduction: np.sum(swf, axis=-1, keepdims=True)
        at pipelines/utils/diarization.py:177, -swf at :261) applied to a SlidingWindowFeature yields a
        SlidingWindowFeature on the same sliding window."""

        def __array_ufunc__(self, ufunc, method, *inputs, **kwargs):
            raw = tuple(x.data if isinstance(x, pcore.SlidingWindowFeature) else x for x in inputs)
            result = getattr(ufunc, method)(*raw, **kwargs)
            if isinstance(result, np.ndarray):
                return SlidingWindowFeature(result, self.sliding_window, labels=self.labels)
            return result

        def crop(self, focus, mode="loose", fixed=None, return_data=True):
            out = super().crop(focus, mode=mode, fixed=fixed, return_data=return_data)
            if isinstance(out, pcore.SlidingWindowFeature):
                out = SlidingWindowFeature(out.data, out.sliding_window, labels=out.labels)
            return out

    core = _module("pyannote.core", Segment=pcore.Segment, SlidingWindow=pcore.SlidingWindow,
                   SlidingWindowFeature=SlidingWindowFeature, Annotation=pcore.Annotation,
                   Timeline=Timeline)
    core.__path__ = []
    utils = _module("pyannote.core.utils")
    utils.__path__ = []
    import importlib as _il

    def get_class_by_name(class_name, default_module_name=None):
        mod, _, cls = class_name.rpartition(".")
        return getattr(_il.import_module(mod or default_module_name), cls)

    return {
        "pyannote.core": core,
        "pyannote.core.utils": utils,
        "pyannote.core.utils.generators": _module("pyannote.core.utils.generators", pairwise=pairwise,
                                                  string_generator=pcore.string_generator),
        "pyannote.core.utils.types": _module("pyannote.core.utils.types", Label=object),
        "pyannote.core.utils.helper": _module("pyannote.core.utils.helper",
                                              get_class_by_name=get_class_by_name),
    }


def _pyannote_pipeline_standins() -> dict:
    """pyannote-pipeline 4.0.0 is absent: its published attribute protocol -- hyper-parameters declared
    as attributes (`Uniform`, `Categorical`, `Integer`, `ParamDict`) live in `_parameters`, sub-pipelines in
    `_pipelines`; `instantiate(params)` stores values in `_instantiated` (a ParamDict becomes an object
    with attribute AND key access) and calls `initialize()`; `freeze` = instantiate; attribute reads
    return the instantiated value."""
    from collections import OrderedDict

    class Parameter:
        def __init__(self, *a, **k):
            pass

    class ParamDict(Parameter):
        def __init__(self, **params):
            self.params = params

    class _Values(dict):
        def __getattr__(self, k):
            try:
                return self[k]
            except KeyError as e:
                raise AttributeError(k) from e

    class Pipeline:
        def __init__(self):
            d = self.__dict__
            d["_parameters"], d["_instantiated"], d["_pipelines"] = OrderedDict(), OrderedDict(), OrderedDict()
            d["training"] = False

        def __getattr__(self, name):
            d = self.__dict__
            for reg in ("_instantiated", "_parameters", "_pipelines"):
                if reg in d and name in d[reg]:
                    return d[reg][name]
            raise AttributeError(f"'{type(self).__name__}' object has no attribute '{name}'")

        def __setattr__(self, name, value):
            d = self.__dict__
            if isinstance(value, (Parameter, Pipeline)):
                for reg in ("_parameters", "_instantiated", "_pipelines"):
                    d[reg].pop(name, None)
                d.pop(name, None)
                d["_parameters" if isinstance(value, Parameter) else "_pipelines"][name] = value
                return
            object.__setattr__(self, name, value)

        def __delattr__(self, name):
            for reg in ("_parameters", "_instantiated", "_pipelines"):
                if name in self.__dict__[reg]:
                    del self.__dict__[reg][name]
                    return
            object.__delattr__(self, name)

        def parameters(self, frozen=False, instantiated=False):
            out = {}
            for name, p in self._parameters.items():
                if instantiated:
                    v = self._instantiated.get(name)
                    out[name] = dict(v) if isinstance(v, dict) else v
                else:
                    out[name] = p
            for name, sub in self._pipelines.items():
                out[name] = sub.parameters(frozen=frozen, instantiated=instantiated)
            return out

        def instantiate(self, params):
            for name, value in params.items():
                if name in self._pipelines:
                    self._pipelines[name].instantiate(value)
                elif isinstance(self._parameters.get(name), ParamDict):
                    self._instantiated[name] = _Values(value)
                elif name in self._parameters:
                    self._instantiated[name] = value
                else:
                    raise ValueError(f"parameter {name!r} does not exist")
            self.initialize()
            return self

        freeze = instantiate

        @property
        def instantiated(self):
            for name, p in self._parameters.items():
                if name not in self._instantiated:
                    return False
                if isinstance(p, ParamDict) and set(p.params) - set(self._instantiated[name]):
                    return False
            return all(sub.instantiated for sub in self._pipelines.values())

        def initialize(self):
            pass

    pipeline = _module("pyannote.pipeline", Pipeline=Pipeline)
    pipeline.__path__ = []
    parameter = _module("pyannote.pipeline.parameter", Parameter=Parameter, Categorical=type("Categorical", (Parameter,), {}),
                        Integer=type("Integer", (Parameter,), {}), Uniform=type("Uniform", (Parameter,), {}),
                        LogUniform=type("LogUniform", (Parameter,), {}), ParamDict=ParamDict)
    return {"pyannote.pipeline": pipeline, "pyannote.pipeline.parameter": parameter}


def _signal_standins() -> dict:
    """asteroid-filterbanks 0.4.0 / torchaudio 2.10.0 / torchcodec are absent: the oracle's restatements
    (oracle/models.py ParamSincFB + Encoder, kaldi_fbank; oracle/audio.py resample)."""
    import oracle.audio as oaudio
    import oracle.models as omodels

    def fbank(waveform, num_mel_bins=23, frame_length=25.0, frame_shift=10.0, sample_frequency=16000.0,
              round_to_power_of_two=True, snip_edges=True, dither=0.0, window_type="povey",
              use_energy=False, **kw):
        # the restatement covers exactly the WeSpeaker settings (wespeaker/__init__.py:88-99)
        assert round_to_power_of_two and snip_edges and dither == 0.0 and not use_energy and not kw
        assert window_type == "hamming"
        return omodels.kaldi_fbank(waveform, num_mel_bins=num_mel_bins, frame_length=frame_length,
                                   frame_shift=frame_shift, sample_frequency=sample_frequency)

    asteroid = _module("asteroid_filterbanks", Encoder=omodels.Encoder, ParamSincFB=omodels.ParamSincFB)
    kaldi = _module("torchaudio.compliance.kaldi", fbank=fbank)
    compliance = _module("torchaudio.compliance", kaldi=kaldi)
    compliance.__path__ = []
    functional = _module("torchaudio.functional", resample=oaudio.resample)
    transforms = _module("torchaudio.transforms", MFCC=_Anything)     # (XVectorMFCC is not on the path)
    # torchaudio's wav2vec 2.0 / WavLM (SSeRiouSS.py:100-124): oracle/wav2vec2.py, unpinned
    import oracle.wav2vec2 as ow
    models = _module("torchaudio.models", wav2vec2_model=ow.wav2vec2_model, wavlm_model=ow.wavlm_model)
    pipelines = _module("torchaudio.pipelines", **ow.PIPELINES)
    torchaudio = _module("torchaudio", compliance=compliance, functional=functional, transforms=transforms,
                         models=models, pipelines=pipelines)
    torchaudio.__path__ = []
    return {"asteroid_filterbanks": asteroid, "torchaudio": torchaudio, "torchaudio.compliance": compliance,
            "torchaudio.compliance.kaldi": kaldi, "torchaudio.functional": functional,
            "torchaudio.transforms": transforms, "torchaudio.models": models,
            "torchaudio.pipelines": pipelines}


class _Anything:
    """What a permissive stand-in module hands out for any name: a class that accepts any construction /
    call / subclassing (base classes of training-only code, telemetry hooks, type aliases)."""

    __args__ = ()      # (core/task.py:54 lists the members of a typing.Literal from pyannote.database)

    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return None

    def __class_getitem__(cls, item):
        return cls


class _PermissiveModule(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        value = type(name, (_Anything,), {})
        setattr(self, name, value)
        return value


class _PermissiveFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    """Training / telemetry / decoding packages that the path's modules import at load time but never
    call on the inference path (pyannote.database, pyannote.metrics, torch_audiomentations, torchmetrics,
    torchcodec, the reference's own opentelemetry wrapper): any sub-module, any name."""
    ROOTS = ("pyannote.database", "pyannote.metrics", "torch_audiomentations", "torchmetrics", "torchcodec",
             "pyannote.audio.telemetry", "pyannote.audio.core.calibration")

    def find_spec(self, fullname, path=None, target=None):
        if any(fullname == r or fullname.startswith(r + ".") for r in self.ROOTS):
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        mod = _PermissiveModule(spec.name)
        mod.__path__ = []
        return mod

    def exec_module(self, module):
        pass


class _Reference:
    def load(self, name: str):
        return importlib.import_module(name)

    def load_core(self):
        """core/io.py, model.py, inference.py, pipeline.py in dependency order, re-exported from the
        `pyannote.audio` shell the way the real __init__.py does (src/pyannote/audio/__init__.py:28-31)."""
        top = sys.modules["pyannote.audio"]
        # (utils/dependencies.py compares installed package versions: pyannote.audio is not installed)
        self.load("pyannote.audio.utils.dependencies").check_dependencies = lambda *a, **k: None
        top.Audio = self.load("pyannote.audio.core.io").Audio
        top.Model = self.load("pyannote.audio.core.model").Model
        top.Inference = self.load("pyannote.audio.core.inference").Inference
        top.Pipeline = self.load("pyannote.audio.core.pipeline").Pipeline
        return top

    def load_pipelines(self):
        """pipelines/utils/__init__.py:23-36 without its training-only getters, then clustering and
        the speaker-diarization pipeline."""
        top = self.load_core()
        utils = sys.modules["pyannote.audio.pipelines.utils"]
        utils.oracle_segmentation = self.load("pyannote.audio.pipelines.utils.oracle").oracle_segmentation
        diar = self.load("pyannote.audio.pipelines.utils.diarization")
        utils.SpeakerDiarizationMixin = diar.SpeakerDiarizationMixin
        getter = self.load("pyannote.audio.pipelines.utils.getter")
        for n in ("PipelineAugmentation", "PipelineCalibration", "PipelineModel", "PipelinePLDA",
                  "get_augmentation", "get_calibration", "get_devices", "get_model", "get_plda"):
            setattr(utils, n, getattr(getter, n))
        pipelines = sys.modules["pyannote.audio.pipelines"]
        pipelines.clustering = self.load("pyannote.audio.pipelines.clustering")
        pipelines.speaker_verification = self.load("pyannote.audio.pipelines.speaker_verification")
        pipelines.speaker_diarization = self.load("pyannote.audio.pipelines.speaker_diarization")
        pipelines.SpeakerDiarization = pipelines.speaker_diarization.SpeakerDiarization
        return top


@contextlib.contextmanager
def reference_modules(third_party: bool = False):
    """Context in which `pyannote.audio.*` resolves to the reference's files.  sys.modules is restored
    on exit (module objects obtained inside stay usable)."""
    if not available():
        raise RuntimeError("/root/reference is not present")
    import pyannote_audio_amd.core  # noqa: F401  (must see the REAL absence of pyannote.core first)
    saved = {k: v for k, v in sys.modules.items() if k.split(".")[0] in _PREFIXES}
    for k in saved:
        del sys.modules[k]
    finder = None
    try:
        for name in _SHELLS + ([] if third_party else ["pyannote.audio.models.embedding.wespeaker"]):
            sys.modules[name] = _shell(name)   # (with stand-ins the real wespeaker/__init__.py can run)
        sys.modules["pyannote.audio"].__version__ = "4.0.7"
        if third_party:
            finder = _PermissiveFinder()
            sys.meta_path.insert(0, finder)
            for table in (_lightning_standins(), _pyannote_core_standins(), _pyannote_pipeline_standins(),
                          _signal_standins()):
                sys.modules.update(table)
        yield _Reference()
    finally:
        if finder in sys.meta_path:
            sys.meta_path.remove(finder)
        for k in [k for k in sys.modules if k.split(".")[0] in _PREFIXES]:
            del sys.modules[k]
        sys.modules.update(saved)
