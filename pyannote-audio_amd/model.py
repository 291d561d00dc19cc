"""Model objects of the hot path: specifications, checkpoint reader/writer, and the two models the
3.1 pipeline uses (PyanNet segmentation, WeSpeaker ResNet34 embedding) as thin host objects over the
HIP engines.  Mirrors core/model.py (specifications :114-153, receptive_field :168-184,
on_save/load_checkpoint :244-262, from_pretrained :497-655) and core/task.py:59-137.

Weights stay on the host (reference state-dict layout) until `.to(torch.device("cuda"))`, exactly like
the reference keeps modules on CPU until `pipeline.to(device)` (core/pipeline.py:591-611).  Calling a
model that has not been moved to a GPU raises: there is no CPU compute path in this package."""
from __future__ import annotations

import contextlib
import os
import pickle
import sys
import types
import warnings
from dataclasses import dataclass
from enum import Enum
from functools import cached_property
from pathlib import Path
from typing import List, Optional, Text, Tuple

import scipy.special
import torch

from . import ffi
from .core import SlidingWindow


# ---------------------------------------------------------------------------------------------
# core/task.py:59-137
# ---------------------------------------------------------------------------------------------
class Problem(Enum):
    BINARY_CLASSIFICATION = 0
    MONO_LABEL_CLASSIFICATION = 1
    MULTI_LABEL_CLASSIFICATION = 2
    REPRESENTATION = 3
    REGRESSION = 4


class Resolution(Enum):
    FRAME = 1
    CHUNK = 2


@dataclass
class Specifications:
    problem: Problem
    resolution: Resolution
    duration: float
    min_duration: Optional[float] = None
    warm_up: Optional[Tuple[float, float]] = (0.0, 0.0)
    classes: Optional[List[Text]] = None
    powerset_max_classes: Optional[int] = None
    permutation_invariant: bool = False

    @cached_property
    def powerset(self) -> bool:
        if self.powerset_max_classes is None:
            return False
        if self.problem != Problem.MONO_LABEL_CLASSIFICATION:
            raise ValueError("`powerset_max_classes` only makes sense with multi-class "
                             "classification problems.")
        return True

    @cached_property
    def num_powerset_classes(self) -> int:
        return int(sum(scipy.special.binom(len(self.classes), i)
                       for i in range(0, self.powerset_max_classes + 1)))

    def __len__(self):
        return 1

    def __iter__(self):
        yield self


# pickled checkpoints reference these classes by their reference module path
_REF_TASK_MODULE = "pyannote.audio.core.task"
for _k in (Problem, Resolution, Specifications):
    _k.__module__ = _REF_TASK_MODULE


class AttributeDict(dict):
    """stand-in for lightning's AttributeDict (hyper_parameters container)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


class _Opaque:
    """placeholder for pickled objects we never need (legacy Introspection, loss functions, ...)."""

    def __init__(self, *a, **k):
        pass

    def __setstate__(self, state):
        self.__dict__["state"] = state


class _ShimUnpickler(pickle.Unpickler):
    """Restricted unpickler for Lightning / pyannote.audio checkpoints: foreign classes of the
    pyannote / lightning ecosystem map to local stand-ins, tensor / container plumbing comes from an
    allow-list of modules, and EVERYTHING ELSE raises `UnpicklingError` -- a crafted checkpoint cannot
    resolve `os.system`, `builtins.eval` and the like (the reference's plain `weights_only=False`
    load can)."""

    _LOCAL = {"Specifications": Specifications, "Problem": Problem, "Resolution": Resolution}
    _SAFE_MODULES = ("torch", "collections", "numpy", "_codecs", "typing", "enum", "datetime",
                     "pathlib", "argparse", "omegaconf", "copyreg")
    _SAFE_BUILTINS = {"dict", "list", "tuple", "set", "frozenset", "int", "float", "bool", "str",
                      "bytes", "bytearray", "complex", "slice", "range", "object", "getattr"}

    def find_class(self, module, name):
        if module.startswith("pyannote.audio") and name in self._LOCAL:
            return self._LOCAL[name]
        if name == "AttributeDict":
            return AttributeDict
        if module.startswith(("pyannote.", "lightning", "pytorch_lightning", "torchmetrics",
                              "pytorch_metric_learning", "asteroid")):
            return _Opaque
        if module == "builtins":
            if name in self._SAFE_BUILTINS and name != "getattr":
                return super().find_class(module, name)
            raise pickle.UnpicklingError(f"checkpoint references builtins.{name}: refused")
        root = module.split(".")[0]
        if root in self._SAFE_MODULES and not (root == "torch" and name in ("load", "hub")):
            if root == "numpy" and name in ("load", "loads", "fromfile", "memmap"):
                raise pickle.UnpicklingError(f"checkpoint references {module}.{name}: refused")
            return super().find_class(module, name)
        raise pickle.UnpicklingError(
            f"checkpoint references {module}.{name}, which is not on the allow-list of the "
            "pyannote_audio_amd checkpoint reader")


_shim_pickle = types.ModuleType("pyannote_audio_amd._shim_pickle")
_shim_pickle.Unpickler = _ShimUnpickler
_shim_pickle.load = lambda f, **kw: _ShimUnpickler(f, **kw).load()
_shim_pickle.__name__ = "pickle"
for _n in ("dumps", "dump", "loads", "Pickler", "PickleError", "UnpicklingError", "HIGHEST_PROTOCOL"):
    setattr(_shim_pickle, _n, getattr(pickle, _n))


@contextlib.contextmanager
def _reference_task_module():
    """Temporarily expose our Specifications/Problem/Resolution as `pyannote.audio.core.task` so that
    pickling by reference produces checkpoints the reference can read (core/model.py:244-256)."""
    created = []
    parts = _REF_TASK_MODULE.split(".")
    for i in range(1, len(parts) + 1):
        name = ".".join(parts[:i])
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
            created.append(name)
    mod = sys.modules[_REF_TASK_MODULE]
    saved = {k: getattr(mod, k, None) for k in ("Specifications", "Problem", "Resolution")}
    mod.Specifications, mod.Problem, mod.Resolution = Specifications, Problem, Resolution
    try:
        yield
    finally:
        for k, v in saved.items():
            if v is None:
                delattr(mod, k)
            else:
                setattr(mod, k, v)
        for name in reversed(created):
            sys.modules.pop(name, None)


def load_checkpoint(path) -> dict:
    """torch.load of a Lightning checkpoint dict {state_dict, hyper_parameters, "pyannote.audio": ...}
    (core/model.py:602-604 uses weights_only=False; so must we, through the restricted unpickler)."""
    if isinstance(path, (str, Path)) and os.path.isdir(path):
        path = Path(path) / "pytorch_model.bin"
    return torch.load(path, map_location="cpu", weights_only=False, pickle_module=_shim_pickle)


def save_checkpoint(path, state_dict: dict, hyper_parameters: dict, architecture: Tuple[str, str],
                    specifications: Specifications, version: str = "3.1.0"):
    """Write a checkpoint in the reference's on-disk format (core/model.py:244-256)."""
    ckpt = {
        "state_dict": {k: v.detach().cpu() for k, v in state_dict.items()},
        "hyper_parameters": dict(hyper_parameters),
        "pytorch-lightning_version": "2.6.1",
        "pyannote.audio": {
            "versions": {"pyannote.audio": version},
            "architecture": {"module": architecture[0], "class": architecture[1]},
            "specifications": specifications,
        },
    }
    with _reference_task_module():
        torch.save(ckpt, path)


# ---------------------------------------------------------------------------------------------
# receptive-field arithmetic (utils/receptive_field.py:26-165)
# ---------------------------------------------------------------------------------------------
def multi_conv_num_frames(n, kernel_size, stride, padding, dilation):
    for k, s, p, d in zip(kernel_size, stride, padding, dilation):
        n = 1 + (n + 2 * p - d * (k - 1) - 1) // s
    return n


def multi_conv_receptive_field_size(num_frames, kernel_size, stride, padding, dilation):
    size = num_frames
    for k, s, p, d in reversed(list(zip(kernel_size, stride, padding, dilation))):
        size = 1 + (k - 1) * d + (size - 1) * s - 2 * p
    return size


def multi_conv_receptive_field_center(frame, kernel_size, stride, padding, dilation):
    c = frame
    for k, s, p, d in reversed(list(zip(kernel_size, stride, padding, dilation))):
        c = c * s + ((1 + (k - 1) * d) - 1) // 2 - p
    return c


class _AudioSpec:
    """what callers read from `model.audio` (core/io.py): sample rate, mono, get_num_samples."""

    def __init__(self, sample_rate: int = 16000, mono: str = "downmix"):
        self.sample_rate = sample_rate
        self.mono = mono

    def get_num_samples(self, duration: float, sample_rate: Optional[int] = None) -> int:
        sample_rate = sample_rate or self.sample_rate
        return round(duration * sample_rate)  # core/io.py:292-304


class Model:
    """Host-side model object (not an nn.Module: the weights are kernel-ready device images)."""

    ARCHITECTURE = ("", "")

    def __init__(self, state_dict: dict, hparams: dict, specifications: Specifications):
        self._state_dict = {k: v.detach().cpu() for k, v in state_dict.items()}
        self.hparams = AttributeDict(hparams)
        self.specifications = specifications
        self.device = torch.device("cpu")
        self.audio = _AudioSpec(int(self.hparams.get("sample_rate", 16000)), "downmix")
        self._engine = None

    # -- reference API surface
    def state_dict(self):
        return self._state_dict

    def eval(self):
        return self

    def to(self, device: torch.device):
        if not isinstance(device, torch.device):
            raise TypeError(
                f"`device` must be an instance of `torch.device`, got `{type(device).__name__}`")
        if device.type == "cuda":
            ffi.require_gpu()
            if device.index is None:
                device = torch.device("cuda", torch.cuda.current_device())
            if self._engine is None or self.device != device:
                self._engine = self._build_engine(device)
        else:
            self._engine = None
        self.device = device
        return self

    def cuda(self, index: Optional[int] = None):
        return self.to(torch.device("cuda") if index is None else torch.device("cuda", index))

    @property
    def engine(self):
        if self._engine is None:
            raise RuntimeError(
                f"{type(self).__name__} lives on {self.device}: this package has no CPU compute path. "
                "Move it to an MI355X first: model.to(torch.device('cuda')).")
        return self._engine

    @cached_property
    def receptive_field(self) -> SlidingWindow:
        size = self.receptive_field_size(num_frames=1)
        step = self.receptive_field_size(num_frames=2) - size
        start = self.receptive_field_center(frame=0) - (size - 1) / 2
        sr = self.audio.sample_rate
        return SlidingWindow(start=start / sr, duration=size / sr, step=step / sr)

    def save(self, path):
        save_checkpoint(path, self._state_dict, dict(self.hparams), self.ARCHITECTURE,
                        self.specifications)

    @classmethod
    def from_pretrained(cls, checkpoint, map_location=None, strict: bool = True,
                        subfolder: Optional[str] = None, revision: Optional[str] = None,
                        token=None, cache_dir=None, **kwargs) -> Optional["Model"]:
        """core/model.py:497-655 restricted to local checkpoints (no network in scope): a path to
        `pytorch_model.bin`, a directory containing it (optionally `subfolder`), or a byte buffer."""
        if isinstance(checkpoint, (str, Path)):
            p = Path(checkpoint)
            if p.is_dir():
                p = p / subfolder / "pytorch_model.bin" if subfolder else p / "pytorch_model.bin"
            if not p.exists():
                raise FileNotFoundError(
                    f"{checkpoint!s}: only local checkpoints are supported (no Hugging Face download "
                    "in this build: SURVEY.md section 2, row 14)")
            checkpoint = p
        ckpt = load_checkpoint(checkpoint)
        arch = ckpt["pyannote.audio"]["architecture"]["class"]
        klass = {"PyanNet": PyanNet, "SSeRiouSS": SSeRiouSS, "WeSpeakerResNet34": WeSpeakerResNet34,
                 "WeSpeakerResNet152": WeSpeakerResNet152, "WeSpeakerResNet221": WeSpeakerResNet221,
                 "WeSpeakerResNet293": WeSpeakerResNet293, "XVectorSincNet": XVectorSincNet}.get(arch)
        # a user's own class (the reference imports `architecture.module` and takes `architecture.class` from it,
        # core/model.py:609-613): its counterpart registered with `register_architecture`
        module = ckpt["pyannote.audio"]["architecture"].get("module", "")
        klass = _USER_ARCHITECTURES.get((module, arch), klass)
        if klass is None:
            raise NotImplementedError(
                f"architecture {arch!r} is outside the accelerated hot path (PyanNet, SSeRiouSS, "
                "WeSpeakerResNet34/152/221/293, XVectorSincNet)")
        return klass(ckpt["state_dict"], dict(ckpt.get("hyper_parameters", {})),
                     ckpt["pyannote.audio"]["specifications"])


_USER_ARCHITECTURES: dict = {}


def register_architecture(klass) -> type:
    """Make `Model.from_pretrained` build `klass` for checkpoints whose architecture is `klass.ARCHITECTURE` =
    (module, class) of a user's model class in the reference (which finds it by importing that module,
    core/model.py:609-613).  Usable as a decorator.  Example: a reference subclass of `BaseWeSpeakerResNet` that
    forwards `fbank_centering_span` <-> a subclass of `WeSpeakerResNet34` here with that key added to `INIT_KEYS`."""
    module, name = klass.ARCHITECTURE
    if not module or not name:
        raise ValueError("ARCHITECTURE = (module, class) of the reference class is required")
    _USER_ARCHITECTURES[(module, name)] = klass
    return klass


class PyanNet(Model):
    """models/segmentation/PyanNet.py:38-240 over the HIP segmentation engine."""

    ARCHITECTURE = ("pyannote.audio.models.segmentation.PyanNet", "PyanNet")
    _K = [251, 3, 5, 3, 5, 3]

    def _conv(self):
        s = int((self.hparams.get("sincnet") or {}).get("stride", 10))
        return self._K, [s, 3, 1, 3, 1, 3], [0] * 6, [1] * 6

    def num_frames(self, num_samples: int) -> int:
        return multi_conv_num_frames(num_samples, *self._conv())

    def receptive_field_size(self, num_frames: int = 1) -> int:
        return multi_conv_receptive_field_size(num_frames, *self._conv())

    def receptive_field_center(self, frame: int = 0) -> int:
        return multi_conv_receptive_field_center(frame, *self._conv())

    @property
    def dimension(self) -> int:
        s = self.specifications
        return s.num_powerset_classes if s.powerset else len(s.classes)

    def _build_engine(self, device):
        from .segmentation import SegmentationEngine
        from .weights import SegmentationPack
        s = self.specifications
        if s.powerset:
            pack = SegmentationPack(self._state_dict, dict(self.hparams), s.num_powerset_classes,
                                    len(s.classes), s.powerset_max_classes, device)
        elif s.problem in (Problem.MULTI_LABEL_CLASSIFICATION, Problem.BINARY_CLASSIFICATION):
            # sigmoid scores per class (default_activation, core/model.py:286-294)
            pack = SegmentationPack(self._state_dict, dict(self.hparams), len(s.classes), len(s.classes),
                                    None, device)
        else:
            raise NotImplementedError(f"{s.problem} heads are outside the accelerated path (powerset or "
                                      "multi-label frame-level segmentation)")
        return SegmentationEngine(pack)

    def __call__(self, waveforms: torch.Tensor) -> torch.Tensor:
        """(batch, 1, samples) -> (batch, frames, classes) log-probabilities, or sigmoid scores for a
        multi-label checkpoint (PyanNet.py:211-240)."""
        return self.engine.forward(waveforms)

    forward = __call__


class SSeRiouSS(PyanNet):
    """models/segmentation/SSeRiouSS.py:42-328 (wav2vec 2.0 / WavLM > LSTM > feed-forward > classifier) over
    the HIP wav2vec engine.  Frame geometry = the feature extractor's convolutions (:217-287)."""

    ARCHITECTURE = ("pyannote.audio.models.segmentation.SSeRiouSS", "SSeRiouSS")

    def _conv(self):
        from .weights import wav2vec_config
        shapes = wav2vec_config(self.hparams.get("wav2vec") or "WAVLM_BASE")["extractor_conv_layer_config"]
        n = len(shapes)
        return [k for _, k, _ in shapes], [s for _, _, s in shapes], [0] * n, [1] * n

    def _build_engine(self, device):
        from .segmentation import SSeRiouSSEngine
        from .weights import SSeRiouSSPack
        s = self.specifications
        if s.powerset:
            args = (s.num_powerset_classes, len(s.classes), s.powerset_max_classes)
        elif s.problem in (Problem.MULTI_LABEL_CLASSIFICATION, Problem.BINARY_CLASSIFICATION):
            args = (len(s.classes), len(s.classes), None)
        else:
            raise NotImplementedError(f"{s.problem} heads are outside the accelerated path")
        return SSeRiouSSEngine(SSeRiouSSPack(self._state_dict, dict(self.hparams), *args, device))


class WeSpeakerResNet34(Model):
    """models/embedding/wespeaker/__init__.py:346-372 over the HIP embedding engine.

    Every WeSpeaker checkpoint of the reference carries the arguments of its kaldi fbank front end as
    hyper-parameters (`BaseWeSpeakerResNet.__init__` saves eleven of them, wespeaker/__init__.py:56-85).  Two rules, both
    the reference's:

    * WHICH of them count.  `Model.from_pretrained` rebuilds the model through Lightning's `load_from_checkpoint`
      (core/model.py:620-626), which passes the saved hyper-parameters to `cls.__init__` FILTERED BY ITS SIGNATURE
      (lightning 2.x `core/saving.py::_load_state`: "filter kwargs according to class init unless it allows
      unspecified arguments via kwargs").  The four stock classes take sample_rate, num_channels, num_mel_bins,
      frame_length, frame_shift, dither, window_type, use_energy (:346-372, :375-470) -- NOT round_to_power_of_two,
      snip_edges or fbank_centering_span, which therefore always have their defaults (True, True, None) in the
      reference whatever the file says.  `INIT_KEYS` mirrors that; a dropped key that differs from its default is
      reported with a warning, and the default is used -- as in the reference.  A user class of the reference that
      forwards more of them to `BaseWeSpeakerResNet` has its counterpart here: subclass, extend `INIT_KEYS`, name the
      reference class in `ARCHITECTURE`, `register_architecture`.
    * WHAT the front end can do.  csrc/emb_fbank.hip is built for the values every published checkpoint has -- 16 kHz,
      mono, 80 mel bins, 25 ms / 10 ms hamming frames, no energy, no dither, snip_edges, power-of-two FFT; a
      hyper-parameter that COUNTS and says anything else is refused at load time (`NotImplementedError`): computing
      something else without a word would be worse.  `fbank_centering_span` (running instead of global mean
      subtraction, :137-157) is implemented (`pa_fbank_center_span`)."""

    ARCHITECTURE = ("pyannote.audio.models.embedding.wespeaker", "WeSpeakerResNet34")

    #: the hyper-parameters the reference class's __init__ takes (wespeaker/__init__.py:346-372): the others are dropped
    #: by the reference's loader
    INIT_KEYS = ("sample_rate", "num_channels", "num_mel_bins", "frame_length", "frame_shift", "dither",
                 "window_type", "use_energy")
    #: hyper-parameter -> the one value the HIP fbank kernel is built for (= the reference's defaults, :56-71)
    FBANK_BUILT_FOR = {"sample_rate": 16000, "num_channels": 1, "num_mel_bins": 80, "frame_length": 25.0,
                       "frame_shift": 10.0, "round_to_power_of_two": True, "snip_edges": True, "dither": 0.0,
                       "window_type": "hamming", "use_energy": False}

    @staticmethod
    def _same(got, built) -> bool:
        if isinstance(built, bool):
            return isinstance(got, (bool, int)) and bool(got) == built
        if isinstance(built, str):
            return got == built
        return isinstance(got, (int, float)) and not isinstance(got, bool) and float(got) == float(built)

    def __init__(self, state_dict: dict, hparams: dict, specifications: Specifications):
        defaults = dict(self.FBANK_BUILT_FOR, fbank_centering_span=None)
        counted = dict(hparams)
        for key, default in defaults.items():
            if key in counted and key not in self.INIT_KEYS:
                got = counted.pop(key)
                differs = (got is not None) if default is None else not self._same(got, default)
                if differs:
                    warnings.warn(
                        f"{type(self).__name__}: the checkpoint says {key} = {got!r}, but "
                        f"{self.ARCHITECTURE[1]}.__init__ of the reference does not take {key} and its loader drops "
                        f"it (Lightning filters hyper-parameters by the signature): {key} = {default!r} is used, as "
                        "in the reference")
        super().__init__(state_dict, counted, specifications)
        for key, built in self.FBANK_BUILT_FOR.items():
            got = self.hparams.get(key, built)
            if not self._same(got, built):
                raise NotImplementedError(
                    f"{type(self).__name__}: hyper-parameter {key} = {got!r}, but the HIP fbank front end is built for "
                    f"{key} = {built!r} only (wespeaker/__init__.py:56-99); refusing to load a checkpoint whose "
                    "features would silently differ")
        span = self.hparams.get("fbank_centering_span", None)
        if span is not None and not (isinstance(span, (int, float)) and not isinstance(span, bool) and span > 0):
            raise ValueError(f"fbank_centering_span must be None or a positive number of seconds, got {span!r}")

    @property
    def fbank_center_kernel(self) -> int:
        """0 = the mean over all frames is subtracted (`fbank_centering_span=None`); otherwise the ODD number of
        frames of the running mean, F.avg_pool1d(kernel 2 (k // 2) + 1, stride 1, padding k // 2,
        count_include_pad=False) with k = frames of `fbank_centering_span` seconds (wespeaker/__init__.py:141-157)"""
        span = self.hparams.get("fbank_centering_span", None)
        if span is None:
            return 0
        sr = int(self.hparams.get("sample_rate", 16000))
        window = int(sr * float(self.hparams.get("frame_length", 25.0)) * 0.001)
        step = int(sr * float(self.hparams.get("frame_shift", 10.0)) * 0.001)
        k = multi_conv_num_frames(int(span * sr), [window], [step], [0], [1])
        return 2 * (max(k, 0) // 2) + 1

    @property
    def dimension(self) -> int:
        return int(self._state_dict["resnet.seg_1.weight"].shape[0])

    @property
    def _bottleneck(self) -> bool:
        return "resnet.layer1.0.conv3.weight" in self._state_dict

    @property
    def _num_blocks(self):
        return [1 + max(int(k.split(".")[2]) for k in self._state_dict if k.startswith(f"resnet.layer{l}."))
                for l in (1, 2, 3, 4)]

    def _conv_chain(self):
        """(kernel, stride, padding) of every convolution along the time axis: stem, then per block
        3,3 (BasicBlock, resnet.py:108-133) or 1,3,1 (Bottleneck, :181-203), the block's stride on its
        first 3x3"""
        ks, ss, ps = [3], [1], [1]
        for n, s in zip(self._num_blocks, (1, 2, 2, 2)):
            for i in range(n):
                st = s if i == 0 else 1
                if self._bottleneck:
                    ks += [1, 3, 1]
                    ss += [1, st, 1]
                    ps += [0, 1, 0]
                else:
                    ks += [3, 3]
                    ss += [st, 1]
                    ps += [1, 1]
        return ks, ss, ps

    def num_frames(self, num_samples: int) -> int:
        t = multi_conv_num_frames(num_samples, [400], [160], [0], [1])
        for s in (1, 1, 2, 2, 2):  # conv1 + the strided 3x3 of each layer (pad 1)
            t = (t + 2 - 3) // s + 1
        return t

    def receptive_field_size(self, num_frames: int = 1) -> int:
        ks, ss, ps = self._conv_chain()
        size = multi_conv_receptive_field_size(num_frames, ks, ss, ps, [1] * len(ks))
        return multi_conv_receptive_field_size(size, [400], [160], [0], [1])

    def receptive_field_center(self, frame: int = 0) -> int:
        ks, ss, ps = self._conv_chain()
        c = multi_conv_receptive_field_center(frame, ks, ss, ps, [1] * len(ks))
        return multi_conv_receptive_field_center(c, [400], [160], [0], [1])

    def _build_engine(self, device):
        from .embedding import EmbeddingEngine
        from .weights import EmbeddingPack
        return EmbeddingEngine(EmbeddingPack(self._state_dict, device, sample_rate=self.audio.sample_rate,
                                             center_kernel=self.fbank_center_kernel))

    def __call__(self, waveforms: torch.Tensor, weights: Optional[torch.Tensor] = None) -> torch.Tensor:
        return self.engine.forward(waveforms, weights)

    forward = __call__


class XVectorSincNet(Model):
    """models/embedding/xvector.py:205-349 over the HIP x-vector engine."""

    ARCHITECTURE = ("pyannote.audio.models.embedding.xvector", "XVectorSincNet")
    _TDNN = ([5, 3, 3, 1, 1], [1, 1, 1, 1, 1], [0, 0, 0, 0, 0], [1, 2, 3, 1, 1])   # kernel, stride, pad, dilation
    _K = [251, 3, 5, 3, 5, 3]

    def _sinc(self):
        s = int((self.hparams.get("sincnet") or {}).get("stride", 10))
        return self._K, [s, 3, 1, 3, 1, 3], [0] * 6, [1] * 6

    @property
    def dimension(self) -> int:
        return int(self._state_dict["embedding.weight"].shape[0])

    def num_frames(self, num_samples: int) -> int:
        return multi_conv_num_frames(multi_conv_num_frames(num_samples, *self._sinc()), *self._TDNN)

    def receptive_field_size(self, num_frames: int = 1) -> int:
        return multi_conv_receptive_field_size(multi_conv_receptive_field_size(num_frames, *self._TDNN),
                                               *self._sinc())

    def receptive_field_center(self, frame: int = 0) -> int:
        return multi_conv_receptive_field_center(multi_conv_receptive_field_center(frame, *self._TDNN),
                                                 *self._sinc())

    def _build_engine(self, device):
        from .embedding import XVectorEngine
        from .weights import XVectorPack
        return XVectorEngine(XVectorPack(self._state_dict, dict(self.hparams), device))

    def __call__(self, waveforms: torch.Tensor, weights: Optional[torch.Tensor] = None) -> torch.Tensor:
        return self.engine.forward(waveforms, weights)

    forward = __call__


class WeSpeakerResNet152(WeSpeakerResNet34):
    """Bottleneck [3, 8, 36, 3] (wespeaker/__init__.py:375-404); depth and block type are read from the
    state-dict keys, so the three Bottleneck classes only differ by the name the checkpoint carries."""
    ARCHITECTURE = ("pyannote.audio.models.embedding.wespeaker", "WeSpeakerResNet152")


class WeSpeakerResNet221(WeSpeakerResNet34):
    ARCHITECTURE = ("pyannote.audio.models.embedding.wespeaker", "WeSpeakerResNet221")


class WeSpeakerResNet293(WeSpeakerResNet34):
    ARCHITECTURE = ("pyannote.audio.models.embedding.wespeaker", "WeSpeakerResNet293")


# ---------------------------------------------------------------------------------------------
# synthetic checkpoints in the reference layout (no pretrained weights are available offline)
# ---------------------------------------------------------------------------------------------
def segmentation_specifications(duration: float = 10.0, powerset: bool = True) -> Specifications:
    """segmentation-3.0: powerset of 3 speakers, at most 2 simultaneously;
    `powerset=False`: the pre-3.0 multi-label form (one sigmoid score per speaker)."""
    if not powerset:
        return Specifications(problem=Problem.MULTI_LABEL_CLASSIFICATION, resolution=Resolution.FRAME,
                              duration=duration, min_duration=None, warm_up=(0.0, 0.0),
                              classes=["speaker#1", "speaker#2", "speaker#3"], permutation_invariant=True)
    return Specifications(problem=Problem.MONO_LABEL_CLASSIFICATION, resolution=Resolution.FRAME,
                          duration=duration, min_duration=None, warm_up=(0.0, 0.0),
                          classes=["speaker#1", "speaker#2", "speaker#3"], powerset_max_classes=2,
                          permutation_invariant=True)


def embedding_specifications() -> Specifications:
    return Specifications(problem=Problem.REPRESENTATION, resolution=Resolution.CHUNK, duration=5.0)
