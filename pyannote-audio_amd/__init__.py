"""MI355X-native speaker-diarization hot path (drop-in for pyannote.audio's SpeakerDiarization).

Host-side mirror of the reference's Python interfaces over a C-ABI library of hand-written
gfx950 kernels (`libpyannote_amd.so`, declared in include/pyannote_amd.h)."""

__version__ = "0.1.0"

from .core import Annotation, Segment, SlidingWindow, SlidingWindowFeature  # noqa: E402,F401
from .audio import Audio  # noqa: E402,F401
from .model import (Model, PyanNet, SSeRiouSS, WeSpeakerResNet34, WeSpeakerResNet152, WeSpeakerResNet221,  # noqa: E402,F401
                    WeSpeakerResNet293, XVectorSincNet, Specifications, Problem, Resolution)  # noqa: E402,F401
from .pipeline import Pipeline  # noqa: E402,F401
from .inference import Inference  # noqa: E402,F401
from .clustering import (AgglomerativeClustering, Clustering, KMeansClustering, OracleClustering,  # noqa: E402,F401
                         VBxClustering)
from .plda import PLDA  # noqa: E402,F401
from .speaker_verification import PretrainedSpeakerEmbedding  # noqa: E402,F401
from .speaker_diarization import SpeakerDiarization, DiarizeOutput  # noqa: E402,F401
from .voice_activity_detection import VoiceActivityDetection  # noqa: E402,F401
from .hook import ArtifactHook, Hooks, ProgressHook, TimingHook  # noqa: E402,F401
