"""MI355X-native speaker-diarization hot path (drop-in for pyannote.audio's SpeakerDiarization).

Host-side mirror of the reference's Python interfaces over a C-ABI library of hand-written
gfx950 kernels (`libpyannote_amd.so`, declared in include/pyannote_amd.h)."""

__version__ = "0.1.0"
