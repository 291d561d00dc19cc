"""CPU oracle for the speaker-diarization-3.1 hot path  --  TEST INFRASTRUCTURE ONLY.

This package is a CPU restatement (torch-CPU fp32 + numpy + scipy) of the reference
algorithm of pyannote.audio's `SpeakerDiarization` pipeline (segmentation -> counting ->
embedding -> agglomerative clustering -> reconstruction).  It is the *checker*:

  * only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
    import anything from here;
  * the product package (`pyannote-audio_amd/`, importable as `pyannote_audio_amd`)
    never imports it and has no CPU fallback: it raises when the HIP library is absent.

Pinning status: since round 3 the REFERENCE'S OWN CODE is executed against this
package in the build container (tests/refharness.py loads it from /root/reference/src with stand-ins for the
absent third-party packages): tests/test_oracle_vs_reference_files.py, tests/test_reference_pipeline.py -- the
whole SpeakerDiarization.apply, the loaders, PyanNet, WeSpeaker ResNets, XVectorSincNet, SSeRiouSS, AHC / VBx
clustering and binarize are bit-identical to the functions here.  What stays unpinned are the third-party
stand-ins themselves:

  * torch `nn.LSTM / Conv1d / Conv2d / InstanceNorm1d / BatchNorm2d / MaxPool1d / Linear /
    LogSoftmax / F.interpolate` and scipy `linkage / fcluster / cdist` ARE the code the
    reference executes -> exact for those.
  * StatsPool, AgglomerativeClustering.cluster and Powerset are pinned by the reference's
    own known-answer tests (tests/test_stats_pool.py, tests/test_clustering.py,
    tests/utils/test_powerset.py -> tests/test_oracle_kats.py here).
  * `asteroid_filterbanks.ParamSincFB` (0.4.0) and `torchaudio.compliance.kaldi.fbank`
    (2.10.0) live in third-party packages that are neither vendored in /root/reference nor
    installed here: they are restated from their published algorithm  ==> PARITY UNPINNED
    for those two functions (and therefore for end-to-end numbers that depend on them); independent
    implementations bound them: tests/test_oracle_sincnet_pin.py, tests/test_oracle_fbank_pin.py.
  * torchaudio's wav2vec 2.0 / WavLM encoder (oracle/wav2vec2.py) is pinned layer by layer to HuggingFace
    transformers (tests/test_oracle_wav2vec2_pin.py).
  * `pyannote.core` (Segment / SlidingWindow / closest_frame) is restated from its published
    semantics; nothing in the reference's tests pins it numerically.

There is no C restatement: the reference path is 100 % Python (no native sources to compile),
so `oracle/_ref/` is not built for this project.
"""

from .models import (  # noqa: F401
    ParamSincFB,
    SincNet,
    PyanNet,
    StatsPool,
    WeSpeakerResNet34,
    kaldi_fbank,
    Powerset,
    seeded_pyannet,
    seeded_wespeaker,
    XVectorSincNet,
    seeded_xvector,
    SSeRiouSS,
    seeded_sseriouss,
)
