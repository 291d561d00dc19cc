"""BASELINE.json configs[0]: `Inference` of the segmentation model over the reference's own 30 s
fixture (src/pyannote/audio/sample/sample.wav, copied to tests/golden/sample.wav; 16 kHz mono PCM16;
sample.rttm = 2 speakers, 10 turns), with the structural checks of the reference's
tests/inference_test.py:59-63 (step > duration -> ValueError) and :94-97 (skip_aggregation -> 3-D
SlidingWindowFeature), then the same file through the full pipeline, HIP path vs the oracle.
Weights are the seeded synthetic checkpoints (no pretrained weights exist offline), so sample.rttm
pins the file format only, not the diarization."""
import os

import numpy as np
import pytest
import torch

from conftest import north_star_ratio

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
SAMPLE = os.path.join(GOLDEN, "sample.wav")


def test_sample_fixture_geometry_cpu():
    """21 chunks of 10 s every 1 s, no orphan chunk; RTTM fixture parses as the wire format."""
    from pyannote_audio_amd.audio import Audio
    from pyannote_audio_amd.inference import Inference
    wav, sr = Audio(16000, mono="downmix")(SAMPLE)
    assert sr == 16000 and tuple(wav.shape) == (1, 480000) and wav.dtype == torch.float32
    assert Inference.num_chunks(480000, 160000, 16000) == (21, False)
    rows = [l.split() for l in open(os.path.join(GOLDEN, "sample.rttm")) if l.strip()]
    assert len(rows) == 10 and {r[0] for r in rows} == {"SPEAKER"} and {r[1] for r in rows} == {"sample"}
    assert len({r[7] for r in rows}) == 2


def test_step_longer_than_duration_raises_cpu(synthetic_models, tmp_path):
    """tests/inference_test.py:59-63"""
    import pyannote_audio_amd as pa
    from conftest import write_pipeline_dir
    write_pipeline_dir(tmp_path, *synthetic_models)
    model = pa.Model.from_pretrained(os.path.join(str(tmp_path), "segmentation"))
    with pytest.raises(ValueError):
        pa.Inference(model, duration=model.specifications.duration,
                     step=model.specifications.duration + 1)


@pytest.mark.gpu
def test_inference_on_sample_wav(pipeline_dir, synthetic_models, gpu_device):
    """configs[0] on the GPU path: SlidingWindowFeature (21, 589, 3), equal to the oracle's
    `Inference.slide` + `Powerset.to_multilabel(hard)`; log-probs within the float tolerance."""
    import pyannote_audio_amd as pa
    from oracle import pipeline as op
    seg_o, _ = synthetic_models
    model = pa.Model.from_pretrained(os.path.join(pipeline_dir, "segmentation")).to(gpu_device)
    inference = pa.Inference(model, skip_aggregation=True)
    swf = inference(SAMPLE)
    assert isinstance(swf, pa.SlidingWindowFeature)
    assert swf.data.ndim == 3 and swf.data.shape == (21, 589, 3)          # inference_test.py:94-97
    assert (swf.sliding_window.duration, swf.sliding_window.step) == (10.0, 1.0)
    wav, sr = pa.Audio(16000, mono="downmix")(SAMPLE)
    ref = op.slide(seg_o, wav, sr, 10.0, 1.0, 32)
    assert ref.shape == (21, 589, 3)
    chunks = wav.unfold(1, 160000, 16000).permute(1, 0, 2)
    # (1) the float contract, outright: a default-gain seeded checkpoint on this real speech must meet
    #     rtol 1e-4 / atol 1e-5 against the float32 oracle (tests/test_golden.py checks the same run against
    #     the REFERENCE-generated vectors).
    from oracle import seeded_pyannet
    from pyannote_audio_amd.segmentation import SegmentationEngine
    from pyannote_audio_amd.weights import SegmentationPack
    plain = seeded_pyannet(seed=1234, num_layers=4)
    eng = SegmentationEngine(SegmentationPack(plain.state_dict(), {"lstm": {"num_layers": 4}}, 7, 3, 2,
                                              gpu_device))
    with torch.inference_mode():
        plain_ref = plain(chunks)
    plain_got, _ = eng.forward_strided(wav.view(-1).to(gpu_device), 16000, 21, 160000)
    assert north_star_ratio("config1_sample_logp_default_gain", plain_got, plain_ref) <= 1.0
    # (2) the calibrated (synthetic, high-gain read-out) checkpoint of the pipeline tests: real speech drives
    #     it so hard that float32 ITSELF is only good to ~1e-2 on its log-probabilities (float32 CPU oracle vs a
    #     float64 evaluation of the same module), so no float32 implementation can meet 1e-4 against another
    #     one there.  What is asserted for it is what the pipeline consumes: hard decisions identical wherever
    #     the oracle's top-2 gap exceeds 1e-4; the errors against float64 are logged.
    with torch.inference_mode():
        ref_logp = seg_o(chunks)
    got_logp = model(chunks.to(gpu_device)).cpu()
    import copy
    with torch.inference_mode():
        ref64 = copy.deepcopy(seg_o).double()(chunks.double())
    err_cpu = (ref_logp.double() - ref64).abs().max().item()
    err_gpu = (got_logp.double() - ref64).abs().max().item()
    top2 = torch.topk(ref_logp, 2, dim=-1).values
    safe = ((top2[..., 0] - top2[..., 1]) > 1e-4).numpy()
    mism = (swf.data != ref).any(axis=-1)
    with open("gpurun_out/parity.log", "a") as fp:
        fp.write(f"config1_sample (calibrated read-out): max|d| vs float64: float32 oracle {err_cpu:.3e}, "
                 f"HIP path {err_gpu:.3e}; hard-decision mismatches {int(mism.sum())} of {mism.size} frames, "
                 f"{int((mism & safe).sum())} outside the 1e-4 top-2 gap\n")
    assert not (mism & safe).any()


@pytest.mark.gpu
def test_full_pipeline_on_sample_wav(pipeline_dir, synthetic_models, gpu_device):
    """path-on-disk front door -> DiarizeOutput, identical turns to the oracle on real speech"""
    import pyannote_audio_amd as pa
    from oracle.pipeline import diarize
    seg_o, emb_o = synthetic_models
    pipeline = pa.Pipeline.from_pretrained(pipeline_dir).to(gpu_device)
    out = pipeline(SAMPLE)
    wav, _ = pa.Audio(16000, mono="downmix")(SAMPLE)
    ref = diarize(seg_o, emb_o, wav, exclude_overlap=True)
    got = [(s.start, s.end, l) for s, _, l in out.speaker_diarization.itertracks(yield_label=True)]
    assert got == ref.diarization
    gotx = [(s.start, s.end, l)
            for s, _, l in out.exclusive_speaker_diarization.itertracks(yield_label=True)]
    assert gotx == ref.exclusive_diarization
    assert out.speaker_diarization.uri == "sample"
    for line in out.speaker_diarization.to_rttm().splitlines():
        f = line.split()
        assert len(f) == 10 and f[0] == "SPEAKER" and f[1] == "sample" and f[2] == "1"
