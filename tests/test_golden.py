"""Golden vectors.

tests/golden/reference_v1.npz -- outputs of the REFERENCE'S OWN CODE (pyannote.audio 4.0.7 loaded from
/root/reference/src by tests/refharness.py; generator: tests/golden/make_reference_golden.py) on seeded
checkpoints and inputs: PyanNet log-probs, WeSpeaker embeddings, the reference's 30 s fixture through
`Model.forward`, and every stage of `SpeakerDiarization.apply` on a synthetic conversation and on sample.wav.
  CPU: the oracle reproduces them (here the reference cannot be loaded on the GPU box, the file can).
  GPU: the HIP path reproduces them with no oracle in the loop (oracle.* supplies WEIGHTS only).
tests/golden/oracle_v1.npz (oracle-generated, round 1) stays as a drift guard for the fbank restatement.

Float tolerance = BASELINE.json north_star / SURVEY.md section 8d: |d| <= 1e-5 + 1e-4 |ref| element-wise;
hard decisions identical wherever the reference's top-2 log-prob gap exceeds 1e-4 (count logged); counts,
discrete diarization and turns identical."""
import os

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden", "oracle_v1.npz")
REF = os.path.join(HERE, "golden", "reference_v1.npz")
SAMPLE = os.path.join(HERE, "golden", "sample.wav")
RTOL, ATOL, GAP = 1e-4, 1e-5, 1e-4


def _inputs():
    g = torch.Generator().manual_seed(7)
    wav = (0.1 * torch.randn(2, 1, 160000, generator=g)).clamp(-1, 1)
    masks = (torch.rand(2, 589, generator=g) < 0.7).float()
    return wav, masks


def _sample():
    from pyannote_audio_amd.audio import Audio
    return Audio(16000, mono="downmix")(SAMPLE)[0]


def _ratio(got, want):
    got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    return float((np.abs(got - want) / (ATOL + RTOL * np.abs(want))).max())


def _log(line):
    os.makedirs(os.path.join(os.path.dirname(HERE), "gpurun_out"), exist_ok=True)
    with open(os.path.join(os.path.dirname(HERE), "gpurun_out", "parity.log"), "a") as fp:
        fp.write(line + "\n")
    print(line)


def _readout_models(ref):
    from oracle.synthetic import models_from_readout
    return models_from_readout(ref["readout_classifier_weight"], ref["readout_classifier_bias"],
                               ref["readout_seg1_bias"])


def _turn_table(annotation):
    labels = annotation.labels()
    return np.array([(s.start, s.end, labels.index(l)) for s, _, l in annotation.itertracks(yield_label=True)],
                    dtype=np.float64).reshape(-1, 3)


# ------------------------------------------------------------------------------------ CPU: the oracle
def test_oracle_reproduces_reference_golden():
    from oracle import seeded_pyannet, seeded_wespeaker
    ref = np.load(REF)
    wav, masks = _inputs()
    seg, emb = seeded_pyannet(seed=1234, num_layers=4), seeded_wespeaker(seed=4321)
    with torch.inference_mode():
        logp = seg(wav).numpy()[:, ::19]
        e = emb(wav[:, :, :48000], weights=masks).numpy()
        sample_logp = seg(_sample().unfold(1, 160000, 16000).permute(1, 0, 2)).numpy()[:, ::7]
    # (thread count / SIMD width of the host may re-associate fp32 sums: tolerance, not bit equality)
    assert _ratio(logp, ref["seg_logp"]) <= 1.0
    assert _ratio(e, ref["embeddings"]) <= 1.0
    assert _ratio(sample_logp, ref["sample_logp"]) <= 1.0


def test_oracle_pipeline_reproduces_reference_golden():
    from oracle.pipeline import diarize
    from oracle.synthetic import synth_conversation
    ref = np.load(REF)
    seg, emb = _readout_models(ref)
    conv, _ = synth_conversation(24.0, seed=3)
    out = diarize(seg, emb, conv, exclude_overlap=True)
    safe = ref["pipeline_top2_gap"] > GAP
    assert np.array_equal(out.segmentations.astype(np.uint8)[safe], ref["pipeline_segmentation"][safe])
    assert np.array_equal(out.count.reshape(-1).astype(np.uint8), ref["pipeline_count"])
    assert _ratio(out.embeddings, ref["pipeline_embeddings"]) <= 1.0
    labels = sorted({l for _, _, l in out.diarization})
    turns = np.array([(s, t, labels.index(l)) for s, t, l in out.diarization], dtype=np.float64)
    assert np.array_equal(turns, ref["pipeline_turns"])
    xturns = np.array([(s, t, labels.index(l)) for s, t, l in out.exclusive_diarization], dtype=np.float64)
    assert np.array_equal(xturns, ref["pipeline_exclusive_turns"])


def test_oracle_reproduces_golden():
    """round-1 oracle-generated vectors: drift guard (this is where the fbank restatement is sampled)"""
    from oracle import kaldi_fbank, seeded_pyannet, seeded_wespeaker
    gold = np.load(GOLD)
    wav, masks = _inputs()
    with torch.inference_mode():
        logp = seeded_pyannet(seed=1234, num_layers=4)(wav).numpy()[:, ::19]
        fb = kaldi_fbank(wav[0, :, :48000] * 32768.0).numpy()[::23]
        e = seeded_wespeaker(seed=4321)(wav[:, :, :48000], weights=masks).numpy()
    assert np.allclose(logp, gold["seg_logp"], rtol=1e-4, atol=1e-5)
    assert np.allclose(fb, gold["fbank"], rtol=1e-4, atol=2e-4)
    assert np.abs(e - gold["embeddings"]).max() <= 1e-4 * np.abs(gold["embeddings"]).max()


def test_reference_and_oracle_golden_agree():
    """the two golden files were made a round apart, one by the oracle and one by the reference's code"""
    gold, ref = np.load(GOLD), np.load(REF)
    assert _ratio(gold["seg_logp"], ref["seg_logp"]) <= 1.0
    assert _ratio(gold["embeddings"], ref["embeddings"]) <= 1.0


# ------------------------------------------------------------------------------------ GPU: the HIP path
@pytest.mark.gpu
def test_hip_path_reproduces_golden(gpu_device):
    """segmentation log-probs and embeddings of the seeded checkpoints, and BASELINE configs[0] (the
    reference's sample.wav, real speech) -- all within rtol 1e-4 / atol 1e-5 of the REFERENCE's output."""
    from oracle import seeded_pyannet, seeded_wespeaker   # weights only; nothing is run on the CPU
    from pyannote_audio_amd.embedding import EmbeddingEngine
    from pyannote_audio_amd.segmentation import SegmentationEngine
    from pyannote_audio_amd.weights import EmbeddingPack, SegmentationPack
    ref = np.load(REF)
    wav, masks = _inputs()
    seg = SegmentationEngine(SegmentationPack(seeded_pyannet(seed=1234, num_layers=4).state_dict(),
                                              {"lstm": {"num_layers": 4}}, 7, 3, 2, gpu_device))
    emb = EmbeddingEngine(EmbeddingPack(seeded_wespeaker(seed=4321).state_dict(), gpu_device))
    logp = seg.forward(wav.to(gpu_device)).cpu().numpy()[:, ::19]
    e = emb.forward(wav[:, :, :48000].to(gpu_device), weights=masks.to(gpu_device)).cpu().numpy()
    r_seg, r_emb = _ratio(logp, ref["seg_logp"]), _ratio(e, ref["embeddings"])
    sample = _sample().to(gpu_device).view(-1)
    logp_s, _ = seg.forward_strided(sample, 16000, 21, 160000)
    r_sample = _ratio(logp_s.cpu().numpy()[:, ::7], ref["sample_logp"])
    _log(f"reference_golden: north-star ratios seg {r_seg:.3f} emb {r_emb:.3f} sample.wav {r_sample:.3f}")
    assert r_seg <= 1.0 and r_emb <= 1.0 and r_sample <= 1.0


@pytest.mark.gpu
def test_hip_pipeline_reproduces_reference_golden(gpu_device, tmp_path):
    """Every stage of the reference's `SpeakerDiarization.apply` on the synthetic conversation, and the
    reference's hard segmentation + turns on sample.wav, from the HIP pipeline."""
    import pyannote_audio_amd as pa
    from conftest import write_pipeline_dir
    from oracle.synthetic import synth_conversation        # input synthesis only
    ref = np.load(REF)
    write_pipeline_dir(tmp_path, *_readout_models(ref))
    pipeline = pa.Pipeline.from_pretrained(str(tmp_path)).to(gpu_device)
    conv, _ = synth_conversation(24.0, seed=3)
    seen = {}

    def hook(name, artefact, file=None, **kw):
        if artefact is not None:
            seen[name] = np.array(getattr(artefact, "data", artefact), copy=True)

    out = pipeline({"waveform": conv, "sample_rate": 16000, "uri": "conv"}, hook=hook)
    safe = ref["pipeline_top2_gap"] > GAP
    mism = (seen["segmentation"].astype(np.uint8) != ref["pipeline_segmentation"]).any(axis=-1)
    _log(f"reference_golden pipeline: hard-decision mismatches {int(mism.sum())} of {mism.size} frames, "
         f"{int((mism & safe).sum())} outside the {GAP:g} top-2 gap")
    assert not (mism & safe).any()
    assert np.array_equal(seen["speaker_counting"].reshape(-1).astype(np.uint8), ref["pipeline_count"])
    r = _ratio(seen["embeddings"], ref["pipeline_embeddings"])
    _log(f"reference_golden pipeline: embeddings north-star ratio {r:.3f}")
    assert r <= 1.0
    assert np.array_equal(seen["discrete_diarization"].astype(np.uint8), ref["pipeline_discrete"])
    assert np.array_equal(_turn_table(out.speaker_diarization), ref["pipeline_turns"])
    assert np.array_equal(_turn_table(out.exclusive_speaker_diarization), ref["pipeline_exclusive_turns"])
    assert _ratio(out.speaker_embeddings, ref["pipeline_centroids"]) <= 1.0

    out = pipeline(SAMPLE, hook=hook)
    safe = ref["sample_top2_gap"] > GAP
    mism = (seen["segmentation"].astype(np.uint8) != ref["sample_segmentation"]).any(axis=-1)
    _log(f"reference_golden sample.wav: hard-decision mismatches {int(mism.sum())} of {mism.size} frames, "
         f"{int((mism & safe).sum())} outside the {GAP:g} top-2 gap")
    assert not (mism & safe).any()
    if not mism.any():
        assert np.array_equal(_turn_table(out.speaker_diarization), ref["sample_turns"])
