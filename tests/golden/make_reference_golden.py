"""Regenerates tests/golden/reference_v1.npz: outputs of the REFERENCE'S OWN CODE on seeded inputs.

pyannote.audio 4.0.7 is loaded from /root/reference/src by tests/refharness.py (stand-ins only for the
absent third-party packages: lightning, pyannote.core / .pipeline, asteroid_filterbanks and
torchaudio.compliance.kaldi.fbank -- see that file).  What runs here is the reference's `Model.from_pretrained`,
`PyanNet`, `WeSpeakerResNet34`, `Inference` and `SpeakerDiarization.apply` on reference-format checkpoints
with seeded weights (no pretrained weights exist offline).  /root/reference exists only in the build
container: the GPU box compares the HIP path with this file (tests/test_golden.py).

Run from the repository root:  python tests/golden/make_reference_golden.py"""
import os
import sys
import tempfile
import warnings

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("PYANNOTE_SKIP_DEPENDENCY_CHECK", "1")

import refharness  # noqa: E402
from conftest import write_pipeline_dir  # noqa: E402
from oracle import seeded_pyannet, seeded_wespeaker  # noqa: E402  (weights only)
from oracle.synthetic import calibrated_pyannet, calibrated_wespeaker, synth_conversation  # noqa: E402

AHC = {"clustering": {"method": "centroid", "min_cluster_size": 12, "threshold": 0.7045654963945799},
       "segmentation": {"min_duration_off": 0.0}}


def unit_inputs():
    g = torch.Generator().manual_seed(7)
    wav = (0.1 * torch.randn(2, 1, 160000, generator=g)).clamp(-1, 1)
    masks = (torch.rand(2, 589, generator=g) < 0.7).float()
    return wav, masks


def main():
    warnings.simplefilter("ignore")
    torch.manual_seed(0)
    from pyannote_audio_amd.audio import Audio
    sample, _ = Audio(16000, mono="downmix")(os.path.join(ROOT, "tests", "golden", "sample.wav"))
    wav, masks = unit_inputs()
    conv, _ = synth_conversation(24.0, seed=3)
    out = {}
    with refharness.reference_modules(third_party=True) as ref:
        top = ref.load_pipelines()
        SpeakerDiarization = sys.modules["pyannote.audio.pipelines"].SpeakerDiarization

        # --- unit vectors: default-gain seeded checkpoints through the reference's loader + forward
        d = tempfile.mkdtemp()
        write_pipeline_dir(d, seeded_pyannet(seed=1234, num_layers=4), seeded_wespeaker(seed=4321))
        seg = top.Model.from_pretrained(os.path.join(d, "segmentation"))
        emb = top.Model.from_pretrained(os.path.join(d, "embedding"))
        with torch.inference_mode():
            out["seg_logp"] = seg(wav).numpy()[:, ::19]               # every 19th frame of 2 chunks x 7
            out["embeddings"] = emb(wav[:, :, :48000], weights=masks).numpy()
            # BASELINE configs[0]: the reference's 30 s fixture, 21 chunks of 10 s every 1 s
            chunks = sample.unfold(1, 160000, 16000).permute(1, 0, 2)
            out["sample_logp"] = seg(chunks).numpy()[:, ::7]          # (21, 85, 7)

        # --- the whole pipeline on a synthetic conversation, calibrated read-outs stored alongside
        seg_c, emb_c = calibrated_pyannet(calib_seconds=40.0), calibrated_wespeaker(calib_seconds=12.0)
        out["readout_classifier_weight"] = seg_c.classifier.weight.detach().numpy()
        out["readout_classifier_bias"] = seg_c.classifier.bias.detach().numpy()
        out["readout_seg1_bias"] = emb_c.resnet.seg_1.bias.detach().numpy()
        d2 = tempfile.mkdtemp()
        write_pipeline_dir(d2, seg_c, emb_c)
        import oracle.vbx as ov
        ov.synth_plda(os.path.join(d2, "plda"))     # (the constructor loads a PLDA unconditionally, :231)
        pipe = SpeakerDiarization(segmentation=os.path.join(d2, "segmentation"),
                                  embedding=os.path.join(d2, "embedding"), plda=os.path.join(d2, "plda"),
                                  clustering="AgglomerativeClustering", embedding_exclude_overlap=True,
                                  segmentation_batch_size=32, embedding_batch_size=32).instantiate(AHC)
        seen = {}

        def hook(name, artefact, file=None, **kw):
            if artefact is not None:
                seen[name] = np.array(getattr(artefact, "data", artefact), copy=True)

        res = pipe({"waveform": conv, "sample_rate": 16000, "uri": "conv"}, hook=hook)
        out["pipeline_segmentation"] = seen["segmentation"].astype(np.uint8)
        out["pipeline_count"] = seen["speaker_counting"].reshape(-1).astype(np.uint8)
        out["pipeline_embeddings"] = seen["embeddings"].astype(np.float32)
        out["pipeline_discrete"] = seen["discrete_diarization"].astype(np.uint8)
        labels = res.speaker_diarization.labels()
        out["pipeline_turns"] = np.array([(s.start, s.end, labels.index(l)) for s, _, l in
                                          res.speaker_diarization.itertracks(yield_label=True)], dtype=np.float64)
        out["pipeline_exclusive_turns"] = np.array(
            [(s.start, s.end, labels.index(l)) for s, _, l in
             res.exclusive_speaker_diarization.itertracks(yield_label=True)], dtype=np.float64)
        out["pipeline_centroids"] = np.asarray(res.speaker_embeddings, dtype=np.float64)

        # top-2 log-prob gaps of the calibrated model (hard decisions are only defined outside 1e-4)
        seg_cal = top.Model.from_pretrained(os.path.join(d2, "segmentation"))

        def gaps(waveform):
            n = waveform.shape[1]
            c = (n - 160000) // 16000 + 1
            x = waveform.unfold(1, 160000, 16000).permute(1, 0, 2)
            if (n - 160000) % 16000:
                tail = waveform[:, c * 16000:]
                x = torch.cat([x, torch.nn.functional.pad(tail, (0, 160000 - tail.shape[1]))[None]])
            with torch.inference_mode():
                t2 = torch.cat([seg_cal(x[i:i + 8]) for i in range(0, len(x), 8)]).topk(2, dim=-1).values
            return (t2[..., 0] - t2[..., 1]).numpy().astype(np.float32)

        out["pipeline_top2_gap"] = gaps(conv)
        out["sample_top2_gap"] = gaps(sample)
        assert out["pipeline_top2_gap"].shape == out["pipeline_segmentation"].shape[:2]

        # sample.wav through the pipeline as well (real speech)
        res = pipe({"waveform": sample, "sample_rate": 16000, "uri": "sample"}, hook=hook)
        labels = res.speaker_diarization.labels()
        out["sample_segmentation"] = seen["segmentation"].astype(np.uint8)
        out["sample_turns"] = np.array([(s.start, s.end, labels.index(l)) for s, _, l in
                                        res.speaker_diarization.itertracks(yield_label=True)], dtype=np.float64)
    path = os.path.join(ROOT, "tests", "golden", "reference_v1.npz")
    np.savez_compressed(path, **out)
    print("written", path, {k: v.shape for k, v in out.items()}, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
