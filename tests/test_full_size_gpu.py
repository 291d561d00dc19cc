"""Full-size (BASELINE.json configs[3]: 1 hour, 3 591 chunks) checks through size-independent
properties + the oracle's host stages, and the edge cases of the reference's chunking
(core/inference.py:244-278: file shorter than a chunk, ragged zero-padded last chunk)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hour_run(pipeline_dir, gpu_device):
    """one full pipeline pass over a synthetic 1-hour conversation, artifacts kept."""
    import copy
    import pyannote_audio_amd as pa
    from bench import synth_hour
    pipeline = pa.Pipeline.from_pretrained(pipeline_dir).to(gpu_device)
    wav = synth_hour(1.0, seed=11, device=gpu_device)
    art = {}

    def hook(step, artifact, file=None, total=None, completed=None):
        if artifact is not None and total is None:
            art[step] = copy.deepcopy(artifact)

    out = pipeline({"waveform": wav, "sample_rate": 16000, "uri": "hour"}, hook=hook)
    return pipeline, wav, art, out


def test_hour_batch_invariance(hour_run, gpu_device):
    """a chunk's segmentation / embedding does not depend on which launch group it is computed in:
    recomputing scattered chunks of the 1-hour run alone (batch of 1..3) is BIT-identical."""
    pipeline, wav, art, _ = hour_run
    seg_eng = pipeline._segmentation.model.engine
    emb_eng = pipeline._embedding.model_.engine
    seg = art["segmentation"].data            # (3591, 589, 3)
    emb = art["embeddings"]                   # (3591, 3, 256)
    C = seg.shape[0]
    assert C == 3591 and seg.shape[1:] == (589, 3) and emb.shape == (C, 3, 256)
    w = wav.view(-1)
    rng = np.random.default_rng(0)
    from pyannote_audio_amd import frames as fo
    dev_seg = fo.as_device_segmentation(seg, gpu_device)
    _, clean = fo.chunk_stats(dev_seg)
    min_frames = int(np.ceil(589 * pipeline._embedding.min_num_samples / 160000))
    masks = fo.embedding_masks(dev_seg, clean, True, min_frames)
    for c in [0, 1, C - 1] + list(rng.integers(2, C - 1, 12)):
        c = int(c)
        n = int(rng.integers(1, 4)) if c + 3 < C else 1
        _, ml = seg_eng.forward_strided(w[c * 16000:], 16000, n, 160000, want_logp=False)
        assert np.array_equal(ml.cpu().numpy().astype(np.float32), seg[c:c + n]), c
        e = emb_eng.forward_strided(w[c * 16000:], 16000, n, 160000, masks[c:c + n].contiguous())
        assert np.array_equal(e.cpu().numpy(), emb[c:c + n]), c


def test_hour_host_stages_match_oracle(hour_run):
    """given the GPU's segmentations and embeddings of the full hour, counting, clustering (10 773
    embeddings, GPU pdist + GPU linkage vs SciPy) and reconstruction equal the oracle's exactly."""
    from oracle import pipeline as op
    pipeline, _, art, out = hour_run
    seg = art["segmentation"].data
    emb = art["embeddings"]
    chunks, frames = op.SW(0.0, 10.0, 1.0), op.SW(0.0, 0.0619375, 0.016875)
    count, _ = op.speaker_count(seg, chunks, frames)
    assert count.shape[0] == 213334
    assert np.array_equal(art["speaker_counting"].data, count)
    hard, _, centroids = op.clustering(emb.copy(), seg, min_clusters=1, max_clusters=np.inf,
                                       method="centroid", threshold=0.7045654963945799,
                                       min_cluster_size=12)
    hard[np.sum(seg, axis=1) == 0] = -2
    want = op.reconstruct(seg, chunks, hard, count.astype(np.int8), frames)
    got = art["discrete_diarization"].data
    assert got.shape == want.shape
    assert np.array_equal(got, want)
    tracks = op.binarize(want, frames)
    got_tracks = [(s.start, s.end) for s, _ in out.speaker_diarization.itertracks()]
    assert sorted(got_tracks) == sorted((a, b) for a, b, _, _ in tracks)


@pytest.mark.parametrize("seconds", [3.0, 10.0, 10.5, 19.99])
def test_short_and_ragged_files_match_oracle(pipeline_dir, synthetic_models, gpu_device, seconds):
    """< 1 chunk (one zero-padded chunk), exactly 1 chunk, 1 chunk + ragged tail, 2 chunks - 1 sample
    short of a third hop."""
    import pyannote_audio_amd as pa
    from oracle.pipeline import diarize
    from oracle.synthetic import synth_conversation
    seg_o, emb_o = synthetic_models
    wav, _ = synth_conversation(seconds, seed=21)
    pipeline = pa.Pipeline.from_pretrained(pipeline_dir).to(gpu_device)
    art = {}
    out = pipeline({"waveform": wav, "sample_rate": 16000, "uri": "short"},
                   hook=lambda step, a, **kw: art.__setitem__(step, a) if kw.get("total") is None else None)
    ref = diarize(seg_o, emb_o, wav, exclude_overlap=True)
    assert art["segmentation"].data.shape == ref.segmentations.shape
    assert np.array_equal(art["segmentation"].data, ref.segmentations)
    got = [(s.start, s.end, l) for s, _, l in out.speaker_diarization.itertracks(yield_label=True)]
    assert got == ref.diarization
    gotx = [(s.start, s.end, l)
            for s, _, l in out.exclusive_speaker_diarization.itertracks(yield_label=True)]
    assert gotx == ref.exclusive_diarization
