"""GPU parity of XVectorSincNet (models/embedding/xvector.py:205-349) -- SincNet
front end, 5 dilated TDNN layers as chained MFMA GEMMs with the BatchNorms folded forward, weighted statistics
pooling over the (tile, t, b16) rows, embedding Linear -- against the oracle module, which
tests/test_reference_pipeline.py pins bit for bit to the reference's class.  Tolerance: the north_star float
contract for embeddings, |d| <= 1e-5 + 1e-4 |ref|."""
import os

import numpy as np
import pytest
import torch

from conftest import north_star_ratio

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def xvec(gpu_device):
    from oracle import seeded_xvector
    from pyannote_audio_amd.embedding import XVectorEngine
    from pyannote_audio_amd.weights import XVectorPack
    model = seeded_xvector()
    pack = XVectorPack(model.state_dict(), {"sincnet": {"stride": 10}}, gpu_device)
    return model, XVectorEngine(pack)


@pytest.mark.parametrize("B,N,S", [(5, 48000, 3), (19, 80000, 2), (3, 160000, 3), (2, 4771, 1)])
def test_xvector_forward(xvec, gpu_device, B, N, S):
    model, eng = xvec
    g = torch.Generator().manual_seed(B * 7 + S)
    wav = (0.1 * torch.randn(B, 1, N, generator=g)).clamp(-1, 1)
    Fm = max(1, eng.num_pool_frames(N) + 14)                   # the segmentation resolution (SincNet frames)
    weights = (torch.rand(B, S, Fm, generator=g) < 0.6).float()
    weights[0, 0] = 0.0                                         # an all-zero mask: mean = std = 0 -> bias
    with torch.inference_mode():
        want = torch.stack([model(wav, weights=weights[:, s]) for s in range(S)], dim=1)
        want_u = model(wav) if eng.num_pool_frames(N) > 1 else None
    got = eng.forward(wav.to(gpu_device), weights.to(gpu_device))
    assert got.shape == (B, S, 512)
    assert north_star_ratio(f"xvector_B{B}_N{N}", got, want) <= 1.0
    if want_u is not None:
        got_u = eng.forward(wav.to(gpu_device))
        assert got_u.shape == (B, 512)
        assert north_star_ratio(f"xvector_unweighted_N{N}", got_u, want_u) <= 1.0


def test_xvector_strided_chunks_and_too_short(xvec, gpu_device):
    model, eng = xvec
    g = torch.Generator().manual_seed(1)
    total, N, step, C = 16000 * 14 - 3000, 80000, 16000, 10     # last chunk zero padded by 3000 samples
    wav = (0.1 * torch.randn(total, generator=g)).clamp(-1, 1)
    chunks = torch.zeros(C, 1, N)
    for c in range(C):
        seg = wav[c * step: c * step + N]
        chunks[c, 0, :seg.numel()] = seg
    masks = (torch.rand(C, 3, 293, generator=g) < 0.7).float()
    with torch.inference_mode():
        want = torch.stack([model(chunks, weights=masks[:, s]) for s in range(3)], dim=1)
    got = eng.forward_strided(wav.to(gpu_device), step, C, N, masks.to(gpu_device))
    assert north_star_ratio("xvector_strided", got, want) <= 1.0
    # chunks whose masks are all empty skip the backbone (SpeakerDiarization._embed_speech_chunks): same rows, bit for bit
    from types import SimpleNamespace
    from pyannote_audio_amd.speaker_diarization import SpeakerDiarization
    sparse = masks.clone()
    sparse[[0, 3, 4, C - 1]] = 0.0
    wd, md = wav.to(gpu_device), sparse.to(gpu_device)
    me = SimpleNamespace(skip_inactive_chunks=True, last_embedded_chunks=(0, 0))
    short = SpeakerDiarization._embed_speech_chunks(me, eng, wd, step, C, N, md)
    assert me.last_embedded_chunks == (C, C - 4 + 1)
    assert torch.equal(short, eng.forward_strided(wd, step, C, N, md))
    assert eng.num_pool_frames(4770) == 0 and eng.num_pool_frames(4771) == 1
    with pytest.raises(ValueError):
        eng.forward(torch.zeros(1, 1, 4770, device=gpu_device))


def test_pipeline_with_xvector_embeddings(synthetic_models, gpu_device, tmp_path):
    """the diarization pipeline with an XVectorSincNet checkpoint as `embedding`: loader (architecture
    class from the checkpoint), wrapper properties (dimension 512, min_num_samples 4771 = the reference's
    bisection), all stages against the oracle."""
    import pyannote_audio_amd as pa
    from conftest import write_pipeline_dir
    from oracle import seeded_xvector
    from oracle.pipeline import diarize
    from oracle.synthetic import synth_conversation
    seg_o, _ = synthetic_models
    emb_o = seeded_xvector()
    write_pipeline_dir(tmp_path, seg_o, emb_o)
    pipeline = pa.Pipeline.from_pretrained(str(tmp_path)).to(gpu_device)
    assert isinstance(pipeline._embedding.model_, pa.XVectorSincNet)
    assert pipeline._embedding.dimension == 512 and pipeline._embedding.min_num_samples == 4771
    conv, _ = synth_conversation(26.0, seed=4)
    seen = {}

    def hook(name, artefact, file=None, **kw):
        if artefact is not None and kw.get("total") is None:
            seen[name] = np.array(getattr(artefact, "data", artefact), copy=True)

    out = pipeline({"waveform": conv, "sample_rate": 16000, "uri": "conv"}, hook=hook)
    want = diarize(seg_o, emb_o, conv, exclude_overlap=True, min_num_samples=4771)
    assert np.array_equal(seen["segmentation"], want.segmentations)
    assert north_star_ratio("xvector_pipeline_embeddings", seen["embeddings"], want.embeddings) <= 1.0
    got = [(s.start, s.end, l) for s, _, l in out.speaker_diarization.itertracks(yield_label=True)]
    assert got == want.diarization
    assert out.speaker_embeddings.shape[1] == 512
