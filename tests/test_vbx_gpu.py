"""GPU: VBxClustering + PLDA (pipelines/clustering.py:550-669,
utils/vbx.py:27-218, core/plda.py:33-60) against the oracle restatement (oracle/vbx.py).

Float contract: the PLDA projection and the VB iterations are float64 on both sides but sum in a
different order (numpy's BLAS vs fixed reduction trees), so features / responsibilities agree to
~1e-12 relative, NOT bit for bit; cluster ASSIGNMENTS (Hungarian on cosine similarities to the
surviving speakers' centroids) and the number of speakers must be identical."""
import copy
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def plda_dir(tmp_path_factory):
    from oracle.vbx import synth_plda
    return synth_plda(str(tmp_path_factory.mktemp("plda")))


def _embeddings(C, K, noise, seed, D_=256, F=60, S=3):
    rng = np.random.default_rng(seed)
    centers = rng.standard_normal((K, D_))
    who = rng.integers(0, K, size=(C, S))
    emb = (centers[who] + noise * rng.standard_normal((C, S, D_))).astype(np.float32)
    emb[rng.integers(0, C), rng.integers(0, S)] = np.nan
    seg = (rng.uniform(size=(C, F, S)) < 0.45).astype(np.float32)
    seg[rng.integers(0, C, C // 8), :, rng.integers(0, S, C // 8)] = 0.0
    return emb, seg


def test_plda_transform_matches_oracle(plda_dir, gpu_device):
    import pyannote_audio_amd as pa
    from oracle.vbx import PLDA as OraclePLDA
    ref = OraclePLDA(os.path.join(plda_dir, "xvec_transform.npz"), os.path.join(plda_dir, "plda.npz"))
    plda = pa.PLDA.from_pretrained(plda_dir).to(gpu_device)
    assert np.array_equal(plda.phi, ref.phi)
    x = np.random.default_rng(0).standard_normal((700, 256)).astype(np.float32)
    got, want = plda(x), ref(x)
    assert got.shape == want.shape == (700, 128) and got.dtype == np.float64
    assert np.abs(got - want).max() <= 1e-11 * np.abs(want).max()


@pytest.mark.parametrize("C,K,noise,seed,kw", [
    (400, 4, 0.9, 0, {}), (900, 7, 1.2, 1, {}), (250, 3, 0.8, 2, {"num_clusters": 5}),
    (300, 5, 1.0, 3, {"max_clusters": 2}), (60, 2, 0.7, 4, {})])
def test_vbx_clustering_matches_oracle(plda_dir, gpu_device, C, K, noise, seed, kw):
    import pyannote_audio_amd as pa
    from pyannote_audio_amd.core import SlidingWindow, SlidingWindowFeature
    from oracle.vbx import PLDA as OraclePLDA, vbx_clustering
    emb, seg = _embeddings(C, K, noise, seed)
    params = {"threshold": 0.6, "Fa": 0.07, "Fb": 0.8}
    clu = pa.VBxClustering(plda=plda_dir).instantiate(params).to(gpu_device)
    chunks = SlidingWindow(start=0.0, duration=10.0, step=1.0)
    hard, soft, cen = clu(embeddings=emb.copy(), segmentations=SlidingWindowFeature(seg, chunks),
                          min_clusters=kw.get("num_clusters", kw.get("min_clusters", 1)),
                          max_clusters=kw.get("num_clusters", kw.get("max_clusters", np.inf)),
                          num_clusters=kw.get("num_clusters"))
    ref = OraclePLDA(os.path.join(plda_dir, "xvec_transform.npz"), os.path.join(plda_dir, "plda.npz"))
    rh, rs, rc = vbx_clustering(emb.copy(), seg, ref, num_clusters=kw.get("num_clusters"),
                                min_clusters=kw.get("num_clusters", kw.get("min_clusters")),
                                max_clusters=kw.get("num_clusters", kw.get("max_clusters")), **params)
    assert cen.shape == rc.shape, "VBx kept a different number of speakers"
    assert np.allclose(cen, rc, rtol=1e-9, atol=1e-12)
    assert np.array_equal(hard, rh)
    assert np.allclose(soft, rs, rtol=1e-9, atol=1e-12, equal_nan=True)
    with open("gpurun_out/parity.log", "a") as fp:
        fp.write(f"vbx[C={C},K={K}]: {cen.shape[0]} speakers, {clu.timings.get('vbx_iterations')} VB "
                 f"iterations, max|dcentroid| = {np.abs(cen - rc).max():.2e}\n")


def test_pipeline_with_vbx_clustering(tmp_path, synthetic_models, plda_dir, gpu_device):
    """config.yaml with `clustering: VBxClustering` + `plda: $model/plda` (the community-1 layout):
    loader contract + end-to-end equality with the oracle given the pipeline's own embeddings."""
    import shutil
    import pyannote_audio_amd as pa
    from conftest import write_pipeline_dir
    from oracle import pipeline as op
    from oracle.synthetic import synth_conversation
    from oracle.vbx import PLDA as OraclePLDA, vbx_clustering
    root = str(tmp_path / "community")
    write_pipeline_dir(root, *synthetic_models, config_extra={
        "pipeline": {"name": "pyannote.audio.pipelines.SpeakerDiarization",
                     "params": {"clustering": "VBxClustering", "embedding": "$model/embedding",
                                "embedding_batch_size": 32, "embedding_exclude_overlap": True,
                                "plda": "$model/plda", "segmentation": "$model/segmentation",
                                "segmentation_batch_size": 32}},
        "params": {"clustering": {"threshold": 0.6, "Fa": 0.07, "Fb": 0.8},
                   "segmentation": {"min_duration_off": 0.0}}})
    shutil.copytree(plda_dir, os.path.join(root, "plda"))
    pipeline = pa.Pipeline.from_pretrained(root).to(gpu_device)
    assert isinstance(pipeline.clustering, pa.VBxClustering) and pipeline.clustering.Fa == 0.07
    wav, _ = synth_conversation(45.0, seed=17)
    art = {}
    out = pipeline({"waveform": wav, "sample_rate": 16000, "uri": "vbx"},
                   hook=lambda step, a, **kw: art.__setitem__(step, copy.deepcopy(a))
                   if (a is not None and kw.get("total") is None) else None)
    seg, emb = art["segmentation"].data, art["embeddings"]
    ref = OraclePLDA(os.path.join(plda_dir, "xvec_transform.npz"), os.path.join(plda_dir, "plda.npz"))
    hard, _, cen = vbx_clustering(emb.copy(), seg, ref, threshold=0.6, Fa=0.07, Fb=0.8)
    hard = hard.astype(np.int64)
    hard[np.sum(seg, axis=1) == 0] = -2
    chunks, frames = op.SW(0.0, 10.0, 1.0), op.SW(0.0, 0.0619375, 0.016875)
    count, _ = op.speaker_count(seg, chunks, frames)
    want = op.reconstruct(seg, chunks, hard, count.astype(np.int8), frames)
    assert np.array_equal(art["discrete_diarization"].data, want)
    assert out.speaker_embeddings.shape[1] == 256
