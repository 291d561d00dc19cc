"""GPU: full product pipeline (Pipeline.from_pretrained -> pipeline(audio)) vs the oracle's
loop-for-loop restatement on the same synthetic conversation and the same synthetic checkpoints.
Bar (BASELINE.json north_star): hard segmentations equal where the top-2 log-prob gap is safe, embeddings
within float tolerance, cluster assignments and output segments identical."""
import numpy as np
import pytest
import torch

from conftest import north_star_ratio, report

pytestmark = pytest.mark.gpu


def test_pdist_cdist_bit_exact_vs_scipy(gpu_device):
    from scipy.spatial.distance import cdist, pdist
    from pyannote_audio_amd import distance
    rng = np.random.default_rng(0)
    for n, d in ((700, 256), (129, 37), (2, 256)):
        X = rng.standard_normal((n, d)).astype(np.float32)
        X /= np.linalg.norm(X, axis=1, keepdims=True)
        got = distance.pdist_euclidean(X, device=gpu_device)
        want = pdist(X, metric="euclidean")
        assert got.dtype == np.float64 and np.array_equal(got, want), f"pdist {n}x{d}"
    A = rng.standard_normal((900, 256)).astype(np.float32)
    A[17] = np.nan
    B = rng.standard_normal((5, 256)).astype(np.float32)
    got = distance.cdist(A, B, metric="cosine", device=gpu_device)
    want = cdist(A, B, metric="cosine")
    assert np.array_equal(got, want, equal_nan=True)


def test_clustering_on_gpu_matches_host(gpu_device):
    import pyannote_audio_amd as pa
    from pyannote_audio_amd.core import SlidingWindow, SlidingWindowFeature
    from test_host_logic import _cluster_data, CHUNKS
    emb, seg = _cluster_data(C=400, D_=256, K=5, seed=9)
    params = {"method": "centroid", "min_cluster_size": 12, "threshold": 0.7045654963945799}
    host = pa.AgglomerativeClustering().instantiate(params).to(torch.device("cpu"))   # explicit SciPy route
    dev = pa.AgglomerativeClustering().instantiate(params).to(gpu_device)
    a = host(embeddings=emb.copy(), segmentations=SlidingWindowFeature(seg, CHUNKS))
    b = dev(embeddings=emb.copy(), segmentations=SlidingWindowFeature(seg, CHUNKS))
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[2], b[2])
    assert np.array_equal(a[1], b[1], equal_nan=True)


@pytest.mark.parametrize("seconds,seed", [(33.0, 5), (27.3, 8)])
def test_full_pipeline_matches_oracle(pipeline_dir, synthetic_models, gpu_device, seconds, seed):
    import pyannote_audio_amd as pa
    from oracle.pipeline import diarize
    from oracle.synthetic import synth_conversation
    seg_o, emb_o = synthetic_models
    wav, _ = synth_conversation(seconds, seed=seed)
    pipeline = pa.Pipeline.from_pretrained(pipeline_dir)
    pipeline.to(gpu_device)
    steps = []
    artifacts = {}

    def hook(step, artifact, file=None, total=None, completed=None):
        steps.append(step)
        if artifact is not None and total is None:
            import copy
            artifacts[step] = copy.deepcopy(artifact)   # `count.data` is re-assigned later (as in the reference)

    out = pipeline({"waveform": wav, "sample_rate": 16000, "uri": "synth"}, hook=hook)
    ref = diarize(seg_o, emb_o, wav, exclude_overlap=True)
    assert isinstance(out, pa.DiarizeOutput)
    for s in ("segmentation", "speaker_counting", "embeddings", "discrete_diarization"):
        assert s in steps
    seg = artifacts["segmentation"].data
    assert seg.shape == ref.segmentations.shape
    mism = int((seg != ref.segmentations).sum())
    with open("gpurun_out/parity.log", "a") as fp:
        fp.write(f"pipeline[{seconds}s]: segmentation mismatching frames = {mism} of {seg.size}\n")
    assert mism == 0
    assert np.array_equal(artifacts["speaker_counting"].data, ref.count)
    assert north_star_ratio(f"pipeline_embeddings_{seconds}", torch.from_numpy(artifacts["embeddings"]),
                            torch.from_numpy(ref.embeddings)) <= 1.0
    got = [(s.start, s.end, l) for s, _, l in out.speaker_diarization.itertracks(yield_label=True)]
    assert got == ref.diarization
    gotx = [(s.start, s.end, l)
            for s, _, l in out.exclusive_speaker_diarization.itertracks(yield_label=True)]
    assert gotx == ref.exclusive_diarization
    assert out.speaker_embeddings.shape == ref.centroids.shape
    assert np.allclose(out.speaker_embeddings, ref.centroids, rtol=1e-4, atol=1e-5)
    ser = out.serialize()
    assert set(ser) == {"diarization", "exclusive_diarization"}
    # legacy=True returns the bare Annotation (speaker_diarization.py:626-627)
    pipeline.legacy = True
    ann = pipeline({"waveform": wav, "sample_rate": 16000, "uri": "synth"})
    assert isinstance(ann, pa.Annotation) and ann.uri == "synth"
    assert "SPEAKER synth 1" in ann.to_rttm()


@pytest.mark.parametrize("clustering", ["AgglomerativeClustering", "VBxClustering"])
def test_verify_checkpoint_tool(pipeline_dir, synthetic_models, gpu_device, tmp_path, clustering):
    """tools/verify_checkpoint.py end to end on the seeded checkpoints and the reference's 30-s fixture: every row of
    its table passes (what the owner of real weights runs on theirs) -- the 3.1 configuration and the community-1 one
    (VBx + PLDA)"""
    import importlib.util
    import json
    import os
    if clustering == "VBxClustering":
        from conftest import write_pipeline_dir
        from oracle.vbx import synth_plda
        pipeline_dir = str(tmp_path / "community")
        write_pipeline_dir(pipeline_dir, *synthetic_models, config_extra={
            "pipeline": {"name": "pyannote.audio.pipelines.SpeakerDiarization",
                         "params": {"clustering": "VBxClustering", "embedding": "$model/embedding",
                                    "embedding_batch_size": 32, "embedding_exclude_overlap": True,
                                    "plda": "$model/plda", "segmentation": "$model/segmentation",
                                    "segmentation_batch_size": 32}},
            "params": {"clustering": {"threshold": 0.6, "Fa": 0.07, "Fb": 0.8},
                       "segmentation": {"min_duration_off": 0.0}}})
        synth_plda(os.path.join(pipeline_dir, "plda"))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("verify_checkpoint", os.path.join(root, "tools", "verify_checkpoint.py"))
    tool = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tool)
    out_json = str(tmp_path / "report.json")
    rc = tool.main([pipeline_dir, os.path.join(root, "tests", "golden", "sample.wav"), "--max-seconds", "22",
                    "--chunks", "3", "--json", out_json])
    rep = json.load(open(out_json))
    with open("gpurun_out/parity.log", "a") as fp:
        for r in rep["rows"]:
            fp.write(f"verify_checkpoint: {r['stage']}: {r['what']}: {r['value']} pass={r['pass']}\n")
    assert rc == 0 and rep["ok"]
    verdicts = [r for r in rep["rows"] if r["pass"] is not None]
    assert len(verdicts) >= 6 and all(r["pass"] for r in verdicts)


def test_silence_returns_empty(pipeline_dir, gpu_device):
    """count == 0 everywhere -> early exit with empty annotations (speaker_diarization.py:617-629)."""
    import pyannote_audio_amd as pa
    pipeline = pa.Pipeline.from_pretrained(pipeline_dir).to(gpu_device)
    seg = pipeline._segmentation
    wav = torch.zeros(1, 16000 * 12)
    # force the "no speaker" class by running on digital silence; if the synthetic model still fires,
    # the test only checks types
    out = pipeline({"waveform": wav, "sample_rate": 16000, "uri": "z"})
    assert isinstance(out, pa.DiarizeOutput)
    assert isinstance(out.speaker_diarization, pa.Annotation)


@pytest.mark.parametrize("n,d,dup,seed", [(2, 8, 0, 0), (3, 8, 0, 1), (50, 16, 0, 2), (257, 256, 0, 3),
                                          (200, 16, 60, 4), (1500, 256, 25, 5), (4000, 64, 0, 6)])
def test_linkage_centroid_bit_exact_vs_scipy(gpu_device, n, d, dup, seed):
    """pa_pdist_f64 + pa_linkage_centroid_f64 == scipy linkage(pdist(X), "centroid"), including exact
    ties produced by duplicated rows (same merge order, same float64 heights)."""
    from scipy.cluster.hierarchy import linkage
    from scipy.spatial.distance import pdist
    from pyannote_audio_amd import distance
    rng = np.random.default_rng(seed)
    centers = rng.standard_normal((5, d))
    X = (centers[rng.integers(0, 5, n)] + 0.4 * rng.standard_normal((n, d))).astype(np.float32)
    if dup:
        X[rng.integers(0, n, dup)] = X[rng.integers(0, n, dup)]
    X /= np.linalg.norm(X, axis=1, keepdims=True)
    want = linkage(pdist(X), method="centroid")
    got = distance.linkage_centroid(X, gpu_device)
    assert got.shape == want.shape and got.dtype == np.float64
    bad = np.nonzero((got != want).any(axis=1))[0]
    assert len(bad) == 0, f"first differing merge {bad[0]}: {got[bad[0]]} vs {want[bad[0]]}"


@pytest.mark.parametrize("n,workgroups", [(12300, None), (3000, 4), (12300, 1)])
def test_linkage_multi_workgroup_vs_scipy(gpu_device, n, workgroups, monkeypatch):
    """the heap kernels of csrc/linkage.hip on data WITH exact ties (duplicated rows: the heap-free merge gives up at
    its first pop and the gated heap kernel recomputes the dendrogram): the default at N = 12 300 (one workgroup,
    heap in global memory), a forced 4-workgroup run (k_linkage_centroid_mw, opt-in through PA_LINKAGE_WGS) on a
    small problem, a forced single workgroup -- bit-identical to SciPy."""
    from scipy.cluster.hierarchy import linkage
    from scipy.spatial.distance import pdist
    from pyannote_audio_amd import distance
    if workgroups is not None:
        monkeypatch.setenv("PA_LINKAGE_WGS", str(workgroups))
    rng = np.random.default_rng(n)
    centers = rng.standard_normal((4, 32))
    X = (centers[rng.integers(0, 4, n)] + 0.5 * rng.standard_normal((n, 32))).astype(np.float32)
    X[rng.integers(0, n, 60)] = X[rng.integers(0, n, 60)]
    X /= np.linalg.norm(X, axis=1, keepdims=True)
    want = linkage(pdist(X), method="centroid")
    got = distance.linkage_centroid(X, gpu_device)
    bad = np.nonzero((got != want).any(axis=1))[0]
    assert len(bad) == 0, f"first differing merge {bad[0]}: {got[bad[0]]} vs {want[bad[0]]}"
    st = distance.last_linkage_stats
    assert st[8] == 1 and st[7] == n, "expected: heap-free merge gave up on a tie, heap kernel ran"


@pytest.mark.parametrize("n,d,wgs", [(3, 8, None), (700, 16, None), (4000, 64, None), (7176, 256, None),
                                     (12300, 32, None), (2500, 16, 3), (4000, 64, 2), (7176, 256, 16), (7176, 256, 1),
                                     (12300, 32, 5)])
def test_linkage_heap_free_merge_vs_scipy(gpu_device, n, d, wgs, monkeypatch):
    """csrc/linkage_fast.hip (arg-min over the lower bounds instead of SciPy's heap, EXACT bit per row, square
    matrix, candidate exchange between workgroups) on tie-free data: it completes the dendrogram by itself (status
    0, the heap kernel returns at its gate) and the result is bit-identical to SciPy.  N = 7 176 is one audio-hour
    (8 workgroups by default), 12 300 takes 16, up to 1 024 rows one; `wgs` forces a number of workgroups."""
    from scipy.cluster.hierarchy import linkage
    from scipy.spatial.distance import pdist
    from pyannote_audio_amd import distance
    if wgs is not None:
        monkeypatch.setenv("PA_LINKAGE_FAST_WGS", str(wgs))
    rng = np.random.default_rng(n)
    centers = rng.standard_normal((4, d))
    X = (centers[rng.integers(0, 4, n)] + 0.5 * rng.standard_normal((n, d))).astype(np.float32)
    X /= np.linalg.norm(X, axis=1, keepdims=True)
    want = linkage(pdist(X), method="centroid")
    got = distance.linkage_centroid(X, gpu_device)
    st = distance.last_linkage_stats
    bad = np.nonzero((got != want).any(axis=1))[0]
    assert len(bad) == 0, f"first differing merge {bad[0]}: {got[bad[0]]} vs {want[bad[0]]} (status {st[8]})"
    assert st[8] == 0 and st[7] == 0, f"heap-free merge did not complete: status {st[8]}, heap kernel n {st[7]}"
    wanted = wgs if wgs is not None else (1 if n <= 1024 else (8 if n < 12000 else 16))
    assert st[13] == min(wanted, -(-n // 1024))      # never more workgroups than chunks of 1 024 rows


@pytest.mark.parametrize("wgs", [None, 4])
def test_linkage_late_tie_falls_back_to_the_heap(gpu_device, wgs, monkeypatch):
    """mirrored pairs (u, v), (-u, -v) have the same float64 distance: they meet as the two smallest lower bounds at
    merge 72 (tests/linkage_model.py), where the heap-free merge gives up in the MIDDLE of the dendrogram and the
    heap kernel recomputes it from the untouched condensed matrix: SciPy's result."""
    from scipy.cluster.hierarchy import linkage
    from scipy.spatial.distance import pdist
    from pyannote_audio_amd import distance
    from linkage_model import late_tie_points
    if wgs is not None:
        monkeypatch.setenv("PA_LINKAGE_FAST_WGS", str(wgs))
    X = late_tie_points()
    want = linkage(pdist(X), method="centroid")
    got = distance.linkage_centroid(X, gpu_device)
    assert np.array_equal(got, want)
    st = distance.last_linkage_stats
    assert st[8] == 1 and st[15] == 72 and st[7] == len(X)   # gave up at merge 72, like the model; the heap ran


def test_linkage_heap_free_merge_under_memory_load(gpu_device):
    """the multi-workgroup merge publishes matrix updates to its peers through relaxed sc1 stores + a mailbox tag
    (csrc/linkage_fast.hip, ORDERING): run it while a second stream saturates HBM with copies -- store
    acknowledgements then take far longer than in an idle chip -- and require SciPy's dendrogram bit for bit,
    completed by the heap-free kernel itself (status 0), three times in a row."""
    import torch
    from scipy.cluster.hierarchy import linkage
    from scipy.spatial.distance import pdist
    from pyannote_audio_amd import distance
    n, d = 12300, 32
    rng = np.random.default_rng(77)
    centers = rng.standard_normal((4, d))
    X = (centers[rng.integers(0, 4, n)] + 0.5 * rng.standard_normal((n, d))).astype(np.float32)
    X /= np.linalg.norm(X, axis=1, keepdims=True)
    want = linkage(pdist(X), method="centroid")
    side = torch.cuda.Stream(device=gpu_device)
    a = torch.empty(1 << 28, dtype=torch.float32, device=gpu_device)   # 1 GB each: far beyond L2 + Infinity Cache
    b = torch.empty_like(a)
    for rep in range(3):
        with torch.cuda.stream(side):
            for _ in range(400):          # ~ 0.3 s of back-to-back 2-GB copies at HBM rate
                b.copy_(a, non_blocking=True)
        got = distance.linkage_centroid(X, gpu_device)
        st = distance.last_linkage_stats
        side.synchronize()
        bad = np.nonzero((got != want).any(axis=1))[0]
        assert len(bad) == 0, f"rep {rep}: first differing merge {bad[0]}: {got[bad[0]]} vs {want[bad[0]]}"
        assert st[8] == 0 and st[13] == 13, f"rep {rep}: status {st[8]}, workgroups {st[13]}"   # (13 chunks of 1 024 rows)


def test_linkage_poll_limit_gives_up_together(gpu_device, monkeypatch):
    """a mailbox poll that runs into its limit (forced: PA_LINKAGE_POLL_MS = 10 ns) ends the heap-free kernel with
    status 3 in EVERY workgroup (the one that gives up first tells the others through its record) and the gated heap
    kernel delivers SciPy's dendrogram; the call returns promptly instead of after one limit per workgroup."""
    import time
    from scipy.cluster.hierarchy import linkage
    from scipy.spatial.distance import pdist
    from pyannote_audio_amd import distance
    monkeypatch.setenv("PA_LINKAGE_POLL_MS", "0.00001")
    n, d = 12300, 32
    rng = np.random.default_rng(78)
    centers = rng.standard_normal((4, d))
    X = (centers[rng.integers(0, 4, n)] + 0.5 * rng.standard_normal((n, d))).astype(np.float32)
    X /= np.linalg.norm(X, axis=1, keepdims=True)
    want = linkage(pdist(X), method="centroid")
    t0 = time.perf_counter()
    got = distance.linkage_centroid(X, gpu_device)
    dt = time.perf_counter() - t0
    st = distance.last_linkage_stats
    assert np.array_equal(got, want)
    assert st[8] == 3 and st[7] == n, f"status {st[8]} (expected 3: gave up), heap kernel n {st[7]}"
    assert dt < 20.0


@pytest.mark.parametrize("n,d,k,seed", [(1, 256, 1, 0), (40, 256, 1, 1), (500, 256, 7, 2), (7176, 256, 50, 3),
                                        (3000, 300, 13, 4), (9, 16, 20, 5)])
def test_centroid_means_bit_exact(gpu_device, n, d, k, seed):
    """pa_centroid_means (north_star: "centroid updates ... run as a HIP kernel"; reference pipelines/clustering.py:
    182-187) == the reference's np.mean(X[labels == c], axis=0) per cluster, bit for bit, on rows selected out of a
    larger device-resident embedding matrix; empty clusters give NaN rows like the host restatement."""
    from pyannote_audio_amd import distance
    from pyannote_audio_amd.clustering import segment_means
    rng = np.random.default_rng(seed)
    total = 3 * n + 5
    X = (rng.standard_normal((total, d)) * rng.uniform(0.1, 30.0, (total, 1))).astype(np.float32)
    rows = np.sort(rng.choice(total, n, replace=False))
    labels = rng.integers(0, k, n)
    if k > 3:
        labels[labels == 2] = 1            # an empty cluster in the middle
    got = distance.centroid_means(torch.from_numpy(X).to(gpu_device), rows, labels, k, gpu_device)
    want = segment_means(X[rows], labels, k)
    for c in range(k):
        sel = X[rows][labels == c]
        if len(sel):
            assert np.array_equal(want[c], np.mean(sel, axis=0))      # the host restatement IS the reference call
    assert got.dtype == np.float32 and got.shape == want.shape
    assert np.array_equal(got, want, equal_nan=True)


def test_non_powerset_pipeline_matches_oracle(synthetic_models, gpu_device, tmp_path):
    """a14: a multi-label segmentation checkpoint -- sigmoid scores out of the
    classifier kernel, hysteresis thresholding (pipelines/speaker_diarization.py:599-606, utils/signal.py:
    78-204) on the device, reconstruction from the raw scores -- against the oracle (itself pinned to the
    reference's pipeline run on the same kind of checkpoint, tests/test_reference_pipeline.py)."""
    import pyannote_audio_amd as pa
    from conftest import north_star_ratio, write_pipeline_dir
    from oracle.pipeline import diarize
    from oracle.synthetic import calibrated_multilabel_pyannet, synth_conversation
    seg_o = calibrated_multilabel_pyannet(calib_seconds=40.0)
    _, emb_o = synthetic_models
    write_pipeline_dir(tmp_path, seg_o, emb_o, powerset=False)
    pipeline = pa.Pipeline.from_pretrained(str(tmp_path)).to(gpu_device)
    assert not pipeline._segmentation.model.specifications.powerset
    conv, _ = synth_conversation(23.0, seed=14)
    seen = {}

    def hook(name, artefact, file=None, **kw):
        if artefact is not None and kw.get("total") is None:
            seen[name] = np.array(getattr(artefact, "data", artefact), copy=True)

    out = pipeline({"waveform": conv, "sample_rate": 16000, "uri": "conv"}, hook=hook)
    want = diarize(seg_o, emb_o, conv, exclude_overlap=True, segmentation_threshold=0.5)
    assert north_star_ratio("non_powerset_scores", seen["segmentation"], want.raw_segmentations) <= 1.0
    assert np.array_equal(seen["speaker_counting"].reshape(-1), want.count.reshape(-1))
    got = [(s.start, s.end, l) for s, _, l in out.speaker_diarization.itertracks(yield_label=True)]
    assert got == want.diarization
    gotx = [(s.start, s.end, l) for s, _, l in out.exclusive_speaker_diarization.itertracks(yield_label=True)]
    assert gotx == want.exclusive_diarization
    # the stand-alone model contract: (B, 1, N) -> (B, F, 3) sigmoid scores
    model = pipeline._segmentation.model
    chunk = conv[:, :160000][None]
    with torch.inference_mode():
        ref = seg_o(chunk)
    assert north_star_ratio("non_powerset_forward", model(chunk.to(gpu_device)), ref) <= 1.0


def test_inactive_chunks_skip_the_backbone_bit_identically(pipeline_dir, gpu_device):
    """chunks whose masks are all empty are left out of the embedding launch (SpeakerDiarization._embed_speech_chunks):
    every row equals the row of the full run, bit for bit -- the kept chunks although they were computed in another
    launch composition, the skipped ones because an empty mask pools to zero whatever the backbone computed"""
    import pyannote_audio_amd as pa
    from oracle.synthetic import synth_conversation
    wav, _ = synth_conversation(41.5, seed=3)            # 33 chunks, the last one runs past the file
    pipeline = pa.Pipeline.from_pretrained(pipeline_dir)
    pipeline.to(gpu_device)
    waveform = wav.to(gpu_device)
    seg = pipeline.get_segmentations({"waveform": wav, "sample_rate": 16000, "uri": "s"}, waveform=waveform)
    dev_seg = pipeline._segmentation.last_device_output.clone()
    C = dev_seg.shape[0]
    assert C >= 30
    silent = [0, 7, 8, 9, 10, 11, 19, C - 1]             # first, last (zero-padded) and interior chunks
    dev_seg[silent] = 0
    for exclude_overlap in (True, False):
        pipeline.skip_inactive_chunks = False
        full, *_ = pipeline._embed(waveform, dev_seg, seg.sliding_window, 0, exclude_overlap, None)
        assert pipeline.last_embedded_chunks == (C, C)
        pipeline.skip_inactive_chunks = True
        short, *_ = pipeline._embed(waveform, dev_seg, seg.sliding_window, 0, exclude_overlap, None)
        kept = int(dev_seg.flatten(1).any(dim=1).sum())
        assert kept <= C - len(silent) and pipeline.last_embedded_chunks == (C, kept + 1)
        assert torch.isfinite(full).all()
        assert torch.equal(full, short)
        # (an empty mask gives the bias of the last Linear: the same row everywhere)
        assert torch.equal(short[silent[0]], short[silent[-1]])
    # nothing to skip: the plain launch
    dev_seg = pipeline._segmentation.last_device_output
    if bool(dev_seg.flatten(1).any(dim=1).all()):
        pipeline._embed(waveform, dev_seg, seg.sliding_window, 0, True, None)
        assert pipeline.last_embedded_chunks == (C, C)


def test_pipeline_over_a_long_pause_equals_the_full_run(pipeline_dir, gpu_device):
    """a recording with 45 s of silence in the middle: the pipeline's outputs with and without the skip are the same
    objects (turns, labels, centroids, the "embeddings" artefact)"""
    import pyannote_audio_amd as pa
    from oracle.synthetic import synth_conversation
    a, _ = synth_conversation(24.0, seed=11)
    b, _ = synth_conversation(21.0, seed=12)
    wav = torch.cat([a, torch.zeros(1, 45 * 16000), b], dim=1)
    pipeline = pa.Pipeline.from_pretrained(pipeline_dir)
    pipeline.to(gpu_device)
    # the seeded read-out was calibrated on conversation only and finds speakers in digital silence (a trained model does
    # not): the chunks that lie inside the pause are forced to "nobody", on the host copy and on the device copy alike
    inference = pipeline._segmentation
    plain_slide = inference.slide

    def slide(*args, **kwargs):
        out = plain_slide(*args, **kwargs)
        out.data[24:60] = 0
        inference.last_device_output[24:60] = 0
        return out
    inference.slide = slide

    def run(skip):
        pipeline.skip_inactive_chunks = skip
        art = {}

        def hook(step, artifact, file=None, total=None, completed=None):
            if artifact is not None and total is None:
                import copy
                art[step] = copy.deepcopy(artifact)
        out = pipeline({"waveform": wav, "sample_rate": 16000, "uri": "pause"}, hook=hook)
        return out, art, pipeline.last_embedded_chunks

    full, art_full, done_full = run(False)
    short, art_short, done_short = run(True)
    assert done_full[0] == done_full[1] and done_short[0] == done_full[0]
    with open("gpurun_out/parity.log", "a") as fp:
        fp.write(f"pipeline[long pause]: chunks through the backbone {done_short[1]} of {done_short[0]}\n")
    assert done_short[1] <= done_short[0] - 35
    assert np.array_equal(art_full["embeddings"], art_short["embeddings"])
    turns = lambda o: [(s.start, s.end, l) for s, _, l in o.itertracks(yield_label=True)]
    assert turns(full.speaker_diarization) == turns(short.speaker_diarization)
    assert turns(full.exclusive_speaker_diarization) == turns(short.exclusive_speaker_diarization)
    assert np.array_equal(full.speaker_embeddings, short.speaker_embeddings)
