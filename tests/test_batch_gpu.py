"""GPU: `SpeakerDiarization.apply_batch` (core/pipeline.py:489-508, 570-578).

* pipelined batch == sequential `apply`, file by file (clustering/back end of file i run on a second
  stream while the front end of file i+1 runs);
* duplicate URIs raise ValueError (core/pipeline.py:570-578);
* `joint_clustering=True` (BASELINE.json configs[4]) == the oracle's clustering called on the
  concatenated embeddings, single process and 2 ranks (one file per rank,
  records exchanged with one all-gather; both ranks share the test GPU, the wire is gloo -- on a
  multi-GPU node the same code runs one rank per GPU over RCCL);
* the global-memory heap instantiation of the linkage kernel (N > ~11 600, the size class of joint
  clustering) is bit-identical to SciPy."""
import copy
import os
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _turns(ann):
    return [(s.start, s.end, l) for s, _, l in ann.itertracks(yield_label=True)]


def _files(seconds_seeds):
    from oracle.synthetic import synth_conversation
    return [{"waveform": synth_conversation(sec, seed=seed)[0], "sample_rate": 16000, "uri": f"f{seed}"}
            for sec, seed in seconds_seeds]


def test_apply_batch_equals_sequential(pipeline_dir, gpu_device):
    import pyannote_audio_amd as pa
    pipeline = pa.Pipeline.from_pretrained(pipeline_dir).to(gpu_device)
    files = _files([(33.0, 5), (12.0, 3), (27.3, 8), (3.0, 21)])
    want = [pipeline(f) for f in files]
    got = list(pipeline(files))
    assert [f["uri"] for f, _ in got] == [f["uri"] for f in files]
    for (f, out), ref in zip(got, want):
        assert isinstance(out, pa.DiarizeOutput)
        assert _turns(out.speaker_diarization) == _turns(ref.speaker_diarization), f["uri"]
        assert _turns(out.exclusive_speaker_diarization) == _turns(ref.exclusive_speaker_diarization)
        assert np.array_equal(out.speaker_embeddings, ref.speaker_embeddings)
    with pytest.raises(ValueError, match="distinct URIs"):
        pipeline([files[0], dict(files[1], uri=files[0]["uri"])])
    # without a hook the results are produced one ahead of the consumer in a worker thread (pipelining.run_ahead); with
    # a hook -- the caller's code, called from the main thread (front end) and from the tail thread (back end) as
    # before -- they are not, and the results are the same either way
    import threading
    seen = set()

    def hook(step, artefact, file=None, total=None, completed=None):
        seen.add(threading.current_thread().name)

    hooked = list(pipeline(files, hook=hook))
    assert threading.current_thread().name in seen and "pa-run-ahead" not in seen
    for (f, out), ref in zip(hooked, want):
        assert _turns(out.speaker_diarization) == _turns(ref.speaker_diarization), f["uri"]
    # a consumer that stops after the first result leaves nothing behind that the next call trips over
    it = iter(pipeline(files))
    first = next(it)
    it.close()
    assert _turns(first[1].speaker_diarization) == _turns(want[0].speaker_diarization)
    again = list(pipeline(files))
    assert _turns(again[-1][1].speaker_diarization) == _turns(want[-1].speaker_diarization)


def test_files_on_disk_equal_resident_waveforms(pipeline_dir, gpu_device, tmp_path):
    """WAV files on disk (the reference CLI's input, __main__.py:684-744): the stored 16-bit samples go to the device as
    they are and are scaled there (`Audio.load_on_device`), and `apply_batch` reads file i + 1 in a worker thread while
    the GPU runs file i -- results identical to the same samples handed over as float32 waveforms, for a list, a
    single path, a stereo file and a file object; a missing file raises where the reference raises (validation)."""
    from scipy.io import wavfile
    import pyannote_audio_amd as pa
    from pyannote_audio_amd.audio import Audio
    pipeline = pa.Pipeline.from_pretrained(pipeline_dir).to(gpu_device)
    files = _files([(33.0, 5), (12.0, 3), (27.3, 8)])
    on_disk, resident = [], []
    for f in files:
        pcm = (f["waveform"][0].clamp(-1, 1) * 32767.0).round().to(torch.int16).numpy()
        if f["uri"] == "f3":                                   # one stereo file: down-mixed on the device
            pcm = np.stack([pcm, pcm[::-1]], axis=1)
        path = str(tmp_path / (f["uri"] + ".wav"))
        wavfile.write(path, 16000, pcm)
        on_disk.append({"audio": path, "uri": f["uri"]})
        host = Audio(16000, "downmix")(path)[0]                # the host conversion of the same file
        assert torch.equal(Audio(16000, "downmix").load_on_device(path, gpu_device)[0].cpu(), host)
        resident.append({"waveform": host, "sample_rate": 16000, "uri": f["uri"]})
    want = [out for _, out in pipeline(resident)]
    got = list(pipeline(on_disk))
    assert [f["uri"] for f, _ in got] == [f["uri"] for f in on_disk]
    for (f, out), ref in zip(got, want):
        assert _turns(out.speaker_diarization) == _turns(ref.speaker_diarization), f["uri"]
        assert np.array_equal(out.speaker_embeddings, ref.speaker_embeddings)
    single = pipeline(on_disk[0]["audio"])
    assert _turns(single.speaker_diarization) == _turns(want[0].speaker_diarization)
    with open(on_disk[1]["audio"], "rb") as fp:
        assert _turns(pipeline(fp).speaker_diarization) == _turns(want[1].speaker_diarization)
    with pytest.raises(ValueError, match="does not exist"):
        list(pipeline([on_disk[0], {"audio": str(tmp_path / "missing.wav"), "uri": "m"}]))


def test_apply_batch_honours_per_file_pipeline_kwargs(pipeline_dir, gpu_device):
    """core/pipeline.py:583: every file of a list is applied with ITS `pipeline_kwargs`; a name given both
    per file and per batch is the TypeError a double keyword is."""
    import pyannote_audio_amd as pa
    pipeline = pa.Pipeline.from_pretrained(pipeline_dir).to(gpu_device)
    files = _files([(33.0, 5), (27.3, 8), (21.0, 2)])
    files[0]["pipeline_kwargs"] = {"num_speakers": 1}
    files[1]["pipeline_kwargs"] = {"min_speakers": 3, "max_speakers": 4}
    want = [pipeline(copy.copy(f)) for f in files]
    got = [out for _, out in pipeline(files)]
    for out, ref in zip(got, want):
        assert _turns(out.speaker_diarization) == _turns(ref.speaker_diarization)
    assert len(got[0].speaker_diarization.labels()) == 1
    with pytest.raises(TypeError, match="multiple values"):
        list(pipeline(files, num_speakers=2))


def _joint_reference(art, uris):
    """oracle clustering on the concatenation + per-file reconstruction"""
    from oracle import pipeline as op
    seg = np.concatenate([art[u]["segmentation"].data for u in uris], axis=0)
    emb = np.concatenate([art[u]["embeddings"] for u in uris], axis=0)
    hard, _, centroids = op.clustering(emb.copy(), seg, min_clusters=1, max_clusters=np.inf,
                                       method="centroid", threshold=0.7045654963945799,
                                       min_cluster_size=12)
    return seg, hard, centroids


def _collect(art):
    def hook(step, artifact, file=None, total=None, completed=None):
        if artifact is not None and total is None:
            art.setdefault(file["uri"], {})[step] = copy.deepcopy(artifact)
    return hook


def test_joint_clustering_matches_oracle(pipeline_dir, gpu_device):
    import pyannote_audio_amd as pa
    from oracle import pipeline as op
    pipeline = pa.Pipeline.from_pretrained(pipeline_dir).to(gpu_device)
    files = _files([(33.0, 5), (27.3, 8), (41.0, 13)])
    art = {}
    outs = dict((f["uri"], o) for f, o in
                pipeline.apply_batch(files, joint_clustering=True, hook=_collect(art)))
    uris = [f["uri"] for f in files]
    seg, hard, centroids = _joint_reference(art, uris)
    assert np.array_equal(pipeline.joint_hard_clusters, hard)
    chunks, frames = op.SW(0.0, 10.0, 1.0), op.SW(0.0, 0.0619375, 0.016875)
    pos = 0
    for u in uris:
        C = art[u]["segmentation"].data.shape[0]
        s = seg[pos:pos + C]
        h = hard[pos:pos + C].copy()
        pos += C
        h[np.sum(s, axis=1) == 0] = -2
        count, _ = op.speaker_count(s, chunks, frames)
        want = op.reconstruct(s, chunks, h, count.astype(np.int8), frames)
        assert np.array_equal(art[u]["discrete_diarization"].data, want), u
        tracks = op.binarize(want, frames)
        got = sorted((a, b, l) for a, b, l in _turns(outs[u].speaker_diarization))
        assert got == sorted((a, b, f"SPEAKER_{l:02d}") for a, b, _, l in tracks), u
        assert np.array_equal(outs[u].speaker_embeddings, centroids)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_pipelined_joint_jobs_equal_sequential_calls(pipeline_dir, gpu_device):
    """`apply_joint_batches`: independent joint-clustering jobs (configs[4], one job per bench step) pipelined --
    front ends of job i+1 beside the clustering / back ends of job i -- give the outputs of the separate calls."""
    import pyannote_audio_amd as pa
    pipeline = pa.Pipeline.from_pretrained(pipeline_dir).to(gpu_device)
    groups = [_files([(21.0, 5), (14.0, 3)]), _files([(18.5, 8)]), _files([(3.0, 21), (16.0, 2)])]
    want = [[(f["uri"], _turns(o.speaker_diarization))
             for f, o in pipeline.apply_batch(copy.copy(g), joint_clustering=True)] for g in groups]
    got = [[(f["uri"], _turns(o.speaker_diarization)) for f, o in job]
           for job in pipeline.apply_joint_batches(groups)]
    assert got == want


def _joint_worker(rank, world, port, pipeline_dir, specs, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import pyannote_audio_amd as pa
    pipeline = pa.Pipeline.from_pretrained(pipeline_dir).to(torch.device("cuda:0"))
    mine = _files(specs[rank])
    res = [(f["uri"], _turns(o.speaker_diarization), o.speaker_embeddings.tolist())
           for f, o in pipeline.apply_batch(mine, joint_clustering=True)]
    # the same job twice through the pipelined form (what bench.py --gpus N runs): collectives stay in the main
    # thread, in the same order on every rank; the clustering / back ends run in the worker thread
    # with several ranks job j is clustered by rank j % world only (pipelining.pipelined_owned) and its labels are
    # broadcast: three jobs, so that rank 0 owns two of them and every rank is a receiver at least once
    again = [[(f["uri"], _turns(o.speaker_diarization), o.speaker_embeddings.tolist()) for f, o in job]
             for job in pipeline.apply_joint_batches([_files(specs[rank]) for _ in range(3)])]
    assert again[0] == again[1] == again[2] == res
    q.put((rank, res))
    dist.barrier()
    dist.destroy_process_group()


def test_joint_clustering_two_ranks(pipeline_dir, gpu_device):
    """rank 0 owns two files, rank 1 one: every rank's outputs equal the single-process joint run"""
    import torch.multiprocessing as mp
    import pyannote_audio_amd as pa
    specs = [[(33.0, 5), (27.3, 8)], [(41.0, 13)]]
    pipeline = pa.Pipeline.from_pretrained(pipeline_dir).to(gpu_device)
    flat = [s for per in specs for s in per]
    want = dict((f["uri"], (_turns(o.speaker_diarization), o.speaker_embeddings.tolist()))
                for f, o in pipeline.apply_batch(_files(flat), joint_clustering=True))
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_joint_worker, args=(r, world, port, pipeline_dir, specs, q))
             for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=600) for _ in range(world)), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    seen = 0
    for rank, items in res:
        for uri, turns, cent in items:
            assert turns == want[uri][0], (rank, uri)
            assert cent == want[uri][1], (rank, uri)
            seen += 1
    assert seen == len(flat)


def test_linkage_global_heap_path_vs_scipy(gpu_device):
    """N = 20 000 > the LDS-heap limit: the <int, global heap> instantiation (joint clustering of
    several hours) equals scipy linkage(pdist(X), "centroid") bit for bit, duplicates included."""
    from scipy.cluster.hierarchy import linkage
    from scipy.spatial.distance import pdist
    from pyannote_audio_amd import distance
    rng = np.random.default_rng(42)
    n, d = 20000, 24
    centers = rng.standard_normal((7, d))
    X = (centers[rng.integers(0, 7, n)] + 0.5 * rng.standard_normal((n, d))).astype(np.float32)
    X[rng.integers(0, n, 40)] = X[rng.integers(0, n, 40)]
    X /= np.linalg.norm(X, axis=1, keepdims=True)
    got = distance.linkage_centroid(X, gpu_device)
    want = linkage(pdist(X), method="centroid")
    bad = np.nonzero((got != want).any(axis=1))[0]
    assert len(bad) == 0, f"first differing merge {bad[0]}: {got[bad[0]]} vs {want[bad[0]]}"
