"""The REFERENCE'S OWN PIPELINE, executed here, against the oracle -- stage by stage and end to end.

tests/refharness.py loads pyannote.audio 4.0.7 from /root/reference/src with stand-ins for the absent
third-party packages (lightning, pyannote.core / .pipeline, asteroid_filterbanks, torchaudio's kaldi.fbank:
ours on both sides of the comparison, i.e. still unpinned).  Everything else that runs below is the
reference's code: `Model.from_pretrained`, `PyanNet` / `SincNet`, `WeSpeakerResNet34`, `Inference.slide /
aggregate`, `SpeakerDiarization.apply` (speaker_count, get_embeddings, reconstruct, to_diarization),
`AgglomerativeClustering` / `VBxClustering` / `PLDA`, `Binarize`, the label mapping.

The checkpoints are the seeded synthetic ones, written by the PRODUCT's `save_checkpoint` and read by the
REFERENCE's loader -- which also pins the checkpoint format.  All comparisons are
bit for bit.  Skipped where /root/reference is absent (the GPU box)."""
import os

import numpy as np
import pytest
import torch

import refharness

pytestmark = pytest.mark.skipif(not refharness.available(), reason="/root/reference is not present")

AHC_PARAMS = {"clustering": {"method": "centroid", "min_cluster_size": 12, "threshold": 0.7045654963945799},
              "segmentation": {"min_duration_off": 0.0}}


@pytest.fixture(scope="module")
def models():
    from oracle.synthetic import calibrated_pyannet, calibrated_wespeaker
    return calibrated_pyannet(calib_seconds=40.0), calibrated_wespeaker(calib_seconds=12.0)


@pytest.fixture(scope="module")
def model_dir(tmp_path_factory, models):
    import oracle.vbx as ov
    from conftest import write_pipeline_dir
    d = str(tmp_path_factory.mktemp("ref_sd31"))
    write_pipeline_dir(d, *models)
    ov.synth_plda(os.path.join(d, "plda"))
    return d


@pytest.fixture(scope="module")
def ref():
    os.environ.setdefault("PYANNOTE_SKIP_DEPENDENCY_CHECK", "1")
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        with refharness.reference_modules(third_party=True) as r:
            r.load_pipelines()
            import sys
            ns = {
                "Model": sys.modules["pyannote.audio"].Model,
                "Inference": sys.modules["pyannote.audio"].Inference,
                "SpeakerDiarization": sys.modules["pyannote.audio.pipelines"].SpeakerDiarization,
                "clustering": sys.modules["pyannote.audio.pipelines.clustering"],
                "signal": r.load("pyannote.audio.utils.signal"),
                "core": sys.modules["pyannote.core"],
                "r": r,
            }
            yield ns


def _pipeline(ref, model_dir, clustering="AgglomerativeClustering", params=AHC_PARAMS, **kw):
    pipe = ref["SpeakerDiarization"](
        segmentation=os.path.join(model_dir, "segmentation"), embedding=os.path.join(model_dir, "embedding"),
        plda=os.path.join(model_dir, "plda"), clustering=clustering, embedding_exclude_overlap=True,
        segmentation_batch_size=32, embedding_batch_size=32, **kw)
    return pipe.instantiate(params)


def _turns(annotation):
    return [(s.start, s.end, l) for s, _, l in annotation.itertracks(yield_label=True)]


def test_reference_loader_reads_product_checkpoints(ref, model_dir, models):
    """core/model.py:497-655 on checkpoints written by pyannote_audio_amd.model.save_checkpoint: the right
    classes, specifications and hyper-parameters come back; PyanNet / WeSpeakerResNet34 forward equal the
    oracle modules bit for bit (PyanNet.py:211-240, sincnet.py:163-184, wespeaker/__init__.py:324-372)."""
    seg_o, emb_o = models
    seg_r = ref["Model"].from_pretrained(os.path.join(model_dir, "segmentation"))
    emb_r = ref["Model"].from_pretrained(os.path.join(model_dir, "embedding"))
    assert type(seg_r).__name__ == "PyanNet" and type(emb_r).__name__ == "WeSpeakerResNet34"
    assert type(seg_r).__module__.startswith("pyannote.audio.models.segmentation")
    spec = seg_r.specifications
    assert spec.powerset and spec.duration == 10.0 and spec.num_powerset_classes == 7
    assert len(spec.classes) == 3 and spec.powerset_max_classes == 2
    rf = seg_r.receptive_field
    assert (round(rf.duration * 16000), round(rf.step * 16000)) == (991, 270) and rf.start == 0.0
    g = torch.Generator().manual_seed(5)
    wav = (0.1 * torch.randn(3, 1, 160000, generator=g)).clamp(-1, 1)
    masks = (torch.rand(3, 589, generator=g) < 0.6).float()
    with torch.inference_mode():
        assert torch.equal(seg_r(wav), seg_o(wav))
        assert torch.equal(emb_r(wav[:, :, :48000], weights=masks), emb_o(wav[:, :, :48000], weights=masks))
        assert torch.equal(emb_r(wav[:2]), emb_o(wav[:2]))
    assert seg_r.num_frames(160000) == 589 and seg_r.num_frames(80000) == 293


@pytest.mark.parametrize("span", [None, 0.4, 1.0, 3.0, 30.0])
def test_reference_compute_fbank_centering_span_equals_oracle(ref, span):
    """wespeaker/__init__.py:113-157 executed from the reference's file: global mean (`None`) and the running mean of
    `fbank_centering_span` seconds (a window shorter than the chunk, about the chunk, far longer than the chunk) --
    the oracle's restatement is bit-identical, and so is the embedding of the whole model.  The stock classes do not
    take `fbank_centering_span` (:346-372): the reference-side model is what a user of the reference would write, a
    subclass that forwards it to `BaseWeSpeakerResNet`."""
    import oracle.models as om
    wespeaker = ref["r"].load("pyannote.audio.models.embedding.wespeaker")
    resnet = ref["r"].load("pyannote.audio.models.embedding.wespeaker.resnet")

    class Centred(wespeaker.BaseWeSpeakerResNet):
        def __init__(self, fbank_centering_span=None):
            super().__init__(fbank_centering_span=fbank_centering_span)
            self.resnet = resnet.ResNet34(80, 256, pooling_func="TSTP", two_emb_layer=False)

    torch.manual_seed(21)
    theirs = Centred(fbank_centering_span=span).eval()
    assert theirs.hparams.fbank_centering_span == span and theirs.hparams.window_type == "hamming"
    ours = om.WeSpeakerResNet34(fbank_centering_span=span).eval()
    ours.load_state_dict(theirs.state_dict())
    g = torch.Generator().manual_seed(9)
    wav = (0.1 * torch.randn(2, 1, 40000, generator=g)).clamp(-1, 1)
    with torch.inference_mode():
        want = theirs.compute_fbank(wav)
        got = ours.compute_fbank(wav)
        assert want.shape == (2, 248, 80) and torch.equal(got, want)
        assert torch.equal(ours(wav), theirs(wav))
        plain = om.WeSpeakerResNet34().compute_fbank(wav)
    # (a window that covers the chunk from every frame is the global mean up to the order of the float32 sums)
    assert (span is None) == torch.equal(plain, want)
    if span == 30.0:
        assert torch.allclose(plain, want, atol=1e-4)


def test_reference_loader_drops_hyper_parameters_its_classes_do_not_take(ref, model_dir, tmp_path):
    """A WeSpeakerResNet34 checkpoint that SAYS fbank_centering_span = 3.0 / snip_edges = False: the reference's class
    does not take either (wespeaker/__init__.py:346-372), its loader drops them (Lightning's `_load_state`, restated in
    tests/refharness.py -- Lightning itself is absent: unpinned) and computes the defaults.  The product does the same
    and says so; a hyper-parameter that counts and that the HIP front end is not built for is refused."""
    import warnings
    import pyannote_audio_amd as pa
    from pyannote_audio_amd.model import load_checkpoint, save_checkpoint
    src = os.path.join(model_dir, "embedding", "pytorch_model.bin")
    ckpt = load_checkpoint(src)
    odd = dict(ckpt["hyper_parameters"], fbank_centering_span=3.0, snip_edges=False)
    path = str(tmp_path / "odd.bin")
    save_checkpoint(path, ckpt["state_dict"], odd, ("pyannote.audio.models.embedding.wespeaker", "WeSpeakerResNet34"),
                    ckpt["pyannote.audio"]["specifications"])
    theirs = ref["Model"].from_pretrained(path)
    assert theirs.hparams.fbank_centering_span is None and theirs.hparams.snip_edges is True
    with warnings.catch_warnings(record=True) as seen:
        warnings.simplefilter("always")
        ours = pa.Model.from_pretrained(path)
    assert ours.fbank_center_kernel == 0
    text = " ".join(str(w.message) for w in seen)
    assert "fbank_centering_span = 3.0" in text and "snip_edges = False" in text and "as in the reference" in text


def test_reference_inference_on_sample_wav(ref, model_dir, models):
    """BASELINE configs[0] on the reference's own `Inference` (core/inference.py:217-373) over its own
    30 s fixture: (21, 589, 3) hard multilabel chunks, equal to oracle.pipeline.slide."""
    from oracle import pipeline as op
    from pyannote_audio_amd.audio import Audio
    seg_o, _ = models
    wav, sr = Audio(16000, mono="downmix")(os.path.join(os.path.dirname(__file__), "golden", "sample.wav"))
    model = ref["Model"].from_pretrained(os.path.join(model_dir, "segmentation"))
    inference = ref["Inference"](model, skip_aggregation=True, batch_size=8)
    swf = inference({"waveform": wav, "sample_rate": sr})
    assert swf.data.shape == (21, 589, 3)
    assert (swf.sliding_window.duration, swf.sliding_window.step) == (10.0, 1.0)
    assert np.array_equal(swf.data, op.slide(seg_o, wav, sr, 10.0, 1.0, 8))
    # a file that ends in an orphan chunk (core/inference.py:270-278)
    short = wav[:, : 16000 * 13 + 4321]
    swf = inference({"waveform": short, "sample_rate": sr})
    want = op.slide(seg_o, short, sr, 10.0, 1.0, 8)
    assert swf.data.shape == want.shape == (5, 589, 3) and np.array_equal(swf.data, want)


@pytest.mark.parametrize("seconds,seed,kwargs", [
    (17.0, 3, {}),
    (19.5, 8, {"num_speakers": 2}),
    (16.0, 5, {"min_speakers": 2, "max_speakers": 3}),
])
def test_reference_speaker_diarization_equals_oracle(ref, model_dir, models, seconds, seed, kwargs):
    """`SpeakerDiarization.apply` (pipelines/speaker_diarization.py:530-784) with the 3.1 configuration:
    every artefact the hook sees and the final annotations equal oracle.pipeline.diarize."""
    from oracle.pipeline import diarize
    from oracle.synthetic import synth_conversation
    seg_o, emb_o = models
    conv, _ = synth_conversation(seconds, seed=seed)
    seen = {}

    def hook(name, artefact, file=None, **kw):
        if artefact is not None:      # (copied: apply() later caps `count.data` in place, :676, :702-707)
            seen[name] = np.array(getattr(artefact, "data", artefact), copy=True)

    out = _pipeline(ref, model_dir)({"waveform": conv, "sample_rate": 16000, "uri": "conv"}, hook=hook,
                                    **kwargs)
    want = diarize(seg_o, emb_o, conv, exclude_overlap=True, **kwargs)
    assert np.array_equal(seen["segmentation"], want.segmentations)
    assert np.array_equal(seen["speaker_counting"].reshape(-1), want.count.reshape(-1))
    assert np.array_equal(seen["embeddings"], want.embeddings, equal_nan=True)
    assert _turns(out.speaker_diarization) == want.diarization
    assert _turns(out.exclusive_speaker_diarization) == want.exclusive_diarization
    assert np.array_equal(out.speaker_embeddings, want.centroids)
    assert len(want.diarization) > 5 and len({l for _, _, l in want.diarization}) >= 2


def test_reference_clustering_classes_equal_oracle(ref, models):
    """`AgglomerativeClustering.__call__` and `VBxClustering.__call__` (pipelines/clustering.py:214-480,
    :550-669) on seeded embeddings: hard clusters, soft scores and centroids equal the oracle's."""
    import oracle.vbx as ov
    from oracle.pipeline import clustering as oracle_ahc
    rng = np.random.default_rng(12)
    C, S, D, F = 140, 3, 256, 589
    centres = rng.standard_normal((4, D))
    who = rng.integers(0, 4, (C, S))
    emb = (centres[who] + 0.35 * rng.standard_normal((C, S, D))).astype(np.float32)
    seg = np.zeros((C, F, S), dtype=np.float32)
    for c in range(C):
        for s in range(S):
            if rng.uniform() < 0.8:
                a = rng.integers(0, 300)
                seg[c, a:a + rng.integers(60, 280), s] = 1.0
    emb[5, 1] = np.nan
    SWF, SW = ref["core"].SlidingWindowFeature, ref["core"].SlidingWindow
    swf = SWF(seg, SW(start=0.0, duration=10.0, step=1.0))
    for kw in ({}, {"num_clusters": 3}, {"min_clusters": 5, "max_clusters": 6}):
        ahc = ref["clustering"].AgglomerativeClustering(metric="cosine").instantiate(AHC_PARAMS["clustering"])
        bounds = {"num_clusters": None, "min_clusters": 1, "max_clusters": np.inf}
        bounds.update(kw)
        if bounds["num_clusters"]:
            bounds["min_clusters"] = bounds["max_clusters"] = bounds["num_clusters"]
        hard, soft, cent = ahc(embeddings=emb.copy(), segmentations=swf, **bounds)
        ohard, osoft, ocent = oracle_ahc(emb.copy(), seg, **bounds, **AHC_PARAMS["clustering"])
        assert np.array_equal(hard, ohard) and np.array_equal(soft, osoft, equal_nan=True)
        assert np.array_equal(cent, ocent)


def test_reference_vbx_pipeline_equals_oracle(ref, model_dir, models):
    """The 4.x / community-1 default (`clustering="VBxClustering"`, speaker_diarization.py:210, 280-285):
    the reference's pipeline with a synthetic PLDA vs the oracle's stages + oracle.vbx.vbx_clustering."""
    import oracle.vbx as ov
    from oracle.synthetic import synth_conversation
    seg_o, emb_o = models
    conv, _ = synth_conversation(17.0, seed=11)
    seen = {}

    def hook(name, artefact, file=None, **kw):
        if artefact is not None:
            seen[name] = np.array(getattr(artefact, "data", artefact), copy=True)

    params = {"segmentation": {"min_duration_off": 0.0}, "clustering": {"threshold": 0.6, "Fa": 0.07, "Fb": 0.8}}
    pipe = _pipeline(ref, model_dir, clustering="VBxClustering", params=params)
    out = pipe({"waveform": conv, "sample_rate": 16000, "uri": "conv"}, hook=hook)
    plda = ov.PLDA(os.path.join(model_dir, "plda", "xvec_transform.npz"),
                   os.path.join(model_dir, "plda", "plda.npz"))
    seg = seen["segmentation"]
    hard, _, cent = ov.vbx_clustering(seen["embeddings"], seg, plda, **params["clustering"])
    # the reference's discrete diarization restated from the oracle's clusters
    from oracle import pipeline as op
    chunks = op.SW(0.0, 10.0, 1.0)
    frames = op.receptive_field(seg_o, 16000)
    count, count_frames = op.speaker_count(seg, chunks, frames)
    assert np.array_equal(seen["speaker_counting"].reshape(-1), count.reshape(-1))
    hard = hard.copy()
    hard[np.sum(seg, axis=1) == 0] = -2
    discrete = op.reconstruct(seg, chunks, hard, count.astype(np.int8), count_frames)
    got = seen["discrete_diarization"]
    assert got.shape == discrete.shape and np.array_equal(got, discrete)
    assert out.speaker_embeddings.shape[1] == 256 and len(_turns(out.speaker_diarization)) > 3


def test_reference_binarize_equals_oracle(ref):
    """`Binarize.__call__` (utils/signal.py:254-318) on a random {0,1} frame matrix vs oracle.pipeline.binarize."""
    from oracle import pipeline as op
    rng = np.random.default_rng(2)
    data = (rng.uniform(size=(4000, 3)) < 0.5).astype(np.float32)
    for k in range(3):                       # long runs instead of salt-and-pepper
        data[:, k] = np.repeat(data[::40, k], 40)
    frames = op.SW(0.0, 991 / 16000, 270 / 16000)
    want = op.binarize(data, frames)
    SWF, SW = ref["core"].SlidingWindowFeature, ref["core"].SlidingWindow
    swf = SWF(data, SW(start=frames.start, duration=frames.duration, step=frames.step))
    ann = ref["signal"].Binarize(onset=0.5, offset=0.5, min_duration_on=0.0, min_duration_off=0.0)(swf)
    got = sorted((s.start, s.end, int(l)) for s, _, l in ann.itertracks(yield_label=True))
    assert got == sorted((s, e, int(l)) for s, e, _, l in want)


def test_reference_binarize_function_equals_oracle(ref):
    """`binarize` (utils/signal.py:78-204; hysteresis thresholding, ndarray and SlidingWindowFeature forms)
    vs oracle.pipeline.hysteresis, incl. NaN scores, scores exactly on a threshold, every initial state."""
    from oracle import pipeline as op
    rng = np.random.default_rng(6)
    scores = rng.uniform(size=(7, 120, 3)).astype(np.float32)
    scores[2, 10:14, 1] = np.nan
    scores[3, ::9, 0] = 0.6            # exactly on the onset
    scores[4, ::7, 2] = 0.4            # exactly on the offset
    SWF, SW = ref["core"].SlidingWindowFeature, ref["core"].SlidingWindow
    swf = SWF(scores, SW(start=0.0, duration=10.0, step=1.0))
    for onset, offset, init in [(0.5, None, False), (0.6, 0.4, None), (0.6, 0.4, True), (0.3, None, None)]:
        want = ref["signal"].binarize(swf, onset=onset, offset=offset, initial_state=init).data
        got = op.hysteresis(scores, onset=onset, offset=offset, initial_state=init)
        assert want.shape == got.shape and np.array_equal(got, want)


def test_reference_non_powerset_pipeline_equals_oracle(ref, tmp_path, models):
    """a14: a multi-label (non-powerset) segmentation checkpoint through the reference's pipeline --
    `binarize(segmentations, onset=threshold, initial_state=False)` (speaker_diarization.py:599-606), counting /
    masks / clustering on the binary form, reconstruction from the RAW scores (:687-691) -- vs the oracle."""
    import oracle.vbx as ov
    from conftest import write_pipeline_dir
    from oracle.pipeline import diarize
    from oracle.synthetic import calibrated_multilabel_pyannet, synth_conversation
    seg_o = calibrated_multilabel_pyannet(calib_seconds=40.0)
    _, emb_o = models
    d = str(tmp_path)
    write_pipeline_dir(d, seg_o, emb_o, powerset=False)
    ov.synth_plda(os.path.join(d, "plda"))
    params = {"clustering": AHC_PARAMS["clustering"], "segmentation": {"threshold": 0.5, "min_duration_off": 0.0}}
    pipe = _pipeline(ref, d, params=params)
    conv, _ = synth_conversation(16.0, seed=14)
    seen = {}

    def hook(name, artefact, file=None, **kw):
        if artefact is not None:
            seen[name] = np.array(getattr(artefact, "data", artefact), copy=True)

    out = pipe({"waveform": conv, "sample_rate": 16000, "uri": "conv"}, hook=hook)
    want = diarize(seg_o, emb_o, conv, exclude_overlap=True, segmentation_threshold=0.5)
    assert seen["segmentation"].dtype == np.float32 and 0.0 < seen["segmentation"].min() < 0.5
    assert np.array_equal(seen["segmentation"], want.raw_segmentations)
    assert np.array_equal(seen["speaker_counting"].reshape(-1), want.count.reshape(-1))
    assert np.array_equal(seen["embeddings"], want.embeddings, equal_nan=True)
    assert _turns(out.speaker_diarization) == want.diarization
    assert _turns(out.exclusive_speaker_diarization) == want.exclusive_diarization
    assert len(want.diarization) > 5 and len({l for _, _, l in want.diarization}) >= 2


def test_reference_xvector_sincnet_equals_oracle(ref):
    """f3: `XVectorSincNet` (models/embedding/xvector.py:205-349) from the reference vs oracle.models.
    XVectorSincNet with the same state dict: same keys, identical embeddings with / without weights, the
    reference's frame geometry and the `min_num_samples` bisection of the pipeline wrapper
    (pipelines/speaker_verification.py:688-702) reproduced by the product's closed form."""
    import oracle.models as om
    import pyannote_audio_amd.model as pm
    xv = ref["r"].load("pyannote.audio.models.embedding.xvector")
    ours = om.seeded_xvector()
    theirs = xv.XVectorSincNet()
    assert list(theirs.state_dict()) == list(ours.state_dict())
    theirs.load_state_dict(ours.state_dict())
    theirs.eval()
    g = torch.Generator().manual_seed(2)
    wav = (0.1 * torch.randn(2, 1, 48000, generator=g)).clamp(-1, 1)
    weights = (torch.rand(2, 173, generator=g) < 0.6).float()
    with torch.inference_mode():
        assert torch.equal(ours(wav, weights=weights), theirs(wav, weights=weights))
        assert torch.equal(ours(wav), theirs(wav))
    assert theirs.dimension == 512
    product = pm.XVectorSincNet.__new__(pm.XVectorSincNet)
    product.hparams = {"sincnet": {"stride": 10}}
    for n in (4771, 4770, 48000, 80000, 160000):
        assert product.num_frames(n) == theirs.num_frames(n)
    assert product.receptive_field_size(1) == theirs.receptive_field_size(1)
    assert product.receptive_field_center(0) == theirs.receptive_field_center(0)
    # the wrapper's bisection: shortest waveform the reference model embeds without raising
    lower, upper = 2, 8000
    with torch.inference_mode():
        while lower + 1 < upper:
            middle = (lower + upper) // 2
            try:
                theirs(torch.randn(1, 1, middle))
                upper = middle
            except Exception:
                lower = middle
    from pyannote_audio_amd.speaker_verification import first_true
    assert upper == first_true(lambda n: product.num_frames(n) > 0, 2, 8000)


@pytest.mark.parametrize("wav2vec,layer", [("WAVLM_BASE", -1), ("tiny", -1), ("tiny", 2)])
def test_reference_sseriouss_equals_oracle(ref, tmp_path, wav2vec, layer):
    """f3: the reference's `SSeRiouSS` (models/segmentation/SSeRiouSS.py:42-328 -- layer weighting :307-313,
    LSTM :315-322, head :324-328, frame geometry :217-287), loaded by the reference's loader from a checkpoint
    the product wrote, vs oracle.models.SSeRiouSS.  BOTH run on oracle/wav2vec2.py, the (unpinned) restatement
    of torchaudio's wav2vec 2.0 / WavLM encoder: this pins the reference's own file, not torchaudio."""
    import oracle.models as om
    from pyannote_audio_amd.model import save_checkpoint, segmentation_specifications
    config = "WAVLM_BASE" if wav2vec == "WAVLM_BASE" else dict(om.TINY_WAV2VEC2)
    ours = om.seeded_sseriouss(wav2vec=config, num_layers=2, wav2vec_layer=layer)
    hparams = {"wav2vec": config, "wav2vec_frozen": False, "wav2vec_layer": layer,
               "lstm": {"hidden_size": 128, "num_layers": 2, "bidirectional": True, "monolithic": True,
                        "dropout": 0.0}, "linear": {"hidden_size": 128, "num_layers": 2},
               "sample_rate": 16000, "num_channels": 1}
    path = os.path.join(str(tmp_path), "pytorch_model.bin")
    save_checkpoint(path, ours.state_dict(), hparams,
                    ("pyannote.audio.models.segmentation.SSeRiouSS", "SSeRiouSS"), segmentation_specifications(10.0))
    theirs = ref["Model"].from_pretrained(path)
    assert type(theirs).__name__ == "SSeRiouSS" and list(theirs.state_dict()) == list(ours.state_dict())
    g = torch.Generator().manual_seed(4)
    wav = (0.1 * torch.randn(2, 1, 24000, generator=g)).clamp(-1, 1)
    with torch.inference_mode():
        want, got = theirs(wav), ours(wav)
    assert want.shape == got.shape and torch.equal(got, want)
    import pyannote_audio_amd.model as pm
    product = pm.SSeRiouSS.__new__(pm.SSeRiouSS)
    product.hparams = hparams
    for n in (24000, 160000, 80000, 400, 399):
        assert product.num_frames(n) == theirs.num_frames(n)
    assert product.receptive_field_size(1) == theirs.receptive_field_size(1)
    assert product.receptive_field_size(2) == theirs.receptive_field_size(2)
    assert product.receptive_field_center(0) == theirs.receptive_field_center(0)
