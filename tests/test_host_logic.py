"""Host-side logic of the product (no GPU): overlap-add, counting, top-k discretisation, hysteresis,
clustering, checkpoint + config loading, error behaviour -- each against the oracle's loop-for-loop
restatement of the reference (oracle/pipeline.py) on seeded inputs.  Integer/index results bit-exact."""
import numpy as np
import pytest
import torch

import pyannote_audio_amd as pa
from oracle import pipeline as O
from pyannote_audio_amd import diarization as D
from pyannote_audio_amd.core import Segment, SlidingWindow, SlidingWindowFeature
from pyannote_audio_amd.inference import Inference


def _random_segmentations(C=40, F=589, S=3, seed=0):
    rng = np.random.default_rng(seed)
    seg = np.zeros((C, F, S), dtype=np.float32)
    for c in range(C):
        for s in range(S):
            pos = 0
            while pos < F:
                gap, on = int(rng.integers(5, 200)), int(rng.integers(5, 250))
                if rng.uniform() < 0.6:
                    seg[c, pos + gap: pos + gap + on, s] = 1.0
                pos += gap + on
    seg[(seg.sum(-1) > 2)] = 0  # at most 2 simultaneous (powerset 3/2)
    return seg


CHUNKS = SlidingWindow(start=0.0, duration=10.0, step=1.0)
FRAMES = SlidingWindow(start=0.0, duration=991 / 16000, step=270 / 16000)
OCH, OFR = O.SW(0.0, 10.0, 1.0), O.SW(0.0, 991 / 16000, 270 / 16000)


def test_aggregate_and_count_match_oracle():
    seg = _random_segmentations()
    count = D.speaker_count(SlidingWindowFeature(seg, CHUNKS), FRAMES, warm_up=(0.0, 0.0))
    ref, ref_frames = O.speaker_count(seg, OCH, OFR)
    assert count.data.dtype == np.uint8 and np.array_equal(count.data, ref)
    assert count.sliding_window.step == ref_frames.step
    # 1 h geometry: 3591 chunks -> 213334 frames, last chunk at frame 212741
    from pyannote_audio_amd.inference import aggregate_start_frames
    st = aggregate_start_frames(SlidingWindow(start=0.0, duration=10.0, step=1.0), FRAMES, 3591)
    assert st[-1] == 212741 and st[-1] + 589 + 4 == 213334
    # generic float data + NaNs + hamming + warm-up
    rng = np.random.default_rng(1)
    x = rng.standard_normal((12, 589, 4)).astype(np.float32)
    x[3, 100:200, 1] = np.nan
    got = Inference.aggregate(SlidingWindowFeature(x.copy(), CHUNKS), FRAMES, warm_up=(0.1, 0.1),
                              hamming=True, missing=0.0)
    want, _ = O.aggregate(x.copy(), OCH, OFR, warm_up=(0.1, 0.1), hamming=True, missing=0.0)
    assert np.array_equal(got.data, want)


def test_reconstruct_and_binarize_match_oracle(monkeypatch):
    seg = _random_segmentations(seed=3)
    C = seg.shape[0]
    rng = np.random.default_rng(4)
    hard = rng.integers(0, 4, size=(C, 3))
    hard[rng.uniform(size=hard.shape) < 0.15] = -2
    count = D.speaker_count(SlidingWindowFeature(seg, CHUNKS), FRAMES, warm_up=(0.0, 0.0))
    count.data = np.minimum(count.data, 3).astype(np.int8)
    # SpeakerDiarization.reconstruct itself runs on the GPU (tests/test_frames_gpu.py); here the host
    # `to_diarization` (general SlidingWindowFeature API) is checked on the same clustered activations
    K = hard.max() + 1
    clustered = np.full((C, seg.shape[1], K), np.nan)
    for c in range(C):
        for k in np.unique(hard[c]):
            if k >= 0:
                clustered[c, :, k] = np.max(seg[c][:, hard[c] == k], axis=1)
    got = D.to_diarization(SlidingWindowFeature(clustered, CHUNKS), count)
    want = O.reconstruct(seg, OCH, hard.copy(), count.data, OFR)
    assert np.array_equal(got.data, want)
    ann = D.to_annotation(got)
    tracks = [(s.start, s.end, t, l) for s, t, l in ann.itertracks(yield_label=True)]
    assert tracks == O.binarize(want, OFR)
    assert ann.labels() == sorted({t[3] for t in tracks}, key=str)


def _cluster_data(C=60, S=3, D_=32, K=4, seed=0):
    rng = np.random.default_rng(seed)
    centers = rng.standard_normal((K, D_))
    who = rng.integers(0, K, size=(C, S))
    emb = (centers[who] + 0.15 * rng.standard_normal((C, S, D_))).astype(np.float32)
    emb[5, 1] = np.nan
    seg = _random_segmentations(C=C, seed=seed + 1)
    return emb, seg


@pytest.mark.parametrize("kw", [dict(), dict(num_clusters=3), dict(min_clusters=6), dict(max_clusters=2)])
def test_clustering_matches_oracle(kw):
    emb, seg = _cluster_data()
    clu = pa.AgglomerativeClustering(metric="cosine").to(torch.device("cpu")).instantiate(
        {"method": "centroid", "min_cluster_size": 12, "threshold": 0.7045654963945799})
    hard, soft, cen = clu(embeddings=emb.copy(), segmentations=SlidingWindowFeature(seg, CHUNKS), **kw)
    rh, rs, rc = O.clustering(emb.copy(), seg, **kw)
    assert np.array_equal(hard, rh)
    assert np.array_equal(cen, rc)
    assert np.array_equal(soft, rs, equal_nan=True)


def test_reference_clustering_kat_on_product_class():
    """/root/reference/tests/test_clustering.py:6-29"""
    embeddings = np.array([[1.0, 1.0, 1.0, 1.0], [1.0, 2.0, 1.0, 2.0]])
    clustering = pa.AgglomerativeClustering().to(torch.device("cpu")).instantiate(
        {"method": "centroid", "min_cluster_size": 0, "threshold": 0.0})
    clusters = clustering.cluster(embeddings=embeddings, min_clusters=2, max_clusters=2, num_clusters=2)
    assert np.array_equal(clusters, np.array([0, 1]))


def test_checkpoint_roundtrip_and_layouts(tmp_path, synthetic_models):
    from oracle.models import PyanNet as OPyanNet
    from pyannote_audio_amd.model import (Model, PyanNet, save_checkpoint, segmentation_specifications)
    from pyannote_audio_amd.weights import SegmentationPack
    from conftest import PYANNET_HPARAMS
    seg, _ = synthetic_models
    p = tmp_path / "pytorch_model.bin"
    save_checkpoint(p, seg.state_dict(), PYANNET_HPARAMS, PyanNet.ARCHITECTURE,
                    segmentation_specifications())
    m = Model.from_pretrained(str(tmp_path))
    assert isinstance(m, PyanNet) and m.specifications.powerset and m.dimension == 7
    assert m.specifications.num_powerset_classes == 7 and m.num_frames(160000) == 589
    rf = m.receptive_field
    assert (rf.start, round(rf.duration * 16000), round(rf.step * 16000)) == (0.0, 991, 270)
    for k, v in seg.state_dict().items():
        assert torch.equal(m.state_dict()[k], v)
    # non-monolithic LSTM key layout (PyanNet.py:101-123) packs to the same device images
    mono = OPyanNet(lstm={"num_layers": 2})
    split = OPyanNet(lstm={"num_layers": 2, "monolithic": False})
    sd = dict(mono.state_dict())
    sd2 = {k: v for k, v in sd.items() if not k.startswith("lstm.")}
    for l in range(2):
        for name in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"):
            for suffix in ("", "_reverse"):
                sd2[f"lstm.{l}.{name}_l0{suffix}"] = sd[f"lstm.{name}_l{l}{suffix}"]
    assert set(sd2) == set(split.state_dict())
    a = SegmentationPack(sd, {"lstm": {"num_layers": 2}}, 7, 3, 2, torch.device("cpu"))
    b = SegmentationPack(sd2, {"lstm": {"num_layers": 2, "monolithic": False}}, 7, 3, 2,
                         torch.device("cpu"))
    assert all(torch.equal(x, y) for x, y in zip(a._keep, b._keep))


def test_pipeline_from_pretrained_contract(pipeline_dir):
    """plugin boundary b1 (cf. /root/reference/tests/test_pipeline_subfolder.py)."""
    import os
    p = pa.Pipeline.from_pretrained(pipeline_dir)
    assert isinstance(p, pa.SpeakerDiarization)
    assert p.instantiated and p.clustering.method == "centroid" and p.clustering.min_cluster_size == 12
    assert p.clustering.threshold == 0.7045654963945799 and p.segmentation.min_duration_off == 0.0
    assert p.embedding_exclude_overlap is True and p.embedding_batch_size == 32
    assert p.segmentation_batch_size == 32
    p.segmentation_batch_size = 256
    assert p._segmentation.batch_size == 256
    assert p._embedding.min_num_samples == 400 and p._embedding.dimension == 256
    assert p._embedding.metric == "cosine" and p._embedding.sample_rate == 16000
    assert p._segmentation.step == 1.0 and p._segmentation.duration == 10.0
    # same thing from the config.yaml path and from a dict
    assert isinstance(pa.Pipeline.from_pretrained(os.path.join(pipeline_dir, "config.yaml")),
                      pa.SpeakerDiarization)
    with pytest.raises(TypeError):
        p.to("cuda")
    with pytest.raises(ValueError):
        pa.Pipeline.from_pretrained(pipeline_dir, revision="main")
    wav = torch.zeros(1, 160000)
    with pytest.raises(ValueError, match="distinct URIs"):
        p([{"waveform": wav, "sample_rate": 16000, "uri": "a"},
           {"waveform": wav, "sample_rate": 16000, "uri": "a"}])
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="no CPU"):
            p({"waveform": wav, "sample_rate": 16000})


def test_error_behaviour(synthetic_models, tmp_path):
    from pyannote_audio_amd.model import PyanNet, segmentation_specifications
    from conftest import PYANNET_HPARAMS
    seg, emb = synthetic_models
    m = PyanNet(seg.state_dict(), PYANNET_HPARAMS, segmentation_specifications())
    with pytest.raises(ValueError, match="Step between consecutive chunks"):
        Inference(m, duration=10.0, step=11.0)       # tests/inference_test.py:59-63
    with pytest.raises(ValueError, match="clustering must be one of"):
        pa.SpeakerDiarization(segmentation=m, embedding=m, clustering="Nope")
    with pytest.raises(ValueError):
        D.set_num_speakers(min_speakers=4, max_speakers=2)
    assert D.set_num_speakers(num_speakers=3) == (3, 3, 3)
    assert D.set_num_speakers() == (None, 1, np.inf)
    with pytest.raises(TypeError):
        Inference(m).to("cuda:0")


def test_bottleneck_checkpoint_loader(tmp_path):
    """WeSpeakerResNet152/221/293 checkpoints (Bottleneck blocks, wespeaker/resnet.py:148-212, 477-507):
    the loader picks the class from the checkpoint's architecture name, reads depth / block type from
    the state-dict keys, and the frame geometry follows the 1-3-1 kernels."""
    from oracle.models import Bottleneck, WeSpeakerResNet34 as OracleNet
    from pyannote_audio_amd.model import (Model, WeSpeakerResNet152, embedding_specifications,
                                          save_checkpoint)
    from pyannote_audio_amd.weights import EmbeddingPack
    from conftest import WESPEAKER_HPARAMS
    net = OracleNet(num_blocks=(1, 2, 1, 1), block=Bottleneck)
    save_checkpoint(tmp_path / "pytorch_model.bin", net.state_dict(), WESPEAKER_HPARAMS,
                    WeSpeakerResNet152.ARCHITECTURE, embedding_specifications())
    m = Model.from_pretrained(str(tmp_path))
    assert isinstance(m, WeSpeakerResNet152) and m._bottleneck and m._num_blocks == [1, 2, 1, 1]
    assert m.dimension == 256
    assert m.num_frames(160000) == 125 and m.num_frames(48000) == 38      # same strides as ResNet34
    # with the reference's formula (utils/receptive_field.py: padded 3x3 convolutions do not widen the
    # field) both block types see one 25 ms fbank window per output frame
    basic = OracleNet(num_blocks=(1, 2, 1, 1))
    from pyannote_audio_amd.model import WeSpeakerResNet34
    mb = WeSpeakerResNet34(basic.state_dict(), WESPEAKER_HPARAMS, embedding_specifications())
    assert m.receptive_field_size() == mb.receptive_field_size() == 400
    assert m.receptive_field_center() == mb.receptive_field_center()
    pack = EmbeddingPack(net.state_dict(), torch.device("cpu"))
    assert pack.bottleneck and pack.num_blocks == (1, 2, 1, 1) and pack.struct.bottleneck == 1
    assert EmbeddingPack(basic.state_dict(), torch.device("cpu")).struct.bottleneck == 0


def test_optimal_mapping_onto_reference_speakers():
    """pipelines/utils/diarization.py:104-148 (pyannote.metrics' Hungarian mapper restated): hypothesis labels
    take the name of the reference speaker they overlap most with, one-to-one; never-overlapping ones keep theirs."""
    from pyannote_audio_amd.diarization import cooccurrence, optimal_mapping
    ref = pa.Annotation(uri="r")
    ref[Segment(0, 10), "_"] = "alice"
    ref[Segment(10, 20), "_"] = "bob"
    hyp = pa.Annotation(uri="h")
    hyp[Segment(0, 9), "_"] = 0          # alice
    hyp[Segment(9, 12), "_"] = 1         # 1 s of alice, 2 s of bob
    hyp[Segment(12, 20), "_"] = 2        # bob
    hyp[Segment(30, 31), "_"] = 3        # outside the reference
    la, lb, m = cooccurrence(hyp, ref)
    assert la == [0, 1, 2, 3] and lb == ["alice", "bob"]
    assert np.allclose(m, [[9, 0], [1, 2], [0, 8], [0, 0]])
    mapped, mapping = optimal_mapping(ref, hyp, return_mapping=True)
    assert mapping == {0: "alice", 2: "bob"}
    assert mapped.labels() == sorted([1, 3, "alice", "bob"], key=str)
    # only the annotated region counts when the file says which part was annotated
    file = {"annotation": ref, "annotated": [Segment(9, 12)]}
    _, mapping = optimal_mapping(file, hyp, return_mapping=True)
    assert mapping == {1: "bob"}         # inside [9, 12] only speaker 1 overlaps anybody (1 s alice, 2 s bob)
    assert optimal_mapping(ref, pa.Annotation(uri="empty")).labels() == []


def test_xvector_pack_folds_batchnorm_forward():
    """XVectorPack (weights.py): every BatchNorm1d of the TDNN stack follows a LeakyReLU, so it is folded into
    the NEXT convolution; the last one stays an affine map in front of the statistics pooling (an all-zero
    mask pools to 0, not to its shift).  The folded chain, evaluated with plain torch ops, equals the oracle
    module (xvector.py:330-349), also for an all-zero mask."""
    import torch.nn.functional as F
    from oracle import seeded_xvector
    from pyannote_audio_amd.weights import XVectorPack
    model = seeded_xvector()
    pack = XVectorPack(model.state_dict(), {"sincnet": {"stride": 10}}, torch.device("cpu"))
    g = torch.Generator().manual_seed(0)
    wav = (0.1 * torch.randn(2, 1, 32000, generator=g)).clamp(-1, 1)
    w = (torch.rand(2, 117, generator=g) < 0.6).float()
    w[1] = 0.0
    with torch.inference_mode():
        want = model(wav, weights=w)
        x = model.sincnet(wav)                                        # (B, 60, T)
        x = F.pad(x, (0, 0, 0, 4))                                    # channels 60 -> 64 (zero)
        for (taps, bias), d in zip(pack.folded_tdnn, XVectorPack.DILATION):
            k, cout, cin = taps.shape
            x = F.leaky_relu(F.conv1d(x, taps.permute(1, 2, 0).contiguous(), bias, dilation=d))
        sc, sh = pack.last_batchnorm
        stats = model.stats_pool(x * sc.view(1, -1, 1) + sh.view(1, -1, 1), weights=w)
        ew, eb = pack.embedding
        got = stats @ ew[:, :stats.shape[1]].T + eb
    assert ew.shape[1] == 3008 and torch.count_nonzero(ew[:, 3000:]) == 0
    assert torch.allclose(got, want, rtol=1e-4, atol=1e-5)


def _host_array(pointer, *shape):
    """a pa_*_weights pointer field of a pack built on the CPU device -> torch tensor view (what the kernels
    would read)"""
    import ctypes
    n = int(np.prod(shape))
    addr = pointer if isinstance(pointer, int) else pointer.value
    return torch.from_numpy(np.ctypeslib.as_array((ctypes.c_float * n).from_address(addr)).reshape(shape).copy())


def _emulate_sser_encoder(pack, wav):
    """pa_sser_forward's encoder, operation for operation, in plain torch on the PACKED weights
    (csrc/sser_forward.cpp + w2v.hip): validates every host-side layout / fold of SSeRiouSSPack."""
    import torch.nn.functional as F
    w, cfg = pack.struct, pack.cfg
    B = wav.shape[0]
    D, H, FF, nl = w.embed_dim, w.num_heads, w.ff_dim, w.num_layers
    hd = D // H
    x = wav
    cin = 1
    for l in range(w.num_conv):
        cout, k, s = w.conv_channels[l], w.conv_kernel[l], w.conv_stride[l]
        W = _host_array(w.conv_w[l], cout, k * cin)
        win = x.unfold(1, k, s) if l == 0 else x.unfold(1, k, s).permute(0, 1, 3, 2).reshape(B, -1, k * cin)
        x = win @ W.T
        if w.conv_b[l]:
            x = x + _host_array(w.conv_b[l], cout)
        if w.extractor_layer_norm:
            x = F.gelu(F.layer_norm(x, (cout,), _host_array(w.conv_norm_g[l], cout), _host_array(w.conv_norm_b[l], cout)))
        elif l == 0:
            m, v = x.mean(1, keepdim=True), x.var(1, unbiased=False, keepdim=True)
            x = F.gelu((x - m) / torch.sqrt(v + 1e-5) * _host_array(w.conv_norm_g[0], cout) + _host_array(w.conv_norm_b[0], cout))
        else:
            x = F.gelu(x)
        cin = cout
    T = x.shape[1]
    x = F.layer_norm(x, (cin,), _host_array(w.proj_ln_g, cin), _host_array(w.proj_ln_b, cin))
    x = x @ _host_array(w.proj_w, D, cin).T + _host_array(w.proj_b, D)
    G, KW = w.pos_groups, w.pos_kernel
    CG = D // G
    w3 = _host_array(w.pos_w, G, KW, CG, CG)
    xp = F.pad(x, (0, 0, KW // 2, KW // 2))                                  # zero padding in time
    pos = torch.zeros_like(x)
    for g in range(G):
        for j in range(KW):
            pos[:, :, g * CG:(g + 1) * CG] += xp[:, j:j + T, g * CG:(g + 1) * CG] @ w3[g, j]
    x = x + F.gelu(pos + _host_array(w.pos_b, D))
    if not w.layer_norm_first:      # (post-LN: the encoder-level LayerNorm precedes the layers, sser_forward.cpp)
        x = F.layer_norm(x, (D,), _host_array(w.enc_ln_g, D), _host_array(w.enc_ln_b, D))
    bias = pack.relative_bias(T)
    outs = []
    for l in range(nl):
        L = w.layers[l]
        ln1 = lambda t: F.layer_norm(t, (D,), _host_array(L.ln1_g, D), _host_array(L.ln1_b, D))   # noqa: E731
        ln2 = lambda t: F.layer_norm(t, (D,), _host_array(L.ln2_g, D), _host_array(L.ln2_b, D))   # noqa: E731
        a = ln1(x) if w.layer_norm_first else x
        qk = a @ _host_array(L.qk_w, 2 * D, D).T + _host_array(L.qk_b, 2 * D)
        v = a @ _host_array(L.v_w, D, D).T
        q, k = (t.view(B, T, H, hd).transpose(1, 2) for t in (qk[..., :D], qk[..., D:]))
        s = q @ k.transpose(-1, -2) / np.sqrt(hd)
        if bias is not None:
            u = a.view(B, T, H, hd).transpose(1, 2) @ _host_array(L.gate_w, 8, hd).T + _host_array(L.gate_b, 8)
            ga, gb = torch.sigmoid(u[..., :4].sum(-1)), torch.sigmoid(u[..., 4:].sum(-1))
            gate = ga * (gb * _host_array(L.gate_const, H).view(1, H, 1) - 1.0) + 2.0
            s = s + gate.unsqueeze(-1) * bias
        o = (torch.softmax(s, -1) @ v.view(B, T, H, hd).transpose(1, 2)).transpose(1, 2).reshape(B, T, D)
        x1 = x + o @ _host_array(L.out_w, D, D).T + _host_array(L.out_b, D)
        def ff(t):
            hidden = F.gelu(t @ _host_array(L.ff1_w, FF, D).T + _host_array(L.ff1_b, FF))
            return hidden @ _host_array(L.ff2_w, D, FF).T + _host_array(L.ff2_b, D)
        if w.layer_norm_first:
            x = x1 + ff(ln2(x1))
        else:
            a1 = ln1(x1)
            x = ln2(a1 + ff(a1))
        outs.append(x)
    if w.use_layer < 0:
        return sum(w.layer_mix[l] * outs[l] for l in range(nl))
    return outs[max(1, w.use_layer) - 1]


@pytest.mark.parametrize("config,layer", [("tiny", -1), ("tiny", 2), ("wavlm_small", -1)])
def test_sseriouss_pack_layouts(config, layer):
    """SSeRiouSSPack (weights.py): strided convolutions as GEMM operands, q|k / v split with the v bias folded
    into the output projection, materialised weight_norm of the positional convolution in [g][j][ci][co] order,
    the relative-position table, softmax'd layer weights -- the packed chain equals the oracle's encoder."""
    import oracle.models as om
    import oracle.wav2vec2 as ow
    from pyannote_audio_amd.weights import SSeRiouSSPack
    if config == "tiny":
        wav2vec = dict(om.TINY_WAV2VEC2)
        model = om.seeded_sseriouss(wav2vec=wav2vec, num_layers=1, wav2vec_layer=layer)
    else:   # a small WavLM: group_norm extractor, post-LN encoder, gated relative position bias
        small = dict(ow.WAVLM_BASE, extractor_conv_layer_config=[(64, 10, 5), (64, 3, 2), (64, 2, 2)],
                     encoder_embed_dim=128, encoder_num_layers=2, encoder_num_heads=4, encoder_pos_conv_kernel=16,
                     encoder_pos_conv_groups=4, encoder_ff_interm_features=256, encoder_num_buckets=40,
                     encoder_max_distance=100)
        ow.PIPELINES["_TEST_SMALL"] = ow._Bundle(small)
        import pyannote_audio_amd.weights as pw
        pw.WAV2VEC_BUNDLES["_TEST_SMALL"] = dict(
            {k: v for k, v in small.items() if k in pw.WAV2VEC_BUNDLES["WAVLM_BASE"]}, wavlm=True)
        wav2vec = "_TEST_SMALL"
        model = om.seeded_sseriouss(wav2vec=wav2vec, num_layers=1, wav2vec_layer=layer)
    hparams = {"wav2vec": wav2vec, "wav2vec_layer": layer, "lstm": {"num_layers": 1}}
    pack = SSeRiouSSPack(model.state_dict(), hparams, 7, 3, 2, torch.device("cpu"))
    assert (pack.struct.lstm_hidden, pack.struct.lstm_bidir, pack.struct.linear_hidden) == (128, 1, 128)
    g = torch.Generator().manual_seed(0)
    wav = (0.1 * torch.randn(2, 4000, generator=g)).clamp(-1, 1)
    with torch.inference_mode():
        outs, _ = model.wav2vec.extract_features(wav, num_layers=None if layer < 0 else layer)
        want = torch.stack(outs, dim=-1) @ torch.softmax(model.wav2vec_weights, 0) if layer < 0 else outs[-1]
        got = _emulate_sser_encoder(pack, wav)
    assert got.shape == want.shape
    assert torch.allclose(got, want, rtol=1e-4, atol=2e-5), (got - want).abs().max()


def test_sseriouss_wav2vec_forms_of_the_reference(tmp_path):
    """`wav2vec` hyper-parameter forms of the reference (models/segmentation/SSeRiouSS.py:97-123): a bundle name, the
    PATH of a self-supervised checkpoint {"config", "state_dict"}, a wav2vec2_model kwargs dict; and
    `wav2vec_layer=0`, for which torchaudio's extract_features(num_layers=0) raises ValueError."""
    import oracle.models as om
    from pyannote_audio_amd.weights import SSeRiouSSPack, wav2vec_config
    cfg = dict(om.TINY_WAV2VEC2)
    path = tmp_path / "ssl.ckpt"
    torch.save({"config": cfg, "state_dict": {}}, path)
    from_path, from_dict = wav2vec_config(str(path)), wav2vec_config(cfg)
    assert from_path == from_dict and from_path["wavlm"] is False
    assert wav2vec_config("WAVLM_BASE")["wavlm"] is True
    with pytest.raises(NotImplementedError, match="neither one of the built torchaudio bundles"):
        wav2vec_config("NOT_A_BUNDLE_NOR_A_FILE")
    torch.save({"state_dict": {}}, path)
    with pytest.raises(ValueError, match="no 'config' entry"):
        wav2vec_config(str(path))
    model = om.seeded_sseriouss(wav2vec=cfg, num_layers=1, wav2vec_layer=1)
    for layer in (0, cfg["encoder_num_layers"] + 1):
        with pytest.raises(ValueError, match="`num_layers` must be between"):
            SSeRiouSSPack(model.state_dict(), {"wav2vec": cfg, "wav2vec_layer": layer, "lstm": {"num_layers": 1}},
                          7, 3, 2, torch.device("cpu"))
