"""The oracle against the REFERENCE'S OWN CODE, executed in this process.

Five files of pyannote.audio 4.0.7 need nothing but torch / numpy / scipy / einops and are loaded from
/root/reference/src where they lie (tests/refharness.py; nothing is copied):

    models/embedding/wespeaker/resnet.py   ResNet34 / 152 / 221 / 293, BasicBlock, Bottleneck, TSTP
    models/blocks/pooling.py               StatsPool
    utils/powerset.py                      Powerset
    utils/receptive_field.py               frame geometry
    utils/vbx.py                           VBx, cluster_vbx, vbx_setup

Every comparison is BIT FOR BIT on seeded inputs (same process, same thread count, same BLAS).  These tests
skip where /root/reference does not exist (the GPU box); what they establish travels there as
tests/golden/reference_v1.npz (tests/golden/make_reference_golden.py, tests/test_golden.py)."""
import os

import numpy as np
import pytest
import torch

import refharness

pytestmark = pytest.mark.skipif(not refharness.available(), reason="/root/reference is not present")


@pytest.fixture(scope="module")
def ref():
    with refharness.reference_modules() as r:
        mods = {
            "resnet": r.load("pyannote.audio.models.embedding.wespeaker.resnet"),
            "pooling": r.load("pyannote.audio.models.blocks.pooling"),
            "powerset": r.load("pyannote.audio.utils.powerset"),
            "receptive_field": r.load("pyannote.audio.utils.receptive_field"),
            "vbx": r.load("pyannote.audio.utils.vbx"),
        }
    return mods


def _randomise_bn(model, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for m in model.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.copy_(0.1 * torch.randn(m.running_mean.shape, generator=g))
                m.running_var.copy_(0.5 + torch.rand(m.running_var.shape, generator=g))
                m.weight.copy_(1.0 + 0.1 * torch.randn(m.weight.shape, generator=g))
                m.bias.copy_(0.1 * torch.randn(m.bias.shape, generator=g))


@pytest.mark.parametrize("name,blocks,bottleneck,frames", [
    ("ResNet34", (3, 4, 6, 3), False, 298),
    ("ResNet152", (3, 8, 36, 3), True, 67),
    ("ResNet221", (6, 16, 48, 3), True, 40),
    ("ResNet293", (10, 20, 64, 3), True, 33),
])
def test_wespeaker_resnets(ref, name, blocks, bottleneck, frames):
    """wespeaker/resnet.py:84-145 (BasicBlock), :148-212 (Bottleneck), :215-260, :399-430 (ResNet),
    :37-66 (TSTP) with the wrapper's arguments (`two_emb_layer=False`, wespeaker/__init__.py:370-372):
    same state-dict keys, identical embeddings with and without pooling weights."""
    import oracle.models as om
    torch.manual_seed(11)
    theirs = getattr(ref["resnet"], name)(80, 256, pooling_func="TSTP", two_emb_layer=False)
    _randomise_bn(theirs, 5)
    theirs.eval()
    ours = om.ResNet(blocks, 32, 80, 256, block=om.Bottleneck if bottleneck else om.BasicBlock)
    assert list(ours.state_dict()) == list(theirs.state_dict())
    ours.load_state_dict(theirs.state_dict())
    ours.eval()
    g = torch.Generator().manual_seed(3)
    fbank = torch.randn(2, frames, 80, generator=g)
    weights = (torch.rand(2, 589, generator=g) < 0.7).float()
    with torch.inference_mode():
        _, want = theirs(fbank.clone(), weights=weights)
        got = ours(fbank.clone(), weights=weights)
        _, want_u = theirs(fbank.clone())
        got_u = ours(fbank.clone())
    assert want.shape == (2, 256) and torch.equal(got, want)
    assert torch.equal(got_u, want_u)


def test_stats_pool(ref):
    """models/blocks/pooling.py:30-130: unweighted, (B,T) weights, (B,S,T) weights, weights on another
    time grid (nearest interpolation), all-zero weights."""
    import oracle.models as om
    theirs, ours = ref["pooling"].StatsPool(), om.StatsPool()
    g = torch.Generator().manual_seed(0)
    x = torch.randn(3, 40, 125, generator=g)
    cases = [None,
             torch.rand(3, 125, generator=g),
             (torch.rand(3, 589, generator=g) < 0.6).float(),
             torch.rand(3, 3, 589, generator=g),
             torch.zeros(3, 589)]
    for w in cases:
        want = theirs(x, weights=w)
        got = ours(x, weights=w)
        assert got.shape == want.shape and torch.equal(got, want)
    tstp_t = ref["resnet"].TSTP(in_dim=2560)
    f = torch.randn(2, 256, 10, 38, generator=g)
    w = (torch.rand(2, 589, generator=g) < 0.5).float()
    assert torch.equal(om.TSTP()(f, weights=w), tstp_t(f, weights=w))
    assert tstp_t.get_out_dim() == 5120


@pytest.mark.parametrize("num_classes,max_set_size", [(3, 2), (4, 2), (3, 3), (2, 1)])
def test_powerset(ref, num_classes, max_set_size):
    """utils/powerset.py:37-140: mapping, hard and soft `to_multilabel`, `to_powerset`."""
    import oracle.models as om
    theirs = ref["powerset"].Powerset(num_classes, max_set_size)
    ours = om.Powerset(num_classes, max_set_size)
    assert ours.num_powerset_classes == theirs.num_powerset_classes
    assert torch.equal(ours.mapping, theirs.mapping)
    g = torch.Generator().manual_seed(1)
    logp = torch.log_softmax(torch.randn(4, 57, theirs.num_powerset_classes, generator=g), dim=-1)
    assert torch.equal(ours.to_multilabel(logp), theirs.to_multilabel(logp))
    assert torch.equal(ours.to_multilabel(logp, soft=True), theirs.to_multilabel(logp, soft=True))
    assert torch.equal(ours(logp), theirs(logp))
    ml = theirs.to_multilabel(logp)
    assert torch.equal(ours.to_powerset(ml), theirs.to_powerset(ml))


def test_powerset_product_lut(ref):
    """The product's hard conversion is an arg-max -> look-up table (k_classifier): the table must be
    the reference's mapping (utils/powerset.py:80-109) for the 3.1 model's (3, 2)."""
    from pyannote_audio_amd.weights import powerset_mapping
    want = ref["powerset"].Powerset(3, 2).mapping.numpy()
    assert np.array_equal(np.asarray(powerset_mapping(3, 2)), want)


def test_receptive_field(ref):
    """utils/receptive_field.py:26-165 with SincNet's kernels / strides (models/blocks/sincnet.py:82-161):
    oracle.pipeline.receptive_field and the product's frame geometry."""
    import oracle.models as om
    from oracle.pipeline import receptive_field
    rf = ref["receptive_field"]
    ks, ss, ps, ds = [251, 3, 5, 3, 5, 3], [10, 3, 1, 3, 1, 3], [0] * 6, [1] * 6
    size = rf.multi_conv_receptive_field_size(1, ks, ss, padding=ps, dilation=ds)
    step = rf.multi_conv_receptive_field_size(2, ks, ss, padding=ps, dilation=ds) - size
    center = rf.multi_conv_receptive_field_center(0, ks, ss, padding=ps, dilation=ds)
    assert (size, step, center) == (991, 270, 495)
    sw = receptive_field(om.seeded_pyannet(num_layers=1), 16000)
    assert sw.duration == size / 16000 and sw.step == step / 16000
    assert sw.start == (center - (size - 1) / 2) / 16000
    import pyannote_audio_amd.model as pm
    from pyannote_audio_amd.segmentation import num_frames
    for n in (160000, 80000, 48000, 16000, 991, 1261, 123457):
        want = rf.multi_conv_num_frames(n, ks, ss, padding=ps, dilation=ds)
        assert num_frames(n) == want == pm.multi_conv_num_frames(n, ks, ss, ps, ds)
    assert pm.multi_conv_receptive_field_size(1, ks, ss, ps, ds) == size
    assert pm.multi_conv_receptive_field_size(2, ks, ss, ps, ds) - size == step
    assert pm.multi_conv_receptive_field_center(0, ks, ss, ps, ds) == center
    # padded / strided 2-D chains of the ResNets (wespeaker/__init__.py:160-322)
    ks2, ss2, ps2 = [3, 3, 3, 3, 1, 3], [1, 1, 1, 2, 1, 2], [1, 1, 1, 1, 0, 1]
    for f in (0, 1, 7):
        assert (pm.multi_conv_receptive_field_center(f, ks2, ss2, ps2, [1] * 6)
                == rf.multi_conv_receptive_field_center(f, ks2, ss2, padding=ps2, dilation=[1] * 6))
        assert (pm.multi_conv_receptive_field_size(f + 1, ks2, ss2, ps2, [1] * 6)
                == rf.multi_conv_receptive_field_size(f + 1, ks2, ss2, padding=ps2, dilation=[1] * 6))
    # the WeSpeaker wrappers' geometry (wespeaker/__init__.py:160-230): fbank window 400 / hop 160
    assert rf.conv1d_num_frames(48000, kernel_size=400, stride=160, padding=0, dilation=1) == 298
    assert rf.conv1d_num_frames(160000, kernel_size=400, stride=160, padding=0, dilation=1) == 998


def test_vbx_setup_and_cluster_vbx(ref, tmp_path):
    """utils/vbx.py:27-157 (VBx, cluster_vbx) and :160-218 (vbx_setup): the oracle's PLDA front end,
    responsibilities, priors and ELBO trace equal the reference's bit for bit."""
    import oracle.vbx as ov
    d = ov.synth_plda(str(tmp_path), seed=9)
    xvec_tf, plda_tf, plda_psi = ref["vbx"].vbx_setup(os.path.join(d, "xvec_transform.npz"),
                                                      os.path.join(d, "plda.npz"))
    plda = ov.PLDA(os.path.join(d, "xvec_transform.npz"), os.path.join(d, "plda.npz"))
    rng = np.random.default_rng(4)
    centres = rng.standard_normal((4, 256))
    labels = rng.integers(0, 4, 300)
    emb = (centres[labels] + 0.3 * rng.standard_normal((300, 256))).astype(np.float32)
    want_fea = plda_tf(xvec_tf(emb), lda_dim=128)
    got_fea = plda(emb)
    assert np.array_equal(got_fea, want_fea)
    assert np.array_equal(plda.phi, plda_psi[:128])
    init = (labels + rng.integers(0, 2, 300)) % 5        # a deliberately imperfect 5-cluster start
    for Fa, Fb in [(0.07, 0.8), (0.3, 2.0)]:
        wq, wpi = ref["vbx"].cluster_vbx(init, want_fea, plda_psi[:128], Fa=Fa, Fb=Fb, maxIters=20)
        gq, gpi, gL = ov.cluster_vbx(init, got_fea, plda.phi, Fa=Fa, Fb=Fb, maxIters=20)
        assert np.array_equal(gq, wq) and np.array_equal(gpi, wpi)
        # the ELBO trace (and with it the stopping iteration) from VBx itself, utils/vbx.py:27-140
        from scipy.special import softmax
        q0 = np.zeros((300, 5))
        q0[range(300), init] = 1.0
        q0 = softmax(q0 * 7.0, axis=1)
        _, _, wL = ref["vbx"].VBx(want_fea, plda_psi[:128], Fa=Fa, Fb=Fb, pi=5, gamma=q0, maxIters=20)
        assert np.array_equal(np.asarray(gL), np.asarray(wL)) and 2 <= len(gL) <= 20
