"""Pin the oracle with the reference's own known-answer tests:
  /root/reference/tests/test_stats_pool.py:28-131, tests/test_clustering.py:6-29,
  tests/utils/test_powerset.py:29-76, plus the frame-count identities of SURVEY.md appendix A."""
import numpy as np
import torch

from oracle.models import Powerset, PyanNet, StatsPool, WeSpeakerResNet34, kaldi_fbank
from oracle.pipeline import ahc_cluster, receptive_field


def test_stats_pool_weightless():
    x = torch.Tensor([[[2.0, 4.0], [2.0, 4.0]], [[1.0, 1.0], [1.0, 1.0]]])
    y = StatsPool()(x)
    assert torch.equal(torch.round(y, decimals=4),
                       torch.Tensor([[3.0, 3.0, 1.4142, 1.4142], [1.0, 1.0, 0.0, 0.0]]))


def test_stats_pool_one_speaker():
    x = torch.Tensor([[[2.0, 4.0], [2.0, 4.0]], [[1.0, 1.0], [1.0, 1.0]]])
    w = torch.Tensor([[0.5, 0.01], [0.2, 0.1]])
    y = StatsPool()(x, weights=w)
    assert torch.equal(torch.round(y, decimals=4),
                       torch.Tensor([[2.0392, 2.0392, 1.4142, 1.4142], [1.0, 1.0, 0.0, 0.0]]))


def test_stats_pool_multi_speaker():
    x = torch.Tensor([[[2.0, 4.0], [2.0, 4.0]], [[1.0, 1.0], [1.0, 1.0]]])
    w = torch.Tensor([[[0.1, 0.2], [0.2, 0.3]], [[0.001, 0.001], [0.2, 0.3]]])
    y = StatsPool()(x, weights=w)
    assert torch.equal(torch.round(y, decimals=4), torch.Tensor(
        [[[3.3333, 3.3333, 1.4142, 1.4142], [3.2, 3.2, 1.4142, 1.4142]],
         [[1.0, 1.0, 0.0, 0.0], [1.0, 1.0, 0.0, 0.0]]]))


def test_stats_pool_frame_mismatch():
    x = torch.Tensor([[[2.0, 2.0], [2.0, 2.0]], [[1.0, 1.0], [1.0, 1.0]]])
    w = torch.Tensor([[0.5, 0.5, 0.0], [0.0, 0.5, 0.5]])
    y = StatsPool()(x, weights=w)
    assert torch.equal(torch.round(y, decimals=4),
                       torch.Tensor([[2.0, 2.0, 0.0, 0.0], [1.0, 1.0, 0.0, 0.0]]))


def test_stats_pool_all_zero_weights():
    x = torch.Tensor([[[2.0, 4.0], [2.0, 4.0]], [[1.0, 1.0], [1.0, 1.0]]])
    w = torch.Tensor([[0.5, 0.01], [0.0, 0.0]])
    y = StatsPool()(x, weights=w)
    assert torch.equal(torch.round(y, decimals=4),
                       torch.Tensor([[2.0392, 2.0392, 1.4142, 1.4142], [0.0, 0.0, 0.0, 0.0]]))


def test_agglomerative_clustering_num_cluster():
    """issue 1525 regression (reference tests/test_clustering.py)"""
    embeddings = np.array([[1.0, 1.0, 1.0, 1.0], [1.0, 2.0, 1.0, 2.0]])
    clusters = ahc_cluster(embeddings, min_clusters=2, max_clusters=2, num_clusters=2,
                           method="centroid", min_cluster_size=0, threshold=0.0)
    assert np.array_equal(clusters, np.array([0, 1]))


def test_powerset_roundtrip():
    for num_classes in range(2, 5):
        for max_set_size in range(1, num_classes + 1):
            powerset = Powerset(num_classes, max_set_size)
            one = [[0] * powerset.num_powerset_classes for _ in range(powerset.num_powerset_classes)]
            for i in range(powerset.num_powerset_classes):
                one[i][i] = 1.0
            batch = torch.tensor([one, one[::-1]])
            assert torch.equal(batch, powerset.to_powerset(powerset.to_multilabel(batch)).to(batch.dtype))


def test_powerset_mapping_order():
    m = Powerset(3, 2).mapping
    assert m.tolist() == [[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1], [1, 1, 0], [1, 0, 1], [0, 1, 1]]


def test_frame_geometry():
    m = PyanNet()
    assert m.num_frames(160000) == 589 and m.num_frames(80000) == 293
    rf = receptive_field(m)
    assert abs(rf.duration - 991 / 16000) < 1e-12 and abs(rf.step - 270 / 16000) < 1e-12
    assert rf.start == 0.0
    x = torch.zeros(1, 1, 160000)
    with torch.inference_mode():
        assert m.eval()(x).shape == (1, 589, 7)
    for n, t in ((160000, 998), (80000, 498), (48000, 298)):
        assert kaldi_fbank(torch.zeros(1, n)).shape == (t, 80)
    e = WeSpeakerResNet34().eval()
    with torch.inference_mode():
        assert e.resnet.forward_frames(torch.zeros(1, 298, 80)).shape == (1, 256, 10, 38)
