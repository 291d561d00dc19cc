"""utils/permutation.py: the reference's own known-answer tests (tests/utils/test_permutation.py:7-113) on
this build's `permutate`, plus the cost matrix and the callable-cost path."""
import numpy as np
import torch

from pyannote_audio_amd.permutation import permutate

ALL_3 = [(0, 1, 2), (0, 2, 1), (1, 0, 2), (1, 2, 0), (2, 0, 1), (2, 1, 0)]


def test_permutate_torch():
    torch.manual_seed(0)
    y2 = torch.randn((10, 3))
    y1 = torch.stack([y2[:, p] for p in ALL_3])
    permutated, permutations = permutate(y1, y2)
    assert permutations == ALL_3 and isinstance(permutated, torch.Tensor)
    for k, p in enumerate(ALL_3):
        np.testing.assert_allclose(permutated[k], y2[:, p])


def test_permutate_numpy():
    rng = np.random.default_rng(0)
    y2 = rng.standard_normal((10, 3))
    y1 = np.stack([y2[:, p] for p in ALL_3])
    permutated, permutations = permutate(y1, y2)
    assert permutations == ALL_3
    for k, p in enumerate(ALL_3):
        np.testing.assert_allclose(permutated[k], y2[:, p])


def test_permutate_less_speakers():
    rng = np.random.default_rng(1)
    wanted = [(0, 1, None), (0, None, 1), (1, 0, None), (1, None, 0), (None, 0, 1), (None, 1, 0)]
    y2 = rng.standard_normal((10, 2))
    y1 = np.zeros((len(wanted), 10, 3))
    for k, p in enumerate(wanted):
        for i, j in enumerate(p):
            if j is not None:
                y1[k, :, i] = y2[:, j]
    _, permutations = permutate(y1, y2)
    assert permutations == wanted


def test_permutate_more_speakers():
    rng = np.random.default_rng(2)
    wanted = [(0, 1), (0, 2), (1, 0), (1, 2), (2, 0), (2, 1)]
    y2 = rng.standard_normal((10, 3))
    y1 = np.stack([np.stack([y2[:, j] for j in p], axis=1) for p in wanted])
    permutated, permutations = permutate(y1, y2)
    assert permutations == wanted
    np.testing.assert_allclose(permutated, y1)


def test_costs_and_callable_cost():
    rng = np.random.default_rng(3)
    y2 = rng.standard_normal((12, 3))
    y1 = y2[None, :, [2, 0, 1]]
    _, permutations, cost = permutate(y1, y2, return_cost=True)
    assert permutations == [(2, 0, 1)] and cost.shape == (1, 3, 3)
    want = np.array([[np.mean((y1[0, :, i] - y2[:, j]) ** 2) for j in range(3)] for i in range(3)])
    np.testing.assert_allclose(cost[0], want)
    mae = lambda Y, y, **kw: torch.mean(torch.abs(Y - y), axis=0)      # utils/permutation.py:84-96
    _, with_callable, cost_mae = permutate(y1, y2, cost_func=mae, return_cost=True)
    _, with_name, cost_named = permutate(y1, y2, cost_func="mae", return_cost=True)
    assert with_callable == with_name == [(2, 0, 1)]
    np.testing.assert_allclose(cost_mae, cost_named)
