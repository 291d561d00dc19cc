"""Cost-minimising speaker permutation between two activation sequences (utils/permutation.py:38-196):
`permutate(y1, y2)` maps the speakers of `y2` onto those of `y1` by Hungarian assignment on the mean
squared (or absolute) frame difference.  Used by `OracleClustering`; pinned by the reference's own
tests/utils/test_permutation.py (tests/test_permutation_cpu.py)."""
from __future__ import annotations

from typing import Callable, List, Optional, Tuple, Union

import numpy as np
import torch
from scipy.optimize import linear_sum_assignment

Array = Union[np.ndarray, torch.Tensor]


def _pair_costs(a: np.ndarray, b: np.ndarray, cost_func) -> np.ndarray:
    """(frames, C1), (frames, C2) -> (C1, C2) cost of pairing class i of `a` with class j of `b`"""
    if cost_func in (None, "mse", "mae"):
        diff = a[:, :, None] - b[:, None, :]
        return np.mean(diff * diff if cost_func != "mae" else np.abs(diff), axis=0)
    # callable: cost_func(Y, y) -> (num_classes,) with y one class of `a` repeated (permutation.py:143-148)
    ta, tb = torch.from_numpy(np.array(a)), torch.from_numpy(np.array(b))     # (writable copies)
    rows = [cost_func(tb, ta[:, i:i + 1].expand(-1, tb.shape[1])) for i in range(ta.shape[1])]
    return torch.stack(rows).numpy()


def permutate(y1: Array, y2: Array, cost_func: Union[Callable, str, None] = "mse", return_cost: bool = False):
    """y1: (batch, frames, C1); y2: (frames, C2) or (batch, frames, C2).
    -> (y2 re-ordered like y1 (batch, frames, C1), permutations[, costs (batch, C1, C2)]) where
    `permutations[b][i] = j` says class j of y2 plays the part of class i of y1 (None: nobody does)."""
    as_torch = isinstance(y1, torch.Tensor)
    a = y1.detach().cpu().numpy() if as_torch else np.asarray(y1)
    b = y2.detach().cpu().numpy() if isinstance(y2, torch.Tensor) else np.asarray(y2)
    if b.ndim == 2:
        b = np.broadcast_to(b, (a.shape[0],) + b.shape)
    if b.ndim != 3:
        raise ValueError("Incorrect shape: should be (batch_size, num_frames, num_classes).")
    if a.shape[:2] != b.shape[:2]:
        raise ValueError(f"Shape mismatch: {tuple(a.shape)} vs. {tuple(b.shape)}.")
    batch, _, c1 = a.shape
    c2 = b.shape[2]
    out = np.zeros(a.shape, dtype=b.dtype)
    permutations: List[Tuple[Optional[int], ...]] = []
    costs = []
    for k in range(batch):
        cost = _pair_costs(a[k], b[k], cost_func)
        costs.append(cost)
        padded = cost
        if c2 > c1:   # every class of y2 must be matched to something: dummy classes of y1 take the rest
            padded = np.concatenate([cost, np.full((c2 - c1, c2), cost.max() + 1, dtype=cost.dtype)], axis=0)
        chosen: List[Optional[int]] = [None] * c1
        for i, j in zip(*linear_sum_assignment(padded)):
            if i < c1:
                chosen[i] = int(j)
                out[k, :, i] = b[k, :, j]
        permutations.append(tuple(chosen))
    result = torch.from_numpy(out) if as_torch else out
    if return_cost:
        stacked = np.stack(costs) if costs else np.zeros((0, c1, c2))
        return result, permutations, (torch.from_numpy(stacked) if as_torch else stacked)
    return result, permutations
