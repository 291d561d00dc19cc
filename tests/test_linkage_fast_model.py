"""The algorithm of csrc/linkage_fast.hip (heap-free centroid linkage with an EXACT bit per row and a tie give-up),
as a Python model, against SciPy -- the reference's `linkage(..., method="centroid")` call
(pipelines/clustering.py:374-382).  The kernel itself is checked on the GPU (tests/test_pipeline_gpu.py)."""
import numpy as np
import pytest
from scipy.cluster.hierarchy import linkage
from scipy.spatial.distance import pdist

from linkage_model import fast_linkage_model


def _points(n, d, seed):
    rng = np.random.default_rng(seed)
    c = rng.standard_normal((4, d))
    X = (c[rng.integers(0, 4, n)] + 0.5 * rng.standard_normal((n, d))).astype(np.float32)
    X /= np.linalg.norm(X, axis=1, keepdims=True)
    return X.astype(np.float64)


@pytest.mark.parametrize("n,d,seed,G,CH", [(3, 4, 0, 1, 1024), (4, 4, 1, 2, 1), (10, 3, 2, 2, 2), (60, 8, 3, 3, 4),
                                            (150, 16, 4, 1, 1024), (300, 4, 5, 16, 4), (200, 8, 6, 5, 3)])
def test_model_is_scipy_when_no_pop_ties(n, d, seed, G, CH):
    """G workgroups owning interleaved chunks of CH rows, candidate exchange with two candidates per workgroup"""
    X = _points(n, d, seed)
    Z, repairs = fast_linkage_model(X, G, CH)
    assert Z is not None, "unexpected tie in random data"
    assert np.array_equal(Z, linkage(pdist(X), "centroid"))
    assert repairs > 0 or n < 10      # the lower-bound repair path is exercised


def test_model_gives_up_on_duplicated_rows():
    X = _points(40, 8, 7)
    X[5] = X[17]
    X[9] = X[17]             # three identical rows: rows 5 and 9 both hold the bound 0.0 at the first pop
    Z, at = fast_linkage_model(X)
    assert Z is None and at == 0


def test_model_gives_up_on_a_late_tie():
    """mirrored pairs (u, v) and (-u, -v): the same float64 distance, tied only once both are the smallest bounds"""
    from linkage_model import late_tie_points
    for G, CH in ((1, 1024), (4, 16)):
        Z, at = fast_linkage_model(late_tie_points().astype(np.float64), G, CH)
        assert Z is None and at == 72
