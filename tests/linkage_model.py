"""Python model of the heap-free centroid-linkage merge of csrc/linkage_fast.hip (test infrastructure): the same
state (lower bound, neighbour candidate, EXACT bit, size, id per row), the same update rules, the same give-up rule
(a pop whose smallest bound is attained twice -> None, the launcher then runs the heap kernel).  The CPU suite checks
the MODEL against scipy.cluster.hierarchy.linkage(., "centroid") (tests/test_linkage_fast_model.py); the GPU suite
checks the KERNEL against SciPy (tests/test_pipeline_gpu.py)."""
import numpy as np
from scipy.spatial.distance import pdist, squareform


def fast_linkage_model(X):
    """-> (Z or None, merge index of the give-up or number of lower-bound repairs)"""
    n = len(X)
    S = squareform(pdist(X))
    size = np.ones(n, int); cid = np.arange(n)
    mind = np.full(n, np.inf); nb = np.full(n, -1); ex = np.zeros(n, bool)
    for x in range(n-1):
        row = S[x, x+1:]
        j = int(np.argmin(row)); mind[x] = row[j]; nb[x] = x+1+j; ex[x] = True
    Z = np.zeros((n-1, 4)); reps = 0
    for k in range(n-1):
        while True:
            m = mind.min()
            idx = np.nonzero(mind == m)[0]
            if len(idx) > 1 or not np.isfinite(m):
                return None, k
            x = int(idx[0]); dist = m; y = nb[x]
            if ex[x] and y >= 0: break
            best, bi = np.inf, -1
            for i in range(x+1, n):
                if size[i] and S[x, i] < best: best, bi = S[x, i], i
            nb[x] = bi; mind[x] = best; ex[x] = bi >= 0; reps += 1
        nx, ny = size[x], size[y]
        a, b = sorted((cid[x], cid[y]))
        Z[k] = (a, b, dist, nx+ny)
        best, bi = np.inf, -1
        for z in range(n):
            if z == x or z == y or size[z] == 0: continue
            dx, dy = S[x, z], S[y, z]
            nd = np.sqrt((((nx*dx*dx) + (ny*dy*dy)) - ((nx*ny)*dist*dist)/(nx+ny)) / (nx+ny))
            S[y, z] = nd; S[z, y] = nd
            if z < y:
                c = nb[z]
                if nd < mind[z]:
                    mind[z] = nd; nb[z] = y; ex[z] = True
                elif c == x or c == y:
                    nb[z] = y; ex[z] = mind[z] == nd
            elif nd < best:
                best, bi = nd, z
        mind[x] = np.inf; size[x] = 0; size[y] = nx+ny; cid[y] = n+k
        if y < n-1 and bi >= 0:
            nb[y] = bi; mind[y] = best; ex[y] = True
    return Z, reps


def late_tie_points():
    """600 unit vectors with four close pairs (u, v) and their mirror images (-u, -v): the mirrored pairs have the
    same float64 distance and meet as the two smallest lower bounds at merge 72 (not at the first pop)."""
    rng = np.random.default_rng(99)
    X = rng.standard_normal((600, 24)).astype(np.float32)
    X /= np.linalg.norm(X, axis=1, keepdims=True)
    for a in range(0, 8, 2):
        v = X[a] + 0.35 * (1 + a / 8) * rng.standard_normal(24).astype(np.float32)
        X[a + 1] = v / np.linalg.norm(v)
        X[500 + a], X[501 + a] = -X[a], -X[a + 1]
    return X
