"""Python model of the heap-free centroid-linkage merge of csrc/linkage_fast.hip (test infrastructure): the same
per-row state (lower bound, neighbour candidate, EXACT bit, size, id), the same update rules, the same candidate
exchange between workgroups (every workgroup publishes its two smallest rows; the workgroup of a repaired row
re-publishes them without that row, which travels as an extra candidate; so does the freshly merged row), the same give-up rule (a pop
whose smallest bound is attained twice -> None, the launcher then runs the heap kernel of csrc/linkage.hip).
The CPU suite checks the MODEL against scipy.cluster.hierarchy.linkage(., "centroid")
(tests/test_linkage_fast_model.py); the GPU suite checks the KERNEL against SciPy (tests/test_pipeline_gpu.py)."""
import numpy as np
from scipy.spatial.distance import pdist, squareform

INF = np.inf


def fast_linkage_model(X, G=1, CH=256, stats=None):
    """-> (Z or None, merge index of the give-up or number of lower-bound repairs).  G workgroups own the rows in
    interleaved chunks of CH rows (row z belongs to workgroup (z // CH) % G).  Candidate table: per workgroup its two
    smallest rows (computed without the row merged last and without the row it is repairing) + one extra entry (the
    row repaired last in that workgroup), + the row merged last."""
    n = len(X)
    S = squareform(pdist(X))
    size = np.ones(n, int)
    cid = np.arange(n)
    mind = np.full(n, INF)
    nb = np.full(n, -1)
    ex = np.zeros(n, bool)
    for x in range(n - 1):
        row = S[x, x + 1:]
        j = int(np.argmin(row))
        mind[x] = row[j]
        nb[x] = x + 1 + j
        ex[x] = True
    owner = lambda z: (z // CH) % G
    rows_of = [[z for z in range(n) if owner(z) == g] for g in range(G)]
    Z = np.zeros((n - 1, 4))
    reps = 0

    def local_top2(g, exclude):
        c = sorted((mind[z], z) for z in rows_of[g] if z not in exclude and mind[z] < INF)
        return (c + [None, None])[:2]

    pend_y, passbest = -1, None
    k = 0
    while k < n - 1:
        # ---- main exchange: everybody's two smallest rows (without the row merged last) + that row's new bound
        y_excl = pend_y
        top2 = [local_top2(g, (y_excl,)) for g in range(G)]
        extra = [None] * G
        yentry = None
        if pend_y >= 0:
            d, i = passbest
            if pend_y < n - 1:
                if i < 0:
                    return None, k
                mind[pend_y], nb[pend_y], ex[pend_y] = d, i, True
                yentry = (d, pend_y)
            pend_y = -1
        tries = 0
        while True:
            C = [c for g in range(G) for c in top2[g] + [extra[g]] if c is not None]
            if yentry is not None:
                C.append(yentry)
            C.sort()
            if not C:
                return None, k
            if len(C) > 1 and C[1][0] == C[0][0]:
                return None, k                      # tie at the pop: take the heap
            dist, x = C[0]
            y = nb[x]
            if ex[x] and y >= 0:
                break
            tries += 1
            if tries >= n - k:
                return None, k
            # ---- scan exchange: repaired bound of row x; its workgroup publishes a fresh top-2 without x
            best, bi = INF, -1
            for i in range(x + 1, n):
                if size[i] and S[x, i] < best:
                    best, bi = S[x, i], i
            nb[x], mind[x], ex[x] = bi, best, bi >= 0
            reps += 1
            g = owner(x)
            top2[g] = local_top2(g, (y_excl, x))
            extra[g] = (best, x) if bi >= 0 else None
        # ---- merge
        nx, ny = size[x], size[y]
        a, b = sorted((cid[x], cid[y]))
        Z[k] = (a, b, dist, nx + ny)
        best, bi = INF, -1
        for z in range(n):
            if z == x or z == y or size[z] == 0:
                continue
            dx, dy = S[x, z], S[y, z]
            nd = np.sqrt((((nx * dx * dx) + (ny * dy * dy)) - ((nx * ny) * dist * dist) / (nx + ny)) / (nx + ny))
            S[y, z] = nd
            S[z, y] = nd
            if z < y:
                c = nb[z]
                if nd < mind[z]:
                    mind[z], nb[z], ex[z] = nd, y, True
                elif c == x or c == y:
                    nb[z], ex[z] = y, mind[z] == nd
            elif nd < best:
                best, bi = nd, z
        mind[x] = INF
        size[x] = 0
        size[y] = nx + ny
        cid[y] = n + k
        pend_y, passbest = y, (best, bi)
        k += 1
    if stats is not None:
        stats.update(repairs=reps)
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
