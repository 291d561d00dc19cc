import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from scipy.cluster.hierarchy import linkage
from scipy.spatial.distance import pdist
from pyannote_audio_amd import distance
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
for n in (1000, 3000, 7000, 10773):
    c = rng.standard_normal((4, 256))
    X = (c[rng.integers(0, 4, n)] + 0.6 * rng.standard_normal((n, 256))).astype(np.float32)
    X /= np.linalg.norm(X, axis=1, keepdims=True)
    distance.linkage_centroid(X[:100], dev)
    t = time.perf_counter(); Z = distance.linkage_centroid(X, dev); tg = time.perf_counter() - t
    if n <= 7000:
        t = time.perf_counter(); Zs = linkage(pdist(X), "centroid"); ts = time.perf_counter() - t
        print(n, f"gpu {tg*1e3:.1f} ms  scipy(pdist+linkage) {ts*1e3:.1f} ms  equal={np.array_equal(Z, Zs)}", flush=True)
    else:
        print(n, f"gpu {tg*1e3:.1f} ms", flush=True)
