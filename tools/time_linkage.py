import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from scipy.cluster.hierarchy import linkage
from scipy.spatial.distance import pdist
from pyannote_audio_amd import distance
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
emb_file = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "bench_train_emb.npy")
if os.path.exists(emb_file):
    X = np.load(emb_file); X = X / np.linalg.norm(X, axis=1, keepdims=True)
    t = time.perf_counter(); Z = distance.linkage_centroid(X.copy(), dev); tg = time.perf_counter() - t
    st = distance.last_linkage_stats; n = len(X)
    print("bench embeddings n=%d gpu %.1f ms retries/merge %.2f heap-updates/merge %.2f overflows %d cycles %s" % (
        n, tg * 1e3, st[0] / n, st[1] / n, st[2], [int(c / n) for c in st[3:7]]), flush=True)
for n in (1000, 3000, 7000):
    c = rng.standard_normal((4, 256))
    X = (c[rng.integers(0, 4, n)] + 0.6 * rng.standard_normal((n, 256))).astype(np.float32)
    X /= np.linalg.norm(X, axis=1, keepdims=True)
    distance.linkage_centroid(X[:100], dev)
    t = time.perf_counter(); Z = distance.linkage_centroid(X, dev); tg = time.perf_counter() - t
    st = distance.last_linkage_stats
    print("   retries/merge %.2f  heap-updates/merge %.2f  overflows %d  cycles(find,record,pass,replay)/merge %s" % (
        st[0] / n, st[1] / n, st[2], [int(c / n) for c in st[3:7]]), flush=True)
    if n <= 7000:
        t = time.perf_counter(); Zs = linkage(pdist(X), "centroid"); ts = time.perf_counter() - t
        print(n, f"gpu {tg*1e3:.1f} ms  scipy(pdist+linkage) {ts*1e3:.1f} ms  equal={np.array_equal(Z, Zs)}", flush=True)
    else:
        print(n, f"gpu {tg*1e3:.1f} ms", flush=True)
