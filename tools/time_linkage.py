"""Times the dendrogram merge kernel (csrc/linkage.hip) for 1 .. 16 workgroups and checks every variant against
SciPy (small n) or against the single-workgroup kernel (large n).
usage (GPU box): python tools/time_linkage.py [n ...]     (default 7000 20000 57000)
If gpurun_out/bench_train_emb.npy exists (PA_BENCH_DUMP_EMB=1 python bench.py ...) it is timed first."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from scipy.cluster.hierarchy import linkage
from scipy.spatial.distance import pdist
from pyannote_audio_amd import distance

dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(X, G):
    os.environ["PA_LINKAGE_WGS"] = str(G)
    distance.linkage_centroid(X[:64].copy(), dev)
    torch.cuda.synchronize()
    t = time.perf_counter()
    Z = distance.linkage_centroid(X.copy(), dev)
    dt = time.perf_counter() - t
    return Z, dt, describe(len(X))


def describe(n):
    st = distance.last_linkage_stats
    fast = (f"heap-free merge: status {st[8]} workgroups {st[13]} repairs/merge {st[9] / n:.2f} re-publishes/merge "
            f"{st[14] / n:.2f} cycles(pop, pass)/merge {[int(c / n) for c in st[10:12]]}") if st[12] else "heap-free merge: off"
    heap = (f"heap kernel: retries/merge {st[0] / n:.2f} heap-updates/merge {st[1] / n:.2f} overflows {st[2]} "
            f"cycles(find, wait, pass, replay)/merge {[int(c / n) for c in st[3:7]]}") if st[7] else "heap kernel: skipped"
    return fast + " | " + heap


def sweep(name, X, scipy_check):
    ref = None
    if scipy_check:
        t = time.perf_counter()
        ref = linkage(pdist(X), "centroid")
        print(f"{name}: scipy pdist + linkage {1e3 * (time.perf_counter() - t):.0f} ms", flush=True)
    for G in [int(g) for g in os.environ.get("LK_GS", "0,-8,-16,1").split(",")]:
        # G = 0: the default (heap-free merge first, its own choice of workgroups); G < 0: the heap-free merge forced
        # to -G workgroups; G >= 1: PA_LINKAGE_FAST=0, the heap kernel with G workgroups
        if G <= 0:
            os.environ.pop("PA_LINKAGE_FAST", None)
            os.environ.pop("PA_LINKAGE_WGS", None)
            os.environ.pop("PA_LINKAGE_FAST_WGS", None)
            if G < 0:
                os.environ["PA_LINKAGE_FAST_WGS"] = str(-G)
            distance.linkage_centroid(X[:64].copy(), dev)
            torch.cuda.synchronize()
            t = time.perf_counter()
            Z = distance.linkage_centroid(X.copy(), dev)
            dt = time.perf_counter() - t
            info = describe(len(X))
            os.environ.pop("PA_LINKAGE_FAST_WGS", None)
        else:
            os.environ["PA_LINKAGE_FAST"] = "0"
            Z, dt, info = run(X, G)
            os.environ.pop("PA_LINKAGE_FAST", None)
        if ref is None:
            ref = Z
        print(f"{name}: n={len(X)} workgroups={G:2d} {1e3 * dt:8.1f} ms  identical={np.array_equal(Z, ref)}  {info}",
              flush=True)


emb_file = os.path.join(ROOT, "gpurun_out", "bench_train_emb.npy")
if os.path.exists(emb_file):
    X = np.load(emb_file)
    X = X / np.linalg.norm(X, axis=1, keepdims=True)
    sweep("bench embeddings", X, True)
for n in [int(a) for a in sys.argv[1:]] or [7000, 20000, 57000]:
    c = rng.standard_normal((4, 256))
    X = (c[rng.integers(0, 4, n)] + 0.6 * rng.standard_normal((n, 256))).astype(np.float32)
    X /= np.linalg.norm(X, axis=1, keepdims=True)
    sweep("synthetic", X, n <= 7000)
    if n >= 20000 and os.environ.get("LK_NODUP") != "1":           # exact ties (duplicated rows): the heap-free merge gives up at its first pop
        X[n // 2: n // 2 + 500] = X[:500]
        sweep("synthetic + 500 duplicated rows", X, False)
os.environ.pop("PA_LINKAGE_WGS", None)
