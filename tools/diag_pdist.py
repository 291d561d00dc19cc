import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from scipy.spatial.distance import pdist, cdist
from pyannote_audio_amd import distance
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
for n, d in ((700, 256), (129, 37), (2, 256)):
    X = rng.standard_normal((n, d)).astype(np.float32)
    X /= np.linalg.norm(X, axis=1, keepdims=True)
    got = distance.pdist_euclidean(X, device=dev)
    want = pdist(X, metric="euclidean")
    bad = np.nonzero(got != want)[0]
    print(n, d, "pdist mismatches", len(bad), "of", len(want))
    if len(bad):
        i = bad[0]
        print("  first", i, repr(got[i]), repr(want[i]), "ulps", (got[i]-want[i])/np.spacing(want[i]))
        print("  sq equal?", np.sum((got**2 != want**2)))
        print("  max ulps", np.max(np.abs(got-want)/np.spacing(want)))
    A = rng.standard_normal((50, d)).astype(np.float32); B = rng.standard_normal((7, d)).astype(np.float32)
    g2 = distance.cdist(A, B, metric="cosine", device=dev); w2 = cdist(A, B, metric="cosine")
    print("  cdist mismatches", int((g2 != w2).sum()), "max ulps", np.max(np.abs(g2-w2)/np.spacing(np.abs(w2)+1e-300)))
# sqrt check
x = torch.rand(1000000, dtype=torch.float64, device=dev) * 4
s = torch.sqrt(x).cpu().numpy(); print("torch sqrt mismatches vs numpy", int((s != np.sqrt(x.cpu().numpy())).sum()))
