"""Per-phase cycle stamps of the Winograd kernel (development aid; needs a library built with
-DPA_WINO_DEBUG: `PA_EXTRA_FLAGS=-DPA_WINO_DEBUG python pyannote-audio_amd/_build.py --force`).
usage: python tools/wino_phases.py [shape index 0..3] [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch
import pyannote_audio_amd.ffi as ffi
from pyannote_audio_amd.weights import winograd_pack, winograd_weights

si = int(sys.argv[1]) if len(sys.argv) > 1 else 3
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
dev = torch.device("cuda:0")
lib = ffi.load()
H, W, ci = [(80, 998, 32), (40, 499, 64), (20, 250, 128), (10, 125, 256)][si]
X = torch.randn(B, H, W, ci, device=dev)
Wg = torch.randn(9, ci, ci) * 0.05
sh = torch.randn(ci, device=dev)
R = torch.randn(B, H, W, ci, device=dev)
Y = torch.empty(B, H, W, ci, device=dev)
Ug = winograd_pack(winograd_weights(Wg.permute(1, 2, 0).reshape(ci, ci, 3, 3))).to(dev)
dbg = torch.zeros(64 * 2 * 8, dtype=torch.int64, device=dev)
def run():
    ffi.check(lib.pa_conv3x3_wino(ffi.ptr(X), B, H, W, ci, ffi.ptr(Ug), ffi.ptr(sh), ffi.ptr(R), ffi.ptr(Y),
                                  ci, 1, ffi.stream()), "wino")
run(); torch.cuda.synchronize()
lib.pa_wino_debug_buffer(C.c_void_p(dbg.data_ptr()))
run(); torch.cuda.synchronize()
d = dbg.cpu().view(64, 2, 8)
names = ["prep: DMA issue", "zero+transform", "wait vmcnt", "barrier", "mfma", "barrier", "epilogue/bookkeeping", "loop"]
for h in range(2):
    valid = int((d[:, h, 0] != 0).sum().item())
    lo, hi = (4, 40) if valid >= 41 else (2, max(3, valid - 1))
    print(f"half {h}: cycles per segment, stages {lo}..{hi} ({valid} stamped)")
    rows = []
    for i in range(lo, hi):
        st = d[i, h]
        seg = [int(st[k + 1] - st[k]) for k in range(7)] + [int(d[i + 1, h, 0] - st[7])]
        rows.append(seg)
    import numpy as np
    rows = np.array(rows)
    for k, n in enumerate(names):
        print(f"  {n:28s} mean {rows[:, k].mean():8.0f}  min {rows[:, k].min():6d}  max {rows[:, k].max():6d}")
    print(f"  stage total mean {rows.sum(1).mean():.0f}")
