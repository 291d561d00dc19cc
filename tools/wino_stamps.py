"""Phase timeline of k_conv3x3_wino from an instrumented build (-DPA_WINO_STAMP=1 -DPA_WINO_PF=0,
tools/build_variants.py tag `stamp`): s_memtime stamps per wave and stage of the first 16 workgroups.
usage: PA_LIB=.../libpa_stamp.so python tools/wino_stamps.py out.npz"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import pyannote_audio_amd.ffi as ffi
from pyannote_audio_amd.weights import winograd_pack, winograd_weights

dev = torch.device("cuda:0")
lib = ffi.load()
lib.pa_wino_read_stamps.argtypes = [C.c_void_p, C.c_int]
B = int(os.environ.get("B", "256"))
shapes = [(80, 998, 32, True), (40, 499, 64, True), (20, 250, 128, True), (10, 125, 256, True), (80, 998, 32, False)]
out = {}
for (H, W, c, res) in shapes:
    X = torch.randn(B, H, W, c, device=dev)
    Wg = torch.randn(c, c, 3, 3) * 0.05
    U = winograd_pack(winograd_weights(Wg)).to(dev)
    sh = torch.randn(c, device=dev)
    R = torch.randn(B, H, W, c, device=dev) if res else None
    Y = torch.empty(B, H, W, c, device=dev)
    def run():
        ffi.check(lib.pa_conv3x3_wino(ffi.ptr(X), B, H, W, c, ffi.ptr(U), ffi.ptr(sh), ffi.ptr(R), ffi.ptr(Y),
                                      c, 1, ffi.stream()), "wino")
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    buf = np.zeros(16 * 4 * 64 * 8, dtype=np.uint64)
    lib.pa_wino_read_stamps(buf.ctypes.data, 1)
    run()
    torch.cuda.synchronize()
    lib.pa_wino_read_stamps(buf.ctypes.data, 1)
    st = buf.reshape(16, 4, 64, 8).astype(np.int64)
    out[f"{H}x{W}x{c}_res{int(res)}"] = st
    # quick summary: median phase durations over stages 4..60 of all recorded waves
    d = np.diff(st[:, :, 4:60, :], axis=-1).reshape(-1, 7)
    nxt = (st[:, :, 5:61, 0] - st[:, :, 4:60, 0]).reshape(-1)
    names = ["barrier1", "dma_issue", "vmcnt_wait", "barrier2", "transform", "mfma", "epilogue/none"]
    print(f"{H}x{W}x{c} res={int(res)}: stage period median {np.median(nxt):.0f} cycles; " +
          ", ".join(f"{n} {np.median(d[:, i]):.0f}" for i, n in enumerate(names)), flush=True)
np.savez_compressed(sys.argv[1], **out)
