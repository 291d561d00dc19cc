"""Phase anatomy of k_conv3x3_wino4p (development aid): runs one layer shape with the instrumented library
(tools/build_variants.py: libpa_w4pstamp.so) and prints, per wave of workgroup 0, the mean cycles of a stage spent
in the ticks: input transform | sync | MFMA run (with the interleaved DMA issue) | sync; and for a tile's last
stage the epilogue marks.
usage: PA_LIB=pyannote-audio_amd/build/variants/libpa_w4pstamp.so python tools/wino4p_stamps.py [cin H W B]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import pyannote_audio_amd.ffi as ffi
from pyannote_audio_amd.weights import winograd4_pack, winograd4_weights

cin, H, W, B = (int(a) for a in sys.argv[1:5]) if len(sys.argv) >= 5 else (128, 20, 250, 512)
dev = torch.device("cuda:0")
lib = ffi.load()
X = torch.randn(B, H, W, cin, device=dev)
U = winograd4_pack(winograd4_weights(torch.randn(cin, cin, 3, 3) * 0.05)).to(dev)
sh = torch.randn(cin, device=dev)
Y = torch.empty(B, H, W, cin, device=dev)
for _ in range(3):
    ffi.check(lib.pa_conv3x3_wino4_kernel(ffi.ptr(X), B, H, W, cin, ffi.ptr(U), ffi.ptr(sh), None, ffi.ptr(Y), cin, 1,
                                          1, ffi.stream()), "wino4p")
torch.cuda.synchronize()
buf = np.zeros(8 * 8 * 64 * 10, dtype=np.uint64)
lib.pa_wino4p_read_stamps.argtypes = [C.c_void_p]
assert lib.pa_wino4p_read_stamps(buf.ctypes.data) == 0
st = buf.reshape(8, 8, 64, 10).astype(np.int64)
nst = cin // 8
names = ["transform", "sync", "mfma run", "sync"]
for wg in (0, 1):
    for wave in range(8):
        d = np.diff(st[wg, wave, :, :5], axis=1)                  # (64, 4)
        e = st[wg, wave, :, 5:10] - st[wg, wave, :, 4:5]           # epilogue marks relative to the tick behind the last run
        full = st[wg, wave, 1:, 0] - st[wg, wave, :-1, 0]          # stage to stage
        last = (np.arange(64) % nst) == nst - 1
        print(f"wg {wg} wave {wave}: " + " | ".join(f"{n} {d[~last, i].mean():.0f}" for i, n in enumerate(names)) +
              f" || stage {full[~last[:-1]].mean():.0f} cycles; tile's last stage to next tile {full[last[:-1]].mean():.0f}; "
              f"epilogue marks (E1 done, E2 starts, E3 done, E4 starts, E4 done) {[int(x) for x in e[last].mean(axis=0)]}")
