"""What does a long single-workgroup kernel on a second stream do to the persistent convolution kernels?
(tools/probes/hog_probe.hip; the real dendrogram merge for comparison).
usage: sh tools/probes/build.sh && python tools/probes/interference_probe.py"""
import ctypes as C, os, sys, time
here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(here)))
import torch
import pyannote_audio_amd.ffi as ffi
from pyannote_audio_amd.weights import winograd_pack, winograd_weights

probes = C.CDLL(os.path.join(here, "..", "_dbg", "libprobes.so"))
probes.hog_probe.argtypes = [C.c_longlong, C.c_int, C.c_void_p, C.c_longlong, C.c_longlong, C.c_void_p, C.c_int, C.c_void_p]
dev = torch.device("cuda:0")
lib = ffi.load()
B, H, W, c = 256, 20, 250, 128
X = torch.randn(B, H, W, c, device=dev)
U = winograd_pack(winograd_weights(torch.randn(c, c, 3, 3) * 0.05)).to(dev)
sh = torch.randn(c, device=dev)
R = torch.randn(B, H, W, c, device=dev)
Y = torch.empty(B, H, W, c, device=dev)
side = torch.cuda.Stream(device=dev)
n = 7176
nd = n * (n - 1) // 2
buf = torch.rand(nd, dtype=torch.float64, device=dev)
sink = torch.zeros(1024, dtype=torch.float64, device=dev)
emb = torch.randn(n, 256, dtype=torch.float64, device=dev)
Z = torch.empty((n - 1, 4), dtype=torch.float64, device=dev)
ws = torch.empty(lib.pa_linkage_workspace_bytes(n), dtype=torch.uint8, device=dev)
cond = torch.empty(nd, dtype=torch.float64, device=dev)


def convs(k):
    for _ in range(k):
        ffi.check(lib.pa_conv3x3_wino(ffi.ptr(X), B, H, W, c, ffi.ptr(U), ffi.ptr(sh), ffi.ptr(R), ffi.ptr(Y), c, 1,
                                      ffi.stream()), "wino")


def timed(label, hog, reserved):
    convs(10)
    torch.cuda.synchronize()
    if hog is not None:
        with torch.cuda.stream(side):
            hog()
    time.sleep(0.01)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    convs(60)
    e1.record()
    torch.cuda.synchronize()
    print(f"{label:46s} reserved={reserved}: {e0.elapsed_time(e1) / 60:.3f} ms per conv", flush=True)


CYC = int(0.25 * 2.3e9)
def spin(mode, stride=7001, lds_kb=150):
    return lambda: probes.hog_probe(CYC, mode, buf.data_ptr(), nd, stride, sink.data_ptr(), lds_kb,
                                    torch.cuda.current_stream().cuda_stream)
def merge():
    ffi.check(lib.pa_pdist_f64(ffi.ptr(emb), n, 256, ffi.ptr(cond), ffi.stream()), "pdist")
    ffi.check(lib.pa_linkage_centroid_f64(ffi.ptr(cond), n, ffi.ptr(Z), ffi.ptr(ws), ws.numel(), ffi.stream()), "link")

which = os.environ.get("PROBE", "full")
if which == "full":
    for reserved in (0, 1):
        timed("alone", None, reserved)
        timed("hog: spin only, 150 KB LDS", spin(0), reserved)
        timed("hog: spin only, 8 KB LDS", spin(0, lds_kb=8), reserved)
        timed("hog: scattered reads", spin(1), reserved)
        timed("hog: scattered read-modify-write", spin(2), reserved)
        timed("hog: contiguous reads", spin(3), reserved)
        timed("real dendrogram merge (n = 7176)", merge, reserved)
else:
    for reserved in (0,):
        timed("alone", None, reserved)
        timed("hog: spin only, 150 KB LDS", spin(0), reserved)
        timed("hog: spin only, 100 KB LDS", spin(0, lds_kb=100), reserved)
        timed("hog: spin only, 60 KB LDS", spin(0, lds_kb=60), reserved)
        timed("real dendrogram merge (n = 7176)", merge, reserved)
