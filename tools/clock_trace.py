"""Shader-clock trace under load (the evidence behind `sustained_mfma_tflops` in bench.py).

Samples `rocm-smi --showclocks` (sclk) every ~50 ms in a background thread while (a) the bare
v_mfma_f32_16x16x4_f32 stream of tools/probes runs on random / zero operands and (b) the Winograd
convolution runs on the ResNet34 layer shapes.  usage (GPU box): python tools/clock_trace.py"""
import ctypes as C, os, re, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

samples, stop = [], False
def sampler():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showclocks"], capture_output=True, text=True, timeout=5).stdout
            m = re.search(r"sclk clock level:\s*\d+:?\s*\(?(\d+)Mhz", out)
            if m:
                samples.append((time.perf_counter(), int(m.group(1))))
        except Exception as e:  # noqa: BLE001
            samples.append((time.perf_counter(), -1))
        time.sleep(0.02)

def window(t0, t1):
    v = [c for t, c in samples if t0 <= t <= t1 and c > 0]
    return (min(v), sum(v) / len(v), max(v), len(v)) if v else None

th = threading.Thread(target=sampler, daemon=True)
th.start()
dev = torch.device("cuda:0")
lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "_dbg", "libprobes.so"))
src = torch.randn(256 * 16384 + 65536, device=dev)
out = torch.zeros(256 * 8, dtype=torch.int64, device=dev)
sink = torch.zeros(512, device=dev)
time.sleep(1.0)
print("idle sclk (MHz) min/mean/max/n:", window(0, time.perf_counter()))
for label, x in (("random operands", src), ("zero operands", torch.zeros_like(src))):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    iters = 400000   # ~1 s of back-to-back MFMAs
    lib.pingpong_probe(C.c_void_p(x.data_ptr()), iters, 1, 0, 0, 0, 0, C.c_void_p(out.data_ptr()),
                       C.c_void_p(sink.data_ptr()), 256, None)
    e1.record()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    ms = e0.elapsed_time(e1)
    cyc = out.view(256, 8)[:, :4].double().mean().item()
    print(f"bare v_mfma_f32_16x16x4_f32 stream, {label}: {256 * 4 * iters * 128 * 2048.0 / ms / 1e9:6.1f} TFLOP/s "
          f"over {ms:.0f} ms; s_memtime cycles / wall = {cyc / ms / 1e6:.3f} GHz; "
          f"rocm-smi sclk min/mean/max/n = {window(t0 + 0.1, t1)}", flush=True)
# the Winograd kernel on the four stride-1 layer shapes (random activations)
import pyannote_audio_amd.ffi as ffi
from pyannote_audio_amd.weights import winograd_pack, winograd_weights
plib = ffi.load()
for (H, W, ci) in ((80, 998, 32), (40, 499, 64), (20, 250, 128), (10, 125, 256)):
    B = 256
    X = torch.randn(B, H, W, ci, device=dev)
    Ug = winograd_pack(winograd_weights(torch.randn(ci, ci, 3, 3) * 0.05)).to(dev)
    sh = torch.randn(ci, device=dev)
    R = torch.randn(B, H, W, ci, device=dev)
    Y = torch.empty(B, H, W, ci, device=dev)
    def run():
        ffi.check(plib.pa_conv3x3_wino(ffi.ptr(X), B, H, W, ci, ffi.ptr(Ug), ffi.ptr(sh), ffi.ptr(R), ffi.ptr(Y),
                                       ci, 1, ffi.stream()), "wino")
    run(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    n = 300
    for _ in range(n):
        run()
    e1.record(); torch.cuda.synchronize()
    t1 = time.perf_counter()
    ms = e0.elapsed_time(e1) / n
    ex = 2.0 * 9 * ci * ci * B * H * W * 16 / 36 / ms / 1e9
    print(f"k_conv3x3_wino {H}x{W}x{ci}: {ms:.3f} ms, {ex:.1f} TFLOP/s executed; rocm-smi sclk "
          f"min/mean/max/n = {window(t0 + 0.1, t1)}", flush=True)
    del X, R, Y
stop = True
