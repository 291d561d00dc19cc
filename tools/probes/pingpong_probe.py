"""What runs beside an f32 MFMA stream on the same SIMD?  (development probe; build: tools/probes/build.sh)"""
import ctypes as C, os
import torch
lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "_dbg", "libprobes.so"))
dev = torch.device("cuda:0")
src = torch.randn(256 * 16384 + 65536, device=dev)
out = torch.zeros(256 * 8, dtype=torch.int64, device=dev)
sink = torch.zeros(512, device=dev)
def run(label, iters=200, mfma=1, breads=0, valu=0, dsr=0, dma=0, grid=256):
    for _ in range(2):
        out.zero_()
        lib.pingpong_probe(C.c_void_p(src.data_ptr()), iters, mfma, breads, valu, dsr, dma,
                           C.c_void_p(out.data_ptr()), C.c_void_p(sink.data_ptr()), grid, None)
        torch.cuda.synchronize()
    o = out.view(256, 8)[:grid].double()
    c = o[:, :4].mean().item() / iters
    p = o[:, 4:].mean().item() / iters
    print(f"{label:58s} MFMA half {c:7.0f} clk/iter ({c / 128:5.1f} per MFMA)   other half {p:7.0f} clk/iter", flush=True)
import time
def wall(label, data_zero, iters=4000):
    x = torch.zeros_like(src) if data_zero else src
    lib.pingpong_probe(C.c_void_p(x.data_ptr()), 50, 1, 0, 0, 0, 0, C.c_void_p(out.data_ptr()),
                       C.c_void_p(sink.data_ptr()), 256, None)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    lib.pingpong_probe(C.c_void_p(x.data_ptr()), iters, 1, 0, 0, 0, 0, C.c_void_p(out.data_ptr()),
                       C.c_void_p(sink.data_ptr()), 256, None)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    cyc = out.view(256, 8)[:, :4].double().mean().item()
    flop = 256 * 4 * iters * 128 * 2048.0
    print(f"{label}: {flop / ms / 1e9:7.1f} TFLOP/s over {ms:.1f} ms; {cyc / iters / 128:.2f} clk per MFMA; "
          f"effective shader clock {cyc / ms / 1e6:.3f} GHz", flush=True)
wall("bare v_mfma_f32_16x16x4_f32 stream, 1 wave/SIMD, random data", False)
wall("bare v_mfma_f32_16x16x4_f32 stream, 1 wave/SIMD, zero data  ", True)
run("128 MFMA alone")
for k in (2, 3, 4, 6, 8, 10):
    run(f"same wave: 1 MFMA + {k - 2} independent v_add, x128", mfma=k)

run("128 MFMA + B-fragment ds_reads", breads=1)
run("other half alone: 256 VALU", mfma=0, valu=256)
run("other half alone: 16 ds_read_b128", mfma=0, dsr=16)
run("other half alone: 10 DMA pieces", mfma=0, dma=10)
run("MFMA | 128 VALU", valu=128)
run("MFMA | 256 VALU", valu=256)
run("MFMA | 512 VALU", valu=512)
run("MFMA | 1024 VALU", valu=1024)
run("MFMA+Breads | 256 VALU", breads=1, valu=256)
run("MFMA | 16 ds_read_b128", dsr=16)
run("MFMA+Breads | 48 ds_read_b128", breads=1, dsr=48)
run("MFMA | 10 DMA pieces", dma=10)
run("MFMA+Breads | 10 DMA pieces", breads=1, dma=10)
run("MFMA+Breads | 10 DMA + 16 ds_read + 256 VALU", breads=1, dma=10, dsr=16, valu=256)
run("MFMA+Breads | 5 DMA + 16 ds_read + 256 VALU", breads=1, dma=5, dsr=16, valu=256)
