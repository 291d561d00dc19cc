"""Stage A of VERDICT round 5, item 1: the Winograd-domain GEMM of the F(4x4) kernel on the bf16 matrix pipe through an
exact 3-way split (development probe; build: tools/probes/build.sh; run on the GPU box, output ->
profiles/r6_split_mfma_probe.txt).

part 1  accuracy on the Winograd-domain operands of real layers (seeded WeSpeaker ResNet34 of the test suite, BatchNorm
        folded; activations of a synthetic conversation at the inputs of layer2/3/4 stride-1 convolutions):
        native v_mfma_f32_16x16x4_f32 (product order) vs 9 / 7 / 6 products of v_mfma_f32_16x16x32_bf16, each against a
        float64 evaluation of the same float32 operands.
part 2  sustained rate of the bare MFMA streams.
part 3  cycles per 8-channel STAGE of a split-form kernel (instruction mix of csrc/emb_winograd4.hip with the split),
        parts switched off one by one."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import torch.nn.functional as F
from oracle.models import seeded_wespeaker
from oracle.synthetic import synth_conversation

lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "_dbg", "libprobes.so"))
dev = torch.device("cuda:0")
BT = torch.tensor([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0],
                   [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], dtype=torch.float32)
G = torch.tensor([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6],
                  [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], dtype=torch.float64)


def layer_operands(model, layer, block, conv, wav):
    """-> U [36][32][K] float32 (first 32 output channels), V [36][K][T] float32 (T a multiple of 16)"""
    blk = getattr(model.resnet, f"layer{layer}")[block]
    cv, bn = getattr(blk, f"conv{conv}"), getattr(blk, f"bn{conv}")
    grabbed = {}
    h = cv.register_forward_pre_hook(lambda m, inp: grabbed.setdefault("x", inp[0].detach().clone()))
    with torch.inference_mode():
        model(wav)
    h.remove()
    x = grabbed["x"]                                               # (B, K, H, W)
    scale = (bn.weight / torch.sqrt(bn.running_var + bn.eps)).detach()
    g = (cv.weight.detach() * scale.view(-1, 1, 1, 1)).double()
    U = torch.einsum("ap,oipq,bq->aboi", G, g, G).reshape(36, g.shape[0], g.shape[1]).float()[:, :32].contiguous()
    B, K, H, W = x.shape
    th, tw = -(-H // 4), -(-W // 4)
    xp = F.pad(x, (1, tw * 4 + 1 - W, 1, th * 4 + 1 - H))
    d = xp.unfold(2, 6, 4).unfold(3, 6, 4)                          # (B, K, th, tw, 6, 6)
    V = torch.einsum("ai,bcyxij,dj->adcbyx", BT, d, BT).reshape(36, K, -1)   # float32 transform
    T = V.shape[2] // 16 * 16
    return U, V[:, :, :T].contiguous()


def run_acc(U, V, mode):
    P, _, K = U.shape
    T = V.shape[2]
    Ud, Vd = U.to(dev), V.to(dev)
    out = torch.zeros((P, 32, T), dtype=torch.float32, device=dev)
    rc = lib.split_probe_acc(C.c_void_p(Ud.data_ptr()), C.c_void_p(Vd.data_ptr()), C.c_void_p(out.data_ptr()), P, K, T,
                             mode, None)
    torch.cuda.synchronize()
    assert rc == 0, rc
    return out.cpu()


NAMES = ["native v_mfma_f32_16x16x4_f32 (product order)", "split RNE, 6 products (2 MFMA per 8 cin)",
         "split RNE, 7 products (6 + lm)", "split RNE, 9 products (3 MFMA per 8 cin)",
         "split truncating, 6 products", "split truncating, 9 products"]
print("== part 1: accuracy against float64 on the same float32 operands ==")
model = seeded_wespeaker()
wav, _ = synth_conversation(12.0, seed=5)
wav = wav[:, :160000].unsqueeze(0)
for (layer, block, conv) in ((2, 1, 1), (3, 1, 1), (3, 4, 2), (4, 1, 1)):
    U, V = layer_operands(model, layer, block, conv, wav)
    ref = torch.einsum("pok,pkt->pot", U.double(), V.double())
    mag = torch.einsum("pok,pkt->pot", U.double().abs(), V.double().abs())
    print(f"layer{layer}[{block}].conv{conv}: K = {U.shape[2]} input channels, 36 points x 32 output channels x "
          f"{V.shape[2]} tiles; max |ref| {ref.abs().max():.3e}, rms {ref.pow(2).mean().sqrt():.3e}")
    base = None
    for mode, name in enumerate(NAMES):
        got = run_acc(U, V, mode).double()
        err = (got - ref).abs()
        r = dict(maxrel=(err.max() / ref.abs().max()).item(), rms=(err.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item(),
                 sumabs=(err / mag.clamp_min(1e-300)).max().item(), bias=((got - ref).mean() / ref.abs().mean()).item())
        if mode == 0:
            base = r
        print(f"  {name:48s} max|err|/max|ref| {r['maxrel']:.3e}  rms err/rms ref {r['rms']:.3e}  "
              f"max |err|/sum|u||v| {r['sumabs']:.3e}  mean err/mean|ref| {r['bias']:+.2e}   "
              f"(vs native: max x{r['maxrel'] / base['maxrel']:.2f}, rms x{r['rms'] / base['rms']:.2f})", flush=True)

print("\n== part 2: bare MFMA streams, one wave per SIMD, 256 workgroups ==")
src = torch.randn(1 << 16, device=dev)
out = torch.zeros(256 * 4, dtype=torch.int64, device=dev)
sink = torch.zeros(1024, device=dev)
rates = {}
for mode, (name, flop, iters) in enumerate((("v_mfma_f32_16x16x4_f32", 2048.0, 40000), ("v_mfma_f32_16x16x32_bf16", 16384.0, 80000))):
    for rep in range(2):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = lib.split_probe_stream(C.c_void_p(src.data_ptr()), iters, mode, C.c_void_p(out.data_ptr()),
                                    C.c_void_p(sink.data_ptr()), 256, None)
        e1.record()
        torch.cuda.synchronize()
        assert rc == 0, rc
    ms = e0.elapsed_time(e1)
    cyc = out.double().mean().item() / (iters * 16)
    tf = 256 * 4 * iters * 16 * flop / ms / 1e9
    rates[mode] = tf
    print(f"{name:28s} {tf:8.1f} TFLOP/s over {ms:6.1f} ms, {cyc:6.2f} cycles per MFMA per SIMD, "
          f"shader clock {out.double().mean().item() / ms / 1e6:.3f} GHz", flush=True)
print(f"f32-equivalent rate of the split GEMM (bf16 stream / products per f32 product): 9 products "
      f"{rates[1] / 9:.1f}, 7 or 6 products at 2 MFMA per 8 channels (8 k-slots of 32 carry 6 products) {rates[1] / 8:.1f}, "
      f"6 products densely packed {rates[1] / 6:.1f} TFLOP/s; native f32 MFMA {rates[0]:.1f}")

print("\n== part 3: one 8-channel stage of a split-form F(4x4) kernel (cycles per stage and wave; the f32 kernel: 7 500) ==")
span = 1 << 26
pool = torch.randn(span // 4, device=dev).bfloat16().float().view(torch.int32)
pool = (pool & -65536) | ((pool >> 16) & 0xffff)       # both halves of every word are finite bf16 numbers
variants = [(31, "everything: DMA + split + transform + U reads + MFMA"), (30, "no DMA"), (29, "no split (B tuples invariant)"),
            (27, "no transform"), (24, "U reads + MFMA only"), (16, "144 MFMAs only"),
            (15, "everything but the MFMAs"), (14, "split + transform + U reads"), (6, "split + transform"),
            (2, "split only"), (4, "transform only")]
libu = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "_dbg", "libprobes_unpacked.so"))
for unpacked, flags, name in [(0, f, n) for f, n in variants] + [(1, f, n + " [unpacked f32 VALU]") for f, n in variants[:6]]:
    for rep in range(2):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn = libu.split_probe_stage_unpacked if unpacked else lib.split_probe_stage
        rc = fn(C.c_void_p(pool.data_ptr()), span, 512, flags, C.c_void_p(out.data_ptr()),
                                   C.c_void_p(sink.data_ptr()), 256, None)
        e1.record()
        torch.cuda.synchronize()
        assert rc == 0, rc
    cyc = out.double() / 512
    print(f"flags {flags:2d} {name:54s} {cyc.mean().item():7.0f} cycles per stage (max {cyc.max().item():7.0f}), "
          f"{e0.elapsed_time(e1):7.2f} ms for 512 stages", flush=True)
