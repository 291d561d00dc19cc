"""GPU parity: HIP segmentation kernels (through the C ABI) vs the torch-CPU oracle.
Tolerances: SURVEY.md section 8d -- log-probs rtol 1e-4 / atol 1e-5 class (fp32 re-association)."""
import ctypes as C
import os

import pytest
import torch
import torch.nn.functional as F

from conftest import north_star_ratio, report

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def seg(gpu_device):
    from oracle import seeded_pyannet
    from pyannote_audio_amd.weights import SegmentationPack
    from pyannote_audio_amd.segmentation import SegmentationEngine
    model = seeded_pyannet(seed=1234, num_layers=4)
    pack = SegmentationPack(model.state_dict(), {"lstm": {"num_layers": 4}}, 7, 3, 2, gpu_device)
    return model, pack, SegmentationEngine(pack)


def _wave(B, N, seed=0):
    g = torch.Generator().manual_seed(seed)
    x = 0.1 * torch.randn(B, 1, N, generator=g)
    x += 0.05 * torch.sin(torch.arange(N) * 0.01)[None, None] + 0.02  # DC + tone
    return x.clamp(-1, 1)


def test_sinc_taps_match_oracle(seg):
    model, pack, _ = seg
    taps = model.sincnet.conv1d[0].filterbank.filters()[:, 0]
    assert torch.equal(taps, pack.sinc_taps)


def test_frontend_stages(seg, gpu_device):
    import pyannote_audio_amd.ffi as ffi
    lib = ffi.load()
    model, pack, _ = seg
    sn = model.sincnet
    B, N = 3, 80000
    x = _wave(B, N)
    with torch.inference_mode():
        xn = sn.wav_norm1d(x)
        c1 = torch.abs(sn.conv1d[0](xn))
        p1 = sn.pool1d[0](c1)
        a1 = F.leaky_relu(sn.norm1d[0](p1))
        p2 = sn.pool1d[1](sn.conv1d[1](a1))
        a2 = F.leaky_relu(sn.norm1d[1](p2))
        p3 = sn.pool1d[2](sn.conv1d[2](a2))
        a3 = F.leaky_relu(sn.norm1d[2](p3))
    dev = gpu_device
    w = pack.struct
    xd = x.view(-1).to(dev)
    st = ffi.stream()
    mean = torch.empty(B, device=dev); rstd = torch.empty(B, device=dev)
    ffi.check(lib.pa_row_stats(ffi.ptr(xd), N, xd.numel(), B, N, 1e-5, ffi.ptr(mean), ffi.ptr(rstd), st))
    report("wav_mean", mean, x.mean(-1).view(-1))
    report("wav_rstd", rstd, 1.0 / torch.sqrt(x.var(-1, unbiased=False) + 1e-5).view(-1))
    P1 = p1.shape[-1]
    s1 = torch.zeros(B, 80, P1, device=dev)
    ffi.check(lib.pa_sinc_fir_pool(ffi.ptr(xd), xd.numel(), N, B, N, 10, ffi.ptr(mean), ffi.ptr(rstd),
                                   w.wav_gamma, w.wav_beta, w.sinc_filt, ffi.ptr(s1), st))
    assert north_star_ratio("sinc_pool", s1, p1) <= 1.0       # element-wise rtol 1e-4 / atol 1e-5
    m1 = torch.empty(B * 80, device=dev); r1 = torch.empty(B * 80, device=dev)
    ffi.check(lib.pa_row_stats(ffi.ptr(s1), P1, s1.numel(), B * 80, P1, 1e-5, ffi.ptr(m1), ffi.ptr(r1), st))
    P2 = p2.shape[-1]
    s2 = torch.zeros(B, 60, P2, device=dev)
    ffi.check(lib.pa_conv5_pool(ffi.ptr(s1), B, 80, P1, ffi.ptr(m1), ffi.ptr(r1), w.norm0,
                                C.c_void_p(w.norm0 + 80 * 4), w.conv1_w, w.conv1_b, ffi.ptr(s2), st))
    assert north_star_ratio("conv1_pool", s2, p2) <= 1.0
    m2 = torch.empty(B * 60, device=dev); r2 = torch.empty(B * 60, device=dev)
    ffi.check(lib.pa_row_stats(ffi.ptr(s2), P2, s2.numel(), B * 60, P2, 1e-5, ffi.ptr(m2), ffi.ptr(r2), st))
    T = p3.shape[-1]
    s3 = torch.zeros(B, 60, T, device=dev)
    ffi.check(lib.pa_conv5_pool(ffi.ptr(s2), B, 60, P2, ffi.ptr(m2), ffi.ptr(r2), w.norm1,
                                C.c_void_p(w.norm1 + 60 * 4), w.conv2_w, w.conv2_b, ffi.ptr(s3), st))
    assert north_star_ratio("conv2_pool", s3, p3) <= 1.0
    m3 = torch.empty(B * 60, device=dev); r3 = torch.empty(B * 60, device=dev)
    ffi.check(lib.pa_row_stats(ffi.ptr(s3), T, s3.numel(), B * 60, T, 1e-5, ffi.ptr(m3), ffi.ptr(r3), st))
    X0 = torch.full((1 * T * 16, 64), float("nan"), device=dev)
    ffi.check(lib.pa_norm_transpose(ffi.ptr(s3), B, T, ffi.ptr(m3), ffi.ptr(r3), w.norm2,
                                    C.c_void_p(w.norm2 + 60 * 4), ffi.ptr(X0), st))
    torch.cuda.synchronize()
    X0 = X0.view(T, 16, 64).cpu()
    assert north_star_ratio("sincnet_out", X0[:, :B, :60].permute(1, 2, 0), a3) <= 1.0
    assert torch.all(X0[:, B:] == 0) and torch.all(X0[:, :, 60:] == 0)


def test_gemm_tn(gpu_device):
    import pyannote_audio_amd.ffi as ffi
    lib = ffi.load()
    g = torch.Generator().manual_seed(3)
    for (M, N, K, act, mode) in [(300, 128, 256, 1, 0), (160, 1024, 64, 0, 1), (37, 256, 5120, 0, 0)]:
        A = torch.randn(M, K, generator=g); W = torch.randn(N, K, generator=g) / K ** 0.5
        b = torch.randn(N, generator=g)
        ref = (A.double() @ W.double().T + b.double()).float()
        if act:
            ref = F.leaky_relu(ref)
        Ad, Wd, bd = A.to(gpu_device), W.to(gpu_device), b.to(gpu_device)
        Cd = torch.full((M * N,), float("nan"), device=gpu_device)
        ffi.check(lib.pa_gemm_tn(ffi.ptr(Ad), K, ffi.ptr(Wd), K, ffi.ptr(bd), ffi.ptr(Cd), N, M, N, K, act,
                                 mode, ffi.stream()), "gemm")
        torch.cuda.synchronize()
        got = Cd.cpu()
        if mode == 1:
            got = got.view(M // 16, N, 16).permute(0, 2, 1).reshape(M, N)
        else:
            got = got.view(M, N)
        if K <= 256:      # element-wise contract against the float64 product
            assert north_star_ratio(f"gemm_{M}x{N}x{K}", got, ref) <= 1.0
        else:             # K = 5 120 random terms: float32 summation itself is only good to ~1e-5 absolute here
            e = report(f"gemm_{M}x{N}x{K}", got, ref)
            assert e < 1e-4 * ref.abs().max().item()


def test_lstm_layer(seg, gpu_device):
    """one bidirectional layer (layer 0 weights) on random input vs torch.nn.LSTM."""
    import pyannote_audio_amd.ffi as ffi
    lib = ffi.load()
    model, pack, _ = seg
    w = pack.struct
    B, T = 20, 37
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, T, 60, generator=g)
    l0 = torch.nn.LSTM(60, 128, 1, bidirectional=True, batch_first=True)
    sd = model.lstm.state_dict()
    l0.load_state_dict({k: v for k, v in sd.items() if k.endswith("_l0") or k.endswith("_l0_reverse")})
    with torch.inference_mode():
        ref, _ = l0(x)
    ntiles = (B + 15) // 16
    X0 = torch.zeros(ntiles, T, 16, 64)
    for b in range(B):
        X0[b // 16, :, b % 16, :60] = x[b]
    X0 = X0.to(gpu_device)
    M = ntiles * T * 16
    xproj = torch.empty(M * 1024, device=gpu_device)
    out = torch.full((M, 256), float("nan"), device=gpu_device)
    st = ffi.stream()
    ffi.check(lib.pa_gemm_tn(ffi.ptr(X0), 64, C.c_void_p(w.lstm_wih[0]), 64, C.c_void_p(w.lstm_bias[0]),
                             ffi.ptr(xproj), 0, M, 1024, 64, 0, 1, st), "xproj")
    ffi.check(lib.pa_lstm_rec(ffi.ptr(xproj), C.c_void_p(w.lstm_whh[0]), ffi.ptr(out), ntiles, 2, T, st),
              "lstm")
    torch.cuda.synchronize()
    got = out.view(ntiles, T, 16, 256).permute(0, 2, 1, 3).reshape(ntiles * 16, T, 256)[:B].cpu()
    assert north_star_ratio("lstm_layer0", got, ref) <= 1.0


@pytest.mark.parametrize("B,N,stride", [(5, 80000, 8000), (18, 160000, 16000)])
def test_seg_forward_end_to_end(seg, gpu_device, B, N, stride):
    model, pack, eng = seg
    total = stride * (B - 1) + N - 5000  # last chunk is zero padded by 5000 samples
    wav = _wave(1, total, seed=7).view(-1)
    chunks = torch.zeros(B, 1, N)
    for b in range(B):
        seg_ = wav[b * stride: b * stride + N]
        chunks[b, 0, :seg_.numel()] = seg_
    with torch.inference_mode():
        ref = model(chunks)
    logp, ml = eng.forward_strided(wav.to(gpu_device), stride, B, N)
    torch.cuda.synchronize()
    # north_star tolerance for segmentation log-probs: rtol 1e-4, atol 1e-5
    assert north_star_ratio(f"seg_logp_B{B}_N{N}", logp, ref) <= 1.0
    # hard powerset decisions identical except where the top-2 gap is below tolerance
    top2 = ref.topk(2, dim=-1).values
    safe = (top2[..., 0] - top2[..., 1]) > 1e-4
    from oracle import Powerset
    ref_ml = Powerset(3, 2)(ref).to(torch.uint8)
    mism = (ml.cpu() != ref_ml).any(dim=-1)
    with open("gpurun_out/parity.log", "a") as fp:
        fp.write(f"seg_hard_B{B}_N{N}: hard-decision mismatches {int(mism.sum())} of {mism.numel()} frames, "
                 f"{int((mism & safe).sum())} outside the 1e-4 top-2 gap\n")
    assert torch.equal(ml.cpu()[safe], ref_ml[safe])
    # reference Model.forward contract: (B,1,N) in -> (B,F,K) out
    out = eng.forward(chunks[:2].to(gpu_device))
    assert north_star_ratio(f"seg_forward_contract_N{N}", out, ref[:2]) <= 1.0


def test_shared_sinc_layer_matches_the_per_chunk_layer(seg, gpu_device):
    """The sinc layer computed ONCE for a span of overlapping chunks + per-chunk affine fix-up (pa_sinc_fir_span,
    pa_sinc_fix_pool; the default of pa_seg_forward) against the per-chunk layer (PA_SEG_SHARED_SINC=0): the pooled magnitudes
    within the element-wise tolerance of the oracle, and the whole forward within it too (the CPU study
    tools/probes/shared_sinc_numerics.py: log-probabilities move by 2.5e-5, no hard decision changes)."""
    import pyannote_audio_amd.ffi as ffi
    from pyannote_audio_amd.segmentation import SegmentationEngine
    lib = ffi.load()
    model, pack, _ = seg
    sn = model.sincnet
    w = pack.struct
    dev = gpu_device
    N, STEP, B = 160000, 16000, 7
    x = _wave(1, (B - 1) * STEP + N - 4000)[0, 0]          # the last chunk is zero padded
    chunks = torch.stack([F.pad(x[c * STEP: c * STEP + N], (0, max(0, c * STEP + N - x.numel()))) for c in range(B)])
    with torch.inference_mode():
        p1 = sn.pool1d[0](torch.abs(sn.conv1d[0](sn.wav_norm1d(chunks.unsqueeze(1)))))
    xd = x.contiguous().to(dev)
    st = ffi.stream()
    mean = torch.empty(B, device=dev); rstd = torch.empty(B, device=dev)
    ffi.check(lib.pa_row_stats(ffi.ptr(xd), STEP, xd.numel(), B, N, 1e-5, ffi.ptr(mean), ffi.ptr(rstd), st))
    span = (B - 1) * STEP + N
    Pc = (span - 251) // 10 + 1
    S = torch.empty(80, Pc, device=dev)
    taps = torch.empty(80, device=dev)
    P1 = p1.shape[-1]
    s1 = torch.zeros(B, 80, P1, device=dev)
    ffi.check(lib.pa_sinc_fir_span(ffi.ptr(xd), xd.numel(), span, w.sinc_filt, ffi.ptr(S), st))
    ffi.check(lib.pa_sinc_fix_pool(ffi.ptr(S), Pc, STEP // 10, B, P1, ffi.ptr(mean), ffi.ptr(rstd), w.wav_gamma,
                                   w.wav_beta, w.sinc_filt, ffi.ptr(taps), ffi.ptr(s1), st))
    assert north_star_ratio("shared_sinc_pool", s1, p1) <= 1.0
    engine = SegmentationEngine(pack)
    outs = {}
    for flag in ("0", "1"):
        os.environ["PA_SEG_SHARED_SINC"] = flag
        try:
            logp, ml = engine.forward_strided(xd, STEP, B, N)
            outs[flag] = (logp.cpu(), ml.cpu())
        finally:
            os.environ.pop("PA_SEG_SHARED_SINC", None)
    assert north_star_ratio("shared_sinc_logp", outs["1"][0], outs["0"][0]) <= 1.0
    top2 = outs["0"][0].topk(2, dim=-1).values
    decided = (top2[..., 0] - top2[..., 1]) > 1e-4
    same = (outs["1"][1] == outs["0"][1]).all(dim=-1)
    assert bool((same | ~decided).all())


@pytest.mark.parametrize("lstm,linear", [
    ({"hidden_size": 64, "num_layers": 2}, {"hidden_size": 64, "num_layers": 2}),
    ({"hidden_size": 256, "num_layers": 2}, {"hidden_size": 256, "num_layers": 1}),
    ({"hidden_size": 96, "num_layers": 1}, {"hidden_size": 128, "num_layers": 0}),
    ({"hidden_size": 64, "num_layers": 2, "bidirectional": False, "monolithic": False}, {"hidden_size": 96, "num_layers": 2}),
    ({"hidden_size": 128, "num_layers": 1, "bidirectional": False}, {"hidden_size": 32, "num_layers": 1}),
])
def test_other_lstm_and_linear_widths_match_oracle(gpu_device, lstm, linear):
    """PyanNet.py:64-72, 98-123 accept any nn.LSTM / Linear configuration: hidden sizes
    other than 128, a single direction, split (non-monolithic) stacks, heads of other widths or none at all run
    through k_lstm_rec_gen (W_hh streamed from L2, two 16-chunk tiles per workgroup; csrc/seg_lstm.hip) and the
    same GEMM / classifier kernels -- log-probabilities against the torch-CPU oracle, 37 chunks (an odd number of
    tiles: the last workgroup owns a single one)."""
    from oracle.models import PyanNet
    from pyannote_audio_amd.weights import SegmentationPack
    from pyannote_audio_amd.segmentation import SegmentationEngine
    torch.manual_seed(77)
    model = PyanNet(num_classes=7, lstm=lstm, linear=linear).eval()
    pack = SegmentationPack(model.state_dict(), {"lstm": lstm, "linear": linear}, 7, 3, 2, gpu_device)
    assert pack.struct.lstm_hidden == lstm["hidden_size"]
    eng = SegmentationEngine(pack)
    B, N, stride = 37, 32000, 4000
    wav = _wave(1, stride * (B - 1) + N, seed=11).view(-1)
    chunks = torch.stack([wav[b * stride: b * stride + N] for b in range(B)]).unsqueeze(1)
    with torch.inference_mode():
        ref = model(chunks)
    logp, ml = eng.forward_strided(wav.to(gpu_device), stride, B, N)
    torch.cuda.synchronize()
    assert logp.shape == ref.shape
    tag = "seg_H%d_%s_lin%dx%d" % (lstm["hidden_size"], "uni" if lstm.get("bidirectional") is False else "bi",
                                   linear["hidden_size"], linear["num_layers"])
    assert north_star_ratio(tag, logp, ref) <= 1.0
    top2 = ref.topk(2, dim=-1).values
    safe = (top2[..., 0] - top2[..., 1]) > 1e-4
    assert torch.equal(logp.cpu().argmax(-1)[safe], ref.argmax(-1)[safe])


def test_lstm_recurrence_any_width_through_the_c_abi(gpu_device):
    """pa_lstm_rec_h directly: one bidirectional layer of H = 48 / 192 / 512 and a unidirectional one of H = 32 on
    random gate inputs vs torch.nn.LSTM (W_ih = identity block: the test feeds the gate pre-activations itself)."""
    import pyannote_audio_amd.ffi as ffi
    from pyannote_audio_amd.weights import _lstm_row_perm_gen, _lstm_whh_image_gen
    lib = ffi.load()
    g = torch.Generator().manual_seed(9)
    for H, ndir, B, T in ((48, 2, 20, 23), (192, 2, 33, 11), (512, 2, 16, 5), (32, 1, 50, 40)):
        K = 32
        ref_lstm = torch.nn.LSTM(K, H, 1, bidirectional=ndir == 2, batch_first=True)
        x = torch.randn(B, T, K, generator=g)
        with torch.inference_mode():
            ref, _ = ref_lstm(x)
        sd = ref_lstm.state_dict()
        perm = _lstm_row_perm_gen(H)
        ntiles = (B + 15) // 16
        # gate pre-activations x W_ih^T + b_ih + b_hh in the kernel's layout [tile][t][ndir * 4H][16]
        xproj = torch.zeros(ntiles, T, ndir * 4 * H, 16)
        whh = []
        for d in range(ndir):
            sfx = "_reverse" if d else ""
            pre = x @ sd["weight_ih_l0" + sfx].T + sd["bias_ih_l0" + sfx] + sd["bias_hh_l0" + sfx]   # (B, T, 4H)
            pre = pre[..., perm]
            for b in range(B):
                xproj[b // 16, :, d * 4 * H:(d + 1) * 4 * H, b % 16] = pre[b]
            whh.append(_lstm_whh_image_gen(sd["weight_hh_l0" + sfx]))
        xd, wd = xproj.to(gpu_device), torch.cat(whh).to(gpu_device)
        out = torch.full((ntiles * T * 16, ndir * H), float("nan"), device=gpu_device)
        ffi.check(lib.pa_lstm_rec_h(ffi.ptr(xd), ffi.ptr(wd), ffi.ptr(out), ntiles, ndir, T, H, ffi.stream()),
                  "pa_lstm_rec_h")
        torch.cuda.synchronize()
        got = out.view(ntiles, T, 16, ndir * H).permute(0, 2, 1, 3).reshape(ntiles * 16, T, ndir * H)[:B].cpu()
        assert north_star_ratio(f"lstm_rec_h_H{H}_d{ndir}", got, ref) <= 1.0


@pytest.mark.parametrize("stride", [5, 16])
def test_other_sincnet_strides_match_oracle(gpu_device, stride):
    """SincNet(stride=...) (models/blocks/sincnet.py:58-69; PyanNet hyper-parameter `sincnet.stride`): the sinc kernel is
    instantiated per stride (the per-chunk form; the shared per-span sinc layer stays with stride 10) -- log-probs against
    the oracle, frame counts from the same geometry."""
    from oracle.models import PyanNet
    from pyannote_audio_amd.weights import SegmentationPack
    from pyannote_audio_amd.segmentation import SegmentationEngine
    torch.manual_seed(5)
    hp = {"sincnet": {"stride": stride}, "lstm": {"hidden_size": 128, "num_layers": 2}}
    model = PyanNet(num_classes=7, sincnet=hp["sincnet"], lstm=hp["lstm"]).eval()
    pack = SegmentationPack(model.state_dict(), hp, 7, 3, 2, gpu_device)
    assert pack.struct.sinc_stride == stride
    eng = SegmentationEngine(pack)
    B, N, step = 19, 48000, 6000
    wav = _wave(1, step * (B - 1) + N, seed=23).view(-1)
    chunks = torch.stack([wav[b * step: b * step + N] for b in range(B)]).unsqueeze(1)
    with torch.inference_mode():
        ref = model(chunks)
    logp, ml = eng.forward_strided(wav.to(gpu_device), step, B, N)
    torch.cuda.synchronize()
    assert logp.shape == ref.shape == (B, model.num_frames(N), 7)
    assert north_star_ratio(f"seg_sinc_stride_{stride}", logp, ref) <= 1.0
