"""GPU parity: HIP embedding kernels (through the C ABI) vs the torch-CPU oracle.
Tolerances: embeddings rtol 1e-4 / atol 1e-5 class; fbank compared on the linear power
scale relative to the frame's peak (fp32 FFT round-off differs between pocketfft and our radix-4)."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

from conftest import north_star_ratio, report

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def emb(gpu_device):
    from oracle import seeded_wespeaker
    from pyannote_audio_amd.weights import EmbeddingPack
    from pyannote_audio_amd.embedding import EmbeddingEngine
    model = seeded_wespeaker(seed=4321)
    pack = EmbeddingPack(model.state_dict(), gpu_device)
    return model, pack, EmbeddingEngine(pack, max_chunks=3)


def _wave(B, N, seed=0):
    g = torch.Generator().manual_seed(seed)
    x = 0.1 * torch.randn(B, 1, N, generator=g)
    x += 0.05 * torch.sin(torch.arange(N) * 0.05)[None, None] + 0.01
    return x.clamp(-1, 1)


def test_mel_tables_match_oracle(emb):
    from oracle.models import kaldi_mel_banks
    _, pack, _ = emb
    assert torch.equal(pack.mel[:, :256], kaldi_mel_banks(80, 512, 16000.0))


def test_fbank(emb, gpu_device):
    import pyannote_audio_amd.ffi as ffi
    lib = ffi.load()
    model, pack, _ = emb
    w = pack.struct
    B, N = 3, 48000
    x = _wave(B, N)
    with torch.inference_mode():
        ref = model.compute_fbank(x)
    xd = x.view(-1).to(gpu_device)
    T = lib.pa_emb_num_fbank_frames(N)
    out = torch.full((B, T, 80), float("nan"), device=gpu_device)
    ffi.check(lib.pa_fbank(ffi.ptr(xd), xd.numel(), N, B, N, w.fb_window, w.fb_tw256, w.fb_tw512,
                           w.fb_mel_w, w.fb_mel_lo, w.fb_mel_hi, 80, ffi.ptr(out), 1, ffi.stream()), "fbank")
    torch.cuda.synchronize()
    assert out.shape == ref.shape
    e = report("fbank_centered", out, ref)
    # (a) energy domain, relative to max(value, 1e-3 x peak): the float tolerance of the contract
    # (b) log domain: bins 1e-4 of the peak carry the FFT round-off of the whole frame in BOTH
    #     float32 implementations (the float32 oracle itself is 5e-3 away from a float64 evaluation
    #     there, tests/test_oracle_fbank_pin.py), so the log-domain bound is loose by construction
    e_ref = torch.exp(ref.double())          # centring is a common factor in the energy domain
    e_got = torch.exp(out.cpu().double())
    rel = ((e_got - e_ref).abs() / torch.maximum(e_ref, 1e-3 * e_ref.amax(dim=(1, 2), keepdim=True))).max().item()
    with open("gpurun_out/parity.log", "a") as fp:
        fp.write(f"fbank_centered: energy-domain rel err = {rel:.3e}\n")
    assert rel < 2e-4   # measured 9.7e-5 (both sides are float32 512-point FFTs)
    assert e < 2e-3


@pytest.mark.parametrize("T,nmel,K", [(298, 80, 39), (998, 80, 299), (298, 80, 1), (248, 80, 2999), (77, 23, 5),
                                      (1201, 80, 301)])
def test_fbank_center_span_kernel(gpu_device, T, nmel, K):
    """pa_fbank_center_span == x - F.avg_pool1d(x, K, 1, K // 2, count_include_pad=False) (wespeaker/__init__.py:
    151-157): frames near both ends (windows clipped to the chunk), windows longer than the chunk, several frame
    tiles and mel groups.  Same float32 sums in the same order: the bound is a few ulp of the features."""
    import pyannote_audio_amd.ffi as ffi
    lib = ffi.load()
    g = torch.Generator().manual_seed(T + K)
    x = 4.0 * torch.randn(3, T, nmel, generator=g) - 9.0
    want = x - F.avg_pool1d(x.transpose(1, 2), kernel_size=K, stride=1, padding=K // 2,
                            count_include_pad=False).transpose(1, 2)
    xd = x.to(gpu_device)
    out = torch.full_like(xd, float("nan"))
    ffi.check(lib.pa_fbank_center_span(ffi.ptr(xd), 3, T, nmel, K, ffi.ptr(out), ffi.stream()), "center_span")
    torch.cuda.synchronize()
    assert torch.equal(xd.cpu(), x)                        # out of place
    err = (out.cpu() - want).abs().max().item()
    report(f"fbank_center_span_T{T}_K{K}", out, want)
    assert err <= 4e-6, err                                # |x| ~ 20: one ulp is 1.9e-6
    # refusals: even windows, in place
    assert lib.pa_fbank_center_span(ffi.ptr(xd), 3, T, nmel, 4, ffi.ptr(out), ffi.stream()) != 0
    assert lib.pa_fbank_center_span(ffi.ptr(xd), 3, T, nmel, K, ffi.ptr(xd), ffi.stream()) != 0


@pytest.mark.parametrize("span", [0.4, 3.0])
def test_emb_forward_with_centering_span(gpu_device, span, tmp_path):
    """a checkpoint of a user class that forwards `fbank_centering_span` (wespeaker/__init__.py:137-157), loaded through
    the registered counterpart: embeddings against the oracle with the same span, north-star tolerance"""
    import pyannote_audio_amd as pa
    from pyannote_audio_amd import model as pm
    from oracle import seeded_wespeaker
    from conftest import WESPEAKER_HPARAMS

    @pm.register_architecture
    class CentredResNet34(pm.WeSpeakerResNet34):
        ARCHITECTURE = ("my_project.models", "CentredResNet34")
        INIT_KEYS = pm.WeSpeakerResNet34.INIT_KEYS + ("fbank_centering_span",)

    try:
        oracle_model = seeded_wespeaker(seed=4321)
        oracle_model.fbank_centering_span = span
        path = str(tmp_path / "centred.bin")
        pm.save_checkpoint(path, oracle_model.state_dict(), dict(WESPEAKER_HPARAMS, fbank_centering_span=span),
                           CentredResNet34.ARCHITECTURE, pm.embedding_specifications())
        model = pa.Model.from_pretrained(path).to(gpu_device)
        assert model.fbank_center_kernel == (39 if span == 0.4 else 299)
        x = _wave(3, 48000, seed=7)
        g = torch.Generator().manual_seed(4)
        masks = (torch.rand(3, 3, 173, generator=g) < 0.7).float()
        with torch.inference_mode():
            ref = oracle_model(x, weights=masks)
            plain = seeded_wespeaker(seed=4321)(x, weights=masks)
        out = model(x.to(gpu_device), masks.to(gpu_device))
        torch.cuda.synchronize()
        assert north_star_ratio(f"emb_center_span_{span}", out, ref) <= 1.0
        assert north_star_ratio(f"emb_center_span_{span}_vs_global_mean", out, plain, ) > 1.0   # (it is not a no-op)
    finally:
        pm._USER_ARCHITECTURES.pop(CentredResNet34.ARCHITECTURE, None)


def test_conv3x3_configs(gpu_device):
    import pyannote_audio_amd.ffi as ffi
    lib = ffi.load()
    g = torch.Generator().manual_seed(11)
    # (cin, cout, H, W, stride): one per kernel instantiation + ragged edges
    # the kernel is persistent (grid = resident workgroups, each loops over tiles with a cross-tile
    # prefetch): the B = 48 / 150 cases give every workgroup several tiles, also without a residual
    cases = [(32, 32, 80, 70, 1, 2, True), (64, 64, 40, 45, 1, 2, True), (128, 128, 20, 37, 1, 2, True),
             (256, 256, 10, 70, 1, 2, True), (32, 64, 80, 71, 2, 2, True), (128, 256, 20, 67, 2, 2, True),
             (64, 128, 40, 50, 2, 2, True), (32, 32, 80, 70, 1, 48, True), (32, 64, 80, 71, 2, 48, False),
             (256, 256, 10, 70, 1, 150, False), (64, 64, 40, 45, 1, 40, True)]
    for cin, cout, H, W, s, B, use_res in cases:
        x = torch.randn(B, cin, H, W, generator=g)
        wt = torch.randn(cout, cin, 3, 3, generator=g) / (3 * cin ** 0.5)
        sh = torch.randn(cout, generator=g)
        Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
        res = torch.randn(B, cout, Ho, Wo, generator=g)
        ref = F.conv2d(x, wt, stride=s, padding=1) + sh.view(1, -1, 1, 1)
        ref = F.relu(ref + res if use_res else ref)
        xd = x.permute(0, 2, 3, 1).contiguous().to(gpu_device)
        wd = wt.permute(2, 3, 0, 1).reshape(9, cout, cin).contiguous().to(gpu_device)
        rd = res.permute(0, 2, 3, 1).contiguous().to(gpu_device)
        shd = sh.to(gpu_device)
        y = torch.full((B, Ho, Wo, cout), float("nan"), device=gpu_device)
        ffi.check(lib.pa_conv3x3(ffi.ptr(xd), B, H, W, cin, ffi.ptr(wd), ffi.ptr(shd),
                                 ffi.ptr(rd) if use_res else None, ffi.ptr(y), cout, s, 1, ffi.stream()),
                  "conv3x3")
        torch.cuda.synchronize()
        assert not torch.isnan(y).any()
        # element-wise north_star contract (rtol 1e-4 / atol 1e-5), also for this intermediate activation
        assert north_star_ratio(f"conv3x3_{cin}_{cout}_{H}x{W}_s{s}_B{B}", y.permute(0, 3, 1, 2), ref) <= 1.0


def test_stats_pool_kats(gpu_device):
    """the reference's own known-answer tests (tests/test_stats_pool.py:28-131) on the HIP kernel."""
    import pyannote_audio_amd.ffi as ffi
    lib = ffi.load()

    def run(x, w):
        # x: (B, D, T) -> feat[b][f=1][t][c=D]
        B, D, T = x.shape
        feat = x.permute(0, 2, 1).contiguous().view(B, 1, T, D).to(gpu_device)
        if w is None:
            S, Fm, md, idx = 1, 0, None, None
        else:
            w3 = w if w.dim() == 3 else w.unsqueeze(1)
            S, Fm = w3.shape[1], w3.shape[2]
            md = w3.contiguous().to(gpu_device)
            ramp = torch.arange(Fm, dtype=torch.float32).view(1, 1, -1)
            idx = F.interpolate(ramp, size=T, mode="nearest").view(-1).to(torch.int32).to(gpu_device)
        out = torch.empty(B, S, 2 * D, device=gpu_device)
        ffi.check(lib.pa_stats_pool(ffi.ptr(feat), B, 1, T, D, ffi.ptr(md), S, Fm, ffi.ptr(idx),
                                    ffi.ptr(out), ffi.stream()), "pool")
        torch.cuda.synchronize()
        out = out.cpu()
        return out[:, 0] if (w is None or w.dim() == 2) else out

    x = torch.Tensor([[[2.0, 4.0], [2.0, 4.0]], [[1.0, 1.0], [1.0, 1.0]]])
    r = lambda y: torch.round(y, decimals=4)
    assert torch.equal(r(run(x, None)), torch.Tensor([[3.0, 3.0, 1.4142, 1.4142], [1.0, 1.0, 0.0, 0.0]]))
    w = torch.Tensor([[0.5, 0.01], [0.2, 0.1]])
    assert torch.equal(r(run(x, w)), torch.Tensor([[2.0392, 2.0392, 1.4142, 1.4142], [1.0, 1.0, 0.0, 0.0]]))
    w = torch.Tensor([[[0.1, 0.2], [0.2, 0.3]], [[0.001, 0.001], [0.2, 0.3]]])
    assert torch.equal(r(run(x, w)), torch.Tensor(
        [[[3.3333, 3.3333, 1.4142, 1.4142], [3.2, 3.2, 1.4142, 1.4142]],
         [[1.0, 1.0, 0.0, 0.0], [1.0, 1.0, 0.0, 0.0]]]))
    x2 = torch.Tensor([[[2.0, 2.0], [2.0, 2.0]], [[1.0, 1.0], [1.0, 1.0]]])
    w = torch.Tensor([[0.5, 0.5, 0.0], [0.0, 0.5, 0.5]])
    assert torch.equal(r(run(x2, w)), torch.Tensor([[2.0, 2.0, 0.0, 0.0], [1.0, 1.0, 0.0, 0.0]]))
    w = torch.Tensor([[0.5, 0.01], [0.0, 0.0]])
    assert torch.equal(r(run(x, w)), torch.Tensor([[2.0392, 2.0392, 1.4142, 1.4142], [0.0, 0.0, 0.0, 0.0]]))


@pytest.mark.parametrize("B,N", [(4, 48000), (2, 160000)])
def test_emb_forward_end_to_end(emb, gpu_device, B, N):
    model, pack, eng = emb
    x = _wave(B, N, seed=5)
    g = torch.Generator().manual_seed(2)
    Fm = 589 if N == 160000 else 173
    masks = (torch.rand(B, 3, Fm, generator=g) < 0.7).float()
    masks[0, 2] = 0.0  # all-zero mask -> embedding = seg_1 bias path (tests/test_stats_pool.py:111-131)
    with torch.inference_mode():
        ref = model(x, weights=masks)
        ref1 = model(x[:2])
    out = eng.forward(x.to(gpu_device), masks.to(gpu_device))
    out1 = eng.forward(x[:2].to(gpu_device))
    torch.cuda.synchronize()
    # north_star tolerance for embeddings: rtol 1e-4, atol 1e-5
    assert north_star_ratio(f"emb_B{B}_N{N}", out, ref) <= 1.0
    assert north_star_ratio(f"emb_unweighted_N{N}", out1, ref1) <= 1.0


def test_conv3x3_winograd(gpu_device):
    """pa_conv3x3_wino (Winograd F(2x2,3x3), fp32) vs torch conv2d: all ResNet34 channel configurations,
    odd / ragged extents, with and without residual, enough images for several tiles per workgroup."""
    import pyannote_audio_amd.ffi as ffi
    from pyannote_audio_amd.weights import winograd_pack, winograd_weights
    lib = ffi.load()
    g = torch.Generator().manual_seed(12)
    cases = [(32, 32, 80, 70, 2, True), (64, 64, 40, 45, 2, True), (128, 128, 20, 37, 2, False),
             (256, 256, 10, 71, 2, True), (32, 32, 17, 9, 3, True), (64, 64, 1, 1, 2, False),
             (32, 32, 80, 70, 48, True), (256, 256, 10, 125, 40, False), (128, 128, 20, 250, 20, True)]
    for cin, cout, H, W, B, use_res in cases:
        x = torch.randn(B, cin, H, W, generator=g)
        wt = torch.randn(cout, cin, 3, 3, generator=g) / (3 * cin ** 0.5)
        sh = torch.randn(cout, generator=g)
        res = torch.randn(B, cout, H, W, generator=g)
        ref = F.conv2d(x, wt, stride=1, padding=1) + sh.view(1, -1, 1, 1)
        ref = F.relu(ref + res if use_res else ref)
        xd = x.permute(0, 2, 3, 1).contiguous().to(gpu_device)
        ud = winograd_pack(winograd_weights(wt)).to(gpu_device)
        rd = res.permute(0, 2, 3, 1).contiguous().to(gpu_device)
        shd = sh.to(gpu_device)
        y = torch.full((B, H, W, cout), float("nan"), device=gpu_device)
        ffi.check(lib.pa_conv3x3_wino(ffi.ptr(xd), B, H, W, cin, ffi.ptr(ud), ffi.ptr(shd),
                                      ffi.ptr(rd) if use_res else None, ffi.ptr(y), cout, 1, ffi.stream()),
                  "conv3x3_wino")
        torch.cuda.synchronize()
        assert not torch.isnan(y).any(), (cin, cout, H, W, B)
        assert north_star_ratio(f"wino3x3_{cin}_{cout}_{H}x{W}_B{B}", y.permute(0, 3, 1, 2), ref) <= 1.0


def test_conv3x3_winograd_f4(gpu_device):
    """pa_conv3x3_wino4 (Winograd F(4x4,3x3), fp32) vs torch conv2d: all ResNet34 channel configurations, odd /
    ragged extents (tiles, tile rows and whole waves outside the image), with and without residual, enough images
    for several tiles per workgroup.  Bound per convolution: |err| <= 1e-4 max |ref| (F(4x4)'s transforms cost a
    digit against F(2x2): 2-6e-5 of the maximum on unit-variance data, tools/probes/winograd_f4_numerics.py; the
    element-wise north-star bound is asserted on the EMBEDDINGS, test_emb_forward_* / test_golden.py)."""
    import pyannote_audio_amd.ffi as ffi
    from pyannote_audio_amd.weights import winograd4_pack, winograd4_weights
    lib = ffi.load()
    g = torch.Generator().manual_seed(12)
    cases = [(32, 32, 80, 70, 2, True), (64, 64, 40, 45, 2, True), (128, 128, 20, 37, 2, False),
             (256, 256, 10, 71, 2, True), (32, 32, 17, 9, 3, True), (64, 64, 1, 1, 2, False),
             (32, 32, 80, 270, 24, True), (256, 256, 10, 125, 40, False), (128, 128, 20, 250, 20, True),
             (40, 32, 9, 129, 3, False)]
    for cin, cout, H, W, B, use_res in cases:
        x = torch.randn(B, cin, H, W, generator=g)
        wt = torch.randn(cout, cin, 3, 3, generator=g) / (3 * cin ** 0.5)
        sh = torch.randn(cout, generator=g)
        res = torch.randn(B, cout, H, W, generator=g)
        ref = F.conv2d(x, wt, stride=1, padding=1) + sh.view(1, -1, 1, 1)
        ref = F.relu(ref + res if use_res else ref)
        xd = x.permute(0, 2, 3, 1).contiguous().to(gpu_device)
        ud = winograd4_pack(winograd4_weights(wt)).to(gpu_device)
        rd = res.permute(0, 2, 3, 1).contiguous().to(gpu_device)
        shd = sh.to(gpu_device)
        y = torch.full((B, H, W, cout), float("nan"), device=gpu_device)
        ffi.check(lib.pa_conv3x3_wino4(ffi.ptr(xd), B, H, W, cin, ffi.ptr(ud), ffi.ptr(shd),
                                       ffi.ptr(rd) if use_res else None, ffi.ptr(y), cout, 1, ffi.stream()),
                  "conv3x3_wino4")
        torch.cuda.synchronize()
        got = y.permute(0, 3, 1, 2).cpu()
        assert not torch.isnan(got).any(), (cin, cout, H, W, B)
        err = (got - ref).abs().max().item() / ref.abs().max().item()
        elementwise = north_star_ratio(f"wino4_3x3_{cin}_{cout}_{H}x{W}_B{B}", got, ref)   # logged, not the bound here
        print(f"wino4 {cin}->{cout} {H}x{W} B{B}: max |err| / max |ref| = {err:.2e} (element-wise ratio {elementwise:.2f})")
        assert err <= 1e-4, (cin, cout, H, W, B, err)
        # ... and element-wise, with a stated looser constant: 8 x the north-star bound (observed 1.2 - 4.6 on these
        # unit-variance cases).  A tile, an edge column or a channel slice gone wrong is off by O(1) = 1e4 x the bound,
        # wherever it sits and however small the values around it are.
        assert elementwise <= 8.0, (cin, cout, H, W, B, elementwise)


@pytest.mark.parametrize("linear", ["1", "0"])
def test_conv3x3_winograd_f4_unit_shapes(gpu_device, linear):
    """the unit shapes of the F(4x4) kernel through the launcher's own choice (the switch PA_WINO4_LINEAR is read once per
    process): narrow maps with at least 5 tiles per row take RUN-shaped units (16 consecutive tiles of the raster order
    -- units straddle tile rows and images -- laid out run by run with shared halos), narrower ones tile-private
    patches, wide maps row-shaped units.  Every shape against torch: the maps of 3 s chunks (40 x 149, 20 x 75, 10 x 38),
    maps smaller than one unit, one tile per image (units of 16 images), 5 / 6 tiles per row (four runs per unit),
    ragged edges, more images than fit a launch group evenly, with and without residual; guard bands around the output
    catch any store that strays out of its image."""
    import pyannote_audio_amd.ffi as ffi
    from pyannote_audio_amd.weights import winograd4_pack, winograd4_weights
    lib = ffi.load()
    g = torch.Generator().manual_seed(17 + int(linear))
    narrow = [(64, 64, 40, 149, 5, True), (128, 128, 20, 75, 7, False), (256, 256, 10, 38, 9, True),
              (32, 32, 3, 3, 37, True), (64, 32, 4, 4, 20, False), (32, 64, 6, 5, 3, True), (128, 128, 20, 75, 130, True),
              (256, 256, 10, 38, 200, False), (64, 64, 9, 18, 23, True), (32, 32, 7, 22, 41, False), (64, 64, 12, 21, 9, True)]
    wide = [(64, 64, 40, 499, 2, True), (128, 128, 20, 250, 3, False), (256, 256, 10, 125, 5, True)]
    for cin, cout, H, W, B, use_res in (narrow if linear == "1" else wide):
        x = torch.randn(B, cin, H, W, generator=g)
        wt = torch.randn(cout, cin, 3, 3, generator=g) / (3 * cin ** 0.5)
        sh = torch.randn(cout, generator=g)
        res = torch.randn(B, cout, H, W, generator=g)
        ref = F.conv2d(x, wt, stride=1, padding=1) + sh.view(1, -1, 1, 1)
        ref = F.relu(ref + res if use_res else ref)
        xd = x.permute(0, 2, 3, 1).contiguous().to(gpu_device)
        ud = winograd4_pack(winograd4_weights(wt)).to(gpu_device)
        rd = res.permute(0, 2, 3, 1).contiguous().to(gpu_device)
        shd = sh.to(gpu_device)
        # guard bands in front of and behind the output: a store that strays out of its image shows up
        ybuf = torch.full((B + 2, H, W, cout), float("nan"), device=gpu_device)
        y = ybuf[1:B + 1]
        ffi.check(lib.pa_conv3x3_wino4(ffi.ptr(xd), B, H, W, cin, ffi.ptr(ud), ffi.ptr(shd),
                                       ffi.ptr(rd) if use_res else None, C.c_void_p(y.data_ptr()), cout, 1, ffi.stream()),
                  "conv3x3_wino4")
        torch.cuda.synchronize()
        assert torch.isnan(ybuf[0]).all() and torch.isnan(ybuf[B + 1]).all(), "stores outside the output tensor"
        got = y.permute(0, 3, 1, 2).cpu()
        assert not torch.isnan(got).any(), (cin, cout, H, W, B)
        err = (got - ref).abs().max().item() / ref.abs().max().item()
        elementwise = north_star_ratio(f"wino4_units_{cin}_{cout}_{H}x{W}_B{B}", got, ref)
        assert err <= 1e-4 and elementwise <= 8.0, (cin, cout, H, W, B, err, elementwise)


def test_conv3x3_row_split_between_f4_and_f2(gpu_device):
    """a map whose height is 2 (mod 4) -- layer 4 on 10 s chunks: 10 x 125 x 256 -- is split by pa_emb_forward: rows
    0 .. H - 3 through pa_conv3x3_wino4_rows, the last two through pa_conv3x3_wino_rows; together they are the whole
    convolution (every output pixel written exactly once: Y starts as NaN), with and without the residual, for
    heights 10, 6 and 18 and a ragged width."""
    import pyannote_audio_amd.ffi as ffi
    from pyannote_audio_amd.weights import winograd4_pack, winograd4_weights, winograd_pack, winograd_weights
    lib = ffi.load()
    g = torch.Generator().manual_seed(41)
    for cin, cout, H, W, B, use_res in [(256, 256, 10, 125, 5, True), (128, 128, 6, 70, 3, False),
                                        (64, 64, 18, 129, 2, True), (256, 256, 10, 125, 40, False)]:
        x = torch.randn(B, cin, H, W, generator=g)
        wt = torch.randn(cout, cin, 3, 3, generator=g) / (3 * cin ** 0.5)
        sh = torch.randn(cout, generator=g)
        res = torch.randn(B, cout, H, W, generator=g)
        ref = F.conv2d(x, wt, stride=1, padding=1) + sh.view(1, -1, 1, 1)
        ref = F.relu(ref + res if use_res else ref)
        xd = x.permute(0, 2, 3, 1).contiguous().to(gpu_device)
        v = winograd4_pack(winograd4_weights(wt)).to(gpu_device)
        u = winograd_pack(winograd_weights(wt)).to(gpu_device)
        rd = res.permute(0, 2, 3, 1).contiguous().to(gpu_device) if use_res else None
        shd = sh.to(gpu_device)
        y = torch.full((B, H, W, cout), float("nan"), device=gpu_device)
        ffi.check(lib.pa_conv3x3_wino4_rows(ffi.ptr(xd), B, H, W, cin, ffi.ptr(v), ffi.ptr(shd), ffi.ptr(rd),
                                            ffi.ptr(y), cout, 1, H - 2, ffi.stream()), "wino4_rows")
        torch.cuda.synchronize()
        assert torch.isnan(y[:, H - 2:]).all() and not torch.isnan(y[:, :H - 2]).any()
        ffi.check(lib.pa_conv3x3_wino_rows(ffi.ptr(xd), B, H, W, cin, ffi.ptr(u), ffi.ptr(shd), ffi.ptr(rd),
                                           ffi.ptr(y), cout, 1, H - 2, ffi.stream()), "wino_rows")
        torch.cuda.synchronize()
        got = y.permute(0, 3, 1, 2).cpu()
        assert not torch.isnan(got).any()
        top = (got[:, :, :H - 2] - ref[:, :, :H - 2]).abs().max().item() / ref.abs().max().item()
        assert top <= 1e-4, (cin, H, W, top)                                     # F(4x4) rows: its per-convolution bound
        assert north_star_ratio(f"row_split_f2_rows_{cin}_{H}x{W}", got[:, :, H - 2:], ref[:, :, H - 2:]) <= 1.0


def test_shortcut_gemm_reads_strided_pixels_in_place(gpu_device):
    """pa_gemm_tn_s2 (the 1x1 stride-2 shortcut convolution + folded BatchNorm of the first block of layers 2-4,
    resnet.py:109-118) directly through the C ABI vs F.conv2d(kernel 1, stride 2): even and ODD maps (Ho = ceil(H/2):
    the last row / column of an odd map is read), the ResNet34 and Bottleneck channel pairs, a ragged last M tile,
    an output pitch wider than N."""
    import pyannote_audio_amd.ffi as ffi
    lib = ffi.load()
    g = torch.Generator().manual_seed(31)
    cases = [(3, 80, 37, 32, 64, 64), (2, 40, 19, 64, 128, 128), (2, 20, 10, 128, 256, 256), (5, 7, 5, 32, 64, 96),
             (1, 1, 1, 64, 32, 32), (2, 80, 250, 128, 512, 512), (4, 9, 33, 256, 512, 512)]
    for B, H, W, cin, cout, ldc in cases:
        x = torch.randn(B, cin, H, W, generator=g)
        wt = torch.randn(cout, cin, 1, 1, generator=g) / cin ** 0.5
        bias = torch.randn(cout, generator=g)
        ref = F.conv2d(x, wt, bias, stride=2)                      # (B, cout, Ho, Wo)
        Ho, Wo = ref.shape[2:]
        assert (Ho, Wo) == ((H - 1) // 2 + 1, (W - 1) // 2 + 1)
        xd = x.permute(0, 2, 3, 1).contiguous().to(gpu_device)
        wd = wt[:, :, 0, 0].contiguous().to(gpu_device)
        bd = bias.to(gpu_device)
        y = torch.full((B * Ho * Wo, ldc), float("nan"), device=gpu_device)
        ffi.check(lib.pa_gemm_tn_s2(ffi.ptr(xd), B, H, W, cin, ffi.ptr(wd), cin, ffi.ptr(bd), ffi.ptr(y), ldc, cout,
                                    ffi.stream()), "pa_gemm_tn_s2")
        torch.cuda.synchronize()
        got = y[:, :cout].reshape(B, Ho, Wo, cout).permute(0, 3, 1, 2).cpu()
        assert not torch.isnan(got).any(), (B, H, W, cin, cout)
        assert torch.isnan(y[:, cout:]).all(), "wrote beyond the N columns of a row"
        assert north_star_ratio(f"gemm_tn_s2_{cin}_{cout}_{H}x{W}_B{B}", got, ref) <= 1.0


def _randomise_bn(model, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for m in model.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.copy_(0.1 * torch.randn(m.running_mean.shape, generator=g))
                m.running_var.copy_(0.5 + torch.rand(m.running_var.shape, generator=g))
                m.weight.copy_(1.0 + 0.1 * torch.randn(m.weight.shape, generator=g))
                m.bias.copy_(0.1 * torch.randn(m.bias.shape, generator=g))


def test_winograd_guard_keeps_f4_on_ordinary_weights(emb, gpu_device):
    """the numerical guard of EmbeddingPack (weights._guard_winograd): on the seeded, BatchNorm-randomised ResNet34
    every stride-1 convolution is measured (29 of 32: three are strided), none is demoted, and the measured errors
    sit well inside the margins -- the table printed here is where the margins come from."""
    _, pack, _ = emb
    rows = pack.winograd_guard
    assert len(rows) == 29
    m4, m2 = pack.WINOGRAD_GUARD_MARGINS["f4"], pack.WINOGRAD_GUARD_MARGINS["f2"]
    for r in rows:
        print("guard layer%d.%d.conv%d: F(4x4) %s  F(2x2) %s -> %s" % (
            r["layer"], r["block"], r["conv"], "%.2e" % r["f4"] if r["f4"] is not None else "   n/a  ",
            "%.2e" % r["f2"] if r["f2"] is not None else "   n/a  ", r["path"]))
        assert r["path"] == ("f2" if r["layer"] == 1 else "f4")
        assert r["f2"] is not None and r["f2"] <= 0.5 * m2
        assert (r["f4"] is None) == (r["layer"] == 1)
        if r["f4"] is not None:
            assert r["f4"] <= 0.5 * m4          # (ordinary weights sit at least a factor 2 inside the margin)
    worst4 = max(r["f4"] for r in rows if r["f4"] is not None)
    worst2 = max(r["f2"] for r in rows)
    print(f"guard: worst F(4x4) {worst4:.2e} (margin {m4:.1e}), worst F(2x2) {worst2:.2e} (margin {m2:.1e})")


def test_winograd_guard_demotes_a_cancelling_convolution(gpu_device, caplog):
    """adversarial weight statistics: layer3.2.conv1 gets a DUPLICATED output channel
    (channels 0 and 1 identical, BatchNorm included) and layer3.2.conv2 weighs the pair with +K g and -K g + w
    (K = 20 000): the function the block computes is unchanged up to float32 rounding of the weights, but the
    convolution now cancels two terms 100 x larger than its output.  F(4x4)'s transform constants cost the digits
    that output needs (CPU emulation: 7e-5 of the output maximum instead of 3e-7; F(2x2) 4e-6, direct 2e-6): the
    guard measures it with the loaded weights, demotes that convolution (and only that one) and logs the decision.
    With the guard the embeddings stay inside the north-star bound; the unguarded ratio is printed beside it."""
    import logging
    from oracle import seeded_wespeaker
    from pyannote_audio_amd.embedding import EmbeddingEngine
    from pyannote_audio_amd.weights import EmbeddingPack
    model = seeded_wespeaker(seed=99)
    with torch.no_grad():
        blk = model.resnet.layer3[2]
        blk.conv1.weight[1].copy_(blk.conv1.weight[0])
        for t in (blk.bn1.weight, blk.bn1.bias, blk.bn1.running_mean, blk.bn1.running_var):
            t[1] = t[0]
        w2 = blk.conv2.weight
        big = 20000.0 * w2[:, 0].clone()
        w2[:, 1] -= big - w2[:, 0]
        w2[:, 0] = big
    with caplog.at_level(logging.WARNING, logger="pyannote_audio_amd"):
        pack = EmbeddingPack(model.state_dict(), gpu_device)
    by_key = {(r["layer"], r["block"], r["conv"]): r for r in pack.winograd_guard}
    bad = by_key[(3, 2, 2)]
    print("guard on the cancelling convolution:", bad)
    assert bad["f4"] > pack.WINOGRAD_GUARD_MARGINS["f4"] and bad["path"] in ("f2", "direct")
    assert any("layer3.2.conv2" in rec.getMessage() for rec in caplog.records)
    demoted = [k for k, r in by_key.items() if r["path"] != ("f2" if k[0] == 1 else "f4")]
    assert demoted == [(3, 2, 2)], demoted
    # the struct the kernels read no longer carries the F(4x4) image of that convolution
    index = 3 + 4 + 2
    assert not pack.struct.blk_v2[index] and pack.struct.blk_v1[index]
    eng = EmbeddingEngine(pack, max_chunks=4)
    x = _wave(3, 48000, seed=21)
    g = torch.Generator().manual_seed(4)
    masks = (torch.rand(3, 3, 173, generator=g) < 0.7).float()
    with torch.inference_mode():
        ref = model(x, weights=masks)
    out = eng.forward(x.to(gpu_device), masks.to(gpu_device))
    torch.cuda.synchronize()
    assert north_star_ratio("emb_guarded_cancelling_conv", out, ref) <= 1.0
    # the same checkpoint with the guard switched off keeps F(4x4) everywhere (what a user would have got in round 4)
    unguarded = EmbeddingPack(model.state_dict(), gpu_device, guard=False)
    assert unguarded.winograd_guard == [] and unguarded.struct.blk_v2[index]
    out_u = EmbeddingEngine(unguarded, max_chunks=4).forward(x.to(gpu_device), masks.to(gpu_device))
    torch.cuda.synchronize()
    print("unguarded embeddings, north-star ratio:", north_star_ratio("emb_unguarded_cancelling_conv", out_u, ref))


def test_embeddings_f4_on_versus_off(emb, gpu_device, monkeypatch):
    """end to end: the embeddings with F(4x4) on layers 2-4 (default) and with F(2x2) everywhere (PA_WINOGRAD4="")
    agree with each other far inside the north-star bound -- the two paths share nothing but the weights"""
    from pyannote_audio_amd.embedding import EmbeddingEngine
    from pyannote_audio_amd.weights import EmbeddingPack
    model, pack, eng = emb
    monkeypatch.setenv("PA_WINOGRAD4", "")
    pack2 = EmbeddingPack(model.state_dict(), gpu_device)
    assert not pack2.winograd4_layers and all(r["f4"] is None for r in pack2.winograd_guard)
    x = _wave(4, 160000, seed=8)
    g = torch.Generator().manual_seed(3)
    masks = (torch.rand(4, 3, 589, generator=g) < 0.7).float()
    a = eng.forward(x.to(gpu_device), masks.to(gpu_device))
    b = EmbeddingEngine(pack2, max_chunks=3).forward(x.to(gpu_device), masks.to(gpu_device))
    torch.cuda.synchronize()
    assert north_star_ratio("emb_f4_vs_f2", a, b) <= 0.5


@pytest.mark.parametrize("num_blocks", [(1, 1, 1, 1), (2, 3, 2, 2)])
def test_bottleneck_resnet_matches_oracle(gpu_device, num_blocks):
    """WeSpeakerResNet152/221/293 (wespeaker/resnet.py:148-212, 477-507): Bottleneck
    blocks = 1x1 GEMMs over the NHWC pixels + the existing 3x3 kernels + a residual/ReLU GEMM epilogue.
    Shallow stacks keep the CPU oracle fast; depth and block type are read from the state-dict keys, so
    the deep variants only differ by the loop counts."""
    from oracle.models import Bottleneck, WeSpeakerResNet34 as OracleNet
    from pyannote_audio_amd.embedding import EmbeddingEngine
    from pyannote_audio_amd.weights import EmbeddingPack
    torch.manual_seed(11)
    model = OracleNet(num_blocks=num_blocks, block=Bottleneck).eval()
    with torch.no_grad():
        for name, m in model.named_modules():
            if isinstance(m, torch.nn.BatchNorm2d):          # exercise the BatchNorm folding
                m.running_mean.normal_(0, 0.1)
                m.running_var.uniform_(0.5, 1.5)
                m.weight.uniform_(0.6, 1.2)
                m.bias.normal_(0, 0.1)
    pack = EmbeddingPack(model.state_dict(), gpu_device)
    assert pack.bottleneck and pack.num_blocks == tuple(num_blocks)
    eng = EmbeddingEngine(pack)
    B, N = 3, 48000
    x = _wave(B, N, seed=13)
    masks = (torch.rand(B, 2, 173, generator=torch.Generator().manual_seed(4)) < 0.7).float()
    with torch.inference_mode():
        ref = model(x, weights=masks)
        ref1 = model(x[:2])
    out = eng.forward(x.to(gpu_device), masks.to(gpu_device))
    out1 = eng.forward(x[:2].to(gpu_device))
    assert out.shape == ref.shape == (B, 2, 256)
    assert north_star_ratio(f"bottleneck{num_blocks}_emb", out, ref) <= 1.0
    assert north_star_ratio(f"bottleneck{num_blocks}_emb_unweighted", out1, ref1) <= 1.0
