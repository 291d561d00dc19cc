"""CPU check of the Winograd F(2x2,3x3) algebra the HIP kernel implements (csrc/emb_winograd.hip): host
weight transform U = G g G^T (weights.winograd_weights) + the kernel's B^T d B / A^T M A formulas,
restated with numpy, reproduce torch's conv2d."""
import torch
import torch.nn.functional as F


def test_winograd_formulas_match_conv2d():
    from pyannote_audio_amd.weights import winograd_weights
    torch.manual_seed(0)
    cin, cout, H, W = 5, 4, 6, 10
    x = torch.randn(1, cin, H, W, dtype=torch.float64)
    w = torch.randn(cout, cin, 3, 3)
    ref = F.conv2d(x, w.double(), padding=1)[0]
    U = winograd_weights(w).double().reshape(4, 4, cout, cin)
    xp = F.pad(x, (1, 1, 1, 1))[0]
    out = torch.zeros(cout, H, W, dtype=torch.float64)
    for ty in range(H // 2):
        for tx in range(W // 2):
            d = xp[:, 2 * ty:2 * ty + 4, 2 * tx:2 * tx + 4]
            r = torch.stack([d[:, :, 0] - d[:, :, 2], d[:, :, 1] + d[:, :, 2], d[:, :, 2] - d[:, :, 1],
                             d[:, :, 1] - d[:, :, 3]], dim=2)            # row transform: r[c][i][b]
            v = torch.stack([r[:, 0] - r[:, 2], r[:, 1] + r[:, 2], r[:, 2] - r[:, 1], r[:, 1] - r[:, 3]],
                            dim=1)                                       # column transform: v[c][a][b]
            m = torch.einsum("aboc,cab->oab", U, v)
            s = m[:, :, 0] + m[:, :, 1] + m[:, :, 2]
            dd = m[:, :, 1] - m[:, :, 2] - m[:, :, 3]
            out[:, 2 * ty, 2 * tx] = s[:, 0] + s[:, 1] + s[:, 2]
            out[:, 2 * ty, 2 * tx + 1] = dd[:, 0] + dd[:, 1] + dd[:, 2]
            out[:, 2 * ty + 1, 2 * tx] = s[:, 1] - s[:, 2] - s[:, 3]
            out[:, 2 * ty + 1, 2 * tx + 1] = dd[:, 1] - dd[:, 2] - dd[:, 3]
    assert (out - ref).abs().max() < 1e-5    # U is rounded to float32


def test_lds_swizzle_is_conflict_free():
    """slot (g + 2*((row>>2)&1)) & 3 of 64-byte LDS rows: every ds_read_b128 lane group of 16 lanes
    (rows consecutive in lane & 15, quad lane >> 4) touches 16 distinct 16-byte bank groups, for every
    row alignment -- the property csrc/emb_winograd.hip relies on (MI355X_MICROARCH.md LDS table)."""
    groups = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)),
              list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
              list(range(32, 36)) + list(range(44, 48)) + list(range(52, 60)),
              list(range(36, 44)) + list(range(48, 52)) + list(range(60, 64))]
    for base in range(64):
        for grp in groups:
            slots = set()
            for lane in grp:
                t, g = lane & 15, lane >> 4
                row = base + t
                slots.add((4 * row + ((g + 2 * ((row >> 2) & 1)) & 3)) % 16)
            assert len(slots) == 16


def test_winograd_pack_is_the_lds_image():
    """weights.winograd_pack: slab (cout slice o, cin stage c) row r = 32 xi + n, physical slot
    (g + 2 ((r >> 2) & 1)) & 3 holds U[xi][32 o + n][16 c + 4 g : 16 c + 4 g + 4]."""
    from pyannote_audio_amd.weights import winograd_pack
    cout, cin = 64, 48
    U = torch.arange(16 * cout * cin, dtype=torch.float32).reshape(16, cout, cin)
    P = winograd_pack(U)
    assert P.shape == (cout // 32, cin // 16, 512, 4, 4) and P.is_contiguous()
    for (o, c, xi, n, g) in [(0, 0, 0, 0, 0), (1, 2, 15, 31, 3), (1, 1, 7, 4, 1), (0, 2, 3, 5, 2), (1, 0, 9, 12, 0)]:
        r = 32 * xi + n
        slot = (g + 2 * ((r >> 2) & 1)) & 3
        assert torch.equal(P[o, c, r, slot], U[xi, 32 * o + n, 16 * c + 4 * g:16 * c + 4 * g + 4])
