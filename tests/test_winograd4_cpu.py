"""Winograd F(4x4, 3x3) (csrc/emb_winograd4.hip) on the host: the weight image (weights.winograd4_weights /
winograd4_pack, and the C twin pa_winograd4_pack_host) and a numpy replay of the kernel's arithmetic -- the 12-
operation B^T recipe on channel pairs, the point order xi = 6a + b, U read at the kernel's slab addresses, the A^T
recipe of the epilogue -- against torch's conv2d (the reference's operation, wespeaker/resnet.py:92-107)."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from pyannote_audio_amd.weights import winograd4_pack, winograd4_weights


def bt(x):
    """the kernel's wino4_bt on the first axis (6, ...) in float32"""
    a = x[4] - 4 * x[2]
    b = x[3] - 4 * x[1]
    c = x[4] - x[2]
    d = x[3] - x[1]
    return np.stack([4 * x[0] + (x[4] - 5 * x[2]), a + b, a - b, c + 2 * d, c - 2 * d,
                     4 * x[1] + (x[5] - 5 * x[3])]).astype(np.float32)


def at(m):
    """the kernel's wino4_at on the first axis (6, ...) -> (4, ...)"""
    s1, d1, s2, d2 = m[1] + m[2], m[1] - m[2], m[3] + m[4], m[3] - m[4]
    return np.stack([m[0] + s1 + s2, d1 + 2 * d2, s1 + 4 * s2, d1 + 8 * d2 + m[5]]).astype(np.float32)


@pytest.mark.parametrize("H,W,cin,cout", [(8, 12, 32, 32), (5, 7, 40, 64), (10, 125, 32, 32)])
def test_kernel_arithmetic_replayed_on_the_host(H, W, cin, cout):
    g = torch.Generator().manual_seed(H * W)
    x = torch.randn(1, cin, H, W, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) / (3 * cin ** 0.5)
    want = F.conv2d(x, w, padding=1)[0].numpy()
    slabs = winograd4_pack(winograd4_weights(w)).numpy()            # [cout/32][cin/8][1152][8]
    th, tw = -(-H // 4), -(-W // 4)
    xp = np.zeros((cin, 4 * th + 2, 4 * tw + 2), dtype=np.float32)
    xp[:, 1:H + 1, 1:W + 1] = x[0].numpy()
    out = np.zeros((cout, 4 * th, 4 * tw), dtype=np.float32)
    for ty in range(th):
        for tx in range(tw):
            d = xp[:, 4 * ty:4 * ty + 6, 4 * tx:4 * tx + 6]                       # (cin, 6, 6)
            tt = bt(d.transpose(1, 2, 0))                                          # columns: (i, j, c) over i
            v = bt(tt.transpose(1, 0, 2)).transpose(1, 0, 2)                       # rows: over j -> v[a][b][c]
            M = np.zeros((36, cout), dtype=np.float32)
            for xi in range(36):
                for ns in range(cout // 32):
                    for st in range(cin // 8):
                        u = slabs[ns, st, 32 * xi:32 * xi + 32, :].copy()          # (32 cout, 8 cin): kernel's rows
                        sw = ((np.arange(32) >> 3) & 1).astype(bool)                # ... quads swapped where bit 3 of n
                        u[sw] = np.concatenate([u[sw, 4:], u[sw, :4]], axis=1)
                        M[xi, 32 * ns:32 * ns + 32] += u @ v[xi // 6, xi % 6, 8 * st:8 * st + 8]
            z = at(M.reshape(6, 6, cout))                                          # (4, 6, cout): over a
            y = at(z.transpose(1, 0, 2))                                           # (4 q, 4 p, cout): over b
            out[:, 4 * ty:4 * ty + 4, 4 * tx:4 * tx + 4] = y.transpose(2, 1, 0)
    got = out[:, :H, :W]
    assert np.abs(got - want).max() <= 1e-4 * np.abs(want).max()


def test_host_packer_of_the_c_abi_matches_the_python_pack():
    import pyannote_audio_amd.ffi as ffi
    lib = ffi.load()
    g = torch.Generator().manual_seed(3)
    w = torch.randn(64, 40, 3, 3, generator=g)
    scale = torch.rand(64, generator=g) + 0.5
    want = winograd4_pack(winograd4_weights(w * scale.view(-1, 1, 1, 1))).numpy().reshape(-1)
    got = np.zeros(36 * 64 * 40, dtype=np.float32)
    fp = C.POINTER(C.c_float)
    lib.pa_winograd4_pack_host.argtypes = [fp, fp, C.c_int, C.c_int, fp]
    lib.pa_winograd4_pack_host.restype = C.c_int
    wc, sc = np.ascontiguousarray(w.numpy()), np.ascontiguousarray(scale.numpy())
    rc = lib.pa_winograd4_pack_host(wc.ctypes.data_as(fp), sc.ctypes.data_as(fp), 64, 40, got.ctypes.data_as(fp))
    assert rc == 0
    # same layout, same float64 arithmetic up to the summation order (G has 1/6 and 1/24: the float32 rounding of
    # the two float64 sums may differ in the last bit; F(2x2)'s G is dyadic and compares bit for bit)
    assert np.allclose(got, want, rtol=3e-7, atol=1e-9)
    assert lib.pa_winograd4_pack_host(wc.ctypes.data_as(fp), None, 60, 40, got.ctypes.data_as(fp)) == 3
