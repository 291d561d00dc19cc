"""The C-ABI library builds (hipcc cross-compiles gfx950 without a GPU), loads, and exports every
symbol that include/pyannote_amd.h declares.  No compute is launched here."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "pyannote_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(pa_[a-z0-9_]+)\s*\(", src)))


def test_library_builds_and_exports_every_declared_symbol():
    import __graft_entry__
    __graft_entry__.build()
    import pyannote_audio_amd.ffi as ffi
    lib = ffi.load()
    names = _declared()
    assert len(names) >= 20
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, f"declared in the header but not exported: {missing}"
    assert lib.pa_version() >= 100
    assert isinstance(lib.pa_last_error(), bytes)
    # ... and the other way round: nothing named pa_* is exported that the header does not declare (internal launchers
    # shared between translation units are hidden: csrc/common.h PA_INTERNAL)
    import subprocess
    from pyannote_audio_amd._build import LIB_PATH
    out = subprocess.run(["nm", "-D", "--defined-only", str(LIB_PATH)], capture_output=True, text=True, check=True).stdout
    exported = {ln.split()[-1] for ln in out.splitlines() if ln.split()[-1].startswith("pa_") and " T " in ln}
    assert exported == set(names), f"exported but undeclared: {sorted(exported - set(names))}"


def test_frame_arithmetic_entry_points():
    import pyannote_audio_amd.ffi as ffi
    lib = ffi.load()
    assert lib.pa_seg_num_frames(160000, 10) == 589
    assert lib.pa_seg_num_frames(80000, 10) == 293
    assert lib.pa_seg_num_frames(100, 10) == 0
    assert lib.pa_emb_num_fbank_frames(160000) == 998
    assert lib.pa_emb_num_fbank_frames(399) == 0


def test_winograd_pack_host_matches_python_packer():
    """include/pyannote_amd.h: `pa_conv3x3_wino` takes the slab image, not [16][cout][cin]; the header's
    host-side packer (what a C / C++ caller uses) equals weights.winograd_pack(winograd_weights(w * scale))."""
    import numpy as np
    import torch
    import pyannote_audio_amd.ffi as ffi
    from pyannote_audio_amd.weights import winograd_pack, winograd_weights
    lib = ffi.load()
    f32p = ctypes.POINTER(ctypes.c_float)
    lib.pa_winograd_pack_host.argtypes = [f32p, f32p, ctypes.c_int, ctypes.c_int, f32p]
    g = torch.Generator().manual_seed(0)
    for cout, cin in ((32, 32), (64, 32), (128, 128)):
        w = torch.randn(cout, cin, 3, 3, generator=g)
        scale = 0.5 + torch.rand(cout, generator=g)
        want = winograd_pack(winograd_weights(w * scale.view(-1, 1, 1, 1))).numpy().reshape(-1)
        got = np.full(16 * cout * cin, np.nan, dtype=np.float32)
        wn, sn = np.ascontiguousarray(w.numpy()), np.ascontiguousarray(scale.numpy())
        rc = lib.pa_winograd_pack_host(wn.ctypes.data_as(f32p), sn.ctypes.data_as(f32p), cout, cin,
                                       got.ctypes.data_as(f32p))
        assert rc == 0 and not np.isnan(got).any()
        assert np.allclose(got, want, rtol=1e-6, atol=1e-7)
    assert lib.pa_winograd_pack_host(wn.ctypes.data_as(f32p), None, 48, 16, got.ctypes.data_as(f32p)) == 3


def test_struct_layout_matches_header():
    """ctypes mirror of pa_seg_weights / pa_emb_weights has the size the C compiler gives."""
    import subprocess, tempfile
    import pyannote_audio_amd.ffi as ffi
    src = '#include <stdio.h>\n#include "pyannote_amd.h"\nint main(){printf("%zu %zu %zu\\n", sizeof(pa_seg_weights), sizeof(pa_emb_weights), sizeof(pa_xvec_weights));return 0;}\n'
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "s.c")
        open(c, "w").write(src)
        exe = os.path.join(d, "s")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", exe])
        a, b, c = map(int, subprocess.check_output([exe]).split())
    assert ctypes.sizeof(ffi.SegWeights) == a
    assert ctypes.sizeof(ffi.EmbWeights) == b
    assert ctypes.sizeof(ffi.XvecWeights) == c


def test_no_cpu_fallback():
    """models refuse to compute on the host; kernels refuse CPU tensors."""
    import pytest
    import torch
    import pyannote_audio_amd.ffi as ffi
    with pytest.raises(RuntimeError):
        ffi.ptr(torch.zeros(4))
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError):
            ffi.require_gpu()
