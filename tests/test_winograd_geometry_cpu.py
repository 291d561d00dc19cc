"""csrc/emb_winograd_geom.h -- the integer geometry of the Winograd convolution kernel (LDS layout and swizzle,
per-lane DMA offsets with their halo class bits, the transform's ds_read addresses, the XCD-aware tile
order) -- compiled UNCHANGED for the host and replayed for every wave and lane
(tests/native/winograd_geom_harness.cpp): each transform read returns exactly the patch element its Winograd
tile needs or a hardware zero in the halo, never an unwritten location, every ds_read_b128 is bank-conflict
free, and the tile order hands out every (pixel tile, cout slice) once with the slices of a pixel tile on one
XCD -- on the four ResNet34 layer shapes, ragged images and all three tile geometries."""
import shutil
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def test_dma_layout_matches_transform_reads_and_is_conflict_free(tmp_path):
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("g++ not available")
    exe = tmp_path / "geom"
    subprocess.check_call([gxx, "-O2", "-std=c++17", "-Wno-unknown-pragmas",
                           "-I", str(ROOT / "pyannote-audio_amd" / "csrc"),
                           str(ROOT / "tests" / "native" / "winograd_geom_harness.cpp"), "-o", str(exe)])
    rc = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert rc.returncode == 0, rc.stdout + rc.stderr


def test_f4_dma_layout_matches_transform_reads(tmp_path):
    """csrc/emb_winograd4_geom.h (Winograd F(4x4, 3x3), k_conv3x3_wino4): patch DMA vs the 6x6 transform reads of
    every wave and lane, zeros for EVERY pixel outside the image, contiguous 512-byte ds_read_b64s, U-slab reads
    (tests/native/winograd4_geom_harness.cpp)."""
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("g++ not available")
    exe = tmp_path / "geom4"
    subprocess.check_call([gxx, "-O2", "-std=c++17", "-Wno-unknown-pragmas",
                           "-I", str(ROOT / "pyannote-audio_amd" / "csrc"),
                           str(ROOT / "tests" / "native" / "winograd4_geom_harness.cpp"), "-o", str(exe)])
    rc = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert rc.returncode == 0, rc.stdout + rc.stderr
