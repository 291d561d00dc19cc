"""csrc/tile_queue.h -- the tile-claiming protocol of the persistent convolution kernels -- compiled for the
host (g++, std::atomic stand-ins for atomicAdd / atomicExch) and driven by real threads: every tile exactly
once, holes of a non-multiple-of-8 index space never, the counter block back to zero after every launch so
that it can be reused without a memset, for grids smaller and larger than the tile count and with late
workgroups.  The same header is what the HIP kernels include."""
import shutil
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("g++ not available")
    exe = tmp_path_factory.mktemp("tq") / "harness"
    subprocess.check_call([gxx, "-O2", "-std=c++17", "-pthread", "-Wno-unknown-pragmas",
                           "-I", str(ROOT / "pyannote-audio_amd" / "csrc"),
                           str(ROOT / "tests" / "native" / "tile_queue_harness.cpp"), "-o", str(exe)])
    return exe


@pytest.mark.parametrize("workgroups,tiles,upto", [
    (16, 4096, 0),     # the usual case: many tiles per workgroup
    (16, 8, 0),        # fewer tiles than workgroups
    (3, 1024, 0),      # XCDs without any workgroup: their tiles are stolen
    (1, 256, 0),       # a single workgroup drains all eight counters
    (13, 1001, 1),     # index space not a multiple of 8 (stride-2 kernel): holes are never handed out
    (16, 5, 1),
    (8, 0, 0),         # nothing to do
])
def test_every_tile_exactly_once_and_block_resets(harness, workgroups, tiles, upto):
    for seed in (1, 2):
        rc = subprocess.run([str(harness), str(workgroups), str(tiles), "6", str(upto), str(seed)],
                            capture_output=True, text=True, timeout=120)
        assert rc.returncode == 0, rc.stdout + rc.stderr
