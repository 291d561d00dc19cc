"""In-tree build of libpyannote_amd.so for gfx950 with hipcc (no cmake, no JIT cache)."""
from __future__ import annotations

import os
import shutil
import subprocess
from pathlib import Path

PKG_DIR = Path(__file__).resolve().parent
CSRC = PKG_DIR / "csrc"
LIB_PATH = PKG_DIR / "libpyannote_amd.so"
ARCH = "gfx950"


def sources() -> list[Path]:
    return sorted(list(CSRC.glob("*.hip")) + list(CSRC.glob("*.cpp")))


def _stale() -> bool:
    if not LIB_PATH.exists():
        return True
    t = LIB_PATH.stat().st_mtime
    deps = sources() + list(CSRC.glob("*.h")) + [PKG_DIR.parent / "include" / "pyannote_amd.h"]
    return any(p.stat().st_mtime > t for p in deps)


def build_library(force: bool = False, verbose: bool = False) -> Path:
    """Compile every HIP/C++ source into one shared library.  hipcc cross-compiles for gfx950
    without a GPU, so this also runs in the CPU-only build container."""
    if not force and not _stale():
        return LIB_PATH
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build libpyannote_amd.so")
    obj_dir = PKG_DIR / "build"
    obj_dir.mkdir(exist_ok=True)
    objs = []
    procs = []
    for src in sources():
        obj = obj_dir / (src.name + ".o")
        objs.append(obj)
        if not force and obj.exists() and obj.stat().st_mtime > max(
                src.stat().st_mtime, *(h.stat().st_mtime for h in CSRC.glob("*.h")),
                (PKG_DIR.parent / "include" / "pyannote_amd.h").stat().st_mtime):
            continue
        extra = []
        for line in src.read_text().splitlines():
            if line.startswith("// hipcc-flags:"):
                extra += line.split(":", 1)[1].split()
        extra += os.environ.get("PA_EXTRA_FLAGS", "").split()   # development aids only (e.g. -DPA_WINO_DEBUG)
        cmd = [hipcc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-x", "hip", *extra,
               "-I", str(PKG_DIR.parent / "include"), "-c", str(src), "-o", str(obj)]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src.name}:\n{out.decode()}")
    cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", str(LIB_PATH)] + [str(o) for o in objs]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout.decode()}")
    return LIB_PATH


if __name__ == "__main__":
    print(build_library(force="--force" in os.sys.argv, verbose=True))
