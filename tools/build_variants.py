"""Build A/B variants of one kernel file next to the product library (development aid): the variant's
object replaces the product object at link time.  usage: python tools/build_variants.py <tag> [<tag> ...] | --all | --list
Variants land in pyannote-audio_amd/build/variants/libpa_<tag>.so; select one with PA_LIB=<path>.  Only the tags named
are built, and the variants of earlier calls are REMOVED first: every .so under the repository travels to the GPU box
with each gpurun call (round 5 shipped 47 MB of them every time)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pyannote_audio_amd import _build

VARIANTS = {   # tag -> (source file, extra flags, git revision of the source or None for the work tree)
    "lfstamp": ("linkage_fast.hip", "-DPA_LF_STAMP=1 -ffp-contract=off", None),   # heap-free merge: cycles per phase of a round (printf)
    "w4defer2": ("emb_winograd4.hip", "-DPA_W4_DEFER_STORES=2", None),   # F(4x4): held stores also with a residual (spills: slower)
    "conv_r4order": ("emb_resnet.hip", "", "59d35f4"),   # k_conv3x3 with the cout slice as the slowest tile index (round 4)
    "conv_prev": ("emb_resnet.hip", "", "b01ddc8"),   # k_conv3x3 with the staging offsets recomputed every stage
    "w4tstores": ("emb_winograd4.hip", "-DPA_W4_STORE_AUX=0", None),        # F(4x4): output stores with the default (temporal) policy
    "w4ntr": ("emb_winograd4.hip", "-DPA_W4_RES_AUX=2", None),              # F(4x4): residual loads non-temporal
    "w32waitstores": ("emb_winograd.hip", "-DPA_WINO32_WAIT_STORES=1", None),   # k_conv3x3_wino32: the step barrier waits for the epilogue's stores too
    "w32patchlast": ("emb_winograd.hip", "-DPA_WINO32_PATCH_FIRST=0", None),   # k_conv3x3_wino32: patch DMA behind the epilogue (round 3)
    "winonty": ("emb_winograd.hip", "-DPA_WINO_STORE_AUX=2", None),         # F(2x2) kernels: output stores non-temporal
    "convnty": ("emb_resnet.hip", "-DPA_CONV_STORE_AUX=2", None),           # direct kernel: output stores non-temporal
    "w4waitstores": ("emb_winograd4.hip", "-DPA_W4_STORES_IN_FLIGHT=0", None),   # F(4x4): a tile's first stage waits for the previous tile's stores
    "w4earlybar": ("emb_winograd4.hip", "-DPA_W4_LATE_BARRIER=0", None),   # F(4x4): stage barrier in front of the transform (round 4)
    "w4stamp": ("emb_winograd4.hip", "-DPA_W4_STAMP=1", None),                   # F(4x4): phase stamps (tools/wino4_stamps.py)
    "stem_r5": ("emb_resnet.hip", "", "3a63397"),   # k_stem with 8 channels of one time step per thread (rounds 2-5)
    "stamp": ("emb_winograd.hip", "-DPA_WINO_STAMP=1", None),
    "norefresh": ("emb_winograd.hip", "-DPA_WINO_REFRESH=0", None),   # 128-channel residual kernel without the pinned residual loads
    "nortouch": ("emb_winograd.hip", "-DPA_WINO_RTOUCH=0", None),   # without the residual line touch
    "norpre": ("emb_winograd.hip", "-DPA_WINO_RPRE=0", None),   # without the residual prefetch through LDS
}


def main():
    import shutil
    args = sys.argv[1:]
    if "--list" in args:
        for tag, (src, flags, rev) in VARIANTS.items():
            print(f"{tag:18s} {src:22s} {flags} {rev or ''}")
        return
    tags = list(VARIANTS) if "--all" in args else args
    unknown = [t for t in tags if t not in VARIANTS]
    if unknown:
        raise SystemExit(f"unknown variant(s) {unknown}; --list shows them")
    _build.build_library()
    out = _build.PKG_DIR / "build" / "variants"
    shutil.rmtree(out, ignore_errors=True)
    if not tags:
        print("no variant requested: pyannote-audio_amd/build/variants/ removed")
        return
    out.mkdir(parents=True, exist_ok=True)
    objs = {p.name: p for p in (_build.PKG_DIR / "build").glob("*.o")}
    for tag, (src, flags, rev) in ((t, VARIANTS[t]) for t in tags):
        path = _build.CSRC / src
        if rev:
            text = subprocess.check_output(["git", "show", f"{rev}:pyannote-audio_amd/csrc/{src}"], cwd=ROOT)
            path = _build.CSRC / f"_variant_{tag}_{src}"
            path.write_bytes(text)
        obj = out / f"{tag}_{src}.o"
        cmd = ["hipcc", f"--offload-arch={_build.ARCH}", "-O3", "-std=c++17", "-fPIC", "-x", "hip", *flags.split(),
               "-I", str(_build.PKG_DIR.parent / "include"), "-c", str(path), "-o", str(obj)]
        subprocess.check_call(cmd)
        if rev:
            path.unlink()
        link = [str(o) for name, o in objs.items() if name != src + ".o"] + [str(obj)]
        lib = out / f"libpa_{tag}.so"
        subprocess.check_call(["hipcc", f"--offload-arch={_build.ARCH}", "-shared", "-fPIC", "-o", str(lib)] + link)
        print("built", lib)


if __name__ == "__main__":
    main()
