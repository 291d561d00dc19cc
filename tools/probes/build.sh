#!/bin/sh
# builds the development probes into tools/_dbg/libprobes.so (git-ignored; travels with gpurun)
set -e
cd "$(dirname "$0")"
mkdir -p ../_dbg
# (the split-MFMA probe a second time with unpacked f32 arithmetic that the compiler may not re-pack)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DSP_UNPACKED=1 -fno-slp-vectorize -o ../_dbg/libprobes_unpacked.so split_mfma_probe.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -o ../_dbg/libprobes.so *.hip
