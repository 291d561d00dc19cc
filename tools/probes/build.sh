#!/bin/sh
# builds the development probes into tools/_dbg/libprobes.so (git-ignored; travels with gpurun)
set -e
cd "$(dirname "$0")"
mkdir -p ../_dbg
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -o ../_dbg/libprobes.so *.hip
