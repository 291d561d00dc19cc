#!/bin/bash
# Spend GPU minutes only on things that build: (1) the product library and every variant of
# tools/build_variants.py, (2) the probes, (3) the CPU import of the package -- and only if all of that succeeds
# hand the command to gpurun (every call is charged 0.3-5 minutes whatever it does; round 2 lost 7 minutes to
# two calls whose variant libraries had failed to compile).
# usage: [PA_VARIANTS="tag ..."] tools/gpurun_checked.sh [--timeout S] -- '<command run on the GPU box>'
set -euo pipefail
cd "$(dirname "$0")/.."
python -c "import __graft_entry__ as g; g.build()"
python tools/build_variants.py ${PA_VARIANTS:-}   # only the A/B variants this call needs (none by default)
sh tools/probes/build.sh 2>&1 | { grep -i " error" && exit 1 || true; }
python - <<'PY'
import glob, subprocess, sys
missing = [p for p in glob.glob("pyannote-audio_amd/build/variants/*_*.o") if not __import__("os").path.exists(p)]
libs = glob.glob("pyannote-audio_amd/build/variants/libpa_*.so")
print("variant libraries:", ", ".join(sorted(libs)) or "none")
PY
exec /usr/local/graft/bin/gpurun "$@"
