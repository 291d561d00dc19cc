#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=gpurun_out/c2; mkdir -p $O
V=pyannote-audio_amd/build/variants
PA_LIB=$V/libpa_stamp.so timeout 200 python tools/wino_stamps.py $O/stamps.npz > $O/stamps.txt 2>&1
timeout 300 python bench.py --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err
cat $O/stamps.txt; python - <<'PY'
import json
d=json.loads(open("gpurun_out/c2/bench_default.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"])
for f in d.get("batch_timeline_s", []): print(f)
PY
