#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=gpurun_out/c10; mkdir -p $O
PROBE=reserve timeout 120 python tools/probes/interference_probe.py > $O/interference.txt 2>&1
timeout 300 python -m pytest tests/test_emb_gpu.py tests/test_batch_gpu.py -q -x > $O/test.txt 2>&1
timeout 300 python bench.py --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err
cat $O/interference.txt; tail -3 $O/test.txt
python - <<'PY'
import json
d=json.loads(open("gpurun_out/c10/bench_default.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["kernels"]["k_conv3x3_wino"], d["kernels"]["k_conv3x3"])
for f in d.get("batch_timeline_s", []): print({k:f[k] for k in ("front_start","segmentation","embeddings_queued","embeddings","tail_start","tail_done")})
PY
