#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=gpurun_out/c9; mkdir -p $O
timeout 300 python -m pytest tests/test_emb_gpu.py tests/test_batch_gpu.py -q -x > $O/test.txt 2>&1
timeout 300 python bench.py --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err
tail -3 $O/test.txt
python - <<'PY'
import json
d=json.loads(open("gpurun_out/c9/bench_default.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["kernels"]["k_conv3x3_wino"])
for f in d.get("batch_timeline_s", []): print({k:f[k] for k in ("front_start","segmentation","embeddings_queued","embeddings","tail_start","tail_done")})
PY
