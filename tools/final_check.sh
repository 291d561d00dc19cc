#!/bin/bash
# round-end validation on the GPU box: full GPU suite, default bench line, smoke, N = 1 under torchrun (the
# configs[4] code path with one rank), then tools/capture_profiles.sh (kernel stats + PMC passes at HEAD)
# usage (via gpurun): [SKIP_LINKAGE=1] bash tools/final_check.sh r5      (the per-configuration PMC traffic of
# bench.py's `configs` entries: sh tools/capture_config_traffic.sh r5, its own call)
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
tag=${1:-r3}
O=gpurun_out/final_$tag; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q > $O/gpu_tests.txt 2>&1
tail -4 $O/gpu_tests.txt
timeout 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1
tail -2 $O/smoke.txt
timeout 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29513 \
    bench.py --gpus 1 --steps 5 --no-cpu-baseline > $O/bench_torchrun_n1.json 2> $O/torchrun.err
timeout 200 python bench.py --sequential --steps 3 --no-cpu-baseline > $O/bench_sequential.json 2>/dev/null
timeout 700 sh tools/capture_profiles.sh $tag > $O/capture.log 2>&1
python - <<PY
import json
for f in ("bench_default.json", "bench_torchrun_n1.json", "bench_sequential.json"):
    try:
        d = json.loads(open("$O/" + f).read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], d["roofline"]["traffic_source"])
    except Exception as e:
        print(f, "ERR", e)
PY
ls gpurun_out/prof_$tag
[ -n "$SKIP_LINKAGE" ] && exit 0
# configs[4] on one GPU (joint clustering over 1..8 one-hour files) and the dendrogram merge alone
timeout 400 python tools/joint_scale.py 1 2 4 8 > $O/joint_scale.txt 2>&1; tail -12 $O/joint_scale.txt
timeout 200 python tools/time_linkage.py 7000 14000 29000 57000 > $O/time_linkage.txt 2>&1; tail -8 $O/time_linkage.txt
