#!/bin/bash
# round-end validation on the GPU box: full GPU suite, default bench line, smoke, rocprof kernel stats
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=gpurun_out/final; mkdir -p $O
R=$PWD
timeout 240 python -m pytest tests -m gpu -x -q > $O/test_all.txt 2>&1
tail -4 $O/test_all.txt
timeout 80 python bench.py > $O/bench_default.json 2> $O/bench_default.err
timeout 30 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1
tail -2 $O/smoke.txt
cd /tmp && timeout 60 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/final_stats -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --sequential > $R/$O/bench_under_rocprof.json 2> $R/$O/stats.err; cd $R
find /tmp/final_stats -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
python - <<'PY'
import json
for f in ("bench_default.json", "bench_under_rocprof.json"):
    try:
        d = json.loads(open("gpurun_out/final/" + f).read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"])
    except Exception as e:
        print(f, "ERR", e)
PY
head -4 $O/kernel_stats.csv | cut -c1-160
