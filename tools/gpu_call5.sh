#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=gpurun_out/c6; mkdir -p $O
R=$PWD
cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/c6_trace -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline > $R/$O/bench_trace.json 2> $R/$O/trace.err; cd $R
python tools/trace_table.py /tmp/c6_trace $O/trace.npz
head -c 600 $O/bench_trace.json
