#!/bin/bash
# round-2 experiment call: Winograd variants A/B + pipelined bench + idle-gap report
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=gpurun_out/c1; mkdir -p $O
V=pyannote-audio_amd/build/variants
( for t in head v0 v1 v2 v3 v5; do echo "== $t"; PA_LIB=$V/libpa_$t.so WINO=1 ONLY_S1=1 timeout 150 python tools/bench_conv.py 256 5 2>&1 | grep -v Warning; done
  echo "== v4 (default build)"; WINO=1 ONLY_S1=1 timeout 150 python tools/bench_conv.py 256 5 2>&1 | grep -v Warning ) > $O/conv.txt 2>&1
timeout 300 python -m pytest tests/test_emb_gpu.py tests/test_batch_gpu.py -q -x > $O/test_default.txt 2>&1
for t in v1 v3 v5; do PA_LIB=$V/libpa_$t.so timeout 120 python -m pytest tests/test_emb_gpu.py -q -x -k "winograd" > $O/test_$t.txt 2>&1; done
timeout 300 python bench.py --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err
PA_LIB=$V/libpa_head.so timeout 300 python bench.py --no-cpu-baseline > $O/bench_head.json 2> $O/bench_head.err
timeout 300 python bench.py --no-cpu-baseline --sequential --steps 3 > $O/bench_default_seq.json 2> $O/bench_default_seq.err
cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/c1_trace -- python $OLDPWD/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OLDPWD/$O/bench_trace.json 2> $OLDPWD/$O/trace.err; cd $OLDPWD
python tools/gap_report.py /tmp/c1_trace 0.3 > $O/gaps.txt 2>&1
tail -3 $O/test_*.txt; cat $O/conv.txt; cut -c1-400 $O/bench_default.json $O/bench_head.json $O/bench_default_seq.json; head -40 $O/gaps.txt
