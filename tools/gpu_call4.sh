#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=gpurun_out/c5; mkdir -p $O
V=pyannote-audio_amd/build/variants
( for t in head v0 v1 v2 v3 head v0; do echo "== $t"; PA_LIB=$V/libpa_$t.so WINO=1 ONLY_S1=1 timeout 150 python tools/bench_conv.py 256 30 2>&1 | grep conv; done ) > $O/conv.txt 2>&1
cat $O/conv.txt
