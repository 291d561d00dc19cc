#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=gpurun_out/c4; mkdir -p $O
V=pyannote-audio_amd/build/variants
PA_LIB=$V/libpa_stamp_prio.so timeout 200 python tools/wino_stamps.py $O/stamps_prio.npz > $O/stamps_prio.txt 2>&1
( for t in head nopf prio_nopf prio; do echo "== $t"; PA_LIB=$V/libpa_$t.so WINO=1 ONLY_S1=1 timeout 150 python tools/bench_conv.py 256 30 2>&1 | grep conv; done
  echo "== default"; WINO=1 ONLY_S1=1 timeout 150 python tools/bench_conv.py 256 30 2>&1 | grep conv ) > $O/conv.txt 2>&1
cat $O/stamps_prio.txt $O/conv.txt
