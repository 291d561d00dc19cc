#!/bin/bash
# First gpurun call of the next round: times the three changes that were written and ISA-checked AFTER round 3's GPU
# budget was spent (all OFF by default) against the default build, in ONE call (~4 GPU-minutes).
#   1. libpa_refresh.so  -- 128-channel residual Winograd kernel with its residual loads pinned (-DPA_WINO_REFRESH=1)
#   2. PA_SEG_SHARED_SINC=1 -- the sinc layer once per span of overlapping chunks (gated GPU test first)
#   3. PA_LINKAGE_WGS / nothing new -- reference numbers of the default build for the same box
# usage: python tools/build_variants.py && gpurun --timeout 600 -- 'bash tools/next_round_first_call.sh'
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=gpurun_out/next_first; mkdir -p $O
V=pyannote-audio_amd/build/variants
echo "== 1. Winograd <2,2,true> with pinned residual loads (left: default, right: libpa_refresh.so)"
for i in 1 2; do
  ONLY_S1=1 WINO=1 timeout 120 python tools/bench_conv.py 512 20 > $O/conv_def_$i.txt 2>&1
  PA_LIB=$V/libpa_refresh.so ONLY_S1=1 WINO=1 timeout 120 python tools/bench_conv.py 512 20 > $O/conv_refresh_$i.txt 2>&1
  paste $O/conv_def_$i.txt $O/conv_refresh_$i.txt | grep "128->128" | cut -c1-46,100-146
done
PA_LIB=$V/libpa_refresh.so timeout 200 python -m pytest tests/test_emb_gpu.py -m gpu -q -x > $O/tests_refresh.txt 2>&1; tail -1 $O/tests_refresh.txt
echo "== 2. shared sinc layer"
PA_TEST_SHARED_SINC=1 timeout 200 python -m pytest tests/test_seg_gpu.py -m gpu -q -x -k shared_sinc > $O/tests_sinc.txt 2>&1; tail -3 $O/tests_sinc.txt
echo "== 3. whole pipeline: default | refresh | shared sinc | both"
timeout 100 python bench.py --steps 6 --no-cpu-baseline > $O/bench_default.json 2>/dev/null
PA_LIB=$V/libpa_refresh.so timeout 100 python bench.py --steps 6 --no-cpu-baseline > $O/bench_refresh.json 2>/dev/null
PA_SEG_SHARED_SINC=1 timeout 100 python bench.py --steps 6 --no-cpu-baseline > $O/bench_sinc.json 2>/dev/null
PA_SEG_SHARED_SINC=1 PA_LIB=$V/libpa_refresh.so timeout 100 python bench.py --steps 6 --no-cpu-baseline > $O/bench_both.json 2>/dev/null
python - <<PY
import json
for f in ("bench_default", "bench_refresh", "bench_sinc", "bench_both"):
    try:
        d = json.loads(open("$O/" + f + ".json").read().strip().splitlines()[-1])
        k = d["kernels"]
        print(f, d["value"], d["ms_per_step"], "seg stage", d["sequential_stages_ms"]["segmentation"], "wino",
              k["k_conv3x3_wino"]["ms"], {n: k[n]["ms"] for n in k if "sinc" in n})
    except Exception as e:
        print(f, "ERR", e)
PY
