# mid-round check on the GPU box: bench per Winograd F(4x4) layer selection, then the whole GPU suite
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=gpurun_out/r4_mid; mkdir -p $O
for w in "" 3 34 234; do
  PA_WINOGRAD4=$w timeout 200 python bench.py --steps 6 --no-cpu-baseline --no-extras > $O/bench_w4_$w.json 2>/dev/null
done
python - <<PY
import json
for w in ("", "3", "34", "234"):
    try:
        d = json.loads(open("$O/bench_w4_%s.json" % w).read().strip().splitlines()[-1])
        k = d["kernels"]
        print("PA_WINOGRAD4=%-4s" % w, d["value"], d["ms_per_step"], {n: round(k[n]["ms"], 1) for n in k if "wino" in n or n == "k_conv3x3"})
    except Exception as e:
        print(w, "ERR", e)
PY
PA_WINOGRAD4=234 timeout 300 python -m pytest tests/test_emb_gpu.py tests/test_golden.py -m gpu -q -x 2>&1 | tail -3
timeout 900 python -m pytest tests -m gpu -q > $O/gpu_tests.txt 2>&1; tail -5 $O/gpu_tests.txt
