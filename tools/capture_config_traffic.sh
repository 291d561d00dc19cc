#!/bin/sh
# PMC traffic (FETCH_SIZE / WRITE_SIZE, separate passes) of the stage configurations of bench.py -- BASELINE.json
# configs[1] (seg5s) and configs[2] (emb3s) -- whose kernels run on other shapes than the pipeline's: writes
# gpurun_out/prof_$1/traffic_<config>.json (copy to profiles/r5_traffic_<config>.json, add "_commit").
# usage (via gpurun): sh tools/capture_config_traffic.sh r5
set -x
tag=${1:-r5}
out=gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
export PA_WINOGRAD_GUARD=0      # (the load-time guard's tiny launches would dilute the per-launch averages)
for cfg in emb3s seg5s; do
  CMD="python bench.py --config $cfg --steps 1 --warmup 0 --no-cpu-baseline"
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch_$cfg -- $CMD > /dev/null 2> $out/fetch_$cfg.err
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write_$cfg -- $CMD > $out/bench_$cfg.json 2> $out/write_$cfg.err
  python tools/pmc_traffic.py $out/fetch_$cfg $out/write_$cfg $out/traffic_$cfg.json > /dev/null
  rm -rf $out/fetch_$cfg $out/write_$cfg
done
ls -la $out
