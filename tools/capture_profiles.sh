#!/bin/sh
# Re-capture the judged measurement artefacts at HEAD on the GPU box (writes under gpurun_out/prof_$1).
# usage (via gpurun): sh tools/capture_profiles.sh r2
set -x
tag=${1:-r2}
out=gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
# (PA_WINOGRAD_GUARD=0: the load-time guard of EmbeddingPack launches every convolution kernel once on two 3-s chunks --
#  ~80 tiny launches that would dilute the per-launch averages of the kernels measured here)
export PA_WINOGRAD_GUARD=0
CMD="python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras --sequential"
sh tools/probes/build.sh > $out/probes_build.log 2>&1   # (tools/_dbg is git-ignored: build here if the caller did not)
python tools/probes/pingpong_probe.py > $out/mfma_probe.txt 2>&1
python tools/clock_trace.py > $out/clock_trace.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- $CMD > $out/bench_under_rocprof.json 2> $out/stats.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- $CMD > /dev/null 2> $out/fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- $CMD > /dev/null 2> $out/write.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $out/sq -- $CMD > /dev/null 2> $out/sq.err
python tools/pmc_traffic.py $out/fetch $out/write $out/traffic.json > /dev/null
python tools/pmc_sq.py $out/sq > $out/pipeline_pmc_sq.txt
find $out/stats -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats.csv \;
# keep only the summaries (the raw traces exceed the 64 MiB pull limit)
rm -rf $out/stats $out/fetch $out/write $out/sq
ls -la $out
