#!/bin/sh
# HBM traffic of the convolution kernels on the layer shapes (development aid): FETCH_SIZE / WRITE_SIZE passes of
# tools/bench_conv.py under rocprofv3, summarised by tools/pmc_traffic.py.   usage: sh tools/conv_traffic.sh <tag> [env...]
tag=$1; shift
out=gpurun_out/traffic_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
CMD="python tools/bench_conv.py 512 3"
env "$@" WARM=1 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- $CMD > /dev/null 2> $out/fetch.err
env "$@" WARM=1 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- $CMD > /dev/null 2> $out/write.err
python tools/pmc_traffic.py $out/fetch $out/write $out/traffic.json | python -c "
import json,sys
d=json.load(open('$out/traffic.json'))
for k,v in d.items(): print('$tag', k, v['launches'], 'GB/launch', round(v['hbm_bytes_per_launch']/1e9,3), 'fetch raw KiB', round(v['fetch_size_kib_per_launch_raw']))
"
rm -rf $out/fetch $out/write
