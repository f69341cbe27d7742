#!/bin/bash
# usage: tools/gpu_prof.sh <tag> <bench args...>   (runs on the GPU box; everything bounded, no stdin reads)
tag=$1; shift
exec < /dev/null
mkdir -p /root/repo/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof_$tag -o $tag -- python /root/repo/bench.py "$@" > /root/repo/gpurun_out/bench_$tag.json 2> /root/repo/gpurun_out/bench_$tag.err
cd /root/repo
cut -c1-400 gpurun_out/bench_$tag.json
f=$(find gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then cut -c1-160 "$f" | head -12; else echo "no kernel_stats csv"; ls -R gpurun_out/prof_$tag | head; fi
