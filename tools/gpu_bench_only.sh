#!/bin/bash
exec < /dev/null
cd /root/repo
mkdir -p gpurun_out
timeout 400 python bench.py > gpurun_out/bench_try.json 2> gpurun_out/bench_try.err; tail -c 3000 gpurun_out/bench_try.json; tail -5 gpurun_out/bench_try.err
