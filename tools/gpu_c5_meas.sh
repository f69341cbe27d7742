#!/bin/bash
# config 5 measurement: throughput + the commit kernel's own phase profile, then rocprofv3 kernel stats; the library-driven
# sharded run (RCCL, one rank) as a bench line
exec < /dev/null
mkdir -p gpurun_out
timeout 120 python tools/bench_c5.py 100000 1024 200000 64 2>&1 | grep -v amdgpu.ids | tee gpurun_out/bench_c5.txt | cut -c1-420
bash tools/gpu_c5_prof.sh 2>&1 | tail -12
cd /root/repo
CCSIM_FORCE_DIST=1 timeout 120 python bench.py --no-variants --no-cpu --seq-rounds 0 --steps 2 > gpurun_out/bench_dist1.json 2> gpurun_out/bench_dist1_err.txt; cut -c1-330 gpurun_out/bench_dist1.json; tail -2 gpurun_out/bench_dist1_err.txt
