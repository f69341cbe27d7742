#!/bin/bash
# round 2, validation call: the whole GPU suite, the persistent kernel's level-batch sweep, the bench line, the library-driven
# sharded run through RCCL at world size 1.  Everything lands in gpurun_out/.
cd "$(dirname "$0")/.." && export TMPDIR=/tmp && mkdir -p gpurun_out
timeout 560 python -m pytest tests -m gpu -q --maxfail=12 -rf --durations=8 2>&1 | tail -45 > gpurun_out/gpu_tests.txt
timeout 90 python tools/persist_prof.py 1000000 8 16,32,64,128 > gpurun_out/persist_sweep.txt 2>&1
timeout 150 python bench.py > gpurun_out/bench_1M.json 2> gpurun_out/bench_err.txt
CCSIM_FORCE_DIST=1 timeout 120 python bench.py --no-variants --no-cpu --seq-rounds 0 --steps 2 > gpurun_out/bench_dist1.json 2> gpurun_out/bench_dist1_err.txt
tail -3 gpurun_out/gpu_tests.txt; cat gpurun_out/persist_sweep.txt; cut -c1-400 gpurun_out/bench_1M.json; cut -c1-300 gpurun_out/bench_dist1.json; tail -3 gpurun_out/bench_dist1_err.txt
