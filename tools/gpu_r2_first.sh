#!/bin/bash
# round 2, first GPU call: barrier microbench, the new BASELINE-size parity tests, the whole gpu suite, a bench line
exec < /dev/null
cd /root/repo
mkdir -p gpurun_out
timeout 60 tools/barrier_bench.bin 2000 2>&1 | tee gpurun_out/barrier_bench.txt
timeout 900 python -m pytest tests/test_baseline_configs.py -m gpu -x -q --durations=5 2>&1 | tail -15 | tee gpurun_out/baseline_tests.txt
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 | tee gpurun_out/gpu_tests.txt
timeout 300 python bench.py > gpurun_out/bench_r2_start.json 2> gpurun_out/bench_r2_start.err; tail -c 600 gpurun_out/bench_r2_start.json
