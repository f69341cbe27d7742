#!/bin/bash
# persistent batched mode: its own tests, then the whole gpu suite, then the bench line
exec < /dev/null
cd /root/repo
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_persist.py -m gpu -x -q --durations=5 2>&1 | tail -25 | tee gpurun_out/persist_tests.txt
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee gpurun_out/gpu_tests.txt
timeout 300 python bench.py --no-cpu > gpurun_out/bench_persist.json 2> gpurun_out/bench_persist.err; tail -c 1500 gpurun_out/bench_persist.json; tail -3 gpurun_out/bench_persist.err
