#!/bin/bash
exec < /dev/null
cd /root/repo
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_multi.py -m gpu -x -q 2>&1 | tail -3
timeout 600 python tools/bench_c5.py 100000 1024 100000 1,16,64 2>&1 | grep -v amdgpu.ids | tee gpurun_out/bench_c5.txt
