#!/bin/bash
exec < /dev/null
cd /root/repo
mkdir -p gpurun_out
timeout 300 python tools/persist_prof.py 1000000 6,8 2>&1 | grep -v amdgpu.ids | tee gpurun_out/persist_prof.txt
timeout 900 python -m pytest tests/test_persist.py tests/test_gpu_parity.py tests/test_sampling.py tests/test_spread.py -m gpu -x -q --durations=8 2>&1 | tail -20 | tee gpurun_out/quick_tests.txt
