#!/bin/bash
exec < /dev/null
cd /root/repo
mkdir -p gpurun_out
timeout 300 python tools/persist_prof.py 1000000 4,8 1,16,32 2>&1 | grep -v amdgpu.ids | tee gpurun_out/persist_prof.txt
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 | tee gpurun_out/gpu_tests.txt
