#!/bin/bash
exec < /dev/null
cd /root/repo
mkdir -p gpurun_out/r02
timeout 1500 python -m pytest tests -m gpu -x -q --durations=8 2>&1 | tail -16 | tee gpurun_out/r02/gpu_tests.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v amdgpu.ids | tail -4 | tee gpurun_out/r02/smoke.txt
timeout 600 python tools/bench_configs.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02/bench_configs.md
