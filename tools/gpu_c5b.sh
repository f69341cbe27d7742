#!/bin/bash
exec < /dev/null
cd /root/repo
timeout 200 python -m pytest tests/test_multi.py -m gpu -x -q 2>&1 | tail -3
CCSIM_MULTI_CACHE=0 timeout 200 python -m pytest tests/test_multi.py -m gpu -x -q -k "c5 or random" 2>&1 | tail -2
bash tools/gpu_c5_prof.sh 2>&1 | tail -9 | head -5 | cut -c1-200
timeout 600 python tools/bench_c5.py 100000 1024 200000 64 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-400
