#!/bin/bash
exec < /dev/null
cd /root/repo
O=/root/repo/gpurun_out/r03
mkdir -p $O
timeout 700 python -m pytest tests/test_coupled.py -m gpu -q --timeout 120 -x -n 4 2>&1 | tail -8 | tee $O/cw_tests.txt
CCSIM_BENCH_SKIP_SEQ=1 timeout 300 python tools/bench_coupled.py 100000 50000 512,32 1024,64 2>&1 | grep -v amdgpu.ids | tee $O/bench_coupled.txt
CCSIM_BENCH_SKIP_SEQ=1 CCSIM_CW_PROF=1 timeout 300 python tools/bench_coupled.py 100000 50000 512,32 2>&1 | grep -v amdgpu.ids | tee $O/bench_coupled_prof.txt
