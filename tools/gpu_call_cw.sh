#!/bin/bash
exec < /dev/null
cd /root/repo
O=/root/repo/gpurun_out/r04c
mkdir -p $O
( timeout 400 python -m pytest tests/test_baseline_configs.py -k coupled -m gpu -q -x --timeout 300 2>&1 | grep -v amdgpu.ids | tail -6 ) | tee $O/gpu_cw_baseline_tests.txt
CCSIM_BENCH_SKIP_SEQ=1 timeout 200 python tools/bench_coupled.py 1000000 50000 2048,64 2>&1 | grep -v amdgpu.ids | tee $O/bench_coupled_1M_64zones.txt | cut -c1-300
CCSIM_BENCH_SKIP_SEQ=1 timeout 200 python tools/bench_coupled.py 100000 50000 2048,64 2>&1 | grep -v amdgpu.ids | tee $O/bench_coupled_100k.txt | cut -c1-300
( timeout 300 python -m pytest tests/test_coupled.py -m gpu -q -x --timeout 200 -n 4 2>&1 | grep -v amdgpu.ids | tail -6 ) | tee $O/gpu_cw_tests.txt
