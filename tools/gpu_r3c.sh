#!/bin/bash
# round 3, after the second form of k_cw_decide_fast and the 384-level batches: parity of both, then the numbers
exec < /dev/null
cd /root/repo
O=/root/repo/gpurun_out/r03
mkdir -p $O
timeout 700 python -m pytest tests/test_coupled.py -m gpu -q --timeout 120 -x -n 4 2>&1 | tail -8 | tee $O/cw_tests.txt
timeout 300 python tools/bench_coupled.py 100000 50000 512,32 1024,32 2>&1 | grep -v amdgpu.ids | tee $O/bench_coupled.txt
CCSIM_BENCH_SKIP_SEQ=1 CCSIM_CW_PROF=1 timeout 300 python tools/bench_coupled.py 100000 50000 512,32 2>&1 | grep -v amdgpu.ids | tee $O/bench_coupled_prof.txt
timeout 900 python -m pytest tests/test_persist.py tests/test_gpu_parity.py tests/test_spread.py tests/test_ipa.py -m gpu -q --timeout 300 -x -n 4 2>&1 | tail -5 | tee $O/batch384_tests.txt
timeout 300 python bench.py 2> $O/bench_err.txt | tee $O/bench_1M.json
