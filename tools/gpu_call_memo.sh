#!/bin/bash
# one GPU call: the multi-spec parity tests on the new library + config 5's throughput line
exec < /dev/null
cd /root/repo
O=/root/repo/gpurun_out/r04
mkdir -p $O
( timeout 700 python -m pytest tests/test_multi.py tests/test_baseline_configs.py::test_c5_100k_nodes_1024_specs_prefix_vs_oracle -m gpu -q -x --timeout 500 2>&1 | grep -v amdgpu.ids | tail -15 ) | tee $O/gpu_multi_memo_tests.txt
timeout 300 python tools/bench_c5.py 100000 1024 200000 64 2>&1 | grep -v amdgpu.ids | tee $O/bench_c5.txt | cut -c1-400
