#!/bin/bash
# one GPU call: the multi-spec parity tests on the new library, then (only if green) the round's profile script
exec < /dev/null
cd /root/repo
mkdir -p gpurun_out/r04
( timeout 700 python -m pytest tests/test_multi.py tests/test_baseline_configs.py::test_c5_100k_nodes_1024_specs_prefix_vs_oracle tests/test_kernel_resources.py -m gpu -q -x --timeout 500 --durations=5 2>&1 | grep -v amdgpu.ids | tail -30 ) | tee gpurun_out/r04/gpu_multi_memo_tests.txt
if grep -q "failed\|error" gpurun_out/r04/gpu_multi_memo_tests.txt; then
  echo "multi-spec tests FAILED: bench only"
  timeout 300 python tools/bench_c5.py 100000 1024 200000 64 2>&1 | grep -v amdgpu.ids | cut -c1-300
  exit 1
fi
bash tools/gpu_round_profile.sh r04 skip-suite
