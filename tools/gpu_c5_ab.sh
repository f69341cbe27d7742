#!/bin/bash
# parity subsets touched by the scoring arithmetic, then config 5 throughput per build variant, then the persistent kernel
exec < /dev/null
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
timeout 250 python -m pytest tests/test_multi.py tests/test_golden.py tests/test_persist.py tests/test_gpu_parity.py -m gpu -q -rf -k "not continued_runs and not 1m_nodes and not sharded" 2>&1 | tail -6 | tee gpurun_out/c5_tests.txt
for v in "" $(ls cluster-capacity_amd/csrc/variants/ 2>/dev/null | sed 's/libccsim_//; s/\.so//'); do
  if [ -n "$v" ]; then export CCSIM_LIB=$PWD/cluster-capacity_amd/csrc/variants/libccsim_$v.so; else unset CCSIM_LIB; fi
  echo "== variant ${v:-default}"
  timeout 100 python tools/bench_c5.py 100000 1024 200000 64 2>&1 | grep -v "amdgpu.ids\|^synth\|^oracle" | cut -c1-330
done | tee gpurun_out/c5_ab.txt
unset CCSIM_LIB
timeout 60 python tools/persist_prof.py 1000000 8 64 2>&1 | grep -v amdgpu.ids | cut -c1-300 | tee gpurun_out/persist_now.txt
