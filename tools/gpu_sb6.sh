#!/bin/bash
# round 6: the sampled search a lap of the ring at a time (k_sb_laps): parity (tests/test_sampling.py), throughput at 1M / 100k nodes against round 5's
# forms, phase profile; A/B of workgroup sizes when cluster-capacity_amd/csrc/variants/libccsim_lap*.so exist
exec < /dev/null
cd /root/repo
O=/root/repo/gpurun_out/${1:-r6a}
mkdir -p $O
[ -n "$SKIP_TESTS" ] || { timeout 1200 python -m pytest tests/test_sampling.py -m gpu -x -q -n 4 > $O/tests.txt 2>&1; tail -8 $O/tests.txt; }
for sb in 1 2; do
CCSIM_SB=$sb timeout 300 python tools/bench_mode_b.py 2>&1 | grep -v amdgpu.ids | tee -a $O/bench_mode_b.txt
done
CCSIM_SB_PROF=1 MB_LIMIT=50000 timeout 300 python tools/bench_mode_b.py 2>&1 | grep -v amdgpu.ids | tee -a $O/bench_mode_b_prof.txt
for v in cluster-capacity_amd/csrc/variants/libccsim_lap*.so; do
  [ -f "$v" ] || continue
  echo "== $v" | tee -a $O/bench_mode_b_variants.txt
  CCSIM_LIB=$PWD/$v CCSIM_SB_PROF=1 timeout 300 python tools/bench_mode_b.py 1000000 2>&1 | grep -v amdgpu.ids | tee -a $O/bench_mode_b_variants.txt
  [ -n "$SKIP_TESTS" ] || { CCSIM_LIB=$PWD/$v timeout 600 python -m pytest tests/test_sampling.py -m gpu -x -q -n 4 -k "not shards" > $O/tests_$(basename $v .so).txt 2>&1; tail -3 $O/tests_$(basename $v .so).txt; }
done
