#!/bin/bash
# round 6: the sampled search of a template with a hard zone spread constraint (k_sz_cycles): parity tests, throughput at 1M / 100k nodes against the three-pass cycle
exec < /dev/null
cd /root/repo
O=/root/repo/gpurun_out/${1:-r6z}
mkdir -p $O
[ -n "$SKIP_TESTS" ] || { timeout 1200 python -m pytest tests/test_sampling.py -m gpu -x -q -n 4 -k "zone or coupled" > $O/tests.txt 2>&1; tail -12 $O/tests.txt; }
for sz in 1 0; do
CCSIM_SZ=$sz CCSIM_SB_PROF=1 timeout 600 python tools/bench_mode_b_zone.py 2>&1 | grep -v amdgpu.ids | tee -a $O/bench_mode_b_zone.txt
done
