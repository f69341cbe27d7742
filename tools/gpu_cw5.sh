#!/bin/bash
# round 5: the windowed coupled mode after the sweeps / the atomic-free class statistics: parity (coupled suite, four workers), throughput at
# 1M nodes / 64 zones and 100k / 16 zones, rocprofv3 kernel stats of the 1M run, the deciding wave's phase profile
exec < /dev/null
cd /root/repo
O=/root/repo/gpurun_out/${1:-r5c}
mkdir -p $O
if [ "$2" != "skip-tests" ]; then
timeout 900 python -m pytest tests/test_coupled.py tests/test_spread.py tests/test_ipa.py "tests/test_baseline_configs.py::test_coupled_template_at_baseline_sizes" "tests/test_baseline_configs.py::test_coupled_template_with_up_to_64_zones_takes_the_64_class_form" -m gpu -x -q -n 4 > $O/tests.txt 2>&1; tail -8 $O/tests.txt
fi
CCSIM_BENCH_SKIP_SEQ=1 timeout 300 python tools/bench_coupled.py 1000000 200000 4096,64 2>&1 | grep -v amdgpu.ids | tee $O/bench_coupled_1M.txt
CCSIM_BENCH_SKIP_SEQ=1 timeout 300 python tools/bench_coupled.py 100000 50000 4096,64 2>&1 | grep -v amdgpu.ids | tee $O/bench_coupled_100k.txt
CCSIM_BENCH_SKIP_SEQ=1 CCSIM_CW_PROF=1 timeout 300 python tools/bench_coupled.py 1000000 50000 4096,64 2>&1 | grep -v amdgpu.ids | tee $O/bench_coupled_1M_phase_profile.txt
cd /tmp && export TMPDIR=/tmp
rm -rf $O/ks
CCSIM_BENCH_SKIP_SEQ=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -o ks -- python /root/repo/tools/bench_coupled.py 1000000 50000 4096,64 > /dev/null 2> $O/ks.err
f=$(find $O/ks -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/cw_1M_64zones_kernel_stats.csv && cut -c1-170 $O/cw_1M_64zones_kernel_stats.csv | head -12
rm -rf $O/ks
