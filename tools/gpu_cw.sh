#!/bin/bash
# windowed coupled mode: parity tests, throughput at 100k nodes with the deciding wave's phase profile, rocprofv3 kernel stats
exec < /dev/null
cd /root/repo
O=/root/repo/gpurun_out/r03
mkdir -p $O
timeout 600 python -m pytest tests/test_coupled.py -m gpu -q --timeout 120 -x 2>&1 | tail -15 | tee $O/cw_tests.txt
CCSIM_CW_PROF=1 timeout 300 python tools/bench_coupled.py 100000 50000 ${CW_SHAPES:-64,16 128,16 256,16} 2>&1 | grep -v amdgpu.ids | tee $O/bench_coupled.txt
cd /tmp && export TMPDIR=/tmp
rm -rf $O/ks
CCSIM_BENCH_SKIP_SEQ=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -o ks -- python /root/repo/tools/bench_coupled.py 100000 50000 ${CW_BEST:-128,16} > /dev/null 2> $O/ks.err
f=$(find $O/ks -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/cw_kernel_stats.csv && cut -c1-160 $O/cw_kernel_stats.csv | head -10
rm -rf $O/ks
