#!/bin/bash
# windowed coupled mode: parity tests (twice, four workers), throughput at 100k nodes, the deciding wave's phase profile,
# rocprofv3 kernel stats
exec < /dev/null
cd /root/repo
O=/root/repo/gpurun_out/r03
mkdir -p $O
python -c "import sys; sys.path.insert(0, '.'); from importlib import import_module as im; b = im('cluster-capacity_amd.build'.replace('-', '_')) if False else None" 2>/dev/null
for rep in 1 2; do timeout 600 python -m pytest tests/test_coupled.py -m gpu -q --timeout 120 -x -n 4 2>&1 | tail -6; done | tee $O/cw_tests.txt
timeout 300 python tools/bench_coupled.py 100000 50000 ${CW_SHAPES:-1024,64 512,32 256,16} 2>&1 | grep -v amdgpu.ids | tee $O/bench_coupled.txt
CCSIM_BENCH_SKIP_SEQ=1 CCSIM_CW_PROF=1 timeout 300 python tools/bench_coupled.py 100000 50000 1024,64 2>&1 | grep -v amdgpu.ids | tee $O/bench_coupled_phase_profile.txt
cd /tmp && export TMPDIR=/tmp
rm -rf $O/ks
CCSIM_BENCH_SKIP_SEQ=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -o ks -- python /root/repo/tools/bench_coupled.py 100000 50000 1024,64 > /dev/null 2> $O/ks.err
f=$(find $O/ks -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/cw_kernel_stats.csv && cut -c1-160 $O/cw_kernel_stats.csv | head -10
rm -rf $O/ks
