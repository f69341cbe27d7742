#!/bin/bash
# per-kernel times of config 5 with the score memo on (the steady state), rocprofv3 --kernel-trace --stats
exec < /dev/null
cd /tmp && export TMPDIR=/tmp
O=/root/repo/gpurun_out
rm -rf $O/c5prof
CCSIM_MULTI_MEMO_MB=65536 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c5prof -o c5 -- python /root/repo/tools/bench_c5.py 100000 1024 200000 128 2>&1 | grep -v amdgpu.ids | grep "window=" | cut -c1-200
f=$(find $O/c5prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/c5_kernel_stats.csv && cut -c1-160 $O/c5_kernel_stats.csv | head -8
rm -rf $O/c5prof
