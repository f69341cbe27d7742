#!/bin/bash
exec < /dev/null
cd /tmp && export TMPDIR=/tmp
O=/root/repo/gpurun_out
rm -rf $O/c5prof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c5prof -o c5 -- python /root/repo/tools/bench_c5.py 100000 1024 100000 64 2>&1 | grep -v amdgpu.ids | tail -3
f=$(find $O/c5prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/c5_kernel_stats.csv && cut -c1-160 $O/c5_kernel_stats.csv | head -12
rm -rf $O/c5prof
