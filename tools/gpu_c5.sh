#!/bin/bash
exec < /dev/null
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof_c5 -o c5 -- python /root/repo/tools/c5_shaped.py 2>&1 | grep "C5-shaped"
f=$(find /root/repo/gpurun_out/prof_c5 -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cut -c1-150 "$f" | head -8
