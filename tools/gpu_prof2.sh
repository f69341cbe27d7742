#!/bin/bash
exec < /dev/null
cd /root/repo
mkdir -p gpurun_out
timeout 300 python tools/persist_prof.py 1000000 4,6,8,12 2>&1 | grep -v amdgpu.ids | tee gpurun_out/persist_prof.txt
