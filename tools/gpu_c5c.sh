#!/bin/bash
exec < /dev/null
cd /root/repo
timeout 600 python tools/bench_c5.py 100000 1024 100000 64 2>&1 | grep -v amdgpu.ids | tail -4
