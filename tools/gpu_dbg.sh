#!/bin/bash
exec < /dev/null
cd /root/repo
timeout 300 python tools/dbg_multi.py 2>&1 | grep -v amdgpu.ids | tail -20
