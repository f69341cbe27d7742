#!/bin/bash
exec < /dev/null
cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_multi.py -m gpu -x -q --durations=5 2>&1 | tail -40 | tee gpurun_out/multi_tests.txt
