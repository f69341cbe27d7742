#!/bin/bash
# runs on the GPU box: GPU parity tests, bounded, no stdin reads
exec < /dev/null
mkdir -p /root/repo/gpurun_out
cd /root/repo
timeout ${1:-600} python -m pytest tests -m gpu -x -q 2>&1 | tail -${2:-15} > gpurun_out/pytest_gpu.log
cat gpurun_out/pytest_gpu.log
