#!/bin/bash
exec < /dev/null
cd /root/repo
mkdir -p gpurun_out
timeout 300 python tools/persist_prof.py 1000000 2,4,8,16 1,8,16,32,64 2>&1 | grep -v amdgpu.ids | tee gpurun_out/persist_prof.txt
CCSIM_LEVEL_BATCH=16 CCSIM_SEQ_STEPS=4 timeout 900 python -m pytest tests/test_persist.py -m gpu -x -q 2>&1 | tail -5 | tee gpurun_out/quick_tests.txt
CCSIM_LEVEL_BATCH=5 CCSIM_SEQ_STEPS=1 timeout 900 python -m pytest tests/test_persist.py -m gpu -x -q -k "random or fast_path" 2>&1 | tail -5 | tee -a gpurun_out/quick_tests.txt
