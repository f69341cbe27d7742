#!/bin/bash
exec < /dev/null
cd /root/repo
timeout 600 python -m pytest tests/test_sampling.py tests/test_gpu_parity.py -m gpu -x -q -k "score_plugins or resource_lists or scalar_resources or profile_variants or sampled" 2>&1 | tail -15
