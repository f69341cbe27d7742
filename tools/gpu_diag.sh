#!/bin/bash
# diagnostic: the tests after the point where the full-suite run stopped, with the Python stack on a fatal signal
cd "$(dirname "$0")/.." && export TMPDIR=/tmp && mkdir -p gpurun_out
bash -c 'timeout 100 python -X faulthandler -m pytest tests/test_sampling.py -m gpu -q -rf -k "scalar_resources" 2>&1; echo "exit=$?"' > gpurun_out/diag_scalar.txt 2>&1
bash -c 'timeout 200 python -X faulthandler -m pytest tests/test_sampling.py tests/test_spread.py -m gpu -q -rf --deselect tests/test_sampling.py::test_sampled_search_scalar_resources -k "not vs_oracle and not random_plugin_mix and not topology_coupled" 2>&1; echo "exit=$?"' > gpurun_out/diag_rest.txt 2>&1
tail -30 gpurun_out/diag_scalar.txt; tail -12 gpurun_out/diag_rest.txt
