#!/bin/bash
# batched-mode parity subset + throughput (quick iteration on ccsim_level.h)
exec < /dev/null
cd /root/repo
mkdir -p gpurun_out
timeout 500 python -m pytest tests/test_gpu_parity.py tests/test_golden.py -m gpu -x -q -k "batched or golden or properties or sharded or eager or narrow" 2>&1 | tail -12 | tee gpurun_out/pytest_batched.log
timeout 200 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee gpurun_out/batched_speed.log
import time
import __graft_entry__ as ge; ge.load_package()
from cluster_capacity_amd import capi, synth
n,p,f = synth.make_config("C4", n_nodes=1_000_000)
e = capi.Engine(device=0); e.load(n,p,f)
for mode in ("batched",):
    ns,b = e.time_scan(300, mode=mode); print(mode, "full pass: %.2f us/launch" % (ns/300/1e3))
for rep in range(3):
    e.reset_state()
    t=time.perf_counter(); r=e.run(max_limit=0, mode="batched", want_log=False); dt=time.perf_counter()-t
    print("batched: %.3e placements/s (%.2f ms, %d placements, %d passes, kernel %.2f ms)" % (r.placed/dt, dt*1e3, r.placed, r.scans, r.kernel_ns/1e6))
PY
