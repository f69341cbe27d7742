#!/bin/bash
# sampled-search parity tests + the full-pass timing check (no regression of the unsampled kernels)
exec < /dev/null
mkdir -p /root/repo/gpurun_out
cd /root/repo
timeout 400 python -m pytest tests/test_sampling.py -m gpu -x -q 2>&1 | tail -25 > gpurun_out/pytest_smp.log
cat gpurun_out/pytest_smp.log
timeout 120 python - <<'PY' 2>&1 | tee gpurun_out/timescan_smp.log
import time
import __graft_entry__ as ge; ge.load_package()
from cluster_capacity_amd import capi, synth
import dataclasses
n,p,f = synth.make_config("C4", n_nodes=1_000_000)
e = capi.Engine(device=0); e.load(n,p,f)
for mode in ("batched","sequential"):
    ns,b = e.time_scan(300, mode=mode); print(mode, "full pass: %.2f us/launch" % (ns/300/1e3))
e.run(max_limit=2048, mode="sequential", want_log=False); e.reset_state()
t=time.perf_counter(); r=e.run(max_limit=2048, mode="sequential", want_log=False); dt=time.perf_counter()-t
print("sequential 100%%: %.0f placements/s" % (r.placed/dt))
e2 = capi.Engine(device=0); e2.load(n,p,dataclasses.replace(f, percentage_of_nodes_to_score=0))
e2.run(max_limit=2048, mode="sequential", want_log=False); e2.reset_state()
t=time.perf_counter(); r=e2.run(max_limit=2048, mode="sequential", want_log=False); dt=time.perf_counter()-t
print("sequential adaptive (K=50000): %.0f placements/s, evaluated/cycle %.0f" % (r.placed/dt, r.evaluated_total/max(1,r.rounds)))
PY
