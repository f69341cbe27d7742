#!/bin/bash
# commit-pass grid sweep (CCSIM_LEVEL_GRID), speed only
exec < /dev/null
cd /root/repo
for g in "$@"; do
CCSIM_LEVEL_GRID=$g timeout 100 python - <<PY 2>&1 | grep -v amdgpu.ids
import time
import __graft_entry__ as ge; ge.load_package()
from cluster_capacity_amd import capi, synth
n,p,f = synth.make_config("C4", n_nodes=1_000_000)
e = capi.Engine(device=0); e.load(n,p,f)
best=None
for rep in range(3):
    e.reset_state()
    r=e.run(max_limit=0, mode="batched", want_log=False)
    best = r.kernel_ns if best is None or r.kernel_ns < best else best
print("grid $g: kernel %.2f ms, %d passes" % (best/1e6, r.scans))
PY
done
