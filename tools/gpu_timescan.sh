#!/bin/bash
# usage: tools/gpu_timescan.sh "<extra hipcc flags>" : rebuild with flags, time a train of the batched full-pass kernel
exec < /dev/null
cd /root/repo
CCSIM_EXTRA_FLAGS="$1" timeout 200 python cluster-capacity_amd/build.py > /dev/null 2>&1
timeout 120 python - <<'PY'
import __graft_entry__ as ge; ge.load_package()
from cluster_capacity_amd import capi, synth
n,p,f = synth.make_config("C4", n_nodes=1_000_000)
e = capi.Engine(device=0); e.load(n,p,f)
for mode in ("batched","sequential"):
    ns,b = e.time_scan(300, mode=mode); print(mode, "full pass: %.2f us/launch" % (ns/300/1e3))
PY
