#!/bin/bash
# timing-only experiments on k_level_commit (results are NOT valid placements): per-variant kernel stats
exec < /dev/null
cd /root/repo
mkdir -p gpurun_out
for v in "" "-DCCSIM_EXP_NORUN" "-DCCSIM_EXP_NOWORK"; do
  CCSIM_EXTRA_FLAGS="$v" timeout 200 python cluster-capacity_amd/build.py > /dev/null 2>&1
  echo "== variant [$v]"
  timeout 100 python - <<'PY' 2>&1 | grep -v amdgpu.ids
import time
import __graft_entry__ as ge; ge.load_package()
from cluster_capacity_amd import capi, synth
n,p,f = synth.make_config("C4", n_nodes=1_000_000)
e = capi.Engine(device=0, rounds_per_sync=64); e.load(n,p,f)
e.run(max_limit=0, mode="batched", want_log=False)
# fixed number of passes on a fresh snapshot: 6 graph launches of 64 passes
for rep in range(2):
    e.reset_state()
    t=time.perf_counter(); r=e.run(max_limit=30_000_000, mode="batched", want_log=False); dt=time.perf_counter()-t
    print("  %d placements, %d passes, kernel %.2f ms -> %.2f us/pass" % (r.placed, r.scans, r.kernel_ns/1e6, r.kernel_ns/1e3/max(1,r.scans)))
PY
done
CCSIM_EXTRA_FLAGS="" timeout 200 python cluster-capacity_amd/build.py > /dev/null 2>&1
