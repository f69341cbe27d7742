#!/bin/bash
exec < /dev/null
cd /root/repo
timeout 40 python - <<'PY' 2>&1 | grep -v amdgpu.ids
import sys, time
sys.path.insert(0, "oracle")
import __graft_entry__ as ge; ge.load_package()
import numpy as np
from cluster_capacity_amd import capi, synth
import ccref_py
n,p,f = synth.make_config("C3", n_nodes=4096, seed=0xC0FFEE)
for lim in (0, 700):
    ref = ccref_py.run(f, n, p, max_limit=lim)
    e = capi.Engine(device=0); e.load(n,p,f)
    got = e.run(max_limit=lim, mode="batched", log_cap=max(1, ref.placed))
    print("parity", lim, got.placed == ref.placed and np.array_equal(got.per_node_count, ref.per_node_count) and np.array_equal(got.log, ref.log))
n,p,f = synth.make_config("C4", n_nodes=1_000_000)
e = capi.Engine(device=0); e.load(n,p,f)
for rep in range(3):
    e.reset_state(); r = e.run(max_limit=0, mode="batched", want_log=False)
    print("round-robin slots: kernel %.2f ms, %d placements, %d passes" % (r.kernel_ns/1e6, r.placed, r.scans))
PY
