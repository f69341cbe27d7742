#!/usr/bin/env python
"""ONE template with zone spread + hostname anti-affinity through the SHARDED protocol (ccsim_dist_run) on a one-rank RCCL communicator:
windows of placements per exchange (round 5, ccsim_dist_cw_*) against one exchange per placement (CCSIM_CW_SHARDS=0), both checked
against the oracle's first placements.

    python tools/bench_coupled_dist.py [nodes] [placements]        (GPU box)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge

ge.load_package()
import numpy as np
import ccref_py
from cluster_capacity_amd import capi, model as M, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
limit = int(sys.argv[2]) if len(sys.argv) > 2 else 50_000
nodes, pod, prof = synth.make_config("C3", n_nodes=n, seed=5)
nodes.label_cols.append(np.arange(1, n + 1, dtype=np.int32))
pod.ipa = M.InterPodAffinity(key_cols=[len(nodes.label_cols) - 1], key_ndom=[n], anti_keys=[0], anti_self=[True], anti_existing=[None])
pod.spread = [synth.zone_spread(n, max_skew=1)]
ref = ccref_py.run(prof, nodes, pod, max_limit=300, threads=min(16, os.cpu_count() or 1))
print(f"{n} nodes, zone spread (maxSkew 1, {synth.zones_for(n)} zones) + hostname anti-affinity, one-rank RCCL communicator")
for knob, lim in (("1", limit), ("0", min(limit, 3000))):
    os.environ["CCSIM_CW_SHARDS"] = knob
    e = capi.Engine(device=0, use_graph=False)
    e.load(nodes, pod, prof)
    e.dist_comm_init(capi.dist_unique_id(), 1, 0)
    e.dist_sync_tables()
    head = e.dist_run(300, "sequential", want_log=True, log_cap=300)
    assert np.array_equal(head.log, ref.log), "engine and oracle placement logs differ"
    best = None
    for rep in range(3):
        e.reset_state()
        t0 = time.perf_counter()
        r = e.dist_run(lim, "sequential", want_log=False, log_cap=0)
        dt = time.perf_counter() - t0
        best = dt if best is None or dt < best else best
    print(f"CCSIM_CW_SHARDS={knob}: {r.placed} placements in {best * 1e3:.1f} ms -> {r.placed / best:.3e} placements/s | exchanges {r.scans} "
          f"({best * 1e6 / max(1, r.scans):.1f} us each) | {e.coupled_info()}", flush=True)
    e.close()
