import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import __graft_entry__ as ge; ge.load_package()
import numpy as np
from cluster_capacity_amd import capi, model as M, synth
n, p, f = synth.make_config("C3", n_nodes=100_000)
n.label_cols.append(np.arange(1, n.n + 1, dtype=np.int32))
p.spread = [synth.zone_spread(n.n, max_skew=2)]
p.ipa = M.InterPodAffinity(key_cols=[2], key_ndom=[n.n], anti_keys=[0], anti_self=[True], anti_existing=[None])
e = capi.Engine(device=0); e.load(n, p, f)
e.run(max_limit=512, mode="sequential", want_log=False); e.reset_state()
t0 = time.perf_counter(); r = e.run(max_limit=2048, mode="sequential", want_log=False); dt = time.perf_counter() - t0
print("C5-shaped:", r.placed, "placements", r.scans, "passes", round(dt * 1e6 / r.scans, 2), "us/pass")
