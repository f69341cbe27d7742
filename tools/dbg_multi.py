import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import __graft_entry__ as ge; ge.load_package()
import numpy as np
import ccref_py
from cluster_capacity_amd import capi, model as M, synth
import test_multi as T
os.environ["CCSIM_MULTI_WINDOW"] = "1"
rng = np.random.default_rng(5000)
nodes, pods, prof = T.random_multi_case(rng, int(rng.integers(20, 700)), int(rng.integers(2, 80)))
limit = int(rng.choice([0, 0, 0, 150]))
ref = ccref_py.run_multi(prof, nodes, pods, max_limit=limit)
e = capi.Engine(device=0); e.load(nodes, pods, prof)
s0 = e.read_state()
g1 = e.run(max_limit=limit, log_cap=max(1, ref.placed)); print("run1", g1.placed, g1.stop, g1.stop_spec, g1.scans, g1.rounds)
e.reset_state()
s1 = e.read_state()
for k in s0: print(k, np.array_equal(s0[k], s1[k]))
g2 = e.run(max_limit=limit, log_cap=max(1, ref.placed)); print("run2", g2.placed, g2.stop, g2.stop_spec, g2.scans, g2.rounds, g2.hist[g2.hist > 0], np.nonzero(g2.hist)[0])
e.reset_state()
g3 = e.run(max_limit=limit, log_cap=max(1, ref.placed)); print("run3", g3.placed, g3.stop, g3.stop_spec, g3.scans, g3.rounds)
