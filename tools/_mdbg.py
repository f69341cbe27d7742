import sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/oracle"); sys.path.insert(0, "/root/repo/tests")
import __graft_entry__ as ge; ge.load_package()
import numpy as np, ccref_py
import helpers as H
from cluster_capacity_amd import cli, model as M
from test_multi import random_multi_case, _refused_specs
rng = np.random.default_rng(4400)
n = int(rng.integers(60, 400))
nodes, _, prof = random_multi_case(rng, n, 1)
pods = _refused_specs(rng, nodes, int(rng.integers(2, 7)), "soft")
limit = int(rng.choice([0, 0, 150]))
ref = ccref_py.run_multi(prof, nodes, pods, max_limit=limit)
got = cli.simulate_specs_one_cycle_at_a_time(nodes, pods, prof, limit)
d = next(i for i in range(min(len(got.log), len(ref.log))) if got.log[i] != ref.log[i])
P = len(pods)
print("first diff at", d, "spec", d % P, "got", got.log[d], "ref", ref.log[d], "P", P)
for j, q in enumerate(pods):
    print(j, [(k.col, k.hard, k.max_skew, k.self_match, k.node_match_count is not None, k.node_included is not None) for k in q.spread], q.ipa is not None, q.has_node_selector, bool(q.preferred))
