import os, sys, dataclasses
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/oracle")
import __graft_entry__ as ge; ge.load_package()
import numpy as np, ccref_py
from cluster_capacity_amd import capi, synth
nodes, pod, prof = synth.make_config("C3", n_nodes=1000, seed=77 + 1000)
prof = dataclasses.replace(prof, percentage_of_nodes_to_score=0)
ref = ccref_py.run(prof, nodes, pod, max_limit=20)
print("ref", ref.log[:20], ref.evaluated_total)
e = capi.Engine(device=0); e.load(nodes, pod, prof)
try:
    got = e.run(max_limit=20, mode="sequential", log_cap=20)
    print("got", got.log[:20], got.evaluated_total, got.pass_launches)
except Exception as ex:
    print("ERR", ex)
