import sys, os, json, numpy as np, yaml, pathlib, tempfile, copy
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import conftest
from cluster_capacity_amd import cli, ingest, model as M, capi
import test_native_host as T
sys.path.insert(0,'/root/repo/oracle')
import ccref_py as ccref
nodes,pods,templates=T._templates_case(n_nodes=30)
templates[2]["spec"]["topologySpreadConstraints"].append({"maxSkew": 1, "topologyKey": "kubernetes.io/hostname", "whenUnsatisfiable": "ScheduleAnyway","labelSelector": {"matchLabels": {"app": "t2"}}})
tmp=pathlib.Path(tempfile.mkdtemp())
cluster,paths=T._write_templates(tmp,nodes,pods,templates)
snap=ingest.build_snapshot(nodes,pods,[cli.parse_pod_spec(q) for q in paths])
prof=M.Profile.default()
r=ccref.run_multi(prof,snap.nodes,snap.pods)
got=cli.simulate_specs_one_cycle_at_a_time(snap.nodes,snap.pods,prof,0)
d=next((i for i,(a,b) in enumerate(zip(r.log.tolist(),got.log.tolist())) if a!=b),None)
print(os.environ.get("TAG"), "first divergence", d, got.log.tolist()[25:34])
