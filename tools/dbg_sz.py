import sys, os, dataclasses
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests'); sys.path.insert(0,'/root/repo/oracle')
import __graft_entry__ as ge; ge.load_package()
import numpy as np, ccref_py
from cluster_capacity_amd import capi
import test_sampling as T
for (n,z,pct,lim,skew,anti) in [(1000,None,0,8,2,False)]:
    nodes,pod,prof=T._zone_template(n,z,max_skew=skew,anti=anti,seed=40+n)
    prof=dataclasses.replace(prof,percentage_of_nodes_to_score=pct)
    for l in range(1,lim+1):
        ref=ccref_py.run(prof,nodes,pod,max_limit=l,threads=8)
        print("oracle cycle",l-1,"winner",ref.log[-1],"evaluated_total",ref.evaluated_total,"last_feasible",ref.last_feasible)
    e=capi.Engine(device=0); e.load(nodes,pod,prof)
    got=e.run(max_limit=lim,mode="sequential",log_cap=lim)
    print("got", got.log, got.evaluated_total)
    zc=np.asarray(nodes.label_cols[pod.spread[0].col]); print("zones of", [int(zc[i]) for i in ref.log])
    e.close()
