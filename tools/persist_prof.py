#!/usr/bin/env python
"""Where the persistent batched launch spends its time (C4, 1M nodes): kernel duration + workgroup 0's phase breakdown,
for a sweep of the run-down's lane-sequential step count.   python tools/persist_prof.py [nodes] [steps,steps,...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge

ge.load_package()
from cluster_capacity_amd import capi, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
sweep = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [6]
batches = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [1]
nodes, pod, prof = synth.make_config("C4", n_nodes=n)
for steps, batch in [(s_, b_) for b_ in batches for s_ in sweep]:
    os.environ["CCSIM_SEQ_STEPS"] = str(steps)
    os.environ["CCSIM_LEVEL_BATCH"] = str(batch)
    os.environ["CCSIM_PERSIST_PROF"] = "0"
    e = capi.Engine(device=0)
    e.load(nodes, pod, prof)
    for rep in range(3):
        e.reset_state()
        r0 = e.run(max_limit=0, mode="batched", want_log=False, log_cap=0)
    e.close()
    os.environ["CCSIM_PERSIST_PROF"] = "1"
    e = capi.Engine(device=0)
    e.load(nodes, pod, prof)
    for rep in range(2):
        e.reset_state()
        r = e.run(max_limit=0, mode="batched", want_log=False, log_cap=0)
    p = e.persist_prof()
    lv = max(1, p["levels"])
    print(f"batch={batch:3d} seq_steps={steps:3d} placed={r.placed} levels={p['levels']} passes={r.scans} kernel={r0.kernel_ns/1e6:.3f} ms (stamped run {r.kernel_ns/1e6:.3f} ms) "
          f"({r.kernel_ns/1e3/lv:.2f} us/level) | per level us: " +
          " ".join(f"{k}={v/lv:.2f}" for k, v in p.items() if k in ("scan_list", "plan", "apply", "block_reduce", "grid_reduce")) +
          f" | whole run us: load {p['load']:.1f}, rescore maxima {p['rescore_maxima']:.1f} + scores/prediction/reduce {p['rescore']:.1f}, write-back + diagnosis {p['write_back']:.1f}", flush=True)
    e.close()
