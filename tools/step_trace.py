#!/usr/bin/env python
"""A few bench.py steps (C4, batched, page-locked result array reused) for a timeline trace:
    rocprofv3 --hip-trace --kernel-trace --memory-copy-trace --output-format csv -d <dir> -- python tools/step_trace.py [nodes] [steps]
The host stamps (perf_counter_ns) of every step go to stdout so that the trace's API / kernel / copy records can be laid beside them."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge

ge.load_package()
from cluster_capacity_amd import capi, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
nodes, pod, prof = synth.make_config("C4", n_nodes=n)
e = capi.Engine(device=0)
e.load(nodes, pod, prof)
for _ in range(5):
    e.reset_state(); e.run(mode="batched", want_log=False, reuse_buffers=True)
ts = []
for _ in range(steps):
    t0 = time.perf_counter_ns()
    e.reset_state()
    t1 = time.perf_counter_ns()
    r = e.run(mode="batched", want_log=False, reuse_buffers=True)
    t2 = time.perf_counter_ns()
    ts.append((t1 - t0, t2 - t1, r.kernel_ns))
for a, b, k in ts:
    print(f"reset {a / 1e3:.1f} us | run {b / 1e3:.1f} us | kernel {k / 1e3:.1f} us")
e.close()
