#!/usr/bin/env python
"""What one bench.py step costs around the persistent kernel (C4, 1M nodes): reset, run with fresh pageable result arrays, run with
the engine's page-locked result array reused.      python tools/step_breakdown.py [nodes]      (GPU box)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge

ge.load_package()
import numpy as np
import torch
from cluster_capacity_amd import capi, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
nodes, pod, prof = synth.make_config("C4", n_nodes=n)
e = capi.Engine(device=0)
e.load(nodes, pod, prof)


def timed(f, reps=5):
    best = None
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = f()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        best = dt if best is None or dt < best else best
    return best * 1e3, r


e.reset_state(); e.run(mode="batched", want_log=False)
t_reset, _ = timed(lambda: e.reset_state())
t_run, r0 = timed(lambda: (e.reset_state(), e.run(mode="batched", want_log=False))[1])
t_pin, r1 = timed(lambda: (e.reset_state(), e.run(mode="batched", want_log=False, reuse_buffers=True))[1])
assert np.array_equal(r0.per_node_count, r1.per_node_count) and r0.placed == r1.placed
print(f"{n} nodes C4, batched: kernel {r1.kernel_ns / 1e6:.3f} ms | reset_state alone {t_reset:.3f} ms | step with fresh pageable result arrays "
      f"{t_run:.3f} ms | step with the page-locked result array reused {t_pin:.3f} ms | placed {r1.placed}")
