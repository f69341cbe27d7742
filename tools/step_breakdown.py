#!/usr/bin/env python
"""What one bench.py step costs around the persistent kernel (C4, 1M nodes): the kernel (HIP events), the whole step with the
engine's page-locked result array reused (what bench.py times), the same with fresh pageable arrays, the eager state restore for
comparison (CCSIM_EAGER_RESET=1), and the caller's own Python overhead (the same calls on a 512-node snapshot).
    python tools/step_breakdown.py [nodes]      (GPU box)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge

ge.load_package()
import numpy as np
from cluster_capacity_amd import build as B, capi, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000


def timed(f, reps=20):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        r = f()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    return ts[len(ts) // 2] * 1e3, r


def measure(n_nodes, label, narrow=0):
    nodes, pod, prof = synth.make_config("C4", n_nodes=n_nodes)
    e = capi.Engine(device=0)
    e.load(nodes, pod, prof)
    e.reset_state(); e.run(mode="batched", want_log=False, reuse_buffers=True, narrow_counts=narrow)
    t_pin, r1 = timed(lambda: (e.reset_state(), e.run(mode="batched", want_log=False, reuse_buffers=True, narrow_counts=narrow))[1])
    label += f" [per-node counts as {r1.per_node_count.dtype}]"
    t_run, r0 = timed(lambda: (e.reset_state(), e.run(mode="batched", want_log=False))[1], reps=5)
    assert np.array_equal(r0.per_node_count, r1.per_node_count) and r0.placed == r1.placed
    print(f"{label}: {n_nodes} nodes C4, batched: kernel {r1.kernel_ns / 1e6:.3f} ms | step (reset + run, page-locked result array reused) {t_pin:.3f} ms "
          f"= kernel + {t_pin - r1.kernel_ns / 1e6:.3f} | step with fresh pageable result arrays {t_run:.3f} ms | placed {r1.placed} passes {r1.scans}", flush=True)
    e.close()


print(f"libccsim sources {B.source_sha16()}")
measure(n, "step frame + one-byte counts (ABI 5 per_node_count_narrow)", narrow=1)
measure(n, "step frame + two-byte counts", narrow=2)
measure(n, "lazy restore (the launch loads the pristine columns)")
os.environ["CCSIM_FRAME_OFF"] = "1"
measure(n, "CCSIM_FRAME_OFF=1 (rounds 4-5: state copy + three fills before the launch, three small copies behind it)")
del os.environ["CCSIM_FRAME_OFF"]
os.environ["CCSIM_EAGER_RESET"] = "1"
measure(n, "eager restore (CCSIM_EAGER_RESET=1: device-to-device copy + mirror rebuild before the launch)")
del os.environ["CCSIM_EAGER_RESET"]
measure(512, "caller overhead (512 nodes: nothing to compute or copy)")
