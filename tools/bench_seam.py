#!/usr/bin/env python
"""The SchedulePod seam (ccsim_schedule_one: what a Go host calls once per pod, scheduler.go:88-91): microseconds per call, the launch and the
read-back of the run state included, on the resident block summaries (csrc/ccsim_search_full.h, ccsim_sampled.h) and on the node passes
(CCSIM_SF=0 / CCSIM_SB=0).   python tools/bench_seam.py [n_nodes ...]"""
import dataclasses
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

ge.load_package()
from cluster_capacity_amd import capi, synth  # noqa: E402

for n in [int(x) for x in sys.argv[1:]] or [1_000_000, 100_000]:
    for pct in (100, 0):
        nodes, pod, prof = synth.make_config("C4", n_nodes=n)
        prof = dataclasses.replace(prof, percentage_of_nodes_to_score=pct)
        e = capi.Engine(device=0)
        e.load(nodes, pod, prof)
        for _ in range(50):
            e.schedule_one()
        calls = int(os.environ.get("SEAM_CALLS", "3000"))
        t0 = time.perf_counter()
        for _ in range(calls):
            e.schedule_one()
        dt = time.perf_counter() - t0
        info = e.sampled_info()
        print(f"CCSIM_SF={os.environ.get('CCSIM_SF', '1')} CCSIM_SB={os.environ.get('CCSIM_SB', '1')} {n} nodes, percentageOfNodesToScore {pct}: {dt / calls * 1e6:.1f} us per ccsim_schedule_one call "
              f"({calls / dt:.3e} calls/s) | resident {info['resident']}, full search {info['full_search_form']}, laps {info['laps_form']}", flush=True)
        e.close()
