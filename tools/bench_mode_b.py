#!/usr/bin/env python
"""Mode B (the reference's DEFAULT percentageOfNodesToScore: adaptive sampling, schedule_one.go:610-723) on one GPU: oracle gate, then
throughput of the form CCSIM_SB selects (1: a lap of the ring at a time, 2: a cycle at a time, 0: three node passes per cycle).
    python tools/bench_mode_b.py [n_nodes ...]"""
import dataclasses
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import __graft_entry__ as ge  # noqa: E402

ge.load_package()
import numpy as np  # noqa: E402
import ccref_py  # noqa: E402
from cluster_capacity_amd import capi, synth  # noqa: E402

sizes = [int(x) for x in sys.argv[1:]] or [1_000_000, 100_000]
GATE = int(os.environ.get("MB_GATE", "3000"))
LIM = int(os.environ.get("MB_LIMIT", "100000"))
for n in sizes:
    nodes, pod, prof = synth.make_config("C4", n_nodes=n)
    prof = dataclasses.replace(prof, percentage_of_nodes_to_score=int(os.environ.get("MB_PCT", "0")))
    ref = ccref_py.run(prof, nodes, pod, max_limit=GATE, threads=16)
    e = capi.Engine(device=0)
    e.load(nodes, pod, prof)
    head = e.run(max_limit=GATE, mode="sequential", log_cap=GATE)
    assert np.array_equal(head.log, ref.log) and head.evaluated_total == ref.evaluated_total, "engine and oracle differ"
    best = None
    for rep in range(3):
        e.reset_state()
        t0 = time.perf_counter()
        r = e.run(max_limit=LIM, mode="sequential", want_log=False, log_cap=0)
        dt = time.perf_counter() - t0
        best = dt if best is None or dt < best else best
    info = e.sampled_info()
    print(f"CCSIM_SB={os.environ.get('CCSIM_SB', '1')} {n} nodes, pct {prof.percentage_of_nodes_to_score}: {r.placed} cycles in {best * 1e3:.1f} ms -> {r.placed / best:.3e} placements/s, "
          f"{best * 1e6 / r.placed:.2f} us/cycle, {r.evaluated_total / r.placed:.0f} nodes visited per cycle, kernel {r.kernel_ns / 1e6:.1f} ms, {info}"
          + (f", {best * 1e6 / info['laps']:.2f} us/lap" if info["laps"] else ""), flush=True)
    e.close()
