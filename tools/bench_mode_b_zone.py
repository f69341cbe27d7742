#!/usr/bin/env python
"""The reference's default percentageOfNodesToScore for ONE template with topology-coupled plugins (BASELINE config 5's pod shape: zone DoNotSchedule
spread + hostname anti-affinity) on one GPU: oracle gate, then throughput of the form CCSIM_SZ selects (1: per-(block, zone) entries under the mask of
eligible zones, csrc/ccsim_sampled_zone.h; 0: three node passes per cycle).    python tools/bench_mode_b_zone.py [n_nodes ...]"""
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
from cluster_capacity_amd import capi, model as M, synth  # noqa: E402

sizes = [int(x) for x in sys.argv[1:]] or [1_000_000, 100_000]
GATE = int(os.environ.get("MB_GATE", "700"))
LIM = int(os.environ.get("MB_LIMIT", "20000"))
for n in sizes:
    nodes, pod, prof = synth.make_config("C3", n_nodes=n, seed=5)
    nodes.label_cols.append(np.arange(1, n + 1, dtype=np.int32))  # kubernetes.io/hostname
    pod.ipa = M.InterPodAffinity(key_cols=[len(nodes.label_cols) - 1], key_ndom=[n], anti_keys=[0], anti_self=[True], anti_existing=[None])
    pod.spread = [synth.zone_spread(n, max_skew=1)]
    prof = dataclasses.replace(prof, percentage_of_nodes_to_score=int(os.environ.get("MB_PCT", "0")))
    ref = ccref_py.run(prof, nodes, pod, max_limit=GATE, threads=16)
    e = capi.Engine(device=0)
    e.load(nodes, pod, prof)
    head = e.run(max_limit=GATE, mode="sequential", log_cap=GATE)
    assert np.array_equal(head.log, ref.log) and head.evaluated_total == ref.evaluated_total, "engine and oracle differ"
    lim = LIM if os.environ.get("CCSIM_SZ", "1") != "0" else min(LIM, 3000)
    best = None
    for rep in range(2):
        e.reset_state()
        t0 = time.perf_counter()
        r = e.run(max_limit=lim, mode="sequential", want_log=False, log_cap=0)
        dt = time.perf_counter() - t0
        best = dt if best is None or dt < best else best
    info = e.sampled_info()
    print(f"CCSIM_SZ={os.environ.get('CCSIM_SZ', '1')} {n} nodes, {synth.zones_for(n)} zones, pct {prof.percentage_of_nodes_to_score}: {r.placed} cycles in {best * 1e3:.1f} ms -> "
          f"{r.placed / best:.3e} placements/s, {best * 1e6 / r.placed:.2f} us/cycle, {r.evaluated_total / r.placed:.0f} nodes visited per cycle, kernel {r.kernel_ns / 1e6:.1f} ms, {info}", flush=True)
    e.close()
