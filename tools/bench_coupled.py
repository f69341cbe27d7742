#!/usr/bin/env python
"""ONE template with topology-coupled plugins (BASELINE config 5's pod shape as a single template: DoNotSchedule zone spread +
required hostname anti-affinity against its own clones) on a synthetic C3-style snapshot: the windowed mode (csrc/ccsim_coupled.h)
against the one-pass-per-placement loop it replaces, both checked against the oracle's first placements.

    python tools/bench_coupled.py [nodes] [placements] [window,list ...]        (GPU box)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge

ge.load_package()
import numpy as np
from cluster_capacity_amd import capi, model as M, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
limit = int(sys.argv[2]) if len(sys.argv) > 2 else 50_000
shapes = [tuple(int(x) for x in a.split(",")) for a in sys.argv[3:]] or [(64, 16)]

nodes, pod, prof = synth.make_config("C3", n_nodes=n, seed=5)
nodes.label_cols.append(np.arange(1, n + 1, dtype=np.int32))  # kubernetes.io/hostname
pod.ipa = M.InterPodAffinity(key_cols=[len(nodes.label_cols) - 1], key_ndom=[n], anti_keys=[0], anti_self=[True], anti_existing=[None])
pod.spread = [synth.zone_spread(n, max_skew=1)]
zones = synth.zones_for(n)
if os.environ.get("CCSIM_BENCH_ZONES"):  # fewer zones than the synthetic cluster's (64 beyond 100k nodes): the same nodes, zones folded
    zones = int(os.environ["CCSIM_BENCH_ZONES"])
    col = pod.spread[0].col
    nodes.label_cols[col] = np.where(nodes.label_cols[col] > 0, (nodes.label_cols[col] - 1) % zones + 1, 0).astype(np.int32)
    pod.spread[0].n_domains = zones
import ccref_py

oracle_rounds = 200
t0 = time.perf_counter()
ref = ccref_py.run(prof, nodes, pod, max_limit=oracle_rounds, threads=min(16, os.cpu_count() or 1))
dt = time.perf_counter() - t0
print(f"{n} nodes, zone spread (maxSkew 1, {zones} zones) + hostname anti-affinity, percentageOfNodesToScore=100")
print(f"oracle (OpenMP x{min(16, os.cpu_count() or 1)}): {ref.placed / dt:.1f} placements/s ({oracle_rounds} cycles, {dt:.1f}s)", flush=True)


def bench(label, env):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    e = capi.Engine(device=0)
    e.load(nodes, pod, prof)
    head = e.run(max_limit=oracle_rounds, mode="sequential", log_cap=oracle_rounds)
    assert np.array_equal(head.log, ref.log), "engine and oracle placement logs differ"
    best = None
    for rep in range(3):
        e.reset_state()
        t0 = time.perf_counter()
        r = e.run(max_limit=limit, mode="sequential", want_log=False, log_cap=0)
        dt = time.perf_counter() - t0
        best = dt if best is None or dt < best else best
    info = e.coupled_info()
    print(f"{label}: {r.placed} placements in {best * 1e3:.1f} ms -> {r.placed / best:.3e} placements/s | passes {r.scans} "
          f"({r.placed / max(1, r.scans):.1f} placements/pass, {best * 1e6 / max(1, r.scans):.1f} us/pass) | {info}", flush=True)
    e.close()
    for k, v in old.items():
        os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)
    return r


full = None
for w, l in shapes:
    full = bench(f"windowed W={w:3d} L={l:2d}", {"CCSIM_CW_WINDOW": str(w), "CCSIM_CW_LIST": str(l)})
if os.environ.get("CCSIM_BENCH_SKIP_SEQ") != "1":
    limit = min(limit, 4000)
    bench("one pass per placement (CCSIM_CW=0)", {"CCSIM_CW": "0"})
