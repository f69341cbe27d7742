#!/usr/bin/env python
"""BASELINE config 5: 100k nodes x 1024 genpod-shaped pod specs (zone DoNotSchedule spread + hostname anti-affinity to
their own label), cycled round-robin.   python tools/bench_c5.py [nodes] [specs] [placements] [windows,...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import __graft_entry__ as ge

ge.load_package()
import numpy as np
from cluster_capacity_amd import capi, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
P = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
L = int(sys.argv[3]) if len(sys.argv) > 3 else 100_000
windows = [int(x) for x in sys.argv[4].split(",")] if len(sys.argv) > 4 else [64]
t0 = time.perf_counter()
nodes, pods, prof = synth.make_c5(n, P)
print(f"synth {time.perf_counter() - t0:.1f}s", flush=True)
import ccref_py

oracle_rounds = 300
t0 = time.perf_counter()
ref = ccref_py.run_multi(prof, nodes, pods, max_limit=oracle_rounds, threads=min(16, os.cpu_count() or 1))
dt = time.perf_counter() - t0
print(f"oracle (OpenMP x{min(16, os.cpu_count() or 1)}): {ref.placed / dt:.1f} placements/s ({oracle_rounds} cycles, {dt:.1f}s)", flush=True)
memo_modes = [None] if os.environ.get("CCSIM_MULTI_MEMO_MB") is not None else [None, "0"]  # default (score memo on), then the round-3 form
for w, memo in [(w, m) for w in windows for m in memo_modes]:
    os.environ["CCSIM_MULTI_WINDOW"] = str(w)
    if memo is not None:
        os.environ["CCSIM_MULTI_MEMO_MB"] = memo
    e = capi.Engine(device=0)
    t0 = time.perf_counter()
    e.load(nodes, pods, prof)
    t_load = time.perf_counter() - t0
    head = e.run(max_limit=oracle_rounds, log_cap=oracle_rounds)
    assert np.array_equal(head.log, ref.log), "engine and oracle placement logs differ"
    best = None
    for rep in range(3):
        e.reset_state()
        t0 = time.perf_counter()
        r = e.run(max_limit=L, want_log=False, log_cap=0)
        dt = time.perf_counter() - t0
        best = dt if best is None or dt < best else best
    mm = e.multi_memo()
    print(f"score memo {'on' if mm['on'] else 'OFF (CCSIM_MULTI_MEMO_MB=0: every scan computes)'}: {mm['bytes'] / 1e6:.0f} MB, pod-scans read from it {mm['memo_scans']}, computed {mm['full_scans']} | scan workgroup (0,0), us per scan: loads+staging, evaluation, merge = {mm.get('scan_us')}")
    print(f"window={w:3d}: {r.placed} placements in {best * 1e3:.1f} ms -> {r.placed / best:.3e} placements/s | windows {r.scans} "
          f"({r.placed / max(1, r.scans):.1f} pods/window, {r.pass_launches} ended early) kernel {r.kernel_ns / 1e6:.1f} ms "
          f"({r.kernel_ns / 1e3 / max(1, r.scans):.1f} us/window) | load+set_pods {t_load:.2f}s | stop={r.stop} spec={r.stop_spec} | stop reasons {e.multi_stops()}", flush=True)
    os.environ['CCSIM_MULTI_PROF'] = '1'
    pr = e.multi_stops()
    del os.environ['CCSIM_MULTI_PROF']
    print('   commit profile, us per window (assign+verify: load pods/cands, tables+min, assign, winner columns, verify, apply | in-order kernel adds to the same slots):', ['%.1f' % (x / 100.0 / max(1, r.scans)) for x in pr[:6]], flush=True)
    e.close()
    if memo is not None:
        del os.environ["CCSIM_MULTI_MEMO_MB"]
