"""Oracle parity AT the BASELINE.json configuration sizes (C2 @ 10 000, C3 @ 100 000, C4 @ 1 000 000 nodes).

The oracle (oracle/ccref.c, OpenMP over nodes -- same results as one thread, tests/test_oracle_known_answers.py) is
the checker; the HIP engine is called through the C ABI in both modes.  What is compared is the placement LOG (the
canonical sequence), the per-node vector and, where the run ends, the FitError histogram.

  C2  whole run until Unschedulable (575 619 placements): log + vector + histogram, both modes;
  C3  first 2 000 cycles: log equality, both modes; + the closed-form exhaustive vector (order-independent plugin set:
      every node fills until NodeResourcesFit rejects it) for the batched mode's full run;
  C4  first 300 cycles: log equality, both modes; + the closed-form exhaustive vector.
"""
import os

import numpy as np
import pytest

from cluster_capacity_amd import capi, model as M, synth

pytestmark = pytest.mark.gpu
MODES = ["sequential", "batched"]
THREADS = min(16, os.cpu_count() or 1)


def _engine(nodes, pod, prof, **kw):
    e = capi.Engine(device=0, **kw)
    e.load(nodes, pod, prof)
    return e


def closed_form_capacity(nodes, pod, prof):
    """Pods every node takes before NodeResourcesFit rejects it (fit.go:564-615); statically infeasible nodes
    (Spec.Unschedulable, untolerated NoSchedule taints -- when the profile enables those filters) take none."""
    big = np.int64(1) << 40
    free_c = nodes.alloc[0] - nodes.req[0]
    free_m = nodes.alloc[1] - nodes.req[1]
    cap = np.minimum(free_c // int(pod.req[0]) if pod.req[0] else big, free_m // int(pod.req[1]) if pod.req[1] else big)
    cap = np.minimum(cap, (nodes.alloc_pods - nodes.pod_count).astype(np.int64))
    cap = np.maximum(cap, 0)
    static_ok = np.ones(nodes.n, bool)
    if prof.filter_mask & M.F_UNSCHEDULABLE:
        static_ok &= nodes.unschedulable == 0
    if prof.filter_mask & M.F_TAINT:
        static_ok &= np.asarray(pod.taint_filter_ok)[nodes.taintset_id] != 0
    return np.where(static_ok, cap, 0)


def test_c2_10k_whole_run_vs_oracle(ccref):
    nodes, pod, prof = synth.make_config("C2", n_nodes=10_000)
    ref = ccref.run(prof, nodes, pod, max_limit=0, threads=THREADS)
    assert ref.stop == M.STOP_UNSCHEDULABLE and ref.placed == int(closed_form_capacity(nodes, pod, prof).sum())
    e = _engine(nodes, pod, prof)
    for mode in MODES:
        e.reset_state()
        got = e.run(max_limit=0, mode=mode, log_cap=ref.placed)
        assert got.placed == ref.placed and got.stop == ref.stop, mode
        assert np.array_equal(got.per_node_count, ref.per_node_count), mode
        assert np.array_equal(got.log, ref.log), mode
        assert np.array_equal(got.hist, ref.hist) and got.n_code_unschedulable == ref.n_code_unschedulable, mode
    e.close()


@pytest.mark.parametrize("cfg,n,cycles", [("C3", 100_000, 2000), ("C4", 1_000_000, 300)])
def test_c3_c4_log_prefix_vs_oracle(ccref, cfg, n, cycles):
    nodes, pod, prof = synth.make_config(cfg, n_nodes=n)
    ref = ccref.run(prof, nodes, pod, max_limit=cycles, threads=THREADS)
    assert ref.placed == cycles and ref.stop == M.STOP_LIMIT
    e = _engine(nodes, pod, prof)
    for mode in MODES:
        e.reset_state()
        got = e.run(max_limit=cycles, mode=mode, log_cap=cycles)
        assert got.placed == cycles and got.stop == M.STOP_LIMIT, mode
        assert np.array_equal(got.log, ref.log), mode
        assert np.array_equal(got.per_node_count, ref.per_node_count), mode
    # the whole run (batched mode): order-independent plugin set -> the exhaustive vector has a closed form
    e.reset_state()
    full = e.run(max_limit=0, mode="batched", want_log=False)
    cap = closed_form_capacity(nodes, pod, prof)
    assert full.stop == M.STOP_UNSCHEDULABLE and full.placed == int(cap.sum())
    assert np.array_equal(full.per_node_count.astype(np.int64), cap)
    assert full.hist[M.R_UNSCHEDULABLE] == int(nodes.unschedulable.sum())
    e.close()


# ---- round 4: the BASELINE-size configurations that were only checked inside tools/ -------------------------------
# These are the sizes where the two-level class-list merges, > 64 classes, kCwMaxKeys and int32 table entries bite.


def _same_multi(got, ref):
    assert got.placed == ref.placed and got.stop == ref.stop and got.stop_spec == ref.stop_spec
    assert np.array_equal(got.log, ref.log)
    assert np.array_equal(got.per_node_count, ref.per_node_count)
    assert np.array_equal(got.per_spec_count, ref.per_spec_count)


def test_c5_100k_nodes_1024_specs_prefix_vs_oracle(ccref):
    """BASELINE configs[4] proper: 100 000 nodes x 1024 genpod-shaped specs (zone DoNotSchedule spread + hostname anti-affinity),
    cycled round-robin (ccref_run_multi: one reference scheduling cycle per pod, pkg/framework/simulator.go:297-381)."""
    nodes, pods, prof = synth.make_c5(100_000, 1024)
    # round 5: every spec at least twice against the ORACLE -- 2 200 cycles = two rounds of the 1024 specs and a bit,
    # >= 34 windows of <= 64 pods, the early-ended windows among them (the engine's own window-of-1 run below is a second opinion, not
    # the checker).  The oracle does ~450 cycles/s here on 16 threads.
    cycles = 2200
    ref = ccref.run_multi(prof, nodes, pods, max_limit=cycles, threads=THREADS)
    assert ref.placed == cycles and ref.stop == M.STOP_LIMIT
    assert int(np.asarray(ref.per_spec_count).min()) >= 2
    e = capi.Engine(device=0)
    e.load(nodes, pods, prof)
    head = e.run(max_limit=cycles, log_cap=cycles)
    _same_multi(head, ref)
    assert 34 <= head.scans < cycles // 8, head.scans  # (windows really were windows: <= 64 pods each, and far more than one pod on average)
    # a longer stretch (beyond one round of the 1024 specs): windows of 64 against the in-order window of 1, same engine semantics
    e.reset_state()
    long_w = e.run(max_limit=5000, log_cap=5000)
    e.close()
    os.environ["CCSIM_MULTI_WINDOW"] = "1"
    try:
        e1 = capi.Engine(device=0)
        e1.load(nodes, pods, prof)
        long_1 = e1.run(max_limit=5000, log_cap=5000)
        e1.close()
    finally:
        del os.environ["CCSIM_MULTI_WINDOW"]
    assert long_w.placed == long_1.placed == 5000
    assert np.array_equal(long_w.log, long_1.log) and np.array_equal(long_w.per_node_count, long_1.per_node_count)
    assert np.array_equal(long_w.log[:cycles], ref.log)


def _coupled_template(n, zones=None, seed=5):
    """BASELINE config 5's pod shape as ONE template: zone DoNotSchedule spread (maxSkew 1) + required hostname anti-affinity
    against its own clones, on the C3-style synthetic snapshot (tools/bench_coupled.py's workload)."""
    nodes, pod, prof = synth.make_config("C3", n_nodes=n, seed=seed)
    nodes.label_cols.append(np.arange(1, n + 1, dtype=np.int32))  # kubernetes.io/hostname
    pod.ipa = M.InterPodAffinity(key_cols=[len(nodes.label_cols) - 1], key_ndom=[n], anti_keys=[0], anti_self=[True], anti_existing=[None])
    pod.spread = [synth.zone_spread(n, max_skew=1)]
    if zones is not None:  # the same nodes dealt to `zones` zones (node i -> zone i mod zones, like the generator's own labelling)
        col = pod.spread[0].col
        nodes.label_cols[col] = (np.arange(n) % zones + 1).astype(np.int32)
        pod.spread[0].n_domains = zones
    return nodes, pod, prof


@pytest.mark.parametrize("n,zones", [(100_000, None), (1_000_000, None), (300_000, 16)], ids=["100k-16zones", "1M-64zones", "300k-16zones"])
def test_coupled_template_at_baseline_sizes(ccref, n, zones):
    """One template with topology-coupled plugins (podtopologyspread/filtering.go:235-356, interpodaffinity/filtering.go:204-432) in
    windows (csrc/ccsim_coupled.h): the oracle's first 200 cycles, and 5000 placements windowed == one pass per placement (CCSIM_CW=0).
    1M nodes with the generator's own 64 zones (synth.zones_for) is the shape with more classes than lanes."""
    nodes, pod, prof = _coupled_template(n, zones)
    # round 5: the ORACLE checks whole windows and their boundaries, not a tenth of one: 4 300 cycles are a full
    # window of the 64-class kernel (4096 cycles since round 5: 64 classes x 64 list members; two of round 4's 2048) plus the start of
    # the next (1M nodes / 64 zones: the oracle does ~90 cycles/s there on 16 threads, ~50 s), and at 16 zones two to three of the
    # ~1000-cycle windows the 16 class lists carry
    cycles = 4300 if zones is None else 2200
    ref = ccref.run(prof, nodes, pod, max_limit=cycles, threads=THREADS)
    assert ref.placed == cycles and ref.stop == M.STOP_LIMIT
    e = _engine(nodes, pod, prof)
    got = e.run(max_limit=cycles, mode="sequential", log_cap=cycles)
    assert got.placed == cycles and np.array_equal(got.log, ref.log) and np.array_equal(got.per_node_count, ref.per_node_count)
    head_info = e.coupled_info()
    assert head_info["windows"] >= 2 and not head_info["fell_back"], head_info  # (>= 1 window boundary inside the oracle-checked stretch)
    assert head_info["swept"] >= cycles - 64 * head_info["windows"], head_info  # (round 5: whole rounds at once -- what the oracle checked here IS the sweep path)
    e.reset_state()
    win = e.run(max_limit=5000, mode="sequential", log_cap=5000)
    info = e.coupled_info()
    assert info["plan"] and info["windows"] > 0 and not info["fell_back"], info
    # which decide kernel: the lane-per-candidate one up to 48 classes / 63 domains (16 zones), its 64-class form for the generator's own
    # 64 zones at 1M nodes (round 4; round 3 ran the general LDS kernel there: 1.5e5 placements/s)
    want = "full_windows" if synth.zones_for(n) == 64 and zones is None else "fast_windows"
    assert info[want] >= info["windows"] - 2, info
    e.close()
    os.environ["CCSIM_CW"] = "0"
    try:
        e0 = _engine(nodes, pod, prof)
        lit = e0.run(max_limit=5000, mode="sequential", log_cap=5000)
        assert not e0.coupled_info()["windows"]
        e0.close()
    finally:
        del os.environ["CCSIM_CW"]
    assert win.placed == lit.placed == 5000
    assert np.array_equal(win.log, lit.log) and np.array_equal(win.per_node_count, lit.per_node_count)
    assert np.array_equal(win.log[:min(cycles, 5000)], ref.log[:5000])


@pytest.mark.parametrize("n,zones,limit", [(20_000, 64, 0), (9_000, 57, 2500), (30_000, 64, 4000)])
def test_coupled_template_with_up_to_64_zones_takes_the_64_class_form(ccref, n, zones, limit):
    """More classes than the lane-per-candidate kernel's standard form takes (48) but no more than lanes (64): its FULL form (every lane a
    class, domains in lane value - 1, a winner that could still win ends the window) against the oracle: whole runs and limits."""
    nodes, pod, prof = _coupled_template(n, zones)
    ref = ccref.run(prof, nodes, pod, max_limit=limit, threads=THREADS)
    e = _engine(nodes, pod, prof)
    got = e.run(max_limit=limit, mode="sequential", log_cap=max(1, ref.placed))
    info = e.coupled_info()
    e.close()
    assert got.placed == ref.placed and got.stop == ref.stop
    assert np.array_equal(got.log, ref.log) and np.array_equal(got.per_node_count, ref.per_node_count)
    if ref.stop == M.STOP_UNSCHEDULABLE:
        assert np.array_equal(got.hist, ref.hist)
    assert info["full_windows"] >= info["windows"] - 2 and not info["fell_back"], info


def test_c3_100k_multi_kernel_form_vs_oracle(ccref):
    """C3 at 100 000 nodes through the multi-kernel form of the batched mode (CCSIM_PERSIST=0: what every sharded run, wide snapshot
    and > 1M-node GPU takes): log prefix, the blind batches cut by a limit inside a level, and the whole run's closed form."""
    nodes, pod, prof = synth.make_config("C3", n_nodes=100_000)
    cycles = 2000
    ref = ccref.run(prof, nodes, pod, max_limit=cycles, threads=THREADS)
    os.environ["CCSIM_PERSIST"] = "0"
    try:
        e = _engine(nodes, pod, prof)
    finally:
        del os.environ["CCSIM_PERSIST"]
    got = e.run(max_limit=cycles, mode="batched", log_cap=cycles)
    assert got.placed == cycles and np.array_equal(got.log, ref.log) and np.array_equal(got.per_node_count, ref.per_node_count)
    e.reset_state()
    blind = e.run(max_limit=cycles, mode="batched", want_log=False)
    assert blind.placed == cycles and np.array_equal(blind.per_node_count, ref.per_node_count)
    e.reset_state()
    full = e.run(max_limit=0, mode="batched", want_log=False)
    cap = closed_form_capacity(nodes, pod, prof)
    assert full.stop == M.STOP_UNSCHEDULABLE and full.placed == int(cap.sum())
    assert np.array_equal(full.per_node_count.astype(np.int64), cap)
    e.close()
