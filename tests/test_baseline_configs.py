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
