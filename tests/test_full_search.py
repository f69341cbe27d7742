"""The full search (percentageOfNodesToScore = 100, every node filtered and scored each cycle: schedule_one.go:430-478) of a template
without topology-coupled plugins on resident block summaries (csrc/ccsim_search_full.h, k_sf_cycles): the form is taken, and it is the
oracle's simulation cycle by cycle -- through ccsim_run and through the SchedulePod seam (ccsim_schedule_one, scheduler.go:88-91)."""
import numpy as np
import pytest

import helpers as H
from cluster_capacity_amd import capi, model as M, report as R, synth

pytestmark = pytest.mark.gpu
THREADS = 16


def _engine(nodes, pod, prof, **kw):
    e = capi.Engine(device=0, **kw)
    e.load(nodes, pod, prof)
    return e


def _assert_same(got, ref, nodes):
    assert got.placed == ref.placed and got.stop == ref.stop
    assert np.array_equal(got.per_node_count, ref.per_node_count)
    assert np.array_equal(got.log, ref.log)
    assert got.evaluated_total == ref.evaluated_total
    if ref.stop == M.STOP_UNSCHEDULABLE:
        assert np.array_equal(got.hist, ref.hist) and got.n_code_unschedulable == ref.n_code_unschedulable
        assert R.stop_reason(got, nodes.n, 0) == R.stop_reason(ref, nodes.n, 0)


def _check_state(e, nodes, pod, cnt):
    st = e.read_state()
    cnt = cnt.astype(np.int64)
    assert np.array_equal(st["req_mcpu"], nodes.req[0] + cnt * int(pod.req[0])) and np.array_equal(st["req_mem"], nodes.req[1] + cnt * int(pod.req[1]))
    assert np.array_equal(st["nz_mcpu"], nodes.nz_mcpu + cnt * pod.nz_mcpu) and np.array_equal(st["pod_count"], nodes.pod_count + cnt)


@pytest.mark.parametrize("seed", range(14))
def test_random_plugin_mix_to_the_end(ccref, seed):
    # random taints / affinity terms / weights: nodes leave the feasible ones one by one, the normalization maxima move (rebuilds), the
    # run ends with the FitError histogram; pods with an ephemeral-storage request take the wide (int64) kernel
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.choice([1, 63, 700, 5000, 17000]))
    nodes, pod, prof = H.random_case(rng, n)
    limit = 0 if n <= 5000 else 6000
    ref = ccref.run(prof, nodes, pod, max_limit=limit, threads=THREADS)
    e = _engine(nodes, pod, prof)
    got = e.run(max_limit=limit, mode="sequential", log_cap=max(1, ref.placed))
    info = e.sampled_info()
    assert info["full_search_form"] and info["laps"] == got.placed, info
    _assert_same(got, ref, nodes)
    _check_state(e, nodes, pod, got.per_node_count)
    e.close()


@pytest.mark.parametrize("knob", [("CCSIM_SF_SHIFT", "10"), ("CCSIM_SB_CYCLES", "7"), ("CCSIM_SF", "0")], ids=["blocks-of-1024", "7-cycles-per-launch", "one-pass-per-cycle"])
def test_forms_agree(ccref, monkeypatch, knob):
    rng = np.random.default_rng(77)
    nodes, pod, prof = H.random_case(rng, 9000)
    ref = ccref.run(prof, nodes, pod, max_limit=4000, threads=THREADS)
    monkeypatch.setenv(*knob)
    e = _engine(nodes, pod, prof)
    got = e.run(max_limit=4000, mode="sequential", log_cap=4000)
    info = e.sampled_info()
    assert info["full_search_form"] == (knob[0] != "CCSIM_SF"), info
    if knob[0] == "CCSIM_SF_SHIFT":
        assert info["block"] == 1024
    if knob[0] == "CCSIM_SB_CYCLES":
        assert info["launches"] >= got.placed // 7
    _assert_same(got, ref, nodes)
    e.close()


@pytest.mark.parametrize("cfg,n,limit", [("C3", 100_000, 3000), ("C4", 300_000, 1500)])
def test_baseline_shapes_vs_oracle(ccref, cfg, n, limit):
    nodes, pod, prof = synth.make_config(cfg, n_nodes=n)
    ref = ccref.run(prof, nodes, pod, max_limit=limit, threads=THREADS)
    e = _engine(nodes, pod, prof)
    got = e.run(max_limit=limit, mode="sequential", log_cap=limit)
    assert e.sampled_info()["full_search_form"]
    _assert_same(got, ref, nodes)
    e.close()


def test_schedule_one_at_1m_nodes_matches_oracle_cycle_by_cycle(ccref):
    # the seam a Go host calls once per pod (scheduler.go:88-91): 2000 calls at BASELINE's full size, each one launch on the summaries
    nodes, pod, prof = synth.make_config("C4", n_nodes=1_000_000)
    cycles = 2000
    ref = ccref.run(prof, nodes, pod, max_limit=cycles + 500, threads=THREADS)  # (one oracle run: the calls' 2000 cycles, then 500 of a run)
    assert ref.placed == cycles + 500
    e = _engine(nodes, pod, prof)
    feasible0 = None
    for r in range(cycles):
        node, evaluated, feasible = e.schedule_one()
        assert node == ref.log[r], r
        assert evaluated == nodes.n and feasible > 0
        feasible0 = feasible if feasible0 is None else feasible0
        assert feasible <= feasible0
    info = e.sampled_info()
    assert info["full_search_form"] and info["laps"] == cycles, info
    _check_state(e, nodes, pod, np.bincount(ref.log[:cycles], minlength=nodes.n))
    # ... and a run on the columns as the calls left them continues the same simulation
    more = e.run(max_limit=500, mode="sequential", log_cap=500)
    assert np.array_equal(more.log, ref.log[cycles:])
    e.close()


def test_schedule_one_to_the_fit_error_and_beyond(ccref):
    nodes, pod, prof = H.readme_nodes(2), H.examples_pod(), M.Profile.default()
    e = _engine(nodes, pod, prof)
    seen = [e.schedule_one() for _ in range(29)]
    assert e.sampled_info()["full_search_form"]
    assert [s[0] for s in seen[:26]].count(0) == 13 and [s[0] for s in seen[:26]].count(1) == 13
    assert all(s[0] == -1 and s[2] == 0 for s in seen[26:])
    assert all(s[2] in (1, 2) for s in seen[:26]) and seen[0][2] == 2  # the feasible nodes the cycle scored
    e.close()


def test_schedule_one_under_the_sampled_search_uses_the_summaries_too(ccref, monkeypatch):
    # percentageOfNodesToScore below 100 at the seam: one cycle per launch of the lap kernel
    import dataclasses
    nodes, pod, prof = synth.make_config("C4", n_nodes=60_000)
    prof = dataclasses.replace(prof, percentage_of_nodes_to_score=10)
    cycles = 400
    ref = ccref.run(prof, nodes, pod, max_limit=cycles, threads=THREADS)
    e = _engine(nodes, pod, prof)
    ev = 0
    for r in range(cycles):
        node, evaluated, feasible = e.schedule_one()
        assert node == ref.log[r], r
        ev += evaluated
    assert ev == ref.evaluated_total
    info = e.sampled_info()
    assert info["resident"] and info["launches"] >= cycles, info
    e.close()
