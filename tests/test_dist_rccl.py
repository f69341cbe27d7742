"""The sharded run driven inside libccsim.so over the engine's own RCCL communicator (ccsim_dist_comm_init /
ccsim_dist_sync_tables / ccsim_dist_run, include/ccsim.h).  The GPU box has ONE GPU, so the communicator has one rank:
this pins the plumbing (dlopen of librccl, ncclCommInitRank, the all-gather on the engine's stream between the scan and
the decision, the poll loop, the table all-reduce) against the oracle; the protocol across several shards is pinned by
tests/test_gpu_parity.py (several engines on one GPU) and tests/test_dist_gloo.py (world 2-3 over gloo)."""
import numpy as np
import pytest

import helpers as H
from cluster_capacity_amd import capi, model as M, synth


def test_unique_id_without_a_gpu():
    a, b = capi.dist_unique_id(), capi.dist_unique_id()
    assert len(a) == len(b) == capi.DIST_ID_BYTES and a != b


def test_unloadable_rccl_is_an_error_not_a_crash():
    """with librccl not loadable the entry points return -EIO (the message built from ONE dlerror() call), they do not
    segfault.  A child process: the binding is cached per process."""
    import os
    import subprocess
    import sys

    code = ("import sys; sys.path.insert(0, %r); import __graft_entry__ as ge; ge.load_package()\n"
            "from cluster_capacity_amd import capi\n"
            "try:\n    capi.dist_unique_id()\nexcept capi.CcsimError as ex:\n    print('REFUSED', ex)\n") % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=H.SUBPROC_TIMEOUT,
                       env=dict(os.environ, CCSIM_RCCL_LIB="/nonexistent/librccl.so.1", CCSIM_DIST_DEBUG="1"))
    assert p.returncode == 0 and "REFUSED" in p.stdout, (p.returncode, p.stdout, p.stderr)
    assert "could not be loaded" in p.stderr and "/nonexistent/librccl.so.1" in p.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("mode,cfg,n,limit", [("sequential", "C3", 1500, 700), ("batched", "C3", 1500, 0), ("batched", "C3", 20_000, 12_345),
                                              ("batched", "C2", 3000, 0)])
def test_library_driven_run_world_1(ccref, mode, cfg, n, limit):
    nodes, pod, prof = synth.make_config(cfg, n_nodes=n, seed=500 + n)
    ref = ccref.run(prof, nodes, pod, max_limit=limit, threads=8)
    e = capi.Engine(device=0, use_graph=False)
    e.load(nodes, pod, prof)
    e.dist_comm_init(capi.dist_unique_id(), 1, 0)
    for rep in range(2):  # a second run on the restored state reuses the communicator
        e.reset_state()
        got = e.dist_run(limit, mode, want_log=True, log_cap=max(1, ref.placed))
        assert got.placed == ref.placed and got.stop == ref.stop
        assert np.array_equal(got.per_node_count, ref.per_node_count) and np.array_equal(got.log, ref.log)
        if ref.stop == M.STOP_UNSCHEDULABLE:
            assert np.array_equal(got.hist, ref.hist)
    e.close()


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(4))
def test_library_driven_run_with_replicated_tables(ccref, seed):
    """Hard spread constraints + inter-pod affinity: the count tables go through ccsim_dist_sync_tables (ncclAllReduce in place)."""
    rng = np.random.default_rng(8800 + seed)
    nodes, pod, prof = H.random_case(rng, int(rng.integers(50, 1500)))
    pod.spread = H.random_spread(rng, nodes, n_constraints=2)
    pod.ipa = H.random_ipa(rng, nodes)
    prof.filter_mask |= M.F_FIT
    ref = ccref.run(prof, nodes, pod, max_limit=400)
    e = capi.Engine(device=0, use_graph=False)
    e.load(nodes, pod, prof)
    e.dist_comm_init(capi.dist_unique_id(), 1, 0)
    assert e.dist_tables()
    e.dist_sync_tables()
    got = e.dist_run(400, "sequential", want_log=True, log_cap=max(1, ref.placed))
    assert got.placed == ref.placed and got.stop == ref.stop and np.array_equal(got.log, ref.log)
    if ref.stop == M.STOP_UNSCHEDULABLE:
        assert np.array_equal(got.hist, ref.hist)
    e.close()


@pytest.mark.gpu
@pytest.mark.parametrize("seed,pct", [(0, 0), (1, 30), (2, 10), (3, 70)])
def test_library_driven_sampled_search_of_a_coupled_template(ccref, seed, pct):
    """Round 6: percentageOfNodesToScore < 100 with hard spread constraints + inter-pod affinity through ccsim_dist_run (one RCCL rank: the
    counting and the scoring pass each end in an ncclAllGather, the tables went through ccsim_dist_sync_tables) == the oracle's visiting loop."""
    import dataclasses
    rng = np.random.default_rng(9100 + seed)
    nodes, pod, prof = H.random_case(rng, int(rng.integers(150, 1500)))
    pod.spread = H.random_spread(rng, nodes, n_constraints=2)
    pod.ipa = H.random_ipa(rng, nodes)
    prof.filter_mask |= M.F_FIT
    prof = dataclasses.replace(prof, percentage_of_nodes_to_score=pct)
    ref = ccref.run(prof, nodes, pod, max_limit=300)
    e = capi.Engine(device=0, use_graph=False)
    e.load(nodes, pod, prof)
    e.dist_comm_init(capi.dist_unique_id(), 1, 0)
    e.dist_sync_tables()
    got = e.dist_run(300, "sequential", want_log=True, log_cap=max(1, ref.placed))
    assert got.placed == ref.placed and got.stop == ref.stop and np.array_equal(got.log, ref.log)
    assert got.evaluated_total == ref.evaluated_total
    if ref.stop == M.STOP_UNSCHEDULABLE:
        assert np.array_equal(got.hist, ref.hist)
    e.close()


@pytest.mark.gpu
def test_dist_run_needs_a_communicator():
    nodes, pod, prof = synth.make_config("C3", n_nodes=600, seed=1)
    e = capi.Engine(device=0)
    e.load(nodes, pod, prof)
    with pytest.raises(capi.CcsimError, match="ccsim_dist_comm_init first"):
        e.dist_run(0, "batched")
    e.close()


@pytest.mark.gpu
@pytest.mark.parametrize("n,limit", [(3000, 0), (50_000, 20_000)])
def test_library_driven_windows_on_shards_world_1(ccref, monkeypatch, n, limit):
    """BASELINE config 5's pod shape as one template through ccsim_dist_run: the windows of placements per exchange (round 5,
    ccsim_dist_cw_*: the agreement all-reduce, per window an ncclAllGather of the window record on the engine's stream, the deciding
    wave) on a one-rank communicator == the oracle; with CCSIM_CW_SHARDS=0 the same run takes one exchange per placement."""
    from test_coupled import c5_single_template
    nodes, pod, prof = c5_single_template(n)
    ref = ccref.run(prof, nodes, pod, max_limit=limit, threads=8)
    scans = {}
    for knob in ("1", "0"):
        monkeypatch.setenv("CCSIM_CW_SHARDS", knob)
        e = capi.Engine(device=0, use_graph=False)
        e.load(nodes, pod, prof)
        e.dist_comm_init(capi.dist_unique_id(), 1, 0)
        e.dist_sync_tables()
        got = e.dist_run(limit, "sequential", want_log=True, log_cap=max(1, ref.placed))
        assert got.placed == ref.placed and got.stop == ref.stop and np.array_equal(got.log, ref.log), knob
        assert np.array_equal(got.per_node_count, ref.per_node_count), knob
        if ref.stop == M.STOP_UNSCHEDULABLE:
            assert np.array_equal(got.hist, ref.hist), knob
        scans[knob] = got.scans
        info = e.coupled_info()
        assert (info["windows"] > 0) == (knob == "1"), (knob, info)
        e.close()
    assert scans["1"] * 2 < scans["0"], scans  # (passes: windows against one or two per placement; with 3 zones the late phase of a run moves the maxima every cycle)
