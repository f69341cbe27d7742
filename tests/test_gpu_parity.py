"""Parity of the HIP path (through the C ABI) against the CPU oracle.  Bit-exact: integer work."""
import numpy as np
import pytest

import helpers as H
from cluster_capacity_amd import capi, model as M, report as R, synth

pytestmark = pytest.mark.gpu


def _engine(nodes, pod, prof, **kw):
    e = capi.Engine(device=0, **kw)
    e.load(nodes, pod, prof)
    return e


def _assert_same(got, ref, nodes, pod, check_log=True):
    assert got.placed == ref.placed
    assert got.stop == ref.stop
    assert np.array_equal(got.per_node_count, ref.per_node_count)
    if check_log and ref.log is not None and got.log is not None:
        assert np.array_equal(got.log, ref.log)
    if ref.stop == M.STOP_UNSCHEDULABLE:
        assert np.array_equal(got.hist, ref.hist)
        assert np.array_equal(got.hist_taintset[: len(ref.hist_taintset)], ref.hist_taintset)
        assert got.n_code_unschedulable == ref.n_code_unschedulable
        assert R.stop_reason(got, nodes.n, 0) == R.stop_reason(ref, nodes.n, 0)


@pytest.mark.parametrize("mode", ["sequential"])
def test_ka1_test_prediction(ccref, mode):
    nodes, pod, prof = H.test_prediction_nodes(), H.test_prediction_pod(), M.Profile.default()
    e = _engine(nodes, pod, prof)
    for limit in (0, 6):
        got = e.run(max_limit=limit, mode=mode)
        ref = ccref.run(prof, nodes, pod, max_limit=limit)
        _assert_same(got, ref, nodes, pod)
        e.load(nodes, pod, prof)
    got = _engine(nodes, pod, prof).run(mode=mode)
    assert R.stop_reason(got, 3, 0) == ("Unschedulable: 0/3 nodes are available: 1 Insufficient cpu, 3 Too many pods. "
                                       "preemption: 0/3 nodes are available: 3 No preemption victims found for incoming pod.")


@pytest.mark.parametrize("mode", ["sequential"])
def test_ka2_readme(ccref, mode):
    nodes, pod, prof = H.readme_nodes(4), H.examples_pod(), M.Profile.default()
    got = _engine(nodes, pod, prof).run(mode=mode)
    assert got.placed == 52 and got.per_node_count.tolist() == [13] * 4
    _assert_same(got, ccref.run(prof, nodes, pod), nodes, pod)


@pytest.mark.parametrize("mode", ["sequential"])
@pytest.mark.parametrize("cfg,n,limit", [("C2", 1000, 0), ("C3", 1000, 0), ("C3", 4096, 700), ("C2", 5000, 300),
                                          ("C3", 777, 0), ("C3", 1, 0), ("C3", 513, 50)])
def test_synthetic_vs_oracle(ccref, mode, cfg, n, limit):
    nodes, pod, prof = synth.make_config(cfg, n_nodes=n, seed=1234 + n)
    ref = ccref.run(prof, nodes, pod, max_limit=limit)
    e = _engine(nodes, pod, prof)
    got = e.run(max_limit=limit, mode=mode)
    _assert_same(got, ref, nodes, pod)
    # final dynamic state == initial + count * pod request (NodeInfo.update)
    st = e.read_state()
    cnt = got.per_node_count.astype(np.int64)
    assert np.array_equal(st["req_mcpu"], nodes.req[0] + cnt * int(pod.req[0]))
    assert np.array_equal(st["req_mem"], nodes.req[1] + cnt * int(pod.req[1]))
    assert np.array_equal(st["nz_mcpu"], nodes.nz_mcpu + cnt * pod.nz_mcpu)
    assert np.array_equal(st["pod_count"], nodes.pod_count + got.per_node_count)


def _random_case(rng, n):
    a_cpu = rng.choice([1000, 2000, 4000, 8000], n)
    a_mem = rng.choice([2, 4, 8, 16], n) * H.GiB
    a_eph = rng.choice([0, 10, 50], n) * H.GiB
    nodes = H.simple_nodes(a_cpu, a_mem, rng.integers(1, 12, n), req_mcpu=rng.integers(0, 900, n),
                           req_mem=rng.integers(0, 3, n) * H.GiB // 2, pod_count=rng.integers(0, 4, n), alloc_eph=a_eph,
                           taintset_id=rng.integers(0, 4, n), unschedulable=(rng.random(n) < 0.05),
                           label_cols=[rng.integers(0, 5, n), rng.integers(0, 3, n)])
    nodes.nz_mcpu = nodes.req[0] + rng.integers(0, 3, n) * 100  # existing pods without cpu requests
    nodes.nz_mem = nodes.req[1] + rng.integers(0, 2, n) * 200 * H.MiB
    t_in = lambda size, ids: np.isin(np.arange(size), ids).astype(np.uint8)
    pod = M.PodSpec(
        req=np.array([int(rng.choice([0, 100, 250, 500])), int(rng.choice([0, 256, 512])) * H.MiB, int(rng.choice([0, 0, 1])) * H.GiB]),
        nz_mcpu=0, nz_mem=0,
        taint_filter_ok=np.array([1, rng.integers(0, 2), 1, rng.integers(0, 2)], np.uint8),
        taint_prefer_cnt=np.array([0, 0, rng.integers(0, 3), rng.integers(0, 3)], np.int32),
        tolerates_unschedulable=bool(rng.integers(0, 2)),
        affinity_filter_active=bool(rng.integers(0, 2)),
        has_node_selector=bool(rng.integers(0, 2)), node_selector=[(1, t_in(3, [1, 2]))],
        has_required_terms=bool(rng.integers(0, 2)),
        required=[[(0, t_in(5, [1, 2, 3])), (1, t_in(3, [0, 1]))], [(0, t_in(5, [4]))], []],
        preferred=[(int(rng.integers(1, 100)), [(0, t_in(5, [2]))]), (int(rng.integers(1, 100)), [(1, t_in(3, [2])), (0, t_in(5, [0, 2, 4]))])]
        if rng.integers(0, 2) else [],
    )
    pod.nz_mcpu = int(pod.req[0]) or 100
    pod.nz_mem = int(pod.req[1]) or 200 * H.MiB
    prof = M.Profile(fit_res_w=(int(rng.integers(1, 4)), int(rng.integers(1, 4))),
                     w_taint=int(rng.integers(0, 4)), w_nodeaffinity=int(rng.integers(0, 3)), w_fit=int(rng.integers(0, 3)),
                     w_balanced=int(rng.integers(0, 2)))
    return nodes, pod, prof


@pytest.mark.parametrize("mode", ["sequential"])
@pytest.mark.parametrize("seed", range(12))
def test_random_plugin_mix_vs_oracle(ccref, mode, seed):
    rng = np.random.default_rng(seed)
    nodes, pod, prof = _random_case(rng, int(rng.integers(1, 1500)))
    limit = int(rng.choice([0, 0, 37, 500]))
    ref = ccref.run(prof, nodes, pod, max_limit=limit)
    got = _engine(nodes, pod, prof).run(max_limit=limit, mode=mode)
    _assert_same(got, ref, nodes, pod)


def test_schedule_one_matches_oracle_round_by_round(ccref):
    nodes, pod, prof = synth.make_config("C3", n_nodes=600, seed=5)
    ref = ccref.run(prof, nodes, pod, max_limit=200)
    e = _engine(nodes, pod, prof)
    for r in range(200):
        node, evaluated, feasible = e.schedule_one()
        assert node == ref.log[r]
        assert evaluated == nodes.n and feasible > 0


def test_schedule_one_fit_error_and_empty_snapshot(ccref):
    nodes, pod, prof = H.readme_nodes(2), H.examples_pod(), M.Profile.default()
    e = _engine(nodes, pod, prof)
    seen = [e.schedule_one()[0] for _ in range(27)]
    assert seen[:26].count(0) == 13 and seen[:26].count(1) == 13 and seen[26] == -1
    empty = H.simple_nodes([], [], [])
    got = _engine(empty, pod, prof).run()
    assert got.placed == 0 and got.stop == M.STOP_NO_NODES
    assert R.stop_reason(got, 0, 0) == "Unschedulable: no nodes available to schedule pods"


def test_eager_and_graph_launch_agree(ccref):
    nodes, pod, prof = synth.make_config("C3", n_nodes=3000, seed=9)
    a = _engine(nodes, pod, prof, use_graph=True, rounds_per_sync=64).run(max_limit=500)
    b = _engine(nodes, pod, prof, use_graph=False, rounds_per_sync=7).run(max_limit=500)
    assert np.array_equal(a.log, b.log) and a.placed == b.placed == 500


def test_full_size_properties_1m_nodes():
    """BASELINE full size: size-independent properties instead of the (too slow) oracle."""
    nodes, pod, prof = synth.make_config("C4", n_nodes=1_000_000)
    e = _engine(nodes, pod, prof)
    L = 1500
    got = e.run(max_limit=L, mode="sequential")
    assert got.placed == L and got.stop == M.STOP_LIMIT and int(got.per_node_count.sum()) == L
    assert np.array_equal(np.bincount(got.log, minlength=nodes.n).astype(np.int32), got.per_node_count)
    # every winner passed the static filters and never exceeds its Fit capacity
    w = np.unique(got.log)
    assert not nodes.unschedulable[w].any()
    free_cpu = nodes.alloc[0] - nodes.req[0]
    assert (got.per_node_count.astype(np.int64) * 150 <= free_cpu).all()
    assert (got.per_node_count + nodes.pod_count <= nodes.alloc_pods).all()
    # greedy property of round 1: the first winner maximizes the (static) total score, lowest index on ties
    st = e.read_state()
    assert np.array_equal(st["pod_count"], nodes.pod_count + got.per_node_count)
