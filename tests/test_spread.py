"""PodTopologySpread hard constraints (P/podtopologyspread/filtering.go:235-356).
CPU: hand-derived known answers pin the oracle's restatement.  GPU: the HIP filter (count tables in HBM,
assume-and-verify global minimum) against the oracle, sequential mode."""
import numpy as np
import pytest

import helpers as H
from cluster_capacity_amd import capi, model as M, report as R, synth

DEFAULT = M.Profile.default()


def _zone_nodes(zone_ids, pods):
    n = len(zone_ids)
    return H.simple_nodes([10000] * n, [64 * H.GiB] * n, pods, label_cols=[np.array(zone_ids, np.int32)])


def _pod(**kw):
    p = H.simple_pod(100, 64 * H.MiB)
    p.spread = [M.SpreadConstraint(col=0, hard=True, **kw)]
    return p


def test_ka_spread_max_skew_1(ccref):
    # 3 zones x 1 node; node 0 holds 5 pods at most.  skew = count + 1 - min <= 1  =>  only minimum domains grow:
    # 5/5/5, then zone 0 is full but still a domain (min stays 5): the others stop at 6.  17 = 5 + 6 + 6.
    nodes = _zone_nodes([1, 2, 3], [5, 110, 110])
    r = ccref.run(DEFAULT, nodes, _pod(max_skew=1, n_domains=3, self_match=True))
    assert r.placed == 17 and r.per_node_count.tolist() == [5, 6, 6]
    assert R.stop_reason(r, 3, 0).startswith(
        "Unschedulable: 0/3 nodes are available: 1 Too many pods, 2 node(s) didn't match pod topology spread constraints.")


def test_ka_spread_min_domains(ccref):
    # 2 domains < minDomains 3  =>  the global minimum counts as 0 (filtering.go:56-69): each zone holds maxSkew = 2 pods
    nodes = _zone_nodes([1, 2], [110, 110])
    r = ccref.run(DEFAULT, nodes, _pod(max_skew=2, min_domains=3, n_domains=2, self_match=True))
    assert r.placed == 4 and r.per_node_count.tolist() == [2, 2]


def test_ka_spread_missing_label_and_no_self_match(ccref):
    # node 2 lacks the topology key: never feasible (UnschedulableAndUnresolvable).  Without self-match the counts
    # never move, so the constraint never binds and capacity (pods) is the limit.
    nodes = _zone_nodes([1, 2, 0], [3, 4, 50])
    r = ccref.run(DEFAULT, nodes, _pod(max_skew=1, n_domains=2, self_match=False))
    assert r.placed == 7 and r.per_node_count.tolist() == [3, 4, 0]
    msg = R.stop_reason(r, 3, 0)
    assert "1 node(s) didn't match pod topology spread constraints (missing required label)" in msg and "2 Too many pods" in msg


def test_ka_spread_existing_pods_count(ccref):
    # zone 1 already runs 2 matching pods, zone 2 none: zone 2 must catch up first (2 pods), then they alternate
    nodes = _zone_nodes([1, 2], [110, 110])
    pod = _pod(max_skew=1, n_domains=2, self_match=True)
    pod.spread[0].node_match_count = np.array([2, 0], np.int32)
    r = ccref.run(DEFAULT, nodes, pod, max_limit=6)
    assert r.log.tolist()[:2] == [1, 1] and sorted(r.log.tolist()[2:4]) == [0, 1] and r.per_node_count.tolist() == [2, 4]


def _gpu_check(ccref, nodes, pod, prof, limit):
    ref = ccref.run(prof, nodes, pod, max_limit=limit)
    e = capi.Engine(device=0)
    e.load(nodes, pod, prof)
    got = e.run(max_limit=limit, mode="sequential", log_cap=max(1, ref.placed))
    assert got.placed == ref.placed and got.stop == ref.stop
    assert np.array_equal(got.per_node_count, ref.per_node_count)
    assert np.array_equal(got.log, ref.log)
    if ref.stop == M.STOP_UNSCHEDULABLE:
        assert np.array_equal(got.hist, ref.hist)
        assert got.n_code_unschedulable == ref.n_code_unschedulable
        assert R.stop_reason(got, nodes.n, 0) == R.stop_reason(ref, nodes.n, 0)
    return e, got


@pytest.mark.gpu
def test_gpu_spread_known_answers(ccref):
    _gpu_check(ccref, _zone_nodes([1, 2, 3], [5, 110, 110]), _pod(max_skew=1, n_domains=3, self_match=True), DEFAULT, 0)
    _gpu_check(ccref, _zone_nodes([1, 2], [110, 110]), _pod(max_skew=2, min_domains=3, n_domains=2, self_match=True), DEFAULT, 0)
    _gpu_check(ccref, _zone_nodes([1, 2, 0], [3, 4, 50]), _pod(max_skew=1, n_domains=2, self_match=False), DEFAULT, 0)


@pytest.mark.gpu
@pytest.mark.parametrize("n,skew,limit", [(600, 1, 0), (2000, 2, 900), (3000, 5, 1500)])
def test_gpu_spread_synthetic_zones(ccref, n, skew, limit):
    nodes, pod, prof = synth.make_config("C3", n_nodes=n, seed=21 + n)
    pod.spread = [synth.zone_spread(n, max_skew=skew)]
    e, got = _gpu_check(ccref, nodes, pod, prof, limit)
    # the invariant the constraint enforces, on the final placement: zone counts within maxSkew of each other
    # while every zone still had a feasible node (checked on the prefix where no zone was exhausted: the first Z*k pods)
    zc = np.bincount(nodes.label_cols[1][got.log[: 3 * 10]], minlength=4)[1:]
    assert zc.max() - zc.min() <= skew


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(10))
def test_gpu_spread_random(ccref, seed):
    rng = np.random.default_rng(500 + seed)
    nodes, pod, prof = H.random_case(rng, int(rng.integers(1, 1200)))
    pod.spread = H.random_spread(rng, nodes, n_constraints=int(rng.integers(1, 3)))
    _gpu_check(ccref, nodes, pod, prof, int(rng.choice([0, 0, 60])))


@pytest.mark.gpu
def test_gpu_spread_rejects_batched_and_soft():
    nodes, pod, prof = synth.make_config("C3", n_nodes=300, seed=3)
    pod.spread = [synth.zone_spread(300)]
    e = capi.Engine(device=0)
    e.load(nodes, pod, prof)
    with pytest.raises(capi.CcsimError):
        e.run(mode="batched")
    pod.spread[0].hard = False
    e2 = capi.Engine(device=0)
    e2.load(nodes, pod, prof)
    with pytest.raises(capi.CcsimError):
        e2.run(mode="batched")


# ---- ScheduleAnyway (soft) constraints -> Score (scoring.go:61-265) ----
def test_ka_soft_spread_prefers_emptier_zone(ccref):
    # 2 zones, one node each, identical; zone 1 already runs 3 matching pods.  Raw score = count * log(2 + 2) + 0:
    # zone 1 -> round(3 * 1.386) = 4, zone 2 -> 0; normalized 100 * (4 + 0 - s) / 4 -> 0 vs 100, weight 2: node 1 wins
    # until the counts (and the small resource-score differences) even out.
    nodes = _zone_nodes([1, 2], [110, 110])
    pod = H.simple_pod(100, 64 * H.MiB)
    pod.spread = [M.SpreadConstraint(col=0, max_skew=1, hard=False, self_match=True, n_domains=2,
                                     node_match_count=np.array([3, 0], np.int32))]
    r = ccref.run(DEFAULT, nodes, pod, max_limit=3)
    assert r.log.tolist() == [1, 1, 1]


@pytest.mark.gpu
@pytest.mark.parametrize("n,limit,hostname", [(500, 300, False), (1500, 0, False), (700, 400, True)])
def test_gpu_soft_spread_synthetic(ccref, n, limit, hostname):
    nodes, pod, prof = synth.make_config("C3", n_nodes=n, seed=40 + n)
    pod.spread = [M.SpreadConstraint(col=1, max_skew=2, hard=False, self_match=True, n_domains=synth.zones_for(n))]
    if hostname:
        nodes.label_cols.append(np.arange(1, n + 1, dtype=np.int32))
        pod.spread.append(M.SpreadConstraint(col=2, max_skew=1, hard=False, self_match=True, is_hostname=True, n_domains=n))
    _gpu_check(ccref, nodes, pod, prof, limit)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(12))
def test_gpu_soft_and_hard_spread_random(ccref, seed):
    rng = np.random.default_rng(700 + seed)
    nodes, pod, prof = H.random_case(rng, int(rng.integers(1, 900)))
    cons = H.random_spread(rng, nodes, n_constraints=2)
    for k in cons:
        k.hard = bool(rng.integers(0, 2))
    if seed % 4 == 0:
        cons[0].is_hostname, cons[0].hard = True, False  # scored per node instead of per domain
    pod.spread = cons
    _gpu_check(ccref, nodes, pod, prof, int(rng.choice([0, 0, 70])))


@pytest.mark.gpu
@pytest.mark.parametrize("reset", [False, True])
def test_gpu_soft_hostname_counts_after_runs_of_another_pod(ccref, reset):
    """The clones of a pod with a self-matching ScheduleAnyway hostname constraint are counted per node as len(Pods) now minus len(Pods) when
    the pod was SET (DevSoft::pod_count0) -- not minus the loaded snapshot's, which would count another pod's earlier clones on this
    engine as this pod's (the hosts' one-cycle-at-a-time loop sets a different template every cycle).  After ccsim_reset_state the
    reference point is the loaded snapshot again."""
    import copy
    n = 300
    nodes, first, prof = synth.make_config("C3", n_nodes=n, seed=91)
    nodes.label_cols.append(np.arange(1, n + 1, dtype=np.int32))
    second = copy.copy(first)
    second.req = list(first.req)
    second.req[0], second.nz_mcpu = first.req[0] // 2 + 1, first.req[0] // 2 + 1
    second.spread = [M.SpreadConstraint(col=len(nodes.label_cols) - 1, max_skew=1, hard=False, self_match=True, is_hostname=True, n_domains=n)]
    a = ccref.run(prof, nodes, first, max_limit=170)
    after = copy.deepcopy(nodes)
    for w in a.log.tolist():  # (types.go:409-428 AddPod, as oracle/ccref.c applies it)
        for c in range(len(after.req)):
            after.req[c][w] += first.req[c]
        after.nz_mcpu[w] += first.nz_mcpu
        after.nz_mem[w] += first.nz_mem
        after.pod_count[w] += 1
    e = capi.Engine(device=0)
    try:
        e.load(nodes, first, prof)
        g = e.run(max_limit=170, mode="sequential", want_log=True, log_cap=170)
        assert g.log.tolist() == a.log.tolist()
        e.set_pod(second)
        if reset:
            e.reset_state()
        ref = ccref.run(prof, nodes if reset else after, second, max_limit=260)
        got = e.run(max_limit=260, mode="sequential", want_log=True, log_cap=260)
        assert got.log.tolist() == ref.log.tolist()
    finally:
        e.close()


# ---- the plugin's SYSTEM DEFAULT constraints: requireAllTopologies = false (scoring.go:61-115,140,147-178,205-219) -------------------------
def _relaxed_case(rng, n):
    """random_case() nodes (label column 1: 0 = the zone key is MISSING on some nodes) + a hostname column every node carries, and the two
    system default constraints (plugin.go:48-59: hostname maxSkew 3, zone maxSkew 5, ScheduleAnyway) as a pod without constraints of its own
    gets them: scored with requireAllTopologies = false."""
    nodes, pod, prof = H.random_case(rng, n)
    nodes.label_cols = list(nodes.label_cols) + [np.arange(1, nodes.n + 1, dtype=np.int32)]
    host = len(nodes.label_cols) - 1
    mc = lambda: rng.integers(0, 3, nodes.n).astype(np.int32) if rng.integers(0, 2) else None
    pod.spread = [M.SpreadConstraint(col=host, max_skew=3, hard=False, self_match=True, is_hostname=True, n_domains=nodes.n, node_match_count=mc()),
                  M.SpreadConstraint(col=1, max_skew=5, hard=False, self_match=True, n_domains=2, node_match_count=mc())]
    pod.soft_relaxed = True
    return nodes, pod, prof


def test_ka_system_default_spreading_scores_nodes_without_the_zone_label(ccref):
    """Three identical nodes, hostnames h1..h3, zones [a, a, -]; one matching pod already runs on node 0.  requireAllTopologies = false:
    node 2 is NOT ignored -- it scores its hostname count only (0) and nothing for the zone it does not have; the "" value is a third
    zone domain when the weight is sized: zone weight log(3 + 2)?  No: the domains over the FILTERED nodes are {a, ""} -> log(2 + 2).
    Raw scores: hostname weight log(3 + 2) = 1.609; node 0: 1 * 1.609 + 2 + 1 * 1.386 + 4 = round(8.995) = 9; node 1: 0 + 2 + 1 * 1.386 + 4
    = round(7.386) = 7; node 2: 0 + 2 = 2 (no zone credit at all, not even maxSkew - 1).  Normalized 100 * (9 + 2 - s) / 9: 22, 44, 100 ->
    the first clone goes to the node WITHOUT a zone label.  With the pod's own constraints (requireAllTopologies = true) node 2 would be
    ignored (score 0) and node 1 would win."""
    nodes = H.simple_nodes([4000] * 3, [8 * H.GiB] * 3, [110] * 3, label_cols=[np.array([1, 2, 3], np.int32), np.array([1, 1, 0], np.int32)])
    pod = H.simple_pod(100, 64 * H.MiB)
    cnt = np.array([1, 0, 0], np.int32)
    pod.spread = [M.SpreadConstraint(col=0, max_skew=3, hard=False, self_match=True, is_hostname=True, n_domains=3, node_match_count=cnt),
                  M.SpreadConstraint(col=1, max_skew=5, hard=False, self_match=True, n_domains=1, node_match_count=cnt)]
    pod.soft_relaxed = True
    raw, norm, w = ccref.unit_pts_scores(nodes, pod, [0, 1, 2])
    assert raw == [9, 7, 2] and norm == [22, 44, 100]
    assert ccref.run(DEFAULT, nodes, pod, max_limit=1).log.tolist() == [2]
    pod.soft_relaxed = False
    assert ccref.unit_pts_scores(nodes, pod, [0, 1, 2])[1][2] == 0 and ccref.run(DEFAULT, nodes, pod, max_limit=1).log.tolist() == [1]


@pytest.mark.gpu
@pytest.mark.parametrize("cw", ["1", "0"])
@pytest.mark.parametrize("seed", range(10))
def test_gpu_system_default_spreading_random(ccref, monkeypatch, seed, cw):
    """Engine (windows of placements / one pass per placement) against the oracle's literal requireAllTopologies = false branch, on clusters
    where nodes lack the zone key: the engine takes the form model.relax_soft derives (include/ccsim.h missing_value)."""
    monkeypatch.setenv("CCSIM_CW", cw)
    rng = np.random.default_rng(5100 + seed)
    nodes, pod, prof = _relaxed_case(rng, int(rng.integers(3, 700)))
    assert (nodes.label_cols[1] == 0).any() or nodes.n < 6
    _gpu_check(ccref, nodes, pod, prof, int(rng.choice([0, 0, 90])))


@pytest.mark.gpu
def test_gpu_system_default_spreading_on_a_cluster_without_zone_labels(ccref):
    """The on-premises case the reference's comment names (scoring.go:137-139): no node has a zone label, hostname spreading still works."""
    nodes, pod, prof = synth.make_config("C3", n_nodes=900, seed=12)
    nodes.label_cols[1] = np.zeros(nodes.n, np.int32)
    nodes.label_cols = list(nodes.label_cols) + [np.arange(1, nodes.n + 1, dtype=np.int32)]
    pod.spread = [M.SpreadConstraint(col=2, max_skew=3, hard=False, self_match=True, is_hostname=True, n_domains=nodes.n),
                  M.SpreadConstraint(col=1, max_skew=5, hard=False, self_match=True, n_domains=0)]
    pod.soft_relaxed = True
    _gpu_check(ccref, nodes, pod, prof, 700)
