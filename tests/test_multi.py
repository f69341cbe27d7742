"""Several pod specs cycled round-robin (BASELINE config 5): ccsim_set_pods / ccsim_run / ccsim_schedule_pod against the
oracle's round-robin loop (oracle/ccref.c ccref_run_multi: one reference scheduling cycle per pod, in order).

The engine resolves WINDOWS of consecutive pods per pass and validates every pod's choice against the placements of the
pods before it (csrc/ccsim_multi.h); the result must not depend on the window size."""
import os

import numpy as np
import pytest

import helpers as H
from cluster_capacity_amd import capi, model as M, synth


def random_specs(rng, nodes, n_specs, images=False):
    """Config-5-shaped specs over random_case()-style nodes (label columns: 0 = 'type' (0..4), 1 = 'zone' (0..2, 0 = key
    absent), 2 = hostname), with per-spec variety: tolerations, selectors, preferred terms, existing matching pods."""
    n = nodes.n
    t_in = lambda size, ids: np.isin(np.arange(size), ids).astype(np.uint8)
    sel_cache = {}
    pods = []
    img_pool = [rng.integers(0, 101, n).astype(np.uint8) for _ in range(2)] if images else []
    for j in range(n_specs):
        cpu, mem = int(rng.choice([50, 100, 250, 500, 900])), int(rng.choice([64, 128, 512, 1024])) * H.MiB
        kw = {}
        inc = None
        if rng.random() < 0.4:
            ids = tuple(sorted(rng.choice(5, size=int(rng.integers(1, 4)), replace=False).tolist()))
            kw = dict(affinity_filter_active=True, has_node_selector=True, node_selector=[(0, t_in(5, list(ids)))])
            if ids not in sel_cache:
                sel_cache[ids] = np.isin(nodes.label_cols[0], list(ids)).astype(np.uint8)
            inc = sel_cache[ids]
        if rng.random() < 0.3:
            kw["preferred"] = [(int(rng.integers(1, 50)), [(0, t_in(5, [int(rng.integers(0, 5))]))])]
        ok = np.array([1, rng.integers(0, 2), 1, rng.integers(0, 2)], np.uint8)
        cnt = np.array([0, 0, rng.integers(0, 3), rng.integers(0, 3)], np.int32)
        spread = []
        if rng.random() < 0.8:
            spread.append(M.SpreadConstraint(col=1, max_skew=int(rng.integers(1, 4)), min_domains=int(rng.integers(1, 3)), hard=True,
                                             self_match=bool(rng.random() < 0.9), n_domains=2, node_included=inc,
                                             node_match_count=rng.integers(0, 2, n).astype(np.int32) if rng.random() < 0.3 else None))
        if rng.random() < 0.2:
            spread.append(M.SpreadConstraint(col=0, max_skew=int(rng.integers(1, 6)), min_domains=1, hard=True, self_match=True, n_domains=4,
                                             node_included=inc))
        ipa = None
        if rng.random() < 0.7:
            ex = (rng.random(n) < 0.05).astype(np.int32) if rng.random() < 0.3 else None
            ipa = M.InterPodAffinity(key_cols=[2], key_ndom=[n], anti_keys=[0], anti_self=[True], anti_existing=[ex], exist_anti=[None],
                                     score_existing=[None], score_self=[0], self_entries=[0])
        if img_pool and rng.random() < 0.35:  # ImageLocality scores: some specs share an array (one static class), some bring their own
            kw["image_score"] = img_pool[int(rng.integers(0, len(img_pool)))] if rng.random() < 0.6 else rng.integers(0, 101, n).astype(np.uint8)
        pods.append(M.PodSpec(req=np.array([cpu, mem, 0], np.int64), nz_mcpu=cpu, nz_mem=mem, taint_filter_ok=ok, taint_prefer_cnt=cnt,
                              tolerates_unschedulable=bool(rng.random() < 0.2), spread=spread, ipa=ipa, **kw))
    return pods


def random_multi_case(rng, n, n_specs, images=False):
    nodes = H.simple_nodes(rng.choice([2000, 4000, 8000, 16000], n), rng.choice([4, 8, 16, 32], n) * H.GiB, rng.integers(3, 30, n),
                           req_mcpu=rng.integers(0, 20, n) * 50, req_mem=rng.integers(0, 8, n) * 256 * H.MiB, pod_count=rng.integers(0, 3, n),
                           taintset_id=rng.integers(0, 4, n), unschedulable=(rng.random(n) < 0.03),
                           label_cols=[rng.integers(0, 5, n), rng.integers(0, 3, n), np.arange(1, n + 1)])
    return nodes, random_specs(rng, nodes, n_specs, images), M.Profile.default()


def _same(got, ref):
    assert got.placed == ref.placed and got.stop == ref.stop and got.stop_spec == ref.stop_spec
    assert np.array_equal(got.log, ref.log)
    assert np.array_equal(got.per_node_count, ref.per_node_count)
    assert np.array_equal(got.per_spec_count, ref.per_spec_count)
    if ref.stop == M.STOP_UNSCHEDULABLE:
        assert np.array_equal(got.hist, ref.hist) and got.n_code_unschedulable == ref.n_code_unschedulable
        assert np.array_equal(got.hist_taintset[: len(ref.hist_taintset)], ref.hist_taintset[: len(got.hist_taintset)])


def test_oracle_round_robin_with_one_spec_is_the_single_spec_loop(ccref):
    nodes, pod, prof = synth.make_config("C3", n_nodes=300, seed=3)
    a, b = ccref.run(prof, nodes, pod, max_limit=0), ccref.run_multi(prof, nodes, [pod], max_limit=0)
    assert a.placed == b.placed and np.array_equal(a.log, b.log) and np.array_equal(a.hist, b.hist) and b.stop_spec == 0


def test_oracle_two_specs_hand_case(ccref):
    # 2 nodes x (1000m, 1 GiB, 10 pods); spec A 300m with hostname anti-affinity to itself, spec B 200m without.
    # A B A B ... : A takes each node once (cycles 0 and 2), its third pod (cycle 4) finds both nodes taken -> Unschedulable.
    nodes = H.simple_nodes([1000, 1000], [H.GiB, H.GiB], [10, 10], label_cols=[np.array([1, 2], np.int32)])
    a = H.simple_pod(300, 64 * H.MiB)
    a.ipa = M.InterPodAffinity(key_cols=[0], key_ndom=[2], anti_keys=[0], anti_self=[True], anti_existing=[None], exist_anti=[None],
                               score_existing=[None], score_self=[0], self_entries=[0])
    b = H.simple_pod(200, 64 * H.MiB)
    r = ccref.run_multi(M.Profile.default(), nodes, [a, b], max_limit=0)
    assert r.placed == 4 and r.stop == M.STOP_UNSCHEDULABLE and r.stop_spec == 0
    assert r.per_spec_count.tolist() == [2, 2] and sorted(r.log[[0, 2]].tolist()) == [0, 1]
    assert r.hist[M.R_IPA_ANTI] == 2


@pytest.mark.gpu
@pytest.mark.parametrize("window", ["1", "7", "64", "128"])  # (128: round 6, the assignment over two waves; the in-order fallback stays at 64)
@pytest.mark.parametrize("seed", range(10))
def test_random_specs_vs_oracle(ccref, monkeypatch, window, seed):
    monkeypatch.setenv("CCSIM_MULTI_WINDOW", window)
    rng = np.random.default_rng(5000 + seed)
    nodes, pods, prof = random_multi_case(rng, int(rng.integers(20, 700)), int(rng.integers(2, 80)))
    limit = int(rng.choice([0, 0, 0, 150]))
    ref = ccref.run_multi(prof, nodes, pods, max_limit=limit)
    e = capi.Engine(device=0)
    e.load(nodes, pods, prof)
    got = e.run(max_limit=limit, log_cap=max(1, ref.placed))
    _same(got, ref)
    e.reset_state()  # the specs' own state (spread tables, anti-affinity bits, round-robin position) is restored too
    _same(e.run(max_limit=limit, log_cap=max(1, ref.placed)), ref)
    e.close()


@pytest.mark.gpu
def test_c5_shape_10k_nodes_1024_specs_vs_oracle(ccref):
    nodes, pods, prof = synth.make_c5(10_000, 1024)
    ref = ccref.run_multi(prof, nodes, pods, max_limit=6000, threads=8)
    e = capi.Engine(device=0)
    e.load(nodes, pods, prof)
    got = e.run(max_limit=6000, log_cap=6000)
    _same(got, ref)
    e.close()


@pytest.mark.gpu
def test_score_memo_on_and_off_agree_and_the_memo_serves_the_scans(ccref, monkeypatch):
    """The resident (spec, node) score memo (csrc/ccsim_multi.h): after each spec's first scan the scans read their rows -- same log,
    counts and stop as with every scan computing (CCSIM_MULTI_MEMO_MB=0) and as the oracle's prefix; a second run on the same engine
    starts from unstamped rows again."""
    nodes, pods, prof = synth.make_c5(10_000, 256)
    ref = ccref.run_multi(prof, nodes, pods, max_limit=1500, threads=8)
    runs = {}
    for memo in ("on", "off"):
        if memo == "off":
            monkeypatch.setenv("CCSIM_MULTI_MEMO_MB", "0")
        e = capi.Engine(device=0)
        e.load(nodes, pods, prof)
        r = e.run(max_limit=20_000, log_cap=20_000)
        mm = e.multi_memo()
        assert mm["on"] == (memo == "on") and mm["memo_scans"] + mm["full_scans"] >= r.placed
        assert ref.placed == 1500 and r.placed > 5000
        if memo == "on":  # every spec computes its row once (and again after a re-derived maximum); the rest is read
            assert 2 * 256 * nodes.n <= mm["bytes"] < 2 * 256 * (nodes.n + 1024)  # 16-bit words (round 5)
            assert mm["memo_scans"] > 10 * mm["full_scans"], mm
            e.reset_state()
            r2 = e.run(max_limit=20_000, log_cap=20_000)
            assert np.array_equal(r2.log, r.log) and np.array_equal(r2.per_node_count, r.per_node_count) and e.multi_memo()["full_scans"] >= 256
        else:
            assert mm["memo_scans"] == 0
        runs[memo] = r
        e.close()
    a, b = runs["on"], runs["off"]
    assert a.placed == b.placed > 5000 and a.stop == b.stop and a.stop_spec == b.stop_spec
    assert np.array_equal(a.log, b.log) and np.array_equal(a.per_node_count, b.per_node_count) and np.array_equal(a.per_spec_count, b.per_spec_count)
    assert np.array_equal(a.log[:1500], ref.log)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(4))
def test_score_memo_random_specs_long_runs_vs_oracle(ccref, monkeypatch, seed):
    """Whole runs (every spec scanned many times, nodes filling up, maxima re-derived) with the memo against the oracle, with windows that
    keep ending early (window 7) and the in-order commit (CCSIM_MULTI_SEQ)."""
    rng = np.random.default_rng(9100 + seed)
    if seed & 1:
        monkeypatch.setenv("CCSIM_MULTI_SEQ", "1")
    monkeypatch.setenv("CCSIM_MULTI_WINDOW", "7" if seed < 2 else "64")
    nodes, pods, prof = random_multi_case(rng, int(rng.integers(300, 900)), int(rng.integers(3, 40)))
    ref = ccref.run_multi(prof, nodes, pods, max_limit=0)
    e = capi.Engine(device=0)
    e.load(nodes, pods, prof)
    _same(e.run(max_limit=0, log_cap=max(1, ref.placed)), ref)
    mm = e.multi_memo()
    assert mm["on"] and mm["memo_scans"] > 0
    e.close()


@pytest.mark.gpu
@pytest.mark.parametrize("window", ["5", "64"])
@pytest.mark.parametrize("seed", range(4))
def test_random_specs_with_image_locality_scores_vs_oracle(ccref, monkeypatch, window, seed):
    """Specs with ImageLocality scores (image_locality.go:54-66: the node's score as it is, weight 1): the score lives in the static
    word of the spec's CLASS, so specs share a class only with equal per-node scores."""
    monkeypatch.setenv("CCSIM_MULTI_WINDOW", window)
    rng = np.random.default_rng(9300 + seed)
    nodes, pods, prof = random_multi_case(rng, int(rng.integers(100, 700)), int(rng.integers(3, 50)), images=True)
    assert any(p.image_score is not None for p in pods)
    ref = ccref.run_multi(prof, nodes, pods, max_limit=0)
    e = capi.Engine(device=0)
    e.load(nodes, pods, prof)
    _same(e.run(max_limit=0, log_cap=max(1, ref.placed)), ref)
    e.close()


@pytest.mark.gpu
def test_c5_shape_whole_run_small(ccref):
    nodes, pods, prof = synth.make_c5(3000, 96)
    ref = ccref.run_multi(prof, nodes, pods, max_limit=0, threads=8)
    assert ref.stop == M.STOP_UNSCHEDULABLE
    e = capi.Engine(device=0)
    e.load(nodes, pods, prof)
    _same(e.run(max_limit=0, log_cap=max(1, ref.placed)), ref)
    e.close()


@pytest.mark.gpu
def test_schedule_pod_cycle_by_cycle(ccref):
    rng = np.random.default_rng(77)
    nodes, pods, prof = random_multi_case(rng, 400, 12)
    ref = ccref.run_multi(prof, nodes, pods, max_limit=300)
    e = capi.Engine(device=0)
    e.load(nodes, pods, prof)
    for i in range(ref.placed):
        node, evaluated, feasible = e.schedule_pod(i % len(pods))
        assert node == ref.log[i] and evaluated == nodes.n and feasible > 0
    if ref.stop == M.STOP_UNSCHEDULABLE:
        assert e.schedule_pod(ref.stop_spec)[0] == -1
    e.close()


@pytest.mark.gpu
def test_unsupported_multi_spec_shapes_are_rejected():
    nodes, pods, prof = synth.make_c5(600, 4)
    e = capi.Engine(device=0)
    soft = synth.make_c5(600, 4)[1]
    soft[1].spread = [M.SpreadConstraint(col=1, max_skew=1, hard=False, n_domains=3)]
    with pytest.raises(capi.CcsimError, match="ScheduleAnyway"):
        e.load(nodes, soft, prof)
    aff = synth.make_c5(600, 4)[1]
    aff[2].ipa = M.InterPodAffinity(key_cols=[1], key_ndom=[3], aff_keys=[0], self_aff=True, score_self=[1], self_entries=[1])
    with pytest.raises(capi.CcsimError, match="inter-pod affinity"):
        e.load(nodes, aff, prof)
    e.load(nodes, pods, M.Profile.default())  # and the supported shape loads
    e.close()


# ---- round 5: pod-spec sets the window engine refuses take the literal loop, one scheduling cycle at a time ------
def _refused_specs(rng, nodes, n_specs, kind):
    """random_specs() bent out of the window engine's shape: `soft` = a ScheduleAnyway constraint (scores against cluster-wide counts),
    `scalar` = a scalar resource request, `affinity` = required inter-pod affinity over a shared key with preferred-term scores,
    `ports` = one template with host ports."""
    pods = random_specs(rng, nodes, n_specs)
    n = nodes.n
    j = int(rng.integers(0, n_specs))
    if kind == "soft":
        pods[j].spread = list(pods[j].spread) + [M.SpreadConstraint(col=1, max_skew=int(rng.integers(1, 3)), hard=False, self_match=True, n_domains=2)]
    elif kind == "scalar":
        for q in pods[:: 2]:
            q.req = np.array(list(q.req) + [int(rng.integers(1, 3))], np.int64)
            q.has_scalar_entries = True
        for q in pods[1:: 2]:
            q.req = np.array(list(q.req) + [0], np.int64)
    elif kind == "affinity":
        pods[j].ipa = M.InterPodAffinity(key_cols=[1], key_ndom=[2], aff_keys=[0], self_aff=True, anti_keys=[], anti_self=[], anti_existing=[],
                                         exist_anti=[None], score_existing=[rng.integers(-2, 3, n).astype(np.int64)], score_self=[int(rng.integers(1, 4))],
                                         self_entries=[1], entries_existing=int(n // 3))
    elif kind == "ports":
        pods[j].has_host_ports = True
        pods[j].host_ports_conflict = (rng.random(n) < 0.1).astype(np.uint8)
    return pods


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["soft", "scalar", "affinity", "ports"])
@pytest.mark.parametrize("seed", range(3))
def test_refused_spec_sets_take_one_cycle_at_a_time_vs_oracle(ccref, kind, seed):
    from cluster_capacity_amd import cli
    rng = np.random.default_rng(4400 + seed)
    n = int(rng.integers(60, 400))
    nodes, _, prof = random_multi_case(rng, n, 1)
    if kind == "scalar":  # a scalar resource column on the nodes
        nodes.alloc.append(rng.integers(0, 9, n).astype(np.int64))
        nodes.req.append(rng.integers(0, 2, n).astype(np.int64))
        nodes.scalar_names = ["example.com/gpu"]
    pods = _refused_specs(rng, nodes, int(rng.integers(2, 7)), kind)
    limit = int(rng.choice([0, 0, 150]))
    ref = ccref.run_multi(prof, nodes, pods, max_limit=limit)
    e = capi.Engine(device=0)
    try:
        e.load(nodes, pods, prof)
        refused = False
    except capi.CcsimError as ex:
        refused = ex.rc == -38
    e.close()
    assert refused, "the case was meant to be outside the window engine's shape"
    got = cli.simulate_specs_one_cycle_at_a_time(nodes, pods, prof, limit)
    _same(got, ref)


@pytest.mark.gpu
@pytest.mark.parametrize("seq", ["0", "1"])
@pytest.mark.parametrize("seed", range(int(os.environ.get("CC_TEST_SEEDS", "6"))))  # (CC_TEST_SEEDS=150: the soak behind profiles/r06/c5_windows_128_soak.txt)
def test_windows_of_128_pods_vs_oracle(ccref, monkeypatch, seed, seq):
    """Round 6: windows of up to 128 pods (the assignment's per-pod rows in the lanes of TWO waves, eight checking threads per pod,
    up to eight pairs per thread in the verification), with more specs than a window holds; `seq`: the in-order commit forced -- it holds
    a pod / a touched node per lane of ONE wave and takes at most 64 pods of a window the scan prepared for 128.  Whole runs (nodes
    filling up, maxima re-derived, windows ending early) and limits inside a window == the oracle's round-robin loop."""
    if seq == "1":
        monkeypatch.setenv("CCSIM_MULTI_SEQ", "1")
    rng = np.random.default_rng(9900 + seed)
    nodes, pods, prof = random_multi_case(rng, int(rng.integers(400, 2500)), int(rng.integers(130, 400)))
    limit = int(rng.choice([0, 0, 777, 3000]))
    ref = ccref.run_multi(prof, nodes, pods, max_limit=limit, threads=8)
    e = capi.Engine(device=0)
    e.load(nodes, pods, prof)
    got = e.run(max_limit=limit, log_cap=max(1, ref.placed))
    _same(got, ref)
    if seq == "0" and ref.placed > 2000:
        assert got.scans < ref.placed / 8, (got.scans, ref.placed)  # windows really were windows (on a few hundred nodes most of them end early: soak seeds 50, 71, 80 average 18 pods)
    e.reset_state()
    _same(e.run(max_limit=limit, log_cap=max(1, ref.placed)), ref)
    e.close()
