"""The windowed mode for ONE template with topology-coupled plugins (csrc/ccsim_coupled.h: k_cw_scan / k_cw_top / k_cw_merge /
k_cw_decide) against the oracle's literal one-cycle-at-a-time loop, placement by placement: the generators of tests/test_spread.py,
tests/test_ipa.py and tests/test_coupled_model.py, with windows of 1 ... 64 cycles and class lists of 1 ... 16 members; the
fallback to one pass per placement when the mode cannot represent a run; and that the windowed path is the one that ran.
(The argument itself is checked on the CPU: tests/coupled_model.py `device_plan=True`, tests/test_coupled_model.py.)"""
import numpy as np
import pytest

import helpers as H
from cluster_capacity_amd import capi, model as M, report as R, synth
from test_coupled_model import coupled_case

WINDOWS = [(1, 1), (7, 2), (64, 16)]


def _run(ccref, nodes, pod, prof, limit, monkeypatch, window, list_len, expect_plan=True, mode="sequential"):
    monkeypatch.setenv("CCSIM_CW_WINDOW", str(window))
    monkeypatch.setenv("CCSIM_CW_LIST", str(list_len))
    ref = ccref.run(prof, nodes, pod, max_limit=limit)
    e = capi.Engine(device=0)
    e.load(nodes, pod, prof)
    got = e.run(max_limit=limit, mode=mode, log_cap=max(1, ref.placed))
    info = e.coupled_info()
    where = (window, list_len, info, got.placed, ref.placed)
    n = min(len(got.log), len(ref.log))
    first = next((i for i in range(n) if got.log[i] != ref.log[i]), None)
    assert first is None, ("first differing placement", first, got.log[max(0, first - 2): first + 3].tolist(), ref.log[max(0, first - 2): first + 3].tolist(), where)
    assert got.placed == ref.placed and got.stop == ref.stop, where
    assert np.array_equal(got.per_node_count, ref.per_node_count), where
    if ref.stop == M.STOP_UNSCHEDULABLE:
        assert np.array_equal(got.hist, ref.hist), (got.hist, ref.hist, where)
        assert got.n_code_unschedulable == ref.n_code_unschedulable
        assert R.stop_reason(got, nodes.n, limit) == R.stop_reason(ref, nodes.n, limit)
    if expect_plan:
        assert info["plan"] and info["windows"] > 0 and not info["fell_back"], where
    e.close()
    return got, info


@pytest.mark.gpu
@pytest.mark.parametrize("window,list_len", WINDOWS)
@pytest.mark.parametrize("seed", range(10))
def test_hard_spread_random(ccref, monkeypatch, seed, window, list_len):
    rng = np.random.default_rng(500 + seed)  # (the cases of tests/test_spread.py::test_gpu_spread_random)
    nodes, pod, prof = H.random_case(rng, int(rng.integers(1, 1200)))
    pod.spread = H.random_spread(rng, nodes, n_constraints=int(rng.integers(1, 3)))
    _run(ccref, nodes, pod, prof, int(rng.choice([0, 0, 60])), monkeypatch, window, list_len)


@pytest.mark.gpu
@pytest.mark.parametrize("window,list_len", WINDOWS)
@pytest.mark.parametrize("seed", range(12))
def test_soft_and_hard_spread_random(ccref, monkeypatch, seed, window, list_len):
    rng = np.random.default_rng(700 + seed)  # (tests/test_spread.py::test_gpu_soft_and_hard_spread_random)
    nodes, pod, prof = H.random_case(rng, int(rng.integers(1, 900)))
    cons = H.random_spread(rng, nodes, n_constraints=2)
    for k in cons:
        k.hard = bool(rng.integers(0, 2))
    if seed % 4 == 0:
        cons[0].is_hostname, cons[0].hard = True, False
    pod.spread = cons
    _run(ccref, nodes, pod, prof, int(rng.choice([0, 0, 70])), monkeypatch, window, list_len)


@pytest.mark.gpu
@pytest.mark.parametrize("window,list_len", WINDOWS)
@pytest.mark.parametrize("seed", range(16))
def test_ipa_random(ccref, monkeypatch, seed, window, list_len):
    rng = np.random.default_rng(900 + seed)  # (tests/test_ipa.py::test_gpu_ipa_random)
    nodes, pod, prof = H.random_case(rng, int(rng.integers(1, 1000)))
    pod.ipa = H.random_ipa(rng, nodes)
    if seed % 3 == 0:
        pod.spread = H.random_spread(rng, nodes, n_constraints=1)
    _run(ccref, nodes, pod, prof, int(rng.choice([0, 0, 80])), monkeypatch, window, list_len)


@pytest.mark.gpu
@pytest.mark.parametrize("window,list_len", [(1, 1), (5, 3), (64, 16), (64, 4), (1024, 64)])  # (the last: the defaults; many classes x long lists = shortened staged lists)
@pytest.mark.parametrize("seed", range(30))
def test_coupled_cases_with_unique_keys(ccref, monkeypatch, seed, window, list_len):
    """tests/test_coupled_model.py's generator: hard / soft constraints and inter-pod terms over shared keys AND over a unique-per-node
    key (hostname), existing pods' counts, missing labels, node inclusion, minDomains, host ports, images; every third case roomy
    (nodes take many clones: several windows, touched nodes that win again)."""
    rng = np.random.default_rng(7100 + seed)
    nodes, pod, prof = coupled_case(rng, int(rng.integers(12, 700)), roomy=seed % 3 == 0)
    limit = 1500 if seed % 3 == 0 else int(rng.choice([0, 0, 29]))
    coupled = bool((any(c.hard for c in pod.spread) and prof.filter_mask & M.F_TOPOLOGYSPREAD) or
                   (any(not c.hard for c in pod.spread) and prof.w_topologyspread) or
                   (pod.ipa is not None and (prof.filter_mask & M.F_INTERPODAFFINITY or prof.w_interpodaffinity)))  # (what the engine activates)
    _run(ccref, nodes, pod, prof, limit, monkeypatch, window, list_len, expect_plan=coupled)


@pytest.mark.gpu
def test_reference_fixtures_and_hand_cases(ccref, monkeypatch):
    from test_ipa import colocation_nodes, self_affinity_pod
    got, _ = _run(ccref, colocation_nodes([1, 2, 3]), self_affinity_pod(3), M.Profile.default(), 100, monkeypatch, 64, 16)
    assert got.placed == 30 and got.per_node_count.tolist() == [30, 0, 0]  # pod_colocation_test.go:18-97 (KA3)
    got, _ = _run(ccref, colocation_nodes([1, 1, 1, 2, 2, 2, 3, 3, 3]), self_affinity_pod(3), M.Profile.default(), 100, monkeypatch, 64, 16)
    assert got.placed == 90 and got.per_node_count.tolist() == [30, 30, 30, 0, 0, 0, 0, 0, 0]  # :99-190 (KA4)
    from test_spread import _pod, _zone_nodes
    for w in (1, 64):
        _run(ccref, _zone_nodes([1, 2, 3], [5, 110, 110]), _pod(max_skew=1, n_domains=3, self_match=True), M.Profile.default(), 0, monkeypatch, w, 16)
        _run(ccref, _zone_nodes([1, 2], [110, 110]), _pod(max_skew=2, min_domains=3, n_domains=2, self_match=True), M.Profile.default(), 0, monkeypatch, w, 16)
        _run(ccref, _zone_nodes([1, 2, 0], [3, 4, 50]), _pod(max_skew=1, n_domains=2, self_match=False), M.Profile.default(), 0, monkeypatch, w, 16)


def c5_single_template(n, zones=None, seed=5):
    """BASELINE config 5's pod shape as ONE template: DoNotSchedule zone spread (maxSkew 1) + required hostname anti-affinity against
    its own clones, on a C3-style snapshot."""
    nodes, pod, prof = synth.make_config("C3", n_nodes=n, seed=seed)
    nodes.label_cols.append(np.arange(1, n + 1, dtype=np.int32))  # hostname
    pod.ipa = M.InterPodAffinity(key_cols=[len(nodes.label_cols) - 1], key_ndom=[n], anti_keys=[0], anti_self=[True], anti_existing=[None])
    pod.spread = [synth.zone_spread(n, max_skew=1)]
    return nodes, pod, prof


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1000, 20_000])
def test_c5_shape_single_template_windows_do_the_work(ccref, monkeypatch, n):
    nodes, pod, prof = c5_single_template(n)
    got, info = _run(ccref, nodes, pod, prof, 0 if n <= 1000 else 3000, monkeypatch, 64, 16)
    assert got.per_node_count.max() == 1
    assert info["windows"] * 16 < got.placed, info  # far fewer node passes than placements: the windows did the work


@pytest.mark.gpu
@pytest.mark.parametrize("group", [2, 5])
def test_class_lists_merged_in_two_levels(ccref, monkeypatch, group):
    """Snapshots with more node blocks than one merge workgroup stages (about 100k nodes at L = 64) merge the blocks' class lists in
    two levels (k_cw_merge: groups of blocks, then the groups).  CCSIM_CW_MERGE_GROUP forces the same on small snapshots."""
    monkeypatch.setenv("CCSIM_CW_MERGE_GROUP", str(group))
    n = 1024 * group * group - 300  # (kCwTile nodes per block: group^2 blocks, the last one ragged)
    nodes, pod, prof = c5_single_template(n)
    got, info = _run(ccref, nodes, pod, prof, 700, monkeypatch, 64, 16)
    assert info["windows"] * 8 < got.placed, info
    rng = np.random.default_rng(4242 + group)
    nodes, pod, prof = coupled_case(rng, n, roomy=True)
    _run(ccref, nodes, pod, prof, 900, monkeypatch, 64, 8, expect_plan=False)


@pytest.mark.gpu
def test_falls_back_exactly_when_the_classes_do_not_fit(ccref, monkeypatch):
    """More classes than the windowed mode represents (a soft hostname constraint over nodes with many different existing counts x
    zones): DevState::cw_fallback, the run continues one pass per placement -- same answer."""
    rng = np.random.default_rng(11)
    n = 3000
    nodes, pod, prof = synth.make_config("C3", n_nodes=n, seed=77)
    nodes.label_cols.append(np.arange(1, n + 1, dtype=np.int32))
    pod.spread = [M.SpreadConstraint(col=len(nodes.label_cols) - 1, max_skew=1, hard=False, self_match=True, is_hostname=True, n_domains=n,
                                     node_match_count=rng.integers(0, 400, n).astype(np.int32)),
                  M.SpreadConstraint(col=1, max_skew=2, hard=False, self_match=True, n_domains=synth.zones_for(n))]
    got, info = _run(ccref, nodes, pod, prof, 150, monkeypatch, 64, 16, expect_plan=False)
    assert info["plan"] and info["fell_back"], info


@pytest.mark.gpu
def test_disabled_by_knob_and_unaffected_modes(ccref, monkeypatch):
    nodes, pod, prof = c5_single_template(600)
    monkeypatch.setenv("CCSIM_CW", "0")
    got, info = _run(ccref, nodes, pod, prof, 200, monkeypatch, 64, 16, expect_plan=False)
    assert not info["plan"] and got.scans >= got.placed  # one pass per placement
    monkeypatch.delenv("CCSIM_CW")
    # continued runs: a windowed run, then more placements on the same engine, then a reset
    ref = ccref.run(prof, nodes, pod, max_limit=300)
    e = capi.Engine(device=0)
    e.load(nodes, pod, prof)
    a = e.run(max_limit=100, mode="sequential", log_cap=100)
    b = e.run(max_limit=200, mode="sequential", log_cap=200)
    assert np.array_equal(np.concatenate([a.log, b.log]), ref.log[:300]) and e.coupled_info()["windows"] > 0
    e.reset_state()
    c = e.run(max_limit=300, mode="sequential", log_cap=300)
    assert np.array_equal(c.log, ref.log)
    # the SchedulePod seam keeps stepping one cycle at a time on the state the windows left
    e.reset_state()
    e.run(max_limit=50, mode="sequential")
    for i in range(50, 60):
        assert e.schedule_one()[0] == ref.log[i]
    e.close()


# ---- round 5: SWEEPS -- whole rounds of placements at once in the lane-per-candidate kernel (csrc/ccsim_coupled.h `sweep`) ------------
def sweep_case(rng, n):
    """The shape sweeps apply to -- ONE hard spread constraint over a shared key, at most a unique-per-node inter-pod key -- with everything
    that cuts a round short or keeps a cycle out of one: 4 ... 40 domains with unequal numbers of existing matching pods (catch-up phases in
    which some feasible domain is below the cap), maxSkew 1 ... 3, minDomains above the domains present, nodes the inclusion policy does
    not count, nodes without the key, PreferNoSchedule taints / preferred affinity held by few nodes (normalization maxima that lose
    their last feasible holder inside a round), nodes that take several clones (winners that stay candidates), hostname anti-affinity
    (winners that leave), existing pods that block their node."""
    nodes, pod, prof = H.random_case(rng, n)
    ndom = int(rng.integers(4, 41))
    zone = rng.integers(1, ndom + 1, n).astype(np.int32)
    if rng.integers(0, 2):
        zone[rng.random(n) < 0.03] = 0
    host = np.arange(1, n + 1, dtype=np.int32)
    nodes.label_cols = list(nodes.label_cols) + [zone, host]
    zc, hc = len(nodes.label_cols) - 2, len(nodes.label_cols) - 1
    pod.spread = [M.SpreadConstraint(col=zc, max_skew=int(rng.choice([1, 1, 1, 1, 2, 3])), min_domains=int(rng.choice([1, 1, 1, ndom + 2])), hard=True,
                                     self_match=bool(rng.integers(0, 8) != 0), n_domains=ndom,
                                     node_match_count=rng.integers(0, 3, n).astype(np.int32) if rng.integers(0, 2) else None,
                                     node_included=(rng.random(n) < 0.9).astype(np.uint8) if rng.integers(0, 3) == 0 else None)]
    kind = int(rng.choice([0, 1, 1, 2, 2]))
    if kind >= 1:  # hostname anti-affinity against the own clones: every winner leaves (BASELINE config 5's pod shape)
        blocked = (rng.random(n) < 0.05).astype(np.int32) if kind == 2 else None  # existing pods the term matches
        pod.ipa = M.InterPodAffinity(key_cols=[hc], key_ndom=[n], anti_keys=[0], anti_self=[True], anti_existing=[blocked], exist_anti=[None],
                                     score_existing=[None], score_self=[0], self_entries=[0])
    else:  # winners stay candidates while they have room: give some of them room for several clones
        roomy = rng.random(n) < 0.3
        nodes.alloc_pods = np.where(roomy, nodes.alloc_pods * 4, nodes.alloc_pods).astype(np.int32)
        nodes.alloc = [np.where(roomy, a * 4, a) for a in nodes.alloc]
    return nodes, pod, prof


@pytest.mark.gpu
@pytest.mark.parametrize("sweep", ["1", "0"])
@pytest.mark.parametrize("seed", range(40))
def test_sweeps_random(ccref, monkeypatch, seed, sweep):
    """Rounds resolved at once == one placement per step == the oracle, placement by placement (log, stop, histogram)."""
    monkeypatch.setenv("CCSIM_CW_SWEEP", sweep)
    rng = np.random.default_rng(8800 + seed)
    nodes, pod, prof = sweep_case(rng, int(rng.integers(40, 4000)))
    limit = int(rng.choice([0, 0, 37, 333, 1000]))
    window, list_len = [(4096, 64), (64, 16), (300, 5)][seed % 3]
    got, info = _run(ccref, nodes, pod, prof, limit, monkeypatch, window, list_len)
    assert (info["swept"] == 0) if sweep == "0" else True
    SWEPT[sweep] = SWEPT.get(sweep, 0) + info["swept"]
    PLACED[sweep] = PLACED.get(sweep, 0) + got.placed


SWEPT, PLACED = {}, {}


@pytest.mark.gpu
def test_sweeps_did_most_of_the_work_where_they_apply(ccref, monkeypatch):
    """Guard against a sweep path that silently never runs: on BASELINE config 5's pod shape (maxSkew 1: every feasible domain is at the
    cap) all but the first placement of a run go through sweeps; and the random cases above (run in the same session) swept too."""
    nodes, pod, prof = c5_single_template(20_000)
    got, info = _run(ccref, nodes, pod, prof, 3000, monkeypatch, 4096, 64)
    assert info["swept"] >= got.placed - 64, info
    if PLACED.get("1", 0) > 2000:  # (only when enough of test_sweeps_random ran in this process: its cases are built to cut rounds short)
        assert SWEPT["1"] > 0, (SWEPT, PLACED)


# ---- round 5: windows on node-range shards (csrc/ccsim_coupled.h "windows on shards", include/ccsim.h ccsim_dist_cw_*) -----------------
@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3, 4, 8])
@pytest.mark.parametrize("n,limit", [(3000, 0), (20_000, 9000), (700, 0)])
def test_c5_pod_shape_in_windows_on_shards(ccref, monkeypatch, world, n, limit):
    """BASELINE config 5's pod shape as one template, the snapshot cut into node-range shards (several engines on the one GPU; the
    all-gather is a device copy): per window the pass over each shard, one exchange of the ranks' window records, the deciding wave
    replicated -- the oracle's log, per-node counts, stop, histogram; far fewer exchanges than placements."""
    from test_gpu_parity import _LocalShards
    nodes, pod, prof = c5_single_template(n)
    ref = ccref.run(prof, nodes, pod, max_limit=limit, threads=8)
    sh = _LocalShards(nodes, pod, prof, world)
    res, log = sh.run(limit, "sequential", max(1, ref.placed))
    assert all(r.placed == ref.placed and r.stop == ref.stop for r in res), ([(r.placed, r.stop) for r in res], ref.placed, ref.stop)
    assert np.array_equal(log[: ref.placed], ref.log)
    assert np.array_equal(np.concatenate([r.per_node_count for r in res]), ref.per_node_count)
    if ref.stop == M.STOP_UNSCHEDULABLE:
        assert np.array_equal(sum(r.hist for r in res), ref.hist)
    for e in sh.engines:
        e.close()
    # as many exchanges as ONE GPU makes node passes for the same run (a window ends where it ends there: exhausted class lists, moved
    # normalization maxima -- with few zones the maxima over the feasible zones alternate late in a run), not one or two per placement
    one = capi.Engine(device=0)
    one.load(nodes, pod, prof)
    got = one.run(max_limit=limit, mode="sequential", log_cap=max(1, ref.placed))
    one.close()
    assert np.array_equal(got.log, ref.log)
    assert 1 <= sh.cw_windows <= got.scans + 8, (sh.cw_windows, got.scans, ref.placed)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(16))
def test_sweep_shapes_on_shards_random(ccref, monkeypatch, seed):
    """The adversarial generator of test_sweeps_random on 2 ... 5 shards: whatever the windows on shards take or decline (a window nobody
    can take sends every rank to one pass per placement alike), the result is the oracle's."""
    from test_gpu_parity import _LocalShards
    rng = np.random.default_rng(8800 + seed)
    nodes, pod, prof = sweep_case(rng, int(rng.integers(200, 3000)))
    limit = int(rng.choice([0, 0, 333, 1000]))
    world = int(rng.integers(2, 6))
    ref = ccref.run(prof, nodes, pod, max_limit=limit)
    sh = _LocalShards(nodes, pod, prof, world)
    res, log = sh.run(limit, "sequential", max(1, ref.placed))
    assert all(r.placed == ref.placed and r.stop == ref.stop for r in res), ([(r.placed, r.stop) for r in res], ref.placed, ref.stop)
    assert np.array_equal(log[: ref.placed], ref.log)
    assert np.array_equal(np.concatenate([r.per_node_count for r in res]), ref.per_node_count)
    if ref.stop == M.STOP_UNSCHEDULABLE:
        assert np.array_equal(sum(r.hist for r in res), ref.hist)
    for e in sh.engines:
        e.close()
