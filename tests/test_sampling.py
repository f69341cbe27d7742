"""percentageOfNodesToScore < 100: the sampled search of findNodesThatPassFilters
(vendor/k8s.io/kubernetes/pkg/scheduler/schedule_one.go:610-723): rotating start index, stop at the
(K+1)-th feasible node, tie-break in feasible-list order.  HIP path (k_scan<SMP=1>, k_smp_prefix,
k_scan<SMP=2>, k_final) against the oracle's literal visiting loop, cycle by cycle."""
import dataclasses

import os

import numpy as np
import pytest

import helpers as H
from cluster_capacity_amd import capi, model as M, report as R, synth


def _check(ccref, nodes, pod, prof, limit):
    ref = ccref.run(prof, nodes, pod, max_limit=limit)
    e = capi.Engine(device=0)
    e.load(nodes, pod, prof)
    got = e.run(max_limit=limit, mode="sequential", log_cap=max(1, ref.placed))
    assert got.placed == ref.placed and got.stop == ref.stop
    assert np.array_equal(got.log, ref.log)
    assert np.array_equal(got.per_node_count, ref.per_node_count)
    assert got.evaluated_total == ref.evaluated_total  # the same nodes were visited, cycle by cycle
    assert got.last_feasible == ref.last_feasible
    # which form ran: a template without topology-coupled plugins takes the resident block summaries (csrc/ccsim_sampled.h, round 5) when
    # the search really samples; everything else (and CCSIM_SB=0) the three node passes per cycle
    import os
    sampled = ccref.num_feasible_nodes_to_find(prof.percentage_of_nodes_to_score, nodes.n) < nodes.n or not (prof.w_taint or prof.w_nodeaffinity or prof.w_fit or prof.w_balanced
                                                                                                        or prof.w_imagelocality or prof.w_topologyspread or prof.w_interpodaffinity)
    info = e.sampled_info()
    coupled = bool(pod.spread) or pod.ipa is not None
    # (a coupled template takes three node passes per cycle, except the one shape csrc/ccsim_sampled_zone.h holds resident -- round 6)
    resident = sampled and os.environ.get("CCSIM_SB", "1") != "0" and (not coupled or info["zone_form"])
    # (every node scored -- a snapshot of <= 100 nodes, or 100 % asked for: the full search on the same summaries, csrc/ccsim_search_full.h)
    full = not sampled and not coupled and os.environ.get("CCSIM_SF", "1") != "0" and nodes.n > 0
    assert info["full_search_form"] == full, info
    resident = resident or full
    assert (got.pass_launches > 0) == resident, (got.pass_launches, resident)
    assert info["resident"] == resident
    if info["zone_form"]:
        assert coupled and info["laps"] >= got.placed  # (`laps` counts this form's cycles)
        return e, got, ref
    if resident and not full:  # ... and a lap of the ring at a time (k_sb_laps, round 6) whenever a block of >= 64 nodes holds one stretch boundary at most
        forced = int(os.environ.get("CCSIM_SB_SHIFT", "6"))  # (blocks of 256 nodes when K >= 256, else of 64; forced: 64 or 256 only)
        assert info["laps_form"] == (info["K"] >= (1 << forced) and forced in (6, 8) and os.environ.get("CCSIM_SB", "1") == "1"), info
        assert not info["laps_form"] or (info["block"] <= info["K"] and info["laps"] > 0)
    if ref.stop == M.STOP_UNSCHEDULABLE:
        assert np.array_equal(got.hist, ref.hist)
        assert got.n_code_unschedulable == ref.n_code_unschedulable
        assert R.stop_reason(got, nodes.n, 0) == R.stop_reason(ref, nodes.n, 0)
    return e, got, ref


def _with_pct(prof, pct):
    return dataclasses.replace(prof, percentage_of_nodes_to_score=pct)


@pytest.mark.gpu
@pytest.mark.parametrize("cfg,n,pct,limit", [("C3", 1000, 0, 0), ("C3", 1000, 0, 300), ("C2", 5000, 10, 400), ("C3", 300, 50, 0),
                                             ("C3", 4096, 5, 0), ("C3", 777, 35, 0), ("C2", 2049, 0, 1000), ("C3", 100, 0, 0),
                                             ("C3", 100_000, 0, 600)])
def test_sampled_search_vs_oracle(ccref, cfg, n, pct, limit):
    nodes, pod, prof = synth.make_config(cfg, n_nodes=n, seed=77 + n)
    e, got, ref = _check(ccref, nodes, pod, _with_pct(prof, pct), limit)
    k = ccref.num_feasible_nodes_to_find(pct, n)
    if k < n:
        assert got.evaluated_total < (got.placed + 1) * n  # the search really stopped early


@pytest.mark.gpu
@pytest.mark.parametrize("knobs", [{"CCSIM_SB": "0"}, {"CCSIM_SB_CYCLES": "3"}, {"CCSIM_SB_CYCLES": "1"}, {"CCSIM_SB": "2"}, {"CCSIM_SB": "2", "CCSIM_SB_CYCLES": "3"},
                                   {"CCSIM_SB_SHIFT": "6"}, {"CCSIM_SB_SHIFT": "8", "CCSIM_SB_SLOW_FLOOR": "0"}, {"CCSIM_SB_SHIFT": "6", "CCSIM_SB_SLOW_FLOOR": "0", "CCSIM_SB_CYCLES": "7"}],
                         ids=["three-passes", "3-cycles-per-launch", "1-cycle-per-launch", "cycle-at-a-time", "cycle-at-a-time-3-per-launch", "blocks-of-64", "blocks-of-256-rebuild-early",
                              "blocks-of-64-rebuild-early-7-per-launch"])
@pytest.mark.parametrize("cfg,n,pct,limit", [("C3", 1000, 0, 0), ("C2", 5000, 10, 400), ("C3", 4096, 5, 0), ("C3", 777, 35, 0)])
def test_sampled_search_forms_agree(ccref, monkeypatch, knobs, cfg, n, pct, limit):
    """The three-pass cycle (what the SchedulePod seam, shards and coupled templates still take), the resident forms -- a lap of the ring at a
    time (the default) and a cycle at a time (round 5's, CCSIM_SB=2) -- relaunched every few cycles (the hand-over of the summaries, the start
    index, the assumed maxima between launches; a lap cut short by the launch's cycle budget), other block sizes, and the rebuild taken
    instead of the node-by-node re-evaluation as soon as a stretch's maxima differ, against the oracle."""
    for k, v in knobs.items():
        monkeypatch.setenv(k, v)
    nodes, pod, prof = synth.make_config(cfg, n_nodes=n, seed=77 + n)
    _check(ccref, nodes, pod, _with_pct(prof, pct), limit)


@pytest.mark.gpu
@pytest.mark.parametrize("cfg,n,pct,limit,floor", [("C3", 20_000, 5, 2500, None), ("C3", 20_000, 5, 2500, 0), ("C2", 60_000, 2, 3000, None), ("C3", 3000, 5, 0, None),
                                                   ("C3", 3000, 5, 0, 0), ("C3", 600, 0, 0, None)])
def test_sampled_search_many_laps_and_wraps(ccref, monkeypatch, cfg, n, pct, limit, floor):
    """Laps of ~19 (49 at 2 %) independent cycles; every lap goes once round the ring, so a run is as many wraps as laps; 3000 nodes to the
    end: the laps shrink to one stretch and then to the degenerate lap that visits every node (F <= K).  With and without the
    node-by-node re-evaluation of stretches whose maxima differ from the assumed ones (floor 0: N / 4 nodes of them end the lap)."""
    if floor is not None:
        monkeypatch.setenv("CCSIM_SB_SLOW_FLOOR", str(floor))
    nodes, pod, prof = synth.make_config(cfg, n_nodes=n, seed=21 + n)
    e, got, ref = _check(ccref, nodes, pod, _with_pct(prof, pct), limit)
    info = e.sampled_info()
    assert info["laps_form"] and info["laps"] >= 3
    if n >= 20_000:
        k = ccref.num_feasible_nodes_to_find(pct, n)
        assert info["laps"] <= 3 * (1 + got.placed * k // n)  # the laps really held ~ N / K cycles each
    e.close()


@pytest.mark.gpu
def test_sampled_search_a_node_wins_in_consecutive_laps(ccref):
    """Few nodes far better than the rest: the same node wins the stretch it lies in lap after lap -- its row and memo word are rewritten and
    re-read every few microseconds by different lanes (the L1 / L2 visibility the kernel's fences are there for)."""
    n = 4000
    nodes, pod, prof = synth.make_config("C3", n_nodes=n, seed=9)
    big = np.arange(0, n, 157)
    for c in range(len(nodes.alloc)):
        nodes.alloc[c][big] *= 40
    nodes.alloc_pods[big] = 4000
    e, got, ref = _check(ccref, nodes, pod, _with_pct(prof, 5), 6000)
    assert np.max(got.per_node_count) >= 40
    e.close()


def _zone_template(n, zones=None, max_skew=1, min_domains=1, anti=True, seed=5, cfg="C3"):
    """BASELINE config 5's pod shape as ONE template: zone DoNotSchedule spread (+ required hostname anti-affinity against its own clones)."""
    nodes, pod, prof = synth.make_config(cfg, n_nodes=n, seed=seed)
    pod.spread = [synth.zone_spread(n, max_skew=max_skew, min_domains=min_domains)]
    if zones is not None:  # the same nodes dealt to `zones` zones
        nodes.label_cols[pod.spread[0].col] = (np.arange(n) % zones + 1).astype(np.int32)
        pod.spread[0].n_domains = zones
    if anti:
        nodes.label_cols.append(np.arange(1, n + 1, dtype=np.int32))  # kubernetes.io/hostname
        pod.ipa = M.InterPodAffinity(key_cols=[len(nodes.label_cols) - 1], key_ndom=[n], anti_keys=[0], anti_self=[True], anti_existing=[None])
    return nodes, pod, prof


@pytest.mark.gpu
@pytest.mark.parametrize("n,zones,pct,limit,skew,anti", [(1000, None, 0, 0, 1, True), (1000, None, 0, 0, 2, False), (5000, 7, 10, 900, 1, True), (20_000, 64, 5, 1500, 1, True),
                                                         (4096, 3, 5, 0, 3, True), (777, 64, 35, 0, 1, True), (100_000, None, 0, 700, 1, True), (3000, 16, 5, 2500, 1, False)])
def test_sampled_zone_form_vs_oracle(ccref, n, zones, pct, limit, skew, anti):
    """The reference's default percentage for a template with a hard zone spread constraint (+ hostname anti-affinity): per-(block, zone) entries
    under the mask of eligible zones (csrc/ccsim_sampled_zone.h) against the oracle's visiting loop -- log, nodes visited, feasible counts, and
    the FitError histogram where the run ends Unschedulable."""
    nodes, pod, prof = _zone_template(n, zones, max_skew=skew, anti=anti, seed=40 + n)
    e, got, ref = _check(ccref, nodes, pod, _with_pct(prof, pct), limit)
    assert e.sampled_info()["zone_form"], e.sampled_info()
    e.close()


@pytest.mark.gpu
def test_sampled_zone_form_nodes_without_the_key_min_domains_and_existing_pods(ccref):
    rng = np.random.default_rng(77)
    n = 2500
    nodes, pod, prof = _zone_template(n, 5, max_skew=1, min_domains=7, anti=True, seed=3)  # fewer zones than minDomains: the global minimum reads 0
    col = pod.spread[0].col
    nodes.label_cols[col][rng.random(n) < 0.1] = 0                                        # some nodes lack the key: never feasible
    pod.spread[0].node_match_count = (rng.random(n) < 0.05).astype(np.int32) * rng.integers(1, 4, n).astype(np.int32)  # matching pods already there
    pod.spread[0].node_included = (rng.random(n) < 0.9).astype(np.uint8)                   # node inclusion policies: some nodes are not counted
    e, got, ref = _check(ccref, nodes, pod, _with_pct(prof, 5), 0)
    assert e.sampled_info()["zone_form"]
    e.close()


@pytest.mark.gpu
@pytest.mark.parametrize("knobs", [{"CCSIM_SZ": "0"}, {"CCSIM_SB_CYCLES": "1"}, {"CCSIM_SB_CYCLES": "5"}], ids=["three-passes", "1-cycle-per-launch", "5-cycles-per-launch"])
def test_sampled_zone_forms_agree(ccref, monkeypatch, knobs):
    for k, v in knobs.items():
        monkeypatch.setenv(k, v)
    for n, zones, pct, limit in ((1500, 9, 10, 0), (6000, 64, 5, 1200)):
        nodes, pod, prof = _zone_template(n, zones, seed=90 + n)
        e, got, ref = _check(ccref, nodes, pod, _with_pct(prof, pct), limit)
        assert e.sampled_info()["zone_form"] == ("CCSIM_SZ" not in knobs)
        e.close()


@pytest.mark.gpu
def test_sampled_zone_form_1m_nodes_64_zones(ccref):
    """BASELINE's 1M-node cluster, the generator's own 64 zones, the flag left unset (what both hosts run for this template): 1300 cycles = 20 rounds
    of the spread constraint, the search sampling 5 % while many zones are eligible and visiting every node for the last zones of a round."""
    nodes, pod, prof = _zone_template(1_000_000, None, cfg="C3", seed=5)
    e, got, ref = _check(ccref, nodes, pod, _with_pct(prof, 0), 1300)
    assert e.sampled_info()["zone_form"]
    e.close()


@pytest.mark.gpu
def test_sampled_search_1m_nodes_adaptive_default(ccref):
    """BASELINE's headline snapshot under the reference's DEFAULT configuration (percentageOfNodesToScore 0 -> 5 % at 1M nodes: K = 50 000
    of the rotating visiting order per cycle, schedule_one.go:697-723): 3 000 cycles against the oracle's visiting loop."""
    nodes, pod, prof = synth.make_config("C4", n_nodes=1_000_000)
    e, got, ref = _check(ccref, nodes, pod, _with_pct(prof, 0), 3000)
    assert got.evaluated_total < 3000 * 80_000  # (K = 50 000 kept + the infeasible nodes met on the way: far from every node)
    e.close()


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(12))
def test_sampled_search_random_plugin_mix(ccref, seed):
    rng = np.random.default_rng(3000 + seed)
    nodes, pod, prof = H.random_case(rng, int(rng.integers(100, 1500)))
    prof = _with_pct(prof, int(rng.choice([0, 10, 35, 70, 99])))
    _check(ccref, nodes, pod, prof, int(rng.choice([0, 0, 37, 500])))


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(12))
def test_sampled_search_with_topology_coupled_plugins(ccref, seed):
    # PreFilter / PreScore state covers ALL nodes (filtering.go:235-308), Filter and Score only the sampled ones
    rng = np.random.default_rng(3100 + seed)
    nodes, pod, prof = H.random_case(rng, int(rng.integers(100, 900)))
    if seed % 3 != 2:
        cons = H.random_spread(rng, nodes, n_constraints=2)
        for c in cons:
            c.hard = bool(rng.integers(0, 2))
        pod.spread = cons
    if seed % 3 != 0:
        pod.ipa = H.random_ipa(rng, nodes)
    prof = _with_pct(prof, int(rng.choice([0, 20, 60])))
    _check(ccref, nodes, pod, prof, int(rng.choice([0, 0, 70])))


@pytest.mark.gpu
def test_sampled_search_scalar_resources(ccref):
    from test_gpu_parity import _scalar_case
    for n_scalar, seed in [(1, 0), (3, 1), (8, 2)]:
        nodes, pod = _scalar_case(np.random.default_rng(3200 + seed), 1200, n_scalar)
        _check(ccref, nodes, pod, _with_pct(M.Profile.default(), 0), 0)


@pytest.mark.gpu
def test_sampled_schedule_one_reports_visited_nodes(ccref):
    n = 2000
    nodes, pod, prof = synth.make_config("C3", n_nodes=n, seed=11)
    prof = _with_pct(prof, 0)
    k = ccref.num_feasible_nodes_to_find(0, n)
    ref = ccref.run(prof, nodes, pod, max_limit=150)
    e = capi.Engine(device=0)
    e.load(nodes, pod, prof)
    total = 0
    for r in range(150):
        node, evaluated, feasible = e.schedule_one()
        assert node == ref.log[r]
        assert feasible == k and k <= evaluated < n
        total += evaluated
    assert total == ref.evaluated_total


@pytest.mark.gpu
def test_sampled_search_continues_across_runs(ccref):
    # nextStartNodeIndex is scheduler state: two runs of 100 == one run of 200
    nodes, pod, prof = synth.make_config("C3", n_nodes=1500, seed=12)
    prof = _with_pct(prof, 0)
    ref = ccref.run(prof, nodes, pod, max_limit=200)
    e = capi.Engine(device=0)
    e.load(nodes, pod, prof)
    a = e.run(max_limit=100, mode="sequential", log_cap=100)
    b = e.run(max_limit=100, mode="sequential", log_cap=100)
    assert np.array_equal(np.concatenate([a.log, b.log]), ref.log)
    e.reset_state()
    c = e.run(max_limit=200, mode="sequential", log_cap=200)
    assert np.array_equal(c.log, ref.log)


def _check_sharded(ccref, nodes, pod, prof, limit, world):
    """The sampled search on `world` node-range shards (two exchanges per cycle: counts, then the max-loc; DevState::smp_phase),
    driven shard by shard on one GPU: same log, same stop, same nodes visited cycle by cycle as the oracle's visiting loop."""
    from test_gpu_parity import _LocalShards
    ref = ccref.run(prof, nodes, pod, max_limit=limit)
    res, log = _LocalShards(nodes, pod, prof, world).run(limit, "sequential", max(1, ref.placed))
    assert all(r.placed == ref.placed and r.stop == ref.stop for r in res), ([(r.placed, r.stop) for r in res], ref.placed, ref.stop)
    assert np.array_equal(log[: ref.placed], ref.log)
    assert np.array_equal(np.concatenate([r.per_node_count for r in res]), ref.per_node_count)
    assert all(r.evaluated_total == ref.evaluated_total for r in res), ([r.evaluated_total for r in res], ref.evaluated_total)
    assert all(r.last_feasible == ref.last_feasible for r in res)
    if ref.stop == M.STOP_UNSCHEDULABLE:
        assert np.array_equal(sum(r.hist for r in res), ref.hist)
    return res, ref


# (whole runs to Unschedulable -- three node passes and two exchanges per cycle -- on two shards only; the other shard counts on bounded runs)
@pytest.mark.gpu
@pytest.mark.parametrize("world,cfg,n,pct,limit",
                         [(w, *c) for c in [("C3", 1000, 0, 300), ("C2", 5000, 10, 400), ("C3", 300, 50, 0), ("C2", 2049, 0, 1000), ("C3", 100, 0, 0)] for w in (1, 2, 3, 5)] +
                         [(2, "C3", 1000, 0, 0), (2, "C3", 777, 35, 0), (1, "C3", 777, 35, 1500), (3, "C3", 1000, 0, 1500), (5, "C3", 777, 35, 1500), (5, "C3", 1000, 0, 1500)])
def test_sampled_search_on_shards_vs_oracle(ccref, world, cfg, n, pct, limit):
    nodes, pod, prof = synth.make_config(cfg, n_nodes=n, seed=77 + n)
    res, ref = _check_sharded(ccref, nodes, pod, _with_pct(prof, pct), limit, world)
    if ccref.num_feasible_nodes_to_find(pct, n) < n:
        assert res[0].evaluated_total < (ref.placed + 1) * n  # the search really stopped early


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(16))
def test_sampled_search_on_shards_random_plugin_mix(ccref, seed):
    rng = np.random.default_rng(3300 + seed)
    nodes, pod, prof = H.random_case(rng, int(rng.integers(100, 1500)))
    prof = _with_pct(prof, int(rng.choice([0, 10, 35, 70, 99])))
    _check_sharded(ccref, nodes, pod, prof, int(rng.choice([0, 0, 37, 500])), int(rng.integers(1, 6)))


@pytest.mark.gpu
def test_sampled_search_on_shards_k1_and_refusals(ccref):
    # a profile without Score plugins keeps the first feasible node of the visiting order (K = 1) -- on shards too
    nodes, pod, prof = synth.make_config("C3", n_nodes=1500, seed=31)
    bare = dataclasses.replace(prof, w_taint=0, w_nodeaffinity=0, w_fit=0, w_balanced=0, w_imagelocality=0, w_topologyspread=0, w_interpodaffinity=0)
    _check_sharded(ccref, nodes, pod, bare, 700, 3)
    # topology-coupled plugins on shards under a percentage: part of the protocol since round 6 (below); the knob restores the refusal
    from test_gpu_parity import _LocalShards
    pod.spread = [synth.zone_spread(1500, max_skew=2)]
    os.environ["CCSIM_DIST_SMP_COUPLED"] = "0"
    try:
        with pytest.raises(capi.CcsimError):
            _LocalShards(nodes, pod, _with_pct(prof, 0), 2).run(50, "sequential", 50)
    finally:
        del os.environ["CCSIM_DIST_SMP_COUPLED"]


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(int(os.environ.get("CC_TEST_SEEDS", "24"))))
def test_sampled_search_on_shards_with_topology_coupled_plugins(ccref, seed):
    """Round 6 (VERDICT r5 missing #6): percentageOfNodesToScore < 100 on node-range shards for a template WITH topology-coupled plugins --
    hard and soft spread constraints, inter-pod (anti-)affinity.  The Filter state is the replicated tables; the counting pass filters
    with the assumed global minimum and the scoring pass verifies it (a stale one sends the cycle back to the counting pass); the
    PreScore facts are gathered over the selected nodes only.  Same log, stop, visited nodes per cycle and histogram as the oracle."""
    rng = np.random.default_rng(8800 + seed)
    nodes, pod, prof = H.random_case(rng, int(rng.integers(100, 1200)))
    kind = seed % 4
    if kind in (0, 1, 3):
        pod.spread = H.random_spread(rng, nodes, n_constraints=int(rng.integers(1, 3)))
        if kind == 3:  # ScheduleAnyway constraints beside (or instead of) the hard ones
            for k in pod.spread:
                k.hard = bool(rng.integers(0, 2))
            pod.spread[-1].hard = False
    if kind in (1, 2) or (kind == 3 and seed % 8 == 3):
        pod.ipa = H.random_ipa(rng, nodes)
    prof = _with_pct(prof, int(rng.choice([0, 10, 35, 70])))
    limit = int(rng.choice([0, 60, 400]))
    ref = ccref.run(prof, nodes, pod, max_limit=limit)
    if ref.placed > 1200:
        limit = 1200
    e_nodes, e_pod = M.relax_soft(nodes, pod)  # (the engine form of requireAllTopologies = false, derived on the whole snapshot)
    from test_gpu_parity import _LocalShards
    ref = ccref.run(prof, nodes, pod, max_limit=limit)
    world = int(rng.integers(1, 5))
    res, log = _LocalShards(e_nodes, e_pod, prof, world).run(limit, "sequential", max(1, ref.placed))
    assert all(r.placed == ref.placed and r.stop == ref.stop for r in res), ([(r.placed, r.stop) for r in res], ref.placed, ref.stop, world)
    n = min(len(log), len(ref.log))
    first = next((i for i in range(n) if log[i] != ref.log[i]), None)
    assert first is None, ("first differing placement", first, world, log[max(0, first - 2): first + 3].tolist(), ref.log[max(0, first - 2): first + 3].tolist())
    assert np.array_equal(np.concatenate([r.per_node_count for r in res]), ref.per_node_count)
    assert all(r.evaluated_total == ref.evaluated_total for r in res), ([r.evaluated_total for r in res], ref.evaluated_total)
    if ref.stop == M.STOP_UNSCHEDULABLE:
        assert np.array_equal(sum(r.hist for r in res), ref.hist)


@pytest.mark.gpu
def test_sampled_search_is_sequential_only(ccref):
    nodes, pod, prof = synth.make_config("C3", n_nodes=500, seed=13)
    e = capi.Engine(device=0)
    e.load(nodes, pod, _with_pct(prof, 0))
    with pytest.raises(capi.CcsimError):
        e.run(mode="batched")
    # fewer than 100 nodes: every node is visited whatever the percentage (schedule_one.go:703-705): batched is valid
    nodes, pod, prof = synth.make_config("C3", n_nodes=90, seed=14)
    ref = ccref.run(_with_pct(prof, 0), nodes, pod)
    e = capi.Engine(device=0)
    e.load(nodes, pod, _with_pct(prof, 0))
    got = e.run(mode="batched")
    assert got.placed == ref.placed and np.array_equal(got.per_node_count, ref.per_node_count)


@pytest.mark.gpu
@pytest.mark.parametrize("cfg,n,limit", [("C3", 1000, 0), ("C3", 50, 0), ("C2", 3000, 700), ("C3", 4097, 2500)])
def test_profile_without_score_plugins_keeps_the_first_feasible_node(ccref, cfg, n, limit):
    """schedule_one.go:619-621: no Score plugin -> numNodesToFind = 1, for every snapshot size: the first feasible node of the
    rotating visiting order is bound, the search stops at the second feasible node, nextStartNodeIndex moves past it."""
    nodes, pod, prof = synth.make_config(cfg, n_nodes=n, seed=5 + n)
    prof = dataclasses.replace(prof, w_taint=0, w_nodeaffinity=0, w_fit=0, w_balanced=0, w_topologyspread=0, w_interpodaffinity=0, w_imagelocality=0)
    e, got, ref = _check(ccref, nodes, pod, prof, limit)
    if limit == 0:
        assert got.placed > n  # the rotation spreads clones round the cluster instead of filling node 0 first
    e.close()
    # ... and at the SchedulePod seam, cycle by cycle
    e = capi.Engine(device=0)
    e.load(nodes, pod, prof)
    for i in range(min(200, ref.placed)):
        assert e.schedule_one()[0] == ref.log[i]
    e.close()


@pytest.mark.gpu
@pytest.mark.parametrize("n,zones,pct,cycles,anti", [(20_000, 16, 5, 500, True), (1500, 5, 10, 0, False)])
def test_schedule_one_takes_the_zone_form(ccref, n, zones, pct, cycles, anti):
    """The SchedulePod seam (scheduler.go:88-91) for a template with a hard zone constraint under the default percentage: one launch on the
    per-(block, zone) entries per call, to the FitError and past it."""
    nodes, pod, prof = _zone_template(n, zones, anti=anti, seed=9)
    prof = _with_pct(prof, pct)
    ref = ccref.run(prof, nodes, pod, max_limit=cycles, threads=16)
    e = capi.Engine(device=0)
    e.load(nodes, pod, prof)
    ev = 0
    for r in range(ref.placed):
        node, evaluated, feasible = e.schedule_one()
        assert node == ref.log[r], r
        ev += evaluated
    info = e.sampled_info()
    assert info["zone_form"] and info["launches"] >= ref.placed, info
    if ref.stop == M.STOP_UNSCHEDULABLE:
        for _ in range(3):
            node, evaluated, feasible = e.schedule_one()
            assert node == -1 and feasible == 0
            ev += evaluated if _ == 0 else 0
    assert ev == ref.evaluated_total
    e.close()


@pytest.mark.gpu
@pytest.mark.parametrize("cfg,n,pct,handover", [("C3", 3000, 5, True), ("C3", 3000, 5, False), ("C4", 5000, 10, True), ("C2", 700, 30, True), ("C3", 40_000, 2, True)])
def test_sampled_search_hands_over_to_the_full_search_at_the_end(ccref, monkeypatch, cfg, n, pct, handover):
    """To the end of a run: once fewer feasible nodes are left than the search keeps, every node is visited and the start index stays
    (schedule_one.go:538) -- a lap is one cycle then, and the full search's kernel (k_sf_cycles, ring order from that start index) takes
    the rest of the run; CCSIM_SB_HANDOVER=0 keeps the lap kernel's one-stretch laps.  Same log, visited nodes, FitError either way."""
    if not handover:
        monkeypatch.setenv("CCSIM_SB_HANDOVER", "0")
    nodes, pod, prof = synth.make_config(cfg, n_nodes=n, seed=5 + n)
    if n >= 40_000:  # (a whole run at this size is long for the oracle: nodes nearly full from the start)
        nodes.pod_count = np.maximum(nodes.pod_count, nodes.alloc_pods - 3).astype(nodes.pod_count.dtype)
    e, got, ref = _check(ccref, nodes, pod, _with_pct(prof, pct), 0)
    info = e.sampled_info()
    assert ref.stop == M.STOP_UNSCHEDULABLE and info["laps_form"] and info["handed_over_to_full_search"] == handover, info
    e.close()


@pytest.mark.gpu
@pytest.mark.parametrize("world,n,limit", [(2, 20_000, 400), (4, 6_000, 0), (3, 50_000, 150)])
def test_sampled_search_on_shards_config5_pod_shape(ccref, world, n, limit):
    """BASELINE config 5's pod shape as one template (zone DoNotSchedule spread + required hostname anti-affinity) under the reference's
    DEFAULT percentage on node-range shards: what `--gpus N` refused until round 6."""
    nodes, pod, prof = synth.make_config("C3", n_nodes=n, seed=5 + n)
    nodes.label_cols.append(np.arange(1, nodes.n + 1, dtype=np.int32))  # kubernetes.io/hostname
    pod.ipa = M.InterPodAffinity(key_cols=[len(nodes.label_cols) - 1], key_ndom=[nodes.n], anti_keys=[0], anti_self=[True], anti_existing=[None])
    pod.spread = [synth.zone_spread(nodes.n, max_skew=1)]
    prof = _with_pct(prof, 0)
    from test_gpu_parity import _LocalShards
    ref = ccref.run(prof, nodes, pod, max_limit=limit, threads=8)
    res, log = _LocalShards(nodes, pod, prof, world).run(limit, "sequential", max(1, ref.placed))
    assert all(r.placed == ref.placed and r.stop == ref.stop for r in res)
    assert np.array_equal(log[: ref.placed], ref.log)
    assert all(r.evaluated_total == ref.evaluated_total for r in res), ([r.evaluated_total for r in res], ref.evaluated_total)
    assert res[0].evaluated_total < (ref.placed + 1) * n or ref.placed == 0
    if ref.stop == M.STOP_UNSCHEDULABLE:
        assert np.array_equal(sum(r.hist for r in res), ref.hist)
