"""The oracle's unit arithmetic against the reference's OWN arithmetic: tests/golden/reference_vectors.json holds inputs and outputs
obtained by executing a mechanical, line-by-line transliteration of the reference's Go functions (leastRequestedScore and the
leastResourceScorer closure, balancedResourceScorer, DefaultNormalizeScore, numFeasibleNodesToFind, calculatePriority,
scaledImageScore) -- tests/golden/make_reference_vectors.py, which records the Go text, its file and line, and the Python it became.
No Go toolchain exists in the build image; this is as close as the oracle can get to "checked against outputs of the reference"
for its scoring arithmetic.  Not covered: math.Log (Go's own implementation; the oracle restates it) and everything that is control
flow over scheduler state rather than arithmetic."""
import json
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIX = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_vectors.json")))
VEC = FIX["vectors"]


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="the reference tree exists in the build container only")
def test_fixture_is_what_the_reference_sources_give():
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_reference_vectors", os.path.join(ROOT, "tests", "golden", "make_reference_vectors.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    env, sources = mod.build()
    assert sources == FIX["sources"]          # the Go text (and so its transliteration) is still what the fixture was made from
    assert mod.vectors(env) == VEC


def test_transliterations_are_line_by_line():
    """Audit aid: every non-blank, non-comment Go line of a function became exactly one Python line (or a brace)."""
    for name, s in FIX["sources"].items():
        opts = s.get("opts", {})
        if opts.get("oneline"):  # `func ... { return X }`: one line, one line
            assert len(s["python"].strip().split("\n")) == 2, name
            continue
        lines = s["go"].split("\n")
        lines = lines if opts.get("block") else lines[opts.get("header", 1):-1]  # (a block is cut with its own first and last line)
        go = [l.strip() for l in lines if l.strip() and not l.strip().startswith("//") and l.strip() != "}"]
        py = [l for l in s["python"].split("\n")[1:] if l.strip()]
        dropped = 5 if name in ("ptsNormalizeScore", "ipaNormalizeScore") else 3 if name == "ipaFilter" else 4 if name in ("ptsFilter", "ptsScore", "ipaScore") else 0  # the cycle-state preamble (make_reference_vectors.DROP), minus its braces
        if name == "volumeZoneFilter":
            dropped = 12  # logger, the no-volume fast path, the PreFilter state or its fallback (15 lines of DROP, 3 of them bare braces)
        two_value_lookups = sum(", ok := " in l or ":= ls.Lookup(" in l for l in go)  # `v, ok := m[k]` becomes a membership test plus a .get: one line more
        two_value_lookups += sum(l.startswith("if ") and ", ok := " in l and not l.startswith("if _, ok") for l in go)  # `if v, ok := m[k]; ok {`: the lookup's two lines, then the test
        two_value_lookups += 2 * sum(bool(re.match(r"if (\w+), (?!ok\b)(\w+) := .+\[[\w.]+\]; \2 \{$", l)) for l in go)  # the same with another flag name (`tpValueExist`, `exist`)
        two_value_lookups += sum(bool(re.match(r"if (\w+, _|_, \w+) := .*; !?\w+ \{$", l)) and ", ok := " not in l for l in go)  # `if match, _ := f(x); !match {`: the call, then the test
        two_value_lookups += sum(bool(re.match(r"for \w+ := .+; ; \w+ = ", l)) for l in go)  # `for x := f(); ; x = f() {`: `while True:` + the call
        named_result = 1 if re.search(r"\) \(\w+ [\[\]\w.]+\) \{$", s["go"].split("\n")[0]) else 0  # `(n int)`: one line that sets its zero value
        joined = s.get("joined", 0)  # lines absorbed into the one before: a condition continued after && / ||, a composite literal's fields
        closers = 1 if name == "findNodesThatPassFilters_checkNode" else 0  # the `})` that ends the func() { ... } argument of SendErrorWithCancel: a brace
        if name == "volumeZoneFilter":
            two_value_lookups += 2  # `if _, ok := node.Labels[k]; ok {` in its three-line form; `v, ok = m[k]` (an assignment, not a declaration): lookup + .get
        if name == "haveOverlap":
            two_value_lookups -= 1  # `if _, ok := m[val]; ok {`: m.Has(val), one line (make_reference_vectors.REWRITE)
        if name == "HostPortInfo_CheckConflict":
            two_value_lookups -= 2  # `if _, ok := m[*pp]; ok {` twice: a membership test on the (protocol, port) pair, one line each (make_reference_vectors.REWRITE)
        assert len(go) - dropped + two_value_lookups + named_result - joined - closers == len(py), name


def test_least_allocated(ccref):
    for rq, al, w, want in VEC["leastResourceScorer"]:
        assert ccref.least_allocated(rq, al, w) == want, (rq, al, w)


def test_balanced_allocation(ccref):
    for rq, al, want in VEC["balancedResourceScorer"]:
        assert ccref.balanced_allocation(rq, al) == want, (rq, al)


def test_default_normalize(ccref):
    for sc, reverse, want in VEC["DefaultNormalizeScore"]:
        assert list(ccref.default_normalize(100, reverse, sc)) == want, (sc, reverse)


def test_num_feasible_nodes_to_find(ccref):
    for pct, n, want in VEC["numFeasibleNodesToFind"]:
        # (-1: the profile leaves percentageOfNodesToScore unset and the global value, 0 = adaptive, applies)
        assert ccref.num_feasible_nodes_to_find(max(pct, 0), n) == want, (pct, n)


def test_image_locality(ccref):
    for sizes, nn, total, ncont, want in VEC["imageLocality"]:
        assert ccref.image_locality_score(sizes, nn, total, ncont) == want, (sizes, nn, total, ncont)


def test_pts_normalize(ccref):
    for sc, ignored, want in VEC["ptsNormalizeScore"]:
        assert ccref.pts_normalize(sc, ignored) == want, (sc, ignored)


def test_ipa_normalize(ccref):
    for sc, want in VEC["ipaNormalizeScore"]:
        assert ccref.ipa_normalize(sc) == want, sc


# ---- round 3: the loop-level pieces ----------------------------------------------------------------------------
def _intern(values):
    """Topology values -> value ids the way the ingests intern them (0 = the node lacks the key)."""
    ids = {}
    col = []
    for v in values:
        col.append(0 if v is None else ids.setdefault(v, len(ids) + 1))
    return col, ids


def _plain_nodes(n, label_cols):
    import numpy as np
    import helpers as H
    return H.simple_nodes([4000] * n, [8 << 30] * n, [110] * n, label_cols=[np.array(c, np.int32) for c in label_cols])


def test_ipa_count_maps_vs_update_with_terms(ccref):
    """interpodaffinity/filtering.go:111-145 (update, updateWithAffinityTerms, updateWithAntiAffinityTerms) driven over every existing pod, as
    PreFilter drives them, against the oracle's ipa_build: the same topology-pair counts."""
    import numpy as np
    import helpers as H
    from cluster_capacity_amd import model as M
    for labels, aff_terms, anti_terms, pods, aff_want, anti_want in VEC["ipaCountMaps"]:
        n = len(labels)
        cols, ids = zip(*[_intern([lb.get(k) for lb in labels]) for k in ("zone", "host")])
        nodes = _plain_nodes(n, cols)
        key = {"zone": 0, "host": 1}
        pod = H.simple_pod(100, 64 << 20)
        pod.ipa = M.InterPodAffinity(
            key_cols=[0, 1], key_ndom=[max(len(ids[0]), 1), max(len(ids[1]), 1)], aff_keys=[key[k] for k in aff_terms], self_aff=False,
            aff_existing=np.array([sum(1 for m_all, _ in pods[i] if m_all and aff_terms) for i in range(n)], np.int32),
            anti_keys=[key[k] for k in anti_terms], anti_self=[False] * len(anti_terms),
            anti_existing=[np.array([sum(1 for _, m_anti in pods[i] if m_anti[t]) for i in range(n)], np.int32) for t in range(len(anti_terms))],
            exist_anti=[None, None], score_existing=[None, None], score_self=[0, 0], self_entries=[0, 0], entries_existing=0)
        tabs, totals = ccref.unit_ipa_build(nodes, pod)
        for want, which in ((aff_want, 0), (anti_want, 1)):
            got = {}
            for k, name in enumerate(("zone", "host")):
                inv = {v: s for s, v in ids[k].items()}
                for vid, cnt in enumerate(tabs[k][which]):
                    if vid and cnt:
                        got[(name, inv[vid])] = cnt
            assert got == {(a, b): c for a, b, c in want}, (labels, aff_terms, anti_terms, pods)
        assert totals[0] == sum(c for _, _, c in aff_want)  # len(affinityCounts) is only ever tested against 0; the oracle keeps the entry total


def test_cal_prefilter_state(ccref):
    """podtopologyspread/filtering.go:235-308 (calPreFilterState: processNode closure, merge, critical paths; with nodeLabelsMatchSpreadConstraints,
    countPodsMatchSelector, matchNodeInclusionPolicies, criticalPaths.update) against the oracle's pts_prefilter.  The per-node inputs the oracle
    takes -- matching pods, node inclusion -- are derived here the way the ingests derive them."""
    import numpy as np
    import helpers as H
    from cluster_capacity_amd import ingest, model as M
    for gate, cons, nodes_, tolerations, want_maps, want_min in VEC["calPreFilterState"]:
        n = len(nodes_)
        keys = sorted({c["key"] for c in cons})
        cols, ids = zip(*[_intern([nd["labels"].get(k) for nd in nodes_]) for k in keys])
        nodes = _plain_nodes(n, cols)
        low = lambda d: {k.lower(): v for k, v in d.items() if v != ""}
        tols = [low(t) for t in tolerations]
        spread = []
        for j, c in enumerate(cons):
            counts = [0 if c["emptySelector"] else sum(1 for p in nd["pods"] if p["match"][j] and p["ns"] == "default" and not p["terminating"]) for nd in nodes_]
            inc = []
            for nd in nodes_:
                untolerated = not ingest.taint_verdict([low(t) for t in nd["taints"]], tols)[0]
                if gate:
                    ok = (c["affinityPolicy"] != "Honor" or nd["affinityMatch"]) and (c["taintsPolicy"] != "Honor" or not untolerated)
                else:
                    ok = nd["affinityMatch"]  # the gate off: required node affinity decides for every constraint (filtering.go:259-265)
                inc.append(1 if ok else 0)
            ki = keys.index(c["key"])
            spread.append(M.SpreadConstraint(col=ki, max_skew=1, min_domains=1, hard=True, self_match=True, n_domains=max(len(ids[ki]), 1),
                                             node_match_count=np.array(counts, np.int32), node_included=np.array(inc, np.uint8)))
        pod = H.simple_pod(100, 64 << 20)
        pod.spread = spread
        got = ccref.unit_pts_prefilter(nodes, pod)
        for j, c in enumerate(cons):
            ki = keys.index(c["key"])
            inv = {v: s for s, v in ids[ki].items()}
            match_num, mn, ndom = got[j]
            have = {inv[vid]: cnt for vid, cnt in enumerate(match_num) if vid and cnt >= 0}
            assert have == {k: v for k, v in want_maps[j]}, (gate, cons, nodes_, j)
            assert mn == want_min[j] and ndom == len(want_maps[j]), (gate, cons, nodes_, j)


@pytest.mark.parametrize("relaxed", [False, True], ids=["requireAllTopologies", "system-defaults"])
def test_pts_prescore_and_score(ccref, relaxed):
    """podtopologyspread/scoring.go:61-265 -- initPreScoreState's loop over the filtered nodes, the weights, PreScore's closure over all nodes,
    Score, NormalizeScore -- against the oracle's pts_scores: the same ignored nodes (score 0), the same log(size + 2) weights bit for bit, the
    same raw (math.Round'ed) and normalized scores.  The per-node inputs the oracle takes are derived the way the ingests derive them.
    `relaxed`: the Go text driven with requireAllTopologies = false (scoring.go:140, the plugin's system default constraints) against the
    oracle's soft_relaxed branch: nobody ignored, "" counted as a domain, no credit for a constraint whose key the node lacks."""
    import numpy as np
    import helpers as H
    from cluster_capacity_amd import ingest, model as M
    host = "kubernetes.io/hostname"
    assert not relaxed or not any(any(row[5]) for row in VEC["ptsPreScoreScoreRelaxed"])  # (IgnoredNodes stays empty)
    assert relaxed or any(any(row[5]) for row in VEC["ptsPreScoreScore"])
    for gate, cons, nodes_, tolerations, filtered, ignored, weights_hex, raw, norm in VEC["ptsPreScoreScoreRelaxed" if relaxed else "ptsPreScoreScore"]:
        n = len(nodes_)
        keys = sorted({c["key"] for c in cons})
        cols, ids = zip(*[_intern([nd["labels"].get(k) for nd in nodes_]) for k in keys])
        nodes = _plain_nodes(n, cols)
        low = lambda d: {k.lower(): v for k, v in d.items() if v != ""}
        tols = [low(t) for t in tolerations]
        spread = []
        for j, c in enumerate(cons):
            counts = [0 if c["emptySelector"] else sum(1 for p in nd["pods"] if p["match"][j] and p["ns"] == "default" and not p["terminating"]) for nd in nodes_]
            inc = []
            for nd in nodes_:
                untolerated = not ingest.taint_verdict([low(t) for t in nd["taints"]], tols)[0]
                ok = ((c["affinityPolicy"] != "Honor" or nd["affinityMatch"]) and (c["taintsPolicy"] != "Honor" or not untolerated)) if gate else nd["affinityMatch"]
                inc.append(1 if ok else 0)
            ki = keys.index(c["key"])
            spread.append(M.SpreadConstraint(col=ki, max_skew=c["maxSkew"], min_domains=1, hard=False, self_match=True, is_hostname=c["key"] == host,
                                             n_domains=max(len(ids[ki]), 1), node_match_count=np.array(counts, np.int32), node_included=np.array(inc, np.uint8)))
        pod = H.simple_pod(100, 64 << 20)
        pod.spread = spread
        pod.soft_relaxed = relaxed
        got_raw, got_norm, got_w = ccref.unit_pts_scores(nodes, pod, filtered)
        where = (gate, cons, nodes_, filtered)
        assert [float.fromhex(h) for h in weights_hex] == got_w, where
        assert [r if not ig else 0 for r, ig in zip(raw, ignored)] == got_raw, (raw, got_raw, where)
        assert norm == got_norm, (norm, got_norm, where)


def test_ipa_prescore_and_score(ccref):
    """interpodaffinity/scoring.go:51-290 -- processTerm / processTerms / append, processExistingPod, PreScore's closure over the nodes, Score,
    NormalizeScore -- against the oracle's score tables (ipa_build), its Score and its Skip: the same topology-pair sums, the same raw and
    normalized scores over a filtered node list, PreScore skipped in the same clusters.  The per-node inputs the oracle takes (the weight an
    existing pod contributes per key, the number of term hits) are derived the way the ingests derive them."""
    import numpy as np
    import helpers as H
    from cluster_capacity_amd import model as M
    for hard_w, inc_aff, inc_anti, nodes_, filtered, skipped, want_map, raw, norm in VEC["ipaPreScoreScore"]:
        n = len(nodes_)
        keys = ("zone", "host")
        cols, ids = zip(*[_intern([nd["labels"].get(k) for nd in nodes_]) for k in keys])
        nodes = _plain_nodes(n, cols)
        score_existing = [np.zeros(n, np.int64) for _ in keys]
        entries = 0
        for i, nd in enumerate(nodes_):
            for p in nd["pods"]:
                hits = [(t["key"], t["weight"]) for t, m in zip(inc_aff, p["matchAff"]) if m] + [(t["key"], -t["weight"]) for t, m in zip(inc_anti, p["matchAnti"]) if m]
                if hard_w > 0:
                    hits += [(t["key"], hard_w) for t in p["required"] if t["matches"]]
                hits += [(t["key"], t["weight"]) for t in p["prefAff"] if t["matches"]] + [(t["key"], -t["weight"]) for t in p["prefAnti"] if t["matches"]]
                for key, w in hits:
                    if key in nd["labels"]:  # the node carries the term's topology key (scoring.go:53-60)
                        score_existing[keys.index(key)][i] += w
                        entries += 1
        pod = H.simple_pod(100, 64 << 20)
        pod.ipa = M.InterPodAffinity(key_cols=[0, 1], key_ndom=[max(len(ids[0]), 1), max(len(ids[1]), 1)], aff_keys=[], self_aff=False, aff_existing=None, anti_keys=[],
                                     anti_self=[], anti_existing=[], exist_anti=[None, None], score_existing=score_existing, score_self=[0, 0], self_entries=[0, 0],
                                     entries_existing=entries)
        where = (hard_w, inc_aff, inc_anti, nodes_, filtered)
        tabs, totals = ccref.unit_ipa_build(nodes, pod)
        got = {}
        for k, name in enumerate(keys):
            inv = {v: s_ for s_, v in ids[k].items()}
            for vid, w in enumerate(tabs[k][3]):
                if vid and w:
                    got[(name, inv[vid])] = w
        want = {(k, val): w for k, m in want_map for val, w in m if w}  # (an entry that sums to 0 scores like no entry)
        assert got == want, where
        assert (totals[2] == 0) == bool(skipped), where
        got_raw, got_norm, got_skipped = ccref.unit_ipa_scores(nodes, pod, filtered)
        assert got_skipped == bool(skipped) and got_raw == raw and got_norm == norm, (got_raw, raw, got_norm, norm, where)


def test_node_search_of_one_cycle(ccref):
    """schedule_one.go:610-693 (findNodesThatPassFilters' per-position closure: the visiting order from nextStartNodeIndex, the search cancelled by
    the (K+1)-th feasible node) and :538-539 (nodes processed -> the next start index), driven by one worker in order, against ONE cycle of the
    oracle from the same start index: the same number of nodes visited, the same number of feasible nodes kept, the same next start index, a
    winner among the nodes the reference kept (the first of them for a profile without Score plugins)."""
    import dataclasses
    import numpy as np
    import helpers as H
    from cluster_capacity_amd import model as M
    for n, pct, scoring, start, feas, kept, processed, next_start in VEC["findNodesThatPassFilters"]:
        # feasibility through NodeResourcesFit: an infeasible node has no pod slot left
        nodes = H.simple_nodes([4000] * n, [8 * H.GiB] * n, [110 if f else 0 for f in feas])
        pod = H.simple_pod(100, 64 * H.MiB)
        prof = dataclasses.replace(M.Profile.default(), percentage_of_nodes_to_score=pct)
        if not scoring:
            prof = dataclasses.replace(prof, w_taint=0, w_nodeaffinity=0, w_fit=0, w_balanced=0, w_topologyspread=0, w_interpodaffinity=0, w_imagelocality=0)
        winner, evaluated, n_feasible, nxt = ccref.schedule_one(prof, nodes, pod, start)
        where = (n, pct, scoring, start)
        if not kept:
            assert winner == -1 and evaluated == n, where  # FitError: every node was visited (the start index is not read again)
            continue
        assert evaluated == processed and n_feasible == len(kept) and nxt == next_start, (evaluated, processed, n_feasible, len(kept), nxt, next_start, where)
        assert winner in kept and (scoring or winner == kept[0]), (winner, kept[:5], where)


def test_weigh_and_sum(ccref):
    """runtime/framework.go:1214-1238: the oracle's ccref_weigh (the function its cycle calls) gives RunScorePlugins' TotalScore."""
    for weights, scores, want in VEC["RunScorePlugins_weigh"]:
        got, _ = ccref.weigh_and_select(scores, weights)
        assert got == want, (weights, scores)


def test_select_host_is_a_possible_outcome_and_the_canonical_one(ccref):
    """schedule_one.go:894-941 driven through every outcome of its reservoir sampling: the oracle's ccref_select_host (the function its cycle calls)
    returns a node the reference can return, namely the first of them in list order (the canonical tie-break)."""
    for totals, possible, canonical in VEC["selectHost"]:
        got = ccref.select_host(totals)
        assert got in possible and got == canonical == min(possible), (totals, possible)


def test_topology_normalizing_weight(ccref):
    """podtopologyspread/scoring.go:294-296: math.Log(float64(size + 2)) -- the fixture's values come from the restated pure-Go algorithm
    (make_reference_vectors.go_math_log); the oracle's ccref_go_log agrees bit for bit, and both lie within an ulp of libm's log."""
    import math
    for size, want_hex in VEC["topologyNormalizingWeight"]:
        want = float.fromhex(want_hex)
        assert ccref.go_log(float(size + 2)).hex() == want_hex, size
        ref = math.log(size + 2)
        assert abs(want - ref) <= math.ulp(ref), size


# ---- string-level helpers the ingests mirror ------------------------------------------------------------------------------------
def test_toleration_matching():
    from cluster_capacity_amd import ingest
    for tol, taint, want in VEC["ToleratesTaint"]:
        t = {k.lower(): v for k, v in tol.items() if v != ""}      # (an object as kubectl prints it: empty fields are absent)
        x = {k.lower(): v for k, v in taint.items() if v != ""}
        assert ingest.tolerates(t, x) == want, (tol, taint)
        full = {k.lower(): v for k, v in tol.items()}               # ... and with the empty strings spelt out
        assert ingest.tolerates(full, {k.lower(): v for k, v in taint.items()}) == want, (tol, taint)


def test_taint_verdicts():
    """The TaintToleration Filter (first untolerated NoSchedule / NoExecute taint: helpers.go:78-101 under taint.go:23-28) and Score
    (untolerated PreferNoSchedule taints over the tolerations PreScore keeps: taint_toleration.go:135-181)."""
    from cluster_capacity_amd import ingest
    low = lambda d: {k.lower(): v for k, v in d.items() if v != ""}
    for taints, tols, found, at, cnt in VEC["taintVerdict"]:
        ok, n, first = ingest.taint_verdict([low(t) for t in taints], [low(t) for t in tols])
        assert ok == (not found) and n == cnt, (taints, tols)
        if found:
            assert first == low(taints[at]), (taints, tols)


def test_fits_request(ccref):
    """NodeResourcesFit's fitsRequest (fit.go:564-660) and the status code its Filter derives (fit.go:520-546: UnschedulableAndUnresolvable when
    any reason is Unresolvable) against the oracle's Fit filter: one one-node cluster per vector, the reasons read out of the FitError
    histogram.  (Which node state the hosts and the engine feed it is covered elsewhere; this is the arithmetic and the reason set.)"""
    import numpy as np
    from cluster_capacity_amd import model as M, report as R
    prof = M.Profile.default()
    for alloc, req, n_pods, pod, want in VEC["fitsRequest"]:
        names = sorted(set(alloc["scalars"]) | set(req["scalars"]) | set(pod["scalars"]))
        col = lambda d: [np.array([d["cpu"]], np.int64), np.array([d["mem"]], np.int64), np.array([d["eph"]], np.int64)] + [np.array([d["scalars"].get(k, 0)], np.int64) for k in names]
        nodes = M.NodesSoA(alloc=col(alloc), alloc_pods=np.array([alloc["pods"]], np.int32), req=col(req), nz_mcpu=np.array([req["cpu"]], np.int64), nz_mem=np.array([req["mem"]], np.int64),
                           pod_count=np.array([n_pods], np.int32), taintset_id=np.zeros(1, np.int32), unschedulable=np.zeros(1, np.uint8), names=["n"], scalar_names=names)
        spec = M.PodSpec(req=np.array([pod["cpu"], pod["mem"], pod["eph"]] + [pod["scalars"].get(k, 0) for k in names], np.int64), nz_mcpu=pod["cpu"] or 100, nz_mem=pod["mem"] or 200 << 20,
                         has_scalar_entries=bool(pod["scalars"]))
        r = ccref.run(prof, nodes, spec, max_limit=1)
        if not want:
            assert r.placed == 1, (alloc, req, n_pods, pod)
            continue
        assert r.placed == 0, (alloc, req, n_pods, pod)
        got = R._reason_histogram(r.hist, (), None, names)
        assert got == {text: 1 for text, _ in want}, (alloc, req, n_pods, pod, got)
        assert r.n_code_unschedulable == (0 if any(u for _, u in want) else 1), (alloc, req, n_pods, pod)


def test_inter_pod_affinity_filter(ccref):
    """InterPodAffinity's Filter (filtering.go:352-432: satisfyPodAffinity incl. the first-pod exception, satisfyPodAntiAffinity,
    satisfyExistingPodsAntiAffinity, their order and status codes) against the oracle, node by node: the other three nodes are made
    unschedulable (they still hold their pods, so the count maps are the cluster's) and the verdict is read from a one-cycle run."""
    import numpy as np
    from cluster_capacity_amd import model as M, report as R
    prof = M.Profile.default()
    zone_id, key_idx = {"a": 1, "b": 2}, {"zone": 0, "host": 1}
    checked = 0
    for labels, aff_terms, self_aff, aff_existing, anti_terms, anti_existing, exist_anti, want in VEC["ipaFilter"]:
        cols = [np.array([zone_id.get(lb.get("zone"), 0) for lb in labels], np.int32), np.array([i + 1 if "host" in lb else 0 for i, lb in enumerate(labels)], np.int32)]
        arr = lambda x: np.array(x, np.int32) if any(x) else None
        ipa = M.InterPodAffinity(key_cols=[0, 1], key_ndom=[2, 4], aff_keys=[key_idx[k] for k in aff_terms], self_aff=self_aff, aff_existing=arr(aff_existing),
                                 anti_keys=[key_idx[k] for k in anti_terms], anti_self=[False] * len(anti_terms), anti_existing=[arr(x) for x in anti_existing],
                                 exist_anti=[arr(exist_anti[k]) if k in exist_anti else None for k in ("zone", "host")], score_existing=[None, None], score_self=[0, 0],
                                 entries_existing=0, self_entries=[0, 0])
        pod = M.PodSpec(req=np.array([100, 1 << 20, 0], np.int64), nz_mcpu=100, nz_mem=1 << 20, ipa=ipa)
        for i in range(4):
            unsched = np.ones(4, np.uint8)
            unsched[i] = 0
            z = np.zeros(4, np.int64)
            nodes = M.NodesSoA(alloc=[z + 8000, z + (8 << 30), z.copy()], alloc_pods=np.full(4, 110, np.int32), req=[z.copy(), z.copy(), z.copy()], nz_mcpu=z.copy(), nz_mem=z.copy(),
                               pod_count=np.zeros(4, np.int32), taintset_id=np.zeros(4, np.int32), unschedulable=unsched, label_cols=[c.copy() for c in cols], names=[f"n{j}" for j in range(4)])
            r = ccref.run(prof, nodes, pod, max_limit=1)
            if want[i] is None:
                assert r.placed == 1 and list(r.log[:1]) == [i], (labels, aff_terms, anti_terms, i)
            else:
                code, text = want[i]
                got = R._reason_histogram(r.hist, (), None, [])
                assert r.placed == 0 and got.get(text) == 1 and sum(got.values()) == 4, (labels, aff_terms, self_aff, aff_existing, anti_terms, anti_existing, exist_anti, i, got)
                assert r.n_code_unschedulable == (1 if code == "Unschedulable" else 0), (code, i)
            checked += 1
    assert checked == 3600


def test_pod_topology_spread_filter(ccref):
    """PodTopologySpread's Filter (filtering.go:311-356) with minMatchNum (:56-69: the global minimum reads 0 below minDomains) against the oracle,
    node by node as in test_inter_pod_affinity_filter: missing label -> UnschedulableAndUnresolvable, skew above maxSkew -> Unschedulable."""
    import numpy as np
    from cluster_capacity_amd import model as M, report as R
    prof = M.Profile.default()
    zone_id = {"a": 1, "b": 2, "c": 3}
    checked = 0
    for labels, cons, want in VEC["ptsFilter"]:
        cols = [np.array([zone_id.get(lb.get("zone"), 0) for lb in labels], np.int32), np.array([i + 1 if "host" in lb else 0 for i, lb in enumerate(labels)], np.int32)]
        spread = [M.SpreadConstraint(col=0 if c["key"] == "zone" else 1, max_skew=c["maxSkew"], min_domains=c["minDomains"], hard=True, self_match=c["selfMatch"],
                                     is_hostname=c["key"] == "host", n_domains=3 if c["key"] == "zone" else 4, node_match_count=np.array(c["counts"], np.int32)) for c in cons]
        pod = M.PodSpec(req=np.array([100, 1 << 20, 0], np.int64), nz_mcpu=100, nz_mem=1 << 20, spread=spread)
        for i in range(4):
            unsched = np.ones(4, np.uint8)
            unsched[i] = 0
            z = np.zeros(4, np.int64)
            nodes = M.NodesSoA(alloc=[z + 8000, z + (8 << 30), z.copy()], alloc_pods=np.full(4, 110, np.int32), req=[z.copy(), z.copy(), z.copy()], nz_mcpu=z.copy(), nz_mem=z.copy(),
                               pod_count=np.zeros(4, np.int32), taintset_id=np.zeros(4, np.int32), unschedulable=unsched, label_cols=[c.copy() for c in cols], names=[f"n{j}" for j in range(4)])
            r = ccref.run(prof, nodes, pod, max_limit=1)
            if want[i] is None:
                assert r.placed == 1 and list(r.log[:1]) == [i], (labels, cons, i)
            else:
                code, text = want[i]
                got = R._reason_histogram(r.hist, (), None, [])
                assert r.placed == 0 and got.get(text) == 1 and sum(got.values()) == 4, (labels, cons, i, got)
                assert r.n_code_unschedulable == (1 if code == "Unschedulable" else 0), (code, i)
            checked += 1
    assert checked == 3600


def test_node_ports():
    """nodeports/node_ports.go:176-185 over HostPortInfo (sanitize, NewProtocolPort, CheckConflict): the ingest's host_ports + ports_conflict give the
    same verdict for every pair of (ports in use on the node, ports the pod wants)."""
    from cluster_capacity_amd import ingest
    spec = lambda ps: {"containers": [{"ports": [{k: v for k, v in p.items() if v != ""} for p in ps]}]}
    for used, want, fits in VEC["fitsPorts"]:
        assert (not ingest.ports_conflict(ingest.host_ports(spec(want)), set(ingest.host_ports(spec(used))))) == fits, (used, want, fits)


def test_zone_key():
    from cluster_capacity_amd import ingest
    for labels, want in VEC["GetZoneKey"]:
        assert ingest.zone_key(labels or {}) == want, labels


def test_normalized_image_name():
    from cluster_capacity_amd import ingest
    for name, want in VEC["normalizedImageName"]:
        assert ingest.normalized_image_name(name) == want, name


def test_requirement_matching():
    """labels.Requirement.Matches (apimachinery/pkg/labels/selector.go:246-293) behind nodeSelector / node affinity / label selectors.  The
    operators a NodeSelectorRequirement or LabelSelectorRequirement can name map onto selection operators
    (component-helpers/scheduling/corev1/nodeaffinity nodeSelectorRequirementsAsSelector, metav1 LabelSelectorAsSelector)."""
    from cluster_capacity_amd import ingest
    name = {"in": "In", "notin": "NotIn", "exists": "Exists", "!": "DoesNotExist", "gt": "Gt", "lt": "Lt"}
    checked = 0
    for op, vals, ls, want in VEC["requirementMatches"]:
        if op not in name:
            continue  # =, ==, != only arise from selector STRINGS, which no object of this path carries
        assert ingest.requirement_matches("k" in ls, ls.get("k"), name[op], vals) == want, (op, vals, ls)
        checked += 1
    assert checked > 1500


# ---- round 6: the volume plugins' Filters on object graphs ---------------------------------------------------------------------
# Every family is a set of small worlds for ONE plugin; the expected per-node verdicts come from the plugin's transliterated Filter
# (isVolumeConflict / satisfyVolumeConflicts, VolumeZone.Filter, CSILimits.Filter + getVolumeLimits, checkBoundClaims).  Both hosts must give them:
# cluster-capacity_amd/volumes.py here in-process, cluster-capacity-native through --dump-snapshot with the other volume plugins taken out of the profile.
_VOL_FAMILIES = {"volumeRestrictions": "VolumeRestrictions", "volumeZone": "VolumeZone", "csiLimits": "NodeVolumeLimits", "boundClaims": "VolumeBinding"}


def _vol_world(row):
    from test_native_host import EXAMPLES_POD, node, running_pod
    import yaml
    nodes = [node(nd["name"], cpu="4", mem="8Gi", pods="10", labels=nd["labels"]) for nd in row["nodes"]]
    pods = []
    for p in row.get("pods", []):
        q = running_pod(p["metadata"]["name"], p["spec"]["nodeName"], cpu="100m")
        q["metadata"]["namespace"] = p["metadata"]["namespace"]
        q["spec"]["volumes"] = p["spec"]["volumes"]
        pods.append(q)
    tmpl = yaml.safe_load(EXAMPLES_POD)
    tmpl["metadata"]["name"], tmpl["metadata"]["namespace"] = "sim", "default"
    tmpl["spec"]["volumes"] = row["volumes"]
    return nodes, pods, row.get("objs", []), tmpl


def _vol_expected(fam, row):
    from cluster_capacity_amd import model as M
    if fam == "volumeRestrictions":
        return [M.VOL_DISK_CONFLICT if c else 0 for c in row["conflict"]]
    if fam == "volumeZone":
        return [M.VOL_ZONE if c else 0 for c in row["reject"]]
    if fam == "csiLimits":
        return [M.VOL_MAX_COUNT if c else 0 for c in row["reject"]]
    return [{0: 0, 1: M.VOL_NODE_AFFINITY, 2: M.VOL_PV_NOT_EXIST}[c] for c in row["verdict"]]


@pytest.mark.parametrize("fam", sorted(_VOL_FAMILIES))
def test_volume_filters_python_host(fam):
    from cluster_capacity_amd import volumes as V
    for k, row in enumerate(VEC["volumeFilters_" + fam]):
        nodes, pods, objs, tmpl = _vol_world(row)
        by = {}
        for o in objs:
            by.setdefault(o["kind"], []).append(o)
        index = {n["metadata"]["name"]: i for i, n in enumerate(nodes)}
        side = V.volume_side(tmpl, nodes, pods, index, pvc_objs=by.get("PersistentVolumeClaim", []), class_objs=by.get("StorageClass", []), pv_objs=by.get("PersistentVolume", []),
                             enabled=(_VOL_FAMILIES[fam],), csinode_objs=by.get("CSINode", []), attachment_objs=by.get("VolumeAttachment", []))
        assert side.prefilter_reject is None, (fam, k, side.prefilter_reject)
        got = [0] * len(nodes) if side.veto is None else [int(x) for x in side.veto]
        assert got == _vol_expected(fam, row), (fam, k)
        if fam == "volumeRestrictions":
            assert bool(side.exclusive) == row["exclusive"], (fam, k)


@pytest.mark.parametrize("fam", sorted(_VOL_FAMILIES))
def test_volume_filters_native_host(fam, tmp_path):
    import subprocess
    import yaml
    from cluster_capacity_amd import build as B
    from helpers import SUBPROC_TIMEOUT
    native = B.build_host()
    others = [p for p in _VOL_FAMILIES.values() if p != _VOL_FAMILIES[fam]]
    cfg = tmp_path / "sched.yaml"
    cfg.write_text(yaml.safe_dump({"apiVersion": "kubescheduler.config.k8s.io/v1", "kind": "KubeSchedulerConfiguration",
                                   "profiles": [{"plugins": {"multiPoint": {"disabled": [{"name": p} for p in others]}}}]}))
    for k, row in enumerate(VEC["volumeFilters_" + fam]):
        nodes, pods, objs, tmpl = _vol_world(row)
        (tmp_path / "pod.yaml").write_text(yaml.safe_dump(json.loads(json.dumps(tmpl))))
        (tmp_path / "cluster.json").write_text(json.dumps({"kind": "List", "items": nodes + pods + objs}))
        p = subprocess.run([native, "--podspec", str(tmp_path / "pod.yaml"), "--snapshot", str(tmp_path / "cluster.json"), "--sync-persistent-volumes", "--default-config", str(cfg),
                            "--dump-snapshot", "-"], capture_output=True, text=True, timeout=SUBPROC_TIMEOUT)
        assert p.returncode == 0, (fam, k, p.stderr[-500:])
        dump = json.loads(p.stdout)
        got = dump["pod"]
        assert got["prefilter_reject"] is None, (fam, k, got["prefilter_reject"])
        veto = got["volume_veto"] if got["volume_veto"] is not None else [0] * len(nodes)
        want = dict(zip([nd["name"] for nd in row["nodes"]], _vol_expected(fam, row)))
        assert veto == [want[name] for name in dump["names"]], (fam, k)  # (the snapshot's node order is the node tree's: zone round-robin)
        if fam == "volumeRestrictions":
            assert bool(got["volume_exclusive"]) == row["exclusive"], (fam, k)
