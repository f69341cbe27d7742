"""InterPodAffinity (P/interpodaffinity).  CPU: the reference's own behavioural fixtures
(test/benchmark/pod_colocation_test.go -- SURVEY KA3 / KA4) and hand-derived answers pin the oracle.
GPU: the HIP filter/score against the oracle."""
import numpy as np
import pytest

import helpers as H
from cluster_capacity_amd import capi, model as M, report as R


def colocation_nodes(zones):
    """pod_colocation_test.go:193-221 BuildTestNode(name, 1000m, 1000 B, 30 pods); one label column = topology id."""
    n = len(zones)
    return H.simple_nodes([1000] * n, [1000] * n, [30] * n, label_cols=[np.array(zones, np.int32)])


def self_affinity_pod(n_domains):
    """BuildTestPod("pod-affinity", 10m, 10 B) with label key=value and a REQUIRED pod-affinity term on the
    topology key selecting key=value: the pod matches its own term.  Each clone's required term matches the
    incoming pod -> HardPodAffinityWeight (1) per clone on the clone's topology pair (scoring.go:104-110)."""
    p = H.simple_pod(10, 10)
    p.ipa = M.InterPodAffinity(key_cols=[0], key_ndom=[n_domains], aff_keys=[0], self_aff=True,
                               score_self=[1], self_entries=[1])
    return p


def test_ka3_colocation_single_node(ccref):
    # pod_colocation_test.go:18-97: 3 nodes, hostname topology, limit 100.  The fixture sets only
    # Profiles[0].Plugins.Filter.Enabled = [InterPodAffinity] (:49-53); the MultiPoint defaults still contribute every
    # default Filter plugin behind it (expandMultiPointPlugins, S/framework/runtime/framework.go:539-624 -- the merge
    # cluster_capacity_amd/schedconfig.py implements), so NodeResourcesFit is ON.
    # First pod: affinityCounts empty + self-match -> allowed everywhere (filtering.go:396-405); afterwards only the
    # node holding the clones passes InterPodAffinity; its 30-pod capacity binds: 30 on ONE node, then Unschedulable.
    prof = M.Profile.default()
    r = ccref.run(prof, colocation_nodes([1, 2, 3]), self_affinity_pod(3), max_limit=100)
    assert r.placed == 30 and r.stop == M.STOP_UNSCHEDULABLE
    assert sorted(r.per_node_count.tolist()) == [0, 0, 30]  # the test's assertion: all pods on one node (:84-90)
    assert r.per_node_count.tolist() == [30, 0, 0]          # canonical tie-break: lowest index
    assert R.stop_reason(r, 3, 100).startswith(
        "Unschedulable: 0/3 nodes are available: 1 Too many pods, 2 node(s) didn't match pod affinity rules.")


def test_hand_case_interpodaffinity_as_the_only_filter(ccref):
    # NOT a reference fixture: a profile whose ONLY filter is InterPodAffinity (MultiPoint defaults disabled).  No Fit
    # filter -> the 30-pod capacity never binds: the limit of 100 is reached on one node.
    prof = M.Profile(filter_mask=M.F_INTERPODAFFINITY)
    r = ccref.run(prof, colocation_nodes([1, 2, 3]), self_affinity_pod(3), max_limit=100)
    assert r.placed == 100 and r.stop == M.STOP_LIMIT
    assert r.per_node_count.tolist() == [100, 0, 0]


def test_ka4_colocation_single_zone(ccref):
    # pod_colocation_test.go:99-190: 9 nodes in 3 zones (custom topology key, so the node tree has one zone and the
    # canonical order is by name: node1-1..node3-3), Filter.Enabled = [InterPodAffinity, NodeResourcesFit] in front of
    # the MultiPoint defaults (see KA3), limit 100.
    # All pods land in the first pod's zone; 30 pods/node -> 90 = 30+30+30, then Unschedulable.
    prof = M.Profile.default()
    nodes = colocation_nodes([1, 1, 1, 2, 2, 2, 3, 3, 3])
    r = ccref.run(prof, nodes, self_affinity_pod(3), max_limit=100)
    assert r.placed == 90 and r.stop == M.STOP_UNSCHEDULABLE
    assert r.per_node_count.tolist() == [30, 30, 30, 0, 0, 0, 0, 0, 0]  # one zone only (:181-187)
    assert R.stop_reason(r, 9, 100).startswith(
        "Unschedulable: 0/9 nodes are available: 3 Too many pods, 6 node(s) didn't match pod affinity rules.")


def test_ipa_required_anti_affinity_on_hostname(ccref):
    # required anti-affinity to its own label on the hostname key: one clone per node (config 5's pod shape)
    nodes = colocation_nodes([1, 2, 3, 4])
    p = H.simple_pod(10, 10)
    p.ipa = M.InterPodAffinity(key_cols=[0], key_ndom=[4], anti_keys=[0], anti_self=[True], anti_existing=[None])
    r = ccref.run(M.Profile.default(), nodes, p)
    assert r.placed == 4 and r.per_node_count.tolist() == [1, 1, 1, 1]
    assert "4 node(s) didn't match pod anti-affinity rules" in R.stop_reason(r, 4, 0)


def test_ipa_existing_pods_anti_affinity_and_missing_key(ccref):
    # node 0: an existing pod has required anti-affinity (this key) matching the incoming pod -> whole domain blocked;
    # node 3 lacks the topology key: anti-affinity cannot bind there, affinity is not required -> feasible.
    nodes = H.simple_nodes([1000] * 4, [1000] * 4, [2] * 4, label_cols=[np.array([1, 1, 2, 0], np.int32)])
    p = H.simple_pod(10, 10)
    p.ipa = M.InterPodAffinity(key_cols=[0], key_ndom=[2], exist_anti=[np.array([1, 0, 0, 0], np.int32)])
    r = ccref.run(M.Profile.default(), nodes, p)
    assert r.per_node_count.tolist() == [0, 0, 2, 2]
    assert "2 node(s) didn't satisfy existing pods anti-affinity rules" in R.stop_reason(r, 4, 0)


def test_ipa_preferred_score_prefers_domain(ccref):
    # existing pods put weight 5 on domain 2 (e.g. a preferred affinity term of the incoming pod matches them):
    # normalized 100 there, 0 elsewhere, x weight 2 dominates the +-1 resource-score differences
    nodes = colocation_nodes([1, 2, 3])
    p = H.simple_pod(10, 10)
    p.ipa = M.InterPodAffinity(key_cols=[0], key_ndom=[3], score_existing=[np.array([0, 5, 0], np.int64)],
                               score_self=[0], self_entries=[0], entries_existing=1)
    r = ccref.run(M.Profile.default(), nodes, p, max_limit=10)
    assert r.per_node_count.tolist() == [0, 10, 0]


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
        assert R.stop_reason(got, nodes.n, limit) == R.stop_reason(ref, nodes.n, limit)
    return got


@pytest.mark.gpu
def test_gpu_ipa_reference_fixtures(ccref):
    # the fixtures' real profile: Filter.Enabled entries in front of the MultiPoint defaults (see test_ka3_*)
    got = _gpu_check(ccref, colocation_nodes([1, 2, 3]), self_affinity_pod(3), M.Profile.default(), 100)
    assert got.placed == 30 and got.per_node_count.tolist() == [30, 0, 0]
    got = _gpu_check(ccref, colocation_nodes([1, 1, 1, 2, 2, 2, 3, 3, 3]), self_affinity_pod(3), M.Profile.default(), 100)
    assert got.placed == 90 and got.per_node_count.tolist() == [30, 30, 30, 0, 0, 0, 0, 0, 0]
    # hand cases (not reference fixtures): InterPodAffinity as the only filter / with NodeResourcesFit only
    _gpu_check(ccref, colocation_nodes([1, 2, 3]), self_affinity_pod(3), M.Profile(filter_mask=M.F_INTERPODAFFINITY), 100)
    _gpu_check(ccref, colocation_nodes([1, 1, 1, 2, 2, 2, 3, 3, 3]), self_affinity_pod(3),
               M.Profile(filter_mask=M.F_INTERPODAFFINITY | M.F_FIT), 100)


@pytest.mark.gpu
def test_gpu_ipa_hand_cases(ccref):
    p = H.simple_pod(10, 10)
    p.ipa = M.InterPodAffinity(key_cols=[0], key_ndom=[4], anti_keys=[0], anti_self=[True], anti_existing=[None])
    _gpu_check(ccref, colocation_nodes([1, 2, 3, 4]), p, M.Profile.default(), 0)
    nodes = H.simple_nodes([1000] * 4, [1000] * 4, [2] * 4, label_cols=[np.array([1, 1, 2, 0], np.int32)])
    p = H.simple_pod(10, 10)
    p.ipa = M.InterPodAffinity(key_cols=[0], key_ndom=[2], exist_anti=[np.array([1, 0, 0, 0], np.int32)])
    _gpu_check(ccref, nodes, p, M.Profile.default(), 0)
    p = H.simple_pod(10, 10)
    p.ipa = M.InterPodAffinity(key_cols=[0], key_ndom=[3], score_existing=[np.array([0, 5, 0], np.int64)],
                               score_self=[0], self_entries=[0], entries_existing=1)
    _gpu_check(ccref, colocation_nodes([1, 2, 3]), p, M.Profile.default(), 10)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(16))
def test_gpu_ipa_random(ccref, seed):
    rng = np.random.default_rng(900 + seed)
    nodes, pod, prof = H.random_case(rng, int(rng.integers(1, 1000)))
    pod.ipa = H.random_ipa(rng, nodes)
    if seed % 3 == 0:
        pod.spread = H.random_spread(rng, nodes, n_constraints=1)
    _gpu_check(ccref, nodes, pod, prof, int(rng.choice([0, 0, 80])))


@pytest.mark.gpu
def test_gpu_ipa_hostname_anti_affinity_1000_nodes(ccref):
    # config 5's pod shape: required anti-affinity to itself on kubernetes.io/hostname (one clone per node) + zone spread
    from cluster_capacity_amd import synth
    n = 1000
    nodes, pod, prof = synth.make_config("C3", n_nodes=n, seed=5)
    nodes.label_cols.append(np.arange(1, n + 1, dtype=np.int32))  # column 2 = hostname
    pod.ipa = M.InterPodAffinity(key_cols=[2], key_ndom=[n], anti_keys=[0], anti_self=[True], anti_existing=[None])
    pod.spread = [synth.zone_spread(n, max_skew=2)]
    got = _gpu_check(ccref, nodes, pod, prof, 0)
    assert got.per_node_count.max() == 1
