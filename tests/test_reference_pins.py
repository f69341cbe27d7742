"""Every literal string the hosts print, every status code the oracle assigns and every constant the path hard-codes is a
transcription from the reference's Go sources.  tests/golden/reference_pins.json holds those values as READ OUT OF /root/reference
(tests/golden/make_reference_pins.py: file, line, value); here the transcriptions -- Python host, C++ host, oracle -- are compared
with the fixture, and, where the reference tree is present (the build container), the fixture with the sources again.

This is the part of "pinning the oracle to the reference" that needs no Go toolchain: it cannot show that the arithmetic is the
reference's (the known answers and the arithmetic tests do what they can there), but it does show that no message, code or default
drifted in transcription."""
import json
import os
import re

import numpy as np
import pytest

import helpers as H
from cluster_capacity_amd import cli, ingest, model as M, preemption, report as R, schedconfig

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PINS = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_pins.json")))
V = {k: v["value"] for k, v in PINS.items()}


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="the reference tree exists in the build container only")
def test_fixture_is_what_the_reference_sources_say():
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_reference_pins", os.path.join(ROOT, "tests", "golden", "make_reference_pins.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.collect() == PINS


def test_python_host_strings():
    t = R.REASON_TEXT
    assert t[M.R_UNSCHEDULABLE] == V["reason.unschedulable"] and t[M.R_NODENAME] == V["reason.nodename"] and t[M.R_NODEAFFINITY] == V["reason.nodeaffinity"]
    assert t[M.R_TOO_MANY_PODS] == V["reason.too_many_pods"] and t[M.R_NODEPORTS] == V["reason.nodeports"]
    assert t[M.R_PTS_SKEW] == V["reason.pts_skew"] and t[M.R_PTS_MISSING_LABEL] == V["reason.pts_skew"] + V["reason.pts_missing_label_suffix"]
    assert t[M.R_IPA_AFFINITY] == V["reason.ipa_affinity"] and t[M.R_IPA_ANTI] == V["reason.ipa_anti"] and t[M.R_IPA_EXISTING_ANTI] == V["reason.ipa_existing_anti"]
    # the rendered FitError: prefix, resource reasons, the preemption tail
    hist = np.zeros(M.NREASON, np.int64)
    hist[M.R_RES0], hist[M.R_RES0 + 1], hist[M.R_RES0 + 2], hist[M.R_RES0 + 3] = 1, 2, 3, 4
    msg = R.fit_error_message(10, hist, [], 4, scalar_names=["example.com/gpu"])
    assert msg.startswith(V["fit_error.prefix_format"].replace("%v", "10") + ":")
    for text in (V["reason.insufficient_cpu"], V["reason.insufficient_memory"], V["reason.insufficient_ephemeral"],
                 V["reason.insufficient_scalar_format"].replace("%v", "example.com/gpu")):
        assert text in msg
    assert f" {V['preemption.prefix']}{V['fit_error.prefix_format'].replace('%v', '10')}: 4 {V['preemption.no_victims']}, 6 {V['preemption.not_helpful']}." in msg
    assert R.fit_error_message(10, hist, [], 4, preemption=preemption.Outcome(kind="never")).endswith(" " + V["preemption.prefix"] + V["preemption.never"])
    res = M.RunResult(placed=7, stop=M.STOP_LIMIT, per_node_count=np.zeros(1, np.int32), log=None, hist=hist, hist_taintset=np.zeros(1, np.int64), n_code_unschedulable=0)
    assert R.stop_reason(res, 10, 7) == V["stop.limit_format"].replace("%v", "7")
    # the taint reason as the ingest spells it
    from test_ingest_cli import EXAMPLES_POD, node
    import yaml
    snap = ingest.build_snapshot([node("n", taints=[{"key": "k", "value": "v", "effect": "NoSchedule"}])], [], yaml.safe_load(EXAMPLES_POD))
    assert snap.taint_reasons[0] == V["reason.taint_format"] % ("k", "v")
    # the pretty printer
    review = {"spec": {"podRequirements": [{"podName": "p", "resources": {"primaryResources": {"cpu": "1", "memory": "1"}, "scalarResources": None}, "nodeSelectors": None}]},
              "status": {"replicas": 3, "failReason": {"failType": "T", "failMessage": "m"}, "pods": [{"podName": "p", "replicasOnNodes": [{"nodeName": "a", "replicas": 3}]}]}}
    out = cli.pretty(review, True)
    for key, args in (("report.headline_format", ("3", "p")), ("report.termination_format", ("T", "m")), ("report.distribution_header", ()),
                      ("report.node_line_format", ("a", "3")), ("report.requirements_format", ("p",))):
        text = V[key]
        for a in args:
            text = text.replace("%v", a, 1)
        assert text in out, key


def test_native_host_strings():
    src = "".join(open(os.path.join(ROOT, "cluster-capacity_amd", "host", f)).read() for f in ("report.hpp", "snapshot.hpp"))
    for key in ("reason.unschedulable", "reason.nodename", "reason.nodeaffinity", "reason.too_many_pods", "reason.nodeports", "reason.pts_skew",
                "reason.ipa_affinity", "reason.ipa_anti", "reason.ipa_existing_anti", "preemption.no_victims", "preemption.not_helpful", "preemption.never",
                "preemption.prefix", "report.distribution_header"):
        assert '"%s' % V[key].replace('"', '\\"') in src or V[key] in src, key
    assert V["reason.pts_skew"] + V["reason.pts_missing_label_suffix"] in src
    assert '"Insufficient "' in src and "node(s) had untolerated taint {" in src and "LimitReached: Maximum number of pods simulated: " in src
    assert "The cluster can schedule " in src and " instance(s) of the pod " in src and "Termination reason: " in src and " instance(s)" in src
    m = re.search(r"kDefaultMilliCPU = (\d+);", src)
    assert int(m.group(1)) == V["default.milli_cpu_request"]
    m = re.search(r"kDefaultMemory = (\d+)ll \* (\d+) \* (\d+);", src)
    assert int(m.group(1)) * int(m.group(2)) * int(m.group(3)) == V["default.memory_request"]
    prof = open(os.path.join(ROOT, "cluster-capacity_amd", "host", "profile.hpp")).read()
    m = re.search(r"f\.w_taint = (\d+), f\.w_nodeaffinity = (\d+), f\.w_fit = (\d+), f\.w_balanced = (\d+), f\.w_topologyspread = (\d+), f\.w_interpodaffinity = (\d+), f\.w_imagelocality = (\d+);", prof)
    assert [int(x) for x in m.groups()] == [V["weight.TaintToleration"], V["weight.NodeAffinity"], V["weight.NodeResourcesFit"], V["weight.NodeResourcesBalancedAllocation"],
                                            V["weight.PodTopologySpread"], V["weight.InterPodAffinity"], V["weight.ImageLocality"]]
    assert re.search(r"hard_pod_affinity_weight = %d;" % V["default.hard_pod_affinity_weight"], prof)


def test_python_host_constants():
    p = M.Profile.default()
    assert (p.w_taint, p.w_nodeaffinity, p.w_fit, p.w_balanced, p.w_topologyspread, p.w_interpodaffinity, p.w_imagelocality) == (
        V["weight.TaintToleration"], V["weight.NodeAffinity"], V["weight.NodeResourcesFit"], V["weight.NodeResourcesBalancedAllocation"], V["weight.PodTopologySpread"],
        V["weight.InterPodAffinity"], V["weight.ImageLocality"])
    assert ingest.DEFAULT_MILLI_CPU == V["default.milli_cpu_request"] and ingest.DEFAULT_MEMORY == V["default.memory_request"]
    prof, hard = schedconfig.profile_from_config(None)
    assert hard == V["default.hard_pod_affinity_weight"]
    # (the reference's default percentageOfNodesToScore is 0 = adaptive; the hosts apply it when the run is order-dependent, see cli.main)
    assert V["default.percentage_of_nodes_to_score"] == 0


def test_oracle_constants(ccref):
    top = V["score.max_node_score"]
    assert ccref.least_allocated([0], [10], [1]) == top and ccref.balanced_allocation([5, 5], [10, 10]) == top
    # numFeasibleNodesToFind (schedule_one.go:697-723): below minFeasibleNodesToFind every node; adaptive 50 - N/125, floored at the percentage floor
    k, floor = V["search.min_feasible_nodes"], V["search.min_feasible_percentage"]
    assert ccref.num_feasible_nodes_to_find(0, k - 1) == k - 1 and ccref.num_feasible_nodes_to_find(0, k) == k
    assert ccref.num_feasible_nodes_to_find(0, 5000) == 5000 * (50 - 5000 // 125) // 100
    assert ccref.num_feasible_nodes_to_find(0, 100000) == 100000 * floor // 100
    assert ccref.num_feasible_nodes_to_find(10, 500) == k  # 10 % of 500 = 50 < the minimum
    # ImageLocality thresholds (image_locality.go:33-35,84-115): one container
    mb, lo, hi = V["image.mb"], V["image.min_threshold_mb"], V["image.max_container_threshold_mb"]
    assert ccref.image_locality_score([lo * mb], [1], 1, 1) == 0 and ccref.image_locality_score([hi * mb], [1], 1, 1) == top
    assert ccref.image_locality_score([(lo + 1) * mb - 1], [1], 1, 1) == top * (mb - 1) // ((hi - lo) * mb)


def _one_node_failing(kind):
    """A 2-node snapshot where node 1 fails exactly the filter `kind` (node 0 takes one pod, then is full)."""
    nodes = H.simple_nodes([1000, 1000], [1 << 30, 1 << 30], [1, 5], label_cols=[np.array([1, 1]), np.array([1, 2])])
    pod = H.simple_pod(100, 1 << 20)
    pod.taint_filter_ok, pod.taint_prefer_cnt = np.array([1, 0], np.uint8), np.zeros(2, np.int32)
    nodes.taintset_id = np.zeros(2, np.int32)
    t_in = lambda size, ids: np.isin(np.arange(size), ids).astype(np.uint8)
    if kind == "unschedulable":
        nodes.unschedulable = np.array([0, 1], np.uint8)
    elif kind == "taint":
        nodes.taintset_id = np.array([0, 1], np.int32)
    elif kind == "nodeaffinity":
        pod.affinity_filter_active, pod.has_node_selector, pod.node_selector = True, True, [(1, t_in(3, [1]))]
    elif kind == "nodeports":
        pod.has_host_ports, pod.host_ports_conflict = True, np.array([0, 1], np.uint8)
    elif kind == "fit_default":
        nodes.req[0][1] = 950
    elif kind == "fit_beyond_allocatable":
        nodes.alloc[0][1] = 50
    elif kind == "pts_missing_label":
        nodes.label_cols[1] = np.array([1, 0], np.int32)
        pod.spread = [M.SpreadConstraint(col=1, max_skew=5, hard=True, self_match=True, n_domains=2)]
    elif kind == "pts_skew":
        pod.spread = [M.SpreadConstraint(col=1, max_skew=1, hard=True, self_match=False, n_domains=2, node_match_count=np.array([0, 3], np.int32))]
    elif kind == "ipa_affinity":
        pod.ipa = M.InterPodAffinity(key_cols=[1], key_ndom=[2], aff_keys=[0], self_aff=False, aff_existing=np.array([1, 0], np.int32),
                                     exist_anti=[None], score_existing=[None], score_self=[0], self_entries=[0])
    elif kind == "ipa_anti":
        pod.ipa = M.InterPodAffinity(key_cols=[1], key_ndom=[2], anti_keys=[0], anti_self=[False], anti_existing=[np.array([0, 1], np.int32)],
                                     exist_anti=[None], score_existing=[None], score_self=[0], self_entries=[0])
    elif kind == "ipa_existing_anti":
        pod.ipa = M.InterPodAffinity(key_cols=[1], key_ndom=[2], exist_anti=[np.array([0, 1], np.int32)], score_existing=[None], score_self=[0], self_entries=[0])
    return nodes, pod


@pytest.mark.parametrize("kind,slot", [("unschedulable", M.R_UNSCHEDULABLE), ("taint", None), ("nodeaffinity", M.R_NODEAFFINITY), ("nodeports", M.R_NODEPORTS),
                                       ("fit_default", M.R_RES0), ("fit_beyond_allocatable", M.R_RES0), ("pts_missing_label", M.R_PTS_MISSING_LABEL),
                                       ("pts_skew", M.R_PTS_SKEW), ("ipa_affinity", M.R_IPA_AFFINITY), ("ipa_anti", M.R_IPA_ANTI),
                                       ("ipa_existing_anti", M.R_IPA_EXISTING_ANTI)])
def test_oracle_status_codes(ccref, kind, slot):
    """Which filter failures are plain Unschedulable (preemption may help) and which are UnschedulableAndUnresolvable: the oracle's
    n_code_unschedulable against the code the reference's plugin returns for that reason."""
    nodes, pod = _one_node_failing(kind)
    r = ccref.run(M.Profile.default(), nodes, pod)
    assert r.placed == 1 and r.log.tolist() == [0] and r.stop == M.STOP_UNSCHEDULABLE
    node1_resolvable = V["code." + kind] == "Unschedulable"
    if kind == "nodeports":  # node 0 holds a clone: its own ports conflict before Fit is asked (NodePorts runs first)
        assert r.hist[slot] == 2 and r.hist[M.R_TOO_MANY_PODS] == 0 and r.n_code_unschedulable == 2 and node1_resolvable
        return
    if slot is None:
        assert r.hist_taintset[1] == 1
    else:
        assert r.hist[slot] == 1
    # node 0 is full after its one pod: "Too many pods", plain Unschedulable (fit.go:520-531 default code)
    assert V["code.fit_default"] == "Unschedulable" and r.hist[M.R_TOO_MANY_PODS] == 1
    assert r.n_code_unschedulable == 1 + int(node1_resolvable), kind
    assert V["code.preemption_no_victims"] == "UnschedulableAndUnresolvable"


# ---- filter order: the first failing plugin decides the reasons (framework.go:897-930), in the order of the default plugin list ----
_FILTERS = {"NodeUnschedulable": "unschedulable", "TaintToleration": "taint", "NodeAffinity": "nodeaffinity", "NodePorts": "nodeports",
            "NodeResourcesFit": "fit_default", "PodTopologySpread": "pts_skew", "InterPodAffinity": "ipa_anti"}
_SLOT = {"unschedulable": M.R_UNSCHEDULABLE, "taint": None, "nodeaffinity": M.R_NODEAFFINITY, "nodeports": M.R_NODEPORTS, "fit_default": M.R_RES0,
         "pts_skew": M.R_PTS_SKEW, "ipa_anti": M.R_IPA_ANTI}


def _apply(kind, nodes, pod):
    """The mutation of _one_node_failing(kind) on an existing (nodes, pod): node 1 additionally fails `kind`."""
    n2, p2 = _one_node_failing(kind)
    if kind == "unschedulable":
        nodes.unschedulable = n2.unschedulable
    elif kind == "taint":
        nodes.taintset_id = n2.taintset_id
    elif kind == "nodeaffinity":
        pod.affinity_filter_active, pod.has_node_selector, pod.node_selector = True, True, p2.node_selector
    elif kind == "nodeports":
        pod.has_host_ports, pod.host_ports_conflict = True, p2.host_ports_conflict
    elif kind == "fit_default":
        nodes.req[0][1] = 950
    elif kind == "pts_skew":
        pod.spread = p2.spread
    elif kind == "ipa_anti":
        pod.ipa = p2.ipa


def test_oracle_filter_order_is_the_default_plugin_order(ccref):
    order = [p for p in V["plugins.multipoint_order"] if p in _FILTERS]
    assert order == ["NodeUnschedulable", "TaintToleration", "NodeAffinity", "NodePorts", "NodeResourcesFit", "PodTopologySpread", "InterPodAffinity"]
    for i, first in enumerate(order):
        for second in order[i + 1:]:
            a, b = _FILTERS[first], _FILTERS[second]
            nodes, pod = _one_node_failing(a)
            _apply(b, nodes, pod)
            r = ccref.run(M.Profile.default(), nodes, pod)
            # node 1 fails both plugins: only the earlier one's reason is recorded for it
            got_a = r.hist_taintset[1] if _SLOT[a] is None else r.hist[_SLOT[a]]
            got_b = r.hist_taintset[1] if _SLOT[b] is None else r.hist[_SLOT[b]]
            # (node 0 took the one clone: with host ports in play ITS reason is NodePorts too, whichever role NodePorts has in the pair)
            assert r.placed == 1 and got_a == (2 if a == "nodeports" else 1) and got_b == (1 if b == "nodeports" else 0), (first, second, r.hist.tolist())


def test_scalar_resource_names():
    """schedutil.IsScalarResourceName (S/util/utils.go:140-143 over pkg/apis/core/v1/helper/helpers.go:36-66,133-135 and
    validation.IsQualifiedName): which request names become resource columns and which the scheduler drops.  The prefixes and the
    qualified-name pieces come from the sources; the verdicts below follow from reading those functions."""
    assert ingest._QNAME.pattern == "(" + V["qname.char"] + V["qname.ext_char"] + "*)?" + V["qname.char"]
    assert ingest._DNS1123_LABEL == V["qname.dns1123_label"] and V["qname.max_length"] == 63 and V["qname.dns1123_subdomain_max_length"] == 253
    assert (V["resource.prefix_native"], V["resource.prefix_hugepages"], V["resource.prefix_attachable"], V["resource.prefix_requests"]) == (
        "kubernetes.io/", "hugepages-", "attachable-volumes-", "requests.")
    src = open(os.path.join(ROOT, "cluster-capacity_amd", "host", "snapshot.hpp")).read() + open(os.path.join(ROOT, "cluster-capacity_amd", "ingest.py")).read()
    for key in ("resource.prefix_native", "resource.prefix_hugepages", "resource.prefix_attachable", "resource.prefix_requests"):
        assert src.count('"%s"' % V[key]) >= 2, key  # both hosts spell the prefix exactly
    yes = ["nvidia.com/gpu", "example.com/gpu", "hugepages-2Mi", "hugepages-1Gi", "kubernetes.io/batch", "x.kubernetes.io/y", "attachable-volumes-csi-x",
           "example.com/" + "n" * 63, "a-b.c/d_e.f", "requests.kubernetes.io/x"]
    no = ["cpu", "memory", "pods", "ephemeral-storage", "storage", "foo", "requests.example.com/x", "example.com/Bad Name", "a/b/c", "Example.com/x", "example.com/",
          "/x", "example.com/" + "n" * 64, "example..com/x", "-a.com/x", "example.com/-x", "example.com/x-"]
    for n in yes:
        assert ingest.is_scalar_resource(n), n
    for n in no:
        assert not ingest.is_scalar_resource(n), n


def test_numeric_label_comparison_parses_integers_like_go():
    """labels.Requirement.Matches, Gt / Lt (apimachinery/pkg/labels/selector.go:264-289): strconv.ParseInt(value, 10, 64) on both sides; a
    value it refuses makes the requirement false.  Python's int() is more generous ("1_0", " 5", beyond int64): go_parse_int is not."""
    f = ingest.go_parse_int
    assert [f(x) for x in ("7", "+7", "-7", "007", "0", str((1 << 63) - 1), str(-(1 << 63)))] == [7, 7, -7, 7, 0, (1 << 63) - 1, -(1 << 63)]
    assert [f(x) for x in ("1_0", " 5", "5 ", "", "+", "3x", "2.0", "0x10", str(1 << 63), "٣", None)] == [None] * 11
    m = ingest.requirement_matches
    assert m(True, "10", "Gt", ["9"]) and not m(True, "1_0", "Gt", ["9"]) and not m(True, "10", "Gt", ["9", "8"]) and not m(False, None, "Lt", ["9"])
    assert m(True, "-2", "Lt", ["+1"]) and not m(True, "10", "Gt", ["9223372036854775808"])


def test_verbose_requirements_block_follows_the_reference_format():
    """report.go:236-252: `%v` of the ScalarResources map prints Go's map syntax with sorted keys; the node selector prints as
    labels.SelectorFromSet(...).String() (sorted k=v pairs joined by commas)."""
    review = {"spec": {"podRequirements": [{"podName": "p", "resources": {"primaryResources": {"cpu": "150m", "memory": "100Mi"},
                                                                          "scalarResources": {"hugepages-2Mi": 2097152, "example.com/gpu": 1}},
                                            "nodeSelectors": {"zone": "a", "disk": "ssd"}}]},
              "status": {"replicas": 0, "failReason": {"failType": "T", "failMessage": "m"}, "pods": [{"podName": "p", "replicasOnNodes": []}]}}
    out = cli.pretty(review, True)
    assert "\t- ScalarResources: map[example.com/gpu:1 hugepages-2Mi:2097152]\n" in out
    assert "\t- NodeSelector: disk=ssd,zone=a\n" in out
    assert "\t- CPU: 150m\n\t- Memory: 100Mi\n" in out


def test_report_keeps_the_references_resource_key_and_yaml_key_order(tmp_path):
    """report.go:34 spells the third primary resource "nvdia.com/gpu" (sic) -- a consumer of the JSON sees that key; -o yaml goes through
    sigs.k8s.io/yaml, i.e. through JSON: every mapping's keys come out sorted (report.go:296-303)."""
    assert V["report.gpu_resource_name"] == "nvdia.com/gpu"
    import io
    import yaml
    from test_ingest_cli import EXAMPLES_POD
    pod = yaml.safe_load(EXAMPLES_POD)
    assert set(cli.pod_requirements(pod)["resources"]["primaryResources"]) == {"cpu", "memory", V["report.gpu_resource_name"]}
    src = open(os.path.join(ROOT, "cluster-capacity_amd", "host", "report.hpp")).read()
    assert 'prim.set("%s"' % V["report.gpu_resource_name"] in src
    text = yaml.safe_dump({"status": {"replicas": 1, "pods": []}, "spec": {"templates": [], "replicas": 0}}, sort_keys=True)
    assert text.index("spec:") < text.index("status:")
    assert "sort_keys=True" in open(os.path.join(ROOT, "cluster-capacity_amd", "cli.py")).read() and "sorted_keys(review)" in open(
        os.path.join(ROOT, "cluster-capacity_amd", "host", "main.cpp")).read()
