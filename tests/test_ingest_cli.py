"""Snapshot ingest + CLI.  CPU: Quantity / selector / toleration semantics and the ingest of
the README demo cluster against the oracle.  GPU: the CLI end to end (README.md:44-66 -> 52 = 13 x 4)."""
import io
import json
import os

import numpy as np
import pytest
import yaml

from cluster_capacity_amd import cli, ingest, model as M, report as R

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXAMPLES_POD = """
apiVersion: v1
kind: Pod
metadata: {name: small-pod, labels: {app: guestbook, tier: frontend}}
spec:
  containers:
  - name: php-redis
    image: gcr.io/google-samples/gb-frontend:v4
    resources: {limits: {cpu: 150m, memory: 100Mi}, requests: {cpu: 150m, memory: 100Mi}}
  restartPolicy: OnFailure
  dnsPolicy: Default
"""  # the reference's examples/pod.yaml


def node(name, cpu="2", mem="4Gi", pods="110", labels=None, taints=None, unschedulable=False):
    return {"kind": "Node", "metadata": {"name": name, "labels": dict(labels or {})},
            "spec": {"taints": taints or [], "unschedulable": unschedulable},
            "status": {"allocatable": {"cpu": cpu, "memory": mem, "pods": pods, "ephemeral-storage": "100Gi"}}}


def running_pod(name, node_name, cpu=None, mem=None, labels=None, ns="default", phase="Running", affinity=None):
    req = {k: v for k, v in (("cpu", cpu), ("memory", mem)) if v is not None}
    return {"kind": "Pod", "metadata": {"name": name, "namespace": ns, "labels": dict(labels or {})},
            "spec": {"nodeName": node_name, "containers": [{"name": "c", "resources": {"requests": req}}], "affinity": affinity or {}},
            "status": {"phase": phase}}


def test_quantity_semantics():
    # quantity.go:813-834: Value / MilliValue round UP
    assert ingest.milli_value("150m") == 150 and ingest.milli_value("2") == 2000 and ingest.milli_value("0.1") == 100
    assert ingest.milli_value("1500u") == 2 and ingest.milli_value("100n") == 1
    assert ingest.value("100Mi") == 104857600 and ingest.value("4Gi") == 4 << 30 and ingest.value("1e3") == 1000
    assert ingest.value("1.5Ki") == 1536 and ingest.value("100m") == 1 and ingest.value("4G") == 4_000_000_000
    with pytest.raises(ValueError):
        ingest.parse_quantity("abc")


def test_selector_and_toleration_semantics():
    rm = ingest.requirement_matches
    assert rm(True, "a", "In", ["a", "b"]) and not rm(False, None, "In", ["a"])
    assert rm(False, None, "NotIn", ["a"]) and rm(False, None, "DoesNotExist", []) and not rm(True, "a", "DoesNotExist", [])
    assert rm(True, "7", "Gt", ["5"]) and not rm(True, "x", "Gt", ["5"]) and rm(True, "3", "Lt", ["5"])
    assert not ingest.label_selector_matches(None, {"a": "b"}) and ingest.label_selector_matches({}, {"a": "b"})
    t = {"key": "dedicated", "value": "infra", "effect": "NoSchedule"}
    assert ingest.tolerates({"key": "dedicated", "operator": "Exists"}, t)
    assert ingest.tolerates({"operator": "Exists"}, t) and not ingest.tolerates({"key": "dedicated", "value": "x"}, t)
    assert not ingest.tolerates({"key": "dedicated", "value": "infra", "effect": "NoExecute"}, t)


def test_canonical_node_order_is_zone_round_robin():
    z = lambda v: {"topology.kubernetes.io/zone": v}
    objs = [node("n5", labels=z("b")), node("n1", labels=z("a")), node("n3", labels=z("a")), node("n2", labels=z("b")), node("n4")]
    order = [n["metadata"]["name"] for n in ingest.canonical_node_order(objs)]
    assert order == ["n1", "n2", "n4", "n3", "n5"]  # zones in first-seen order a, b, "" ; round robin


def test_ingest_readme_cluster_matches_oracle(ccref):
    # README.md:44-66 demo: 4 nodes x (2 CPU, 4 GB), examples/pod.yaml -> 52 = 13 x 4, "Insufficient cpu"
    objs = [node(f"kube-node-{i}", cpu="2", mem="4G") for i in range(1, 5)]
    snap = ingest.build_snapshot(objs, [], yaml.safe_load(EXAMPLES_POD))
    assert snap.pod.req.tolist() == [150, 104857600, 0] and snap.pod.nz_mcpu == 150
    r = ccref.run(M.Profile.default(), snap.nodes, snap.pod)
    assert r.placed == 52 and r.per_node_count.tolist() == [13] * 4
    msg = R.stop_reason(r, 4, 0, taint_reasons=snap.taint_reasons)
    assert msg.startswith("Unschedulable: 0/4 nodes are available: 4 Insufficient cpu.")


def test_ingest_existing_pods_taints_and_selectors(ccref):
    objs = [node("a", labels={"disk": "ssd"}), node("b", labels={"disk": "hdd"}),
            node("c", labels={"disk": "ssd"}, taints=[{"key": "dedicated", "value": "infra", "effect": "NoSchedule"}]),
            node("d", labels={"disk": "ssd"}, unschedulable=True)]
    pods = [running_pod("p1", "a", cpu="500m", mem="1Gi"), running_pod("p2", "a"),  # p2: no requests -> NonZero defaults only
            running_pod("done", "b", cpu="1", phase="Succeeded"), running_pod("elsewhere", "zzz", cpu="1")]
    sim = yaml.safe_load(EXAMPLES_POD)
    sim["spec"]["nodeSelector"] = {"disk": "ssd"}
    snap = ingest.build_snapshot(objs, pods, sim)
    i = snap.names.index("a")
    assert snap.nodes.req[0][i] == 500 and snap.nodes.nz_mcpu[i] == 600 and snap.nodes.pod_count[i] == 2
    assert snap.nodes.nz_mem[i] == (1 << 30) + 200 * (1 << 20) and snap.nodes.req[0][snap.names.index("b")] == 0
    r = ccref.run(M.Profile.default(), snap.nodes, snap.pod)
    # only node a is feasible: (2000 - 500) // 150 = 10
    assert r.placed == 10 and r.per_node_count[i] == 10
    msg = R.stop_reason(r, 4, 0, taint_reasons=snap.taint_reasons)
    assert "1 node(s) had untolerated taint {dedicated: infra}" in msg and "1 node(s) were unschedulable" in msg
    assert "1 node(s) didn't match Pod's node affinity/selector" in msg and "1 Insufficient cpu" in msg
    assert ingest.build_snapshot(objs, pods, sim, exclude_nodes=["a"]).names == ["b", "c", "d"]


def test_ingest_self_affinity_matches_reference_fixture(ccref):
    # test/benchmark/pod_colocation_test.go:99-190 through the string path
    objs = [node(f"node{z}-{i}", cpu="1", mem="1000", pods="30", labels={"topology-domain": f"zone{z}"}) for z in (1, 2, 3) for i in (1, 2, 3)]
    sim = {"kind": "Pod", "metadata": {"name": "pod-affinity", "namespace": "default", "labels": {"key": "value"}},
           "spec": {"containers": [{"resources": {"requests": {"cpu": "10m", "memory": "10"}}}],
                    "affinity": {"podAffinity": {"requiredDuringSchedulingIgnoredDuringExecution": [
                        {"topologyKey": "topology-domain", "labelSelector": {"matchLabels": {"key": "value"}}}]}}}}
    snap = ingest.build_snapshot(objs, [], sim)
    assert snap.pod.ipa.self_aff and snap.pod.ipa.score_self == [1] and snap.pod.ipa.self_entries == [1]
    r = ccref.run(M.Profile.default(), snap.nodes, snap.pod, max_limit=100)  # the fixture's profile keeps the MultiPoint defaults
    assert r.placed == 90 and r.per_node_count.tolist() == [30, 30, 30, 0, 0, 0, 0, 0, 0]


def test_pretty_printer_format():
    snap = ingest.build_snapshot([node("n1"), node("n2")], [], yaml.safe_load(EXAMPLES_POD))
    res = M.RunResult(placed=3, stop=M.STOP_LIMIT, per_node_count=np.array([2, 1], np.int32), log=np.array([1, 0, 0], np.int32),
                      hist=np.zeros(M.NREASON, np.int64), hist_taintset=np.zeros(1, np.int64), n_code_unschedulable=0)
    rev = cli.build_review(yaml.safe_load(EXAMPLES_POD), snap, res, 3)
    assert cli.pretty(rev, False) == "3\n"
    txt = cli.pretty(rev, True)
    assert "small-pod pod requirements:\n\t- CPU: 150m\n\t- Memory: 100Mi\n" in txt
    assert "The cluster can schedule 3 instance(s) of the pod small-pod." in txt
    assert "Termination reason: LimitReached: Maximum number of pods simulated: 3" in txt
    assert "\t- n2: 1 instance(s)\n\t- n1: 2 instance(s)" in txt  # first-placement order (report.go:157-171)


@pytest.mark.gpu
def test_cli_end_to_end_readme_demo(tmp_path):
    (tmp_path / "pod.yaml").write_text(EXAMPLES_POD)
    (tmp_path / "cluster.yaml").write_text(yaml.safe_dump({"kind": "List", "items": [node(f"kube-node-{i}", cpu="2", mem="4G") for i in range(1, 5)]}))
    buf = io.StringIO()
    assert cli.main(["--podspec", str(tmp_path / "pod.yaml"), "--snapshot", str(tmp_path / "cluster.yaml"), "--verbose"], out=buf) == 0
    txt = buf.getvalue()
    assert "The cluster can schedule 52 instance(s) of the pod small-pod." in txt
    assert "Termination reason: Unschedulable: 0/4 nodes are available: 4 Insufficient cpu." in txt
    assert txt.count("13 instance(s)") == 4
    buf = io.StringIO()
    cli.main(["--podspec", str(tmp_path / "pod.yaml"), "--snapshot", str(tmp_path / "cluster.yaml"), "--max-limit", "7", "-o", "json"], out=buf)
    rev = json.loads(buf.getvalue())
    assert rev["status"]["replicas"] == 7 and rev["status"]["failReason"]["failType"] == "LimitReached"
    assert sum(r["replicas"] for r in rev["status"]["pods"][0]["replicasOnNodes"]) == 7


def test_replicas_on_nodes_with_a_capped_log():
    # ccsim_report.log_cap < placed (1M nodes x 110 pods exceeds the 2^26 cap): nodes first placed beyond the cap must
    # still be listed, so that the pretty-printer's headline (the sum of the list) equals status.replicas
    per_node = np.array([2, 3, 0, 1], np.int32)
    got = R.replicas_on_nodes(per_node, ["a", "b", "c", "d"], np.array([1, 1], np.int32))  # log cut after two placements
    assert [g["nodeName"] for g in got] == ["b", "a", "d"]
    assert sum(g["replicas"] for g in got) == int(per_node.sum())


def test_non_zero_requests_with_pod_level_resources():
    # types.go:1095-1124: pod-level memory is set, so the 100m / 200Mi defaults apply only to a resource that NOBODY names.
    # containers [cpu 500m, -]: cpu is named by a container -> no per-container default: Non0CPU = 500m (not 600m);
    # memory comes from the pod level: 1Gi
    spec = {"resources": {"requests": {"memory": "1Gi"}},
            "containers": [{"resources": {"requests": {"cpu": "500m"}}}, {"resources": {}}]}
    req, nz_cpu, nz_mem = ingest.pod_requests(spec, ["cpu", "memory"])
    assert (req["cpu"], req["memory"], nz_cpu, nz_mem) == (500, 1 << 30, 500, 1 << 30)
    # nobody names cpu: the default stands in for every container
    spec = {"resources": {"requests": {"memory": "1Gi"}}, "containers": [{"resources": {}}, {"resources": {}}]}
    assert ingest.pod_requests(spec, ["cpu", "memory"])[1:] == (200, 1 << 30)
    # no pod-level requests: per-container defaults as ever
    spec = {"containers": [{"resources": {"requests": {"cpu": "500m"}}}, {"resources": {}}]}
    assert ingest.pod_requests(spec, ["cpu", "memory"])[1:] == (600, 2 * 200 * 1024 * 1024)
