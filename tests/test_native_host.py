"""The native (C++) host above the C ABI -- cluster-capacity_amd/host/: ingest, CLI, report -- against the Python host it
mirrors (cluster_capacity_amd/{ingest,cli,report}.py) on the same objects.  CPU: the integer snapshot (`--dump-snapshot`)
and the rendered reports (`--fake-result`) must be identical, from JSON and from YAML input.  GPU: the binary end to end."""
import dataclasses
import json
import os
import subprocess

import numpy as np
import pytest

from helpers import SUBPROC_TIMEOUT
import yaml

from cluster_capacity_amd import build as B, cli, ingest, model as M
from test_ingest_cli import EXAMPLES_POD, node, running_pod


@pytest.fixture(scope="module")
def native():
    return B.build_host()


def _pod_dump(p):
    lst = lambda a: None if a is None else [int(x) for x in a]
    term = lambda t: [{"col": int(c), "table": lst(tab)} for c, tab in t]
    ipa = None
    if p.ipa is not None:
        a = p.ipa
        ipa = {"key_cols": lst(a.key_cols), "key_ndom": lst(a.key_ndom), "aff_keys": lst(a.aff_keys), "self_aff": bool(a.self_aff),
               "aff_existing": lst(a.aff_existing), "anti_keys": lst(a.anti_keys), "anti_self": lst(a.anti_self),
               "anti_existing": [lst(x) for x in a.anti_existing], "exist_anti": [lst(x) for x in a.exist_anti],
               "score_existing": [lst(x) for x in a.score_existing], "score_self": lst(a.score_self), "self_entries": lst(a.self_entries),
               "entries_existing": int(a.entries_existing)}
    return {"req": lst(p.req), "nz_mcpu": int(p.nz_mcpu), "nz_mem": int(p.nz_mem), "has_scalar_entries": bool(p.has_scalar_entries),
            "taint_filter_ok": lst(p.taint_filter_ok), "taint_prefer_cnt": lst(p.taint_prefer_cnt),
            "tolerates_unschedulable": bool(p.tolerates_unschedulable), "affinity_filter_active": bool(p.affinity_filter_active),
            "has_node_selector": bool(p.has_node_selector), "has_required_terms": bool(p.has_required_terms),
            "node_selector": term(p.node_selector), "required": [term(t) for t in p.required],
            "preferred": [{"weight": int(w), "term": term(t)} for w, t in p.preferred],
            "spread": [{"col": int(k.col), "max_skew": int(k.max_skew), "min_domains": int(k.min_domains), "hard": bool(k.hard),
                        "self_match": bool(k.self_match), "is_hostname": bool(k.is_hostname), "n_domains": int(k.n_domains),
                        "node_match_count": lst(k.node_match_count), "node_included": lst(k.node_included)} for k in p.spread],
            "soft_relaxed": bool(getattr(p, "soft_relaxed", False)),
            "ipa": ipa, "has_host_ports": bool(p.has_host_ports), "host_ports_conflict": lst(p.host_ports_conflict),
            "image_score": lst(p.image_score),
            "volume_veto": lst(p.volume_veto), "volume_exclusive": bool(p.volume_exclusive), "prefilter_reject": p.prefilter_reject,
            "rwop_capacity_one": bool(p.rwop_capacity_one),
            "preempt": {"priority": p.preempt.priority, "never": p.preempt.never, "victim_count": lst(p.preempt.victim_count),
                        "victim_req": [lst(v) for v in p.preempt.victim_req], "ports_conflict_rest": lst(p.preempt.ports_conflict_rest),
                        "victim_interacts": lst(p.preempt.victim_interacts), "volume_veto_rest": lst(p.preempt.volume_veto_rest)}}


def py_dump(snap):
    """The structure snapshot_json() of host/snapshot.hpp emits, from the Python Snapshot."""
    n = snap.nodes
    lst = lambda a: None if a is None else [int(x) for x in a]
    d = {
        "names": list(snap.names), "res_names": ["cpu", "memory", "ephemeral-storage"] + list(snap.scalar_names),
        "scalar_names": list(snap.scalar_names), "taint_reasons": list(snap.taint_reasons),
        "alloc": [lst(c) for c in n.alloc], "req": [lst(c) for c in n.req], "label_cols": [lst(c) for c in n.label_cols],
        "alloc_pods": lst(n.alloc_pods), "pod_count": lst(n.pod_count), "taintset_id": lst(n.taintset_id),
        "nz_mcpu": lst(n.nz_mcpu), "nz_mem": lst(n.nz_mem), "unschedulable": lst(n.unschedulable),
        "pod": _pod_dump(snap.pod)}
    if len(snap.pods) > 1:
        d["more_pods"] = [_pod_dump(p) for p in snap.pods[1:]]
    return d


def rich_cluster():
    """Zones, taints of every effect, label variety, existing pods with requests / init containers / affinity terms."""
    rng = np.random.default_rng(5)
    nodes, pods = [], []
    for i in range(23):
        labels = {"topology.kubernetes.io/zone": f"zone-{i % 3}", "topology.kubernetes.io/region": "r1", "kubernetes.io/hostname": f"n{i:02d}",
                  "disk": ["ssd", "hdd", "nvme"][i % 3], "gen": str(3 + i % 5)}
        if i % 7 == 0:
            del labels["disk"]
        taints = []
        if i % 5 == 1:
            taints.append({"key": "dedicated", "value": "infra", "effect": "NoSchedule"})
        if i % 4 == 2:
            taints.append({"key": "maintenance", "value": "soon", "effect": "PreferNoSchedule"})
        if i % 11 == 3:
            taints.append({"key": "flaky", "effect": "NoExecute"})
        nd = node(f"n{i:02d}", cpu=["4", "8", "1500m", "0.5"][i % 4], mem=["8Gi", "16G", "3.5Gi", "1e9"][i % 4], pods=str(8 + i % 5),
                  labels=labels, taints=taints, unschedulable=(i == 9))
        nd["status"]["allocatable"]["example.com/gpu"] = str(i % 3)
        nd["status"]["allocatable"]["hugepages-2Mi"] = "64Mi"
        nodes.append(nd)
    anti = {"podAntiAffinity": {"requiredDuringSchedulingIgnoredDuringExecution": [
        {"topologyKey": "kubernetes.io/hostname", "labelSelector": {"matchLabels": {"app": "web"}}}]}}
    pref = {"podAffinity": {"preferredDuringSchedulingIgnoredDuringExecution": [
        {"weight": 30, "podAffinityTerm": {"topologyKey": "topology.kubernetes.io/zone", "labelSelector": {"matchExpressions": [
            {"key": "app", "operator": "In", "values": ["web", "api"]}]}}}]},
        "podAntiAffinity": {"preferredDuringSchedulingIgnoredDuringExecution": [
            {"weight": 5, "podAffinityTerm": {"topologyKey": "kubernetes.io/hostname", "labelSelector": {"matchLabels": {"tier": "frontend"}}}}]}}
    for j in range(40):
        nd = f"n{int(rng.integers(0, 25)):02d}"  # some land on nodes that do not exist
        p = running_pod(f"p{j}", nd, cpu=[None, "250m", "1", "1500u"][j % 4], mem=[None, "64Mi", "1Gi", "100M"][j % 4],
                        labels={"app": ["web", "api", "db"][j % 3], "tier": ["frontend", "backend"][j % 2]},
                        ns=["default", "other"][j % 5 == 0], phase=["Running", "Pending", "Succeeded", "Failed"][j % 9 if j % 9 < 4 else 0],
                        affinity=[None, anti, pref][j % 3])
        if j % 6 == 0:
            p["spec"]["initContainers"] = [{"name": "init", "resources": {"requests": {"cpu": "2", "memory": "10Mi"}}}]
        if j % 8 == 0:
            p["spec"]["overhead"] = {"cpu": "10m", "memory": "1Mi"}
        if j % 10 == 0:
            p["metadata"]["deletionTimestamp"] = "2025-01-01T00:00:00Z"
        pods.append(p)
    return nodes, pods


def rich_pod():
    pod = yaml.safe_load(EXAMPLES_POD)
    pod["metadata"]["labels"] = {"app": "web", "tier": "frontend"}
    spec = pod["spec"]
    spec["containers"].append({"name": "side", "resources": {"requests": {"example.com/gpu": "1", "hugepages-2Mi": "2Mi"}}})
    spec["initContainers"] = [{"name": "init", "resources": {"requests": {"cpu": "300m", "memory": "10Mi"}}}]
    spec["overhead"] = {"cpu": "5m"}
    spec["nodeSelector"] = {"topology.kubernetes.io/region": "r1"}
    spec["tolerations"] = [{"key": "dedicated", "operator": "Equal", "value": "infra", "effect": "NoSchedule"},
                           {"key": "node.kubernetes.io/unschedulable", "operator": "Exists"}]
    spec["affinity"] = {
        "nodeAffinity": {
            "requiredDuringSchedulingIgnoredDuringExecution": {"nodeSelectorTerms": [
                {"matchExpressions": [{"key": "disk", "operator": "In", "values": ["ssd", "nvme"]}, {"key": "gen", "operator": "Gt", "values": ["3"]}]},
                {"matchExpressions": [{"key": "disk", "operator": "DoesNotExist"}]},
                {"matchFields": [{"key": "metadata.name", "operator": "In", "values": ["n01", "n02"]}]}]},
            "preferredDuringSchedulingIgnoredDuringExecution": [
                {"weight": 10, "preference": {"matchExpressions": [{"key": "gen", "operator": "Lt", "values": ["6"]}]}},
                {"weight": 40, "preference": {"matchExpressions": [{"key": "disk", "operator": "NotIn", "values": ["hdd"]}]}}]},
        "podAffinity": {"preferredDuringSchedulingIgnoredDuringExecution": [
            {"weight": 20, "podAffinityTerm": {"topologyKey": "topology.kubernetes.io/zone", "labelSelector": {"matchLabels": {"app": "web"}}}}]},
        "podAntiAffinity": {"requiredDuringSchedulingIgnoredDuringExecution": [
            {"topologyKey": "kubernetes.io/hostname", "labelSelector": {"matchLabels": {"app": "web"}}, "namespaces": ["default", "other"]}]}}
    spec["topologySpreadConstraints"] = [
        {"maxSkew": 2, "topologyKey": "topology.kubernetes.io/zone", "whenUnsatisfiable": "DoNotSchedule", "labelSelector": {"matchLabels": {"app": "web"}}},
        {"maxSkew": 1, "minDomains": 2, "topologyKey": "kubernetes.io/hostname", "whenUnsatisfiable": "ScheduleAnyway",
         "nodeAffinityPolicy": "Ignore", "labelSelector": {"matchExpressions": [{"key": "tier", "operator": "Exists"}]}}]
    return pod


def ports_images_case():
    """Host ports (a conflict on one node through 0.0.0.0, none through another ip / protocol) and node images."""
    mb = 1024 * 1024
    nodes = [node(f"w{i}") for i in range(5)]
    nodes[0]["status"]["images"] = [{"names": ["gcr.io/40:latest", "gcr.io/40@sha256:abc"], "sizeBytes": 40 * mb}]
    nodes[1]["status"]["images"] = [{"names": ["gcr.io/250:latest"], "sizeBytes": 250 * mb}, {"names": ["gcr.io/40:latest"], "sizeBytes": 41 * mb}]
    nodes[2]["status"]["images"] = [{"names": ["gcr.io/2000"], "sizeBytes": 2000 * mb}]
    pods = [running_pod("holder", "w3"), running_pod("other-ip", "w4"), running_pod("udp", "w2")]
    pods[0]["spec"]["containers"][0]["ports"] = [{"containerPort": 80, "hostPort": 8080}]
    pods[1]["spec"]["containers"][0]["ports"] = [{"containerPort": 80, "hostPort": 8080, "hostIP": "10.0.0.9"}]
    pods[2]["spec"]["containers"][0]["ports"] = [{"containerPort": 80, "hostPort": 8080, "protocol": "UDP"}]
    pod = yaml.safe_load(EXAMPLES_POD)
    pod["spec"]["containers"][0]["image"] = "gcr.io/40"
    pod["spec"]["containers"][0]["ports"] = [{"containerPort": 80, "hostPort": 8080, "hostIP": "10.0.0.1"}]
    pod["spec"]["containers"].append({"name": "big", "image": "gcr.io/250:latest"})
    return nodes, pods, pod, []


def disks_case():
    """VolumeRestrictions (round 5): w1's pod mounts the PD read-write (conflict), w3's read-only (fine next to the template's read-only
    mount); the template's EBS volume makes its clones exclusive."""
    nodes = [node(f"w{i}", cpu="4", mem="8Gi", pods="20", labels={"kubernetes.io/hostname": f"w{i}"}) for i in range(6)]
    rw = running_pod("writer", "w1", cpu="100m")
    rw["spec"]["volumes"] = [{"name": "d", "gcePersistentDisk": {"pdName": "disk-1"}}, {"name": "tmp", "emptyDir": {}}]
    ro = running_pod("reader", "w3", cpu="100m")
    ro["spec"]["volumes"] = [{"name": "d", "gcePersistentDisk": {"pdName": "disk-1", "readOnly": True}}]
    pod = yaml.safe_load(EXAMPLES_POD)
    pod["spec"]["volumes"] = [{"name": "d", "gcePersistentDisk": {"pdName": "disk-1", "readOnly": True}}, {"name": "e", "awsElasticBlockStore": {"volumeID": "vol-1"}}]
    return nodes, [rw, ro], pod, []


CASES = {
    "disks": lambda: disks_case(),
    "readme": lambda: ([node(f"kube-node-{i}", cpu="2", mem="4G") for i in range(1, 5)], [], yaml.safe_load(EXAMPLES_POD), []),
    "taints-selectors": lambda: (
        [node("a", labels={"disk": "ssd"}), node("b", labels={"disk": "hdd"}),
         node("c", labels={"disk": "ssd"}, taints=[{"key": "dedicated", "value": "infra", "effect": "NoSchedule"}]),
         node("d", labels={"disk": "ssd"}, unschedulable=True)],
        [running_pod("p1", "a", cpu="500m", mem="1Gi"), running_pod("p2", "a"), running_pod("done", "b", cpu="1", phase="Succeeded"),
         running_pod("elsewhere", "zzz", cpu="1")],
        dict(yaml.safe_load(EXAMPLES_POD), spec=dict(yaml.safe_load(EXAMPLES_POD)["spec"], nodeSelector={"disk": "ssd"})), ["b"]),
    "rich": lambda: (*rich_cluster(), rich_pod(), ["n04"]),
    "ports-images": lambda: ports_images_case(),
}


def _write(tmp_path, fmt, nodes, pods, pod):
    if fmt == "json":
        (tmp_path / "cluster.json").write_text(json.dumps({"kind": "List", "items": nodes + pods}))
        (tmp_path / "pod.json").write_text(json.dumps(pod))
        return str(tmp_path / "pod.json"), [str(tmp_path / "cluster.json")]
    (tmp_path / "nodes.yaml").write_text(yaml.safe_dump({"kind": "NodeList", "items": nodes}))
    (tmp_path / "pods.yaml").write_text(yaml.safe_dump_all(pods) if pods else "")
    (tmp_path / "pod.yaml").write_text(yaml.safe_dump(pod, default_flow_style=False))
    return str(tmp_path / "pod.yaml"), [str(tmp_path / "nodes.yaml"), str(tmp_path / "pods.yaml")]


def _run(native, args):
    p = subprocess.run([native] + args, capture_output=True, text=True, timeout=SUBPROC_TIMEOUT)
    assert p.returncode == 0, p.stderr
    return p.stdout


@pytest.mark.parametrize("fmt", ["json", "yaml"])
@pytest.mark.parametrize("case", sorted(CASES))
def test_native_ingest_equals_python_ingest(native, tmp_path, case, fmt):
    nodes, pods, pod, exclude = CASES[case]()
    podspec, snaps = _write(tmp_path, fmt, nodes, pods, pod)
    args = ["--podspec", podspec] + [x for s in snaps for x in ("--snapshot", s)] + ["--dump-snapshot", "-"]
    if exclude:
        args += ["--exclude-nodes", ",".join(exclude)]
    got = json.loads(_run(native, args))
    ref = py_dump(ingest.build_snapshot(*cli.load_objects(snaps), cli.parse_pod_spec(podspec), exclude))
    got.pop("label_keys")
    assert got.keys() == ref.keys()
    for k in ref:
        assert got[k] == ref[k], k


def _fake_result(snap, limit_stop):
    n = len(snap.names)
    rng = np.random.default_rng(3)
    per = rng.integers(0, 4, n).astype(np.int32)
    log = rng.permutation(np.repeat(np.arange(n), per)).astype(np.int32)
    hist = np.zeros(M.NREASON, np.int64)
    hist[M.R_TOO_MANY_PODS], hist[M.R_RES0], hist[M.R_RES0 + 1], hist[M.R_NODEAFFINITY], hist[M.R_UNSCHEDULABLE] = 2, n, 1, 3, 1
    hist[M.R_PTS_SKEW], hist[M.R_IPA_ANTI] = 2, 1
    if snap.scalar_names:
        hist[M.R_RES0 + 3] = 4
    ht = np.zeros(len(snap.taint_reasons), np.int64)
    ht[-1] = 2
    return M.RunResult(placed=int(per.sum()), stop=M.STOP_LIMIT if limit_stop else M.STOP_UNSCHEDULABLE, per_node_count=per, log=log,
                       hist=hist, hist_taintset=ht, n_code_unschedulable=min(n, 5))


@pytest.mark.parametrize("limit_stop", [False, True])
@pytest.mark.parametrize("case", ["readme", "rich"])
def test_native_report_equals_python_report(native, tmp_path, case, limit_stop):
    nodes, pods, pod, exclude = CASES[case]()
    podspec, snaps = _write(tmp_path, "json", nodes, pods, pod)
    pypod = cli.parse_pod_spec(podspec)
    snap = ingest.build_snapshot(*cli.load_objects(snaps), pypod, exclude)
    res = _fake_result(snap, limit_stop)
    (tmp_path / "result.json").write_text(json.dumps({
        "placed": res.placed, "stop": res.stop, "n_code_unschedulable": res.n_code_unschedulable, "per_node_count": res.per_node_count.tolist(),
        "log": res.log.tolist(), "hist": res.hist.tolist(), "hist_taintset": res.hist_taintset.tolist()}))
    limit = 17 if limit_stop else 0
    base = ["--podspec", podspec, "--snapshot", snaps[0], "--fake-result", str(tmp_path / "result.json"), "--max-limit", str(limit)]
    if exclude:
        base += ["--exclude-nodes", ",".join(exclude)]
    review = cli.build_review(pypod, snap, res, limit)
    assert _run(native, base) == cli.pretty(review, False)
    assert _run(native, base + ["--verbose"]) == cli.pretty(review, True)
    for fmt, load in (("json", json.loads), ("yaml", yaml.safe_load)):
        got = load(_run(native, base + ["-o", fmt]))
        got["status"].pop("creationTimestamp"), review["status"].pop("creationTimestamp", None)
        assert got["status"] == json.loads(json.dumps(review["status"]))
        assert got["spec"]["podRequirements"] == json.loads(json.dumps(review["spec"]["podRequirements"]))
        assert got["spec"]["replicas"] == 0 and got["spec"]["templates"][0]["metadata"] == pypod["metadata"]


def test_native_host_fails_loudly_without_the_engine(native, tmp_path):
    nodes, pods, pod, _ = CASES["readme"]()
    podspec, snaps = _write(tmp_path, "json", nodes, pods, pod)
    env = dict(os.environ, CCSIM_LIB="/nonexistent/libccsim.so", HIP_VISIBLE_DEVICES="-1")
    p = subprocess.run([native, "--podspec", podspec, "--snapshot", snaps[0]], capture_output=True, text=True, env=env, timeout=SUBPROC_TIMEOUT)
    assert p.returncode != 0 and ("no CPU fallback" in p.stderr or "ccsim_create failed" in p.stderr)
    p = subprocess.run([native, "--snapshot", snaps[0]], capture_output=True, text=True)
    assert p.returncode == 2 and "Pod spec file is missing" in p.stderr


@pytest.mark.gpu
def test_native_cli_end_to_end(native, tmp_path):
    """README.md:44-66 through the C++ host, the C ABI and the HIP engine; then the same run through the Python host."""
    nodes, pods, pod, _ = CASES["readme"]()
    podspec, snaps = _write(tmp_path, "yaml", nodes, pods, pod)
    txt = _run(native, ["--podspec", podspec, "--snapshot", snaps[0], "--verbose"])
    assert "The cluster can schedule 52 instance(s) of the pod small-pod." in txt
    assert "Termination reason: Unschedulable: 0/4 nodes are available: 4 Insufficient cpu." in txt and txt.count("13 instance(s)") == 4
    rev = json.loads(_run(native, ["--podspec", podspec, "--snapshot", snaps[0], "--max-limit", "7", "-o", "json"]))
    assert rev["status"]["replicas"] == 7 and rev["status"]["failReason"]["failType"] == "LimitReached"
    # a coupled pod on the rich cluster: native and Python hosts drive the same engine to the same review
    import io
    nodes, pods, pod, exclude = CASES["rich"]()
    podspec, snaps = _write(tmp_path, "json", nodes, pods, pod)
    args = ["--podspec", podspec, "--snapshot", snaps[0], "--exclude-nodes", ",".join(exclude), "-o", "json"]
    got = json.loads(_run(native, args))
    buf = io.StringIO()
    assert cli.main(args, out=buf) == 0
    ref = json.loads(buf.getvalue())
    got["status"].pop("creationTimestamp"), ref["status"].pop("creationTimestamp")
    assert got["status"] == ref["status"]


KUBECTL_STYLE_YAML = """\
# kubectl get nodes,pods -o yaml (abridged)
apiVersion: v1
items:
- apiVersion: v1
  kind: Node
  metadata:
    annotations:
      kubectl.kubernetes.io/last-applied-configuration: |
        {"apiVersion":"v1","kind":"Node","metadata":{"name":"n1"}}
      note: >-
        folded text
        continues here
      node.alpha.kubernetes.io/ttl: "0"
      "quoted: key": 'it''s quoted'
    creationTimestamp: "2025-01-01T00:00:00Z"
    labels:
      kubernetes.io/hostname: n1   # trailing comment
      topology.kubernetes.io/zone: zone-a
      numeric-looking: "123"
      empty-value: ""
    name: n1
  spec:
    taints:
    - effect: NoSchedule
      key: dedicated
      value: infra
    - {effect: PreferNoSchedule, key: maintenance}
    podCIDRs: [10.0.0.0/24, "10.0.1.0/24"]
    unschedulable: true
  status:
    allocatable:
      cpu: 3920m
      ephemeral-storage: "47093746742"
      memory: 15842200Ki
      pods: "110"
      fractional: 0.5
    capacity: {}
    conditions: []
    images:
    -
      names:
      - registry.k8s.io/pause:3.9
      sizeBytes: 322000
- apiVersion: v1
  kind: Pod
  metadata: {name: p1, namespace: kube-system, labels: {app: dns}}
  spec:
    nodeName: n1
    containers:
    - name: c
      args: ["--flag=a: b", '--x']
      resources:
        requests: {cpu: 100m, memory: 70Mi}
      command:
      - /bin/sh
      - -c
      - echo hello
  status: {phase: Running}
kind: List
metadata:
  resourceVersion: ""
---
kind: Pod
metadata:
  name: long-plain
  description: this plain scalar is long enough that the emitter
    wrapped it onto a second line
spec:
  nodeName: null
  priority: 0
  enableServiceLinks: false
...
"""


def _stringify(v):
    """PyYAML types -> the native model: numbers keep their text, everything else as is."""
    if isinstance(v, dict):
        return {str(k): _stringify(x) for k, x in v.items()}
    if isinstance(v, list):
        return [_stringify(x) for x in v]
    if isinstance(v, bool) or v is None or isinstance(v, str):
        return v
    return v  # int / float: json round-trips them


def test_native_yaml_reader_agrees_with_pyyaml_on_kubectl_style_input(native, tmp_path):
    (tmp_path / "dump.yaml").write_text(KUBECTL_STYLE_YAML)
    got = json.loads(_run(native, ["--parse", str(tmp_path / "dump.yaml")]))
    ref = [_stringify(d) for d in yaml.safe_load_all(KUBECTL_STYLE_YAML) if d]
    assert got == ref
    # and the ingest of that file: one kept node with its quantities interpreted exactly
    (tmp_path / "pod.yaml").write_text(EXAMPLES_POD)
    d = json.loads(_run(native, ["--podspec", str(tmp_path / "pod.yaml"), "--snapshot", str(tmp_path / "dump.yaml"), "--dump-snapshot", "-"]))
    assert d["names"] == ["n1"] and d["alloc"][0] == [3920] and d["alloc"][1] == [15842200 * 1024] and d["alloc"][2] == [47093746742]
    assert d["alloc_pods"] == [110] and d["req"][0] == [100] and d["req"][1] == [70 << 20] and d["unschedulable"] == [1]
    assert d["taint_reasons"] == ["node(s) had untolerated taint {dedicated: infra}"] and d["pod"]["taint_prefer_cnt"] == [1]


def test_native_quantity_arithmetic(native, tmp_path):
    """quantity.go:813-834 through the native ingest: Value / MilliValue round UP, every suffix family."""
    q = {"cpu": ["150m", "2", "0.1", "1500u", "100n", "1e-1", "2E0", "0.0005"], "memory": ["100Mi", "4Gi", "1e3", "1.5Ki", "100m", "4G", "1E", "0.5"]}
    nodes = [node(f"n{i}", cpu=q["cpu"][i], mem=q["memory"][i]) for i in range(8)]
    (tmp_path / "c.json").write_text(json.dumps({"kind": "List", "items": nodes}))
    (tmp_path / "pod.yaml").write_text(EXAMPLES_POD)
    d = json.loads(_run(native, ["--podspec", str(tmp_path / "pod.yaml"), "--snapshot", str(tmp_path / "c.json"), "--dump-snapshot", "-"]))
    assert d["alloc"][0] == [ingest.milli_value(x) for x in q["cpu"]] == [150, 2000, 100, 2, 1, 100, 2000, 1]
    assert d["alloc"][1] == [ingest.value(x) for x in q["memory"]] == [104857600, 4 << 30, 1000, 1536, 1, 4_000_000_000, 10**18, 1]


SCHED_CONFIG = """\
apiVersion: kubescheduler.config.k8s.io/v1
kind: KubeSchedulerConfiguration
percentageOfNodesToScore: 40
profiles:
- schedulerName: default-scheduler
  plugins:
    multiPoint:
      enabled:
      - name: TaintToleration
        weight: 7
      disabled:
      - name: ImageLocality
    filter:
      disabled:
      - name: NodeUnschedulable
    score:
      disabled:
      - name: NodeResourcesBalancedAllocation
      enabled:
      - name: NodeAffinity
      - name: PodTopologySpread
        weight: 5
  pluginConfig:
  - name: NodeResourcesFit
    args:
      scoringStrategy:
        type: LeastAllocated
        resources:
        - name: memory
          weight: 3
        - name: cpu
          weight: 1
  - name: InterPodAffinity
    args:
      hardPodAffinityWeight: 10
"""


def _profile(native, tmp_path, text=None, extra=()):
    args = ["--dump-profile", *extra]
    if text is not None:
        (tmp_path / "sched.yaml").write_text(text)
        args += ["--default-config", str(tmp_path / "sched.yaml")]
    return json.loads(_run(native, args))


def test_native_scheduler_config(native, tmp_path):
    """--default-config (options.go:73): plugin sets, weights, scoring resources, percentageOfNodesToScore."""
    d = _profile(native, tmp_path)
    p = M.Profile.default()
    assert d == {"filter_mask": p.filter_mask, "w_taint": 3, "w_nodeaffinity": 2, "w_fit": 1, "w_balanced": 1, "w_topologyspread": 2,
                 "w_interpodaffinity": 2, "w_imagelocality": 1, "fit_res": [0, 1], "fit_res_w": [1, 1], "bal_res": [0, 1],
                 "percentage_of_nodes_to_score": 100, "hard_pod_affinity_weight": 1}
    d = _profile(native, tmp_path, SCHED_CONFIG)
    assert d["filter_mask"] == M.F_ALL & ~M.F_UNSCHEDULABLE
    assert (d["w_taint"], d["w_nodeaffinity"], d["w_fit"], d["w_balanced"], d["w_topologyspread"], d["w_interpodaffinity"]) == (7, 1, 1, 0, 5, 2)
    assert d["fit_res"] == [1, 0] and d["fit_res_w"] == [3, 1] and d["bal_res"] == [0, 1]
    assert d["percentage_of_nodes_to_score"] == 40 and d["hard_pod_affinity_weight"] == 10
    assert _profile(native, tmp_path, SCHED_CONFIG, ["--percentage-of-nodes-to-score", "100"])["percentage_of_nodes_to_score"] == 100
    fit_only = """{"kind": "KubeSchedulerConfiguration", "profiles": [{"percentageOfNodesToScore": 0, "plugins": {
        "multiPoint": {"disabled": [{"name": "*"}], "enabled": [{"name": "NodeResourcesFit"}]}}}]}"""
    d = _profile(native, tmp_path, fit_only)
    q = M.Profile.fit_only()
    assert d["filter_mask"] == q.filter_mask and d["w_fit"] == 1 and d["percentage_of_nodes_to_score"] == 0
    assert all(d[k] == 0 for k in ("w_taint", "w_nodeaffinity", "w_balanced", "w_topologyspread", "w_interpodaffinity"))
    for bad in ('{"profiles": [{"plugins": {"score": {"enabled": [{"name": "NoSuchPlugin"}]}}}]}',
                '{"profiles": [{"pluginConfig": [{"name": "NodeResourcesFit", "args": {"scoringStrategy": {"type": "MostAllocated"}}}]}]}',
                '{"percentageOfNodesToScore": 101}', '{"profiles": [{}, {}]}'):
        (tmp_path / "bad.json").write_text(bad)
        p = subprocess.run([native, "--dump-profile", "--default-config", str(tmp_path / "bad.json")], capture_output=True, text=True)
        assert p.returncode == 1 and "scheduler config" in p.stderr


@pytest.mark.gpu
def test_native_scheduler_config_end_to_end(native, tmp_path, ccref):
    """A non-default profile through the native host == the same profile through the Python host's engine binding."""
    from cluster_capacity_amd import capi
    nodes, pods, pod, exclude = CASES["taints-selectors"]()
    nodes += [node(f"x{i}", cpu=["3", "5", "7"][i % 3], mem=["6Gi", "9Gi", "20Gi"][i % 3], labels={"disk": "ssd"}) for i in range(9)]
    podspec, snaps = _write(tmp_path, "json", nodes, pods, pod)
    (tmp_path / "sched.yaml").write_text(SCHED_CONFIG.replace("percentageOfNodesToScore: 40", "percentageOfNodesToScore: 100"))
    rev = json.loads(_run(native, ["--podspec", podspec, "--snapshot", snaps[0], "--default-config", str(tmp_path / "sched.yaml"), "-o", "json"]))
    snap = ingest.build_snapshot(*cli.load_objects(snaps), cli.parse_pod_spec(podspec), hard_pod_affinity_weight=10)
    prof = M.Profile(filter_mask=M.F_ALL & ~M.F_UNSCHEDULABLE, w_taint=7, w_nodeaffinity=1, w_fit=1, w_balanced=0, w_topologyspread=5,
                     w_interpodaffinity=2, fit_res=(1, 0), fit_res_w=(3, 1))
    ref = ccref.run(prof, snap.nodes, snap.pod)
    got = {r["nodeName"]: r["replicas"] for r in rev["status"]["pods"][0]["replicasOnNodes"]}
    assert rev["status"]["replicas"] == ref.placed
    assert got == {snap.names[i]: int(c) for i, c in enumerate(ref.per_node_count) if c}
    order = [r["nodeName"] for r in rev["status"]["pods"][0]["replicasOnNodes"]]
    _, first = np.unique(ref.log, return_index=True)
    assert order == [snap.names[i] for i in ref.log[np.sort(first)]]  # first-placement order == the oracle's sequence


def _py_profile_dump(cfg):
    from cluster_capacity_amd import schedconfig
    p, hard = schedconfig.profile_from_config(cfg)
    return {"filter_mask": p.filter_mask, "w_taint": p.w_taint, "w_nodeaffinity": p.w_nodeaffinity, "w_fit": p.w_fit, "w_balanced": p.w_balanced,
            "w_topologyspread": p.w_topologyspread, "w_interpodaffinity": p.w_interpodaffinity, "w_imagelocality": p.w_imagelocality,
            "fit_res": list(p.fit_res),
            "fit_res_w": list(p.fit_res_w), "bal_res": list(p.bal_res), "percentage_of_nodes_to_score": p.percentage_of_nodes_to_score,
            "hard_pod_affinity_weight": hard}


def test_scheduler_config_native_and_python_hosts_agree(native, tmp_path):
    """Differential: random KubeSchedulerConfigurations through host/profile.hpp and schedconfig.py."""
    from cluster_capacity_amd import schedconfig
    assert _profile(native, tmp_path, SCHED_CONFIG) == _py_profile_dump(yaml.safe_load(SCHED_CONFIG))
    names = list(schedconfig.PLUGINS) + ["ImageLocality", "VolumeBinding", "NodePorts"]
    rng = np.random.default_rng(11)

    def pset(with_weight):
        out = {}
        if rng.random() < 0.6:
            out["disabled"] = [{"name": str(rng.choice(names + ["*"]))} for _ in range(int(rng.integers(0, 3)))]
        if rng.random() < 0.7:
            en = []
            for _ in range(int(rng.integers(0, 4))):
                e = {"name": str(rng.choice(names))}
                if with_weight and rng.random() < 0.6:
                    e["weight"] = int(rng.integers(0, 12))
                en.append(e)
            out["enabled"] = en
        return out

    for i in range(60):
        prof = {"plugins": {k: pset(k != "filter") for k in ("multiPoint", "filter", "score") if rng.random() < 0.7}}
        if rng.random() < 0.5:
            prof["percentageOfNodesToScore"] = int(rng.integers(0, 101))
        pc = []
        if rng.random() < 0.5:
            res = [{"name": str(n), "weight": int(rng.integers(1, 5))} for n in rng.permutation(["cpu", "memory"])[: int(rng.integers(1, 3))]]
            pc.append({"name": "NodeResourcesFit", "args": {"scoringStrategy": {"type": "LeastAllocated", "resources": res}}})
        if rng.random() < 0.4:
            pc.append({"name": "NodeResourcesBalancedAllocation", "args": {"resources": [{"name": "memory", "weight": 1}, {"name": "cpu", "weight": 1}]}})
        if rng.random() < 0.4:
            pc.append({"name": "InterPodAffinity", "args": {"hardPodAffinityWeight": int(rng.integers(0, 20))}})
        prof["pluginConfig"] = pc
        cfg = {"apiVersion": "kubescheduler.config.k8s.io/v1", "kind": "KubeSchedulerConfiguration", "profiles": [prof]}
        if rng.random() < 0.4:
            cfg["percentageOfNodesToScore"] = int(rng.integers(0, 101))
        text = yaml.safe_dump(cfg) if i % 2 else json.dumps(cfg)
        assert _profile(native, tmp_path, text) == _py_profile_dump(cfg), cfg


@pytest.fixture(scope="module")
def recorder(tmp_path_factory):
    """tests/abi_recorder.c: records what a host passes through the C ABI and schedules nothing (test infrastructure)."""
    out = tmp_path_factory.mktemp("rec") / "libabi_recorder.so"
    here = os.path.dirname(os.path.abspath(__file__))
    subprocess.check_call(["gcc", "-O1", "-shared", "-fPIC", "-o", str(out), os.path.join(here, "abi_recorder.c")])
    return str(out)


@pytest.mark.parametrize("case", sorted(CASES))
def test_native_host_marshals_the_c_abi_like_the_python_binding(native, recorder, tmp_path, case):
    """The same snapshot through main.cpp:marshal() and through capi.marshal_*: every struct field, pointer target and
    requirement table that reaches ccsim_load_nodes / ccsim_set_profile / ccsim_set_pod must be identical."""
    import ctypes as C
    from cluster_capacity_amd import capi
    nodes, pods, pod, exclude = CASES[case]()
    podspec, snaps = _write(tmp_path, "json", nodes, pods, pod)
    args = ["--podspec", podspec, "--snapshot", snaps[0], "--max-limit", "1000", "-o", "json"] + (["--exclude-nodes", ",".join(exclude)] if exclude else [])
    env = dict(os.environ, CCSIM_LIB=recorder, CCSIM_RECORD=str(tmp_path / "native.json"))
    p = subprocess.run([native] + args, capture_output=True, text=True, env=env, timeout=SUBPROC_TIMEOUT)
    assert p.returncode == 0, p.stderr
    native_rec = json.load(open(tmp_path / "native.json"))
    # the recorder's canned result came back through ccsim_report into the review
    rev = json.loads(p.stdout)
    snap = ingest.build_snapshot(*cli.load_objects(snaps), cli.parse_pod_spec(podspec), exclude)
    assert rev["status"]["replicas"] == len(snap.names) and rev["status"]["failReason"]["failType"] == "LimitReached"
    assert [r["nodeName"] for r in rev["status"]["pods"][0]["replicasOnNodes"]] == snap.names

    lib = C.CDLL(recorder)
    os.environ["CCSIM_RECORD"] = str(tmp_path / "python.json")
    try:
        cfg = capi.CConfig()
        cfg.abi_version, cfg.use_graph = capi.ABI_VERSION, 1
        h = C.c_void_p()
        assert lib.ccsim_create(C.byref(cfg), C.byref(h)) == 0
        keep = []
        assert lib.ccsim_load_nodes(h, C.byref(capi.marshal_nodes(snap.nodes, keep))) == 0
        # --max-limit makes the run order-dependent: without an explicit percentageOfNodesToScore the host applies the reference's
        # default (0 = adaptive sampling), see host/engine.hpp simulate() -- round 5: also for a template with topology-coupled
        # plugins (the windowed every-node-scored form is the caller's choice: --percentage-of-nodes-to-score 100)
        assert lib.ccsim_set_profile(h, C.byref(capi.marshal_profile(dataclasses.replace(M.Profile.default(), percentage_of_nodes_to_score=0)))) == 0
        assert lib.ccsim_set_pod(h, C.byref(capi.marshal_pod(snap.pod, keep))) == 0
        lib.ccsim_destroy(h)
    finally:
        os.environ.pop("CCSIM_RECORD")
    python_rec = json.load(open(tmp_path / "python.json"))
    for k in ("nodes", "profile", "pod"):
        assert native_rec[k] == python_rec[k], k
    assert native_rec["run"]["max_limit"] == 1000 and native_rec["run"]["log_cap"] == 1000 and native_rec["run"]["per_node_cap"] >= len(snap.names)
    coupled = bool(snap.pod.spread) or snap.pod.ipa is not None
    assert native_rec["run"]["mode"] == (0 if coupled or len(snap.names) >= 100 else 1)  # order-dependent pods / a sampled search run the literal loop


def _random_objects(rng):
    """A random cluster + pod spec drawing on every ingest feature (selectors, tolerations, affinities, constraints)."""
    keys = ["disk", "gen", "team", "topology.kubernetes.io/zone", "kubernetes.io/hostname", "rack"]
    # "gen" is compared numerically by Gt / Lt: values strconv.ParseInt takes ("+3", "-1", "007") and values it refuses although Python's int() or
    # std::stoll would not ("1_0", "3x", beyond int64)
    vals = {"disk": ["ssd", "hdd", "nvme"], "gen": ["1", "2", "3", "10", "x", "+3", "-1", "007", "1_0", "3x", "99999999999999999999"], "team": ["a", "b"],
            "rack": ["r1", "r2", "r3", "r4"]}
    effects = ["NoSchedule", "PreferNoSchedule", "NoExecute"]
    qty_cpu = ["250m", "1", "2", "0.5", "1500m", "4", "3.3", "100u"]
    qty_mem = ["512Mi", "1Gi", "2G", "1500M", "3.5Gi", "1e9", "123456789", "64Ki"]

    def rand_labels(i, extra=()):
        lab = {}
        for k in keys:
            if k == "kubernetes.io/hostname":
                if rng.random() < 0.9:
                    lab[k] = f"n{i}"
            elif k == "topology.kubernetes.io/zone":
                if rng.random() < 0.85:
                    lab[k] = f"z{int(rng.integers(0, 4))}"
            elif rng.random() < 0.7:
                lab[k] = str(rng.choice(vals[k]))
        lab.update(extra)
        return lab

    def rand_selector():
        r = rng.random()
        if r < 0.15:
            return None
        if r < 0.3:
            return {}
        sel = {}
        if rng.random() < 0.6:
            sel["matchLabels"] = {"app": str(rng.choice(["web", "api", "db"]))}
        if rng.random() < 0.5:
            op = str(rng.choice(["In", "NotIn", "Exists", "DoesNotExist"]))
            e = {"key": str(rng.choice(["app", "tier"])), "operator": op}
            if op in ("In", "NotIn"):
                e["values"] = [str(x) for x in rng.choice(["web", "api", "db", "frontend"], int(rng.integers(1, 3)), replace=False)]
            sel["matchExpressions"] = [e]
        return sel

    def rand_node_term():
        exprs = []
        for _ in range(int(rng.integers(0, 3))):
            k = str(rng.choice(["disk", "gen", "team", "rack"]))
            op = str(rng.choice(["In", "NotIn", "Exists", "DoesNotExist", "Gt", "Lt"]))
            e = {"key": k, "operator": op}
            if op in ("In", "NotIn"):
                e["values"] = [str(x) for x in rng.choice(vals[k], int(rng.integers(1, 3)), replace=False)]
            elif op in ("Gt", "Lt"):
                e["values"] = [str(rng.choice([str(int(rng.integers(0, 5))), "+2", "1_0", "-5", "9223372036854775808", "2.0"], p=[.7, .06, .06, .06, .06, .06]))]
            exprs.append(e)
        t = {"matchExpressions": exprs} if exprs or rng.random() < 0.5 else {}
        if rng.random() < 0.2:
            t["matchFields"] = [{"key": "metadata.name", "operator": str(rng.choice(["In", "NotIn"])), "values": [f"n{int(rng.integers(0, 12))}"]}]
        return t

    def rand_pod_term():
        t = {"topologyKey": str(rng.choice(["topology.kubernetes.io/zone", "kubernetes.io/hostname", "rack"])), "labelSelector": rand_selector()}
        r = rng.random()
        if r < 0.2:
            t["namespaces"] = [str(x) for x in rng.choice(["default", "other", "third"], int(rng.integers(1, 3)), replace=False)]
        elif r < 0.3:
            t["namespaceSelector"] = {}
        elif r < 0.45:
            t["namespaceSelector"] = {"matchLabels": {"team": str(rng.choice(["a", "b"]))}}
            if rng.random() < 0.5:
                t["namespaces"] = ["third"]
        return t

    def rand_pod_affinity():
        aff = {}
        for kind in ("podAffinity", "podAntiAffinity"):
            a = {}
            if rng.random() < 0.35:
                a["requiredDuringSchedulingIgnoredDuringExecution"] = [rand_pod_term() for _ in range(int(rng.integers(1, 3)))]
            if rng.random() < 0.35:
                a["preferredDuringSchedulingIgnoredDuringExecution"] = [{"weight": int(rng.integers(1, 100)), "podAffinityTerm": rand_pod_term()}
                                                                        for _ in range(int(rng.integers(1, 3)))]
            if a:
                aff[kind] = a
        return aff

    n = int(rng.integers(1, 14))
    nodes = []
    for i in range(n):
        taints = [{"key": str(rng.choice(["dedicated", "maintenance", "gpu"])), "value": str(rng.choice(["infra", "soon", ""])), "effect": str(rng.choice(effects))}
                  for _ in range(int(rng.integers(0, 3)))]
        nd = node(f"n{i}", cpu=str(rng.choice(qty_cpu)), mem=str(rng.choice(qty_mem)), pods=str(int(rng.integers(0, 20))), labels=rand_labels(i),
                  taints=taints, unschedulable=bool(rng.random() < 0.1))
        if rng.random() < 0.5:
            nd["status"]["allocatable"]["example.com/gpu"] = str(int(rng.integers(0, 5)))
        nodes.append(nd)
    pods = []
    for j in range(int(rng.integers(0, 25))):
        p = running_pod(f"p{j}", f"n{int(rng.integers(0, n + 2))}", cpu=rng.choice([None, "100m", "1", "2500u"]), mem=rng.choice([None, "128Mi", "1G"]),
                        labels={"app": str(rng.choice(["web", "api", "db"])), "tier": str(rng.choice(["frontend", "backend"]))},
                        ns=str(rng.choice(["default", "other"])), phase=str(rng.choice(["Running", "Pending", "Succeeded", "Failed", "Running"])),
                        affinity=rand_pod_affinity())
        if rng.random() < 0.2:
            p["spec"]["initContainers"] = [{"name": "i", "resources": {"requests": {"cpu": str(rng.choice(qty_cpu)), "example.com/gpu": "1"}}}]
        if rng.random() < 0.15:  # sidecars between ordinary init containers (KEP-753)
            p["spec"]["initContainers"] = [dict({"name": f"i{k}", "resources": {"requests": {"cpu": str(rng.choice(qty_cpu))} if rng.random() < 0.7 else {}}},
                                                **({"restartPolicy": "Always"} if rng.random() < 0.5 else {})) for k in range(int(rng.integers(1, 5)))]
        if rng.random() < 0.1:
            p["spec"]["resources"] = {"requests": {"cpu": str(rng.choice(qty_cpu)), "memory": str(rng.choice(qty_mem))}}
        if rng.random() < 0.15:
            p["spec"]["overhead"] = {"cpu": "1m", "memory": "1Ki"}
        if rng.random() < 0.1:
            p["metadata"]["deletionTimestamp"] = "2025-01-01T00:00:00Z"
        pods.append(p)
    pod = yaml.safe_load(EXAMPLES_POD)
    pod["metadata"]["labels"] = {"app": str(rng.choice(["web", "api"])), "tier": "frontend"}
    if rng.random() < 0.3:
        pod["metadata"]["namespace"] = "other"
    spec = pod["spec"]
    if rng.random() < 0.4:
        spec["containers"].append({"name": "x", "resources": {"requests": {"example.com/gpu": "1", "cpu": str(rng.choice(qty_cpu))}}})
    if rng.random() < 0.3:  # resource names the scheduler drops (not scalar by schedutil.IsScalarResourceName) next to ones it keeps
        weird = ["foo", "storage", "requests.example.com/x", "example.com/Bad Name", "a/b/c", "Example.com/x", "kubernetes.io/batch", "attachable-volumes-csi-x",
                 "example.com/" + "n" * 63, "example.com/" + "n" * 64, "x.kubernetes.io/y", "hugepages-1Gi"]
        spec["containers"][0].setdefault("resources", {}).setdefault("requests", {}).update(
            {str(k): "1" for k in rng.choice(weird, int(rng.integers(1, 4)), replace=False)})
    if rng.random() < 0.3:
        spec["containers"].append({"name": "besteffort"})
    if rng.random() < 0.3:
        spec["initContainers"] = [{"name": "i", "resources": {"requests": {"memory": str(rng.choice(qty_mem))}}}]
    if rng.random() < 0.2:
        spec["initContainers"] = [dict({"name": f"i{k}", "resources": {"requests": {"cpu": str(rng.choice(qty_cpu)), "memory": str(rng.choice(qty_mem))}}},
                                       **({"restartPolicy": "Always"} if rng.random() < 0.5 else {})) for k in range(int(rng.integers(1, 4)))]
    if rng.random() < 0.1:
        spec["resources"] = {"requests": {"cpu": str(rng.choice(qty_cpu))}}
    if rng.random() < 0.5:
        spec["nodeSelector"] = {str(k): str(rng.choice(vals[k])) for k in rng.choice(["disk", "team", "rack"], int(rng.integers(1, 3)), replace=False)}
    spec["tolerations"] = [{k: v for k, v in (("key", str(rng.choice(["dedicated", "maintenance", "", "node.kubernetes.io/unschedulable"]))),
                                              ("operator", str(rng.choice(["Equal", "Exists", ""]))), ("value", str(rng.choice(["infra", "soon", ""]))),
                                              ("effect", str(rng.choice(effects + [""])))) if v}
                           for _ in range(int(rng.integers(0, 3)))]
    aff = rand_pod_affinity()
    na = {}
    if rng.random() < 0.5:
        na["requiredDuringSchedulingIgnoredDuringExecution"] = {"nodeSelectorTerms": [rand_node_term() for _ in range(int(rng.integers(0, 3)))]}
    if rng.random() < 0.5:
        na["preferredDuringSchedulingIgnoredDuringExecution"] = [{"weight": int(rng.integers(1, 100)), "preference": rand_node_term()}
                                                                 for _ in range(int(rng.integers(1, 3)))]
    if na:
        aff["nodeAffinity"] = na
    if aff:
        spec["affinity"] = aff
    if rng.random() < 0.5:
        spec["topologySpreadConstraints"] = [
            {k: v for k, v in (("maxSkew", int(rng.integers(1, 4))), ("minDomains", int(rng.integers(1, 4)) if rng.random() < 0.3 else None),
                               ("topologyKey", str(rng.choice(["topology.kubernetes.io/zone", "kubernetes.io/hostname", "rack"]))),
                               ("whenUnsatisfiable", str(rng.choice(["DoNotSchedule", "ScheduleAnyway"])) if rng.random() < 0.8 else None),
                               ("nodeAffinityPolicy", str(rng.choice(["Honor", "Ignore"])) if rng.random() < 0.4 else None),
                               ("nodeTaintsPolicy", str(rng.choice(["Honor", "Ignore"])) if rng.random() < 0.4 else None),
                               ("matchLabelKeys", [str(x) for x in rng.choice(["app", "tier", "absent"], int(rng.integers(1, 3)), replace=False)]
                                if rng.random() < 0.3 else None),
                               ("labelSelector", rand_selector())) if v is not None or k == "labelSelector"}
            for _ in range(int(rng.integers(1, 3)))]
    exclude = [f"n{int(rng.integers(0, n))}"] if rng.random() < 0.3 else []
    # Namespace objects travel with the pods (simulator.go:177-185); "third" has none -> no labels
    pods += [{"kind": "Namespace", "metadata": {"name": "default", "labels": {"team": "a", "kubernetes.io/metadata.name": "default"}}},
             {"kind": "Namespace", "metadata": {"name": "other", "labels": {"team": str(rng.choice(["a", "b"]))}}}]
    # host ports (NodePorts) and node images (ImageLocality)
    def rand_ports():
        return [{k: v for k, v in (("containerPort", 80), ("hostPort", int(rng.choice([0, 8080, 8080, 9090]))),
                                   ("protocol", str(rng.choice(["TCP", "UDP", ""]))), ("hostIP", str(rng.choice(["", "0.0.0.0", "10.0.0.1", "10.0.0.2"])))) if v != ""}
                for _ in range(int(rng.integers(1, 3)))]
    for p in pods:
        if p.get("kind") == "Pod" and rng.random() < 0.3:
            p["spec"]["containers"][0]["ports"] = rand_ports()
    if rng.random() < 0.5:
        spec["containers"][int(rng.integers(0, len(spec["containers"])))]["ports"] = rand_ports()
    if rng.random() < 0.2 and spec.get("initContainers"):
        spec["initContainers"][0]["ports"] = rand_ports()
    images = ["gcr.io/google-samples/gb-frontend:v4", "registry.k8s.io/pause", "registry.k8s.io/pause:latest", "localhost:5000/app", "busybox"]
    for c in spec["containers"][1:] + (spec.get("initContainers") or []):
        if rng.random() < 0.7:
            c["image"] = str(rng.choice(images))
    for nd in nodes:
        if rng.random() < 0.6:
            nd["status"]["images"] = [{"names": [str(x) for x in rng.choice(images + ["busybox:latest", "localhost:5000/app:latest"], int(rng.integers(1, 3)), replace=False)],
                                       "sizeBytes": int(rng.choice([5_000_000, 120_000_000, 900_000_000, 2_500_000_000]))}
                                      for _ in range(int(rng.integers(1, 4)))]
    if rng.random() < 0.05:
        spec["volumes"] = [{"name": "data", "persistentVolumeClaim": {"claimName": "pvc"}}]
    # priorities (DefaultPreemption's dry run: existing pods below the template's priority are victims)
    for p in pods:
        if p.get("kind") == "Pod" and rng.random() < 0.4:
            p["spec"]["priority"] = int(rng.choice([-10, 0, 5, 1000]))
    if rng.random() < 0.5:
        spec["priority"] = int(rng.choice([0, 5, 100]))
    if rng.random() < 0.1:
        spec["preemptionPolicy"] = str(rng.choice(["Never", "PreemptLowerPriority"]))
    return nodes, pods, pod, exclude


@pytest.mark.parametrize("seed", range(40))
def test_native_ingest_random_differential(native, tmp_path, seed):
    """Random clusters and pod specs through both ingests: same integer snapshot, or the same refusal."""
    rng = np.random.default_rng(9000 + seed)
    nodes, pods, pod, exclude = _random_objects(rng)
    podspec, snaps = _write(tmp_path, "json" if seed % 2 else "yaml", nodes, pods, pod)
    args = ["--podspec", podspec] + [x for s in snaps for x in ("--snapshot", s)] + ["--dump-snapshot", "-"] + (["--exclude-nodes", ",".join(exclude)] if exclude else [])
    # every third case through the host's worker threads (parallel parse of the List's items, parallel per-pod / per-node walks)
    env = dict(os.environ, CCHOST_PARALLEL_MIN_BYTES="0", CCHOST_PARALLEL_MIN_ITEMS="0", CCHOST_THREADS=str(2 + seed % 4)) if seed % 3 == 0 else None
    p = subprocess.run([native] + args, capture_output=True, text=True, timeout=SUBPROC_TIMEOUT, env=env)
    try:
        no, po, ns = cli.load_all(snaps)
        ref = py_dump(ingest.build_snapshot(no, po, cli.parse_pod_spec(podspec), exclude, namespace_objs=ns))
    except NotImplementedError as e:  # both hosts refuse the same inputs (e.g. more topology keys than the engine holds)
        assert p.returncode == 1 and str(e).split(" ")[-1] in p.stderr
        return
    assert p.returncode == 0, p.stderr
    got = json.loads(p.stdout)
    got.pop("label_keys")
    for k in ref:
        assert got[k] == ref[k], (k, seed)


def test_native_yaml_reader_fuzz_against_pyyaml(native, tmp_path):
    """Random block-style documents (what kubectl and PyYAML emit): tricky plain / quoted / folded scalars, escapes,
    nested sequences, odd keys.  The native reader must build the same tree as PyYAML."""
    import random
    import string
    rnd = random.Random(20250923)
    alphabet = string.ascii_letters + string.digits + "  :#-_/.,'\"{}[]!&*?|>%@`=+~\\"

    def rstr():
        s = "".join(rnd.choice(alphabet) for _ in range(rnd.choice([0, 1, 3, 8, 20, 60, 150])))
        if rnd.random() < 0.1:
            s += "\n" + "".join(rnd.choice(alphabet) for _ in range(10))
        if rnd.random() < 0.1:
            s = "é✓ " + s
        return s

    def rkey():
        return rnd.choice(["app", "kubernetes.io/name", "a b", "x:y", "123", "true", "é", 'q"uote', "it's"]) + str(rnd.randint(0, 99))

    def rval(d=0):
        r = rnd.random()
        if d < 3 and r < 0.25:
            return {rkey(): rval(d + 1) for _ in range(rnd.randint(0, 4))}
        if d < 3 and r < 0.45:
            return [rval(d + 1) for _ in range(rnd.randint(0, 4))]
        if r < 0.55:
            return rnd.randint(-5, 10**6)
        if r < 0.6:
            return rnd.choice([True, False, None])
        if r < 0.65:
            return rnd.choice(["true", "null", "123", "1e3", "0.5", "~", "yes", "- x", "a: b", "#c", " lead", "trail "])
        return rstr()

    for it in range(150):
        docs = [{rkey(): rval() for _ in range(rnd.randint(1, 6))} for _ in range(rnd.choice([1, 1, 3]))]
        text = yaml.safe_dump_all(docs, default_flow_style=False, width=rnd.choice([30, 80, 1000]), allow_unicode=rnd.choice([True, False]))
        (tmp_path / "d.yaml").write_text(text)
        got = json.loads(_run(native, ["--parse", str(tmp_path / "d.yaml")]))
        assert got == [d for d in yaml.safe_load_all(text) if d], (it, text)


def test_namespace_selector_known_answer(native, tmp_path):
    """AffinityTerm.Matches (S/framework/types.go:927-935): namespaces UNION namespaceSelector, labels from the Namespace
    objects of the snapshot; a term without either means the owner's namespace."""
    nodes = [node("a"), node("b")]
    ns = [{"kind": "Namespace", "metadata": {"name": n, "labels": {"team": t}}} for n, t in (("default", "x"), ("red", "r"), ("blue", "b"))]
    pods = [running_pod("p-default", "a", labels={"app": "web"}, ns="default"), running_pod("p-red", "a", labels={"app": "web"}, ns="red"),
            running_pod("p-blue", "b", labels={"app": "web"}, ns="blue"), running_pod("p-nolabel-ns", "b", labels={"app": "web"}, ns="ghost")]
    sel = {"matchLabels": {"app": "web"}}
    terms = [{"topologyKey": "kubernetes.io/hostname", "labelSelector": sel},                                                # own namespace only
             {"topologyKey": "kubernetes.io/hostname", "labelSelector": sel, "namespaceSelector": {"matchLabels": {"team": "r"}}},  # red
             {"topologyKey": "kubernetes.io/hostname", "labelSelector": sel, "namespaceSelector": {}},                           # every namespace
             {"topologyKey": "kubernetes.io/hostname", "labelSelector": sel, "namespaces": ["blue"],
              "namespaceSelector": {"matchLabels": {"team": "r"}}}]                                                             # blue UNION red
    pod = yaml.safe_load(EXAMPLES_POD)
    pod["spec"]["affinity"] = {"podAntiAffinity": {"requiredDuringSchedulingIgnoredDuringExecution": terms}}
    for n in nodes:
        n["metadata"]["labels"]["kubernetes.io/hostname"] = n["metadata"]["name"]
    snap = ingest.build_snapshot(nodes, pods, pod, namespace_objs=ns)
    assert [a.tolist() if a is not None else None for a in snap.pod.ipa.anti_existing] == [[1, 0], [1, 0], [2, 2], [1, 1]]
    (tmp_path / "c.json").write_text(json.dumps({"kind": "List", "items": nodes + pods + ns}))
    (tmp_path / "pod.json").write_text(json.dumps(pod))
    got = json.loads(_run(native, ["--podspec", str(tmp_path / "pod.json"), "--snapshot", str(tmp_path / "c.json"), "--dump-snapshot", "-"]))
    assert got["pod"]["ipa"]["anti_existing"] == [[1, 0], [1, 0], [2, 2], [1, 1]]


def test_spread_node_inclusion_policies_known_answer(native, tmp_path):
    """matchNodeInclusionPolicies (podtopologyspread/common.go:107-122): nodeAffinityPolicy (default Honor) and
    nodeTaintsPolicy (default Ignore) decide which nodes count towards a constraint's domains."""
    z = lambda v, **kw: dict({"topology.kubernetes.io/zone": v}, **kw)
    nodes = [node("n0", labels=z("a", disk="ssd")), node("n1", labels=z("a", disk="hdd")),
             node("n2", labels=z("b", disk="ssd"), taints=[{"key": "dedicated", "value": "infra", "effect": "NoSchedule"}]),
             node("n3", labels=z("b", disk="ssd"), taints=[{"key": "soft", "effect": "PreferNoSchedule"}])]
    pod = yaml.safe_load(EXAMPLES_POD)
    pod["spec"]["nodeSelector"] = {"disk": "ssd"}
    base = {"maxSkew": 1, "topologyKey": "topology.kubernetes.io/zone", "labelSelector": {"matchLabels": {"app": "guestbook"}}}
    pod["spec"]["topologySpreadConstraints"] = [dict(base), dict(base, nodeAffinityPolicy="Ignore"), dict(base, nodeTaintsPolicy="Honor"),
                                                dict(base, nodeAffinityPolicy="Ignore", nodeTaintsPolicy="Honor")]
    want = [[1, 0, 1, 1], None, [1, 0, 0, 1], [1, 1, 0, 1]]  # canonical order n0 n2 n1 n3 -> reorder below
    snap = ingest.build_snapshot(nodes, [], pod)
    order = [snap.names.index(n) for n in ("n0", "n1", "n2", "n3")]
    got_py = [None if k.node_included is None else [int(k.node_included[i]) for i in order] for k in snap.pod.spread]
    assert got_py == want
    (tmp_path / "c.json").write_text(json.dumps({"kind": "List", "items": nodes}))
    (tmp_path / "pod.json").write_text(json.dumps(pod))
    d = json.loads(_run(native, ["--podspec", str(tmp_path / "pod.json"), "--snapshot", str(tmp_path / "c.json"), "--dump-snapshot", "-"]))
    assert [None if k["node_included"] is None else [k["node_included"][i] for i in order] for k in d["pod"]["spread"]] == want


def test_spread_match_label_keys_known_answer(native, tmp_path):
    """matchLabelKeys (podtopologyspread/common.go:95-105): the incoming pod's values of the listed keys join the selector."""
    nodes = [node("n0", labels={"topology.kubernetes.io/zone": "a"}), node("n1", labels={"topology.kubernetes.io/zone": "b"})]
    pods = [running_pod("v1-a", "n0", labels={"app": "web", "rev": "1"}), running_pod("v2-a", "n0", labels={"app": "web", "rev": "2"}),
            running_pod("v2-b", "n1", labels={"app": "web", "rev": "2"}), running_pod("other", "n1", labels={"app": "db", "rev": "2"})]
    pod = yaml.safe_load(EXAMPLES_POD)
    pod["metadata"]["labels"] = {"app": "web", "rev": "2"}
    base = {"maxSkew": 1, "topologyKey": "topology.kubernetes.io/zone", "labelSelector": {"matchLabels": {"app": "web"}}}
    pod["spec"]["topologySpreadConstraints"] = [dict(base), dict(base, matchLabelKeys=["rev"]), dict(base, matchLabelKeys=["absent"]),
                                                {"maxSkew": 1, "topologyKey": "topology.kubernetes.io/zone", "labelSelector": {}, "matchLabelKeys": ["rev"]},
                                                {"maxSkew": 1, "topologyKey": "topology.kubernetes.io/zone", "matchLabelKeys": ["rev"]}]
    # web pods: 2 / 1; web AND rev=2: 1 / 1; key absent on the pod: unchanged; {} AND rev=2: 1 / 2 (db counts); nil selector stays Nothing
    want = [([2, 1], True), ([1, 1], True), ([2, 1], True), ([1, 2], True), (None, False)]
    snap = ingest.build_snapshot(nodes, pods, pod)
    assert [(None if k.node_match_count is None else k.node_match_count.tolist(), bool(k.self_match)) for k in snap.pod.spread] == want
    (tmp_path / "c.json").write_text(json.dumps({"kind": "List", "items": nodes + pods}))
    (tmp_path / "pod.json").write_text(json.dumps(pod))
    d = json.loads(_run(native, ["--podspec", str(tmp_path / "pod.json"), "--snapshot", str(tmp_path / "c.json"), "--dump-snapshot", "-"]))
    assert [(k["node_match_count"], k["self_match"]) for k in d["pod"]["spread"]] == want


def test_sidecar_and_pod_level_requests_known_answer(native, tmp_path):
    """PodRequests (component-helpers/resource/helpers.go:144-251, KEP-753): restartable init containers add to the sum,
    InitContainerUse(i) = init container i + the sidecars before it, pod-level requests override, overhead on top."""
    pod = yaml.safe_load(EXAMPLES_POD)
    spec = pod["spec"]
    spec["containers"] = [{"name": "c", "resources": {"requests": {"cpu": "1"}}}]
    spec["initContainers"] = [{"name": "ic1", "resources": {"requests": {"cpu": "2"}}},
                              {"name": "s1", "restartPolicy": "Always", "resources": {"requests": {"cpu": "500m"}}},
                              {"name": "ic2", "resources": {"requests": {"cpu": "3"}}},
                              {"name": "s2", "restartPolicy": "Always", "resources": {"requests": {"cpu": "250m"}}}]
    spec["overhead"] = {"cpu": "10m"}
    req, nzc, nzm = ingest.pod_requests(spec, ["cpu", "memory"])
    assert req == {"cpu": 3510, "memory": 0} and nzc == 3510 and nzm == 600 * (1 << 20)
    spec["resources"] = {"requests": {"cpu": "5", "example.com/gpu": "9"}}  # pod-level: cpu counts, extended resources do not
    req, nzc, nzm = ingest.pod_requests(spec, ["cpu", "memory", "example.com/gpu"])
    assert req == {"cpu": 5010, "memory": 0, "example.com/gpu": 0} and nzc == 5010
    (tmp_path / "c.json").write_text(json.dumps({"kind": "List", "items": [node("n0")]}))
    (tmp_path / "pod.json").write_text(json.dumps(pod))
    d = json.loads(_run(native, ["--podspec", str(tmp_path / "pod.json"), "--snapshot", str(tmp_path / "c.json"), "--dump-snapshot", "-"]))
    assert d["pod"]["req"][:2] == [5010, 0] and d["pod"]["nz_mcpu"] == 5010 and d["pod"]["nz_mem"] == 600 * (1 << 20)


# ---- genpod (cmd/genpod, pkg/client/nspod.go:34-126) --------------------------------------------------------------------
def _genpod_objects(rng=None):
    ns = [{"kind": "Namespace", "metadata": {"name": "limited", "annotations": {"openshift.io/node-selector": "region=primary, disk = ssd"}}},
          {"kind": "Namespace", "metadata": {"name": "open"}},
          {"kind": "Namespace", "metadata": {"name": "zero", "annotations": {"openshift.io/node-selector": ""}}},
          {"kind": "Namespace", "metadata": {"name": "broken", "annotations": {"openshift.io/node-selector": "a=b=c"}}}]
    lrs = [{"kind": "LimitRange", "metadata": {"name": "a", "namespace": "limited"}, "spec": {"limits": [
                {"type": "Pod", "max": {"cpu": "2", "memory": "1Gi"}}, {"type": "Container", "max": {"cpu": "100m"}},
                {"type": "Pod", "max": {"cpu": "1500m", "nvdia.com/gpu": "1"}}]}},
           {"kind": "LimitRange", "metadata": {"name": "b", "namespace": "limited"}, "spec": {"limits": [{"type": "Pod", "max": {"memory": "900Mi", "cpu": "1.5"}}]}},
           {"kind": "LimitRange", "metadata": {"name": "z", "namespace": "zero"}, "spec": {"limits": [{"type": "Pod", "max": {"cpu": "0", "memory": "0"}}]}},
           {"kind": "LimitRange", "metadata": {"name": "o", "namespace": "other"}, "spec": {"limits": [{"type": "Pod", "max": {"cpu": "1"}}]}}]
    return ns, lrs


def test_genpod_known_answers():
    """By hand from nspod.go: minimum of the Pod-type maxima per resource (1500m < 2, the later equal 1.5 does not replace it:
    Cmp == 1 is strict; 900Mi < 1Gi), limits == requests, the node selector from the annotation; all-zero limits leave the
    container without resources; a namespace without LimitRanges yields the bare stub."""
    from cluster_capacity_amd import genpod
    ns, lrs = _genpod_objects()
    pod = genpod.namespace_pod("limited", ns, lrs)
    c = pod["spec"]["containers"][0]
    assert c["resources"]["limits"] == c["resources"]["requests"] == {"memory": "900Mi", "cpu": "1500m", "nvdia.com/gpu": "1"}
    assert pod["spec"]["nodeSelector"] == {"region": "primary", "disk": "ssd"}
    assert (pod["metadata"]["name"], pod["metadata"]["namespace"], c["image"], c["imagePullPolicy"]) == \
        ("cluster-capacity-stub-container", "limited", "gcr.io/google_containers/pause:2.0", "Always")
    assert pod["spec"]["restartPolicy"] == "OnFailure" and pod["spec"]["dnsPolicy"] == "Default"
    assert "resources" not in genpod.namespace_pod("zero", ns, lrs)["spec"]["containers"][0]
    assert genpod.namespace_pod("zero", ns, lrs)["spec"]["nodeSelector"] == {}
    bare = genpod.namespace_pod("open", ns, lrs)
    assert "resources" not in bare["spec"]["containers"][0] and "nodeSelector" not in bare["spec"]
    with pytest.raises(genpod.GenpodError, match="Namespace missing not found"):
        genpod.namespace_pod("missing", ns, lrs)
    with pytest.raises(genpod.GenpodError, match="Unable to parse openshift.io/node-selector"):
        genpod.namespace_pod("broken", ns, lrs)


@pytest.mark.parametrize("fmt", ["json", "yaml"])
def test_native_genpod_equals_python_genpod(native, tmp_path, fmt):
    from cluster_capacity_amd import genpod
    ns, lrs = _genpod_objects()
    (tmp_path / "objs.yaml").write_text(yaml.safe_dump({"kind": "List", "items": ns + lrs}))
    load = json.loads if fmt == "json" else yaml.safe_load
    for name in ("limited", "open", "zero"):
        got = load(_run(native, ["--genpod", name, "--snapshot", str(tmp_path / "objs.yaml"), "-o", fmt]))
        assert got == genpod.namespace_pod(name, ns, lrs), name
        buf = __import__("io").StringIO()
        assert cli.main(["--genpod", name, "--snapshot", str(tmp_path / "objs.yaml"), "-o", fmt], out=buf) == 0
        assert load(buf.getvalue()) == got
    for name, msg in (("missing", "Namespace missing not found"), ("broken", "Unable to parse openshift.io/node-selector")):
        p = subprocess.run([native, "--genpod", name, "--snapshot", str(tmp_path / "objs.yaml")], capture_output=True, text=True)
        assert p.returncode == 1 and msg in p.stderr
    # the generated pod is a valid --podspec for the simulator's ingest (both hosts): README-style nodes labelled for the selector
    nodes = [node(f"n{i}", labels={"region": "primary", "disk": "ssd" if i % 2 else "hdd"}) for i in range(4)]
    nodes[0]["status"]["allocatable"]["nvdia.com/gpu"] = "2"
    (tmp_path / "nodes.json").write_text(json.dumps({"kind": "List", "items": nodes}))
    (tmp_path / "pod.yaml").write_text(_run(native, ["--genpod", "limited", "--snapshot", str(tmp_path / "objs.yaml")]))
    d = json.loads(_run(native, ["--podspec", str(tmp_path / "pod.yaml"), "--snapshot", str(tmp_path / "nodes.json"), "--dump-snapshot", "-"]))
    assert d["pod"]["req"][:2] == [1500, 900 * 1024 * 1024] and d["scalar_names"] == ["nvdia.com/gpu"] and d["pod"]["has_node_selector"]


# ---- several templates (--podspec repeated): scheduled pod i is a clone of template i mod P (report.go:146-171) -------
def _templates_case(n_nodes=12):
    """Config-5-shaped templates (zone DoNotSchedule spread + required hostname anti-affinity to their own label, one with a
    node selector) over a small zoned cluster with one existing pod per template label."""
    nodes = [node(f"m{i:02d}", cpu="4", mem="8Gi", pods="6", labels={"topology.kubernetes.io/zone": f"z{i % 3}", "kubernetes.io/hostname": f"m{i:02d}",
                                                                     "disk": "ssd" if i % 2 else "hdd"}) for i in range(n_nodes)]
    pods = [running_pod("old-a", "m01", cpu="500m", mem="256Mi", labels={"app": "t0"}), running_pod("old-b", "m05", cpu="1", labels={"app": "t1"})]
    templates = []
    for t, (cpu, mem) in enumerate([("500m", "512Mi"), ("1", "1Gi"), ("250m", "256Mi")]):
        p = yaml.safe_load(EXAMPLES_POD)
        p["metadata"]["name"], p["metadata"]["labels"] = f"tmpl-{t}", {"app": f"t{t}"}
        p["spec"]["containers"][0]["resources"] = {"requests": {"cpu": cpu, "memory": mem}}
        sel = {"matchLabels": {"app": f"t{t}"}}
        p["spec"]["topologySpreadConstraints"] = [{"maxSkew": 1 + t, "topologyKey": "topology.kubernetes.io/zone", "whenUnsatisfiable": "DoNotSchedule", "labelSelector": sel}]
        p["spec"]["affinity"] = {"podAntiAffinity": {"requiredDuringSchedulingIgnoredDuringExecution": [{"topologyKey": "kubernetes.io/hostname", "labelSelector": sel}]}}
        if t == 1:
            p["spec"]["nodeSelector"] = {"disk": "ssd"}
        templates.append(p)
    return nodes, pods, templates


def _write_templates(tmp_path, nodes, pods, templates):
    (tmp_path / "cluster.json").write_text(json.dumps({"kind": "List", "items": nodes + pods}))
    paths = []
    for t, p in enumerate(templates):
        (tmp_path / f"t{t}.yaml").write_text(yaml.safe_dump(json.loads(json.dumps(p))))  # (a deep copy: no YAML anchors for shared sub-dicts)
        paths.append(str(tmp_path / f"t{t}.yaml"))
    return str(tmp_path / "cluster.json"), paths


def test_several_templates_ingest_abi_and_report_agree(native, recorder, tmp_path):
    import ctypes as C
    from cluster_capacity_amd import capi
    nodes, pods, templates = _templates_case()
    cluster, paths = _write_templates(tmp_path, nodes, pods, templates)
    flags = [x for p in paths for x in ("--podspec", p)] + ["--snapshot", cluster]
    # (1) the integer snapshot: shared node columns, one pod side per template
    got = json.loads(_run(native, flags + ["--dump-snapshot", "-"]))
    pypods = [cli.parse_pod_spec(p) for p in paths]
    snap = ingest.build_snapshot(*cli.load_objects([cluster]), pypods)
    ref = py_dump(snap)
    got.pop("label_keys")
    assert got.keys() == ref.keys() and len(got["more_pods"]) == 2
    for k in ref:
        assert got[k] == ref[k], k
    assert got["pod"]["spread"][0]["col"] == got["more_pods"][0]["spread"][0]["col"]  # one zone column for all templates
    assert [p["has_node_selector"] for p in [got["pod"]] + got["more_pods"]] == [False, True, False]
    # (2) what reaches ccsim_set_pods is identical from both hosts
    env = dict(os.environ, CCSIM_LIB=recorder, CCSIM_RECORD=str(tmp_path / "native.json"))
    p = subprocess.run([native] + flags + ["--max-limit", "12", "-o", "json"], capture_output=True, text=True, env=env, timeout=SUBPROC_TIMEOUT)
    assert p.returncode == 0, p.stderr
    native_rec = json.load(open(tmp_path / "native.json"))
    lib = C.CDLL(recorder)
    os.environ["CCSIM_RECORD"] = str(tmp_path / "python.json")
    try:
        cfg = capi.CConfig()
        cfg.abi_version, cfg.use_graph = capi.ABI_VERSION, 1
        h = C.c_void_p()
        assert lib.ccsim_create(C.byref(cfg), C.byref(h)) == 0
        keep = []
        assert lib.ccsim_load_nodes(h, C.byref(capi.marshal_nodes(snap.nodes, keep))) == 0
        assert lib.ccsim_set_profile(h, C.byref(capi.marshal_profile(M.Profile.default()))) == 0  # several templates: searched completely
        arr = (capi.CPod * 3)(*[capi.marshal_pod(q, keep) for q in snap.pods])
        assert lib.ccsim_set_pods(h, arr, 3) == 0
        lib.ccsim_destroy(h)
    finally:
        os.environ.pop("CCSIM_RECORD")
    python_rec = json.load(open(tmp_path / "python.json"))
    for k in ("nodes", "profile", "pods"):
        assert native_rec[k] == python_rec[k], k
    # (3) the review: the recorder's canned log (node i at position i) split round-robin over the templates
    rev = json.loads(p.stdout)
    assert [q["podName"] for q in rev["status"]["pods"]] == ["tmpl-0", "tmpl-1", "tmpl-2"] and len(rev["spec"]["templates"]) == 3
    for t in range(3):
        assert [r["nodeName"] for r in rev["status"]["pods"][t]["replicasOnNodes"]] == snap.names[t::3]
    res = M.RunResult(placed=12, stop=M.STOP_LIMIT, per_node_count=np.ones(12, np.int32), log=np.arange(12, dtype=np.int32),
                      hist=np.zeros(M.NREASON, np.int64), hist_taintset=np.zeros(1, np.int64), n_code_unschedulable=0)
    pyrev = cli.build_review(pypods, snap, res, 12)
    assert pyrev["status"]["pods"] == rev["status"]["pods"] and pyrev["status"]["failReason"] == rev["status"]["failReason"]
    pretty = subprocess.run([native] + flags + ["--max-limit", "12", "--verbose"], capture_output=True, text=True, env=env, timeout=SUBPROC_TIMEOUT).stdout
    assert pretty == cli.pretty(pyrev, True)
    # (4) templates whose selectors match one another's clones are refused by both hosts
    templates[1]["metadata"]["labels"] = {"app": "t0"}
    cluster, paths = _write_templates(tmp_path, nodes, pods, templates)
    bad = subprocess.run([native] + [x for q in paths for x in ("--podspec", q)] + ["--snapshot", cluster, "--dump-snapshot", "-"], capture_output=True, text=True)
    assert bad.returncode == 1 and "several templates" in bad.stderr
    with pytest.raises(NotImplementedError, match="several templates"):
        ingest.build_snapshot(*cli.load_objects([cluster]), [cli.parse_pod_spec(q) for q in paths])


def test_several_templates_oracle_round_robin(ccref):
    """The ingest's pod sides drive the oracle's round-robin loop (ccref_run_multi): per-template counts from the log match
    what parsePodsReview would report, every template's clones respect ITS spread constraint and anti-affinity."""
    nodes, pods, templates = _templates_case()
    snap = ingest.build_snapshot(nodes, pods, templates)
    r = ccref.run_multi(M.Profile.default(), snap.nodes, snap.pods)
    assert r.stop == M.STOP_UNSCHEDULABLE and r.placed == int(r.per_spec_count.sum()) > 6
    zone = snap.nodes.label_cols[snap.pods[0].spread[0].col]
    for t in range(3):
        mine = r.log[t::3]
        assert len(mine) == r.per_spec_count[t] and len(set(mine.tolist())) == len(mine)  # hostname anti-affinity: one clone per node
        if t == 1:
            assert all(nodes_i % 2 == 1 for nodes_i in [int(snap.names[i][1:]) for i in mine])  # disk=ssd nodes only
    rev = cli.build_review(templates, snap, M.RunResult(placed=r.placed, stop=r.stop, per_node_count=r.per_node_count, log=r.log, hist=r.hist,
                                                        hist_taintset=r.hist_taintset, n_code_unschedulable=r.n_code_unschedulable,
                                                        stop_spec=r.stop_spec), 0)
    assert [sum(x["replicas"] for x in q["replicasOnNodes"]) for q in rev["status"]["pods"]] == r.per_spec_count.tolist()
    assert rev["status"]["replicas"] == r.placed and rev["status"]["failReason"]["failType"] == "Unschedulable"


@pytest.mark.gpu
def test_several_templates_cli_end_to_end_both_hosts(ccref, native, tmp_path):
    """--podspec a --podspec b --podspec c through the C++ host, the C ABI (ccsim_set_pods) and the HIP engine == the Python host
    == the oracle's round-robin loop."""
    import io
    nodes, pods, templates = _templates_case(n_nodes=30)
    cluster, paths = _write_templates(tmp_path, nodes, pods, templates)
    flags = [x for p in paths for x in ("--podspec", p)] + ["--snapshot", cluster]
    got = json.loads(_run(native, flags + ["-o", "json"]))
    buf = io.StringIO()
    assert cli.main(flags + ["-o", "json"], out=buf) == 0
    ref = json.loads(buf.getvalue())
    got["status"].pop("creationTimestamp"), ref["status"].pop("creationTimestamp")
    assert got["status"] == ref["status"]
    snap = ingest.build_snapshot(nodes, pods, [cli.parse_pod_spec(p) for p in paths])
    r = ccref.run_multi(M.Profile.default(), snap.nodes, snap.pods)
    assert got["status"]["replicas"] == r.placed
    assert [sum(x["replicas"] for x in q["replicasOnNodes"]) for q in got["status"]["pods"]] == r.per_spec_count.tolist()
    for t in range(3):  # first-placement order per template == the oracle's sequence
        assert [x["nodeName"] for x in got["status"]["pods"][t]["replicasOnNodes"]] == [snap.names[i] for i in r.log[t::3]]
    txt = _run(native, flags + ["--verbose", "--max-limit", "7"])
    assert "Termination reason: LimitReached: Maximum number of pods simulated: 7" in txt and txt.count("The cluster can schedule") == 3


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["soft", "scalar"])
def test_refused_template_sets_take_one_cycle_at_a_time_in_both_hosts(ccref, native, tmp_path, kind):
    """A template set outside the window engine's shape (ccsim_set_pods answers -ENOSYS): both hosts then drive the loop of
    simulator.go:186-256 one cycle at a time over the single-template path, the earlier clones of every template folded into the
    per-node counts of the one in turn. Same review as the oracle's round-robin loop."""
    import io
    nodes, pods, templates = _templates_case(n_nodes=30)
    if kind == "soft":
        templates[2]["spec"]["topologySpreadConstraints"].append({"maxSkew": 1, "topologyKey": "kubernetes.io/hostname", "whenUnsatisfiable": "ScheduleAnyway",
                                                                  "labelSelector": {"matchLabels": {"app": "t2"}}})
    else:
        for nd in nodes[::2]:
            nd["status"]["allocatable"]["example.com/widget"] = "3"
        templates[0]["spec"]["containers"][0]["resources"]["requests"]["example.com/widget"] = "1"
    cluster, paths = _write_templates(tmp_path, nodes, pods, templates)
    flags = [x for p in paths for x in ("--podspec", p)] + ["--snapshot", cluster]
    for extra in ([], ["--max-limit", "11"]):
        p = subprocess.run([native] + flags + extra + ["-o", "json"], capture_output=True, text=True, timeout=SUBPROC_TIMEOUT)
        assert p.returncode == 0, p.stderr
        assert "one scheduling cycle at a time" in p.stderr
        got = json.loads(p.stdout)
        buf = io.StringIO()
        assert cli.main(flags + extra + ["-o", "json"], out=buf) == 0
        ref = json.loads(buf.getvalue())
        got["status"].pop("creationTimestamp"), ref["status"].pop("creationTimestamp")
        assert got["status"] == ref["status"]
        snap = ingest.build_snapshot(nodes, pods, [cli.parse_pod_spec(q) for q in paths])
        r = ccref.run_multi(M.Profile.default(), snap.nodes, snap.pods, max_limit=11 if extra else 0)
        assert got["status"]["replicas"] == r.placed
        assert [sum(x["replicas"] for x in q["replicasOnNodes"]) for q in got["status"]["pods"]] == r.per_spec_count.tolist()
        for t in range(3):
            assert [x["nodeName"] for x in got["status"]["pods"][t]["replicasOnNodes"]] == list(dict.fromkeys(snap.names[i] for i in r.log[t::3]))


@pytest.mark.gpu
def test_native_sharded_templates_with_one_rank_equal_the_window_engine(ccref, native, tmp_path):
    """--force-sharded with several templates: one rank, one thread, the one-cycle-at-a-time loop over ccsim_dist_run == the window engine
    of the plain run == the oracle's round-robin loop."""
    nodes, pods, templates = _templates_case(n_nodes=30)
    cluster, paths = _write_templates(tmp_path, nodes, pods, templates)
    flags = [x for p in paths for x in ("--podspec", p)] + ["--snapshot", cluster, "-o", "json"]
    for extra in ([], ["--max-limit", "10"]):
        plain = json.loads(_run(native, flags + extra))
        p = subprocess.run([native] + flags + extra + ["--force-sharded"], capture_output=True, text=True, timeout=SUBPROC_TIMEOUT)
        assert p.returncode == 0 and "one scheduling cycle at a time" in p.stderr, p.stderr
        got = json.loads(p.stdout)
        got["status"].pop("creationTimestamp"), plain["status"].pop("creationTimestamp")
        assert got["status"] == plain["status"]
    snap = ingest.build_snapshot(nodes, pods, [cli.parse_pod_spec(q) for q in paths])
    r = ccref.run_multi(M.Profile.default(), snap.nodes, snap.pods)
    assert json.loads(_run(native, flags + ["--force-sharded"]))["status"]["replicas"] == r.placed


# ---- --gpus N: the snapshot sharded over the GPUs of one box, the run driven inside libccsim.so over RCCL ------------------
def test_native_sharded_run_fails_loudly_without_gpus(native, tmp_path):
    nodes, pods, pod, _ = CASES["readme"]()
    podspec, snaps = _write(tmp_path, "json", nodes, pods, pod)
    env = dict(os.environ, HIP_VISIBLE_DEVICES="-1")
    p = subprocess.run([native, "--podspec", podspec, "--snapshot", snaps[0], "--gpus", "2"], capture_output=True, text=True, env=env, timeout=SUBPROC_TIMEOUT)
    assert p.returncode == 1 and ("ccsim_create failed on device" in p.stderr or "ccsim_dist_unique_id failed" in p.stderr)


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["readme", "taints-selectors", "ports-images", "disks"])
def test_native_sharded_path_with_one_rank_equals_the_plain_run(native, tmp_path, case):
    """--force-sharded: node-range shard [0, N), one thread, one RCCL rank: shard views of the marshalled arrays, ccsim_dist_comm_init /
    sync_tables / dist_run and the merge of the per-rank reports must give the review of the plain single-GPU run."""
    nodes, pods, pod, exclude = CASES[case]()
    podspec, snaps = _write(tmp_path, "json", nodes, pods, pod)
    args = ["--podspec", podspec, "--snapshot", snaps[0], "-o", "json"] + (["--exclude-nodes", ",".join(exclude)] if exclude else [])
    for extra in ([], ["--max-limit", "9"]):
        plain = json.loads(_run(native, args + extra))
        shard = json.loads(_run(native, args + extra + ["--force-sharded"]))
        plain["status"].pop("creationTimestamp"), shard["status"].pop("creationTimestamp")
        assert shard["status"] == plain["status"], (case, extra)


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [[], ["--max-limit", "40"], ["--percentage-of-nodes-to-score", "30"], ["--percentage-of-nodes-to-score", "100"]])
def test_native_sharded_coupled_template_under_the_default_percentage_equals_the_plain_run(native, tmp_path, extra):
    """Round 6: a template with a DoNotSchedule zone constraint + required hostname anti-affinity on a cluster large enough to sample
    (>= 100 nodes), through the sharded path (one RCCL rank): the reference's default adaptive sampling -- or the percentage said -- as
    on the plain path, same review (total, per-node counts in first-placement order, stop reason and message)."""
    nodes = [node(f"n{i:03d}", cpu=str(2 + i % 5), mem=f"{4 + i % 7}Gi", pods="6", labels={"topology.kubernetes.io/zone": f"z{i % 4}", "kubernetes.io/hostname": f"n{i:03d}"})
             for i in range(160)]
    pod = yaml.safe_load(EXAMPLES_POD)
    pod["metadata"]["labels"] = {"app": "sim"}
    pod["spec"]["topologySpreadConstraints"] = [{"maxSkew": 1, "topologyKey": "topology.kubernetes.io/zone", "whenUnsatisfiable": "DoNotSchedule",
                                                 "labelSelector": {"matchLabels": {"app": "sim"}}}]
    pod["spec"]["affinity"] = {"podAntiAffinity": {"requiredDuringSchedulingIgnoredDuringExecution": [
        {"topologyKey": "kubernetes.io/hostname", "labelSelector": {"matchLabels": {"app": "sim"}}}]}}
    podspec, snaps = _write(tmp_path, "json", nodes, [], pod)
    args = ["--podspec", podspec, "--snapshot", snaps[0], "-o", "json"]
    plain = json.loads(_run(native, args + extra))
    shard = json.loads(_run(native, args + extra + ["--force-sharded"]))
    plain["status"].pop("creationTimestamp"), shard["status"].pop("creationTimestamp")
    assert plain["status"]["replicas"] > 0 and shard["status"] == plain["status"], extra


def _slice_nodes(rec, lo, hi):
    """What rank [lo, hi) of a node-range sharding must pass to ccsim_load_nodes, from the unsharded record."""
    out = {}
    for k, v in rec.items():
        if k in ("alloc", "req", "label_cols"):
            out[k] = [{"v": None if c["v"] is None else c["v"][lo:hi]} for c in v]
        elif isinstance(v, list):
            out[k] = v[lo:hi]
        else:
            out[k] = v
    out["n_nodes"], out["global_offset"], out["n_global"] = hi - lo, lo, rec["n_nodes"]
    return out


def _slice_pod(rec, lo, hi):
    """... and to ccsim_set_pod: per-node side arrays follow the nodes, everything else is replicated; the cluster-wide
    entries_existing count is contributed by rank 0 only (ccsim_dist_sync_tables sums it)."""
    out = json.loads(json.dumps(rec))
    sl = lambda v: None if v is None else v[lo:hi]
    for c in out["spread"]:
        c["node_match_count"], c["node_included"] = sl(c["node_match_count"]), sl(c["node_included"])
    if out["has_ipa"]:
        a = out["ipa"]
        a["aff_existing"] = sl(a["aff_existing"])
        for k in ("anti_existing", "exist_anti", "score_existing"):
            a[k] = [{"v": sl(c["v"])} for c in a[k]]
        if lo > 0:
            a["entries_existing"] = 0
    out["host_ports_conflict"], out["image_score"], out["volume_veto"] = sl(out["host_ports_conflict"]), sl(out["image_score"]), sl(out["volume_veto"])
    return out


def _sharded_records(native, recorder, tmp_path, objs, n_gpus):
    """Run the native host against the recorder, unsharded and with --gpus n_gpus; check each rank's record against the slices of
    the unsharded one.  Returns (stdout of the sharded run, n, podspec, snaps), or None when the host refuses the input."""
    nodes, pods, pod, exclude = objs
    podspec, snaps = _write(tmp_path, "json", nodes, pods, pod)
    args = ["--podspec", podspec] + [x for s in snaps for x in ("--snapshot", s)] + ["-o", "json"] + (["--exclude-nodes", ",".join(exclude)] if exclude else [])
    env = dict(os.environ, CCSIM_LIB=recorder, CCSIM_RECORD=str(tmp_path / "plain.json"))
    p = subprocess.run([native] + args, capture_output=True, text=True, env=env, timeout=SUBPROC_TIMEOUT)
    if p.returncode != 0:
        return None
    plain = json.load(open(tmp_path / "plain.json"))
    env = dict(env, CCSIM_RECORD=str(tmp_path / "shard.json"), CCSIM_RECORD_PER_DEVICE="1")
    p = subprocess.run([native] + args + ["--gpus", str(n_gpus)], capture_output=True, text=True, env=env, timeout=SUBPROC_TIMEOUT)
    assert p.returncode == 0, p.stderr
    n = plain["nodes"]["n_nodes"]
    per = -(-n // n_gpus)
    for g in range(n_gpus):
        rec = json.load(open(f"{tmp_path}/shard.json.{g}"))
        lo, hi = min(n, g * per), min(n, g * per + per)
        assert rec["config"]["device"] == g
        assert rec["nodes"] == _slice_nodes(plain["nodes"], lo, hi), (g, "nodes")
        assert rec["pod"] == _slice_pod(plain["pod"], lo, hi), (g, "pod")
        assert rec["profile"] == plain["profile"]  # (round 6: the percentage too -- the sampled search of a coupled template runs on shards)
        assert rec["dist_comm_init"] == {"n_ranks": n_gpus, "rank": g, "id_ok": 1} and rec["dist_sync_tables"] == n_gpus
        assert rec["dist_run"]["max_limit"] == 0 and rec["dist_run"]["mode"] == plain["run"]["mode"] and rec["dist_run"]["per_node_cap"] >= hi - lo
        assert "run" not in rec
    return p.stdout, n, podspec, snaps


@pytest.mark.parametrize("n_gpus", [2, 3])
@pytest.mark.parametrize("case", sorted(CASES))
def test_native_sharded_host_side_views_threads_and_merge(native, recorder, tmp_path, case, n_gpus):
    """--gpus N without a GPU: tests/abi_recorder.c stands in for libccsim.so.  Every rank's thread must hand the library exactly the
    [lo, hi) slice of the arrays the unsharded host marshals (uneven last shard included), go through comm_init / sync_tables /
    dist_run with the same unique id, and the merge of the per-rank reports (per-node counts concatenated, logs by element-wise
    maximum, FitError histograms summed) must give the review of the recorder's canned global result."""
    objs = CASES[case]()
    exclude = objs[3]
    stdout, n, podspec, snaps = _sharded_records(native, recorder, tmp_path, objs, n_gpus)
    # the merged review: placement i on node i, one per node; the FitError of the canned result
    pypod = cli.parse_pod_spec(podspec)
    snap = ingest.build_snapshot(*cli.load_objects(snaps), pypod, exclude)
    hist = np.zeros(M.NREASON, np.int64)
    hist[M.R_TOO_MANY_PODS] = n
    ht = np.zeros(len(snap.taint_reasons), np.int64)
    if len(ht):
        ht[0] = n_gpus * (n_gpus + 1) // 2
    res = M.RunResult(placed=n, stop=M.STOP_UNSCHEDULABLE, per_node_count=np.ones(n, np.int32), log=np.arange(n, dtype=np.int32), hist=hist,
                      hist_taintset=ht, n_code_unschedulable=1)
    want = cli.build_review(pypod, snap, res, 0)
    got = json.loads(stdout)
    got["status"].pop("creationTimestamp"), want["status"].pop("creationTimestamp", None)
    assert got["status"] == json.loads(json.dumps(want["status"]))


@pytest.mark.parametrize("n_gpus", [1, 2, 3])
def test_native_sharded_templates_one_cycle_at_a_time_threads_and_merge(native, recorder, tmp_path, n_gpus):
    """Several templates with --gpus N (round 5): the rank threads walk the cycles in lock-step -- per cycle every rank sets ITS slice of
    template i mod P with the template's own earlier clones folded in, synchronizes the tables and runs ccsim_dist_run(max_limit = 1).
    Against tests/abi_recorder.c, whose canned cycle c lands on global node c: 12 cycles, one pod per node, templates 0, 1, 2 by turns, the
    13th cycle (template 0) finds nothing; the LAST pod every rank was handed is template 0 with its four clones (nodes 0, 3, 6, 9) in the
    per-node counts of its slice."""
    nodes, pods, templates = _templates_case(n_nodes=12)
    cluster, paths = _write_templates(tmp_path, nodes, pods, templates)
    flags = [x for p in paths for x in ("--podspec", p)] + ["--snapshot", cluster, "-o", "json", "--gpus", str(n_gpus)] + (["--force-sharded"] if n_gpus == 1 else [])
    env = dict(os.environ, CCSIM_LIB=recorder, CCSIM_RECORD=str(tmp_path / "shard.json"), CCSIM_RECORD_PER_DEVICE="1")
    p = subprocess.run([native] + flags, capture_output=True, text=True, env=env, timeout=SUBPROC_TIMEOUT)
    assert p.returncode == 0, p.stderr
    assert "one scheduling cycle at a time" in p.stderr
    st = json.loads(p.stdout)["status"]
    snap = ingest.build_snapshot(nodes, pods, [cli.parse_pod_spec(q) for q in paths])
    assert st["replicas"] == 12 and [[x["nodeName"] for x in q["replicasOnNodes"]] for q in st["pods"]] == [[snap.names[i] for i in range(t, 12, 3)] for t in range(3)]
    assert st["failReason"]["failMessage"].startswith("0/12 nodes are available: 12 Too many pods.")
    per = -(-12 // n_gpus)
    for g in range(n_gpus):
        rec = json.load(open(f"{tmp_path}/shard.json.{g}"))  # (repeated keys: the LAST set_pod / dist_run of the rank)
        lo, hi = min(12, g * per), min(12, g * per + per)
        assert rec["dist_comm_init"] == {"n_ranks": n_gpus, "rank": g, "id_ok": 1} and rec["dist_run"]["max_limit"] == 1 and rec["profile"]["pct"] == 100
        anti = rec["pod"]["ipa"]["anti_existing"][0]["v"]  # template 0: its required hostname anti-affinity sees its own clones
        base = np.asarray(snap.pods[0].ipa.anti_existing[0] if snap.pods[0].ipa.anti_existing[0] is not None else np.zeros(12), np.int64)
        want = base.copy()
        want[[0, 3, 6, 9]] += 1
        assert anti == want[lo:hi].tolist()
        cnt = rec["pod"]["spread"][0]["node_match_count"]
        base = np.asarray(snap.pods[0].spread[0].node_match_count if snap.pods[0].spread[0].node_match_count is not None else np.zeros(12), np.int64)
        want = base.copy()
        want[[0, 3, 6, 9]] += 1
        assert cnt == want[lo:hi].tolist()


def test_default_percentage_of_nodes_to_score_follows_the_reference(native, recorder, tmp_path):
    """the reference passes ComponentConfig.PercentageOfNodesToScore through unchanged (simulator.go:424): its default is the
    adaptive sampling.  Left unset, both hosts score every node only where total and distribution cannot depend on the order: no
    --max-limit and no topology-coupled FILTER.  A template with a DoNotSchedule constraint / required inter-pod (anti-)affinity keeps
    0; one whose coupled plugins only score (ScheduleAnyway, preferred inter-pod terms) is searched completely; a named value wins."""
    nodes, pods, rich, _ = CASES["rich"]()
    soft = rich_pod()
    soft["spec"]["affinity"].pop("podAntiAffinity")
    soft["spec"]["topologySpreadConstraints"] = [c for c in soft["spec"]["topologySpreadConstraints"] if c["whenUnsatisfiable"] == "ScheduleAnyway"]
    plain = yaml.safe_load(EXAMPLES_POD)
    k = [0]

    def pct(pod, extra, existing=True):
        k[0] += 1
        d = tmp_path / f"run{k[0]}"
        d.mkdir()
        podspec, snaps = _write(d, "json", nodes, pods if existing else [], pod)
        env = dict(os.environ, CCSIM_LIB=recorder, CCSIM_RECORD=str(d / "rec.json"))
        p = subprocess.run([native, "--podspec", podspec] + [x for s in snaps for x in ("--snapshot", s)] + ["-o", "json"] + extra, capture_output=True, text=True, env=env,
                           timeout=SUBPROC_TIMEOUT)
        assert p.returncode == 0, p.stderr
        # the Python host applies the same rule (cli.hard_coupled is what it branches on)
        snap = ingest.build_snapshot(*cli.load_objects(snaps), cli.parse_pod_spec(podspec), [])
        return json.load(open(d / "rec.json"))["profile"]["pct"], cli.hard_coupled(snap.pod)

    assert pct(plain, []) == (100, False) and pct(plain, ["--max-limit", "3"]) == (0, False)
    assert pct(rich, []) == (0, True) and pct(rich, ["--max-limit", "3"]) == (0, True)
    assert pct(rich, ["--percentage-of-nodes-to-score", "100"]) == (100, True)
    assert pct(soft, []) == (0, True)  # (existing pods of the rich cluster carry required anti-affinity terms that match the template: a coupled filter)
    assert pct(soft, [], existing=False) == (100, False) and pct(soft, ["--max-limit", "3"], existing=False) == (0, False)


def test_native_sharded_percentage_of_nodes_to_score(native, recorder, tmp_path):
    """--gpus N keeps percentageOfNodesToScore as on one GPU -- unset: every node, or the reference's adaptive default when --max-limit
    makes the order matter; set: as set -- since the sampled search runs on shards (two exchanges per cycle); except for a template
    with topology-coupled plugins, whose shards score every node."""
    def pcts(case, extra):
        nodes, pods, pod, _ = CASES[case]()
        d = tmp_path / (case + "-".join(extra).replace("/", "_"))
        d.mkdir()
        podspec, snaps = _write(d, "json", nodes, pods, pod)
        env = dict(os.environ, CCSIM_LIB=recorder, CCSIM_RECORD=str(d / "shard.json"), CCSIM_RECORD_PER_DEVICE="1")
        p = subprocess.run([native, "--podspec", podspec] + [x for s in snaps for x in ("--snapshot", s)] + ["-o", "json", "--gpus", "2"] + extra,
                           capture_output=True, text=True, env=env, timeout=SUBPROC_TIMEOUT)
        assert p.returncode == 0, p.stderr
        return [json.load(open(f"{d}/shard.json.{g}"))["profile"]["pct"] for g in range(2)]

    assert pcts("readme", []) == [100, 100]
    assert pcts("readme", ["--max-limit", "3"]) == [0, 0]
    assert pcts("readme", ["--percentage-of-nodes-to-score", "30"]) == [30, 30]
    assert pcts("rich", ["--percentage-of-nodes-to-score", "30", "--max-limit", "3"]) == [30, 30]  # (spread constraints + inter-pod affinity: as said, since round 6)


def test_native_sharded_run_keeps_a_coupled_templates_percentage(native, recorder, tmp_path):
    """On one GPU a template with a topology-coupled FILTER keeps the reference's default adaptive sampling; rounds 4-5 scored
    every node on shards (another total for the same input: refused since ADVICE r5 unless 100 was said).  Round 6: the sampled search
    of a coupled template runs on shards (tests/test_sampling.py::test_sampled_search_on_shards_with_topology_coupled_plugins), so
    --gpus N hands the library the SAME percentage as a run without it -- unset: the adaptive default; set: as set."""
    nodes = [node(f"n{i}", cpu="4", mem="8Gi", pods="10", labels={"topology.kubernetes.io/zone": f"z{i % 3}", "kubernetes.io/hostname": f"n{i}"}) for i in range(120)]
    pod = yaml.safe_load(EXAMPLES_POD)
    pod["metadata"]["labels"] = {"app": "sim"}
    pod["spec"]["topologySpreadConstraints"] = [{"maxSkew": 1, "topologyKey": "topology.kubernetes.io/zone", "whenUnsatisfiable": "DoNotSchedule",
                                                 "labelSelector": {"matchLabels": {"app": "sim"}}}]
    podspec, snaps = _write(tmp_path, "json", nodes, [], pod)
    env = dict(os.environ, CCSIM_LIB=recorder, CCSIM_RECORD=str(tmp_path / "shard.json"), CCSIM_RECORD_PER_DEVICE="1")
    base = [native, "--podspec", podspec] + [x for s in snaps for x in ("--snapshot", s)] + ["-o", "json"]
    for extra, want in (([], 0), (["--percentage-of-nodes-to-score", "30"], 30), (["--percentage-of-nodes-to-score", "100"], 100)):
        p = subprocess.run(base + ["--gpus", "2"] + extra, capture_output=True, text=True, env=env, timeout=SUBPROC_TIMEOUT)
        assert p.returncode == 0, p.stderr
        assert [json.load(open(f"{tmp_path}/shard.json.{g}"))["profile"]["pct"] for g in range(2)] == [want, want]
        assert ("adaptive node sampling" in p.stderr) == (want == 0)
        env1 = {k: v for k, v in env.items() if k != "CCSIM_RECORD_PER_DEVICE"}
        one = subprocess.run(base + extra, capture_output=True, text=True, env=dict(env1, CCSIM_RECORD=str(tmp_path / "one.json")), timeout=SUBPROC_TIMEOUT)
        assert one.returncode == 0, one.stderr
        assert json.load(open(tmp_path / "one.json"))["profile"]["pct"] == want  # ... the one-GPU host's choice


@pytest.mark.parametrize("seed", range(16))
def test_native_sharded_host_side_views_random_clusters(native, recorder, tmp_path, seed):
    """The same slicing check over the random clusters / pod specs of the ingest fuzz (spread constraints, inter-pod affinity with
    existing pods, host ports, images), 2 .. 5 shards, more shards than nodes included."""
    rng = np.random.default_rng(9000 + seed)
    objs = _random_objects(rng)
    got = _sharded_records(native, recorder, tmp_path, objs, 2 + seed % 4)
    try:
        podspec, snaps = _write(tmp_path, "json", *objs[:3])
        no, po, ns = cli.load_all(snaps)
        ingest.build_snapshot(no, po, cli.parse_pod_spec(podspec), objs[3], namespace_objs=ns)
    except NotImplementedError:
        assert got is None  # refused by both hosts (volumes, more topology keys than the engine holds, ...)
        return
    assert got is not None


# ---- pruned parsing of cluster dumps (host/value.hpp JsonParser, prune_cluster_objects) -------------------------------------------
def _decorate(obj, rng):
    """What a real `kubectl get -o json` carries around the fields the ingest reads -- all of it must be skipped without a trace."""
    o = json.loads(json.dumps(obj))
    tricky = 'a "quoted" \\ back\\\\slash } ] { [ , : é \n end"'
    md = o.setdefault("metadata", {})
    md["managedFields"] = [{"manager": "kubelet", "operation": "Update", "fieldsType": "FieldsV1", "time": "2025-01-01T00:00:00Z",
                            "fieldsV1": {"f:metadata": {"f:labels": {".": {}, 'f:k"ey': {}}}, "f:status": {"f:conditions": {'k:{"type":"Ready"}': {".": {}}}}}}]
    md["ownerReferences"] = [{"apiVersion": "apps/v1", "kind": "ReplicaSet", "name": "rs", "uid": "1-2-3", "controller": True}]
    md["finalizers"] = ["x/y"]
    md["uid"], md["resourceVersion"], md["creationTimestamp"] = "u-1", "12345", "2025-01-01T00:00:00Z"
    if o.get("kind") != "Namespace":
        md["annotations"] = {"kubectl.kubernetes.io/last-applied-configuration": json.dumps(obj) + tricky, "note": tricky}
    st = o.setdefault("status", {})
    st["conditions"] = [{"type": "Ready", "status": "True", "message": tricky, "lastTransitionTime": None}]
    st["addresses"] = [{"type": "InternalIP", "address": "10.0.0.1"}]
    st["nodeInfo"] = {"kubeletVersion": "v1.34.0", "nested": {"deep": [[1, 2, {"x": [tricky]}], -1.5e3, True, None]}}
    st["capacity"] = {"cpu": "9999", "memory": "9999Gi", "pods": "9999"}
    st["containerStatuses"] = [{"name": "c", "ready": True, "state": {"running": {"startedAt": "2025-01-01T00:00:00Z"}}, "imageID": "sha256:" + "0" * 64}]
    sp = o.setdefault("spec", {})
    if o.get("kind") == "Pod":
        # (kept when a template has volumes of its own to compare them with -- round 5 -- skipped otherwise)
        sp["volumes"] = (sp.get("volumes") or []) + [{"name": "data", "persistentVolumeClaim": {"claimName": "pvc"}},
                                                     {"name": "kube-api-access", "projected": {"sources": [{"serviceAccountToken": {"path": "token"}}]}}]
        sp["tolerations"] = [{"key": "node.kubernetes.io/not-ready", "operator": "Exists", "effect": "NoExecute", "tolerationSeconds": 300}]
        sp["securityContext"], sp["imagePullSecrets"] = {"runAsUser": 1000}, [{"name": "regcred"}]
        for c in (sp.get("containers") or []) + (sp.get("initContainers") or []):
            c["env"] = [{"name": "A", "value": tricky}, {"name": "B", "valueFrom": {"fieldRef": {"fieldPath": "metadata.name"}}}]
            c["volumeMounts"] = [{"name": "data", "mountPath": "/data"}]
            c["livenessProbe"] = {"httpGet": {"path": "/healthz", "port": 8080}}
            c["command"], c["args"] = ["sh", "-c", tricky], ["--flag={}"]
            c["securityContext"] = {"capabilities": {"drop": ["ALL"]}}
    # random junk of random shape and length in skipped places: the skipper works on 64-byte blocks, so strings, escapes and
    # brackets must be met at every alignment, across block boundaries, and in blocks with and without a backslash
    def junk(depth=0):
        r = rng.random()
        if depth > 3 or r < 0.35:
            n = int(rng.integers(0, 150))
            alphabet = ['a', 'b', ' ', '"', '\\', '{', '}', '[', ']', ',', ':', '\n', 'é', '\t', '/']
            return "".join(rng.choice(alphabet, n, p=[.3, .2, .1, .06, .06, .04, .04, .04, .04, .03, .03, .02, .02, .01, .01]))
        if r < 0.5:
            return [None, True, -1.5e3, 0, int(rng.integers(0, 1 << 40))][int(rng.integers(0, 5))]
        if r < 0.75:
            return [junk(depth + 1) for _ in range(int(rng.integers(0, 5)))]
        return {junk(9) or f"k{i}": junk(depth + 1) for i in range(int(rng.integers(0, 5)))}
    md["managedFields"].append(junk())
    st["junk"], st["moreJunk"] = junk(), junk()
    if o.get("kind") != "Namespace":
        md["annotations"]["junk"] = junk(9)
    if rng.random() < 0.5:  # member order as other producers emit it: kind after metadata
        o = {k: o[k] for k in sorted(o, key=lambda k: (k == "kind", k == "apiVersion"))}
    return o


@pytest.mark.parametrize("seed", range(6))
@pytest.mark.parametrize("case", sorted(CASES))
def test_pruned_parse_of_decorated_dumps_gives_the_same_snapshot(native, tmp_path, case, seed):
    nodes, pods, pod, exclude = CASES[case]()
    rng = np.random.default_rng(seed)
    outs = []
    for name, deco in (("plain", False), ("decorated", True)):
        d = tmp_path / name
        d.mkdir()
        ns = [{"kind": "Namespace", "apiVersion": "v1", "metadata": {"name": "default", "labels": {"team": "a"}, "annotations": {"openshift.io/node-selector": "disk=ssd"}}}]
        objs = [dict(o, kind=o.get("kind") or k) for k, lst in (("Node", nodes), ("Pod", pods), ("Namespace", ns)) for o in lst]
        if deco:
            objs = [_decorate(o, rng) for o in objs]
        text = json.dumps({"kind": "List", "apiVersion": "v1", "metadata": {"resourceVersion": ""}, "items": objs}, indent=2 if deco else None,
                          ensure_ascii=bool(rng.integers(0, 2)))
        assert json.loads(text)["items"] == objs
        (d / "cluster.json").write_text(text)
        (d / "pod.json").write_text(json.dumps(pod))
        args = ["--podspec", str(d / "pod.json"), "--snapshot", str(d / "cluster.json"), "--dump-snapshot", "-"] + (["--exclude-nodes", ",".join(exclude)] if exclude else [])
        outs.append(_run(native, args))
        if deco:  # ... and the same through the parallel parse of the List's items (off for small files unless asked for)
            env = dict(os.environ, CCHOST_PARALLEL_MIN_BYTES="0", CCHOST_PARALLEL_MIN_ITEMS="0", CCHOST_THREADS=str(2 + seed % 3))
            par = subprocess.run([native] + args, capture_output=True, text=True, env=env, timeout=SUBPROC_TIMEOUT)
            assert par.returncode == 0 and par.stdout == outs[-1], par.stderr
        if deco:  # ... and as YAML (block style as kubectl / PyYAML emit it, long strings folded, quoted where needed): pruned by indentation
            (d / "cluster.yaml").write_text(yaml.safe_dump({"kind": "List", "apiVersion": "v1", "items": objs}, default_flow_style=False, width=int(rng.choice([60, 80, 1000]))))
            assert yaml.safe_load(open(d / "cluster.yaml"))["items"] == objs
            yargs = ["--podspec", str(d / "pod.json"), "--snapshot", str(d / "cluster.yaml"), "--dump-snapshot", "-"] + (["--exclude-nodes", ",".join(exclude)] if exclude else [])
            assert _run(native, yargs) == outs[0]
        # genpod reads Namespace annotations: they survive the pruning whatever the member order
        g = yaml.safe_load(_run(native, ["--genpod", "default", "--snapshot", str(d / "cluster.json")]))
        assert g["spec"]["nodeSelector"] == {"disk": "ssd"}
    assert outs[0] == outs[1]


def test_pruned_parse_still_rejects_broken_json(native, tmp_path):
    (tmp_path / "pod.json").write_text(EXAMPLES_POD)
    for text in ('{"kind": "List", "items": [{"kind": "Node", "metadata": {"name": "n", "managedFields": [{"a": "unterminated}]}}]}',
                 '{"kind": "Node", "metadata": {"name": "n"}, "status": {"conditions": [1, 2'):
        (tmp_path / "c.json").write_text(text)
        for env in (None, dict(os.environ, CCHOST_PARALLEL_MIN_BYTES="0", CCHOST_THREADS="2")):
            p = subprocess.run([native, "--podspec", str(tmp_path / "pod.json"), "--snapshot", str(tmp_path / "c.json"), "--dump-snapshot", "-"], capture_output=True, text=True,
                               timeout=SUBPROC_TIMEOUT, env=env)
            assert p.returncode != 0
    # an error inside an element must surface from the worker threads too
    items = [{"kind": "Node", "metadata": {"name": f"n{i}"}, "status": {"allocatable": {"cpu": "1", "pods": "1"}}} for i in range(40)]
    text = json.dumps({"kind": "List", "items": items}).replace('"n17"', '"n17" oops')
    (tmp_path / "c.json").write_text(text)
    p = subprocess.run([native, "--podspec", str(tmp_path / "pod.json"), "--snapshot", str(tmp_path / "c.json"), "--dump-snapshot", "-"], capture_output=True, text=True,
                       timeout=SUBPROC_TIMEOUT, env=dict(os.environ, CCHOST_PARALLEL_MIN_BYTES="0", CCHOST_THREADS="4"))
    assert p.returncode != 0 and "cluster-capacity:" in p.stderr


def test_several_templates_unschedulable_stop_names_the_failing_template(native, tmp_path, capsys):
    """The FitError of a several-templates run describes the template whose pod did not fit (its taint reasons, its preemption side);
    templates of different priority -> clones of one could be victims of another: both hosts flag the dry run as not modelled."""
    nodes, pods, templates = _templates_case()
    nodes[0]["spec"]["taints"] = [{"key": "dedicated", "value": "x", "effect": "NoSchedule"}]
    templates[1]["spec"]["tolerations"] = [{"key": "dedicated", "operator": "Exists"}]
    templates[2]["spec"]["priority"] = 10
    cluster, paths = _write_templates(tmp_path, nodes, pods, templates)
    pypods = [cli.parse_pod_spec(p) for p in paths]
    snap = ingest.build_snapshot(*cli.load_objects([cluster]), pypods)
    n = len(snap.names)
    hist = np.zeros(M.NREASON, np.int64)
    hist[M.R_RES0] = n - 1
    for failing in (0, 1, 2):
        ht = np.zeros(len(snap.taint_reasons_all[failing]), np.int64)
        ts = int(snap.nodes.taintset_id[snap.names.index(nodes[0]["metadata"]["name"])])
        ht[ts] = 0 if failing == 1 else 1
        res = M.RunResult(placed=6, stop=M.STOP_UNSCHEDULABLE, per_node_count=np.bincount([0, 1, 2, 3, 4, 5], minlength=n).astype(np.int32),
                          log=np.arange(6, dtype=np.int32), hist=hist, hist_taintset=ht, n_code_unschedulable=n - 1, stop_spec=failing)
        (tmp_path / "result.json").write_text(json.dumps({
            "placed": res.placed, "stop": res.stop, "n_code_unschedulable": res.n_code_unschedulable, "per_node_count": res.per_node_count.tolist(),
            "log": res.log.tolist(), "hist": res.hist.tolist(), "hist_taintset": res.hist_taintset.tolist(), "stop_spec": failing}))
        p = subprocess.run([native] + [x for q in paths for x in ("--podspec", q)] + ["--snapshot", cluster, "--fake-result", str(tmp_path / "result.json"), "-o", "json"],
                           capture_output=True, text=True, timeout=SUBPROC_TIMEOUT)
        assert p.returncode == 0, p.stderr
        want = cli.build_review(pypods, snap, res, 0)
        assert json.loads(p.stdout)["status"]["failReason"] == want["status"]["failReason"], failing
        assert ("not modelled" in p.stderr) and ("not modelled" in capsys.readouterr().err)
        assert ("untolerated taint {dedicated: x}" in want["status"]["failReason"]["failMessage"]) == (failing != 1)


# ---- podRequirements: sums of Quantities printed by Quantity.String() -----------------------------------------------------------------
def _requirements(native, tmp_path, cpu, mem):
    pod = yaml.safe_load(EXAMPLES_POD)
    k = max(len(cpu), len(mem))
    cpu, mem = list(cpu) + [None] * (k - len(cpu)), list(mem) + [None] * (k - len(mem))
    pod["spec"]["containers"] = [{"name": f"c{i}", "resources": {"requests": {k: v for k, v in (("cpu", c), ("memory", m)) if v is not None}}}
                                 for i, (c, m) in enumerate(zip(cpu, mem))]
    (tmp_path / "pod.json").write_text(json.dumps(pod))
    (tmp_path / "cluster.json").write_text(json.dumps({"kind": "List", "items": [dict(node("n1"), kind="Node")]}))
    (tmp_path / "result.json").write_text(json.dumps({"placed": 0, "stop": M.STOP_LIMIT, "n_code_unschedulable": 0, "per_node_count": [0], "log": [],
                                                      "hist": [0] * M.NREASON, "hist_taintset": [0]}))
    out = json.loads(_run(native, ["--podspec", str(tmp_path / "pod.json"), "--snapshot", str(tmp_path / "cluster.json"), "--fake-result", str(tmp_path / "result.json"),
                                   "--max-limit", "1", "-o", "json"]))
    got = out["spec"]["podRequirements"][0]["resources"]["primaryResources"]
    want = cli.pod_requirements(cli.parse_pod_spec(str(tmp_path / "pod.json")))["resources"]["primaryResources"]
    assert got == want, (cpu, mem)
    return got["cpu"], got["memory"]


def test_pod_requirements_print_quantities_like_the_reference(native, tmp_path):
    """report.go:111-144 + Quantity.String() (quantity.go:424-461,600-613; amount.go:257-293), derived by hand from those lines:
    the sum takes the format of the last non-zero addend; BinarySI removes factors of 1024 and falls back to decimal below 1024 or
    for fractions; the decimal forms strip trailing zeros and lower the exponent to a multiple of three."""
    f = lambda cpu, mem: _requirements(native, tmp_path, cpu, mem)
    assert f(["150m"], ["100Mi"]) == ("150m", "100Mi")
    assert f(["1", "500m"], ["1Gi", "512Mi"]) == ("1500m", "1536Mi")
    assert f(["2"], ["512M"]) == ("2", "512M")                      # a decimal memory request stays decimal
    assert f(["12000"], ["1G", "1Gi"]) == ("12k", "2073741824")     # 1e9 + 2^30 in BinarySI: no factor of 1024 left
    assert f(["0.1"], ["500"]) == ("100m", "500")
    assert f(["1e3"], ["1.5Gi"]) == ("1e3", "1536Mi")
    assert f(["100m", "1e3"], ["0.5"]) == ("1000100e-3", "500m")
    assert f(["1500u"], ["1000Ki"]) == ("1500u", "1000Ki")
    assert f(["0", None], [None, "128974848"]) == ("0", "128974848")
    assert f(["1Gi"], ["1Gi", "0"]) == ("1Gi", "1Gi")               # a zero addend leaves the format alone
    assert f([None], [None]) == ("0", "0")


def test_pod_requirements_random_quantities(native, tmp_path):
    rng = np.random.default_rng(31)
    sufs = ["", "m", "u", "n", "k", "M", "G", "Ki", "Mi", "Gi", "e3", "e-3", "E2"]
    def q():
        if rng.random() < 0.15:
            return None
        num = str(int(rng.integers(0, 5000))) if rng.random() < 0.6 else f"{rng.integers(0, 500)}.{rng.integers(0, 1000):03d}"
        return num + str(rng.choice(sufs))
    for k in range(60):
        n = int(rng.integers(1, 4))
        d = tmp_path / str(k)
        d.mkdir()
        _requirements(native, d, [q() for _ in range(n)], [q() for _ in range(n)])


def test_system_default_spreading_is_flagged_by_both_hosts(native, tmp_path, capsys):
    """PodTopologySpread's system defaults (plugin.go:48-59) apply to a pod without constraints that a Service of its namespace selects
    (helper/spread.go:37-116): not modelled -> both hosts say so; no Service / another namespace / own constraints -> silence."""
    nodes, pods, pod, _ = CASES["readme"]()
    svc = lambda ns, sel: {"kind": "Service", "apiVersion": "v1", "metadata": {"name": "s", "namespace": ns}, "spec": {"selector": sel}}
    rs = lambda ns, sel: {"kind": "ReplicaSet", "apiVersion": "apps/v1", "metadata": {"name": "web-abc", "namespace": ns}, "spec": {"selector": sel}}
    owned = {"ownerReferences": [{"apiVersion": "apps/v1", "kind": "ReplicaSet", "name": "web-abc", "controller": True}]}
    cases = [([svc("default", {"app": "guestbook"})], True), ([svc("default", {"app": "guestbook", "tier": "frontend"})], True),
             ([rs("default", {"matchLabels": {"app": "guestbook"}}), owned], True), ([rs("other", {"matchLabels": {"app": "guestbook"}}), owned], False),
             ([rs("default", {}), owned], False), ([rs("default", {"matchLabels": {"app": "guestbook"}})], False),
             ([svc("other", {"app": "guestbook"})], False), ([svc("default", {"app": "nope"})], False), ([svc("default", {})], False),
             ([svc("default", None)], False), ([], False)]
    (tmp_path / "result.json").write_text(json.dumps({"placed": 0, "stop": M.STOP_LIMIT, "n_code_unschedulable": 0, "per_node_count": [0] * len(nodes), "log": [],
                                                      "hist": [0] * M.NREASON, "hist_taintset": [0]}))
    for k, (services, expect) in enumerate(cases):
        for own in (False, True):
            d = tmp_path / f"{k}{int(own)}"
            d.mkdir()
            tpl = json.loads(json.dumps(pod))
            objs = [o for o in services if "kind" in o]
            for o in services:
                if "kind" not in o:  # template metadata (owner references) rides along in the case list
                    tpl["metadata"].update(json.loads(json.dumps(o)))
            if own:
                tpl["spec"]["topologySpreadConstraints"] = [{"maxSkew": 1, "topologyKey": "zone", "whenUnsatisfiable": "ScheduleAnyway", "labelSelector": {"matchLabels": {"app": "guestbook"}}}]
            (d / "cluster.json").write_text(json.dumps({"kind": "List", "items": [dict(n, kind="Node") for n in nodes] + objs}))
            (d / "pod.json").write_text(json.dumps(tpl))
            p = subprocess.run([native, "--podspec", str(d / "pod.json"), "--snapshot", str(d / "cluster.json"), "--fake-result", str(tmp_path / "result.json"), "--max-limit", "1"],
                               capture_output=True, text=True, timeout=SUBPROC_TIMEOUT)
            assert p.returncode == 0, p.stderr
            assert ("system default spreading" in p.stderr) == (expect and not own), (k, own, p.stderr)
            owners = [o for kind in ("ReplicationController", "ReplicaSet", "StatefulSet") for o in cli.load_kind([str(d / "cluster.json")], kind)]
            assert ingest.default_spreading_applies(cli.parse_pod_spec(str(d / "pod.json")), cli.load_kind([str(d / "cluster.json")], "Service"), owners) == (expect and not own)


def test_system_default_spreading_becomes_two_soft_constraints_when_every_node_is_labelled(native, tmp_path, ccref):
    """Every node carries kubernetes.io/hostname and topology.kubernetes.io/zone: requireAllTopologies = false and = true read the same
    (scoring.go:61-115), so the system defaults are exactly two ScheduleAnyway constraints of the pod with the merged Service selector.
    Both ingests must build the same snapshot; the total equals the run without the Service, the order of the placements need not."""
    nodes = [node(f"n{i}", cpu="2", mem="4G", labels={"kubernetes.io/hostname": f"n{i}", "topology.kubernetes.io/zone": f"z{i % 3}"}) for i in range(9)]
    pods = [running_pod(f"p{j}", f"n{j % 4}", cpu="100m", mem="64Mi", labels={"app": "guestbook", "tier": "frontend" if j % 2 else "backend"}) for j in range(7)]
    pod = yaml.safe_load(EXAMPLES_POD)
    svcs = [{"kind": "Service", "apiVersion": "v1", "metadata": {"name": "fe", "namespace": "default"}, "spec": {"selector": {"app": "guestbook"}}},
            {"kind": "Service", "apiVersion": "v1", "metadata": {"name": "fe2", "namespace": "default"}, "spec": {"selector": {"tier": "frontend"}}},
            {"kind": "Service", "apiVersion": "v1", "metadata": {"name": "other", "namespace": "default"}, "spec": {"selector": {"app": "db"}}}]
    (tmp_path / "pod.json").write_text(json.dumps(pod))
    outs = {}
    for name, objs in (("with", svcs), ("without", [])):
        path = tmp_path / f"{name}.json"
        path.write_text(json.dumps({"kind": "List", "items": [dict(n, kind="Node") for n in nodes] + [dict(p, kind="Pod") for p in pods] + objs}))
        p = subprocess.run([native, "--podspec", str(tmp_path / "pod.json"), "--snapshot", str(path), "--dump-snapshot", "-"], capture_output=True, text=True, timeout=SUBPROC_TIMEOUT)
        assert p.returncode == 0 and p.stderr == "", p.stderr
        got = json.loads(p.stdout)
        got.pop("label_keys")
        no, po, ns = cli.load_all([str(path)])
        snap = ingest.build_snapshot(no, po, cli.parse_pod_spec(str(tmp_path / "pod.json")), namespace_objs=ns, service_objs=cli.load_kind([str(path)], "Service"))
        ref = py_dump(snap)
        for k in ref:
            assert got[k] == ref[k], (name, k)
        outs[name] = snap
    a, b = outs["with"].pod, outs["without"].pod
    assert not b.spread and [(c.max_skew, c.hard, c.is_hostname, c.self_match) for c in a.spread] == [(3, False, True, True), (5, False, False, True)]
    # the merged selector app=guestbook,tier=frontend: p1, p3, p5 match -> on n1, n3, n1
    assert a.spread[0].node_match_count.tolist() == [0, 2, 0, 1, 0, 0, 0, 0, 0]
    assert a.soft_relaxed and not b.soft_relaxed
    ra, rb = ccref.run(M.Profile.default(), outs["with"].nodes, a), ccref.run(M.Profile.default(), outs["without"].nodes, b)
    assert ra.placed == rb.placed and ra.per_node_count.tolist() == rb.per_node_count.tolist() and ra.log.tolist() != rb.log.tolist()


def test_system_default_spreading_on_nodes_without_a_zone_label(native, recorder, tmp_path, ccref):
    """requireAllTopologies = false (scoring.go:140): nodes WITHOUT topology.kubernetes.io/zone are not ignored -- they score their
    hostname count and nothing for the zone.  Both ingests keep the reference's form (label id 0 = key missing, PodSide::soft_relaxed);
    both hosts hand the engine the derived form: one more label column in which the nodes without the key carry one more value id, named
    by ccsim_spread_constraint::missing_value (the ABI recorder shows what the native host marshals)."""
    nodes = [node(f"n{i}", cpu="2", mem="4G", labels=dict({"kubernetes.io/hostname": f"n{i}"}, **({"topology.kubernetes.io/zone": f"z{i % 2}"} if i % 3 else {}))) for i in range(9)]
    pods = [running_pod(f"p{j}", f"n{j % 4}", cpu="100m", mem="64Mi", labels={"app": "guestbook"}) for j in range(5)]
    pod = yaml.safe_load(EXAMPLES_POD)
    svcs = [{"kind": "Service", "apiVersion": "v1", "metadata": {"name": "fe", "namespace": "default"}, "spec": {"selector": {"app": "guestbook"}}}]
    (tmp_path / "pod.json").write_text(json.dumps(pod))
    path = tmp_path / "cluster.json"
    path.write_text(json.dumps({"kind": "List", "items": [dict(n, kind="Node") for n in nodes] + [dict(p, kind="Pod") for p in pods] + svcs}))
    p = subprocess.run([native, "--podspec", str(tmp_path / "pod.json"), "--snapshot", str(path), "--dump-snapshot", "-"], capture_output=True, text=True, timeout=SUBPROC_TIMEOUT)
    assert p.returncode == 0 and "system default spreading" not in p.stderr, p.stderr
    got = json.loads(p.stdout)
    got.pop("label_keys")
    no, po, ns = cli.load_all([str(path)])
    snap = ingest.build_snapshot(no, po, cli.parse_pod_spec(str(tmp_path / "pod.json")), namespace_objs=ns, service_objs=cli.load_kind([str(path)], "Service"))
    ref = py_dump(snap)
    for k in ref:
        assert got[k] == ref[k], k
    assert snap.pod.soft_relaxed and not snap.default_spreading_unmodelled and [c.is_hostname for c in snap.pod.spread] == [True, False]
    zone = snap.pod.spread[1]
    assert (np.asarray(snap.nodes.label_cols[zone.col]) == 0).sum() == 3 and zone.n_domains == 2
    # the oracle on the reference's form: the first clone goes to a node without a zone label and without matching pods (hostname count 0, no
    # zone credit) -- with requireAllTopologies = true those nodes would be ignored (score 0)
    r = ccref.run(M.Profile.default(), snap.nodes, snap.pod, max_limit=3)
    assert int(r.log[0]) in (6,) and snap.names[int(r.log[0])] == "n6"
    # what the native host hands the engine: a fourth... one more label column, the zone constraint on it with the extra value id named
    rec = tmp_path / "rec.json"
    (tmp_path / "result.json").write_text(json.dumps({"placed": 0, "stop": M.STOP_LIMIT, "n_code_unschedulable": 0, "per_node_count": [0] * 9, "log": [],
                                                      "hist": [0] * M.NREASON, "hist_taintset": [0]}))
    p = subprocess.run([native, "--podspec", str(tmp_path / "pod.json"), "--snapshot", str(path), "--max-limit", "1"], capture_output=True, text=True, timeout=SUBPROC_TIMEOUT,
                       env=dict(os.environ, CCSIM_LIB=recorder, CCSIM_RECORD=str(rec)))
    assert p.returncode == 0, p.stderr
    native_rec = json.load(open(rec))
    e_nodes, e_pod = M.relax_soft(snap.nodes, snap.pod)
    k = native_rec["pod"]["spread"][1]["k"]
    ez = e_pod.spread[1]
    assert k == [ez.col, 5, 1, 0, 1, 3, 0, 3] and ez.missing_value == 3 and ez.n_domains == 3
    col = native_rec["nodes"]["label_cols"][ez.col]
    col = col["v"] if isinstance(col, dict) else col
    assert col == [int(v) for v in e_nodes.label_cols[ez.col]] and col.count(3) == 3


@pytest.mark.gpu
def test_system_default_spreading_without_zone_labels_end_to_end(native, tmp_path, ccref):
    """The on-premises cluster of the reference's comment (scoring.go:137-139): some nodes carry no zone label, a Service selects the
    simulated pod.  C++ host -> C ABI -> HIP engine == Python host == the oracle's literal requireAllTopologies = false branch: the same
    replicas per node in the same (first placement) order."""
    import io
    nodes = [node(f"n{i:02d}", cpu="2", mem="4G", labels=dict({"kubernetes.io/hostname": f"n{i:02d}"}, **({"topology.kubernetes.io/zone": f"z{i % 3}"} if i % 4 else {}))) for i in range(14)]
    pods = [running_pod(f"p{j}", f"n{j % 5:02d}", cpu="100m", mem="64Mi", labels={"app": "guestbook"}) for j in range(8)]
    pod = yaml.safe_load(EXAMPLES_POD)
    svc = {"kind": "Service", "apiVersion": "v1", "metadata": {"name": "fe", "namespace": "default"}, "spec": {"selector": {"app": "guestbook"}}}
    (tmp_path / "pod.json").write_text(json.dumps(pod))
    path = tmp_path / "cluster.json"
    path.write_text(json.dumps({"kind": "List", "items": [dict(n, kind="Node") for n in nodes] + [dict(p, kind="Pod") for p in pods] + [svc]}))
    flags = ["--podspec", str(tmp_path / "pod.json"), "--snapshot", str(path), "--max-limit", "40", "-o", "json", "--percentage-of-nodes-to-score", "100"]
    got = json.loads(_run(native, flags))
    buf = io.StringIO()
    assert cli.main(flags, out=buf) == 0
    ref = json.loads(buf.getvalue())
    got["status"].pop("creationTimestamp"), ref["status"].pop("creationTimestamp")
    assert got["status"] == ref["status"] and got["status"]["replicas"] == 40
    no, po, ns = cli.load_all([str(path)])
    snap = ingest.build_snapshot(no, po, cli.parse_pod_spec(str(tmp_path / "pod.json")), namespace_objs=ns, service_objs=cli.load_kind([str(path)], "Service"))
    assert snap.pod.soft_relaxed
    r = ccref.run(M.Profile.default(), snap.nodes, snap.pod, max_limit=40)
    order = []
    for i in r.log.tolist():
        if snap.names[i] not in order:
            order.append(snap.names[i])
    rows = got["status"]["pods"][0]["replicasOnNodes"]
    assert [x["nodeName"] for x in rows] == order and [x["replicas"] for x in rows] == [int(r.per_node_count[snap.names.index(nm)]) for nm in order]


def test_genpod_prints_quantities_canonically(native, tmp_path):
    """The stub pod leaves through the serializer, which prints Quantity.String(), not the text the LimitRange carried."""
    from cluster_capacity_amd import genpod
    ns = [{"kind": "Namespace", "metadata": {"name": "q"}}]
    lrs = [{"kind": "LimitRange", "metadata": {"name": "l", "namespace": "q"}, "spec": {"limits": [{"type": "Pod", "max": {"cpu": "0.5", "memory": "1024Mi", "nvdia.com/gpu": "2e0"}}]}}]
    want = {"cpu": "500m", "memory": "1Gi", "nvdia.com/gpu": "2"}
    assert genpod.namespace_pod("q", ns, lrs)["spec"]["containers"][0]["resources"]["requests"] == want
    (tmp_path / "ns.json").write_text(json.dumps({"kind": "List", "items": ns + lrs}))
    out = json.loads(_run(native, ["--genpod", "q", "--snapshot", str(tmp_path / "ns.json"), "-o", "json"]))
    assert out["spec"]["containers"][0]["resources"] == {"limits": want, "requests": want}
