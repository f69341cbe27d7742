"""The native host (the product's ingest + CLI, cluster-capacity_amd/host/) against damaged input: it must refuse cleanly -- exit code 1 and a
message -- never crash, never wrap an integer around, never read a string as "absent".  The reference decodes objects into typed structs
(a string where a mapping belongs is a decode error there); value.hpp's accessors refuse the same way, for every member the ingest reads
(what it never looks at -- pruned members, the template's labels when no selector refers to them -- is not validated).

Run the same tests (and tests/test_native_host.py) with the host built under the sanitizers:

    CCHOST_SANITIZE=address,undefined ASAN_OPTIONS=detect_leaks=0 python -m pytest tests/test_host_robustness.py tests/test_native_host.py -m "not gpu"
    CCHOST_SANITIZE=thread python -m pytest tests/test_native_host.py -m "not gpu"

(build.build_host() then builds bin/cluster-capacity-native-<sanitizers> with -fno-sanitize-recover: any report is a non-zero exit the
tests below see as a failure.)"""
import copy
import json
import os
import random
import subprocess

import numpy as np
import pytest
import yaml

from cluster_capacity_amd import build as B, cli, ingest
from test_native_host import _random_objects, py_dump


@pytest.fixture(scope="module")
def native():
    return B.build_host()


ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=0")  # (the host leaves through _Exit: nothing is freed on purpose)
TOKENS = [b"{", b"}", b"[", b"]", b":", b",", b'"', b"\\", b"\n", b"  ", b"- ", b"null", b"-", b"1e999", b"\x00", b"\xff", b"\t", b"#", b"|", b">", b"&a", b"*a",
          b"---\n", b"'", b"\\u12", b"\\ud800", b"9" * 40, b"0.5m", b"Ki"]
JUNK = [None, True, 0, -1, 1.5, "", "x", "1Gi", [], {}, [1], ["a"], {"a": "b"}, [{}], [[]], {"matchLabels": 3}, "9" * 30, 1 << 70, [None], {"": ""}]


def _damage_bytes(rng, b):
    b = bytearray(b)
    for _ in range(rng.choice([1, 1, 2, 3, 8])):
        if not b:
            break
        k, i = rng.randrange(6), rng.randrange(len(b))
        if k == 0:
            b[i] = rng.randrange(256)
        elif k == 1:
            del b[i:i + rng.choice([1, 1, 2, 16, 200])]
        elif k == 2:
            b[i:i] = rng.choice(TOKENS)
        elif k == 3:
            b = b[:i]
        elif k == 4:
            j = rng.randrange(len(b))
            b[i:i] = b[j:j + rng.choice([1, 8, 64])]
        else:
            b[i] ^= 1 << rng.randrange(8)
    return bytes(b)


def _paths(o, acc, pre=()):
    for k, v in (o.items() if isinstance(o, dict) else enumerate(o) if isinstance(o, list) else ()):
        acc.append(pre + (k,))
        _paths(v, acc, pre + (k,))


def _damage_structure(rng, obj):
    """Another kind of value at a random place of the object: a string where a mapping belongs, a list where a number does, a member gone."""
    for _ in range(rng.choice([1, 1, 2, 4])):
        acc = []
        _paths(obj, acc)
        if not acc:
            return
        p = rng.choice(acc)
        o = obj
        for k in p[:-1]:
            o = o[k]
        r = rng.random()
        if r < 0.7:
            o[p[-1]] = copy.deepcopy(rng.choice(JUNK))
        elif r < 0.85 and isinstance(o, dict):
            del o[p[-1]]
        elif isinstance(o, dict):
            o[rng.choice(["", "spec", "metadata", "labels", "name", "x"])] = copy.deepcopy(rng.choice(JUNK))


def _run(native, tmp_path, cluster: bytes, pod: bytes, fmt, threads=False):
    (tmp_path / f"c.{fmt}").write_bytes(cluster)
    (tmp_path / f"p.{fmt}").write_bytes(pod)
    env = dict(ENV, CCHOST_PARALLEL_MIN_BYTES="0", CCHOST_PARALLEL_MIN_ITEMS="0") if threads else ENV
    return subprocess.run([native, "--podspec", str(tmp_path / f"p.{fmt}"), "--snapshot", str(tmp_path / f"c.{fmt}"), "--dump-snapshot", "-"], capture_output=True, env=env,
                          timeout=120)


def _clean(p):
    return p.returncode in (0, 1) and b"Sanitizer" not in p.stderr and b"runtime error:" not in p.stderr and (p.returncode == 0 or p.stderr.strip())


@pytest.mark.parametrize("seed", range(6))
def test_damaged_bytes_are_refused_cleanly(native, tmp_path, seed):
    rng = random.Random(4100 + seed)
    for it in range(12):
        nodes, pods, pod, _ = _random_objects(np.random.default_rng(rng.randrange(1 << 30)))
        fmt = rng.choice(["json", "yaml"])
        if fmt == "json":
            cl, pd = json.dumps({"kind": "List", "items": nodes + pods}).encode(), json.dumps(pod).encode()
        else:
            cl, pd = yaml.safe_dump_all(nodes + pods).encode(), yaml.safe_dump(pod).encode()
        which = rng.randrange(3)
        cl = cl if which == 1 else _damage_bytes(rng, cl)
        pd = pd if which == 0 else _damage_bytes(rng, pd)
        p = _run(native, tmp_path, cl, pd, fmt, threads=rng.random() < 0.3)
        assert _clean(p), (seed, it, p.returncode, p.stderr[-400:])


@pytest.mark.parametrize("seed", range(6))
def test_objects_of_the_wrong_shape_are_refused_or_read_alike(native, tmp_path, seed):
    """...and where both hosts take the damaged object (a member gone is a zero value), they read the same snapshot."""
    rng = random.Random(5200 + seed)
    both = 0
    for it in range(14):
        nodes, pods, pod, _ = _random_objects(np.random.default_rng(rng.randrange(1 << 30)))
        nodes, pods, pod = copy.deepcopy(nodes), copy.deepcopy(pods), copy.deepcopy(pod)
        _damage_structure(rng, pod if rng.randrange(3) == 0 or not (nodes + pods) else rng.choice(nodes + pods))
        p = _run(native, tmp_path, json.dumps({"kind": "List", "items": nodes + pods}).encode(), json.dumps(pod).encode(), "json")
        assert _clean(p), (seed, it, p.returncode, p.stderr[-400:])
        if p.returncode:
            continue
        try:
            no, po, ns = cli.load_all([str(tmp_path / "c.json")])
            ref = py_dump(ingest.build_snapshot(no, po, cli.parse_pod_spec(str(tmp_path / "p.json")), [], namespace_objs=ns))
        except BaseException:  # the Python mirror walks plain dicts: it trips over shapes the native host reads as zero values
            continue
        got = json.loads(p.stdout)
        for k in ref:
            assert got[k] == ref[k], (seed, it, k)
        both += 1
    assert both >= 2


def _case(tmp_path, native, pod_patch=None, node_patch=None, n_pods=0):
    node = {"kind": "Node", "metadata": {"name": "n0", "labels": {"kubernetes.io/hostname": "n0"}}, "status": {"allocatable": {"cpu": "4", "memory": "8Gi", "pods": "110"}}}
    pod = {"kind": "Pod", "metadata": {"name": "p", "namespace": "default"}, "spec": {"containers": [{"name": "c", "resources": {"requests": {"cpu": "100m", "memory": "64Mi"}}}]}}
    (node_patch or (lambda n: None))(node)
    (pod_patch or (lambda p: None))(pod)
    existing = [dict(copy.deepcopy(pod), metadata={"name": f"e{i}", "namespace": "default"}, spec=dict(copy.deepcopy(pod["spec"]), nodeName="n0"), status={"phase": "Running"})
                for i in range(n_pods)]
    p = _run(native, tmp_path, json.dumps({"kind": "List", "items": [node] + existing}).encode(), json.dumps(pod).encode(), "json")
    try:
        no, po, ns = cli.load_all([str(tmp_path / "c.json")])
        ingest.build_snapshot(no, po, cli.parse_pod_spec(str(tmp_path / "p.json")), [], namespace_objs=ns)
        py = None
    except (ValueError, TypeError, AttributeError, KeyError, OverflowError, NotImplementedError) as e:
        py = str(e)
    return p, py


def test_quantities_and_integers_out_of_range_are_refused_by_both_hosts(native, tmp_path):
    def big_request(pod):
        pod["spec"]["containers"][0]["resources"]["requests"]["memory"] = 1 << 70
    p, py = _case(tmp_path, native, pod_patch=big_request)
    assert p.returncode == 1 and b"out of range (beyond 2^60)" in p.stderr and "out of range (beyond 2^60)" in py
    # the largest suffix still reads (1E = 10^18 < 2^60); the sum of nine pods of that size leaves int64: refused, not wrapped around

    def exa(pod):
        pod["spec"]["containers"][0]["resources"]["requests"]["memory"] = "1E"
    p, py = _case(tmp_path, native, pod_patch=exa)
    assert p.returncode == 0 and py is None
    p, py = _case(tmp_path, native, pod_patch=exa, n_pods=10)
    assert p.returncode == 1 and b"sum beyond int64" in p.stderr and py is not None

    def pods_2_40(node):
        node["status"]["allocatable"]["pods"] = str(1 << 40)
    p, py = _case(tmp_path, native, node_patch=pods_2_40)
    assert p.returncode == 1 and b"allocatable pods" in p.stderr and "allocatable pods" in py

    def port(pod):
        pod["spec"]["containers"][0]["ports"] = [{"containerPort": 80, "hostPort": 1 << 70}]
    p, py = _case(tmp_path, native, pod_patch=port)
    assert p.returncode == 1 and b"expected an integer" in p.stderr and "malformed object" in py

    def skew(pod):
        pod["spec"]["topologySpreadConstraints"] = [{"maxSkew": 1 << 40, "topologyKey": "kubernetes.io/hostname", "whenUnsatisfiable": "DoNotSchedule", "labelSelector": {}}]
    p, py = _case(tmp_path, native, pod_patch=skew)
    assert p.returncode == 1 and b"does not fit an int32" in p.stderr and "does not fit an int32" in py

    def weight(pod):
        pod["spec"]["affinity"] = {"nodeAffinity": {"preferredDuringSchedulingIgnoredDuringExecution": [{"weight": "1Gi", "preference": {"matchExpressions": []}}]}}
    p, py = _case(tmp_path, native, pod_patch=weight)
    assert p.returncode == 1 and b"expected an integer" in p.stderr and "expected an integer" in py

    def quoted_port(pod):  # a JSON string where an int32 is declared: the reference's typed decoder refuses it, digits or not
        pod["spec"]["containers"][0]["ports"] = [{"containerPort": 80, "hostPort": "8080"}]
    p, py = _case(tmp_path, native, pod_patch=quoted_port)
    assert p.returncode == 1 and b"expected an integer" in p.stderr and "expected an integer" in py


@pytest.mark.parametrize("where,junk,message", [
    (("spec", "containers"), "x", b"expected a list, found a string"),
    (("spec",), "x", b"expected a mapping, found a string"),
    (("spec", "nodeSelector"), ["a"], b"expected a mapping, found a list"),
    (("spec", "containers", 0, "resources"), 3, b"expected a mapping, found a number"),
    (("spec", "tolerations"), {"key": "a"}, b"expected a list, found a mapping"),
    (("spec", "containers", 0, "resources", "requests", "cpu"), ["1"], b"expected a scalar, found a list"),
])
def test_a_value_of_the_wrong_kind_is_a_decode_error_not_an_absent_field(native, tmp_path, where, junk, message):
    def patch(pod):
        o = pod
        for k in where[:-1]:
            o = o[k]
        o[where[-1]] = junk
    p, py = _case(tmp_path, native, pod_patch=patch)
    assert p.returncode == 1 and message in p.stderr, p.stderr
    assert py is not None  # (the Python mirror trips over the same value)
    # null, on the other hand, is the zero value -- absent -- as in the reference's decoder
    p, py = _case(tmp_path, native, pod_patch=lambda pod: pod["spec"].update(nodeSelector=None, tolerations=None, affinity=None))
    assert p.returncode == 0 and py is None


def test_python_cli_refuses_a_malformed_object_cleanly(tmp_path, capsys):
    node = {"kind": "Node", "metadata": {"name": "n0"}, "status": {"allocatable": {"cpu": "4", "memory": "8Gi", "pods": "110"}}}
    pod = {"kind": "Pod", "metadata": {"name": "p"}, "spec": {"containers": "x"}}
    (tmp_path / "c.json").write_text(json.dumps(node))
    (tmp_path / "p.json").write_text(json.dumps(pod))
    assert cli.main(["--podspec", str(tmp_path / "p.json"), "--snapshot", str(tmp_path / "c.json")]) == 1
    assert "malformed object" in capsys.readouterr().err


@pytest.mark.parametrize("fmt,text", [
    ("json", "[" * 1_000_000),
    ("json", '{"a":' * 1_000_000),
    ("json", '{"kind":"List","items":[{"kind":"Node","metadata":{"labels":' + '{"a":' * 500_000),
    ("yaml", "".join(" " * i + "a:\n" for i in range(3000))),
    ("yaml", "".join(" " * i + "-\n" for i in range(3000))),
    ("yaml", "a: " + "[" * 200_000),
])
def test_nesting_depth_is_bounded(native, tmp_path, fmt, text):
    """The readers recurse; the depth is the input's to choose, the stack is not (encoding/json stops at 10000 levels, the hosts at 2000)."""
    pod = {"kind": "Pod", "metadata": {"name": "p"}, "spec": {"containers": [{"name": "c"}]}}
    p = _run(native, tmp_path, text.encode(), json.dumps(pod).encode(), fmt) if fmt == "json" else None
    if fmt == "yaml":
        (tmp_path / "p.json").write_text(json.dumps(pod))
        (tmp_path / "c.yaml").write_text(text)
        p = subprocess.run([native, "--podspec", str(tmp_path / "p.json"), "--snapshot", str(tmp_path / "c.yaml"), "--dump-snapshot", "-"], capture_output=True, env=ENV, timeout=120)
    assert p.returncode == 1 and b"exceeded max nesting depth" in p.stderr, (p.returncode, p.stderr[-300:])


def test_deep_but_legal_nesting_is_read_and_skipped_members_do_not_count(native, tmp_path):
    pod = {"kind": "Pod", "metadata": {"name": "p"}, "spec": {"containers": [{"name": "c"}]}}
    node = '{"kind":"Node","metadata":{"name":"n","annotations":' + '{"a":' * 1900 + "1" + "}" * 1900 + ',"managedFields":' + '[{"a":' * 100_000 + "1" + "}]" * 100_000 + "}}"
    p = _run(native, tmp_path, node.encode(), json.dumps(pod).encode(), "json")  # (managedFields is pruned: skipped without recursion)
    assert p.returncode == 0, p.stderr[-300:]
    assert json.loads(p.stdout)["names"] == ["n"]


# ---- round 5: the objects the volume plugins read ---------------------------------------------------------------------------------------
def _volume_world():
    from test_volume_ingest import _claim_vol, _class, _csi_pv, _csinode, _nodes, _pod, _pvc
    from test_native_host import running_pod
    nodes = _nodes()
    user = running_pod("user", "n1", cpu="100m")
    user["spec"]["volumes"] = [_claim_vol("other"), {"name": "d", "gcePersistentDisk": {"pdName": "disk-1"}}]
    objs = [user, _class("local"), _pvc("mine", volume_name="pv-1", modes=("ReadWriteOncePod",)), _pvc("other", volume_name="pv-2"), _csi_pv("pv-1", "h1"), _csi_pv("pv-2", "h2"),
            _csinode("n1", 2), {"apiVersion": "storage.k8s.io/v1", "kind": "VolumeAttachment", "metadata": {"name": "va"},
                                "spec": {"attacher": "ebs.csi.aws.com", "nodeName": "n2", "source": {"persistentVolumeName": "pv-2"}}}]
    pod = _pod([_claim_vol("mine"), {"name": "d", "gcePersistentDisk": {"pdName": "disk-1", "readOnly": True}}])
    return nodes, objs, pod


@pytest.mark.parametrize("seed", range(60))
def test_damaged_volume_objects_are_refused_or_read_alike(native, tmp_path, seed):
    """A value of the wrong kind somewhere in a claim / class / volume / CSINode / attachment / a pod's volume list: the native host refuses
    with a message (exit 1) or reads the object exactly as the Python host does -- never crashes (run under the sanitizers too)."""
    rng = random.Random(7100 + seed)
    nodes, objs, pod = _volume_world()
    junk = rng.choice(["text", 7, -1, 3.5, True, None, [], {}, [1, 2], {"a": {"b": []}}, "9" * 40])
    victim = rng.choice([o for o in objs])

    def paths(o, pre=()):
        out = []
        if isinstance(o, dict):
            for k, v in o.items():
                out.append(pre + (k,))
                out += paths(v, pre + (k,))
        elif isinstance(o, list):
            for i, v in enumerate(o):
                out.append(pre + (i,))
                out += paths(v, pre + (i,))
        return out
    where = rng.choice([p for p in paths(victim) if p[0] in ("spec", "metadata", "provisioner", "volumeBindingMode", "status")])
    cur = victim
    for k in where[:-1]:
        cur = cur[k]
    cur[where[-1]] = junk
    (tmp_path / "pod.yaml").write_text(yaml.safe_dump(json.loads(json.dumps(pod))))
    (tmp_path / "cluster.json").write_text(json.dumps({"kind": "List", "items": nodes + objs}))
    flags = ["--podspec", str(tmp_path / "pod.yaml"), "--snapshot", str(tmp_path / "cluster.json"), "--sync-persistent-volumes", "--dump-snapshot", "-"]
    p = subprocess.run([native] + flags, capture_output=True, text=True, timeout=120)
    assert p.returncode in (0, 1), (where, junk, p.returncode, p.stderr[-400:])  # (a sanitizer report or a signal is neither)
    if p.returncode == 1:
        assert p.stderr.strip().startswith("cluster-capacity:"), p.stderr[-300:]
        return
    by = cli.load_by_kind([flags[3]])
    try:
        snap = ingest.build_snapshot(by.get("Node", []), by.get("Pod", []), cli.parse_pod_spec(flags[1]), pvc_objs=by.get("PersistentVolumeClaim", []),
                                     class_objs=by.get("StorageClass", []), pv_objs=by.get("PersistentVolume", []), csinode_objs=by.get("CSINode", []),
                                     attachment_objs=by.get("VolumeAttachment", []))
    except Exception:  # noqa: BLE001  (the Python mirror may stumble over what the native host read as harmless: only agreement on success is claimed)
        return
    got = json.loads(p.stdout)["pod"]
    q = snap.pod
    assert got["volume_veto"] == (None if q.volume_veto is None else [int(x) for x in q.volume_veto]) and got["prefilter_reject"] == q.prefilter_reject, (where, junk)
