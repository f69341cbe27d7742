"""The volume plugins' object side in both hosts (cluster-capacity_amd/volumes.py, host/volumes.hpp): which objects give which verdict.
Known answers follow the cited lines of the vendored plugins (the reference holds no test that drives them through a cycle).  CPU: the
Python evaluation, the native host's (--dump-snapshot) against it, the messages of the report.  GPU: both CLIs end to end."""
import io
import json
import subprocess

import numpy as np
import pytest
import yaml

from cluster_capacity_amd import cli, ingest, model as M, report as R, schedconfig, volumes as V
from helpers import SUBPROC_TIMEOUT
from test_native_host import EXAMPLES_POD, _run, native, node, running_pod  # noqa: F401  (native: fixture)

ZONE = "topology.kubernetes.io/zone"


def _nodes(k=6):
    return [node(f"n{i}", cpu="4", mem="8Gi", pods="10", labels={ZONE: f"z{i % 3}", "kubernetes.io/hostname": f"n{i}"}) for i in range(k)]


def _pod(volumes, name="sim"):
    p = yaml.safe_load(EXAMPLES_POD)
    p["metadata"]["name"] = name
    p["metadata"]["namespace"] = "default"
    p["spec"]["volumes"] = volumes
    return p


def _pvc(name, volume_name="", cls=None, modes=("ReadWriteOnce",), bound=True, **meta):
    md = {"name": name, "namespace": "default"}
    if volume_name and bound:
        md["annotations"] = {V.ANN_BIND_COMPLETED: "yes"}
    md.update(meta)
    spec = {"accessModes": list(modes)}
    if volume_name:
        spec["volumeName"] = volume_name
    if cls is not None:
        spec["storageClassName"] = cls
    return {"apiVersion": "v1", "kind": "PersistentVolumeClaim", "metadata": md, "spec": spec, "status": {"phase": "Bound" if volume_name else "Pending"}}


def _class(name, mode="WaitForFirstConsumer", provisioner="kubernetes.io/no-provisioner"):
    o = {"apiVersion": "storage.k8s.io/v1", "kind": "StorageClass", "metadata": {"name": name}, "provisioner": provisioner}
    if mode is not None:
        o["volumeBindingMode"] = mode
    return o


def _pv(name, labels=None, terms=None, cls=""):
    spec = {"capacity": {"storage": "1Gi"}, "storageClassName": cls}
    if terms is not None:
        spec["nodeAffinity"] = {"required": {"nodeSelectorTerms": terms}}
    return {"apiVersion": "v1", "kind": "PersistentVolume", "metadata": {"name": name, "labels": labels or {}}, "spec": spec}


def _claim_vol(claim, name="data"):
    return {"name": name, "persistentVolumeClaim": {"claimName": claim}}


def _side(pod, nodes=None, live=(), **kw):
    nodes = nodes or _nodes()
    index = {n["metadata"]["name"]: i for i, n in enumerate(nodes)}
    return V.volume_side(pod, nodes, list(live), index, **kw)


# ---- PreFilter outcomes, as the reference's fake cluster produces them (claims and classes copied, volumes not) ---------------------------
def test_prefilter_rejections_in_plugin_order():
    s = _side(_pod([_claim_vol("ghost")]))
    assert s.prefilter_reject == 'persistentvolumeclaim "ghost" not found'  # VolumeRestrictions.PreFilter (volume_restrictions.go:175-181)
    s = _side(_pod([_claim_vol("c")]), pvc_objs=[_pvc("c", cls="fast")], class_objs=[_class("fast", mode="Immediate")])
    assert s.prefilter_reject == "pod has unbound immediate PersistentVolumeClaims"  # VolumeBinding.PreFilter (volume_binding.go:366-372)
    s = _side(_pod([_claim_vol("c")]), pvc_objs=[_pvc("c")])  # no class at all: immediate too
    assert s.prefilter_reject == "pod has unbound immediate PersistentVolumeClaims"
    s = _side(_pod([_claim_vol("c")]), pvc_objs=[_pvc("c", volume_name="pv-1", bound=False)])  # pre-bound, not yet completed: immediate
    assert s.prefilter_reject == "pod has unbound immediate PersistentVolumeClaims"
    s = _side(_pod([_claim_vol("c")]), pvc_objs=[_pvc("c", volume_name="pv-1")])
    assert s.prefilter_reject == 'persistentvolume "pv-1" not found'  # VolumeZone.PreFilter: the volume is not in the fake cluster
    lost = _pvc("c", volume_name="pv-1")
    lost["status"]["phase"] = "Lost"
    assert _side(_pod([_claim_vol("c")]), pvc_objs=[lost]).prefilter_reject == 'persistentvolumeclaim "c" bound to non-existent persistentvolume "pv-1"'
    gone = _pvc("c", volume_name="pv-1", deletionTimestamp="2024-01-01T00:00:00Z")
    assert _side(_pod([_claim_vol("c")]), pvc_objs=[gone]).prefilter_reject == 'persistentvolumeclaim "c" is being deleted'
    # claims live in the pod's namespace
    other = _pvc("c", volume_name="pv-1", namespace="elsewhere")
    assert _side(_pod([_claim_vol("c")]), pvc_objs=[other]).prefilter_reject == 'persistentvolumeclaim "c" not found'
    # plugins taken out of the profile: the next one in order speaks
    s = _side(_pod([_claim_vol("ghost")]), enabled=("VolumeZone",))
    assert s.prefilter_reject == 'persistentvolumeclaim "ghost" not found'
    s = _side(_pod([_claim_vol("c")]), pvc_objs=[_pvc("c", cls="fast")], class_objs=[_class("fast", mode="Immediate")], enabled=("VolumeZone",))
    assert s.prefilter_reject == "PersistentVolume had no name"  # volume_zone.go:153
    s = _side(_pod([_claim_vol("c")]), pvc_objs=[_pvc("c")], enabled=("VolumeZone",))
    assert s.prefilter_reject == "PersistentVolumeClaim had no pv name and storageClass name"
    s = _side(_pod([_claim_vol("c")]), pvc_objs=[_pvc("c", cls="nope")], enabled=("VolumeZone",))
    assert s.prefilter_reject == 'storageclass.storage.k8s.io "nope" not found'
    # a bound claim without VolumeZone: VolumeBinding.Filter finds the claim bound to a volume that does not exist (binder.go:830-845)
    s = _side(_pod([_claim_vol("c")]), pvc_objs=[_pvc("c", volume_name="pv-1")], enabled=("VolumeRestrictions", "VolumeBinding"))
    assert s.prefilter_reject is None and s.veto.tolist() == [M.VOL_PV_NOT_EXIST] * 6


def test_wait_for_first_consumer_claims():
    pod = _pod([_claim_vol("c")])
    s = _side(pod, pvc_objs=[_pvc("c", cls="local")], class_objs=[_class("local")])
    assert s.prefilter_reject is None and s.veto.tolist() == [M.VOL_NO_PV] * 6  # no volume to bind, nothing provisions one
    with pytest.raises(NotImplementedError, match="PV controller"):
        _side(pod, pvc_objs=[_pvc("c", cls="ebs")], class_objs=[_class("ebs", provisioner="ebs.csi.aws.com")])
    with pytest.raises(NotImplementedError, match="not in the snapshot"):
        _side(pod, pvc_objs=[_pvc("c", cls="ebs")])
    # a generic ephemeral volume: its claim is named after the clone and nobody creates it (volume_binding.go:306-331)
    s = _side(_pod([{"name": "tmp", "emptyDir": {}}, {"name": "scratch", "ephemeral": {"volumeClaimTemplate": {}}}, _claim_vol("ghost")]), enabled=("NodeVolumeLimits", "VolumeBinding", "VolumeZone"))
    assert s.prefilter_reject == 'waiting for ephemeral volume controller to create the persistentvolumeclaim "sim-0-scratch"'
    s = _side(_pod([{"name": "scratch", "ephemeral": {"volumeClaimTemplate": {}}}]), clone_index=2)
    assert s.prefilter_reject == 'waiting for ephemeral volume controller to create the persistentvolumeclaim "sim-2-scratch"'
    with pytest.raises(NotImplementedError, match="created for the simulated pod"):
        _side(_pod([{"name": "scratch", "ephemeral": {"volumeClaimTemplate": {}}}]), pvc_objs=[_pvc("sim-0-scratch")])
    with pytest.raises(NotImplementedError, match="without the VolumeBinding plugin"):
        _side(_pod([{"name": "scratch", "ephemeral": {"volumeClaimTemplate": {}}}]), enabled=("VolumeRestrictions",))


def test_pods_without_volume_plugins_business():
    for vols in ([], [{"name": "tmp", "emptyDir": {}}], [{"name": "cfg", "configMap": {"name": "x"}}], [{"name": "inline", "csi": {"driver": "d"}}]):
        s = _side(_pod(vols))
        assert s.prefilter_reject is None and s.veto is None and not s.exclusive and not s.rwop_capacity_one


# ---- VolumeRestrictions.Filter: the in-tree disks (volume_restrictions.go:105-150) --------------------------------------------------------
def test_disk_conflicts():
    c = V.volume_conflict
    gce = lambda name, ro=False: {"gcePersistentDisk": {"pdName": name, "readOnly": ro}}  # noqa: E731
    assert c(gce("a"), gce("a")) and c(gce("a", True), gce("a")) and not c(gce("a", True), gce("a", True)) and not c(gce("a"), gce("b"))
    ebs = lambda vid, ro=False: {"awsElasticBlockStore": {"volumeID": vid, "readOnly": ro}}  # noqa: E731
    assert c(ebs("v"), ebs("v")) and c(ebs("v", True), ebs("v", True)) and not c(ebs("v"), ebs("w"))  # EBS: read-only does not help
    isc = lambda iqn, ro=False: {"iscsi": {"iqn": iqn, "targetPortal": "p", "lun": 0, "readOnly": ro}}  # noqa: E731
    assert c(isc("q"), isc("q")) and not c(isc("q", True), isc("q", True)) and not c(isc("q"), isc("r"))
    rbd = lambda mons, pool, img, ro=False: {"rbd": {"monitors": mons, "pool": pool, "image": img, "readOnly": ro}}  # noqa: E731
    assert c(rbd(["m1", "m2"], "p", "i"), rbd(["m2"], "p", "i")) and not c(rbd(["m1"], "p", "i"), rbd(["m2"], "p", "i"))
    assert not c(rbd(["m1"], "p", "i"), rbd(["m1"], "q", "i")) and not c(rbd(["m1"], "p", "i", True), rbd(["m1"], "p", "i", True))
    assert not c(gce("a"), ebs("a"))  # different kinds never conflict


def test_disk_conflicts_per_node_and_between_clones():
    nodes = _nodes()
    old = running_pod("old", "n2", cpu="100m")
    old["spec"]["volumes"] = [{"name": "d", "gcePersistentDisk": {"pdName": "disk-1"}}]
    reader = running_pod("reader", "n4", cpu="100m")
    reader["spec"]["volumes"] = [{"name": "d", "gcePersistentDisk": {"pdName": "disk-1", "readOnly": True}}]
    s = _side(_pod([{"name": "d", "gcePersistentDisk": {"pdName": "disk-1", "readOnly": True}}]), nodes, [old, reader])
    assert s.veto.tolist() == [0, 0, M.VOL_DISK_CONFLICT, 0, 0, 0] and not s.exclusive  # read-only next to read-only is fine
    s = _side(_pod([{"name": "d", "gcePersistentDisk": {"pdName": "disk-1"}}]), nodes, [old, reader])
    assert s.veto.tolist() == [0, 0, 1, 0, 1, 0] and s.exclusive
    s = _side(_pod([{"name": "d", "awsElasticBlockStore": {"volumeID": "vol-9", "readOnly": True}}]), nodes, [old])
    assert s.veto is None and s.exclusive
    s = _side(_pod([{"name": "d", "gcePersistentDisk": {"pdName": "disk-1"}}]), nodes, [old], enabled=("VolumeBinding", "VolumeZone"))
    assert s.veto is None and not s.exclusive  # the plugin is out of the profile


def test_read_write_once_pod_claims():
    nodes = _nodes()
    cls = [_class("local")]
    claim = _pvc("solo", cls="local", modes=("ReadWriteOncePod",))
    # (an unbound WaitForFirstConsumer claim of a class without provisioner fails VolumeBinding on every node anyway: code 5 after 2)
    user = running_pod("user", "n1", cpu="100m")
    user["spec"]["volumes"] = [_claim_vol("solo")]
    s = _side(_pod([_claim_vol("solo")]), nodes, [user], pvc_objs=[claim], class_objs=cls)
    assert s.veto.tolist() == [M.VOL_RWOP] * 6 and not s.rwop_capacity_one
    s = _side(_pod([_claim_vol("solo")]), nodes, [], pvc_objs=[claim], class_objs=cls)
    assert s.rwop_capacity_one and s.veto.tolist() == [M.VOL_NO_PV] * 6
    elsewhere = running_pod("user", "n1", cpu="100m")
    elsewhere["metadata"]["namespace"] = "other"
    elsewhere["spec"]["volumes"] = [_claim_vol("solo")]
    assert _side(_pod([_claim_vol("solo")]), nodes, [elsewhere], pvc_objs=[claim], class_objs=cls).rwop_capacity_one  # another namespace's claim


# ---- with the snapshot's PersistentVolumes (--sync-persistent-volumes: beyond the reference) -----------------------------------------------
def test_bound_claims_against_synced_volumes():
    nodes = _nodes()
    claim = _pvc("c", volume_name="pv-1")
    pod = _pod([_claim_vol("c")])
    s = _side(pod, nodes, pvc_objs=[claim], pv_objs=[_pv("pv-1", labels={ZONE: "z1"})])
    assert s.prefilter_reject is None and s.veto.tolist() == [M.VOL_ZONE, 0, M.VOL_ZONE, M.VOL_ZONE, 0, M.VOL_ZONE]
    s = _side(pod, nodes, pvc_objs=[claim], pv_objs=[_pv("pv-1", labels={V.ZONE_BETA: "z0__z2"})])  # beta label, two zones, GA label on the nodes
    assert s.veto.tolist() == [0, M.VOL_ZONE, 0, 0, M.VOL_ZONE, 0]
    bare = [node("m0", cpu="4", mem="8Gi", pods="10"), node("m1", cpu="4", mem="8Gi", pods="10", labels={ZONE: "z9"})]
    s = _side(pod, bare, pvc_objs=[claim], pv_objs=[_pv("pv-1", labels={ZONE: "z1"})])
    assert s.veto.tolist() == [0, M.VOL_ZONE]  # a node without any zone label passes (volume_zone.go:212-224)
    terms = [{"matchExpressions": [{"key": "kubernetes.io/hostname", "operator": "In", "values": ["n3", "n5"]}]}]
    s = _side(pod, nodes, pvc_objs=[claim], pv_objs=[_pv("pv-1", terms=terms)])
    assert s.veto.tolist() == [M.VOL_NODE_AFFINITY] * 3 + [0, M.VOL_NODE_AFFINITY, 0]
    s = _side(pod, nodes, pvc_objs=[claim], pv_objs=[_pv("pv-1", terms=terms, labels={ZONE: "z0"})])  # VolumeBinding speaks before VolumeZone
    assert s.veto.tolist() == [4, 4, 4, 0, 4, M.VOL_ZONE]
    s = _side(pod, nodes, pvc_objs=[claim], pv_objs=[_pv("another")])
    assert s.prefilter_reject == 'persistentvolume "pv-1" not found'
    assert _side(pod, nodes, pvc_objs=[claim], pv_objs=[_pv("pv-1")]).veto is None


def _csi_pv(name, handle, driver="ebs.csi.aws.com"):
    return {"apiVersion": "v1", "kind": "PersistentVolume", "metadata": {"name": name}, "spec": {"csi": {"driver": driver, "volumeHandle": handle}, "storageClassName": ""}}


def _csinode(name, count, driver="ebs.csi.aws.com"):
    d = {"name": driver, "nodeID": name}
    if count is not None:
        d["allocatable"] = {"count": count}
    return {"apiVersion": "storage.k8s.io/v1", "kind": "CSINode", "metadata": {"name": name}, "spec": {"drivers": [d]}}


def _csi_case():
    """6 nodes, a driver limit of 2 volumes on n0..n4 (none declared on n5).  The template mounts claim `mine` (volume h-mine).
    n0: two other volumes attached (pods)            -> over the limit
    n1: one other volume + `mine` itself (a pod uses the same claim) -> nothing NEW to attach: fits
    n2: one volume by a pod, one more by a VolumeAttachment only     -> over
    n3: one volume by a pod that a VolumeAttachment names too        -> counted once: fits
    n4: a pod whose claim is unknown (not counted) + one volume      -> fits
    n5: three volumes, but its CSINode declares no count             -> no limit"""
    nodes = _nodes()
    claims = [_pvc("mine", volume_name="pv-mine")] + [_pvc(f"c{k}", volume_name=f"pv-{k}") for k in range(8)]
    pvs = [_csi_pv("pv-mine", "h-mine")] + [_csi_pv(f"pv-{k}", f"h-{k}") for k in range(8)]
    pods = []

    def user(node_name, *claim_names):
        p = running_pod(f"u{len(pods)}", node_name, cpu="100m")
        p["spec"]["volumes"] = [_claim_vol(c, f"v{j}") for j, c in enumerate(claim_names)]
        pods.append(p)
    user("n0", "c0", "c1")
    user("n1", "c2", "mine")
    user("n2", "c3")
    user("n3", "c4")
    user("n4", "c5", "nowhere")
    user("n5", "c5", "c6", "c7")
    vas = [{"apiVersion": "storage.k8s.io/v1", "kind": "VolumeAttachment", "metadata": {"name": f"va{k}"},
            "spec": {"attacher": "ebs.csi.aws.com", "nodeName": n, "source": {"persistentVolumeName": pv}}} for k, (n, pv) in enumerate([("n2", "pv-6"), ("n3", "pv-4")])]
    csinodes = [_csinode(f"n{i}", 2) for i in range(5)] + [_csinode("n5", None)]
    return nodes, pods, claims, pvs, csinodes, vas


def test_node_volume_limits_against_synced_csinodes():
    nodes, pods, claims, pvs, csinodes, vas = _csi_case()
    pod = _pod([_claim_vol("mine")])
    s = _side(pod, nodes, pods, pvc_objs=claims, pv_objs=pvs, csinode_objs=csinodes, attachment_objs=vas)
    assert s.prefilter_reject is None and s.veto.tolist() == [M.VOL_MAX_COUNT, 0, M.VOL_MAX_COUNT, 0, 0, 0]
    # without the volumes synced (the reference): the bound claim ends at VolumeZone's PreFilter, CSINodes are not even looked at
    assert _side(pod, nodes, pods, pvc_objs=claims, csinode_objs=csinodes).prefilter_reject == 'persistentvolume "pv-mine" not found'
    # the plugin out of the profile
    assert _side(pod, nodes, pods, pvc_objs=claims, pv_objs=pvs, csinode_objs=csinodes, attachment_objs=vas,
                 enabled=("VolumeRestrictions", "VolumeBinding", "VolumeZone")).veto is None
    # an unbound claim counts as one volume of its class's provisioner, named after the claim (csi.go:507-541)
    wait = _pvc("later", cls="ebs")
    s = _side(_pod([_claim_vol("later")]), nodes, pods, pvc_objs=claims + [wait], class_objs=[_class("ebs", mode="WaitForFirstConsumer", provisioner="ebs.csi.aws.com")],
              pv_objs=pvs, csinode_objs=csinodes, attachment_objs=vas, enabled=("VolumeRestrictions", "NodeVolumeLimits"))
    assert s.veto.tolist() == [M.VOL_MAX_COUNT, M.VOL_MAX_COUNT, M.VOL_MAX_COUNT, 0, 0, 0]
    with pytest.raises(NotImplementedError, match="migrated in-tree"):
        _side(_pod([_claim_vol("later")]), nodes, pods, pvc_objs=claims + [wait], class_objs=[_class("ebs", provisioner="kubernetes.io/aws-ebs")],
              pv_objs=pvs, csinode_objs=csinodes, enabled=("NodeVolumeLimits",))


# ---- the report ----------------------------------------------------------------------------------------------------------------------
def test_prefilter_rejection_message():
    r = cli.rejected_by_prefilter(ingest.build_snapshot(_nodes(), [], _pod([])).nodes, M.PodSpec(req=np.zeros(3, np.int64), nz_mcpu=0, nz_mem=0, has_scalar_entries=False,
                                  taint_filter_ok=np.ones(1, np.uint8), taint_prefer_cnt=np.zeros(1, np.int32)), 'persistentvolume "pv-1" not found')
    assert R.stop_reason(r, 6, 0) == ('Unschedulable: 0/6 nodes are available: persistentvolume "pv-1" not found. preemption: 0/6 nodes are available: '
                                      "6 Preemption is not helpful for scheduling.")


def test_scheduler_config_takes_volume_plugins_out():
    prof, _ = schedconfig.profile_from_config(None)
    assert prof.volume_plugins == V.PLUGINS and not prof.volume_plugins_partial
    cfg = {"kind": "KubeSchedulerConfiguration", "profiles": [{"plugins": {"multiPoint": {"disabled": [{"name": "VolumeZone"}, {"name": "NodeVolumeLimits"}]}}}]}
    prof, _ = schedconfig.profile_from_config(cfg)
    assert prof.volume_plugins == ("VolumeRestrictions", "VolumeBinding")
    cfg = {"kind": "KubeSchedulerConfiguration", "profiles": [{"plugins": {"filter": {"disabled": [{"name": "*"}], "enabled": [{"name": "NodeResourcesFit"}]}}}]}
    prof, _ = schedconfig.profile_from_config(cfg)
    assert prof.volume_plugins == () and prof.volume_plugins_partial
    with pytest.raises(NotImplementedError, match="filter point"):
        ingest.build_snapshot(_nodes(), [], _pod([_claim_vol("c")]), volume_plugins=prof.volume_plugins, volume_plugins_partial=True)


def test_snapshot_carries_the_verdicts():
    nodes = _nodes()
    old = running_pod("old", "n2", cpu="100m")
    old["spec"]["volumes"] = [{"name": "d", "awsElasticBlockStore": {"volumeID": "vol-1"}}]
    snap = ingest.build_snapshot(nodes, [old], _pod([{"name": "d", "awsElasticBlockStore": {"volumeID": "vol-1"}}]))
    assert snap.pod.volume_exclusive and snap.pod.volume_veto.tolist() == [0, 0, 1, 0, 0, 0] and snap.pod.prefilter_reject is None
    snap = ingest.build_snapshot(nodes, [], _pod([_claim_vol("c")]), pvc_objs=[_pvc("c", volume_name="pv-1")])
    assert snap.pod.prefilter_reject == 'persistentvolume "pv-1" not found' and snap.pod.volume_veto is None
    with pytest.raises(NotImplementedError, match="same disk"):
        ingest.build_snapshot(nodes, [], [_pod([{"name": "d", "awsElasticBlockStore": {"volumeID": "vol-1"}}], "a"),
                                          _pod([{"name": "e", "awsElasticBlockStore": {"volumeID": "vol-1"}}], "b")])


# ---- the native host: same verdicts from the same objects (--dump-snapshot), same reports ---------------------------------------------
def _write_case(tmp_path, pod, nodes, objs):
    (tmp_path / "pod.yaml").write_text(yaml.safe_dump(json.loads(json.dumps(pod))))
    (tmp_path / "cluster.json").write_text(json.dumps({"kind": "List", "items": nodes + objs}))
    return ["--podspec", str(tmp_path / "pod.yaml"), "--snapshot", str(tmp_path / "cluster.json")]


def _both_sides(native, tmp_path, pod, nodes, objs, extra=()):
    flags = _write_case(tmp_path, pod, nodes, objs) + list(extra)
    got = json.loads(_run(native, flags + ["--dump-snapshot", "-"]))["pod"]
    by = cli.load_by_kind([flags[3]])
    snap = ingest.build_snapshot(by.get("Node", []), by.get("Pod", []), cli.parse_pod_spec(flags[1]), pvc_objs=by.get("PersistentVolumeClaim", []),
                                 class_objs=by.get("StorageClass", []), pv_objs=by.get("PersistentVolume", []) if "--sync-persistent-volumes" in extra else None,
                                 csinode_objs=by.get("CSINode", []) if "--sync-persistent-volumes" in extra else (),
                                 attachment_objs=by.get("VolumeAttachment", []) if "--sync-persistent-volumes" in extra else ())
    p = snap.pod
    ref = {"volume_veto": None if p.volume_veto is None else [int(x) for x in p.volume_veto], "volume_exclusive": bool(p.volume_exclusive),
           "prefilter_reject": p.prefilter_reject, "rwop_capacity_one": bool(p.rwop_capacity_one)}
    assert {k: got[k] for k in ref} == ref
    return ref


def test_native_host_gives_the_same_verdicts(native, tmp_path):
    nodes = _nodes()
    old = running_pod("old", "n2", cpu="100m")
    old["spec"]["volumes"] = [{"name": "d", "gcePersistentDisk": {"pdName": "disk-1"}}, _claim_vol("solo")]
    ro = running_pod("ro", "n4", cpu="100m")
    ro["spec"]["volumes"] = [{"name": "d", "rbd": {"monitors": ["m1", "m2"], "pool": "p", "image": "i", "readOnly": True}}]
    classes = [_class("local"), _class("fast", mode="Immediate"), _class("ebs", provisioner="ebs.csi.aws.com")]
    claims = [_pvc("solo", cls="local", modes=("ReadWriteOncePod",)), _pvc("free", cls="local", modes=("ReadWriteOncePod",)), _pvc("b", volume_name="pv-1"),
              _pvc("imm", cls="fast"), _pvc("old-style", **{"annotations": {V.ANN_BETA_STORAGE_CLASS: "local"}})]
    terms = [{"matchExpressions": [{"key": "kubernetes.io/hostname", "operator": "In", "values": ["n3", "n5"]}]}]
    pvs = [_pv("pv-1", labels={V.ZONE_BETA: "z0__z2"}, terms=terms)]
    objs = [old, ro] + classes + claims + pvs
    cases = [
        ([{"name": "d", "gcePersistentDisk": {"pdName": "disk-1"}}], ()),
        ([{"name": "d", "gcePersistentDisk": {"pdName": "disk-1", "readOnly": True}}], ()),
        ([{"name": "d", "rbd": {"monitors": ["m2"], "pool": "p", "image": "i"}}, {"name": "e", "awsElasticBlockStore": {"volumeID": "v"}}], ()),
        ([_claim_vol("solo")], ()), ([_claim_vol("free")], ()), ([_claim_vol("ghost")], ()), ([_claim_vol("imm")], ()), ([_claim_vol("old-style")], ()),
        ([_claim_vol("b")], ()), ([_claim_vol("b")], ("--sync-persistent-volumes",)),
        ([{"name": "tmp", "emptyDir": {}}, {"name": "inline", "csi": {"driver": "d"}}], ()),
        ([{"name": "scratch", "ephemeral": {"volumeClaimTemplate": {}}}], ()),
    ]
    seen = []
    for k, (vols, extra) in enumerate(cases):
        d = tmp_path / f"c{k}"
        d.mkdir()
        seen.append(_both_sides(native, d, _pod(vols), nodes, objs, extra))
    assert seen[0]["volume_veto"] == [0, 0, 1, 0, 0, 0] and seen[0]["volume_exclusive"]
    assert seen[2]["volume_veto"] == [0, 0, 0, 0, 1, 0] and seen[2]["volume_exclusive"]
    assert seen[3]["volume_veto"] == [M.VOL_RWOP] * 6 and seen[4]["rwop_capacity_one"] and seen[4]["volume_veto"] == [M.VOL_NO_PV] * 6
    assert seen[5]["prefilter_reject"] == 'persistentvolumeclaim "ghost" not found' and seen[6]["prefilter_reject"] == "pod has unbound immediate PersistentVolumeClaims"
    assert seen[7]["volume_veto"] == [M.VOL_NO_PV] * 6  # the class named by the beta annotation
    assert seen[8]["prefilter_reject"] == 'persistentvolume "pv-1" not found'
    assert seen[9]["volume_veto"] == [4, 4, 4, 0, 4, 0] and seen[9]["prefilter_reject"] is None  # n3 (z0) and n5 (z2) carry the volume's node affinity and zones
    assert seen[10] == {"volume_veto": None, "volume_exclusive": False, "prefilter_reject": None, "rwop_capacity_one": False}
    assert seen[11]["prefilter_reject"] == 'waiting for ephemeral volume controller to create the persistentvolumeclaim "sim-0-scratch"'


def test_native_host_counts_csi_volumes_like_the_python_host(native, tmp_path):
    nodes, pods, claims, pvs, csinodes, vas = _csi_case()
    objs = pods + claims + pvs + csinodes + vas + [_pvc("later", cls="ebs"), _class("ebs", mode="WaitForFirstConsumer", provisioner="ebs.csi.aws.com")]
    a = tmp_path / "a"
    a.mkdir()
    got = _both_sides(native, a, _pod([_claim_vol("mine")]), nodes, objs, ("--sync-persistent-volumes",))
    assert got["volume_veto"] == [M.VOL_MAX_COUNT, 0, M.VOL_MAX_COUNT, 0, 0, 0]
    b = tmp_path / "b"
    b.mkdir()
    assert _both_sides(native, b, _pod([_claim_vol("mine")]), nodes, objs)["prefilter_reject"] == 'persistentvolume "pv-mine" not found'
    # two claims of the template: on n1 `mine` is attached already (one NEW volume, but the node sits at its limit), two new ones elsewhere
    c = tmp_path / "c"
    c.mkdir()
    got = _both_sides(native, c, _pod([_claim_vol("mine"), _claim_vol("c7", "second")]), nodes, objs, ("--sync-persistent-volumes",))
    assert got["volume_veto"] == [3, 3, 3, 3, 3, 0]


def test_native_host_refuses_what_the_python_host_refuses(native, tmp_path):
    nodes = _nodes()
    for vols, objs, what in (([_claim_vol("c")], [_pvc("c", cls="ebs"), _class("ebs", provisioner="ebs.csi.aws.com")], "PV controller"),
                             ([_claim_vol("c")], [_pvc("c", cls="nowhere")], "not in the snapshot"),
                             ([{"name": "s", "ephemeral": {"volumeClaimTemplate": {}}}], [_pvc("sim-0-s")], "created for the simulated pod")):
        flags = _write_case(tmp_path, _pod(vols), nodes, objs)
        p = subprocess.run([native] + flags, capture_output=True, text=True, timeout=SUBPROC_TIMEOUT)
        assert p.returncode == 1 and what in p.stderr
        with pytest.raises(NotImplementedError, match=what):
            cli.main(flags, out=io.StringIO())


def test_both_hosts_report_a_prefilter_rejection_without_a_device(native, tmp_path):
    """Zero replicas, the plugin's message, "Preemption is not helpful" on every node: no scheduling cycle reaches the engine."""
    nodes = _nodes()
    flags = _write_case(tmp_path, _pod([_claim_vol("data")]), nodes, [_pvc("data", volume_name="pv-7")])
    want = ('0/6 nodes are available: persistentvolume "pv-7" not found. preemption: 0/6 nodes are available: 6 Preemption is not helpful for scheduling.')
    got = json.loads(_run(native, flags + ["-o", "json"]))
    buf = io.StringIO()
    assert cli.main(flags + ["-o", "json"], out=buf) == 0
    ref = json.loads(buf.getvalue())
    got["status"].pop("creationTimestamp"), ref["status"].pop("creationTimestamp")
    assert got["status"] == ref["status"]
    assert got["status"]["replicas"] == 0 and got["status"]["failReason"] == {"failType": "Unschedulable", "failMessage": want}
    txt = _run(native, flags + ["--verbose"])
    assert "Termination reason: Unschedulable: " + want in txt


def test_resource_claims_end_at_the_dra_prefilter_in_both_hosts(native, tmp_path):
    """DynamicResources.PreFilter in the reference's fake cluster (no ResourceClaim is copied): the first claim it looks up is missing."""
    nodes = _nodes()
    cases = [([{"name": "gpu", "resourceClaimName": "shared-gpu"}], 'could not find ResourceClaim "default/shared-gpu"'),
             ([{"name": "gpu", "resourceClaimTemplateName": "gpu-template"}], 'pod "default/sim-0": ResourceClaim not created yet'),
             ([{"name": "gpu"}], 'pod "default/sim-0", spec.resourceClaim "gpu": none of the supported fields are set')]
    for k, (claims, msg) in enumerate(cases):
        pod = _pod([])
        pod["spec"]["resourceClaims"] = claims
        d = tmp_path / f"c{k}"
        d.mkdir()
        assert _both_sides(native, d, pod, nodes, [])["prefilter_reject"] == msg
    # the volume plugins' PreFilters run first (default_plugins.go:41-47)
    pod = _pod([_claim_vol("ghost")])
    pod["spec"]["resourceClaims"] = cases[0][0]
    d = tmp_path / "both"
    d.mkdir()
    assert _both_sides(native, d, pod, nodes, [])["prefilter_reject"] == 'persistentvolumeclaim "ghost" not found'
    # end to end, no device: zero replicas with the message
    pod = _pod([])
    pod["spec"]["resourceClaims"] = cases[0][0]
    d = tmp_path / "e2e"
    d.mkdir()
    flags = _write_case(d, pod, nodes, [])
    got = json.loads(_run(native, flags + ["-o", "json"]))["status"]
    buf = io.StringIO()
    assert cli.main(flags + ["-o", "json"], out=buf) == 0
    ref = json.loads(buf.getvalue())["status"]
    got.pop("creationTimestamp"), ref.pop("creationTimestamp")
    assert got == ref and got["replicas"] == 0
    assert got["failReason"]["failMessage"].startswith('0/6 nodes are available: could not find ResourceClaim "default/shared-gpu". preemption: 0/6 nodes are available: 6 Preemption is not helpful')
    # the plugin taken out of the profile (multiPoint): the claims are nobody's business
    prof, _ = schedconfig.profile_from_config({"kind": "KubeSchedulerConfiguration", "profiles": [{"plugins": {"multiPoint": {"disabled": [{"name": "DynamicResources"}]}}}]})
    assert not prof.dra_enabled and ingest.build_snapshot(nodes, [], pod, dra_enabled=prof.dra_enabled).pod.prefilter_reject is None
    cfg = d / "sched.yaml"
    cfg.write_text(yaml.safe_dump({"apiVersion": "kubescheduler.config.k8s.io/v1", "kind": "KubeSchedulerConfiguration",
                                   "profiles": [{"plugins": {"multiPoint": {"disabled": [{"name": "DynamicResources"}]}}}]}))
    dump = json.loads(_run(native, flags + ["--default-config", str(cfg), "--dump-snapshot", "-"]))
    assert dump["pod"]["prefilter_reject"] is None


# ---- GPU: both CLIs end to end ------------------------------------------------------------------------------------------------------
def _status(native, flags):
    got = json.loads(_run(native, flags + ["-o", "json"]))
    buf = io.StringIO()
    assert cli.main(flags + ["-o", "json"], out=buf) == 0
    ref = json.loads(buf.getvalue())
    got["status"].pop("creationTimestamp"), ref["status"].pop("creationTimestamp")
    assert got["status"] == ref["status"]
    return got["status"]


@pytest.mark.gpu
def test_gpu_cli_disks_both_hosts_vs_oracle(ccref, native, tmp_path):
    """A read-write GCE PD: nodes whose pods mount it are out, every other node takes ONE clone; the terminal message names the disk
    conflict for the nodes that still had room."""
    nodes = _nodes(9)
    users = []
    for k, n in enumerate(("n2", "n5")):
        u = running_pod(f"user-{k}", n, cpu="100m")
        u["spec"]["volumes"] = [{"name": "d", "gcePersistentDisk": {"pdName": "disk-1", "readOnly": True}}]
        users.append(u)
    pod = _pod([{"name": "d", "gcePersistentDisk": {"pdName": "disk-1"}}])
    flags = _write_case(tmp_path, pod, nodes, users)
    st = _status(native, flags)
    snap = ingest.build_snapshot(nodes, users, cli.parse_pod_spec(flags[1]))
    r = ccref.run(M.Profile.default(), snap.nodes, snap.pod)
    assert st["replicas"] == r.placed == 7
    assert [x["nodeName"] for x in st["pods"][0]["replicasOnNodes"]] == [snap.names[i] for i in r.log]
    assert st["failReason"]["failMessage"] == R.stop_reason(r, 9, 0)[len("Unschedulable: "):]
    assert "9 node(s) had no available disk" in st["failReason"]["failMessage"]
    st = _status(native, flags + ["--max-limit", "3"])
    assert st["replicas"] == 3 and st["failReason"]["failType"] == "LimitReached"


@pytest.mark.gpu
def test_gpu_cli_read_write_once_pod_claim_has_capacity_one(native, tmp_path):
    nodes = _nodes()
    classes = [_class("local")]
    # (bound claims end at VolumeZone's PreFilter in the reference's fake cluster; with the volumes synced the claim is usable)
    claim = _pvc("solo", volume_name="pv-1", modes=("ReadWriteOncePod",))
    objs = classes + [claim, _pv("pv-1", labels={ZONE: "z1"})]
    flags = _write_case(tmp_path, _pod([_claim_vol("solo")]), nodes, objs) + ["--sync-persistent-volumes"]
    st = _status(native, flags)
    assert st["replicas"] == 1 and st["pods"][0]["replicasOnNodes"][0]["nodeName"] in ("n1", "n4")
    # VolumeRestrictions speaks before VolumeZone: all six nodes report the claim in use, none the zone
    assert "6 node(s) unavailable due to PersistentVolumeClaim with ReadWriteOncePod" in st["failReason"]["failMessage"]
    assert "volume zone" not in st["failReason"]["failMessage"]
    st = _status(native, flags + ["--max-limit", "1"])
    assert st["replicas"] == 1 and st["failReason"]["failType"] == "LimitReached"
    user = running_pod("user", "n0", cpu="100m")
    user["spec"]["volumes"] = [_claim_vol("solo")]
    d = tmp_path / "used"
    d.mkdir()
    st = _status(native, _write_case(d, _pod([_claim_vol("solo")]), nodes, objs + [user]) + ["--sync-persistent-volumes"])
    assert st["replicas"] == 0 and "6 node(s) unavailable due to PersistentVolumeClaim with ReadWriteOncePod" in st["failReason"]["failMessage"]


@pytest.mark.gpu
def test_gpu_cli_several_templates_with_volumes_vs_oracle(ccref, native, tmp_path):
    """Templates with volumes are outside the window engine's shape (ccsim_set_pods: -ENOSYS): one cycle at a time, the clones' own disks
    folded into the verdicts the template is set with."""
    nodes = _nodes(8)
    a = _pod([{"name": "d", "awsElasticBlockStore": {"volumeID": "vol-a"}}], "tmpl-a")
    b = _pod([], "tmpl-b")
    for t, p in enumerate((a, b)):
        p["metadata"]["labels"] = {"app": f"t{t}"}
        (tmp_path / f"t{t}.yaml").write_text(yaml.safe_dump(json.loads(json.dumps(p))))
    (tmp_path / "cluster.json").write_text(json.dumps({"kind": "List", "items": nodes}))
    flags = ["--podspec", str(tmp_path / "t0.yaml"), "--podspec", str(tmp_path / "t1.yaml"), "--snapshot", str(tmp_path / "cluster.json")]
    st = _status(native, flags)
    snap = ingest.build_snapshot(nodes, [], [cli.parse_pod_spec(flags[1]), cli.parse_pod_spec(flags[3])])
    r = ccref.run_multi(M.Profile.default(), snap.nodes, snap.pods)
    assert st["replicas"] == r.placed and r.per_spec_count.tolist() == [8, 8] and r.stop_spec == 0
    assert [sum(x["replicas"] for x in q["replicasOnNodes"]) for q in st["pods"]] == r.per_spec_count.tolist()
    assert "8 node(s) had no available disk" in st["failReason"]["failMessage"]
    # a template a PreFilter plugin rejects ends the run at ITS first cycle: template 0's first clone is placed, nothing more
    c = _pod([_claim_vol("ghost")], "tmpl-c")
    (tmp_path / "t1.yaml").write_text(yaml.safe_dump(json.loads(json.dumps(c))))
    st = _status(native, flags)
    assert st["replicas"] == 1 and 'persistentvolumeclaim "ghost" not found' in st["failReason"]["failMessage"]


def test_verdicts_with_a_nodes_victims_gone_in_both_hosts(native, tmp_path):
    """DefaultPreemption's dry run removes the lower-priority pods of ONE node and filters again: a disk conflict leaves with the victim that
    holds the disk; a ReadWriteOncePod claim stays in use on every node but the one whose victims are ALL of its users."""
    nodes = _nodes()
    writer = running_pod("writer", "n1", cpu="100m")           # lower priority: a victim; holds the template's disk
    writer["spec"]["volumes"] = [{"name": "d", "gcePersistentDisk": {"pdName": "disk-1"}}]
    keeper = running_pod("keeper", "n2", cpu="100m")            # same priority as the template: stays; holds the disk too
    keeper["spec"]["volumes"] = [{"name": "d", "gcePersistentDisk": {"pdName": "disk-1"}}]
    keeper["spec"]["priority"] = 5
    user = running_pod("user", "n4", cpu="100m")                # a victim that uses the template's ReadWriteOncePod claim
    user["spec"]["volumes"] = [_claim_vol("solo")]
    pod = _pod([{"name": "d", "gcePersistentDisk": {"pdName": "disk-1", "readOnly": True}}])
    pod["spec"]["priority"] = 5
    a = tmp_path / "a"
    a.mkdir()
    flags = _write_case(a, pod, nodes, [writer, keeper, user])
    got = json.loads(_run(native, flags + ["--dump-snapshot", "-"]))["pod"]
    snap = ingest.build_snapshot(nodes, [writer, keeper, user], cli.parse_pod_spec(flags[1]))
    assert snap.pod.volume_veto.tolist() == [0, 1, 1, 0, 0, 0] and snap.pod.preempt.volume_veto_rest.tolist() == [0, 0, 1, 0, 0, 0]
    assert got["volume_veto"] == [0, 1, 1, 0, 0, 0] and got["preempt"]["volume_veto_rest"] == [0, 0, 1, 0, 0, 0]
    # the ReadWriteOncePod claim: in use (by `user` on n4) -> every node fails now; with a node's victims gone only n4 is free of it
    pod = _pod([_claim_vol("solo")])
    pod["spec"]["priority"] = 5
    objs = [writer, keeper, user, _class("local"), _pvc("solo", volume_name="pv-1", modes=("ReadWriteOncePod",)), _pv("pv-1")]
    b = tmp_path / "b"
    b.mkdir()
    flags = _write_case(b, pod, nodes, objs) + ["--sync-persistent-volumes"]
    got = json.loads(_run(native, flags + ["--dump-snapshot", "-"]))["pod"]
    by = cli.load_by_kind([flags[3]])
    snap = ingest.build_snapshot(by["Node"], by["Pod"], cli.parse_pod_spec(flags[1]), pvc_objs=by["PersistentVolumeClaim"], class_objs=by["StorageClass"], pv_objs=by["PersistentVolume"])
    assert snap.pod.volume_veto.tolist() == [2] * 6 and snap.pod.preempt.volume_veto_rest.tolist() == [2, 2, 2, 2, 0, 2]
    assert got["volume_veto"] == [2] * 6 and got["preempt"]["volume_veto_rest"] == [2, 2, 2, 2, 0, 2]


def test_read_write_once_pod_count_as_the_reference_keeps_it(native, tmp_path):
    """the dry run's ReadWriteOncePod state is a COUNT (volume_restrictions.go:70-84, 219-232): one reference per claim in use however
    many pods use it, minus one per victim volume whose claimName matches -- by name only.  So with two users of the claim (one a victim on n1,
    one staying on n2) node n1 passes the dry run, and a victim of ANOTHER namespace with a same-named claim frees its node too."""
    nodes = _nodes()
    u1 = running_pod("u1", "n1", cpu="100m")   # lower priority: a victim
    u1["spec"]["volumes"] = [_claim_vol("solo")]
    u2 = running_pod("u2", "n2", cpu="100m")   # stays (same priority as the template)
    u2["spec"]["volumes"] = [_claim_vol("solo")]
    u2["spec"]["priority"] = 5
    other = running_pod("other", "n3", cpu="100m")  # a victim in another namespace whose own claim has the same name
    other["metadata"]["namespace"] = "elsewhere"
    other["spec"]["volumes"] = [_claim_vol("solo")]
    pod = _pod([_claim_vol("solo")])
    pod["spec"]["priority"] = 5
    objs = [u1, u2, other, _class("local"), _pvc("solo", volume_name="pv-1", modes=("ReadWriteOncePod",)), _pv("pv-1")]
    flags = _write_case(tmp_path, pod, nodes, objs) + ["--sync-persistent-volumes"]
    got = json.loads(_run(native, flags + ["--dump-snapshot", "-"]))["pod"]
    by = cli.load_by_kind([flags[3]])
    snap = ingest.build_snapshot(by["Node"], by["Pod"], cli.parse_pod_spec(flags[1]), pvc_objs=by["PersistentVolumeClaim"], class_objs=by["StorageClass"], pv_objs=by["PersistentVolume"])
    want = [2, 0, 2, 0, 2, 2]  # count 1; n1's victim and n3's victim each take one reference away, n2's user is no victim
    assert snap.pod.volume_veto.tolist() == [2] * 6 and snap.pod.preempt.volume_veto_rest.tolist() == want
    assert got["volume_veto"] == [2] * 6 and got["preempt"]["volume_veto_rest"] == want


def test_an_empty_cluster_ends_before_any_prefilter_in_both_hosts(native, tmp_path):
    """schedulePod returns ErrNoNodesAvailable before a PreFilter plugin runs (schedule_one.go:438-440): zero nodes and a pod whose claim
    is missing is `no nodes available to schedule pods`, not the claim's message."""
    pod = _pod([_claim_vol("missing")])
    flags = _write_case(tmp_path, pod, [], [])
    buf = io.StringIO()
    assert cli.main(flags + ["-o", "json"], out=buf) == 0
    ref = json.loads(buf.getvalue())["status"]
    got = json.loads(_run(native, flags + ["-o", "json"]))["status"]
    got.pop("creationTimestamp"), ref.pop("creationTimestamp")
    assert got == ref and got["replicas"] == 0 and got["failReason"]["failMessage"] == "no nodes available to schedule pods", got


# ---- random differential: the two hosts on random object graphs -----------------------------------------------------------------------
def _random_volume_world(rng):
    n = int(rng.integers(3, 12))
    nodes = []
    for i in range(n):
        labels = {"kubernetes.io/hostname": f"n{i}"}
        if rng.random() < 0.8:
            labels[ZONE if rng.random() < 0.7 else V.ZONE_BETA] = f"z{int(rng.integers(0, 3))}"
        if rng.random() < 0.3:
            labels[V.REGION_GA] = f"r{int(rng.integers(0, 2))}"
        nodes.append(node(f"n{i}", cpu="4", mem="8Gi", pods="10", labels=labels))
    drivers = ["ebs.csi.aws.com", "pd.csi.storage.gke.io"]
    classes = [_class("local"), _class("fast", mode="Immediate", provisioner=drivers[0]), _class("csi-wait", provisioner=drivers[1]),
               _class("nomode", mode=None), _class("intree", provisioner="kubernetes.io/aws-ebs")]
    pvs, claims = [], []
    for k in range(int(rng.integers(2, 9))):
        labels = {}
        if rng.random() < 0.4:
            labels[ZONE if rng.random() < 0.6 else V.ZONE_BETA] = "__".join(f"z{z}" for z in sorted(set(rng.integers(0, 3, int(rng.integers(1, 3))).tolist())))
        terms = None
        if rng.random() < 0.3:
            terms = [{"matchExpressions": [{"key": "kubernetes.io/hostname", "operator": str(rng.choice(["In", "NotIn"])), "values": [f"n{int(x)}" for x in rng.integers(0, n, 2)]}]}]
        pv = _csi_pv(f"pv-{k}", f"h-{k}", drivers[int(rng.integers(0, 2))]) if rng.random() < 0.7 else _pv(f"pv-{k}")
        pv["metadata"]["labels"] = labels
        if terms is not None:
            pv["spec"]["nodeAffinity"] = {"required": {"nodeSelectorTerms": terms}}
        if rng.random() < 0.1:
            pv["spec"].pop("csi", None)
            pv["spec"]["gcePersistentDisk"] = {"pdName": f"pd-{k}"}
        pvs.append(pv)
    for k in range(int(rng.integers(2, 10))):
        kind = rng.random()
        modes = ("ReadWriteOncePod",) if rng.random() < 0.25 else ("ReadWriteOnce",)
        if kind < 0.45:
            c = _pvc(f"c{k}", volume_name=f"pv-{int(rng.integers(0, len(pvs) + 1))}", modes=modes, bound=bool(rng.random() < 0.85))
        else:
            c = _pvc(f"c{k}", cls=str(rng.choice(["local", "fast", "csi-wait", "nomode", "intree", "gone"])) if rng.random() < 0.85 else None, modes=modes)
        if rng.random() < 0.07:
            c["status"]["phase"] = "Lost"
        if rng.random() < 0.05:
            c["metadata"]["deletionTimestamp"] = "2025-01-01T00:00:00Z"
        if rng.random() < 0.1:
            c["metadata"].setdefault("annotations", {})[V.ANN_BETA_STORAGE_CLASS] = "local"
        claims.append(c)

    def random_volumes(k_max, own):
        vols = []
        for j in range(int(rng.integers(0, k_max + 1))):
            r = rng.random()
            name = f"v{j}"
            if r < 0.45:
                vols.append(_claim_vol(f"c{int(rng.integers(0, len(claims) + (1 if own else 0)))}", name))
            elif r < 0.6:
                vols.append({"name": name, "gcePersistentDisk": {"pdName": f"disk-{int(rng.integers(0, 2))}", "readOnly": bool(rng.random() < 0.5)}})
            elif r < 0.7:
                vols.append({"name": name, "awsElasticBlockStore": {"volumeID": f"vol-{int(rng.integers(0, 2))}"}})
            elif r < 0.78:
                vols.append({"name": name, "rbd": {"monitors": [f"m{int(x)}" for x in rng.integers(0, 3, 2)], "pool": "p", "image": f"i{int(rng.integers(0, 2))}", "readOnly": bool(rng.random() < 0.5)}})
            elif r < 0.85:
                vols.append({"name": name, "iscsi": {"iqn": f"iqn-{int(rng.integers(0, 2))}", "targetPortal": "p", "lun": 0, "readOnly": bool(rng.random() < 0.5)}})
            elif r < 0.9 and own:
                vols.append({"name": name, "ephemeral": {"volumeClaimTemplate": {}}})
            else:
                vols.append({"name": name, "emptyDir": {}})
        return vols
    pods = []
    for k in range(int(rng.integers(0, 10))):
        p = running_pod(f"p{k}", f"n{int(rng.integers(0, n))}", cpu="100m")
        p["spec"]["volumes"] = random_volumes(3, False)
        if rng.random() < 0.2:
            p["metadata"]["namespace"] = "other"
        pods.append(p)
    csinodes = [_csinode(f"n{i}", int(rng.integers(1, 4)) if rng.random() < 0.8 else None, drivers[int(rng.integers(0, 2))]) for i in range(n) if rng.random() < 0.7]
    vas = [{"apiVersion": "storage.k8s.io/v1", "kind": "VolumeAttachment", "metadata": {"name": f"va{k}"},
            "spec": {"attacher": drivers[int(rng.integers(0, 2))], "nodeName": f"n{int(rng.integers(0, n))}", "source": {"persistentVolumeName": f"pv-{int(rng.integers(0, len(pvs)))}"}}}
           for k in range(int(rng.integers(0, 5)))]
    template = _pod(random_volumes(3, True))
    if rng.random() < 0.6:  # lower-priority pods around: DefaultPreemption's dry run needs the verdicts with a node's victims gone
        template["spec"]["priority"] = 5
        for p in pods:
            if rng.random() < 0.4:
                p["spec"]["priority"] = 7
    if rng.random() < 0.15:
        template["spec"]["resourceClaims"] = [{"name": "dev", **({"resourceClaimName": "gpu"} if rng.random() < 0.5 else {"resourceClaimTemplateName": "tpl"})}]
    return nodes, pods + classes + claims + pvs + csinodes + vas, template


@pytest.mark.parametrize("seed", range(120))
def test_both_hosts_agree_on_random_volume_object_graphs(native, tmp_path, seed):
    """Random claims / classes / volumes / CSINodes / attachments / pods with volumes, with and without --sync-persistent-volumes: the native
    host's verdicts == the Python host's; what one refuses the other refuses."""
    rng = np.random.default_rng(52_000 + seed)
    nodes, objs, template = _random_volume_world(rng)
    extra = ("--sync-persistent-volumes",) if seed % 2 else ()
    flags = _write_case(tmp_path, template, nodes, objs) + list(extra)
    by = cli.load_by_kind([flags[3]])
    try:
        snap = ingest.build_snapshot(by.get("Node", []), by.get("Pod", []), cli.parse_pod_spec(flags[1]), pvc_objs=by.get("PersistentVolumeClaim", []),
                                     class_objs=by.get("StorageClass", []), pv_objs=by.get("PersistentVolume", []) if extra else None,
                                     csinode_objs=by.get("CSINode", []) if extra else (), attachment_objs=by.get("VolumeAttachment", []) if extra else ())
    except NotImplementedError as e:
        p = subprocess.run([native] + flags + ["--dump-snapshot", "-"], capture_output=True, text=True, timeout=SUBPROC_TIMEOUT)
        assert p.returncode == 1, (str(e), p.stdout[-300:])
        return
    got = json.loads(_run(native, flags + ["--dump-snapshot", "-"]))["pod"]
    p = snap.pod
    ref = {"volume_veto": None if p.volume_veto is None else [int(x) for x in p.volume_veto], "volume_exclusive": bool(p.volume_exclusive),
           "prefilter_reject": p.prefilter_reject, "rwop_capacity_one": bool(p.rwop_capacity_one)}
    assert {k: got[k] for k in ref} == ref
    rest = p.preempt.volume_veto_rest
    assert got["preempt"]["volume_veto_rest"] == (None if rest is None else [int(x) for x in rest])
