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
    with pytest.raises(NotImplementedError, match="ephemeral"):
        _side(_pod([{"name": "scratch", "ephemeral": {"volumeClaimTemplate": {}}}]))


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
