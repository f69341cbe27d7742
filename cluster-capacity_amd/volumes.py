"""The volume plugins' object side: VolumeRestrictions, NodeVolumeLimits, VolumeBinding, VolumeZone.

What they decide is string and object-graph work that does not change while clones are placed -- except a clone's own disks and a
ReadWriteOncePod claim -- so it is evaluated here, once per template, into what the engine takes (include/ccsim.h ccsim_pod):
`volume_veto[n]` (the code of the first of the four that rejects node n against the snapshot's pods) and `volume_exclusive`.  The
PreFilter outcomes that reject the pod on every node (zero replicas, with the plugin's message as FitError.Diagnosis.PreFilterMsg) never
reach the engine.  Python mirror of host/volumes.hpp.

What the REFERENCE's scheduler sees: SyncWithClient copies PersistentVolumeClaims and StorageClasses into the fake cluster but NOT
PersistentVolumes, CSINodes, CSIDrivers or CSIStorageCapacities (pkg/framework/simulator.go:228-295).  Hence, with the default plugins:
  * a claim that does not exist                    -> VolumeRestrictions.PreFilter: `persistentvolumeclaim "x" not found`
                                                      (volumerestrictions/volume_restrictions.go:175-181)
  * a lost / terminating claim                     -> VolumeBinding.PreFilter (volumebinding/volume_binding.go:333-339, 356-357)
  * a generic ephemeral volume                     -> VolumeBinding.PreFilter: its claim "<clone>-<volume>" is never created (:306-331)
  * an unbound claim of an Immediate class         -> VolumeBinding.PreFilter: "pod has unbound immediate PersistentVolumeClaims" (:366-372)
  * a BOUND claim                                  -> VolumeZone.PreFilter: `persistentvolume "pv" not found` (volumezone/volume_zone.go:156-159,
                                                      253-258) -- the volume is not in the fake cluster
  * an unbound WaitForFirstConsumer claim          -> passes every PreFilter; VolumeBinding.Filter finds no volume to bind
                                                      (binder.go checkVolumeProvisions): a class without a provisioner fails every node with
                                                      "node(s) didn't find available persistent volumes to bind"; a class WITH one passes, and
                                                      the pod then waits in PreBind for a PV controller the fake cluster does not run -- the
                                                      reference hangs; refused here with that reason
  * NodeVolumeLimits                               -> never rejects: no CSINode, no limits (nodevolumelimits/csi.go:265-290)
  * GCE PD / EBS / RBD / ISCSI volumes             -> VolumeRestrictions.Filter against the node's pods and the clones (:105-150, 310-313)
  * a ReadWriteOncePod claim                       -> in use by a pod of the snapshot: every node fails (:283-291); else the first clone
                                                      takes it and the second cycle fails everywhere: capacity 1
`sync_persistent_volumes` (the hosts' --sync-persistent-volumes) goes one step beyond the reference: PersistentVolume, CSINode and
VolumeAttachment objects of the snapshot are taken too, so bound claims are judged as kube-scheduler judges them on the live cluster --
VolumeBinding's node affinity of the bound volume (binder.go checkBoundClaims), VolumeZone's label match (volume_zone.go:191-240) and
NodeVolumeLimits' per-driver counts (nodevolumelimits/csi.go:255-339) -- as static per-node verdicts."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import model as M

PLUGINS = ("VolumeRestrictions", "NodeVolumeLimits", "VolumeBinding", "VolumeZone")
ANN_BIND_COMPLETED = "pv.kubernetes.io/bind-completed"  # volume.AnnBindCompleted
ANN_BETA_STORAGE_CLASS = "volume.beta.kubernetes.io/storage-class"  # v1.BetaStorageClassAnnotation
NO_PROVISIONER = "kubernetes.io/no-provisioner"  # volume.NotSupportedProvisioner
ZONE_BETA, REGION_BETA = "failure-domain.beta.kubernetes.io/zone", "failure-domain.beta.kubernetes.io/region"
ZONE_GA, REGION_GA = "topology.kubernetes.io/zone", "topology.kubernetes.io/region"
TOPOLOGY_LABELS = (ZONE_BETA, REGION_BETA, ZONE_GA, REGION_GA)  # volume_zone.go:84-89
RESTRICTED_KINDS = ("gcePersistentDisk", "awsElasticBlockStore", "rbd", "iscsi")  # needsRestrictionsCheck (volume_restrictions.go:152-155)
# in-tree plugins whose volumes count against the limits of the CSI driver they were migrated to (csi-translation-lib): judging them needs
# the translation of the volume source and the node's migrated-plugins annotation -- not modelled, refused where it would matter
MIGRATABLE_PROVISIONERS = ("kubernetes.io/aws-ebs", "kubernetes.io/gce-pd", "kubernetes.io/azure-disk", "kubernetes.io/azure-file",
                           "kubernetes.io/cinder", "kubernetes.io/vsphere-volume", "kubernetes.io/portworx-volume")
MIGRATABLE_PV_SOURCES = ("awsElasticBlockStore", "gcePersistentDisk", "azureDisk", "azureFile", "cinder", "vsphereVolume", "portworxVolume")


@dataclass
class VolumeSide:
    prefilter_reject: Optional[str] = None  # the pod is UnschedulableAndUnresolvable on every node: FitError.Diagnosis.PreFilterMsg
    veto: Optional[np.ndarray] = None       # uint8[n]: model.VOL_* of the first volume plugin that rejects the node; None = none does
    exclusive: bool = False                 # a clone's disks conflict with the next one's on the same node
    rwop_capacity_one: bool = False         # a ReadWriteOncePod claim nobody uses yet: the first clone takes it, then every node fails


def _read_only(src: dict) -> bool:
    return bool(src.get("readOnly"))


def dra_prefilter(sim_pod: dict, clone_index: int = 0) -> Optional[str]:
    """DynamicResources.PreFilter for a pod with spec.resourceClaims in the reference's fake cluster, which holds NO ResourceClaim
    (SyncWithClient does not copy them, simulator.go:176-295): the first claim the plugin looks up is missing and the pod is
    UnschedulableAndUnresolvable on every node (dynamicresources.go:397-412, 562-565, 1703-1712) -- zero replicas with that message.
      resourceClaimName: c          -> `could not find ResourceClaim "ns/c"` (the scheduler's assume cache, util/assumecache NotFoundError)
      resourceClaimTemplateName: t  -> `pod "ns/<name>-<k>": ResourceClaim not created yet` (resourceclaim.Name: no status entry; the clone
                                       of cycle k is named <template>-<k>, podgenerator.go:34)
      neither                       -> `pod "ns/<name>-<k>", spec.resourceClaim "x": none of the supported fields are set`
    None: the pod names no claim (PreFilter Skip)."""
    md = sim_pod.get("metadata") or {}
    ns, name = md.get("namespace") or "default", f'{md.get("name", "")}-{clone_index}'
    for rc in (sim_pod.get("spec") or {}).get("resourceClaims") or []:
        if rc.get("resourceClaimName") is not None:
            return f'could not find ResourceClaim "{ns}/{rc["resourceClaimName"]}"'
        if rc.get("resourceClaimTemplateName") is not None:
            return f'pod "{ns}/{name}": ResourceClaim not created yet'
        return f'pod "{ns}/{name}", spec.resourceClaim "{rc.get("name", "")}": none of the supported fields are set'
    return None


def volume_conflict(v: dict, ev: dict) -> bool:
    """isVolumeConflict for one pair of volumes (volume_restrictions.go:105-150)."""
    a, b = v.get("gcePersistentDisk"), ev.get("gcePersistentDisk")
    if a is not None and b is not None and a.get("pdName", "") == b.get("pdName", "") and not (_read_only(a) and _read_only(b)):
        return True
    a, b = v.get("awsElasticBlockStore"), ev.get("awsElasticBlockStore")
    if a is not None and b is not None and a.get("volumeID", "") == b.get("volumeID", ""):
        return True
    a, b = v.get("iscsi"), ev.get("iscsi")
    if a is not None and b is not None and a.get("iqn", "") == b.get("iqn", "") and not (_read_only(a) and _read_only(b)):
        return True
    a, b = v.get("rbd"), ev.get("rbd")
    if a is not None and b is not None:
        mon, emon = a.get("monitors") or [], b.get("monitors") or []
        pool, epool = a.get("pool") or "", b.get("pool") or ""  # (as written: ParseAPISpec applies no API defaults, options.go:79-147)
        if set(mon) & set(emon) and pool == epool and a.get("image", "") == b.get("image", "") and not (_read_only(a) and _read_only(b)):
            return True
    return False


def pod_conflicts(volumes: Sequence[dict], other_volumes: Sequence[dict]) -> bool:
    """!satisfyVolumeConflicts for one existing pod (volume_restrictions.go:266-280)."""
    return any(any(v.get(k) is not None for k in RESTRICTED_KINDS) and any(volume_conflict(v, ev) for ev in other_volumes) for v in volumes)


def claim_class(pvc: dict) -> str:
    """storagehelpers.GetPersistentVolumeClaimClass: the beta annotation wins over spec.storageClassName."""
    ann = (pvc.get("metadata") or {}).get("annotations") or {}
    if ANN_BETA_STORAGE_CLASS in ann:
        return ann[ANN_BETA_STORAGE_CLASS] or ""
    return (pvc.get("spec") or {}).get("storageClassName") or ""


def _zones(value: str) -> Optional[set]:
    """volumehelpers.LabelZonesToSet: "a__b" -> {a, b}; an empty element is an error (the label is then ignored, volume_zone.go:384-388)."""
    out = set()
    for z in str(value).split("__"):
        z = z.strip()
        if not z:
            return None
        out.add(z)
    return out


def _requirement_matches(labels: dict, r: dict) -> bool:
    from .ingest import requirement_matches

    key = r.get("key", "")
    return requirement_matches(key in labels, labels.get(key), r.get("operator", ""), [str(x) for x in r.get("values") or []])


def pv_node_affinity_matches(pv: dict, node_labels: dict) -> bool:
    """storagehelpers.CheckNodeAffinity (component-helpers/storage/volume/helpers.go:68-84): the node object it builds carries the labels
    only, so a matchFields requirement on metadata.name is compared with the empty name."""
    req = ((pv.get("spec") or {}).get("nodeAffinity") or {}).get("required")
    if req is None:
        return True
    for term in req.get("nodeSelectorTerms") or []:
        exprs, fields = term.get("matchExpressions") or [], term.get("matchFields") or []
        if not exprs and not fields:
            continue  # (an empty term matches nothing: nodeaffinity.go:118-121)
        if all(_requirement_matches(node_labels, r) for r in exprs) and all(_requirement_matches({"metadata.name": ""}, r) for r in fields):
            return True
    return False


def veto_with_victims_gone(sim_pod: dict, nodes: List[dict], live: Sequence[dict], victims: Sequence[dict], index: Dict[str, int], full: "VolumeSide",
                           **kw) -> Optional[np.ndarray]:
    """The verdicts as DefaultPreemption's dry run sees them on a node once ITS lower-priority pods are removed (default_preemption.go:217-310:
    per node; the PreFilter state follows through RemovePod, volume_restrictions.go:205-213).  Disk conflicts, volume limits and what a bound
    volume says about the node depend on that node alone: the evaluation over the remaining pods.  A ReadWriteOncePod claim in use is a
    cluster-wide COUNT that the victims of node n decrement (the reference's own arithmetic, below)."""
    gone = {id(p) for p in victims}
    rest = volume_side(sim_pod, nodes, [p for p in live if id(p) not in gone], index, skip_rwop_filter=True, **kw)
    veto = np.zeros(len(nodes), np.uint8) if rest.veto is None else rest.veto.copy()
    if full.veto is not None and (full.veto == M.VOL_RWOP).any():  # (a ReadWriteOncePod claim of the pod is in use by some pod of the snapshot)
        # the reference's arithmetic (volume_restrictions.go:70-84, 219-232): PreFilter counts ONE reference per claim of the pod that is in use
        # (IsPVCUsedByPods, by namespace/name); RemovePod subtracts one for every volume of the removed pod whose claimName is in the pod's set --
        # by NAME only, whatever the victim's namespace; the node is rejected while the count is above zero (:282-291)
        ns = (sim_pod.get("metadata") or {}).get("namespace") or "default"
        pvcs = {((o.get("metadata") or {}).get("namespace") or "default", (o.get("metadata") or {}).get("name", "")): o for o in kw.get("pvc_objs") or ()}
        mine = {(v["persistentVolumeClaim"] or {}).get("claimName", "") for v in (sim_pod.get("spec") or {}).get("volumes") or [] if v.get("persistentVolumeClaim") is not None}
        rwop = {c for c in mine if "ReadWriteOncePod" in (((pvcs.get((ns, c)) or {}).get("spec") or {}).get("accessModes") or [])}
        used = set()
        for p in live:
            pns = (p.get("metadata") or {}).get("namespace") or "default"
            for v in (p.get("spec") or {}).get("volumes") or []:
                if v.get("persistentVolumeClaim") is not None:
                    used.add((pns, (v["persistentVolumeClaim"] or {}).get("claimName", "")))
        count0 = sum(1 for c in rwop if (ns, c) in used)
        released = np.zeros(len(nodes), np.int64)
        for u in victims:
            released[index[u["spec"]["nodeName"]]] += sum(1 for v in (u.get("spec") or {}).get("volumes") or []
                                                          if v.get("persistentVolumeClaim") is not None and (v["persistentVolumeClaim"] or {}).get("claimName", "") in rwop)
        for i in range(len(nodes)):
            if count0 - released[i] > 0 and veto[i] != M.VOL_DISK_CONFLICT:
                veto[i] = M.VOL_RWOP
    return veto if veto.any() else None


def csi_volume(pvc: dict, pvs: Optional[dict], classes: dict) -> Optional[Tuple[str, str]]:
    """CSILimits.getCSIDriverInfo (nodevolumelimits/csi.go:446-505, 507-541): -> (driver, unique volume name) of a claim's volume, None =
    not counted.  A claim without a (known) volume counts as one volume of its class's provisioner, named after the claim."""
    md, spec = pvc.get("metadata") or {}, pvc.get("spec") or {}
    pv = pvs.get(spec.get("volumeName") or "") if pvs is not None and spec.get("volumeName") else None
    if pv is None:
        cls = classes.get(claim_class(pvc))
        prov = (cls or {}).get("provisioner") or ""
        if not prov:
            return None
        if prov in MIGRATABLE_PROVISIONERS:
            raise NotImplementedError(f'StorageClass provisioner "{prov}": volume limits of migrated in-tree plugins are not modelled')
        return prov, f'{prov}/claim-{md.get("namespace") or "default"}/{md.get("name", "")}'
    csi = (pv.get("spec") or {}).get("csi")
    if csi is None:
        if any((pv.get("spec") or {}).get(k) is not None for k in MIGRATABLE_PV_SOURCES):
            raise NotImplementedError(f'PersistentVolume "{(pv.get("metadata") or {}).get("name", "")}": volume limits of migrated in-tree plugins are not modelled')
        return None
    driver, handle = csi.get("driver") or "", csi.get("volumeHandle") or ""
    return (driver, f"{driver}/{handle}") if driver and handle else None


def volume_side(sim_pod: dict, nodes: List[dict], live: Sequence[dict], index: Dict[str, int], pvc_objs: Sequence[dict] = (),
                class_objs: Sequence[dict] = (), pv_objs: Optional[Sequence[dict]] = None,
                enabled: Sequence[str] = PLUGINS, csinode_objs: Sequence[dict] = (), attachment_objs: Sequence[dict] = (),
                clone_index: int = 0, skip_rwop_filter: bool = False) -> VolumeSide:
    """`live`: the snapshot's non-terminal pods on kept nodes; `skip_rwop_filter`: DefaultPreemption's dry run keeps the ReadWriteOncePod verdict as a
    count of its own (veto_with_victims_gone); `pv_objs` None: persistent volumes are not synced (the reference) --
    `csinode_objs` / `attachment_objs` (CSINode, VolumeAttachment) are then ignored too: NodeVolumeLimits has no limits to check."""
    spec = sim_pod.get("spec") or {}
    ns = (sim_pod.get("metadata") or {}).get("namespace") or "default"
    volumes = list(spec.get("volumes") or [])
    out = VolumeSide()
    if not volumes:
        return out
    N = len(nodes)
    pvcs = {((o.get("metadata") or {}).get("namespace") or "default", (o.get("metadata") or {}).get("name", "")): o for o in pvc_objs}
    classes = {(o.get("metadata") or {}).get("name", ""): o for o in class_objs}
    pvs = None if pv_objs is None else {(o.get("metadata") or {}).get("name", ""): o for o in pv_objs}
    claim_vols = [v for v in volumes if v.get("persistentVolumeClaim") is not None]
    claim_names = [(v["persistentVolumeClaim"] or {}).get("claimName", "") for v in claim_vols]
    not_found = lambda kind, name: f'{kind} "{name}" not found'  # noqa: E731  (apierrors.NewNotFound(...).Error())

    # ---- PreFilter, in plugin order (framework.go:726-787: the first rejection ends the cycle) ------------------------------------------
    rwop: List[str] = []
    if "VolumeRestrictions" in enabled:  # volume_restrictions.go:166-193, 249-264
        for name in claim_names:
            pvc = pvcs.get((ns, name))
            if pvc is None:
                out.prefilter_reject = not_found("persistentvolumeclaim", name)
                return out
            if "ReadWriteOncePod" in ((pvc.get("spec") or {}).get("accessModes") or []):
                rwop.append(name)
    delayed: List[dict] = []
    bound: List[dict] = []
    # Generic ephemeral volumes: the claim is named after the CLONE -- "<pod>-<volume>", ephemeral.VolumeClaimName; the clone of cycle k is
    # <template>-<k> (podgenerator.go:34) -- and made by a controller the fake cluster does not run: VolumeBinding.PreFilter's podHasPVCs
    # (volume_binding.go:306-331) meets it missing and rejects the pod.  The volumes are walked in their order, claims and ephemeral alike.
    eph = [v for v in volumes if v.get("ephemeral") is not None]
    if eph and "VolumeBinding" not in enabled:
        raise NotImplementedError("generic ephemeral volumes without the VolumeBinding plugin are not modelled")
    if "VolumeBinding" in enabled and (claim_names or eph):  # volume_binding.go:306-383, binder.go:719-828
        clone = f'{(sim_pod.get("metadata") or {}).get("name", "")}-{clone_index}'
        for v in volumes:
            if v.get("ephemeral") is not None:
                made = f'{clone}-{v.get("name", "")}'
                if (ns, made) in pvcs:
                    raise NotImplementedError(f'persistentvolumeclaim "{made}" exists: whether it was created for the simulated pod is not modelled')
                out.prefilter_reject = f'waiting for ephemeral volume controller to create the persistentvolumeclaim "{made}"'
                return out
            if v.get("persistentVolumeClaim") is None:
                continue
            name = (v["persistentVolumeClaim"] or {}).get("claimName", "")
            pvc = pvcs.get((ns, name))
            if pvc is None:
                out.prefilter_reject = not_found("persistentvolumeclaim", name)
                return out
            if (pvc.get("status") or {}).get("phase") == "Lost":
                out.prefilter_reject = f'persistentvolumeclaim "{name}" bound to non-existent persistentvolume "{(pvc.get("spec") or {}).get("volumeName") or ""}"'
                return out
            if (pvc.get("metadata") or {}).get("deletionTimestamp") is not None:
                out.prefilter_reject = f'persistentvolumeclaim "{name}" is being deleted'
                return out
        immediate = False
        for name in claim_names:
            pvc = pvcs[(ns, name)]
            vol_name = (pvc.get("spec") or {}).get("volumeName") or ""
            if vol_name and ANN_BIND_COMPLETED in ((pvc.get("metadata") or {}).get("annotations") or {}):
                bound.append(pvc)
                continue
            cname = claim_class(pvc)
            delay = False
            if cname:  # volume.IsDelayBindingMode (an error of the class lister is a scheduler ERROR, not a rejection)
                cls = classes.get(cname)
                if cls is None:
                    raise NotImplementedError(f'persistentvolumeclaim "{name}": StorageClass "{cname}" is not in the snapshot (the reference\'s scheduler fails the cycle with an error)')
                if cls.get("volumeBindingMode") is None:
                    raise NotImplementedError(f'StorageClass "{cname}" has no volumeBindingMode (the reference\'s scheduler fails the cycle with an error)')
                delay = cls["volumeBindingMode"] == "WaitForFirstConsumer"
            if delay and not vol_name:
                delayed.append(pvc)
            else:
                immediate = True
        if immediate:
            out.prefilter_reject = "pod has unbound immediate PersistentVolumeClaims"
            return out
    topologies: List[Tuple[str, set]] = []
    if "VolumeZone" in enabled:  # volume_zone.go:111-165
        for name in claim_names:
            if name == "":
                out.prefilter_reject = "PersistentVolumeClaim had no name"
                return out
            pvc = pvcs.get((ns, name))
            if pvc is None:
                out.prefilter_reject = not_found("persistentvolumeclaim", name)
                return out
            vol_name = (pvc.get("spec") or {}).get("volumeName") or ""
            if not vol_name:
                cname = claim_class(pvc)
                if not cname:
                    out.prefilter_reject = "PersistentVolumeClaim had no pv name and storageClass name"
                    return out
                cls = classes.get(cname)
                if cls is None:
                    out.prefilter_reject = not_found("storageclass.storage.k8s.io", cname)
                    return out
                if cls.get("volumeBindingMode") is None:
                    out.prefilter_reject = f'VolumeBindingMode not set for StorageClass "{cname}"'
                    return out
                if cls["volumeBindingMode"] == "WaitForFirstConsumer":
                    continue
                out.prefilter_reject = "PersistentVolume had no name"
                return out
            pv = None if pvs is None else pvs.get(vol_name)
            if pv is None:
                out.prefilter_reject = not_found("persistentvolume", vol_name)
                return out
            labels = (pv.get("metadata") or {}).get("labels") or {}
            for key in TOPOLOGY_LABELS:
                if key in labels:
                    zs = _zones(labels[key])
                    if zs is not None:
                        topologies.append((key, zs))

    # ---- Filter: the first of the four that rejects the node (default_plugins.go:41-44) --------------------------------------------------
    veto = np.zeros(N, np.uint8)

    def mark(mask, code):
        veto[(veto == 0) & mask] = code

    if "VolumeRestrictions" in enabled:
        if any(v.get(k) is not None for v in volumes for k in RESTRICTED_KINDS):
            conflict = np.zeros(N, bool)
            for p in live:
                if pod_conflicts(volumes, (p.get("spec") or {}).get("volumes") or []):
                    conflict[index[p["spec"]["nodeName"]]] = True
            mark(conflict, M.VOL_DISK_CONFLICT)
            out.exclusive = pod_conflicts(volumes, volumes)
        if rwop:  # IsPVCUsedByPods (S/backend/cache/snapshot.go usedPVCSet: every pod of every node, keyed namespace/name)
            used = set()
            for p in live:
                pns = (p.get("metadata") or {}).get("namespace") or "default"
                for v in (p.get("spec") or {}).get("volumes") or []:
                    if v.get("persistentVolumeClaim") is not None:
                        used.add((pns, (v["persistentVolumeClaim"] or {}).get("claimName", "")))
            if any((ns, name) in used for name in rwop):
                if not skip_rwop_filter:
                    mark(np.ones(N, bool), M.VOL_RWOP)
            else:
                out.rwop_capacity_one = True
    # NodeVolumeLimits (nodevolumelimits/csi.go:255-339): no CSINode in the reference's fake cluster, hence no limits (:265-290).  With the
    # snapshot's volumes synced, CSINodes and VolumeAttachments are taken too: per node, the pod's NEW volumes (not attached there yet) per
    # driver against the driver's allocatable count minus what the node's pods and attachments hold.  Static: the clones share the
    # template's claims, so a node's first clone attaches them and every later one adds nothing.
    if "NodeVolumeLimits" in enabled and pvs is not None and claim_names and csinode_objs:
        limits_of = {}
        for o in csinode_objs:
            lim = {d.get("name", ""): int((d.get("allocatable") or {})["count"]) for d in ((o.get("spec") or {}).get("drivers") or [])
                   if (d.get("allocatable") or {}).get("count") is not None}
            if lim and (o.get("metadata") or {}).get("name", "") in index:
                limits_of[index[o["metadata"]["name"]]] = lim
        new: Dict[str, str] = {}
        for name in claim_names:
            pvc = pvcs.get((ns, name))
            if pvc is None:  # (filterAttachableVolumes for the new pod: UnschedulableAndUnresolvable on every node that has a CSINode ... with limits or not)
                raise NotImplementedError(f'persistentvolumeclaim "{name}" is missing and only NodeVolumeLimits would notice: not modelled')
            dv = csi_volume(pvc, pvs, classes)
            if dv is not None:
                new[dv[1]] = dv[0]
        if new and limits_of:
            held: Dict[int, Dict[str, str]] = {}
            for p in live:
                i = index[p["spec"]["nodeName"]]
                if i not in limits_of:
                    continue
                pns = (p.get("metadata") or {}).get("namespace") or "default"
                for v in (p.get("spec") or {}).get("volumes") or []:
                    if v.get("persistentVolumeClaim") is None:
                        continue
                    q = pvcs.get((pns, (v["persistentVolumeClaim"] or {}).get("claimName", "")))
                    dv = csi_volume(q, pvs, classes) if q is not None else None  # (an existing pod's unknown claim is not counted: :393-399)
                    if dv is not None:
                        held.setdefault(i, {})[dv[1]] = dv[0]
            extra: Dict[int, Dict[str, str]] = {}
            for va in attachment_objs:  # getNodeVolumeAttachmentInfo (:572-601)
                sp = va.get("spec") or {}
                i = index.get(sp.get("nodeName") or "")
                pv = pvs.get((sp.get("source") or {}).get("persistentVolumeName") or "")
                csi = ((pv or {}).get("spec") or {}).get("csi")
                if i is None or i not in limits_of or not sp.get("attacher") or csi is None:
                    continue
                extra.setdefault(i, {})[f'{sp["attacher"]}/{csi.get("volumeHandle") or ""}'] = sp["attacher"]
            over = np.zeros(N, bool)
            for i, lim in limits_of.items():
                attached = held.get(i, {})
                count: Dict[str, int] = {}
                for drv in attached.values():
                    count[drv] = count.get(drv, 0) + 1
                for uniq, drv in extra.get(i, {}).items():
                    if uniq not in attached:
                        count[drv] = count.get(drv, 0) + 1
                fresh: Dict[str, int] = {}
                for uniq, drv in new.items():
                    if uniq not in attached:
                        fresh[drv] = fresh.get(drv, 0) + 1
                over[i] = any(drv in lim and count.get(drv, 0) + k > lim[drv] for drv, k in fresh.items())
            mark(over, M.VOL_MAX_COUNT)
    if "VolumeBinding" in enabled:
        if bound:  # binder.go checkBoundClaims (:830-865): per node, claim by claim in the pod's order -- the FIRST failure is the node's verdict
            node_labels = [(n.get("metadata") or {}).get("labels") or {} for n in nodes]
            verdict = np.zeros(N, np.uint8)
            for i, lb in enumerate(node_labels):
                for c in bound:
                    pv = None if pvs is None else pvs.get((c.get("spec") or {}).get("volumeName") or "")
                    if pv is None:
                        verdict[i] = M.VOL_PV_NOT_EXIST
                        break
                    if not pv_node_affinity_matches(pv, lb):
                        verdict[i] = M.VOL_NODE_AFFINITY
                        break
            mark(verdict == M.VOL_PV_NOT_EXIST, M.VOL_PV_NOT_EXIST)
            mark(verdict == M.VOL_NODE_AFFINITY, M.VOL_NODE_AFFINITY)
        for pvc in delayed:  # binder.go findMatchingVolumes (no volume of the class to match) -> checkVolumeProvisions
            cname = claim_class(pvc)
            if pvs is not None and any(((pv.get("spec") or {}).get("storageClassName") or "") == cname for pv in pvs.values()):
                raise NotImplementedError(f'persistentvolumeclaim "{pvc["metadata"]["name"]}": matching an unbound claim against the persistent volumes of class "{cname}" is not modelled')
            prov = (classes[cname].get("provisioner") or "")
            if prov == "" or prov == NO_PROVISIONER:
                mark(np.ones(N, bool), M.VOL_NO_PV)
            else:
                raise NotImplementedError(f'persistentvolumeclaim "{pvc["metadata"]["name"]}" waits for its first consumer: StorageClass "{cname}" would provision the volume in '
                                          "PreBind, which waits for a PV controller the simulated cluster does not run (the reference does not terminate)")
    if "VolumeZone" in enabled and topologies:  # volume_zone.go:191-240
        bad = np.zeros(N, bool)
        for i, n in enumerate(nodes):
            labels = (n.get("metadata") or {}).get("labels") or {}
            if not any(k in labels for k in TOPOLOGY_LABELS):
                continue  # (a node without any zone label is fine: a single-zone cluster)
            for key, zs in topologies:
                ga = {ZONE_BETA: ZONE_GA, REGION_BETA: REGION_GA}.get(key, key)
                val = labels.get(key) if key in labels else labels.get(ga)
                if val is None or val not in zs:
                    bad[i] = True
                    break
        mark(bad, M.VOL_ZONE)
    out.veto = veto if veto.any() else None
    return out
