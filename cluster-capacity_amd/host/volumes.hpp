// volumes.hpp -- the volume plugins' object side: VolumeRestrictions, NodeVolumeLimits, VolumeBinding, VolumeZone.
//
// What they decide is string and object-graph work that does not change while clones are placed -- except a clone's own disks and a
// ReadWriteOncePod claim -- so it is evaluated here, once per template, into what the engine takes (include/ccsim.h ccsim_pod):
// volume_veto[n] (the code of the first of the four that rejects node n against the snapshot's pods) and volume_exclusive.  The PreFilter
// outcomes that reject the pod on every node (zero replicas, the plugin's message as FitError.Diagnosis.PreFilterMsg) never reach the
// engine.  Same semantics as cluster_capacity_amd/volumes.py, checked against it in tests/test_volume_ingest.py.
//
// What the REFERENCE's scheduler sees: SyncWithClient copies PersistentVolumeClaims and StorageClasses into the fake cluster but NOT
// PersistentVolumes, CSINodes, CSIDrivers or CSIStorageCapacities (pkg/framework/simulator.go:228-295).  With the default plugins:
//   a claim that does not exist            VolumeRestrictions.PreFilter: persistentvolumeclaim "x" not found (volume_restrictions.go:175-181)
//   a lost / terminating claim             VolumeBinding.PreFilter (volumebinding/volume_binding.go:333-339, 356-357)
//   a generic ephemeral volume             VolumeBinding.PreFilter: its claim "<clone>-<volume>" is never created (:306-331)
//   an unbound claim of an Immediate class VolumeBinding.PreFilter: "pod has unbound immediate PersistentVolumeClaims" (:366-372)
//   a BOUND claim                          VolumeZone.PreFilter: persistentvolume "pv" not found (volumezone/volume_zone.go:156-159, 253-258)
//   an unbound WaitForFirstConsumer claim  passes every PreFilter; VolumeBinding.Filter finds nothing to bind (binder.go
//                                          checkVolumeProvisions): without a provisioner every node fails with "node(s) didn't find
//                                          available persistent volumes to bind"; with one the pod passes and then waits in PreBind for
//                                          a PV controller the fake cluster does not run -- the reference hangs; refused with that reason
//   NodeVolumeLimits                       never rejects: no CSINode, no limits (nodevolumelimits/csi.go:265-290)
//   GCE PD / EBS / RBD / ISCSI volumes     VolumeRestrictions.Filter against the node's pods and the clones (:105-150, 310-313)
//   a ReadWriteOncePod claim               in use by a pod of the snapshot: every node fails (:283-291); else the first clone takes it
//                                          and the second cycle fails everywhere: capacity 1
// --sync-persistent-volumes goes one step beyond the reference: PersistentVolume, CSINode and VolumeAttachment objects of the snapshot are
// taken too, so bound claims are judged as kube-scheduler judges them on the live cluster -- VolumeBinding's node affinity of the bound
// volume (binder.go checkBoundClaims), VolumeZone's label match (volume_zone.go:191-240) and NodeVolumeLimits' per-driver counts
// (nodevolumelimits/csi.go:255-339) -- as static per-node verdicts.
#pragma once
#include <algorithm>
#include <map>
#include <set>
#include <string>
#include <vector>

#include "value.hpp"

namespace cchost {

// (included by snapshot.hpp below its requirement_matches / string_list)

static const char *const kVolumePlugins[] = {"VolumeRestrictions", "NodeVolumeLimits", "VolumeBinding", "VolumeZone"};
static const char *const kAnnBindCompleted = "pv.kubernetes.io/bind-completed";           // volume.AnnBindCompleted
static const char *const kAnnBetaStorageClass = "volume.beta.kubernetes.io/storage-class"; // v1.BetaStorageClassAnnotation
static const char *const kNoProvisioner = "kubernetes.io/no-provisioner";                  // volume.NotSupportedProvisioner
static const char *const kZoneBeta = "failure-domain.beta.kubernetes.io/zone";
static const char *const kRegionBeta = "failure-domain.beta.kubernetes.io/region";
static const char *const kZoneGA = "topology.kubernetes.io/zone";
static const char *const kRegionGA = "topology.kubernetes.io/region";
static const char *const kTopologyLabels[] = {kZoneBeta, kRegionBeta, kZoneGA, kRegionGA}; // volume_zone.go:84-89
static const char *const kRestrictedKinds[] = {"gcePersistentDisk", "awsElasticBlockStore", "rbd", "iscsi"}; // needsRestrictionsCheck

// the objects of the snapshot the volume plugins read, and which of the plugins the profile runs
struct VolumeObjects {
    std::vector<Value> claims, classes, volumes;
    std::vector<Value> csinodes, attachments; // CSINode / VolumeAttachment: taken with the volumes only (NodeVolumeLimits has no limits to check without)
    bool sync_volumes = false; // --sync-persistent-volumes (the reference's SyncWithClient copies none)
    std::vector<std::string> plugins = {"VolumeRestrictions", "NodeVolumeLimits", "VolumeBinding", "VolumeZone"};
    bool plugins_partial = false; // the configuration disables only the filter point of one: pods with volumes are refused
    bool dra_enabled = true, dra_partial = false; // DynamicResources in the profile (a pod with spec.resourceClaims: dra_prefilter)
    bool on(const char *name) const { return std::find(plugins.begin(), plugins.end(), name) != plugins.end(); }
};

struct VolumeSide {
    bool rejected = false;        // a PreFilter plugin rejects the pod on every node ...
    std::string prefilter_reject; // ... with this message
    std::vector<uint8_t> veto;    // per node: code 1..CCSIM_VOL_CODES of the first volume plugin that rejects it; empty = none does
    bool exclusive = false;       // a clone's disks conflict with the next one's on the same node
    bool rwop_capacity_one = false; // a ReadWriteOncePod claim nobody uses yet
};

inline bool restricted(const Value &v) {
    for (const char *k : kRestrictedKinds)
        if (!v[k].is_null()) return true;
    return false;
}

// DynamicResources.PreFilter for a pod with spec.resourceClaims in the reference's fake cluster, which holds NO ResourceClaim (SyncWithClient
// does not copy them, simulator.go:176-295): the first claim the plugin looks up is missing and the pod is UnschedulableAndUnresolvable on
// every node (dynamicresources.go:397-412, 562-565, 1703-1712) -- zero replicas with that message.  "" = the pod names no claim (Skip).
//   resourceClaimName: c          could not find ResourceClaim "ns/c"            (the scheduler's assume cache, util/assumecache NotFoundError)
//   resourceClaimTemplateName: t  pod "ns/<name>-<k>": ResourceClaim not created yet   (the clone of cycle k is <template>-<k>, podgenerator.go:34)
//   neither                       pod "ns/<name>-<k>", spec.resourceClaim "x": none of the supported fields are set
inline std::string dra_prefilter(const Value &sim_pod, size_t clone_index) {
    const std::string ns = sim_pod["metadata"]["namespace"].truthy() ? sim_pod["metadata"]["namespace"].text() : "default";
    const std::string name = sim_pod["metadata"]["name"].text() + "-" + std::to_string(clone_index);
    for (const auto &rc : sim_pod["spec"]["resourceClaims"].items()) {
        if (!rc["resourceClaimName"].is_null()) return "could not find ResourceClaim \"" + ns + "/" + rc["resourceClaimName"].text() + "\"";
        if (!rc["resourceClaimTemplateName"].is_null()) return "pod \"" + ns + "/" + name + "\": ResourceClaim not created yet";
        return "pod \"" + ns + "/" + name + "\", spec.resourceClaim \"" + rc["name"].text() + "\": none of the supported fields are set";
    }
    return "";
}

// isVolumeConflict for one pair of volumes (volume_restrictions.go:105-150)
inline bool volume_conflict(const Value &v, const Value &ev) {
    auto ro = [](const Value &s) { return s["readOnly"].truthy(); };
    {
        const Value &a = v["gcePersistentDisk"], &b = ev["gcePersistentDisk"];
        if (!a.is_null() && !b.is_null() && a["pdName"].text() == b["pdName"].text() && !(ro(a) && ro(b))) return true;
    }
    {
        const Value &a = v["awsElasticBlockStore"], &b = ev["awsElasticBlockStore"];
        if (!a.is_null() && !b.is_null() && a["volumeID"].text() == b["volumeID"].text()) return true;
    }
    {
        const Value &a = v["iscsi"], &b = ev["iscsi"];
        if (!a.is_null() && !b.is_null() && a["iqn"].text() == b["iqn"].text() && !(ro(a) && ro(b))) return true;
    }
    {
        const Value &a = v["rbd"], &b = ev["rbd"];
        if (!a.is_null() && !b.is_null()) {
            bool overlap = false;
            for (const auto &m : a["monitors"].items())
                for (const auto &e : b["monitors"].items()) overlap = overlap || m.text() == e.text();
            // (pool as written: ParseAPISpec applies no API defaults, options.go:79-147)
            if (overlap && a["pool"].text() == b["pool"].text() && a["image"].text() == b["image"].text() && !(ro(a) && ro(b))) return true;
        }
    }
    return false;
}

// !satisfyVolumeConflicts for one existing pod (volume_restrictions.go:266-280)
inline bool pod_conflicts(const Value &volumes, const Value &other_volumes) {
    for (const auto &v : volumes.items()) {
        if (!restricted(v)) continue;
        for (const auto &ev : other_volumes.items())
            if (volume_conflict(v, ev)) return true;
    }
    return false;
}

// storagehelpers.GetPersistentVolumeClaimClass: the beta annotation wins over spec.storageClassName
inline std::string claim_class(const Value &pvc) {
    const Value &ann = pvc["metadata"]["annotations"];
    if (ann.has(kAnnBetaStorageClass)) return ann[kAnnBetaStorageClass].text();
    return pvc["spec"]["storageClassName"].text();
}

// volumehelpers.LabelZonesToSet: "a__b" -> {a, b}; an empty element is an error (the label is then ignored, volume_zone.go:384-388)
inline bool label_zones(const std::string &value, std::set<std::string> &out) {
    size_t pos = 0;
    for (;;) {
        const size_t next = value.find("__", pos);
        std::string z = value.substr(pos, next == std::string::npos ? std::string::npos : next - pos);
        const size_t b = z.find_first_not_of(" \t\r\n\v\f"), e = z.find_last_not_of(" \t\r\n\v\f");
        if (b == std::string::npos) return false;
        out.insert(z.substr(b, e - b + 1));
        if (next == std::string::npos) return true;
        pos = next + 2;
    }
}

// storagehelpers.CheckNodeAffinity (component-helpers/storage/volume/helpers.go:68-84): the node object it builds carries the labels only,
// so a matchFields requirement on metadata.name is compared with the empty name
inline bool pv_node_affinity_matches(const Value &pv, const Value &node_labels) {
    const Value &req = pv["spec"]["nodeAffinity"]["required"];
    if (req.is_null()) return true;
    for (const auto &term : req["nodeSelectorTerms"].items()) {
        const Value &exprs = term["matchExpressions"], &fields = term["matchFields"];
        if (exprs.items().empty() && fields.items().empty()) continue; // (an empty term matches nothing: nodeaffinity.go:118-121)
        bool ok = true;
        for (const auto &r : exprs.items()) {
            const std::string k = r["key"].text();
            ok = ok && requirement_matches(node_labels.has(k), node_labels[k].text(), r["operator"].text(), string_list(r["values"]));
        }
        for (const auto &r : fields.items()) ok = ok && requirement_matches(r["key"].text() == "metadata.name", "", r["operator"].text(), string_list(r["values"]));
        if (ok) return true;
    }
    return false;
}

struct VolumeUnsupported : std::runtime_error {
    using std::runtime_error::runtime_error;
};

// in-tree plugins whose volumes count against the limits of the CSI driver they were migrated to (csi-translation-lib): judging them needs the
// translation of the volume source and the node's migrated-plugins annotation -- not modelled, refused where it would matter
inline bool migratable_provisioner(const std::string &p) {
    for (const char *x : {"kubernetes.io/aws-ebs", "kubernetes.io/gce-pd", "kubernetes.io/azure-disk", "kubernetes.io/azure-file", "kubernetes.io/cinder",
                          "kubernetes.io/vsphere-volume", "kubernetes.io/portworx-volume"})
        if (p == x) return true;
    return false;
}

// CSILimits.getCSIDriverInfo (nodevolumelimits/csi.go:446-505, 507-541): (driver, unique volume name) of a claim's volume; false = not
// counted.  A claim without a (known) volume counts as one volume of its class's provisioner, named after the claim.
inline bool csi_volume(const Value &pvc, const std::map<std::string, const Value *> &pvs, const std::map<std::string, const Value *> &classes, std::string &driver,
                       std::string &unique) {
    const std::string vol_name = pvc["spec"]["volumeName"].text();
    auto pv = vol_name.empty() ? pvs.end() : pvs.find(vol_name);
    if (pv == pvs.end()) {
        auto cls = classes.find(claim_class(pvc));
        const std::string prov = cls == classes.end() ? std::string() : (*cls->second)["provisioner"].text();
        if (prov.empty()) return false;
        if (migratable_provisioner(prov)) throw VolumeUnsupported("StorageClass provisioner \"" + prov + "\": volume limits of migrated in-tree plugins are not modelled");
        const std::string ns = pvc["metadata"]["namespace"].truthy() ? pvc["metadata"]["namespace"].text() : "default";
        driver = prov, unique = prov + "/claim-" + ns + "/" + pvc["metadata"]["name"].text();
        return true;
    }
    const Value &spec = (*pv->second)["spec"];
    const Value &csi = spec["csi"];
    if (csi.is_null()) {
        for (const char *k : {"awsElasticBlockStore", "gcePersistentDisk", "azureDisk", "azureFile", "cinder", "vsphereVolume", "portworxVolume"})
            if (!spec[k].is_null())
                throw VolumeUnsupported("PersistentVolume \"" + (*pv->second)["metadata"]["name"].text() + "\": volume limits of migrated in-tree plugins are not modelled");
        return false;
    }
    driver = csi["driver"].text();
    const std::string handle = csi["volumeHandle"].text();
    if (driver.empty() || handle.empty()) return false;
    unique = driver + "/" + handle;
    return true;
}

// `live`: the snapshot's non-terminal pods on kept nodes, `live_node[j]`: the node index of live[j]
inline VolumeSide volume_side(const Value &sim_pod, const std::vector<const Value *> &nodes, const std::vector<const Value *> &live,
                              const std::vector<size_t> &live_node, const VolumeObjects &vo, size_t clone_index = 0, bool skip_rwop_filter = false) {
    VolumeSide out;
    const Value &spec = sim_pod["spec"];
    const std::string ns = sim_pod["metadata"]["namespace"].truthy() ? sim_pod["metadata"]["namespace"].text() : "default";
    const Value &volumes = spec["volumes"];
    if (volumes.items().empty()) return out;
    if (vo.plugins_partial)
        throw VolumeUnsupported("the scheduler configuration disables only the filter point of a volume plugin: a pod with volumes is not modelled under it");
    const size_t N = nodes.size();
    auto obj_ns = [](const Value &o) { return o["metadata"]["namespace"].truthy() ? o["metadata"]["namespace"].text() : std::string("default"); };
    std::map<std::pair<std::string, std::string>, const Value *> pvcs;
    for (const auto &o : vo.claims) pvcs[{obj_ns(o), o["metadata"]["name"].text()}] = &o;
    std::map<std::string, const Value *> classes, pvs;
    for (const auto &o : vo.classes) classes[o["metadata"]["name"].text()] = &o;
    if (vo.sync_volumes)
        for (const auto &o : vo.volumes) pvs[o["metadata"]["name"].text()] = &o;
    std::vector<std::string> claim_names;
    for (const auto &v : volumes.items())
        if (!v["persistentVolumeClaim"].is_null()) claim_names.push_back(v["persistentVolumeClaim"]["claimName"].text());
    auto find_pvc = [&](const std::string &name) -> const Value * {
        auto it = pvcs.find({ns, name});
        return it == pvcs.end() ? nullptr : it->second;
    };
    auto reject = [&](std::string msg) {
        out.rejected = true, out.prefilter_reject = std::move(msg);
        return out;
    };
    auto not_found = [](const char *kind, const std::string &name) { return std::string(kind) + " \"" + name + "\" not found"; }; // apierrors.NewNotFound(...).Error()

    // ---- PreFilter, in plugin order (framework.go:726-787: the first rejection ends the cycle) ---------------------------------------
    std::vector<std::string> rwop;
    if (vo.on("VolumeRestrictions")) // volume_restrictions.go:166-193, 249-264
        for (const auto &name : claim_names) {
            const Value *pvc = find_pvc(name);
            if (!pvc) return reject(not_found("persistentvolumeclaim", name));
            for (const auto &m : (*pvc)["spec"]["accessModes"].items())
                if (m.text() == "ReadWriteOncePod") {
                    rwop.push_back(name);
                    break;
                }
        }
    std::vector<const Value *> delayed, bound;
    // Generic ephemeral volumes: the claim is named after the CLONE -- "<pod>-<volume>", ephemeral.VolumeClaimName; the clone of cycle k is
    // <template>-<k> (podgenerator.go:34) -- and made by a controller the fake cluster does not run: VolumeBinding.PreFilter's podHasPVCs
    // (volume_binding.go:306-331) meets it missing and rejects the pod.  The volumes are walked in their order, claims and ephemeral alike.
    bool has_eph = false;
    for (const auto &v : volumes.items()) has_eph = has_eph || !v["ephemeral"].is_null();
    if (has_eph && !vo.on("VolumeBinding")) throw VolumeUnsupported("generic ephemeral volumes without the VolumeBinding plugin are not modelled");
    if (vo.on("VolumeBinding") && (!claim_names.empty() || has_eph)) { // volume_binding.go:306-383, binder.go:719-828
        const std::string clone = sim_pod["metadata"]["name"].text() + "-" + std::to_string(clone_index);
        for (const auto &v : volumes.items()) {
            if (!v["ephemeral"].is_null()) {
                const std::string made = clone + "-" + v["name"].text();
                if (find_pvc(made)) throw VolumeUnsupported("persistentvolumeclaim \"" + made + "\" exists: whether it was created for the simulated pod is not modelled");
                return reject("waiting for ephemeral volume controller to create the persistentvolumeclaim \"" + made + "\"");
            }
            if (v["persistentVolumeClaim"].is_null()) continue;
            const std::string name = v["persistentVolumeClaim"]["claimName"].text();
            const Value *pvc = find_pvc(name);
            if (!pvc) return reject(not_found("persistentvolumeclaim", name));
            if ((*pvc)["status"]["phase"].text() == "Lost")
                return reject("persistentvolumeclaim \"" + name + "\" bound to non-existent persistentvolume \"" + (*pvc)["spec"]["volumeName"].text() + "\"");
            if (!(*pvc)["metadata"]["deletionTimestamp"].is_null()) return reject("persistentvolumeclaim \"" + name + "\" is being deleted");
        }
        bool immediate = false;
        for (const auto &name : claim_names) {
            const Value &pvc = *find_pvc(name);
            const std::string vol_name = pvc["spec"]["volumeName"].text();
            if (!vol_name.empty() && pvc["metadata"]["annotations"].has(kAnnBindCompleted)) {
                bound.push_back(&pvc);
                continue;
            }
            const std::string cname = claim_class(pvc);
            bool delay = false;
            if (!cname.empty()) { // volume.IsDelayBindingMode (an error of the class lister is a scheduler ERROR, not a rejection)
                auto it = classes.find(cname);
                if (it == classes.end())
                    throw VolumeUnsupported("persistentvolumeclaim \"" + name + "\": StorageClass \"" + cname + "\" is not in the snapshot (the reference's scheduler fails the cycle with an error)");
                if ((*it->second)["volumeBindingMode"].is_null())
                    throw VolumeUnsupported("StorageClass \"" + cname + "\" has no volumeBindingMode (the reference's scheduler fails the cycle with an error)");
                delay = (*it->second)["volumeBindingMode"].text() == "WaitForFirstConsumer";
            }
            if (delay && vol_name.empty()) delayed.push_back(&pvc);
            else immediate = true;
        }
        if (immediate) return reject("pod has unbound immediate PersistentVolumeClaims");
    }
    std::vector<std::pair<std::string, std::set<std::string>>> topologies;
    if (vo.on("VolumeZone")) // volume_zone.go:111-165
        for (const auto &name : claim_names) {
            if (name.empty()) return reject("PersistentVolumeClaim had no name");
            const Value *pvc = find_pvc(name);
            if (!pvc) return reject(not_found("persistentvolumeclaim", name));
            const std::string vol_name = (*pvc)["spec"]["volumeName"].text();
            if (vol_name.empty()) {
                const std::string cname = claim_class(*pvc);
                if (cname.empty()) return reject("PersistentVolumeClaim had no pv name and storageClass name");
                auto it = classes.find(cname);
                if (it == classes.end()) return reject(not_found("storageclass.storage.k8s.io", cname));
                if ((*it->second)["volumeBindingMode"].is_null()) return reject("VolumeBindingMode not set for StorageClass \"" + cname + "\"");
                if ((*it->second)["volumeBindingMode"].text() == "WaitForFirstConsumer") continue;
                return reject("PersistentVolume had no name");
            }
            auto pv = pvs.find(vol_name);
            if (pv == pvs.end()) return reject(not_found("persistentvolume", vol_name));
            const Value &labels = (*pv->second)["metadata"]["labels"];
            for (const char *key : kTopologyLabels)
                if (labels.has(key)) {
                    std::set<std::string> zs;
                    if (label_zones(labels[key].text(), zs)) topologies.emplace_back(key, std::move(zs));
                }
        }

    // ---- Filter: the first of the four that rejects the node (default_plugins.go:41-44) ------------------------------------------------
    std::vector<uint8_t> veto(N, 0);
    bool any = false;
    auto mark = [&](size_t i, int code) {
        if (!veto[i]) veto[i] = (uint8_t)code, any = true;
    };
    auto mark_all = [&](int code) {
        for (size_t i = 0; i < N; i++) mark(i, code);
    };
    if (vo.on("VolumeRestrictions")) {
        bool needs = false;
        for (const auto &v : volumes.items()) needs = needs || restricted(v);
        if (needs) {
            for (size_t j = 0; j < live.size(); j++)
                if (pod_conflicts(volumes, (*live[j])["spec"]["volumes"])) mark(live_node[j], 1);
            out.exclusive = pod_conflicts(volumes, volumes);
        }
        if (!rwop.empty()) { // IsPVCUsedByPods (the snapshot's usedPVCSet: every pod of every node, keyed namespace/name)
            bool used = false;
            for (size_t j = 0; j < live.size() && !used; j++) {
                const std::string pns = obj_ns(*live[j]);
                if (pns != ns) continue;
                for (const auto &v : (*live[j])["spec"]["volumes"].items())
                    if (!v["persistentVolumeClaim"].is_null() &&
                        std::find(rwop.begin(), rwop.end(), v["persistentVolumeClaim"]["claimName"].text()) != rwop.end())
                        used = true;
            }
            if (used && !skip_rwop_filter) mark_all(2); // (skip: DefaultPreemption's dry run keeps this verdict as a count of its own, veto_with_victims_gone)
            else if (!used) out.rwop_capacity_one = true;
        }
    }
    // NodeVolumeLimits (nodevolumelimits/csi.go:255-339): no CSINode in the reference's fake cluster, hence no limits (:265-290).  With the
    // snapshot's volumes synced, CSINodes and VolumeAttachments are taken too: per node, the pod's NEW volumes (not attached there yet) per
    // driver against the driver's allocatable count minus what the node's pods and attachments hold.  Static: the clones share the
    // template's claims, so a node's first clone attaches them and every later one adds nothing.
    if (vo.on("NodeVolumeLimits") && vo.sync_volumes && !claim_names.empty() && !vo.csinodes.empty()) {
        std::map<std::string, size_t> node_index;
        for (size_t i = 0; i < N; i++) node_index[(*nodes[i])["metadata"]["name"].text()] = i;
        std::map<size_t, std::map<std::string, long long>> limits_of;
        for (const auto &o : vo.csinodes) {
            auto it = node_index.find(o["metadata"]["name"].text());
            if (it == node_index.end()) continue;
            std::map<std::string, long long> lim;
            for (const auto &d : o["spec"]["drivers"].items())
                if (!d["allocatable"]["count"].is_null()) lim[d["name"].text()] = d["allocatable"]["count"].as_int();
            if (!lim.empty()) limits_of[it->second] = std::move(lim);
        }
        std::map<std::string, std::string> fresh_all; // unique volume name -> driver
        for (const auto &name : claim_names) {
            const Value *pvc = find_pvc(name);
            if (!pvc) throw VolumeUnsupported("persistentvolumeclaim \"" + name + "\" is missing and only NodeVolumeLimits would notice: not modelled");
            std::string drv, uniq;
            if (csi_volume(*pvc, pvs, classes, drv, uniq)) fresh_all[uniq] = drv;
        }
        if (!fresh_all.empty() && !limits_of.empty()) {
            std::map<size_t, std::map<std::string, std::string>> held, extra;
            for (size_t j = 0; j < live.size(); j++) {
                if (!limits_of.count(live_node[j])) continue;
                const std::string pns = obj_ns(*live[j]);
                for (const auto &v : (*live[j])["spec"]["volumes"].items()) {
                    if (v["persistentVolumeClaim"].is_null()) continue;
                    auto q = pvcs.find({pns, v["persistentVolumeClaim"]["claimName"].text()});
                    std::string drv, uniq; // (an existing pod's unknown claim is not counted: csi.go:393-399)
                    if (q != pvcs.end() && csi_volume(*q->second, pvs, classes, drv, uniq)) held[live_node[j]][uniq] = drv;
                }
            }
            for (const auto &va : vo.attachments) { // getNodeVolumeAttachmentInfo (:572-601)
                const Value &sp = va["spec"];
                auto it = node_index.find(sp["nodeName"].text());
                auto pv = pvs.find(sp["source"]["persistentVolumeName"].text());
                if (it == node_index.end() || !limits_of.count(it->second) || !sp["attacher"].truthy() || pv == pvs.end() || (*pv->second)["spec"]["csi"].is_null()) continue;
                extra[it->second][sp["attacher"].text() + "/" + (*pv->second)["spec"]["csi"]["volumeHandle"].text()] = sp["attacher"].text();
            }
            for (const auto &kv : limits_of) {
                const size_t i = kv.first;
                const auto &attached = held[i];
                std::map<std::string, long long> count, fresh;
                for (const auto &a : attached) count[a.second] += 1;
                for (const auto &x : extra[i])
                    if (!attached.count(x.first)) count[x.second] += 1;
                for (const auto &f : fresh_all)
                    if (!attached.count(f.first)) fresh[f.second] += 1;
                bool over = false;
                for (const auto &f : fresh) {
                    auto lim = kv.second.find(f.first);
                    over = over || (lim != kv.second.end() && count[f.first] + f.second > lim->second);
                }
                if (over) mark(i, 3);
            }
        }
    }
    if (vo.on("VolumeBinding")) {
        if (!bound.empty()) // binder.go checkBoundClaims (:830-865): per node, claim by claim in the pod's order -- the FIRST failure is the node's verdict
            for (size_t i = 0; i < N; i++)
                for (const Value *c : bound) {
                    const auto pv = pvs.find((*c)["spec"]["volumeName"].text());
                    if (pv == pvs.end()) {
                        mark(i, 6);
                        break;
                    }
                    if (!pv_node_affinity_matches(*pv->second, (*nodes[i])["metadata"]["labels"])) {
                        mark(i, 4);
                        break;
                    }
                }
        for (const Value *pvc : delayed) { // binder.go findMatchingVolumes (no volume of the class to match) -> checkVolumeProvisions
            const std::string cname = claim_class(*pvc);
            for (const auto &kv : pvs)
                if ((*kv.second)["spec"]["storageClassName"].text() == cname)
                    throw VolumeUnsupported("persistentvolumeclaim \"" + (*pvc)["metadata"]["name"].text() + "\": matching an unbound claim against the persistent volumes of class \"" +
                                            cname + "\" is not modelled");
            const std::string prov = (*classes[cname])["provisioner"].text();
            if (prov.empty() || prov == kNoProvisioner) mark_all(5);
            else
                throw VolumeUnsupported("persistentvolumeclaim \"" + (*pvc)["metadata"]["name"].text() + "\" waits for its first consumer: StorageClass \"" + cname +
                                        "\" would provision the volume in PreBind, which waits for a PV controller the simulated cluster does not run (the reference does not terminate)");
        }
    }
    if (vo.on("VolumeZone") && !topologies.empty()) // volume_zone.go:191-240
        for (size_t i = 0; i < N; i++) {
            const Value &labels = (*nodes[i])["metadata"]["labels"];
            bool constrained = false;
            for (const char *k : kTopologyLabels) constrained = constrained || labels.has(k);
            if (!constrained) continue; // (a node without any zone label is fine: a single-zone cluster)
            for (const auto &t : topologies) {
                const std::string ga = t.first == kZoneBeta ? kZoneGA : t.first == kRegionBeta ? kRegionGA : t.first;
                const bool ok = labels.has(t.first) ? t.second.count(labels[t.first].text()) : (labels.has(ga) && t.second.count(labels[ga].text()));
                if (!ok) {
                    mark(i, 7);
                    break;
                }
            }
        }
    if (any) out.veto = std::move(veto);
    return out;
}

// The verdicts as DefaultPreemption's dry run sees them on a node once ITS lower-priority pods are removed (default_preemption.go:217-310: per
// node; the PreFilter state follows through RemovePod, volume_restrictions.go:205-213).  Disk conflicts, volume limits and what a bound volume
// says about the node depend on that node alone: the evaluation over the remaining pods.  A ReadWriteOncePod claim in use is a cluster-wide
// count: it stays in conflict on node n unless EVERY pod using the claim is a victim sitting on n.  Empty = no node is rejected then.
inline std::vector<uint8_t> veto_with_victims_gone(const Value &sim_pod, const std::vector<const Value *> &nodes, const std::vector<const Value *> &live,
                                                   const std::vector<size_t> &live_node, const std::vector<uint8_t> &is_victim, const VolumeObjects &vo,
                                                   const VolumeSide &full, size_t clone_index) {
    std::vector<const Value *> rest_live;
    std::vector<size_t> rest_node;
    for (size_t j = 0; j < live.size(); j++)
        if (!is_victim[j]) rest_live.push_back(live[j]), rest_node.push_back(live_node[j]);
    VolumeSide rest = volume_side(sim_pod, nodes, rest_live, rest_node, vo, clone_index, true);
    const size_t N = nodes.size();
    std::vector<uint8_t> veto = rest.veto.empty() ? std::vector<uint8_t>(N, 0) : rest.veto;
    if (std::find(full.veto.begin(), full.veto.end(), (uint8_t)2) != full.veto.end()) { // (the claim is in use by some pod of the snapshot)
        const std::string ns = sim_pod["metadata"]["namespace"].truthy() ? sim_pod["metadata"]["namespace"].text() : "default";
        std::set<std::string> rwop;
        for (const auto &v : sim_pod["spec"]["volumes"].items()) {
            if (v["persistentVolumeClaim"].is_null()) continue;
            const std::string name = v["persistentVolumeClaim"]["claimName"].text();
            for (const auto &o : vo.claims)
                if (o["metadata"]["name"].text() == name && (o["metadata"]["namespace"].truthy() ? o["metadata"]["namespace"].text() : "default") == ns)
                    for (const auto &m : o["spec"]["accessModes"].items())
                        if (m.text() == "ReadWriteOncePod") rwop.insert(name);
        }
        // the reference's arithmetic (volume_restrictions.go:70-84, 219-232): PreFilter counts ONE reference per claim of the pod that is in use
        // (IsPVCUsedByPods, by namespace/name); RemovePod subtracts one for every volume of the removed pod whose claimName is in the pod's
        // set -- by NAME only, whatever the victim's namespace; the node is rejected while the count is above zero (:282-291)
        std::set<std::string> used_here; // the pod's ReadWriteOncePod claims some pod of ITS namespace uses
        for (size_t j = 0; j < live.size(); j++) {
            const Value &p = *live[j];
            if ((p["metadata"]["namespace"].truthy() ? p["metadata"]["namespace"].text() : "default") != ns) continue;
            for (const auto &v : p["spec"]["volumes"].items())
                if (!v["persistentVolumeClaim"].is_null() && rwop.count(v["persistentVolumeClaim"]["claimName"].text())) used_here.insert(v["persistentVolumeClaim"]["claimName"].text());
        }
        const long count0 = (long)used_here.size();
        std::vector<long> released(N, 0);
        for (size_t j = 0; j < live.size(); j++) {
            if (!is_victim[j]) continue;
            for (const auto &v : (*live[j])["spec"]["volumes"].items())
                if (!v["persistentVolumeClaim"].is_null() && rwop.count(v["persistentVolumeClaim"]["claimName"].text())) released[live_node[j]] += 1;
        }
        for (size_t i = 0; i < N; i++)
            if (count0 - released[i] > 0 && veto[i] != 1) veto[i] = 2;
    }
    bool any = false;
    for (uint8_t x : veto) any = any || x;
    return any ? veto : std::vector<uint8_t>();
}

} // namespace cchost
