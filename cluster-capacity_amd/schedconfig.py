"""KubeSchedulerConfiguration (--default-config, cmd/cluster-capacity/app/options/options.go:73, server.go:106-113)
-> model.Profile.  Python mirror of cluster-capacity_amd/host/profile.hpp (same semantics, checked against it in
tests/test_native_host.py): which of the engine's Filter plugins run, the Score plugins' weights, the resource lists of
NodeResourcesFit (LeastAllocated) / NodeResourcesBalancedAllocation, percentageOfNodesToScore (global or per profile,
schedule_one.go:702-708), InterPodAffinity's hardPodAffinityWeight.  Merge semantics follow the scheduler's
(S/apis/config/v1/default_plugins.go:30-58,77-140; S/framework/runtime/framework.go:506-640): multiPoint applies to every
extension point a plugin implements, `disabled: [{name: "*"}]` clears a point, a specific point overrides multiPoint, a
Score plugin enabled without a weight gets 1."""
from __future__ import annotations

import dataclasses
from typing import Optional, Tuple

from . import model as M

# name -> (filter bit, Profile weight field, default weight)
PLUGINS = {
    "NodeUnschedulable": (M.F_UNSCHEDULABLE, None, 0),
    "NodeName": (M.F_NODENAME, None, 0),
    "TaintToleration": (M.F_TAINT, "w_taint", 3),
    "NodeAffinity": (M.F_NODEAFFINITY, "w_nodeaffinity", 2),
    "NodeResourcesFit": (M.F_FIT, "w_fit", 1),
    "NodeResourcesBalancedAllocation": (0, "w_balanced", 1),
    "PodTopologySpread": (M.F_TOPOLOGYSPREAD, "w_topologyspread", 2),
    "InterPodAffinity": (M.F_INTERPODAFFINITY, "w_interpodaffinity", 2),
    "NodePorts": (M.F_NODEPORTS, None, 0),
    "ImageLocality": (0, "w_imagelocality", 1),
}
# default plugins whose Filter / Score is a no-op for the pods this simulator accepts: accepted, ignored
# (pods that would activate a volume / DRA plugin are refused at ingest)
FOLDED_AWAY = {"SchedulingGates", "PrioritySort", "VolumeRestrictions", "NodeVolumeLimits", "EBSLimits", "GCEPDLimits",
               "AzureDiskLimits", "VolumeBinding", "VolumeZone", "DynamicResources", "DefaultPreemption",
               "DefaultBinder", "ClusterCapacityBinder"}
# ... of which these four are evaluated on the host for pods with volumes (volumes.py): disabling one under multiPoint takes it out;
# with only its filter point disabled a pod with volumes is refused (the PreFilter half would still run in the scheduler)
VOLUME_PLUGINS = ("VolumeRestrictions", "NodeVolumeLimits", "VolumeBinding", "VolumeZone")


class ConfigError(ValueError):
    pass


def _column(name: str) -> int:
    if name == "cpu":
        return 0
    if name == "memory":
        return 1
    if name == "ephemeral-storage":
        return 2
    # (the engine scores scalar resources too -- columns 3+ of the ABI -- but a config names them before the snapshot's
    # scalar columns exist: not wired through this host)
    raise ConfigError(f"scheduler config: scoring resource '{name}' is not supported by this host (cpu, memory and ephemeral-storage are)")


def sets_percentage(cfg: Optional[dict]) -> bool:
    """Does the configuration file name percentageOfNodesToScore (globally or in Profiles[0])?"""
    if not cfg:
        return False
    profs = cfg.get("profiles") or [{}]
    return cfg.get("percentageOfNodesToScore") is not None or (profs[0] or {}).get("percentageOfNodesToScore") is not None


def profile_from_config(cfg: Optional[dict]) -> Tuple[M.Profile, int]:
    """-> (Profile, hardPodAffinityWeight)."""
    p = dataclasses.asdict(M.Profile.default())
    hard = 1
    system_default_spreading = True  # PodTopologySpreadArgs.defaultingType System (the default)
    volume_plugins = set(VOLUME_PLUGINS)
    volume_partial = []
    dra = [True, False]  # DynamicResources: enabled, only its filter point disabled

    def done():
        out = M.Profile(**p)
        out.system_default_spreading = system_default_spreading  # host-side note (not an ABI field): see ingest.default_spreading_applies
        out.volume_plugins = tuple(n for n in VOLUME_PLUGINS if n in volume_plugins)  # host-side note: which of them volumes.py evaluates
        out.volume_plugins_partial = bool(volume_partial)
        out.dra_enabled, out.dra_partial = dra[0], dra[1]  # host-side notes: a pod with spec.resourceClaims (volumes.dra_prefilter)
        return out, hard
    if not cfg:
        return done()
    if cfg.get("kind") and cfg["kind"] != "KubeSchedulerConfiguration":
        raise ConfigError("scheduler config: kind is not KubeSchedulerConfiguration")
    if cfg.get("percentageOfNodesToScore") is not None:
        p["percentage_of_nodes_to_score"] = int(cfg["percentageOfNodesToScore"])
    profiles = cfg.get("profiles") or []
    if len(profiles) > 1:
        raise ConfigError("scheduler config: one profile only (the simulated pod is scheduled by Profiles[0])")
    prof = (profiles[0] if profiles else None) or {}
    if prof.get("percentageOfNodesToScore") is not None:
        p["percentage_of_nodes_to_score"] = int(prof["percentageOfNodesToScore"])

    def lookup(name):
        if name in PLUGINS:
            return PLUGINS[name]
        if name in FOLDED_AWAY:
            return None
        raise ConfigError(f"scheduler config: unknown plugin '{name}'")

    def set_filter(info, on):
        if info[0]:
            p["filter_mask"] = (p["filter_mask"] | info[0]) if on else (p["filter_mask"] & ~info[0])

    def set_score(info, w):
        if info[1]:
            p[info[1]] = w

    def apply(pset, do_filter, do_score, multipoint):
        pset = pset or {}
        for d in pset.get("disabled") or []:
            name = d.get("name", "")
            if do_filter and name in ("*", "DynamicResources"):
                if multipoint:
                    dra[0] = False
                else:
                    dra[1] = True
            if do_filter and (name == "*" or name in VOLUME_PLUGINS):
                if not multipoint:
                    volume_partial.append(name)  # (its PreFilter would still run: refused when a pod with volumes arrives)
                volume_plugins.difference_update(VOLUME_PLUGINS if name == "*" else (name,))
            infos = list(PLUGINS.values()) if name == "*" else [lookup(name)]
            for info in infos:
                if info is None:
                    continue
                if do_filter:
                    set_filter(info, False)
                if do_score:
                    set_score(info, 0)
        for e in pset.get("enabled") or []:
            if multipoint and e.get("name", "") in VOLUME_PLUGINS:
                volume_plugins.add(e["name"])
            if multipoint and e.get("name", "") == "DynamicResources":
                dra[0] = True
            info = lookup(e.get("name", ""))
            if info is None:
                continue
            if do_filter:
                set_filter(info, True)
            if do_score and info[1]:
                w = int(e.get("weight") or 0)
                if w < 0 or w > 100 * 1000:
                    raise ConfigError(f"scheduler config: bad weight for {e.get('name')}")
                set_score(info, w if w > 0 else (info[2] if multipoint else 1))

    plugins = prof.get("plugins") or {}
    apply(plugins.get("multiPoint"), True, True, True)
    apply(plugins.get("filter"), True, False, False)
    apply(plugins.get("score"), False, True, False)

    for pc in prof.get("pluginConfig") or []:
        name, args = pc.get("name", ""), pc.get("args") or {}
        if name == "NodeResourcesFit":
            st = args.get("scoringStrategy") or {}
            if st:
                typ = st.get("type") or "LeastAllocated"
                if typ != "LeastAllocated":
                    raise ConfigError(f"scheduler config: NodeResourcesFit scoringStrategy {typ} is not implemented (LeastAllocated is)")
                if st.get("resources"):
                    p["fit_res"] = tuple(_column(r["name"]) for r in st["resources"])
                    p["fit_res_w"] = tuple(int(r.get("weight") or 1) for r in st["resources"])
            if args.get("ignoredResources") or args.get("ignoredResourceGroups"):
                raise ConfigError("scheduler config: NodeResourcesFit ignoredResources are not implemented")
        elif name == "NodeResourcesBalancedAllocation":
            if args.get("resources"):
                p["bal_res"] = tuple(_column(r["name"]) for r in args["resources"])
        elif name == "InterPodAffinity":
            if "hardPodAffinityWeight" in args:
                hard = int(args["hardPodAffinityWeight"] or 0)
            if args.get("ignorePreferredTermsOfExistingPods"):
                raise ConfigError("scheduler config: ignorePreferredTermsOfExistingPods is not implemented")
        elif name == "PodTopologySpread":
            if args.get("defaultConstraints"):
                raise ConfigError("scheduler config: PodTopologySpread defaultConstraints are not implemented")
            if args.get("defaultingType") == "List":  # an empty list: no default constraints at all (plugin.go:105-113)
                system_default_spreading = False
        elif name == "NodeAffinity":
            if args.get("addedAffinity"):
                raise ConfigError("scheduler config: NodeAffinity addedAffinity is not implemented")
        elif name not in FOLDED_AWAY and name not in PLUGINS:
            raise ConfigError(f"scheduler config: pluginConfig for unknown plugin '{name}'")
    if not 0 <= p["percentage_of_nodes_to_score"] <= 100:
        raise ConfigError("scheduler config: percentageOfNodesToScore out of [0,100] (validation.go:86-90)")
    return done()
