"""Stop-reason / report formatting: Python mirror of the strings the reference produces.

* FitError.Error                    vendor/k8s.io/kubernetes/pkg/scheduler/framework/types.go:787-836
* DefaultPreemption PostFilter msg  .../plugins/defaultpreemption/default_preemption.go:131-141,257
                                    .../framework/preemption/preemption.go:266-279
* StopReason / getMainFailReason    pkg/framework/simulator.go:297-342, pkg/framework/report.go:100-109
* parsePodsReview                   pkg/framework/report.go:146-180

Inputs are the integer histograms returned through the C-ABI; strings live only here.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import numpy as np

from . import model as M

RES_NAMES = ["cpu", "memory", "ephemeral-storage"]

REASON_TEXT = {
    M.R_UNSCHEDULABLE: "node(s) were unschedulable",
    M.R_NODENAME: "node(s) didn't match the requested node name",
    M.R_NODEAFFINITY: "node(s) didn't match Pod's node affinity/selector",
    M.R_TOO_MANY_PODS: "Too many pods",
    M.R_PTS_MISSING_LABEL: "node(s) didn't match pod topology spread constraints (missing required label)",
    M.R_PTS_SKEW: "node(s) didn't match pod topology spread constraints",
    # interpodaffinity/filtering.go:37-45
    M.R_IPA_AFFINITY: "node(s) didn't match pod affinity rules",
    M.R_IPA_ANTI: "node(s) didn't match pod anti-affinity rules",
    M.R_IPA_EXISTING_ANTI: "node(s) didn't satisfy existing pods anti-affinity rules",
    M.R_NODEPORTS: "node(s) didn't have free ports for the requested pod ports",  # nodeports/node_ports.go:39
    # volumerestrictions/volume_restrictions.go:57-59, nodevolumelimits/csi.go:44, volumebinding/binder.go:65-71, volumezone/volume_zone.go:61
    M.R_VOL0 + M.VOL_DISK_CONFLICT - 1: "node(s) had no available disk",
    M.R_VOL0 + M.VOL_RWOP - 1: "node(s) unavailable due to PersistentVolumeClaim with ReadWriteOncePod access mode already in-use by another pod",
    M.R_VOL0 + M.VOL_MAX_COUNT - 1: "node(s) exceed max volume count",
    M.R_VOL0 + M.VOL_NODE_AFFINITY - 1: "node(s) didn't match PersistentVolume's node affinity",
    M.R_VOL0 + M.VOL_NO_PV - 1: "node(s) didn't find available persistent volumes to bind",
    M.R_VOL0 + M.VOL_PV_NOT_EXIST - 1: "node(s) unavailable due to one or more pvc(s) bound to non-existent pv(s)",
    M.R_VOL0 + M.VOL_ZONE - 1: "node(s) had no available volume zone",
}


def _histogram_message(reasons: Dict[str, int]) -> str:
    # types.go:820-829: "<count> <reason>" strings sorted lexicographically, joined by ", "
    items = sorted(f"{v} {k}" for k, v in reasons.items() if v)
    return ", ".join(items)


def _reason_histogram(hist, hist_taintset, taint_reasons, scalar_names) -> Dict[str, int]:
    reasons: Dict[str, int] = {}
    for slot, cnt in enumerate(hist):
        cnt = int(cnt)
        if not cnt:
            continue
        if slot in REASON_TEXT:
            text = REASON_TEXT[slot]
        elif M.R_RES0 <= slot < M.R_RES0 + M.MAX_RES:
            c = slot - M.R_RES0
            name = RES_NAMES[c] if c < 3 else (scalar_names[c - 3] if c - 3 < len(scalar_names) else f"scalar-{c - 3}")
            text = f"Insufficient {name}"
        else:
            text = f"reason-{slot}"
        reasons[text] = reasons.get(text, 0) + cnt
    for ts, cnt in enumerate(hist_taintset):
        cnt = int(cnt)
        if not cnt:
            continue
        # taint_toleration.go:119: "node(s) had untolerated taint {key: value}" (first untolerated taint)
        text = taint_reasons[ts] if taint_reasons is not None else f"node(s) had untolerated taint {{taintset-{ts}}}"
        reasons[text] = reasons.get(text, 0) + cnt
    return reasons


def fit_error_message(
    n_nodes: int,
    hist: Sequence[int],
    hist_taintset: Sequence[int],
    n_code_unschedulable: int,
    taint_reasons: Optional[Sequence[str]] = None,
    scalar_names: Sequence[str] = (),
    with_preemption: bool = True,
    preemption=None,
    prefilter_msg: Optional[str] = None,
) -> str:
    """FitError.Error() for the terminal round, including the DefaultPreemption tail.  `preemption`: the outcome of the dry
    run (preemption.Outcome); None = no node holds a pod of lower priority than the simulated one."""
    msg = f"0/{n_nodes} nodes are available:"
    if prefilter_msg:
        # a PreFilter plugin rejected the pod (schedule_one.go:495-508): its message stands for every node (types.go:789-794), and every
        # node is UnschedulableAndUnresolvable for the preemption that follows (n_code_unschedulable = 0)
        msg += f" {prefilter_msg}."
    else:
        body = _histogram_message(_reason_histogram(hist, hist_taintset, taint_reasons, scalar_names))
        if body:
            msg += f" {body}."
    if not with_preemption or (preemption is not None and preemption.kind == "nominated"):
        return msg  # a candidate node was found: PostFilter returns Success with an empty message (preemption.go:281-303)
    if preemption is not None and preemption.kind == "never":
        return msg + " preemption: not eligible due to preemptionPolicy=Never."  # default_preemption.go:355-357
    # Nodes that failed with plain Unschedulable are dry-run candidates: without a lower-priority pod on them each reports
    # "No preemption victims found"; with victims that do not help, the filter status after their removal; the rest are
    # absent from the map -> "Preemption is not helpful for scheduling".
    no_victims = int(n_code_unschedulable) if preemption is None else preemption.no_victims
    pre = {} if preemption is None else _reason_histogram(preemption.hist, (), None, scalar_names)
    if no_victims:
        pre["No preemption victims found for incoming pod"] = no_victims
    if n_nodes - n_code_unschedulable > 0:
        pre["Preemption is not helpful for scheduling"] = int(n_nodes - n_code_unschedulable)
    pmsg = f"0/{n_nodes} nodes are available:"
    pbody = _histogram_message(pre)
    if pbody:
        pmsg += f" {pbody}."
    return msg + " preemption: " + pmsg


def stop_reason(result, n_nodes: int, max_limit: int, **kw) -> str:
    """ClusterCapacity.Status.StopReason (simulator.go:301,331)."""
    if result.stop == M.STOP_LIMIT:
        return f"LimitReached: Maximum number of pods simulated: {max_limit}"
    if result.stop == M.STOP_NO_NODES:
        return "Unschedulable: no nodes available to schedule pods"
    return "Unschedulable: " + fit_error_message(n_nodes, result.hist, result.hist_taintset, result.n_code_unschedulable,
                                                 prefilter_msg=getattr(result, "prefilter_msg", None), **kw)


def main_fail_reason(message: str):
    """report.go:100-109 getMainFailReason."""
    first = message.split("\n")[0]
    colon = first.index(":")
    return {"failType": first[:colon], "failMessage": first[colon + 1 :].strip(" ")}


def replicas_on_nodes(per_node_count: np.ndarray, names: Optional[List[str]] = None, log: Optional[np.ndarray] = None):
    """report.go:146-180: per-node replica counts; first-placement order when a log is available.  The placement
    log may be capped below the number of placements (ccsim_report.log_cap): nodes first placed beyond the cap follow
    in canonical node order, so the list always covers every node with a replica and sums to the headline count."""
    idx = np.nonzero(per_node_count)[0]
    if log is not None and len(log):
        _, first = np.unique(log, return_index=True)
        order = log[np.sort(first)]
        if len(order) < len(idx):
            order = np.concatenate([order, np.setdiff1d(idx, order)])
    else:
        order = idx
    return [{"nodeName": names[i] if names else str(int(i)), "replicas": int(per_node_count[i])} for i in order]
