"""DefaultPreemption's dry run for the terminal cycle -- host side, report only.

When no node passes the filters the scheduler runs the PostFilter plugins before it reports the pod Unschedulable
(S/schedule_one.go:186-204); the reference stops on that report whatever the outcome (pkg/framework/simulator.go:327-342),
so preemption never changes the COUNT -- it decides the tail of the FitError message:

* no candidate anywhere  -> " preemption: 0/N nodes are available: <histogram>."   (S/framework/preemption/preemption.go:266-279;
  "preemption: " prefix: P/defaultpreemption/default_preemption.go:131-141)
* a candidate exists     -> the plugin nominates a node and returns Success with an empty message: no tail at all
  (preemption.go:281-303; FitError.Error appends PostFilterMsg only when non-empty, S/framework/types.go:831-834)
* preemptionPolicy=Never -> " preemption: not eligible due to preemptionPolicy=Never."  (default_preemption.go:355-357)

The dry run (preemption.go:741-794, default_preemption.go:217-310 SelectVictimsOnNode) looks at the nodes whose filter status
is plain Unschedulable, removes every pod of lower priority than the incoming one from a COPY of the node and runs the Filter
plugins again: no such pod -> "No preemption victims found for incoming pod"; still failing -> that filter status; passing ->
a candidate (reprieving victims afterwards never empties the victim list of a node that failed with them all present).
Nodes that failed UnschedulableAndUnresolvable are not tried: "Preemption is not helpful for scheduling".  The random offset
and the candidate cap of the dry run (default_preemption.go:186-205) do not matter to the message: with at least one candidate
the tail is empty, with none every potential node was visited.

Clones have the template's priority, so only pods of the snapshot can be victims (ingest: PreemptionSide).  Removing a victim
changes the node's Requested, pod count and used host ports; with topology-coupled FILTERS in play (hard spread constraints,
required inter-pod (anti)affinity, existing pods' anti-affinity) it would also change the plugins' PreFilter state
(RunPreFilterExtensionRemovePod) -- that case is not modelled: kind = "unmodelled", the caller says so and keeps the
no-victims form of the message.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

from . import model as M


@dataclass
class Outcome:
    kind: str = "none"  # "none" (no candidate: histogram tail) | "nominated" (no tail) | "never" | "unmodelled" (treated as "none")
    hist: np.ndarray = field(default_factory=lambda: np.zeros(M.NREASON, np.int64))  # dry-run nodes that still fail: their reasons
    no_victims: int = 0   # potential nodes without a lower-priority pod
    not_helpful: int = 0  # nodes whose status was UnschedulableAndUnresolvable


def static_ok(nodes: M.NodesSoA, pod: M.PodSpec, idx: np.ndarray, filter_mask: int) -> np.ndarray:
    """NodeUnschedulable, TaintToleration and NodeAffinity for the nodes `idx` (the plugins before NodePorts / Fit in the
    default order, all UnschedulableAndUnresolvable when they fail).  NodeName: generated pods never set spec.nodeName."""
    ok = np.ones(len(idx), bool)
    if filter_mask & M.F_UNSCHEDULABLE and not pod.tolerates_unschedulable:
        ok &= nodes.unschedulable[idx] == 0
    if filter_mask & M.F_TAINT:
        ok &= pod.taint_filter_ok[nodes.taintset_id[idx]] != 0
    if filter_mask & M.F_NODEAFFINITY and pod.affinity_filter_active:
        def term(reqs, empty):
            if not reqs:
                return np.full(len(idx), empty)
            m = np.ones(len(idx), bool)
            for col, table in reqs:
                m &= np.asarray(table)[nodes.label_cols[col][idx]] != 0
            return m
        if pod.has_node_selector:
            ok &= term(pod.node_selector, True)
        if pod.has_required_terms:
            anyt = np.zeros(len(idx), bool)
            for t in pod.required:
                anyt |= term(t, False)
            ok &= anyt
    return ok


def _coupled_filters(pod: M.PodSpec, filter_mask: int) -> bool:
    if filter_mask & M.F_TOPOLOGYSPREAD and any(c.hard for c in pod.spread):
        return True
    a = pod.ipa
    if filter_mask & M.F_INTERPODAFFINITY and a is not None and (a.aff_keys or a.anti_keys or any(x is not None for x in a.exist_anti)):
        return True
    return False


def dry_run(nodes: M.NodesSoA, pod: M.PodSpec, per_node_count, n_code_unschedulable: int, filter_mask: int = M.F_ALL,
            n_templates: int = 1, mixed_priorities: bool = False) -> Outcome:
    """`nodes`: the snapshot as loaded (before the run); `per_node_count`: clones per node at the terminal cycle.  Several
    templates: modelled only when no pod anywhere can be a victim (same priority everywhere, no lower-priority pod in the
    snapshot) -- the terminal node state is then irrelevant."""
    n = nodes.n
    pre = pod.preempt or M.PreemptionSide()
    out = Outcome(no_victims=int(n_code_unschedulable), not_helpful=int(n - n_code_unschedulable))
    if pre.never:
        out.kind = "never"
        return out
    if mixed_priorities:
        out.kind = "unmodelled"
        return out
    if pre.victim_count is None or not pre.victim_count.any():
        return out
    if n_templates > 1 or _coupled_filters(pod, filter_mask):
        out.kind = "unmodelled"
        return out
    idx = np.nonzero(pre.victim_count)[0]
    cnt = np.asarray(per_node_count, np.int64)[idx]
    sok = static_ok(nodes, pod, idx, filter_mask)
    ncol = len(nodes.alloc)
    all_zero = not (pod.req[:3] > 0).any() and not pod.has_scalar_entries  # fit.go:578-583 (requests are never negative)

    def fit(req_cols, pods):
        """fitsRequest (noderesources/fit.go:564-660): -> (bit 0 too many pods | bit 1+c insufficient column c, request > allocatable)"""
        mask = np.where(pods + 1 > nodes.alloc_pods[idx], 1, 0).astype(np.int64)
        beyond = np.zeros(len(idx), bool)
        if filter_mask & M.F_FIT and not all_zero:
            for c in range(ncol):
                rq = int(pod.req[c])
                if rq == 0:
                    continue
                short = rq > nodes.alloc[c][idx] - req_cols[c]
                mask |= np.where(short, 1 << (1 + c), 0)
                beyond |= short & (rq > nodes.alloc[c][idx])
        if not filter_mask & M.F_FIT:
            mask[:] = 0
        return mask, beyond

    ports_on = bool(filter_mask & M.F_NODEPORTS and pod.has_host_ports)
    conflict_now = np.zeros(len(idx), bool)
    conflict_rest = np.zeros(len(idx), bool)
    if ports_on:
        conflict_now = cnt > 0
        if pod.host_ports_conflict is not None:
            conflict_now = conflict_now | (pod.host_ports_conflict[idx] != 0)
        conflict_rest = cnt > 0
        if pre.ports_conflict_rest is not None:
            conflict_rest = conflict_rest | (pre.ports_conflict_rest[idx] != 0)
    # the terminal cycle's status code of these nodes: plain Unschedulable = a dry-run node
    term_req = [nodes.req[c][idx] + cnt * int(pod.req[c]) for c in range(ncol)]
    term_pods = nodes.pod_count[idx].astype(np.int64) + cnt
    m0, beyond0 = fit(term_req, term_pods)
    potential = sok & (conflict_now | ((m0 != 0) & ~beyond0))
    # ... with the victims gone
    m1, _ = fit([term_req[c] - pre.victim_req[c][idx] for c in range(ncol)], term_pods - pre.victim_count[idx])
    fits = potential & ~conflict_rest & (m1 == 0)
    if fits.any():
        out.kind = "nominated"
        return out
    out.no_victims = int(n_code_unschedulable) - int(potential.sum())
    out.hist[M.R_NODEPORTS] = int((potential & conflict_rest).sum())
    still = potential & ~conflict_rest
    out.hist[M.R_TOO_MANY_PODS] = int((still & ((m1 & 1) != 0)).sum())
    for c in range(ncol):
        out.hist[M.R_RES0 + c] = int((still & ((m1 >> (1 + c)) & 1 != 0)).sum())
    return out
