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
changes the node's Requested, pod count and used host ports.  With topology-coupled FILTERS in play (hard spread constraints,
required inter-pod (anti)affinity, existing pods' anti-affinity) the second Filter run also evaluates those plugins against the
cycle's PreFilter state, which the removal changes only through victims that take part in it (RunPreFilterExtensionRemovePod:
the victim matches a hard spread selector or a required (anti)affinity term of the template, or carries an anti-affinity term
matching it).  The usual victim -- a placeholder pod with labels of its own -- does not: then the state is the terminal cycle's,
rebuilt here from the snapshot and the clone counts (CoupledState below: podtopologyspread/filtering.go:235-356,
interpodaffinity/filtering.go:204-432, as oracle/ccref.c states them), and the dry run is exact.  A potential node whose victims
DO take part is not modelled: kind = "unmodelled", the caller says so and keeps the no-victims form of the message.
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


MAXINT32 = 2147483647


class CoupledState:
    """The PreFilter state of PodTopologySpread (hard constraints) and InterPodAffinity at the terminal cycle: per-domain tables over
    ALL nodes from the snapshot's per-node counts and the clones per node (a clone is an existing pod of the next cycle)."""

    def __init__(self, nodes: M.NodesSoA, pod: M.PodSpec, per_node_count, filter_mask: int):
        self.nodes, self.pod = nodes, pod
        n = nodes.n
        clones = np.asarray(per_node_count, np.int64)[:n]
        self.hard = [c for c in pod.spread if c.hard] if filter_mask & M.F_TOPOLOGYSPREAD else []
        hard_keys = np.ones(n, bool)
        for c in self.hard:
            hard_keys &= nodes.label_cols[c.col] != 0
        self.match, self.min_eff = [], []
        for c in self.hard:  # pts_prefilter: TpValueToMatchNum over the counted nodes, the global minimum (0 below minDomains)
            dom = nodes.label_cols[c.col].astype(np.int64)
            counted = hard_keys & (np.ones(n, bool) if c.node_included is None else c.node_included != 0)
            val = (np.zeros(n, np.int64) if c.node_match_count is None else c.node_match_count.astype(np.int64)) + (clones if c.self_match else 0)
            size = int(dom.max()) + 1 if n else 1
            tab = np.bincount(dom[counted], weights=val[counted], minlength=size).astype(np.int64)
            present = np.bincount(dom[counted], minlength=size) > 0
            self.match.append(tab)
            mn = int(tab[present].min()) if present.any() else MAXINT32
            self.min_eff.append(0 if int(present.sum()) < c.min_domains else mn)
        a = self.ipa = pod.ipa if (filter_mask & M.F_INTERPODAFFINITY) else None
        self.ipa_active = False
        if a is not None:  # ipa_build: the count maps per topology key
            K = len(a.key_cols)
            doms = [nodes.label_cols[col].astype(np.int64) for col in a.key_cols]
            size = [int(d.max()) + 1 if n else 1 for d in doms]
            z = np.zeros(n, np.int64)
            self.aff = [np.zeros(size[k], np.int64) for k in range(K)]
            self.anti = [np.zeros(size[k], np.int64) for k in range(K)]
            self.exist = [np.zeros(size[k], np.int64) for k in range(K)]
            am = (z if a.aff_existing is None else a.aff_existing.astype(np.int64)) + (clones if a.self_aff else 0)
            self.aff_total = 0
            for k in a.aff_keys:
                has = doms[k] != 0
                self.aff[k] += np.bincount(doms[k][has], weights=am[has], minlength=size[k]).astype(np.int64)
                self.aff_total += int(am[has].sum())
            for t, k in enumerate(a.anti_keys):
                m = (z if a.anti_existing[t] is None else a.anti_existing[t].astype(np.int64)) + (clones if a.anti_self[t] else 0)
                has = doms[k] != 0
                self.anti[k] += np.bincount(doms[k][has], weights=m[has], minlength=size[k]).astype(np.int64)
            self.exist_total = 0
            for k in range(K):
                m = (z if a.exist_anti[k] is None else a.exist_anti[k].astype(np.int64)) + clones * sum(1 for t, kk in enumerate(a.anti_keys) if kk == k and a.anti_self[t])
                has = doms[k] != 0
                self.exist[k] += np.bincount(doms[k][has], weights=m[has], minlength=size[k]).astype(np.int64)
                self.exist_total += int(m[has].sum())
            self.doms = doms
            self.ipa_active = not (self.exist_total == 0 and not a.aff_keys and not a.anti_keys)

    @property
    def active(self) -> bool:
        return bool(self.hard) or self.ipa_active

    def verdict(self, i: int):
        """The coupled filters on node i -> None (passes) or (reason slot, unresolvable)."""
        for c, tab, mn in zip(self.hard, self.match, self.min_eff):  # pts_filter
            v = int(self.nodes.label_cols[c.col][i])
            if v == 0:
                return M.R_PTS_MISSING_LABEL, True
            if int(tab[v]) + (1 if c.self_match else 0) - mn > c.max_skew:
                return M.R_PTS_SKEW, False
        if self.ipa_active:  # ipa_filter
            a = self.ipa
            pods_exist = True
            for k in a.aff_keys:
                v = int(self.doms[k][i])
                if v == 0:
                    return M.R_IPA_AFFINITY, True
                if self.aff[k][v] <= 0:
                    pods_exist = False
            if not pods_exist and not (self.aff_total == 0 and a.self_aff):
                return M.R_IPA_AFFINITY, True
            for k in a.anti_keys:
                v = int(self.doms[k][i])
                if v and self.anti[k][v] > 0:
                    return M.R_IPA_ANTI, False
            if self.exist_total > 0:
                for k in range(len(a.key_cols)):
                    v = int(self.doms[k][i])
                    if v and self.exist[k][v] > 0:
                        return M.R_IPA_EXISTING_ANTI, False
        return None


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
    if n_templates > 1:
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
    # the coupled filters run after NodePorts / Fit, against the terminal cycle's PreFilter state (unchanged by the removal of
    # victims that take no part in it): one verdict per victim-bearing node
    coupled = CoupledState(nodes, pod, per_node_count, filter_mask)
    verdicts = [coupled.verdict(int(i)) for i in idx] if coupled.active else [None] * len(idx)
    c_fail = np.array([v is not None for v in verdicts], bool)
    c_unres = np.array([v is not None and v[1] for v in verdicts], bool)
    # the terminal cycle's status code of these nodes: plain Unschedulable = a dry-run node
    term_req = [nodes.req[c][idx] + cnt * int(pod.req[c]) for c in range(ncol)]
    term_pods = nodes.pod_count[idx].astype(np.int64) + cnt
    m0, beyond0 = fit(term_req, term_pods)
    # the volume plugins follow NodeResourcesFit (default_plugins.go:40-45): a node that holds a clone whose disks conflict with the next
    # one's, else the hosts' verdict against the node's pods -- now, and with the victims gone (volumes.veto_with_victims_gone)
    vol_now = np.zeros(len(idx), np.int64) if pod.volume_veto is None else np.asarray(pod.volume_veto, np.int64)[idx]
    vol_rest = np.zeros(len(idx), np.int64) if pre.volume_veto_rest is None else np.asarray(pre.volume_veto_rest, np.int64)[idx]
    if pod.volume_exclusive:
        vol_now, vol_rest = np.where(cnt > 0, M.VOL_DISK_CONFLICT, vol_now), np.where(cnt > 0, M.VOL_DISK_CONFLICT, vol_rest)
    local_fail = conflict_now | (m0 != 0)
    vol_fail = ~local_fail & (vol_now != 0)
    potential = sok & np.where(local_fail, conflict_now | ~beyond0, np.where(vol_fail, vol_now <= M.VOL_LAST_UNSCHEDULABLE, c_fail & ~c_unres))
    if coupled.active and pre.victim_interacts is not None and (potential & (pre.victim_interacts[idx] != 0)).any():
        out.kind = "unmodelled"
        return out
    # ... with the victims gone
    m1, _ = fit([term_req[c] - pre.victim_req[c][idx] for c in range(ncol)], term_pods - pre.victim_count[idx])
    fits = potential & ~conflict_rest & (m1 == 0) & (vol_rest == 0) & ~c_fail
    if fits.any():
        out.kind = "nominated"
        return out
    out.no_victims = int(n_code_unschedulable) - int(potential.sum())
    out.hist[M.R_NODEPORTS] = int((potential & conflict_rest).sum())
    still = potential & ~conflict_rest
    out.hist[M.R_TOO_MANY_PODS] = int((still & ((m1 & 1) != 0)).sum())
    for c in range(ncol):
        out.hist[M.R_RES0 + c] = int((still & ((m1 >> (1 + c)) & 1 != 0)).sum())
    for code in range(1, M.VOL_CODES + 1):
        out.hist[M.R_VOL0 + code - 1] = int((still & (m1 == 0) & (vol_rest == code)).sum())
    for j in np.nonzero(still & (m1 == 0) & (vol_rest == 0))[0]:  # node-locally fine now: the coupled filter's reason (the first failing plugin's)
        out.hist[verdicts[j][0]] += 1
    return out
