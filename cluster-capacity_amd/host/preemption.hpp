// preemption.hpp -- DefaultPreemption's dry run for the terminal cycle: host side, report only.
//
// When no node passes the filters the scheduler runs the PostFilter plugins before it reports the pod Unschedulable
// (S/schedule_one.go:186-204); the reference stops on that report whatever the outcome (pkg/framework/simulator.go:327-342),
// so preemption never changes the COUNT -- it decides the tail of the FitError message:
//   no candidate anywhere   " preemption: 0/N nodes are available: <histogram>."  (S/framework/preemption/preemption.go:266-279,
//                           prefix: P/defaultpreemption/default_preemption.go:131-141)
//   a candidate exists      the plugin nominates a node, Success with an empty message: no tail (preemption.go:281-303;
//                           FitError.Error appends PostFilterMsg only when non-empty, S/framework/types.go:831-834)
//   preemptionPolicy=Never  " preemption: not eligible due to preemptionPolicy=Never."  (default_preemption.go:355-357)
// The dry run (preemption.go:741-794, default_preemption.go:217-310) visits the nodes whose filter status is plain
// Unschedulable, removes every pod of lower priority than the incoming one from a copy of the node and runs the Filter plugins
// again: no such pod -> "No preemption victims found for incoming pod"; still failing -> that status; passing -> a candidate.
// Unresolvable nodes are not tried ("Preemption is not helpful for scheduling").  The dry run's random offset and candidate
// cap (default_preemption.go:186-205) do not reach the message: one candidate empties the tail, none means every node was seen.
//
// Clones have the template's priority: only pods of the snapshot can be victims (snapshot.hpp: PodSide::victim_*).  Removing a
// victim changes the node's Requested, pod count and used host ports; with topology-coupled FILTERS (hard spread constraints,
// required inter-pod (anti)affinity, existing pods' anti-affinity) it would also change those plugins' PreFilter state
// (RunPreFilterExtensionRemovePod): not modelled -> Unmodelled, the caller says so and keeps the no-victims form.
#pragma once
#include <cstdint>
#include <vector>

#include "../../include/ccsim.h"
#include "snapshot.hpp"

namespace cchost {

struct PreemptionOutcome {
    enum Kind { None, Nominated, Never, Unmodelled } kind = None;
    std::vector<int64_t> hist = std::vector<int64_t>((size_t)CCSIM_NREASON, 0); // dry-run nodes that still fail: their reasons
    int64_t no_victims = 0;                                                     // potential nodes without a lower-priority pod
    int64_t not_helpful = 0;                                                    // nodes that were UnschedulableAndUnresolvable
};

// NodeUnschedulable, TaintToleration, NodeAffinity for node i (the plugins before NodePorts / Fit in the default order; all
// UnschedulableAndUnresolvable).  NodeName: generated pods never set spec.nodeName.
inline bool preemption_static_ok(const Snapshot &s, const PodSide &p, size_t i, uint32_t fm) {
    if ((fm & CCSIM_F_UNSCHEDULABLE) && !p.tolerates_unschedulable && s.unschedulable[i]) return false;
    if ((fm & CCSIM_F_TAINT) && !p.taint_filter_ok[(size_t)s.taintset_id[i]]) return false;
    if ((fm & CCSIM_F_NODEAFFINITY) && p.affinity_filter_active) {
        auto term = [&](const Term &t, bool empty) {
            if (t.empty()) return empty;
            for (const auto &r : t)
                if (!r.table[(size_t)s.label_cols[(size_t)r.col][i]]) return false;
            return true;
        };
        if (p.has_node_selector && !term(p.node_selector, true)) return false;
        if (p.has_required_terms) {
            bool any = false;
            for (const auto &t : p.required) any = any || term(t, false);
            if (!any) return false;
        }
    }
    return true;
}

inline PreemptionOutcome preemption_dry_run(const Snapshot &s, const PodSide &p, const std::vector<int32_t> &per_node_count, int64_t n_code_unschedulable,
                                            uint32_t fm, size_t n_templates = 1, bool mixed_priorities = false) {
    const size_t N = s.n(), R = s.res_names.size();
    PreemptionOutcome out;
    out.no_victims = n_code_unschedulable, out.not_helpful = (int64_t)N - n_code_unschedulable;
    if (p.preempt_never) return out.kind = PreemptionOutcome::Never, out;
    if (mixed_priorities) return out.kind = PreemptionOutcome::Unmodelled, out;
    if (p.victim_count.empty()) return out;
    bool coupled = false;
    if (fm & CCSIM_F_TOPOLOGYSPREAD)
        for (const auto &c : p.spread) coupled = coupled || c.hard;
    if ((fm & CCSIM_F_INTERPODAFFINITY) && p.has_ipa) {
        coupled = coupled || !p.ipa.aff_keys.empty() || !p.ipa.anti_keys.empty();
        for (const auto &v : p.ipa.exist_anti) coupled = coupled || !v.empty();
    }
    if (n_templates > 1 || coupled) return out.kind = PreemptionOutcome::Unmodelled, out;

    bool all_zero = !p.has_scalar_entries; // fit.go:578-583
    for (size_t c = 0; c < 3 && c < R; c++) all_zero = all_zero && !(p.preq[c] > 0);
    const bool ports_on = (fm & CCSIM_F_NODEPORTS) && p.has_host_ports;
    // fitsRequest (noderesources/fit.go:564-660): bit 0 too many pods | bit 1+c insufficient column c; beyond = request > allocatable
    auto fit = [&](size_t i, const std::vector<int64_t> &used, int64_t pods, bool &beyond) {
        uint32_t mask = 0;
        beyond = false;
        if (!(fm & CCSIM_F_FIT)) return mask;
        if (pods + 1 > (int64_t)s.alloc_pods[i]) mask |= 1u;
        if (!all_zero)
            for (size_t c = 0; c < R; c++) {
                const int64_t rq = p.preq[c];
                if (rq == 0) continue;
                if (rq > s.alloc[c][i] - used[c]) mask |= 1u << (1 + c), beyond = beyond || rq > s.alloc[c][i];
            }
        return mask;
    };
    int64_t potential = 0;
    std::vector<int64_t> used(R);
    for (size_t i = 0; i < N; i++) {
        if (!p.victim_count[i] || !preemption_static_ok(s, p, i, fm)) continue;
        const int64_t cnt = i < per_node_count.size() ? per_node_count[i] : 0;
        const bool conflict_now = ports_on && (cnt > 0 || (!p.host_ports_conflict.empty() && p.host_ports_conflict[i]));
        const bool conflict_rest = ports_on && (cnt > 0 || (!p.ports_conflict_rest.empty() && p.ports_conflict_rest[i]));
        for (size_t c = 0; c < R; c++) used[c] = s.req[c][i] + cnt * p.preq[c]; // the terminal NodeInfo (types.go:409-428)
        const int64_t pods = (int64_t)s.pod_count[i] + cnt;
        bool beyond = false;
        const uint32_t m0 = fit(i, used, pods, beyond);
        if (!(conflict_now || (m0 && !beyond))) continue; // the terminal status was not plain Unschedulable: not a dry-run node
        potential++;
        for (size_t c = 0; c < R; c++) used[c] -= p.victim_req[c][i]; // ... with the victims gone
        const uint32_t m1 = conflict_rest ? 0 : fit(i, used, pods - p.victim_count[i], beyond);
        if (!conflict_rest && !m1) return out.kind = PreemptionOutcome::Nominated, out;
        if (conflict_rest) out.hist[CCSIM_R_NODEPORTS]++; // the first failing plugin's reasons (framework.go:897-930)
        else {
            if (m1 & 1u) out.hist[CCSIM_R_TOO_MANY_PODS]++;
            for (size_t c = 0; c < R; c++)
                if (m1 & (1u << (1 + c))) out.hist[CCSIM_R_RES0 + c]++;
        }
    }
    out.no_victims = n_code_unschedulable - potential;
    return out;
}

} // namespace cchost
