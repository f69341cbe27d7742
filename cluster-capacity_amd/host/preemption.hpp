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
// victim changes the node's Requested, pod count and used host ports.  With topology-coupled FILTERS (hard spread constraints,
// required inter-pod (anti)affinity, existing pods' anti-affinity) the second Filter run also evaluates those plugins against the
// cycle's PreFilter state, which the removal changes only through victims that take part in it (RunPreFilterExtensionRemovePod).
// The usual victim -- a placeholder pod with labels of its own -- does not: the state is the terminal cycle's, rebuilt here from
// the snapshot and the clone counts (CoupledState: podtopologyspread/filtering.go:235-356, interpodaffinity/filtering.go:204-432),
// and the dry run is exact.  A potential node whose victims DO take part (PodSide::victim_interacts) is not modelled ->
// Unmodelled, the caller says so and keeps the no-victims form.
#pragma once
#include <algorithm>
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

// The PreFilter state of PodTopologySpread (hard constraints) and InterPodAffinity at the terminal cycle: per-domain tables over ALL
// nodes from the snapshot's per-node counts and the clones per node (a clone is an existing pod of the next cycle).
struct CoupledState {
    const Snapshot &s;
    const PodSide &p;
    std::vector<const Spread *> hard;
    std::vector<std::vector<int64_t>> match; // TpValueToMatchNum per hard constraint
    std::vector<int64_t> min_eff;            // the global minimum (0 below minDomains)
    bool ipa_active = false;
    std::vector<std::vector<int64_t>> aff, anti, exist;
    int64_t aff_total = 0, exist_total = 0;

    CoupledState(const Snapshot &snap, const PodSide &pod, const std::vector<int32_t> &per_node_count, uint32_t fm) : s(snap), p(pod) {
        const size_t N = s.n();
        auto clones = [&](size_t i) { return (int64_t)(i < per_node_count.size() ? per_node_count[i] : 0); };
        if (fm & CCSIM_F_TOPOLOGYSPREAD)
            for (const auto &c : p.spread)
                if (c.hard) hard.push_back(&c);
        std::vector<uint8_t> hard_keys(N, 1);
        for (const Spread *c : hard)
            for (size_t i = 0; i < N; i++) hard_keys[i] = hard_keys[i] && s.label_cols[(size_t)c->col][i] != 0;
        for (const Spread *c : hard) { // pts_prefilter
            const auto &dom = s.label_cols[(size_t)c->col];
            int32_t top = 0;
            for (const int32_t v : dom) top = std::max(top, v);
            std::vector<int64_t> tab((size_t)top + 1, 0);
            std::vector<uint8_t> present((size_t)top + 1, 0);
            for (size_t i = 0; i < N; i++) {
                if (!hard_keys[i] || (c->use_included && !c->node_included[i])) continue;
                present[(size_t)dom[i]] = 1;
                tab[(size_t)dom[i]] += (c->node_match_count.empty() ? 0 : c->node_match_count[i]) + (c->self_match ? clones(i) : 0);
            }
            int64_t mn = 2147483647LL, n_dom = 0; // math.MaxInt32, filtering.go:105
            for (size_t v = 0; v < tab.size(); v++)
                if (present[v]) n_dom++, mn = std::min(mn, tab[v]);
            match.push_back(tab), min_eff.push_back(n_dom < c->min_domains ? 0 : mn);
        }
        if ((fm & CCSIM_F_INTERPODAFFINITY) && p.has_ipa) { // ipa_build: the count maps per topology key
            const Ipa &a = p.ipa;
            const size_t K = a.key_cols.size();
            aff.resize(K), anti.resize(K), exist.resize(K);
            for (size_t k = 0; k < K; k++) {
                int32_t top = 0;
                for (const int32_t v : s.label_cols[(size_t)a.key_cols[k]]) top = std::max(top, v);
                aff[k].assign((size_t)top + 1, 0), anti[k].assign((size_t)top + 1, 0), exist[k].assign((size_t)top + 1, 0);
            }
            for (size_t i = 0; i < N; i++) {
                const int64_t am = (a.aff_existing.empty() ? 0 : a.aff_existing[i]) + (a.self_aff ? clones(i) : 0);
                if (am)
                    for (const int k : a.aff_keys) {
                        const int32_t v = s.label_cols[(size_t)a.key_cols[(size_t)k]][i];
                        if (v) aff[(size_t)k][(size_t)v] += am, aff_total += am;
                    }
                for (size_t t = 0; t < a.anti_keys.size(); t++) {
                    const int64_t m = (a.anti_existing[t].empty() ? 0 : a.anti_existing[t][i]) + (a.anti_self[t] ? clones(i) : 0);
                    const int k = a.anti_keys[t];
                    const int32_t v = s.label_cols[(size_t)a.key_cols[(size_t)k]][i];
                    if (m && v) anti[(size_t)k][(size_t)v] += m;
                }
                for (size_t k = 0; k < K; k++) {
                    int64_t m = a.exist_anti[k].empty() ? 0 : a.exist_anti[k][i];
                    for (size_t t = 0; t < a.anti_keys.size(); t++)
                        if ((size_t)a.anti_keys[t] == k && a.anti_self[t]) m += clones(i);
                    const int32_t v = s.label_cols[(size_t)a.key_cols[k]][i];
                    if (m && v) exist[k][(size_t)v] += m, exist_total += m;
                }
            }
            ipa_active = !(exist_total == 0 && a.aff_keys.empty() && a.anti_keys.empty());
        }
    }
    bool active() const { return !hard.empty() || ipa_active; }
    // the coupled filters on node i -> 0 (passes) or the reason slot; `unresolvable` says which status code
    int verdict(size_t i, bool &unresolvable) const {
        unresolvable = false;
        for (size_t h = 0; h < hard.size(); h++) { // pts_filter
            const Spread &c = *hard[h];
            const int32_t v = s.label_cols[(size_t)c.col][i];
            if (v == 0) return unresolvable = true, CCSIM_R_PTS_MISSING_LABEL;
            if (match[h][(size_t)v] + (c.self_match ? 1 : 0) - min_eff[h] > c.max_skew) return CCSIM_R_PTS_SKEW;
        }
        if (ipa_active) { // ipa_filter
            const Ipa &a = p.ipa;
            bool pods_exist = true;
            for (const int k : a.aff_keys) {
                const int32_t v = s.label_cols[(size_t)a.key_cols[(size_t)k]][i];
                if (v == 0) return unresolvable = true, CCSIM_R_IPA_AFFINITY;
                if (aff[(size_t)k][(size_t)v] <= 0) pods_exist = false;
            }
            if (!pods_exist && !(aff_total == 0 && a.self_aff)) return unresolvable = true, CCSIM_R_IPA_AFFINITY;
            for (const int k : a.anti_keys) {
                const int32_t v = s.label_cols[(size_t)a.key_cols[(size_t)k]][i];
                if (v && anti[(size_t)k][(size_t)v] > 0) return CCSIM_R_IPA_ANTI;
            }
            if (exist_total > 0)
                for (size_t k = 0; k < a.key_cols.size(); k++) {
                    const int32_t v = s.label_cols[(size_t)a.key_cols[k]][i];
                    if (v && exist[k][(size_t)v] > 0) return CCSIM_R_IPA_EXISTING_ANTI;
                }
        }
        return 0;
    }
};

inline PreemptionOutcome preemption_dry_run(const Snapshot &s, const PodSide &p, const std::vector<int32_t> &per_node_count, int64_t n_code_unschedulable,
                                            uint32_t fm, size_t n_templates = 1, bool mixed_priorities = false) {
    const size_t N = s.n(), R = s.res_names.size();
    PreemptionOutcome out;
    out.no_victims = n_code_unschedulable, out.not_helpful = (int64_t)N - n_code_unschedulable;
    if (p.preempt_never) return out.kind = PreemptionOutcome::Never, out;
    if (mixed_priorities) return out.kind = PreemptionOutcome::Unmodelled, out;
    if (p.victim_count.empty()) return out;
    if (n_templates > 1) return out.kind = PreemptionOutcome::Unmodelled, out;

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
    // the coupled filters run after NodePorts / Fit, against the terminal cycle's PreFilter state
    const CoupledState coupled(s, p, per_node_count, fm);
    int64_t potential = 0;
    bool nominated = false;
    std::vector<int64_t> used(R);
    for (size_t i = 0; i < N; i++) {
        if (!p.victim_count[i] || !preemption_static_ok(s, p, i, fm)) continue;
        const int64_t cnt = i < per_node_count.size() ? per_node_count[i] : 0;
        const bool conflict_now = ports_on && (cnt > 0 || (!p.host_ports_conflict.empty() && p.host_ports_conflict[i]));
        const bool conflict_rest = ports_on && (cnt > 0 || (!p.ports_conflict_rest.empty() && p.ports_conflict_rest[i]));
        for (size_t c = 0; c < R; c++) used[c] = s.req[c][i] + cnt * p.preq[c]; // the terminal NodeInfo (types.go:409-428)
        const int64_t pods = (int64_t)s.pod_count[i] + cnt;
        bool beyond = false, c_unres = false;
        const uint32_t m0 = fit(i, used, pods, beyond);
        const int c_reason = coupled.active() ? coupled.verdict(i, c_unres) : 0;
        // the volume plugins follow NodeResourcesFit (default_plugins.go:40-45): a node that holds a clone whose disks conflict with the next
        // one's, else the hosts' verdict against the node's pods -- now, and with the victims gone (volumes.hpp veto_with_victims_gone)
        const int vol_now = p.volume_exclusive && cnt > 0 ? 1 : (p.volume_veto.empty() ? 0 : (int)p.volume_veto[i]);
        const int vol_rest = p.volume_exclusive && cnt > 0 ? 1 : (p.volume_veto_rest.empty() ? 0 : (int)p.volume_veto_rest[i]);
        // the terminal status of the node: the first failing plugin decides the code; plain Unschedulable = a dry-run node
        const bool is_potential = conflict_now || m0 ? (conflict_now || !beyond) : vol_now ? vol_now <= CCSIM_VOL_LAST_UNSCHEDULABLE : (c_reason != 0 && !c_unres);
        if (!is_potential) continue;
        if (coupled.active() && !p.victim_interacts.empty() && p.victim_interacts[i]) {
            std::fill(out.hist.begin(), out.hist.end(), 0); // (the no-victims form: nothing of the partial walk may reach the message)
            return out.kind = PreemptionOutcome::Unmodelled, out;
        }
        potential++;
        for (size_t c = 0; c < R; c++) used[c] -= p.victim_req[c][i]; // ... with the victims gone
        const uint32_t m1 = conflict_rest ? 0 : fit(i, used, pods - p.victim_count[i], beyond);
        if (conflict_rest) out.hist[CCSIM_R_NODEPORTS]++; // the first failing plugin's reasons (framework.go:897-930)
        else if (m1) {
            if (m1 & 1u) out.hist[CCSIM_R_TOO_MANY_PODS]++;
            for (size_t c = 0; c < R; c++)
                if (m1 & (1u << (1 + c))) out.hist[CCSIM_R_RES0 + c]++;
        } else if (vol_rest) out.hist[(size_t)(CCSIM_R_VOL0 + vol_rest - 1)]++;
        else if (c_reason) out.hist[(size_t)c_reason]++;
        else nominated = true;
    }
    if (nominated) return out.kind = PreemptionOutcome::Nominated, out;
    out.no_victims = n_code_unschedulable - potential;
    return out;
}

} // namespace cchost
