// report.hpp -- the strings the reference produces around the path, from the integer results of the C ABI.
//   FitError.Error                    vendor/k8s.io/kubernetes/pkg/scheduler/framework/types.go:787-836
//   DefaultPreemption PostFilter msg  .../plugins/defaultpreemption/default_preemption.go:131-141,257
//                                     .../framework/preemption/preemption.go:266-279
//   StopReason / getMainFailReason    pkg/framework/simulator.go:297-342, pkg/framework/report.go:100-109
//   parsePodsReview / GetReport       pkg/framework/report.go:111-225
//   clusterCapacityReviewPrettyPrint  pkg/framework/report.go:235-283
#pragma once
#include <algorithm>
#include <cstdio>
#include <ctime>
#include <map>
#include <string>
#include <vector>

#include "../../include/ccsim.h"
#include "preemption.hpp"
#include "quantity.hpp"
#include "snapshot.hpp"
#include "value.hpp"

namespace cchost {

struct RunResult { // what ccsim_run hands back (host copies)
    int64_t placed = 0;
    int32_t stop = CCSIM_STOP_UNSCHEDULABLE;
    std::vector<int32_t> per_node_count, log;
    std::vector<int64_t> hist, hist_taintset;
    int64_t n_code_unschedulable = 0;
    std::vector<int32_t> per_spec_count; // several templates: placements per template ...
    int32_t stop_spec = -1;              // ... and the template whose pod was Unschedulable
    std::string prefilter_msg;           // host only: the terminal cycle was rejected by a PreFilter plugin (FitError.Diagnosis.PreFilterMsg)
};

inline std::string histogram_message(const std::map<std::string, int64_t> &reasons) {
    // types.go:820-829: "<count> <reason>" strings sorted lexicographically, joined by ", "
    std::vector<std::string> items;
    for (const auto &kv : reasons)
        if (kv.second) items.push_back(std::to_string(kv.second) + " " + kv.first);
    std::sort(items.begin(), items.end());
    std::string out;
    for (size_t i = 0; i < items.size(); i++) out += (i ? ", " : "") + items[i];
    return out;
}

inline const char *reason_text(int slot) {
    switch (slot) {
    case CCSIM_R_UNSCHEDULABLE: return "node(s) were unschedulable";
    case CCSIM_R_NODENAME: return "node(s) didn't match the requested node name";
    case CCSIM_R_NODEAFFINITY: return "node(s) didn't match Pod's node affinity/selector";
    case CCSIM_R_TOO_MANY_PODS: return "Too many pods";
    case CCSIM_R_PTS_MISSING_LABEL: return "node(s) didn't match pod topology spread constraints (missing required label)";
    case CCSIM_R_PTS_SKEW: return "node(s) didn't match pod topology spread constraints";
    case CCSIM_R_IPA_AFFINITY: return "node(s) didn't match pod affinity rules"; // interpodaffinity/filtering.go:37-45
    case CCSIM_R_IPA_ANTI: return "node(s) didn't match pod anti-affinity rules";
    case CCSIM_R_IPA_EXISTING_ANTI: return "node(s) didn't satisfy existing pods anti-affinity rules";
    case CCSIM_R_NODEPORTS: return "node(s) didn't have free ports for the requested pod ports"; // nodeports/node_ports.go:39
    // volumerestrictions/volume_restrictions.go:57-59, nodevolumelimits/csi.go:44, volumebinding/binder.go:65-71, volumezone/volume_zone.go:61
    case CCSIM_R_VOL_DISK_CONFLICT: return "node(s) had no available disk";
    case CCSIM_R_VOL_RWOP: return "node(s) unavailable due to PersistentVolumeClaim with ReadWriteOncePod access mode already in-use by another pod";
    case CCSIM_R_VOL_MAX_COUNT: return "node(s) exceed max volume count";
    case CCSIM_R_VOL_NODE_AFFINITY: return "node(s) didn't match PersistentVolume's node affinity";
    case CCSIM_R_VOL_NO_PV: return "node(s) didn't find available persistent volumes to bind";
    case CCSIM_R_VOL_PV_NOT_EXIST: return "node(s) unavailable due to one or more pvc(s) bound to non-existent pv(s)";
    case CCSIM_R_VOL_ZONE: return "node(s) had no available volume zone";
    }
    return nullptr;
}

inline std::map<std::string, int64_t> reason_histogram(const std::vector<int64_t> &hist, const std::vector<int64_t> &hist_taintset,
                                                       const std::vector<std::string> &taint_reasons, const std::vector<std::string> &scalar_names) {
    static const char *res_names[3] = {"cpu", "memory", "ephemeral-storage"};
    std::map<std::string, int64_t> reasons;
    for (int slot = 0; slot < (int)hist.size(); slot++) {
        if (!hist[(size_t)slot]) continue;
        std::string text;
        if (const char *t = reason_text(slot)) text = t;
        else if (slot >= CCSIM_R_RES0 && slot < CCSIM_R_RES0 + CCSIM_MAX_RES) {
            const int c = slot - CCSIM_R_RES0;
            text = std::string("Insufficient ") + (c < 3 ? res_names[c] : (c - 3 < (int)scalar_names.size() ? scalar_names[(size_t)c - 3] : "scalar-" + std::to_string(c - 3)));
        } else
            text = "reason-" + std::to_string(slot);
        reasons[text] += hist[(size_t)slot];
    }
    for (size_t ts = 0; ts < hist_taintset.size(); ts++)
        if (hist_taintset[ts]) // taint_toleration.go:119: the first untolerated taint of the set
            reasons[ts < taint_reasons.size() ? taint_reasons[ts] : "node(s) had untolerated taint {taintset-" + std::to_string(ts) + "}"] += hist_taintset[ts];
    return reasons;
}

// FitError.Error() for the terminal cycle, including the DefaultPreemption tail.  `pre`: the outcome of the dry run
// (preemption.hpp); nullptr = no node holds a pod of lower priority than the simulated one.
inline std::string fit_error_message(int64_t n_nodes, const RunResult &r, const std::vector<std::string> &taint_reasons,
                                     const std::vector<std::string> &scalar_names, const PreemptionOutcome *pre = nullptr) {
    std::string msg = "0/" + std::to_string(n_nodes) + " nodes are available:";
    if (!r.prefilter_msg.empty()) {
        // a PreFilter plugin rejected the pod (schedule_one.go:495-508): its message stands for every node (types.go:789-794), and every
        // node is UnschedulableAndUnresolvable for the preemption that follows (n_code_unschedulable = 0)
        msg += " " + r.prefilter_msg + ".";
    } else {
        const std::string body = histogram_message(reason_histogram(r.hist, r.hist_taintset, taint_reasons, scalar_names));
        if (!body.empty()) msg += " " + body + ".";
    }
    if (pre && pre->kind == PreemptionOutcome::Nominated) return msg; // a candidate: PostFilter Success, empty message (preemption.go:281-303)
    if (pre && pre->kind == PreemptionOutcome::Never) return msg + " preemption: not eligible due to preemptionPolicy=Never."; // default_preemption.go:355-357
    // Nodes that failed with plain Unschedulable are dry-run nodes: without a lower-priority pod each reports "No preemption
    // victims found"; with victims that do not help, the filter status after their removal; the rest are absent from the map.
    std::map<std::string, int64_t> h;
    if (pre) h = reason_histogram(pre->hist, {}, {}, scalar_names);
    const int64_t no_victims = pre ? pre->no_victims : r.n_code_unschedulable;
    if (no_victims) h["No preemption victims found for incoming pod"] = no_victims;
    if (n_nodes - r.n_code_unschedulable > 0) h["Preemption is not helpful for scheduling"] = n_nodes - r.n_code_unschedulable;
    std::string pmsg = "0/" + std::to_string(n_nodes) + " nodes are available:";
    const std::string pbody = histogram_message(h);
    if (!pbody.empty()) pmsg += " " + pbody + ".";
    return msg + " preemption: " + pmsg;
}

// ClusterCapacity.Status.StopReason (simulator.go:301,331)
inline std::string stop_reason(const RunResult &r, int64_t n_nodes, int64_t max_limit, const std::vector<std::string> &taint_reasons,
                               const std::vector<std::string> &scalar_names, const PreemptionOutcome *pre = nullptr) {
    if (r.stop == CCSIM_STOP_LIMIT) return "LimitReached: Maximum number of pods simulated: " + std::to_string(max_limit);
    if (r.stop == CCSIM_STOP_NO_NODES) return "Unschedulable: no nodes available to schedule pods";
    return "Unschedulable: " + fit_error_message(n_nodes, r, taint_reasons, scalar_names, pre);
}

// The PostFilter of the terminal cycle for a finished run: StopReason with DefaultPreemption's dry run folded in.
inline std::string terminal_stop_reason(const Snapshot &snap, const RunResult &r, int64_t max_limit, uint32_t filter_mask, bool warn = false) {
    const size_t P = snap.n_templates();
    const size_t failing = P > 1 && r.stop_spec >= 0 ? (size_t)r.stop_spec : 0; // the FitError describes the template that did not fit
    if (r.stop != CCSIM_STOP_UNSCHEDULABLE) return stop_reason(r, (int64_t)snap.n(), max_limit, snap.side(failing).taint_reasons, snap.scalar_names);
    bool mixed = false; // clones of one template below another template's priority
    for (size_t t = 1; t < P; t++) mixed = mixed || snap.side(t).priority != snap.side(0).priority;
    PodSide terminal; // (only when the terminal cycle saw another pod than the snapshot's)
    const PodSide *pf = &snap.side(failing);
    if (pf->rwop_capacity_one && r.placed >= 1 && r.prefilter_msg.empty()) {
        // the terminal cycle saw the pod with its ReadWriteOncePod claim held by its own clone (rwop_now_in_use): a clone is no victim, so the
        // claim stays in conflict on every node whatever the dry run removes (static disk conflicts may leave with a victim)
        terminal = *pf;
        rwop_now_in_use(terminal, snap.n(), P == 1 ? &r.per_node_count : nullptr);
        std::vector<uint8_t> rest(snap.n(), 2);
        for (size_t i = 0; i < rest.size() && i < pf->volume_veto_rest.size(); i++)
            if (pf->volume_veto_rest[i] == 1) rest[i] = 1;
        terminal.volume_veto_rest = std::move(rest);
        pf = &terminal;
    }
    const PreemptionOutcome pre = preemption_dry_run(snap, *pf, r.per_node_count, r.n_code_unschedulable, filter_mask, P, mixed);
    if (warn && pre.kind == PreemptionOutcome::Unmodelled)
        std::fprintf(stderr, "warning: a lower-priority pod takes part in a topology-coupled filter of the simulated pod (or several templates "
                             "run): the preemption dry run is not modelled, the 'preemption:' part of the message assumes no victims\n");
    return stop_reason(r, (int64_t)snap.n(), max_limit, snap.side(failing).taint_reasons, snap.scalar_names, &pre);
}

// report.go:100-109 getMainFailReason
inline Value main_fail_reason(const std::string &message) {
    const std::string first = message.substr(0, message.find('\n'));
    const size_t colon = first.find(':');
    std::string rest = colon == std::string::npos ? "" : first.substr(colon + 1);
    while (!rest.empty() && rest.front() == ' ') rest.erase(rest.begin());
    while (!rest.empty() && rest.back() == ' ') rest.pop_back();
    Value o = Value::object();
    o.set("failType", Value::str(first.substr(0, colon))), o.set("failMessage", Value::str(rest));
    return o;
}

// report.go:146-180: per-node replica counts, in first-placement order when the log is available.  The log may be
// capped below the number of placements (ccsim_report.log_cap): nodes first placed beyond the cap follow in canonical
// node order, so the list always covers every node with a replica and sums to the headline count.
inline Value replicas_on_nodes(const RunResult &r, const std::vector<std::string> &names) {
    std::vector<size_t> order;
    if (!r.log.empty()) {
        std::vector<char> seen(r.per_node_count.size(), 0);
        for (const int32_t g : r.log)
            if (g >= 0 && (size_t)g < seen.size() && !seen[(size_t)g]) seen[(size_t)g] = 1, order.push_back((size_t)g);
        for (size_t i = 0; i < r.per_node_count.size(); i++)
            if (r.per_node_count[i] && !seen[i]) order.push_back(i);
    } else
        for (size_t i = 0; i < r.per_node_count.size(); i++)
            if (r.per_node_count[i]) order.push_back(i);
    Value a = Value::array();
    for (const size_t i : order) {
        Value o = Value::object();
        o.set("nodeName", Value::str(i < names.size() ? names[i] : std::to_string(i))), o.set("replicas", Value::num(r.per_node_count[i]));
        a.a.push_back(o);
    }
    return a;
}

// report.go:111-144 getResourceRequest (containers only) + :182-194
// cpu and memory are sums of Quantities printed by Quantity.String(): the container's quantity is the receiver of Add
// (`rQuantity.Add(sum so far)`), so the sum carries the Format of the LAST container that names the resource with a non-zero amount
// (quantity.go:600-613), starting from DecimalSI (cpu) / BinarySI (memory): `memory: 512M` prints as 512M, 1Gi + 512Mi as 1536Mi.
inline Value pod_requirements(const Value &pod) {
    __int128 cpu = 0, mem = 0; // nano units
    QuantityFormat cpu_fmt = QuantityFormat::DecimalSI, mem_fmt = QuantityFormat::BinarySI;
    Value scalars = Value::object();
    for (const auto &c : pod["spec"]["containers"].items())
        for (const auto &kv : c["resources"]["requests"].fields()) {
            if (kv.first == "cpu" || kv.first == "memory") {
                const __int128 v = quantity_nano(parse_quantity(kv.second.text()));
                __int128 &acc = kv.first == "cpu" ? cpu : mem;
                QuantityFormat &fmt = kv.first == "cpu" ? cpu_fmt : mem_fmt;
                if (v != 0) fmt = quantity_format(kv.second.text());
                acc += v;
            } else if (is_scalar_resource(kv.first))
                scalars.set(kv.first, Value::num(scalars[kv.first].as_int() + quantity_value(kv.second.text())));
        }
    Value prim = Value::object();
    prim.set("cpu", Value::str(quantity_canonical(cpu, cpu_fmt))), prim.set("memory", Value::str(quantity_canonical(mem, mem_fmt))), prim.set("nvdia.com/gpu", Value::str("0")); // sic: ResourceNvidiaGPU = "nvdia.com/gpu" (report.go:34)
    Value res = Value::object();
    res.set("primaryResources", prim), res.set("scalarResources", scalars.o.empty() ? Value() : scalars);
    Value o = Value::object();
    o.set("podName", Value::str(pod["metadata"]["name"].text())), o.set("resources", res), o.set("nodeSelectors", pod["spec"]["nodeSelector"]);
    return o;
}

inline std::string utc_now_iso() {
    const std::time_t t = std::time(nullptr);
    std::tm tm{};
    gmtime_r(&t, &tm);
    char buf[40];
    std::strftime(buf, sizeof buf, "%Y-%m-%dT%H:%M:%SZ", &tm); // time.Time marshals as RFC 3339 with "Z" for UTC
    return buf;
}

// report.go:196-225 GetReport.  Scheduled pod i is a clone of template i mod P (parsePodsReview, report.go:146-171): exactly
// the order the engine cycles the templates in.
inline Value build_review(const std::vector<Value> &templates_in, const Snapshot &snap, const RunResult &r, int64_t max_limit,
                          uint32_t filter_mask = ~0u) {
    const size_t P = templates_in.size();
    const std::string stop = terminal_stop_reason(snap, r, max_limit, filter_mask);
    Value spec = Value::object();
    Value templates = Value::array(), reqs = Value::array();
    for (const auto &pod : templates_in) templates.a.push_back(pod), reqs.a.push_back(pod_requirements(pod));
    spec.set("templates", templates), spec.set("replicas", Value::num(0)), spec.set("podRequirements", reqs);
    if (P > 1 && (int64_t)r.log.size() < r.placed) throw std::runtime_error("several templates: the placement log does not cover the run");
    Value pods = Value::array();
    for (size_t t = 0; t < P; t++) {
        Value p = Value::object();
        p.set("podName", Value::str(templates_in[t]["metadata"]["name"].text()));
        if (P == 1) p.set("replicasOnNodes", replicas_on_nodes(r, snap.names));
        else { // this template's clones: log entries t, t + P, t + 2P, ...
            RunResult rt;
            rt.per_node_count.assign(snap.n(), 0);
            for (size_t i = t; i < r.log.size(); i += P) rt.log.push_back(r.log[i]), rt.per_node_count[(size_t)r.log[i]] += 1;
            p.set("replicasOnNodes", replicas_on_nodes(rt, snap.names));
        }
        p.set("failSummary", Value());
        pods.a.push_back(p);
    }
    Value status = Value::object();
    status.set("creationTimestamp", Value::str(utc_now_iso())), status.set("replicas", Value::num(r.placed));
    status.set("failReason", main_fail_reason(stop)), status.set("pods", pods);
    Value review = Value::object();
    review.set("spec", spec), review.set("status", status);
    return review;
}
inline Value build_review(const Value &pod, const Snapshot &snap, const RunResult &r, int64_t max_limit, uint32_t filter_mask = ~0u) {
    return build_review(std::vector<Value>{pod}, snap, r, max_limit, filter_mask);
}

// report.go:235-283 clusterCapacityReviewPrettyPrint
inline std::string pretty(const Value &review, bool verbose) {
    std::string out;
    auto line = [&](const std::string &s) { out += s + "\n"; };
    if (verbose)
        for (const auto &req : review["spec"]["podRequirements"].items()) {
            line(req["podName"].text() + " pod requirements:");
            line("\t- CPU: " + req["resources"]["primaryResources"]["cpu"].text());
            line("\t- Memory: " + req["resources"]["primaryResources"]["memory"].text());
            if (!req["resources"]["scalarResources"].is_null()) { // fmt's %v of a map[v1.ResourceName]int64: map[key:value key:value], keys sorted
                std::vector<std::pair<std::string, std::string>> kv;
                for (const auto &f : req["resources"]["scalarResources"].fields()) kv.emplace_back(f.first, f.second.text());
                std::sort(kv.begin(), kv.end());
                std::string d;
                for (size_t i = 0; i < kv.size(); i++) d += (i ? " " : "") + kv[i].first + ":" + kv[i].second;
                line("\t- ScalarResources: map[" + d + "]");
            }
            if (!req["nodeSelectors"].is_null()) {
                std::vector<std::pair<std::string, std::string>> kv;
                for (const auto &f : req["nodeSelectors"].fields()) kv.emplace_back(f.first, f.second.text());
                std::sort(kv.begin(), kv.end());
                std::string s;
                for (size_t i = 0; i < kv.size(); i++) s += (i ? "," : "") + kv[i].first + "=" + kv[i].second;
                line("\t- NodeSelector: " + s);
            }
            line("");
        }
    for (const auto &p : review["status"]["pods"].items()) {
        long long total = 0;
        for (const auto &r : p["replicasOnNodes"].items()) total += r["replicas"].as_int();
        line(verbose ? "The cluster can schedule " + std::to_string(total) + " instance(s) of the pod " + p["podName"].text() + "." : std::to_string(total));
    }
    if (verbose) {
        const Value &fr = review["status"]["failReason"];
        line("\nTermination reason: " + fr["failType"].text() + ": " + fr["failMessage"].text());
        if (review["status"]["replicas"].as_int() > 0) {
            line("\nPod distribution among nodes:");
            for (const auto &p : review["status"]["pods"].items()) {
                line(p["podName"].text());
                for (const auto &r : p["replicasOnNodes"].items()) line("\t- " + r["nodeName"].text() + ": " + r["replicas"].text() + " instance(s)");
            }
        }
    }
    return out;
}

} // namespace cchost
