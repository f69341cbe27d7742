// profile.hpp -- KubeSchedulerConfiguration (--default-config, cmd/cluster-capacity/app/options/options.go:73,
// server.go:106-113 loadConfigFromFile) -> ccsim_profile.  What the engine can express: which of its Filter plugins run,
// the Score plugins' weights, the resource lists of NodeResourcesFit (LeastAllocated) / NodeResourcesBalancedAllocation,
// percentageOfNodesToScore (global or per profile, schedule_one.go:702-708) and InterPodAffinity's hardPodAffinityWeight.
// Merge semantics follow the scheduler's (S/apis/config/v1/default_plugins.go:30-58 defaults as MultiPoint plugins with
// weights; mergePlugins :77-140; S/framework/runtime/framework.go:506-640 expandMultiPointPlugins): a plugin listed under
// multiPoint applies to every extension point it implements, `disabled: [{name: "*"}]` clears a point, a specific point
// overrides multiPoint, a Score plugin enabled without a weight gets 1.
#pragma once
#include <map>
#include <string>

#include "../../include/ccsim.h"
#include "value.hpp"

namespace cchost {

struct HostProfile {
    ccsim_profile c{};
    int hard_pod_affinity_weight = 1; // InterPodAffinityArgs (defaults.go:229-231), used by the ingest
    bool percentage_set = false;      // percentageOfNodesToScore came from the config file / the command line (else: see simulate())
    bool system_default_spreading = true; // PodTopologySpreadArgs.defaultingType System (the default); List with no constraints turns it off
    // which of VolumeRestrictions / NodeVolumeLimits / VolumeBinding / VolumeZone the host evaluates for pods with volumes (volumes.hpp):
    // disabling one under multiPoint takes it out; with only its filter point disabled a pod with volumes is refused
    std::vector<std::string> volume_plugins = {"VolumeRestrictions", "NodeVolumeLimits", "VolumeBinding", "VolumeZone"};
    bool volume_plugins_partial = false;
    bool dra_enabled = true, dra_partial = false; // DynamicResources: in the profile; only its filter point disabled (pods with resourceClaims are refused then)
};

inline HostProfile default_profile() {
    HostProfile p;
    ccsim_profile &f = p.c;
    f.filter_mask = CCSIM_F_UNSCHEDULABLE | CCSIM_F_NODENAME | CCSIM_F_TAINT | CCSIM_F_NODEAFFINITY | CCSIM_F_FIT | CCSIM_F_TOPOLOGYSPREAD | CCSIM_F_INTERPODAFFINITY |
                    CCSIM_F_NODEPORTS;
    f.w_taint = 3, f.w_nodeaffinity = 2, f.w_fit = 1, f.w_balanced = 1, f.w_topologyspread = 2, f.w_interpodaffinity = 2, f.w_imagelocality = 1;
    f.n_fit_res = 2, f.fit_res[0] = 0, f.fit_res[1] = 1, f.fit_res_w[0] = 1, f.fit_res_w[1] = 1;
    f.n_bal_res = 2, f.bal_res[0] = 0, f.bal_res[1] = 1;
    f.percentage_of_nodes_to_score = 100; // this host's default: the order-independent full search
    return p;
}

struct PluginInfo {
    uint32_t filter_bit; // 0 = no Filter this engine models
    int32_t ccsim_profile::*weight; // nullptr = no Score this engine models
    int default_weight;
};

inline const std::map<std::string, PluginInfo> &plugin_table() {
    static const std::map<std::string, PluginInfo> t = {
        {"NodeUnschedulable", {CCSIM_F_UNSCHEDULABLE, nullptr, 0}},
        {"NodeName", {CCSIM_F_NODENAME, nullptr, 0}},
        {"TaintToleration", {CCSIM_F_TAINT, &ccsim_profile::w_taint, 3}},
        {"NodeAffinity", {CCSIM_F_NODEAFFINITY, &ccsim_profile::w_nodeaffinity, 2}},
        {"NodeResourcesFit", {CCSIM_F_FIT, &ccsim_profile::w_fit, 1}},
        {"NodeResourcesBalancedAllocation", {0, &ccsim_profile::w_balanced, 1}},
        {"PodTopologySpread", {CCSIM_F_TOPOLOGYSPREAD, &ccsim_profile::w_topologyspread, 2}},
        {"InterPodAffinity", {CCSIM_F_INTERPODAFFINITY, &ccsim_profile::w_interpodaffinity, 2}},
        {"NodePorts", {CCSIM_F_NODEPORTS, nullptr, 0}},
        {"ImageLocality", {0, &ccsim_profile::w_imagelocality, 1}},
    };
    return t;
}
// default plugins whose Filter / Score is a no-op for the pods this simulator accepts: accepted, ignored
// (pods that would activate a volume / DRA plugin are refused at ingest)
inline bool folded_away(const std::string &n) {
    static const char *names[] = {"SchedulingGates", "PrioritySort", "VolumeRestrictions", "NodeVolumeLimits", "EBSLimits",
                                  "GCEPDLimits", "AzureDiskLimits", "VolumeBinding", "VolumeZone", "DynamicResources", "DefaultPreemption",
                                  "DefaultBinder", "ClusterCapacityBinder"};
    for (const char *x : names)
        if (n == x) return true;
    return false;
}

inline int resource_column(const std::string &name) {
    if (name == "cpu") return 0;
    if (name == "memory") return 1;
    if (name == "ephemeral-storage") return 2;
    // (the engine scores scalar resources too -- columns 3+ of the ABI -- but a config names them before the snapshot's
    // scalar columns exist: not wired through this host)
    throw std::runtime_error("scheduler config: scoring resource '" + name + "' is not supported by this host (cpu, memory and ephemeral-storage are)");
}

inline HostProfile profile_from_config(const Value &cfg) {
    HostProfile p = default_profile();
    if (cfg.is_null()) return p;
    if (cfg["kind"].truthy() && cfg["kind"].text() != "KubeSchedulerConfiguration") throw std::runtime_error("scheduler config: kind is not KubeSchedulerConfiguration");
    if (cfg.has("percentageOfNodesToScore") && !cfg["percentageOfNodesToScore"].is_null())
        p.c.percentage_of_nodes_to_score = cfg["percentageOfNodesToScore"].as_int32(), p.percentage_set = true;
    const Value &profiles = cfg["profiles"];
    if (profiles.items().size() > 1) throw std::runtime_error("scheduler config: one profile only (the simulated pod is scheduled by Profiles[0], pkg/utils/utils.go:103-108)");
    const Value &prof = profiles.items().empty() ? Value::null_value() : profiles.items()[0];
    if (prof.has("percentageOfNodesToScore") && !prof["percentageOfNodesToScore"].is_null())
        p.c.percentage_of_nodes_to_score = prof["percentageOfNodesToScore"].as_int32(), p.percentage_set = true;

    auto lookup = [&](const std::string &name) -> const PluginInfo * {
        auto it = plugin_table().find(name);
        if (it != plugin_table().end()) return &it->second;
        if (folded_away(name)) return nullptr;
        throw std::runtime_error("scheduler config: unknown plugin '" + name + "'");
    };
    auto set_filter = [&](const PluginInfo &i, bool on) {
        if (i.filter_bit) p.c.filter_mask = on ? (p.c.filter_mask | i.filter_bit) : (p.c.filter_mask & ~i.filter_bit);
    };
    auto set_score = [&](const PluginInfo &i, int w) {
        if (i.weight) p.c.*(i.weight) = w;
    };
    auto apply = [&](const Value &set, bool do_filter, bool do_score, bool multipoint) {
        for (const auto &d : set["disabled"].items()) {
            const std::string name = d["name"].text();
            const bool volume_plugin = name == "VolumeRestrictions" || name == "NodeVolumeLimits" || name == "VolumeBinding" || name == "VolumeZone";
            if (do_filter && (name == "*" || name == "DynamicResources")) {
                if (multipoint) p.dra_enabled = false;
                else p.dra_partial = true;
            }
            if (do_filter && (name == "*" || volume_plugin)) {
                if (!multipoint) p.volume_plugins_partial = true; // (its PreFilter would still run: refused when a pod with volumes arrives)
                if (name == "*") p.volume_plugins.clear();
                else p.volume_plugins.erase(std::remove(p.volume_plugins.begin(), p.volume_plugins.end(), name), p.volume_plugins.end());
            }
            if (name == "*") {
                for (const auto &kv : plugin_table()) {
                    if (do_filter) set_filter(kv.second, false);
                    if (do_score) set_score(kv.second, 0);
                }
                continue;
            }
            if (const PluginInfo *i = lookup(name)) {
                if (do_filter) set_filter(*i, false);
                if (do_score) set_score(*i, 0);
            }
        }
        for (const auto &e : set["enabled"].items()) {
            if (multipoint && e["name"].text() == "DynamicResources") p.dra_enabled = true;
            if (multipoint) {
                const std::string en = e["name"].text();
                static const char *order[] = {"VolumeRestrictions", "NodeVolumeLimits", "VolumeBinding", "VolumeZone"};
                for (const char *v : order)
                    if (en == v && std::find(p.volume_plugins.begin(), p.volume_plugins.end(), en) == p.volume_plugins.end()) p.volume_plugins.push_back(en);
            }
            const PluginInfo *i = lookup(e["name"].text());
            if (!i) continue;
            if (do_filter) set_filter(*i, true);
            if (do_score && i->weight) {
                const int w = e["weight"].as_int32();
                if (w < 0 || w > 100 * 1000) throw std::runtime_error("scheduler config: bad weight for " + e["name"].text());
                // multiPoint: an unset weight keeps the plugin's default; a Score entry without weight gets 1 (defaults.go:120-131)
                set_score(*i, w > 0 ? w : (multipoint ? i->default_weight : 1));
            }
        }
    };
    const Value &plugins = prof["plugins"];
    apply(plugins["multiPoint"], true, true, true);
    apply(plugins["filter"], true, false, false);
    apply(plugins["score"], false, true, false);

    for (const auto &pc : prof["pluginConfig"].items()) {
        const std::string name = pc["name"].text();
        const Value &args = pc["args"];
        if (name == "NodeResourcesFit") {
            const Value &st = args["scoringStrategy"];
            if (st.truthy()) {
                const std::string type = st["type"].truthy() ? st["type"].text() : "LeastAllocated";
                if (type != "LeastAllocated") throw std::runtime_error("scheduler config: NodeResourcesFit scoringStrategy " + type + " is not implemented (LeastAllocated is)");
                if (st["resources"].truthy()) {
                    p.c.n_fit_res = 0;
                    for (const auto &r : st["resources"].items()) {
                        if (p.c.n_fit_res >= CCSIM_MAX_RES) throw std::runtime_error("scheduler config: too many scoring resources");
                        p.c.fit_res[p.c.n_fit_res] = resource_column(r["name"].text());
                        p.c.fit_res_w[p.c.n_fit_res] = r["weight"].truthy() ? r["weight"].as_int() : 1;
                        p.c.n_fit_res++;
                    }
                }
            }
            if (args["ignoredResources"].truthy() || args["ignoredResourceGroups"].truthy()) throw std::runtime_error("scheduler config: NodeResourcesFit ignoredResources are not implemented");
        } else if (name == "NodeResourcesBalancedAllocation") {
            if (args["resources"].truthy()) {
                p.c.n_bal_res = 0;
                for (const auto &r : args["resources"].items()) {
                    if (p.c.n_bal_res >= CCSIM_MAX_RES) throw std::runtime_error("scheduler config: too many balanced resources");
                    p.c.bal_res[p.c.n_bal_res++] = resource_column(r["name"].text());
                }
            }
        } else if (name == "InterPodAffinity") {
            if (args.has("hardPodAffinityWeight")) p.hard_pod_affinity_weight = args["hardPodAffinityWeight"].as_int32();
            if (args["ignorePreferredTermsOfExistingPods"].truthy()) throw std::runtime_error("scheduler config: ignorePreferredTermsOfExistingPods is not implemented");
        } else if (name == "PodTopologySpread") {
            if (args["defaultConstraints"].truthy()) throw std::runtime_error("scheduler config: PodTopologySpread defaultConstraints are not implemented (defaultingType System needs Services / ReplicaSets)");
            if (args["defaultingType"].text() == "List") p.system_default_spreading = false; // (an empty list: no default constraints at all, plugin.go:105-113)
        } else if (name == "NodeAffinity") {
            if (args["addedAffinity"].truthy()) throw std::runtime_error("scheduler config: NodeAffinity addedAffinity is not implemented");
        } else if (!folded_away(name) && !plugin_table().count(name))
            throw std::runtime_error("scheduler config: pluginConfig for unknown plugin '" + name + "'");
    }
    const int pct = p.c.percentage_of_nodes_to_score;
    if (pct < 0 || pct > 100) throw std::runtime_error("scheduler config: percentageOfNodesToScore out of [0,100] (validation.go:86-90)");
    return p;
}

inline Value profile_json(const HostProfile &p) {
    Value o = Value::object();
    const ccsim_profile &f = p.c;
    o.set("filter_mask", Value::num(f.filter_mask));
    o.set("w_taint", Value::num(f.w_taint)), o.set("w_nodeaffinity", Value::num(f.w_nodeaffinity)), o.set("w_fit", Value::num(f.w_fit));
    o.set("w_balanced", Value::num(f.w_balanced)), o.set("w_topologyspread", Value::num(f.w_topologyspread)), o.set("w_interpodaffinity", Value::num(f.w_interpodaffinity));
    o.set("w_imagelocality", Value::num(f.w_imagelocality));
    Value fr = Value::array(), fw = Value::array(), br = Value::array();
    for (int i = 0; i < f.n_fit_res; i++) fr.a.push_back(Value::num(f.fit_res[i])), fw.a.push_back(Value::num(f.fit_res_w[i]));
    for (int i = 0; i < f.n_bal_res; i++) br.a.push_back(Value::num(f.bal_res[i]));
    o.set("fit_res", fr), o.set("fit_res_w", fw), o.set("bal_res", br);
    o.set("percentage_of_nodes_to_score", Value::num(f.percentage_of_nodes_to_score)), o.set("hard_pod_affinity_weight", Value::num(p.hard_pod_affinity_weight));
    return o;
}

} // namespace cchost
