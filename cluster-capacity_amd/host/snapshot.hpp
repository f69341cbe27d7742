// snapshot.hpp -- SyncWithClient (pkg/framework/simulator.go:176-295) and every per-pod-spec precomputation the scheduler
// plugins do with strings, turned into the integer world of include/ccsim.h.  Nothing here is on the hot path.
//
// Reference (S/ = vendor/k8s.io/kubernetes/pkg/scheduler):
//   which objects are copied          pkg/framework/simulator.go:176-295 (non-terminal pods, nodes minus --exclude-nodes)
//   pod requests                      vendor/k8s.io/component-helpers/resource/helpers.go:144-251 (sum containers, max init, + overhead)
//   NonZero requests                  S/framework/types.go:1095-1124 (100m / 200Mi per container without the request)
//   NodeInfo.AddPod                   S/framework/types.go:345-350,409-428
//   node order                        S/backend/cache/node_tree.go:119-143, component-helpers/node/topology/helpers.go:31-58
//   taints / tolerations              component-helpers/scheduling/corev1/helpers.go:63-101, api/core/v1/toleration.go:38-57
//   node selector requirements        apimachinery/pkg/labels/selector.go:246-293, component-helpers/.../nodeaffinity.go
//   label selectors                   apimachinery/pkg/apis/meta/v1/helpers.go:36-75
//   spread constraints                S/framework/plugins/podtopologyspread/common.go:42-159
//   inter-pod affinity terms          vendor/k8s.io/kube-scheduler/framework/types.go:379-384, S/framework/types.go:927-935
#pragma once
#include <algorithm>
#include <map>
#include <thread>
#include <unordered_map>
#include <set>
#include <string>
#include <vector>

#include "../../include/ccsim.h"
#include "quantity.hpp"
#include "value.hpp"

namespace cchost {

constexpr int64_t kDefaultMilliCPU = 100; // S/util/pod_resources.go:28-31
constexpr int64_t kDefaultMemory = 200ll * 1024 * 1024;
static const char *const kHostname = "kubernetes.io/hostname";
static const char *const kUnschedTaint = "node.kubernetes.io/unschedulable";

struct Unsupported : std::runtime_error {
    using std::runtime_error::runtime_error;
};

inline int64_t res_of(const Value &rl, const std::string &name) { // a ResourceList entry, 0 when absent
    if (!rl.truthy() || !rl.has(name)) return 0;
    const std::string q = rl[name].text();
    return name == "cpu" ? quantity_milli_value(q) : quantity_value(q);
}

// S/util/utils.go:140-143: extended / hugepages / prefixed native / attachable-volumes resources
// validation.IsQualifiedName (apimachinery/pkg/util/validation/validation.go:29-70): [dns-1123-subdomain "/"] name; the name is at
// most 63 characters of [-A-Za-z0-9_.] starting and ending alphanumeric, the prefix at most 253 of lower-case labels joined by dots
inline bool is_qualified_name(const std::string &value) {
    auto alnum = [](char c) { return (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z') || (c >= '0' && c <= '9'); };
    auto lower_alnum = [](char c) { return (c >= 'a' && c <= 'z') || (c >= '0' && c <= '9'); };
    const size_t slash = value.find('/');
    std::string name = value;
    if (slash != std::string::npos) {
        if (value.find('/', slash + 1) != std::string::npos) return false;
        const std::string prefix = value.substr(0, slash);
        name = value.substr(slash + 1);
        if (prefix.empty() || prefix.size() > 253) return false;
        size_t at = 0; // labels [a-z0-9]([-a-z0-9]*[a-z0-9])? joined by '.'
        while (true) {
            const size_t dot = prefix.find('.', at);
            const std::string label = prefix.substr(at, dot == std::string::npos ? std::string::npos : dot - at);
            if (label.empty() || !lower_alnum(label.front()) || !lower_alnum(label.back())) return false;
            for (const char c : label)
                if (!(lower_alnum(c) || c == '-')) return false;
            if (dot == std::string::npos) break;
            at = dot + 1;
        }
    }
    if (name.empty() || name.size() > 63 || !alnum(name.front()) || !alnum(name.back())) return false;
    for (const char c : name)
        if (!(alnum(c) || c == '-' || c == '_' || c == '.')) return false;
    return true;
}
// schedutil.IsScalarResourceName (S/util/utils.go:140-143) = extended || hugepages-* || *kubernetes.io/* || attachable-volumes-*
// (pkg/apis/core/v1/helper/helpers.go:36-66,133-135).  Anything else that is not cpu / memory / ephemeral-storage is DROPPED by the
// scheduler's Resource.Add, e.g. an unqualified "foo" or a "requests."-prefixed name.
inline bool is_scalar_resource(const std::string &name) {
    const bool prefixed_native = name.find("kubernetes.io/") != std::string::npos;
    const bool native = name.find('/') == std::string::npos || prefixed_native;
    const bool extended = !native && name.rfind("requests.", 0) != 0 && is_qualified_name("requests." + name);
    return extended || name.rfind("hugepages-", 0) == 0 || prefixed_native || name.rfind("attachable-volumes-", 0) == 0;
}

struct PodRequests {
    std::vector<int64_t> req; // per resource name
    int64_t nz_cpu = 0, nz_mem = 0;
};

// IsSupportedPodLevelResource (component-helpers/resource/helpers.go): cpu, memory, hugepages-*
inline bool pod_level_supported(const std::string &n) { return n == "cpu" || n == "memory" || n.rfind("hugepages-", 0) == 0; }

// PodRequests for one resource (helpers.go:144-251): sum of the containers; restartable (sidecar) init containers add to the
// sum; InitContainerUse(i) = the i-th init container + the sidecars before it, and the pod needs at least the largest of those;
// pod-level requests (spec.resources) override the aggregate for the resources they may carry; + overhead.  `dflt` >= 0 stands in
// for a container that does not name the resource (NonMissingContainerRequests, types.go:1095-1124).
inline int64_t aggregate_request(const Value &spec, const std::string &name, int64_t dflt = -1) {
    auto creq = [&](const Value &c) -> int64_t {
        const Value &r = c["resources"]["requests"];
        if (r.truthy() && r.has(name)) return res_of(r, name);
        return dflt >= 0 ? dflt : 0;
    };
    int64_t total = 0, restartable = 0, init_max = 0;
    for (const auto &c : spec["containers"].items()) total = add64(total, creq(c));
    for (const auto &ic : spec["initContainers"].items()) {
        const int64_t r = creq(ic);
        int64_t use;
        if (ic["restartPolicy"].text() == "Always") total = add64(total, r), restartable = add64(restartable, r), use = restartable;
        else use = add64(r, restartable);
        init_max = std::max(init_max, use);
    }
    total = std::max(total, init_max);
    const Value &pod_level = spec["resources"]["requests"];
    if (pod_level.truthy() && pod_level.has(name) && pod_level_supported(name)) total = res_of(pod_level, name);
    return add64(total, res_of(spec["overhead"], name));
}

// does the pod's aggregated ResourceList hold `name` at all (some container, init container or the pod level names it)?
inline bool named_anywhere(const Value &spec, const std::string &name) {
    for (const char *list : {"containers", "initContainers"})
        for (const auto &c : spec[list].items())
            if (c["resources"]["requests"].truthy() && c["resources"]["requests"].has(name)) return true;
    const Value &pod_level = spec["resources"]["requests"];
    return pod_level.truthy() && pod_level.has(name) && pod_level_supported(name);
}

// types.go:700-734, 1095-1124: without pod-level requests every container lacking cpu / memory counts 100m / 200Mi; WITH
// pod-level requests (spec.resources.requests non-empty) a default is used only for a resource nobody names
inline PodRequests pod_requests(const Value &spec, const std::vector<std::string> &names) {
    PodRequests out;
    // The usual pod of a cluster dump -- containers only: no init containers, no pod-level requests, no overhead -- in ONE pass
    // over its containers (the general form below walks them once per resource name and again for the non-zero defaults; with
    // hundreds of thousands of existing pods that was most of the ingest).  Same sums by construction: every aggregate_request
    // degenerates to "sum over the containers of (the named request, else the default)".
    if (!spec["initContainers"].truthy() && !spec["resources"]["requests"].truthy() && !spec["overhead"].truthy()) {
        out.req.assign(names.size(), 0);
        for (const auto &c : spec["containers"].items()) {
            const Value &r = c["resources"]["requests"];
            const bool any = r.truthy();
            bool has_cpu = false, has_mem = false;
            for (size_t k = 0; any && k < names.size(); k++)
                if (const Value *q = r.find(names[k])) {
                    const int64_t v = names[k] == "cpu" ? quantity_milli_value(q->text()) : quantity_value(q->text());
                    out.req[k] = add64(out.req[k], v);
                    if (names[k] == "cpu") has_cpu = true, out.nz_cpu = add64(out.nz_cpu, v);
                    else if (names[k] == "memory") has_mem = true, out.nz_mem = add64(out.nz_mem, v);
                }
            if (!has_cpu) out.nz_cpu = add64(out.nz_cpu, kDefaultMilliCPU);
            if (!has_mem) out.nz_mem = add64(out.nz_mem, kDefaultMemory);
        }
        return out;
    }
    for (const auto &n : names) out.req.push_back(aggregate_request(spec, n));
    const Value &pod_level = spec["resources"]["requests"];
    const bool pod_level_set = pod_level.truthy() && !pod_level.fields().empty();
    auto non_zero = [&](const std::string &name, int64_t dflt) {
        return aggregate_request(spec, name, !pod_level_set || !named_anywhere(spec, name) ? dflt : -1);
    };
    out.nz_cpu = non_zero("cpu", kDefaultMilliCPU);
    out.nz_mem = non_zero("memory", kDefaultMemory);
    return out;
}

inline std::string label_or(const Value &labels, const char *a, const char *b) {
    if (labels.has(a)) return labels[a].text();
    if (labels.has(b)) return labels[b].text();
    return "";
}
inline std::string zone_key(const Value &labels) { // component-helpers/node/topology/helpers.go:31-58
    const std::string zone = label_or(labels, "failure-domain.beta.kubernetes.io/zone", "topology.kubernetes.io/zone");
    const std::string region = label_or(labels, "failure-domain.beta.kubernetes.io/region", "topology.kubernetes.io/region");
    if (region.empty() && zone.empty()) return "";
    return region + std::string(":\0:", 3) + zone;
}

// node_tree.go:119-143: nodes arrive sorted by name (the fake tracker lists lexicographically,
// client-go/testing/fixture.go:847-855), zones in first-seen order, then round robin across zones
inline std::vector<const Value *> canonical_node_order(std::vector<const Value *> nodes) {
    // (sorted through pointers to the name strings: a comparator that looks the name up and copies it, as the first version did,
    // costs more than the rest of the node pass at 100k nodes)
    static const std::string no_name;
    std::vector<std::pair<const std::string *, const Value *>> by_name;
    by_name.reserve(nodes.size());
    for (const Value *n : nodes) {
        const Value &nm = (*n)["metadata"]["name"];
        by_name.emplace_back(nm.t == Value::Str || nm.t == Value::Num ? &nm.s : &no_name, n);
    }
    std::stable_sort(by_name.begin(), by_name.end(), [](const auto &a, const auto &b) { return *a.first < *b.first; });
    std::unordered_map<std::string, size_t> zone_of; // zone key -> position in first-seen order
    std::vector<std::vector<const Value *>> zones;
    for (const auto &kv : by_name) {
        const auto at = zone_of.emplace(zone_key((*kv.second)["metadata"]["labels"]), zones.size());
        if (at.second) zones.emplace_back();
        zones[at.first->second].push_back(kv.second);
    }
    std::vector<const Value *> out;
    out.reserve(nodes.size());
    for (size_t i = 0; out.size() < nodes.size(); i++)
        for (const auto &z : zones)
            if (i < z.size()) out.push_back(z[i]);
    return out;
}

// fn(k) for k in [0, n) on several threads (CCHOST_THREADS, default one per core, at most 32; small n: the calling thread).  An
// exception ends the thread's chunk; the one from the smallest k -- the one a serial loop would have raised -- is rethrown.
template <class Fn> inline void parallel_for(size_t n, Fn fn) {
    unsigned threads = std::thread::hardware_concurrency();
    if (const char *e = std::getenv("CCHOST_THREADS")) threads = (unsigned)std::atoi(e);
    size_t min_items = 4096; // (CCHOST_PARALLEL_MIN_ITEMS: tests set 0 to run small inputs through the threads)
    if (const char *e = std::getenv("CCHOST_PARALLEL_MIN_ITEMS")) min_items = (size_t)std::atoll(e);
    threads = n < min_items || n == 0 ? 1u : std::max(1u, std::min(threads, 32u));
    std::vector<std::pair<size_t, std::string>> errors(threads, {n, ""});
    auto work = [&](unsigned t) {
        for (size_t k = n * t / threads; k < n * (t + 1) / threads; k++) {
            try {
                fn(k);
            } catch (const std::exception &e) {
                errors[t] = {k, e.what()};
                return;
            }
        }
    };
    std::vector<std::thread> pool;
    for (unsigned t = 1; t < threads; t++) pool.emplace_back(work, t);
    work(0);
    for (auto &th : pool) th.join();
    const auto first = std::min_element(errors.begin(), errors.end(), [](const auto &a, const auto &b) { return a.first < b.first; });
    if (first->first < n) throw std::runtime_error(first->second);
}

// toleration.go:38-57 ToleratesTaint
inline bool tolerates(const Value &tol, const Value &taint) {
    if (tol["effect"].truthy() && tol["effect"].text() != taint["effect"].text()) return false;
    if (tol["key"].truthy() && tol["key"].text() != taint["key"].text()) return false;
    const std::string op = tol["operator"].truthy() ? tol["operator"].text() : "Equal";
    if (op == "Exists") return true;
    if (op == "Equal") return tol["value"].text() == taint["value"].text();
    return false;
}

struct TaintVerdict {
    bool filter_ok;
    int prefer_cnt;
    const Value *first; // first untolerated NoSchedule / NoExecute taint
};
inline TaintVerdict taint_verdict(const Value &taints, const Value &tolerations) {
    TaintVerdict v{true, 0, nullptr};
    auto any_tol = [&](const Value &t, bool prefer_only) {
        for (const auto &x : tolerations.items()) {
            if (prefer_only && x["effect"].truthy() && x["effect"].text() != "PreferNoSchedule") continue;
            if (tolerates(x, t)) return true;
        }
        return false;
    };
    for (const auto &t : taints.items()) {
        const std::string e = t["effect"].text();
        if ((e == "NoSchedule" || e == "NoExecute") && !any_tol(t, false)) {
            v.first = &t, v.filter_ok = false;
            break;
        }
    }
    for (const auto &t : taints.items())
        if (t["effect"].text() == "PreferNoSchedule" && !any_tol(t, true)) v.prefer_cnt++;
    return v;
}

// labels.Requirement.Matches (selector.go:246-293)
inline bool requirement_matches(bool key_present, const std::string &val, const std::string &op, const std::vector<std::string> &values) {
    auto in = [&] { return std::find(values.begin(), values.end(), val) != values.end(); };
    if (op == "In") return key_present && in();
    if (op == "NotIn") return !key_present || !in();
    if (op == "Exists") return key_present;
    if (op == "DoesNotExist") return !key_present;
    if (op == "Gt" || op == "Lt") {
        if (!key_present || values.size() != 1) return false;
        auto parse = [](const std::string &s, long long &out) { // strconv.ParseInt(s, 10, 64)
            if (s.empty()) return false;
            size_t i = (s[0] == '-' || s[0] == '+') ? 1 : 0;
            if (i >= s.size()) return false;
            for (size_t k = i; k < s.size(); k++)
                if (s[k] < '0' || s[k] > '9') return false;
            try {
                out = std::stoll(s);
            } catch (...) {
                return false;
            }
            return true;
        };
        long long a, b;
        if (!parse(val, a) || !parse(values[0], b)) return false;
        return op == "Gt" ? a > b : a < b;
    }
    return false;
}

inline std::vector<std::string> string_list(const Value &v) {
    std::vector<std::string> out;
    for (const auto &x : v.items()) out.push_back(x.text());
    return out;
}

// metav1.LabelSelectorAsSelector: nil -> Nothing, {} -> Everything (helpers.go:36-75)
inline bool label_selector_matches(const Value &sel, const Value &labels) {
    if (sel.is_null()) return false;
    for (const auto &kv : sel["matchLabels"].fields())
        if (!labels.has(kv.first) || labels[kv.first].text() != kv.second.text()) return false;
    for (const auto &e : sel["matchExpressions"].items()) {
        const std::string k = e["key"].text();
        if (!requirement_matches(labels.has(k), labels[k].text(), e["operator"].text(), string_list(e["values"]))) return false;
    }
    return true;
}
inline bool selector_empty(const Value &sel) { return !sel.is_null() && !sel["matchLabels"].truthy() && !sel["matchExpressions"].truthy(); }
} // namespace cchost
#include "volumes.hpp"
namespace cchost {

struct Requirement {
    int col;
    std::vector<uint8_t> table; // matches iff table[value id of the node's label] != 0
};
using Term = std::vector<Requirement>;

// label key -> column; label value -> id (0 = key absent).  Columns are created on demand.
struct Interner {
    const std::vector<const Value *> &nodes;
    std::vector<std::string> keys;
    std::vector<std::vector<std::string>> values; // per column: value strings, index = id - 1
    std::vector<std::vector<int32_t>> arrays;
    explicit Interner(const std::vector<const Value *> &n) : nodes(n) {}
    int col(const std::string &key) {
        for (size_t c = 0; c < keys.size(); c++)
            if (keys[c] == key) return (int)c;
        std::vector<std::string> vals;
        std::map<std::string, int> ids;
        std::vector<int32_t> arr(nodes.size(), 0);
        for (size_t i = 0; i < nodes.size(); i++) {
            const Value &md = (*nodes[i])["metadata"];
            std::string v;
            bool present;
            if (key == "metadata.name") present = true, v = md["name"].text();
            else present = md["labels"].has(key), v = md["labels"][key].text();
            if (!present) continue;
            auto it = ids.find(v);
            if (it == ids.end()) {
                vals.push_back(v);
                it = ids.emplace(v, (int)vals.size()).first;
            }
            arr[i] = it->second;
        }
        keys.push_back(key), values.push_back(vals), arrays.push_back(arr);
        return (int)keys.size() - 1;
    }
    Requirement table(const std::string &key, const std::string &op, const std::vector<std::string> &vals) {
        const int c = col(key);
        Requirement r{c, std::vector<uint8_t>(values[(size_t)c].size() + 1, 0)};
        r.table[0] = requirement_matches(false, "", op, vals);
        for (size_t i = 0; i < values[(size_t)c].size(); i++) r.table[i + 1] = requirement_matches(true, values[(size_t)c][i], op, vals);
        return r;
    }
};

inline Term node_selector_term(Interner &it, const Value &term) {
    Term reqs;
    for (const auto &e : term["matchExpressions"].items()) reqs.push_back(it.table(e["key"].text(), e["operator"].text(), string_list(e["values"])));
    for (const auto &f : term["matchFields"].items()) // only metadata.name with In / NotIn (nodeaffinity.go:260-293)
        reqs.push_back(it.table("metadata.name", f["operator"].text(), string_list(f["values"])));
    return reqs;
}

struct Spread {
    int col, max_skew, min_domains, n_domains;
    bool hard, self_match, is_hostname;
    std::vector<int32_t> node_match_count; // empty = none
    bool use_included = false;             // a node inclusion policy applies (common.go:107-122) -> node_included
    std::vector<uint8_t> node_included;
};

struct Ipa {
    std::vector<int> key_cols, key_ndom;
    std::vector<int> aff_keys;
    bool self_aff = false;
    std::vector<int32_t> aff_existing; // empty = none
    std::vector<int> anti_keys;
    std::vector<int> anti_self;
    std::vector<std::vector<int32_t>> anti_existing; // per term, empty = none
    std::vector<std::vector<int32_t>> exist_anti;    // per key, empty = none
    std::vector<std::vector<int64_t>> score_existing; // per key, empty = none
    std::vector<int64_t> score_self;
    std::vector<int32_t> self_entries;
    int64_t entries_existing = 0;
};

// What ONE template (simulated pod spec) contributes: requests, the verdict of its tolerations on every distinct taint set,
// selector / affinity terms as requirement tables over the shared label columns, spread constraints, inter-pod terms, ports, images.
struct PodSide {
    std::vector<int64_t> preq;
    int64_t pod_nz_cpu = 0, pod_nz_mem = 0;
    bool has_scalar_entries = false, tolerates_unschedulable = false, affinity_filter_active = false, has_node_selector = false,
         has_required_terms = false;
    std::vector<uint8_t> taint_filter_ok;
    std::vector<int32_t> taint_prefer_cnt;
    std::vector<std::string> taint_reasons; // per taint set: the FitError text of its first untolerated taint
    Term node_selector;
    std::vector<Term> required;
    std::vector<std::pair<int, Term>> preferred;
    std::vector<Spread> spread;
    bool soft_relaxed = false;     // `spread` holds the plugin's system default constraints: scored with requireAllTopologies = false
    std::vector<uint8_t> included; // RequiredNodeAffinity.Match per node (spread inclusion policy); empty = all
    bool has_ipa = false;
    Ipa ipa;
    bool has_host_ports = false;              // NodePorts: util.GetHostPorts(pod) is not empty
    std::vector<uint8_t> host_ports_conflict; // per node: an existing pod holds a conflicting port; empty = none does
    std::vector<uint8_t> image_score;         // ImageLocality score per node (0..100); empty = no image of the pod anywhere
    // the volume plugins (volumes.hpp): per node the code of the first of them that rejects it (empty = none does); the pod's own disks
    // conflict with a clone's; HOST ONLY: a PreFilter rejection (its message; empty = none) and the ReadWriteOncePod claim nobody uses yet
    std::vector<uint8_t> volume_veto;
    bool volume_exclusive = false;
    std::string prefilter_reject;
    bool rwop_capacity_one = false;
    // DefaultPreemption's dry run of the terminal cycle (HOST ONLY, never crosses the C ABI; preemption.hpp): a victim is an
    // existing pod of lower priority than the template (default_preemption.go:392-396)
    int64_t priority = 0;
    bool preempt_never = false;                   // spec.preemptionPolicy == Never
    std::vector<int32_t> victim_count;            // per node; empty = no node holds a victim
    std::vector<std::vector<int64_t>> victim_req; // per resource column, per node
    std::vector<uint8_t> ports_conflict_rest;     // a REMAINING pod of the node holds a conflicting host port; empty = not evaluated
    // a victim of the node takes part in the PreFilter state of a topology-coupled filter of the template (removing it would
    // change that state: not modelled by the dry run); empty = no such node
    std::vector<uint8_t> victim_interacts;
    std::vector<uint8_t> volume_veto_rest; // volume_veto with the node's victims gone (volumes.hpp veto_with_victims_gone); empty = no node is rejected then
    // a topology-coupled FILTER is active: a DoNotSchedule spread constraint (podtopologyspread/filtering.go:311-356), required inter-pod
    // (anti-)affinity of the template or anti-affinity terms of existing pods that match it (interpodaffinity/filtering.go:352-432).
    // With one, WHICH nodes a cycle saw decides not only the order of the placements but how many fit: the total depends on
    // percentageOfNodesToScore (host/engine.hpp simulate() keeps the reference's default for such templates).
    bool hard_coupled() const {
        for (const auto &c : spread)
            if (c.hard) return true;
        if (has_ipa) {
            if (!ipa.aff_keys.empty() || !ipa.anti_keys.empty()) return true;
            for (const auto &v : ipa.exist_anti)
                if (!v.empty()) return true;
        }
        return false;
    }
};

// The snapshot: node columns shared by every template + the first template (as base class: the single-template code reads
// s.preq, s.spread, ... as before) + the further templates of a `--podspec a --podspec b ...` run (`more`).
// The pod once a clone holds its ReadWriteOncePod claim: VolumeRestrictions fails every node -- after its own disk check (`clones`: where
// the pod's earlier clones sit, for a pod whose disks are exclusive: the engine counts clones from the moment a pod is set).
inline void rwop_now_in_use(PodSide &side, size_t N, const std::vector<int32_t> *clones = nullptr) {
    if (side.volume_veto.empty()) side.volume_veto.assign(N, 0);
    for (size_t i = 0; i < N; i++) {
        if (side.volume_exclusive && clones && (*clones)[i] > 0) side.volume_veto[i] = 1;
        if (side.volume_veto[i] != 1) side.volume_veto[i] = 2;
    }
    side.rwop_capacity_one = false;
}

struct Snapshot : PodSide {
    // nodes (canonical order)
    std::vector<std::string> names, res_names, scalar_names;
    std::vector<std::vector<int64_t>> alloc, req;
    std::vector<int32_t> alloc_pods, pod_count, taintset_id;
    std::vector<int64_t> nz_mcpu, nz_mem;
    std::vector<uint8_t> unschedulable;
    std::vector<std::string> label_keys;
    std::vector<std::vector<int32_t>> label_cols;
    std::vector<PodSide> more; // templates 1 .. P-1
    bool default_spreading_unmodelled = false; // system default spreading applies but a node lacks the hostname / zone label
    size_t n() const { return names.size(); }
    size_t n_templates() const { return 1 + more.size(); }
    const PodSide &side(size_t t) const { return t == 0 ? static_cast<const PodSide &>(*this) : more[t - 1]; }
};

// ---- host ports (NodePorts) --------------------------------------------------------------------------------------
struct HostPort {
    std::string ip, protocol;
    int64_t port;
};
// util.GetHostPorts (S/util/utils.go:175-210): hostPort > 0 of the restartable init containers and of the containers,
// sanitized as HostPortInfo does ("" -> 0.0.0.0 / TCP, kube-scheduler/framework/types.go:530-538)
inline std::vector<HostPort> host_ports(const Value &spec) {
    std::vector<HostPort> out;
    auto take = [&](const Value &c) {
        for (const auto &p : c["ports"].items()) {
            const int64_t hp = p["hostPort"].as_int32();
            if (hp > 0) out.push_back({p["hostIP"].truthy() ? p["hostIP"].text() : "0.0.0.0", p["protocol"].truthy() ? p["protocol"].text() : "TCP", hp});
        }
    };
    for (const auto &c : spec["initContainers"].items())
        if (c["restartPolicy"].text() == "Always") take(c);
    for (const auto &c : spec["containers"].items()) take(c);
    return out;
}
// fitsPorts (node_ports.go:164-176) over HostPortInfo.CheckConflict (types.go:499-528): 0.0.0.0 conflicts with every ip
inline bool ports_conflict(const std::vector<HostPort> &want, const std::vector<HostPort> &used) {
    for (const auto &w : want)
        for (const auto &u : used)
            if (u.protocol == w.protocol && u.port == w.port && (w.ip == "0.0.0.0" || u.ip == "0.0.0.0" || u.ip == w.ip)) return true;
    return false;
}

// ---- image locality (P/imagelocality/image_locality.go:54-127) ---------------------------------------------------
inline std::string normalized_image_name(const std::string &name) { // :122-127
    const size_t colon = name.rfind(':'), slash = name.rfind('/');
    const long long ci = colon == std::string::npos ? -1 : (long long)colon, si = slash == std::string::npos ? -1 : (long long)slash;
    return ci <= si ? name + ":latest" : name;
}
// calculatePriority(sumImageScores) :84-115; scaledImageScore is fp64: int64(float64(Size) * (NumNodes / total))
inline int64_t image_locality_score(const std::vector<std::pair<int64_t, int64_t>> &sizes_and_spread, int64_t total_nodes, int64_t n_containers) {
    const int64_t mb = 1024 * 1024, lo = 23 * mb, hi = 1000 * mb * n_containers;
    int64_t sum = 0;
    for (const auto &x : sizes_and_spread) {
        const double spread = (double)x.second / (double)total_nodes;
        const double scaled = (double)x.first * spread;
        sum += (int64_t)scaled;
    }
    sum = sum < lo ? lo : (sum > hi ? hi : sum);
    return 100 * (sum - lo) / (hi - lo);
}

// AffinityTerm.Matches (S/framework/types.go:927-935): the pod's namespace is in the term's set, or its namespace's labels
// match the term's namespaceSelector; then the label selector decides.  newAffinityTerm (:879-895): no namespaces and no
// namespaceSelector -> the namespace of the pod that owns the term.  A namespace without a Namespace object in the snapshot
// has no labels (as the scheduler's lister would report).
using NamespaceLabels = std::map<std::string, Value>;
inline bool term_matches_pod(const Value &term, const std::string &owner_ns, const std::string &pod_ns, const Value &pod_labels,
                             const NamespaceLabels &ns_labels) {
    std::vector<std::string> ns_set = string_list(term["namespaces"]);
    const Value &ns_sel = term["namespaceSelector"];
    if (ns_set.empty() && ns_sel.is_null()) ns_set.push_back(owner_ns);
    bool in = std::find(ns_set.begin(), ns_set.end(), pod_ns) != ns_set.end();
    if (!in && !ns_sel.is_null()) {
        const auto it = ns_labels.find(pod_ns);
        in = label_selector_matches(ns_sel, it == ns_labels.end() ? Value::null_value() : it->second);
    }
    return in && label_selector_matches(term["labelSelector"], pod_labels);
}
inline std::string ns_of(const Value &obj) { return obj["metadata"]["namespace"].truthy() ? obj["metadata"]["namespace"].text() : "default"; }

static const char *const kZone = "topology.kubernetes.io/zone";

// helper.DefaultSelector (P/helper/spread.go:37-116) as a LabelSelector value, Null when it is empty: the merged selectors of the Services
// of the pod's namespace that select it, plus the selector of the pod's controller (a pod spec copied from a live pod carries its
// ownerReferences): a ReplicationController's map, a ReplicaSet's / StatefulSet's label selector.  `objs`: Services and controllers.
inline Value default_selector(const Value &sim_pod, const std::vector<Value> &objs) {
    const std::string ns = ns_of(sim_pod);
    const Value &labels = sim_pod["metadata"]["labels"];
    Value merged = Value::object(), exprs = Value::array();
    for (const auto &svc : objs) {
        if (svc["kind"].text() != "Service" || ns_of(svc) != ns) continue;
        const Value &sel = svc["spec"]["selector"];
        if (sel.t != Value::Obj) continue; // a nil selector matches nothing (spread.go:105-108)
        bool all = true;
        for (const auto &kv : sel.o) all = all && labels.has(kv.first) && labels[kv.first].text() == kv.second.text();
        if (all)
            for (const auto &kv : sel.o) merged.set(kv.first, kv.second);
    }
    for (const auto &ref : sim_pod["metadata"]["ownerReferences"].items()) {
        if (!ref["controller"].truthy()) continue;
        const std::string kind = ref["kind"].text(), api = ref["apiVersion"].text();
        for (const auto &o : objs) {
            if (o["kind"].text() != kind || o["metadata"]["name"].text() != ref["name"].text() || ns_of(o) != ns) continue;
            const Value &sel = o["spec"]["selector"];
            if (kind == "ReplicationController" && (api.empty() || api == "v1")) {
                for (const auto &kv : sel.fields()) merged.set(kv.first, kv.second);
            } else if ((kind == "ReplicaSet" || kind == "StatefulSet") && api == "apps/v1") {
                for (const auto &kv : sel["matchLabels"].fields()) merged.set(kv.first, kv.second);
                for (const auto &e : sel["matchExpressions"].items()) exprs.a.push_back(e);
            }
        }
        break; // (GetControllerOf: the first reference marked controller)
    }
    if (merged.o.empty() && exprs.a.empty()) return Value();
    Value out = Value::object();
    out.set("matchLabels", merged), out.set("matchExpressions", exprs);
    return out;
}

// PodTopologySpread's SYSTEM DEFAULT constraints for a pod WITHOUT constraints of its own (P/podtopologyspread/plugin.go:48-59,
// common.go:61-74): hostname maxSkew 3 and zone maxSkew 5, ScheduleAnyway, selector = helper.DefaultSelector; empty when that is empty
inline Value system_default_constraints(const Value &sim_pod, const std::vector<Value> &objs) {
    Value out = Value::array();
    if (sim_pod["spec"]["topologySpreadConstraints"].truthy()) return out;
    const Value sel = default_selector(sim_pod, objs);
    if (sel.is_null()) return out;
    for (const auto &kv : {std::pair<const char *, int>{kHostname, 3}, std::pair<const char *, int>{kZone, 5}}) {
        Value c = Value::object();
        c.set("maxSkew", Value::num(kv.second)), c.set("topologyKey", Value::str(kv.first)), c.set("whenUnsatisfiable", Value::str("ScheduleAnyway")), c.set("labelSelector", sel);
        out.a.push_back(c);
    }
    return out;
}


inline bool any_nonzero(const std::vector<int32_t> &v) {
    for (auto x : v)
        if (x) return true;
    return false;
}

inline Snapshot build_snapshot(const std::vector<Value> &node_objs, const std::vector<Value> &pod_objs, const std::vector<Value> &sim_pods,
                               const std::vector<std::string> &exclude_nodes, int hard_pod_affinity_weight = 1,
                               const std::vector<Value> &namespace_objs = {}, const std::vector<Value> &spreading_objs = {}, bool system_default_spreading = true,
                               const VolumeObjects *volume_objs = nullptr) {
    // spreading_objs: the Services and controllers (ReplicationController / ReplicaSet / StatefulSet) of the dump: helper.DefaultSelector
    // volume_objs: the claims, classes (and, beyond the reference, volumes) of the dump + which volume plugins run; nullptr = none, all four
    static const VolumeObjects no_volume_objs;
    const VolumeObjects &vol = volume_objs ? *volume_objs : no_volume_objs;
    if (sim_pods.empty()) throw std::runtime_error("no pod spec");
    Snapshot S;
    NamespaceLabels ns_labels;
    for (const auto &n : namespace_objs) ns_labels[n["metadata"]["name"].text()] = n["metadata"]["labels"];
    auto tm = [&](const Value &term, const std::string &owner_ns, const std::string &pod_ns, const Value &pod_labels) {
        return term_matches_pod(term, owner_ns, pod_ns, pod_labels, ns_labels);
    };
    std::vector<const Value *> kept;
    for (const auto &n : node_objs)
        if (std::find(exclude_nodes.begin(), exclude_nodes.end(), n["metadata"]["name"].text()) == exclude_nodes.end()) kept.push_back(&n);
    const std::vector<const Value *> nodes = canonical_node_order(kept);
    const size_t N = nodes.size();
    std::unordered_map<std::string, size_t> index; // node name -> position (hundreds of thousands of pod -> node lookups)
    index.reserve(nodes.size() * 2);
    for (size_t i = 0; i < N; i++) {
        S.names.push_back((*nodes[i])["metadata"]["name"].text());
        index[S.names.back()] = i;
    }
    // resources: cpu, memory, ephemeral-storage + every scalar resource a template names
    std::set<std::string> req_names;
    for (const Value &sp : sim_pods) {
        const Value &spec = sp["spec"];
        for (const char *list : {"containers", "initContainers"})
            for (const auto &c : spec[list].items())
                for (const auto &kv : c["resources"]["requests"].fields()) req_names.insert(kv.first);
        for (const auto &kv : spec["resources"]["requests"].fields()) req_names.insert(kv.first); // pod-level requests (hugepages-*)
        for (const auto &kv : spec["overhead"].fields()) req_names.insert(kv.first);
    }
    for (const auto &n : req_names) // std::set iterates sorted
        if (is_scalar_resource(n)) S.scalar_names.push_back(n);
    if ((int)S.scalar_names.size() > CCSIM_MAX_SCALAR) // never drop a resource silently: the Fit filter would over-estimate
        throw std::runtime_error("the pod names " + std::to_string(S.scalar_names.size()) + " scalar/extended resources; at most " +
                                 std::to_string(CCSIM_MAX_SCALAR) + " are supported");
    S.res_names = {"cpu", "memory", "ephemeral-storage"};
    S.res_names.insert(S.res_names.end(), S.scalar_names.begin(), S.scalar_names.end());
    const size_t R = S.res_names.size();

    S.alloc.assign(R, std::vector<int64_t>(N, 0));
    S.req.assign(R, std::vector<int64_t>(N, 0));
    S.alloc_pods.assign(N, 0), S.pod_count.assign(N, 0), S.nz_mcpu.assign(N, 0), S.nz_mem.assign(N, 0);
    parallel_for(N, [&](size_t i) {
        const Value &a = (*nodes[i])["status"]["allocatable"];
        for (size_t c = 0; c < R; c++) S.alloc[c][i] = res_of(a, S.res_names[c]);
        const int64_t ap = a.has("pods") ? quantity_value(a["pods"].text()) : 0;
        if (ap > INT32_MAX || ap < 0) throw std::runtime_error("node " + (*nodes[i])["metadata"]["name"].text() + ": allocatable pods " + a["pods"].text() + " is out of range");
        S.alloc_pods[i] = (int32_t)ap;
    });
    std::vector<const Value *> live; // non-terminal pods bound to a kept node (simulator.go:193-200)
    std::vector<size_t> live_node;
    std::vector<int64_t> live_prio; // spec.priority of the live pods (who is a victim depends on the template: DefaultPreemption)
    bool live_has_terms = false;    // some live pod carries inter-pod (anti)affinity terms
    {
        // Walking every pod's object tree (phase, node name, the containers' request quantities) is memory-latency bound and
        // independent per pod: it runs on several threads into flat per-pod records; folding the records into the node columns
        // (and the `live` list, in pod order) is a cheap serial pass, so the result does not depend on the thread count.
        const size_t P = pod_objs.size(), W = R + 2;
        std::vector<int64_t> rec(P * W);
        std::vector<int64_t> pod_node(P, -1), prio(P, 0);
        std::vector<uint8_t> terms(P, 0);
        parallel_for(P, [&](size_t k) {
            const Value &p = pod_objs[k];
            const std::string phase = p["status"]["phase"].text();
            if (phase == "Succeeded" || phase == "Failed") return;
            const auto at = index.find(p["spec"]["nodeName"].text());
            if (at == index.end()) return;
            const PodRequests r = pod_requests(p["spec"], S.res_names);
            for (size_t c = 0; c < R; c++) rec[k * W + c] = r.req[c];
            rec[k * W + R] = r.nz_cpu, rec[k * W + R + 1] = r.nz_mem;
            prio[k] = p["spec"]["priority"].as_int32(0); // corev1helpers.PodPriority
            const Value &aff = p["spec"]["affinity"];
            terms[k] = aff["podAffinity"].truthy() || aff["podAntiAffinity"].truthy();
            pod_node[k] = (int64_t)at->second;
        });
        for (size_t k = 0; k < P; k++) {
            if (pod_node[k] < 0) continue;
            const size_t i = (size_t)pod_node[k];
            live.push_back(&pod_objs[k]), live_node.push_back(i);
            live_prio.push_back(prio[k]), live_has_terms = live_has_terms || terms[k];
            for (size_t c = 0; c < R; c++) S.req[c][i] = add64(S.req[c][i], rec[k * W + c]);
            S.nz_mcpu[i] = add64(S.nz_mcpu[i], rec[k * W + R]), S.nz_mem[i] = add64(S.nz_mem[i], rec[k * W + R + 1]), S.pod_count[i] += 1;
        }
    }

    // taints -> distinct taint sets (per node); what a template's tolerations make of each set is the template's business
    std::map<std::string, int> sets;
    std::vector<const Value *> set_taints;
    S.taintset_id.assign(N, 0);
    for (size_t i = 0; i < N; i++) {
        const Value &taints = (*nodes[i])["spec"]["taints"];
        std::string key;
        for (const auto &t : taints.items()) key += t["key"].text() + '\x1f' + t["value"].text() + '\x1f' + t["effect"].text() + '\x1e';
        auto it = sets.find(key);
        if (it == sets.end()) {
            it = sets.emplace(key, (int)sets.size()).first;
            set_taints.push_back(&taints);
        }
        S.taintset_id[i] = it->second;
    }
    S.unschedulable.assign(N, 0);
    for (size_t i = 0; i < N; i++) S.unschedulable[i] = (*nodes[i])["spec"]["unschedulable"].truthy();
    Interner it(nodes); // shared by the templates: a label column per key any of them touches

    auto template_side = [&](const Value &sim_pod, PodSide &s, size_t template_index) {
    const Value &spec = sim_pod["spec"];
    const std::string sim_ns = ns_of(sim_pod);
    const Value &sim_labels = sim_pod["metadata"]["labels"];
    const PodRequests pr = pod_requests(spec, S.res_names);
    s.preq = pr.req, s.pod_nz_cpu = pr.nz_cpu, s.pod_nz_mem = pr.nz_mem;
    for (const auto &n : S.scalar_names) { // len(ScalarResources) != 0: the template itself names a scalar resource
        bool named = spec["resources"]["requests"].has(n) || spec["overhead"].has(n);
        for (const char *list : {"containers", "initContainers"})
            for (const auto &c : spec[list].items()) named = named || c["resources"]["requests"].has(n);
        s.has_scalar_entries = s.has_scalar_entries || named;
    }
    const Value &tolerations = spec["tolerations"];
    for (const Value *taints : set_taints) {
        const TaintVerdict v = taint_verdict(*taints, tolerations);
        s.taint_filter_ok.push_back(v.filter_ok), s.taint_prefer_cnt.push_back(v.prefer_cnt);
        // taint_toleration.go:119
        s.taint_reasons.push_back(v.first ? "node(s) had untolerated taint {" + (*v.first)["key"].text() + ": " + (*v.first)["value"].text() + "}" : "");
    }
    {
        Value unsched = Value::object();
        unsched.set("key", Value::str(kUnschedTaint)), unsched.set("effect", Value::str("NoSchedule"));
        for (const auto &t : tolerations.items())
            if (tolerates(t, unsched)) s.tolerates_unschedulable = true;
    }

    // node affinity / node selector
    const Value &aff = spec["affinity"]["nodeAffinity"];
    const Value &node_selector = spec["nodeSelector"];
    const Value &req_aff = aff["requiredDuringSchedulingIgnoredDuringExecution"];
    const Value &required = req_aff.truthy() ? req_aff["nodeSelectorTerms"] : Value::null_value();
    for (const auto &kv : node_selector.fields()) s.node_selector.push_back(it.table(kv.first, "In", {kv.second.text()}));
    for (const auto &t : required.items()) s.required.push_back(node_selector_term(it, t));
    for (const auto &t : aff["preferredDuringSchedulingIgnoredDuringExecution"].items())
        s.preferred.emplace_back(t["weight"].as_int32(), node_selector_term(it, t["preference"]));
    s.has_node_selector = node_selector.truthy();
    s.has_required_terms = !required.is_null();
    s.affinity_filter_active = s.has_node_selector || s.has_required_terms;

    auto node_matches_required = [&](size_t i) { // RequiredNodeAffinity.Match, for the spread inclusion policy
        auto term_ok = [&](const Term &reqs, bool empty) {
            if (reqs.empty()) return empty;
            for (const auto &r : reqs)
                if (!r.table[(size_t)it.arrays[(size_t)r.col][i]]) return false;
            return true;
        };
        if (s.has_node_selector && !term_ok(s.node_selector, true)) return false;
        if (s.has_required_terms) {
            bool any = false;
            for (const auto &t : s.required) any = any || term_ok(t, false);
            if (!any) return false;
        }
        return true;
    };
    if (s.affinity_filter_active) {
        s.included.assign(N, 0);
        for (size_t i = 0; i < N; i++) s.included[i] = node_matches_required(i);
    }

    // (DefaultPreemption dry run) which live pods are victims of this template, and on which nodes a victim takes part in the
    // PreFilter state of one of the template's topology-coupled FILTERS: it matches a hard spread selector or a required
    // (anti)affinity term, or carries an anti-affinity term matching the template
    std::vector<uint8_t> is_victim(live.size(), 0), interacts(N, 0);
    // NodePorts: which nodes' existing pods already hold one of the pod's host ports
    {
        const std::vector<HostPort> want = host_ports(spec);
        if (!want.empty()) {
            s.has_host_ports = true;
            std::vector<std::vector<HostPort>> used(N);
            for (size_t j = 0; j < live.size(); j++)
                for (const auto &hp : host_ports((*live[j])["spec"])) used[live_node[j]].push_back(hp);
            std::vector<uint8_t> conflict(N, 0);
            bool any = false;
            for (size_t i = 0; i < N; i++) conflict[i] = !used[i].empty() && ports_conflict(want, used[i]), any = any || conflict[i];
            if (any) s.host_ports_conflict = conflict;
        }
        // DefaultPreemption dry run (report only): what removing every lower-priority pod of a node would free
        // (corev1helpers.PodPriority: spec.priority, 0 when unset)
        s.priority = spec["priority"].as_int32(0);
        s.preempt_never = spec["preemptionPolicy"].text() == "Never";
        bool any_victim = false;
        for (size_t j = 0; j < live.size(); j++) is_victim[j] = live_prio[j] < s.priority, any_victim = any_victim || is_victim[j];
        if (any_victim) {
            s.victim_count.assign(N, 0);
            s.victim_req.assign(S.res_names.size(), std::vector<int64_t>(N, 0));
            for (size_t j = 0; j < live.size(); j++) {
                if (!is_victim[j]) continue;
                const PodRequests r = pod_requests((*live[j])["spec"], S.res_names);
                s.victim_count[live_node[j]] += 1;
                for (size_t c = 0; c < r.req.size(); c++) s.victim_req[c][live_node[j]] = add64(s.victim_req[c][live_node[j]], r.req[c]);
            }
            if (!want.empty()) {
                std::vector<std::vector<HostPort>> rest(N);
                for (size_t j = 0; j < live.size(); j++)
                    if (!is_victim[j])
                        for (const auto &hp : host_ports((*live[j])["spec"])) rest[live_node[j]].push_back(hp);
                s.ports_conflict_rest.assign(N, 0);
                for (size_t i = 0; i < N; i++) s.ports_conflict_rest[i] = !rest[i].empty() && ports_conflict(want, rest[i]);
            }
        }
    }
    // ImageLocality: the scheduler cache's image states (cache.go:680-703): Size = what the FIRST node added (nodes arrive
    // sorted by name) reports for the image name, NumNodes = nodes listing the name
    {
        // (only the images the template names are looked at: a cluster dump carries tens of image names per node)
        std::vector<std::string> wanted;
        for (const char *list : {"initContainers", "containers"})
            for (const auto &c : spec[list].items()) wanted.push_back(normalized_image_name(c["image"].text()));
        bool present = false; // no node lists an image of the template (the usual case): nothing to sort, nothing to score
        for (size_t i = 0; i < N && !present; i++)
            for (const auto &img : (*nodes[i])["status"]["images"].items())
                for (const auto &nm : img["names"].items()) present = present || (nm.t == Value::Str && std::find(wanted.begin(), wanted.end(), nm.s) != wanted.end());
        std::vector<size_t> by_name(present ? N : 0);
        for (size_t i = 0; i < by_name.size(); i++) by_name[i] = i;
        std::sort(by_name.begin(), by_name.end(), [&](size_t a, size_t b) { return S.names[a] < S.names[b]; });
        std::map<std::string, int64_t> size;
        std::map<std::string, std::set<size_t>> holders;
        for (size_t i : by_name)
            for (const auto &img : (*nodes[i])["status"]["images"].items())
                for (const auto &nm : img["names"].items()) {
                    if (nm.t != Value::Str || std::find(wanted.begin(), wanted.end(), nm.s) == wanted.end()) continue;
                    size.emplace(nm.s, img["sizeBytes"].truthy() ? img["sizeBytes"].as_int() : 0);
                    holders[nm.s].insert(i);
                }
        bool any = false;
        for (const auto &w : wanted) any = any || holders.count(w);
        if (any) {
            s.image_score.assign(N, 0);
            for (size_t i = 0; i < N; i++) {
                std::vector<std::pair<int64_t, int64_t>> held; // (size, number of nodes holding it) of the template's images on node i
                for (const auto &w : wanted) {
                    const auto h = holders.find(w);
                    if (h != holders.end() && h->second.count(i)) held.emplace_back(size[w], (int64_t)h->second.size());
                }
                s.image_score[i] = (uint8_t)image_locality_score(held, (int64_t)N, (int64_t)wanted.size());
            }
        }
    }
    // VolumeRestrictions / NodeVolumeLimits / VolumeBinding / VolumeZone: the object side (volumes.hpp) -> per-node verdict codes, the
    // clones' own disks, the PreFilter rejections.  (inline csi volumes only count against CSINode limits, which the simulated cluster
    // does not have: nodevolumelimits/csi.go:265-290.)  DynamicResources stays refused.
    {
        VolumeSide vs = volume_side(sim_pod, nodes, live, live_node, vol, template_index);
        if (!s.victim_count.empty() && !vs.veto.empty() && !vs.rejected) // (DefaultPreemption's dry run: the verdicts once a node's victims are gone)
            s.volume_veto_rest = veto_with_victims_gone(sim_pod, nodes, live, live_node, is_victim, vol, vs, template_index);
        s.volume_veto = std::move(vs.veto), s.volume_exclusive = vs.exclusive;
        s.prefilter_reject = vs.rejected ? vs.prefilter_reject : std::string(), s.rwop_capacity_one = vs.rwop_capacity_one;
    }
    if (spec["resourceClaims"].truthy() && vol.dra_enabled) {
        // DynamicResources' PreFilter runs after the volume plugins' (default_plugins.go:45-47); the fake cluster holds no ResourceClaim
        if (vol.dra_partial)
            throw Unsupported("the scheduler configuration disables only the filter point of DynamicResources: a pod with resourceClaims is not modelled under it");
        if (s.prefilter_reject.empty()) s.prefilter_reject = dra_prefilter(sim_pod, template_index);
    }

    // topology spread constraints (common.go:86-127); NodeAffinityPolicy defaults to Honor, NodeTaintsPolicy to Ignore
    // System default spreading (a Service / the controller selects the template): two more ScheduleAnyway constraints of the pod, scored
    // with requireAllTopologies = false (scoring.go:61-115,140: a node without a key is not ignored, the missing key counts as the value
    // "" when the domains are sized and scores nothing) -- PodSide::soft_relaxed; marshal() derives the engine's form (include/ccsim.h
    // missing_value).  Left out, and the caller told (Snapshot::default_spreading_unmodelled), only when a node lacks the HOSTNAME label
    // (the per-node constraint has no column to read then) or several templates run.
    Value constraints = Value::array();
    constraints.a = spec["topologySpreadConstraints"].items(); // (null = none; any other kind than a list is refused)
    if (constraints.a.empty() && system_default_spreading) {
        const Value defaults = system_default_constraints(sim_pod, spreading_objs);
        if (!defaults.a.empty()) {
            bool all = sim_pods.size() == 1; // (several templates: the engine keeps each template's spread state apart, a shared selector would couple them)
            for (size_t i = 0; i < N && all; i++) all = (*nodes[i])["metadata"]["labels"].has(kHostname);
            if (all) constraints = defaults, s.soft_relaxed = true;
            else S.default_spreading_unmodelled = true;
        }
    }
    for (const auto &c : constraints.items()) {
        // matchLabelKeys (common.go:95-105): the incoming pod's own values of these keys are ANDed into the selector
        Value merged_sel = c["labelSelector"];
        if (!merged_sel.is_null()) {
            Value exprs = merged_sel["matchExpressions"].t == Value::Arr ? merged_sel["matchExpressions"] : Value::array();
            bool any = false;
            for (const auto &k : c["matchLabelKeys"].items())
                if (sim_labels.has(k.text())) {
                    Value e = Value::object(), vals = Value::array();
                    vals.a.push_back(Value::str(sim_labels[k.text()].text()));
                    e.set("key", Value::str(k.text())), e.set("operator", Value::str("In")), e.set("values", vals);
                    exprs.a.push_back(e), any = true;
                }
            if (any) merged_sel.set("matchExpressions", exprs);
        }
        const Value &sel = merged_sel;
        Spread k{};
        k.col = it.col(c["topologyKey"].text());
        std::vector<int32_t> existing(N, 0);
        for (size_t p = 0; p < live.size(); p++) { // countPodsMatchSelector (common.go:144-159)
            const Value &pod = *live[p];
            if (!selector_empty(sel) && ns_of(pod) == sim_ns && !pod["metadata"]["deletionTimestamp"].truthy() && label_selector_matches(sel, pod["metadata"]["labels"])) {
                existing[live_node[p]] += 1;
                if (is_victim[p] && (c["whenUnsatisfiable"].truthy() ? c["whenUnsatisfiable"].text() : "DoNotSchedule") == "DoNotSchedule") interacts[live_node[p]] = 1;
            }
        }
        // matchNodeInclusionPolicies (common.go:107-122): required node affinity / selector (default Honor) and the
        // NoSchedule / NoExecute taints the pod does not tolerate (default Ignore)
        const std::string aff_policy = c["nodeAffinityPolicy"].truthy() ? c["nodeAffinityPolicy"].text() : "Honor";
        const std::string taint_policy = c["nodeTaintsPolicy"].truthy() ? c["nodeTaintsPolicy"].text() : "Ignore";
        k.max_skew = c["maxSkew"].as_int32();
        k.min_domains = c["minDomains"].truthy() ? c["minDomains"].as_int32() : 1;
        k.hard = (c["whenUnsatisfiable"].truthy() ? c["whenUnsatisfiable"].text() : "DoNotSchedule") == "DoNotSchedule";
        k.self_match = !selector_empty(sel) && label_selector_matches(sel, sim_labels);
        k.is_hostname = c["topologyKey"].text() == kHostname;
        k.n_domains = (int)it.values[(size_t)k.col].size();
        if (any_nonzero(existing)) k.node_match_count = existing;
        const bool honor_aff = aff_policy == "Honor" && s.affinity_filter_active, honor_taints = taint_policy == "Honor";
        k.use_included = honor_aff || honor_taints;
        if (k.use_included) {
            k.node_included.assign(N, 1);
            for (size_t i = 0; i < N; i++) {
                if (honor_aff && !s.included[i]) k.node_included[i] = 0;
                if (honor_taints && !s.taint_filter_ok[(size_t)S.taintset_id[i]]) k.node_included[i] = 0;
            }
        }
        s.spread.push_back(std::move(k));
    }

    // inter-pod affinity (filtering.go:204-432, scoring.go:81-125)
    const Value &pa = spec["affinity"]["podAffinity"], &paa = spec["affinity"]["podAntiAffinity"];
    const Value &r_aff = pa["requiredDuringSchedulingIgnoredDuringExecution"], &r_anti = paa["requiredDuringSchedulingIgnoredDuringExecution"];
    const Value &p_aff = pa["preferredDuringSchedulingIgnoredDuringExecution"], &p_anti = paa["preferredDuringSchedulingIgnoredDuringExecution"];
    const bool others_have_terms = live_has_terms;
    if (r_aff.truthy() || r_anti.truthy() || p_aff.truthy() || p_anti.truthy() || others_have_terms) {
        std::vector<std::string> keys;
        auto kidx = [&](const std::string &k) {
            for (size_t i = 0; i < keys.size(); i++)
                if (keys[i] == k) return (int)i;
            keys.push_back(k);
            return (int)keys.size() - 1;
        };
        Ipa &ipa = s.ipa;
        for (const auto &t : r_aff.items()) ipa.aff_keys.push_back(kidx(t["topologyKey"].text()));
        ipa.self_aff = r_aff.truthy();
        for (const auto &t : r_aff.items()) ipa.self_aff = ipa.self_aff && tm(t, sim_ns, sim_ns, sim_labels);
        for (const auto &t : r_anti.items()) {
            ipa.anti_keys.push_back(kidx(t["topologyKey"].text()));
            ipa.anti_self.push_back(tm(t, sim_ns, sim_ns, sim_labels));
        }
        std::vector<int32_t> aff_existing(N, 0);
        std::vector<std::vector<int32_t>> anti_existing(r_anti.items().size(), std::vector<int32_t>(N, 0));
        std::map<int, std::vector<int32_t>> exist_anti;
        std::map<int, std::vector<int64_t>> score_existing;
        int64_t entries = 0;
        auto add_score = [&](int k, size_t i, int64_t w) {
            const int col = it.col(keys[(size_t)k]);
            if (it.arrays[(size_t)col][i]) {
                auto &v = score_existing[k];
                if (v.empty()) v.assign(N, 0);
                v[i] += w;
                entries++;
            }
        };
        for (size_t pi = 0; pi < live.size(); pi++) {
            const Value &p = *live[pi];
            const size_t i = live_node[pi];
            const std::string p_ns = ns_of(p);
            const Value &pl = p["metadata"]["labels"];
            if (r_aff.truthy()) {
                bool all = true;
                for (const auto &t : r_aff.items()) all = all && tm(t, sim_ns, p_ns, pl);
                if (all) aff_existing[i] += 1, interacts[i] |= is_victim[pi];
            }
            for (size_t t = 0; t < r_anti.items().size(); t++)
                if (tm(r_anti.items()[t], sim_ns, p_ns, pl)) anti_existing[t][i] += 1, interacts[i] |= is_victim[pi];
            const Value &e_aff = p["spec"]["affinity"]["podAffinity"], &e_anti = p["spec"]["affinity"]["podAntiAffinity"];
            for (const auto &t : e_anti["requiredDuringSchedulingIgnoredDuringExecution"].items())
                if (tm(t, p_ns, sim_ns, sim_labels)) {
                    auto &v = exist_anti[kidx(t["topologyKey"].text())];
                    if (v.empty()) v.assign(N, 0);
                    v[i] += 1;
                    interacts[i] |= is_victim[pi];
                }
            // scoring.go:81-125 processExistingPod
            for (const auto &wt : p_aff.items())
                if (tm(wt["podAffinityTerm"], sim_ns, p_ns, pl)) add_score(kidx(wt["podAffinityTerm"]["topologyKey"].text()), i, wt["weight"].as_int32());
            for (const auto &wt : p_anti.items())
                if (tm(wt["podAffinityTerm"], sim_ns, p_ns, pl)) add_score(kidx(wt["podAffinityTerm"]["topologyKey"].text()), i, -wt["weight"].as_int32());
            if (hard_pod_affinity_weight > 0)
                for (const auto &t : e_aff["requiredDuringSchedulingIgnoredDuringExecution"].items())
                    if (tm(t, p_ns, sim_ns, sim_labels)) add_score(kidx(t["topologyKey"].text()), i, hard_pod_affinity_weight);
            for (const auto &wt : e_aff["preferredDuringSchedulingIgnoredDuringExecution"].items())
                if (tm(wt["podAffinityTerm"], p_ns, sim_ns, sim_labels)) add_score(kidx(wt["podAffinityTerm"]["topologyKey"].text()), i, wt["weight"].as_int32());
            for (const auto &wt : e_anti["preferredDuringSchedulingIgnoredDuringExecution"].items())
                if (tm(wt["podAffinityTerm"], p_ns, sim_ns, sim_labels)) add_score(kidx(wt["podAffinityTerm"]["topologyKey"].text()), i, -wt["weight"].as_int32());
        }
        // what ONE clone adds (it is an existing pod of the next cycle, with the incoming pod's own terms): both directions --
        // the incoming pod's term vs the clone, and the clone's term vs the incoming pod
        std::map<int, int64_t> score_self;
        std::map<int, int32_t> self_entries;
        auto self_term = [&](const Value &wt, int sign) {
            if (tm(wt["podAffinityTerm"], sim_ns, sim_ns, sim_labels)) {
                const int k = kidx(wt["podAffinityTerm"]["topologyKey"].text());
                score_self[k] += 2 * sign * wt["weight"].as_int32();
                self_entries[k] += 2;
            }
        };
        for (const auto &wt : p_aff.items()) self_term(wt, 1);
        for (const auto &wt : p_anti.items()) self_term(wt, -1);
        if (hard_pod_affinity_weight > 0)
            for (const auto &t : r_aff.items())
                if (tm(t, sim_ns, sim_ns, sim_labels)) {
                    const int k = kidx(t["topologyKey"].text());
                    score_self[k] += hard_pod_affinity_weight;
                    self_entries[k] += 1;
                }
        if ((int)keys.size() > CCSIM_MAX_IPA_KEYS) throw Unsupported("more than " + std::to_string(CCSIM_MAX_IPA_KEYS) + " distinct inter-pod affinity topology keys");
        if ((int)ipa.aff_keys.size() > CCSIM_MAX_IPA_TERMS || (int)ipa.anti_keys.size() > CCSIM_MAX_IPA_TERMS) throw Unsupported("too many inter-pod affinity terms");
        for (const auto &k : keys) {
            ipa.key_cols.push_back(it.col(k));
            ipa.key_ndom.push_back((int)it.values[(size_t)ipa.key_cols.back()].size());
        }
        if (any_nonzero(aff_existing)) ipa.aff_existing = aff_existing;
        for (auto &a : anti_existing) ipa.anti_existing.push_back(any_nonzero(a) ? a : std::vector<int32_t>());
        for (int k = 0; k < (int)keys.size(); k++) {
            ipa.exist_anti.push_back(exist_anti.count(k) ? exist_anti[k] : std::vector<int32_t>());
            ipa.score_existing.push_back(score_existing.count(k) ? score_existing[k] : std::vector<int64_t>());
            ipa.score_self.push_back(score_self.count(k) ? score_self[k] : 0);
            ipa.self_entries.push_back(self_entries.count(k) ? self_entries[k] : 0);
        }
        ipa.entries_existing = entries;
        s.has_ipa = true;
    }
    if ((int)s.spread.size() > CCSIM_MAX_TSC) throw Unsupported("too many topology spread constraints");
    if (std::find(interacts.begin(), interacts.end(), (uint8_t)1) != interacts.end()) s.victim_interacts = interacts;
    }; // template_side

    template_side(sim_pods[0], S, 0);
    for (size_t t = 1; t < sim_pods.size(); t++) {
        S.more.emplace_back();
        template_side(sim_pods[t], S.more.back(), t);
    }
    S.label_keys = it.keys;
    S.label_cols = it.arrays;
    if ((int)S.label_cols.size() > CCSIM_MAX_LABEL_COLS) throw Unsupported("too many distinct label keys in selectors");
    // several templates: a clone's disks exclude clones of the SAME template from its node (volume_exclusive); a disk or a
    // ReadWriteOncePod claim shared by two templates would exclude the other's clones too, which nothing tracks
    if (sim_pods.size() > 1) {
        std::set<std::pair<std::string, std::string>> rwop;
        for (const auto &o : vol.claims)
            for (const auto &m : o["spec"]["accessModes"].items())
                if (m.text() == "ReadWriteOncePod") rwop.insert({ns_of(o), o["metadata"]["name"].text()});
        auto claims = [&](const Value &p) {
            std::set<std::pair<std::string, std::string>> out;
            for (const auto &v : p["spec"]["volumes"].items())
                if (!v["persistentVolumeClaim"].is_null()) out.insert({ns_of(p), v["persistentVolumeClaim"]["claimName"].text()});
            return out;
        };
        for (size_t a = 0; a < sim_pods.size(); a++)
            for (size_t b = a + 1; b < sim_pods.size(); b++) {
                if (pod_conflicts(sim_pods[a]["spec"]["volumes"], sim_pods[b]["spec"]["volumes"]))
                    throw Unsupported("templates " + std::to_string(a) + " and " + std::to_string(b) + " mount the same disk: conflicts between clones of different templates are not modelled");
                const auto ca = claims(sim_pods[a]), cb = claims(sim_pods[b]);
                for (const auto &c : ca)
                    if (cb.count(c) && rwop.count(c))
                        throw Unsupported("templates " + std::to_string(a) + " and " + std::to_string(b) + " share a ReadWriteOncePod claim: not modelled");
            }
    }
    // several templates: what a clone contributes to the plugin state of later cycles is kept per template (the engine's
    // ccsim_set_pods contract): no selector of one template may match the clones of another
    if (sim_pods.size() > 1)
        for (size_t a = 0; a < sim_pods.size(); a++) {
            std::vector<const Value *> sels;
            const Value &spec = sim_pods[a]["spec"];
            for (const auto &c : spec["topologySpreadConstraints"].items()) sels.push_back(&c["labelSelector"]);
            for (const char *kind : {"podAffinity", "podAntiAffinity"}) {
                const Value &k = spec["affinity"][kind];
                for (const auto &t : k["requiredDuringSchedulingIgnoredDuringExecution"].items()) sels.push_back(&t["labelSelector"]);
                for (const auto &t : k["preferredDuringSchedulingIgnoredDuringExecution"].items()) sels.push_back(&t["podAffinityTerm"]["labelSelector"]);
            }
            for (size_t b = 0; b < sim_pods.size(); b++)
                for (const Value *sel : sels)
                    if (b != a && !sel->is_null() && !selector_empty(*sel) && label_selector_matches(*sel, sim_pods[b]["metadata"]["labels"]))
                        throw Unsupported("several templates: a selector of template " + std::to_string(a) + " matches the labels of template " + std::to_string(b));
        }
    return S;
}

inline Snapshot build_snapshot(const std::vector<Value> &node_objs, const std::vector<Value> &pod_objs, const Value &sim_pod,
                               const std::vector<std::string> &exclude_nodes, int hard_pod_affinity_weight = 1,
                               const std::vector<Value> &namespace_objs = {}, const std::vector<Value> &spreading_objs = {}, bool system_default_spreading = true,
                               const VolumeObjects *volume_objs = nullptr) {
    return build_snapshot(node_objs, pod_objs, std::vector<Value>{sim_pod}, exclude_nodes, hard_pod_affinity_weight, namespace_objs, spreading_objs, system_default_spreading,
                          volume_objs);
}

// ---- the dump the CPU tests compare with the Python ingest (tests/test_native_host.py) -------------------------
template <class T> inline Value int_array(const std::vector<T> &v) {
    Value a = Value::array();
    for (auto x : v) a.a.push_back(Value::num((long long)x));
    return a;
}
inline Value term_json(const Term &t) {
    Value a = Value::array();
    for (const auto &r : t) {
        Value o = Value::object();
        o.set("col", Value::num(r.col)), o.set("table", int_array(r.table));
        a.a.push_back(o);
    }
    return a;
}
inline Value pod_side_json(const PodSide &s) {
    Value p = Value::object();
    p.set("req", int_array(s.preq)), p.set("nz_mcpu", Value::num(s.pod_nz_cpu)), p.set("nz_mem", Value::num(s.pod_nz_mem));
    p.set("has_scalar_entries", Value::boolean(s.has_scalar_entries)), p.set("taint_filter_ok", int_array(s.taint_filter_ok));
    p.set("taint_prefer_cnt", int_array(s.taint_prefer_cnt)), p.set("tolerates_unschedulable", Value::boolean(s.tolerates_unschedulable));
    p.set("affinity_filter_active", Value::boolean(s.affinity_filter_active)), p.set("has_node_selector", Value::boolean(s.has_node_selector));
    p.set("has_required_terms", Value::boolean(s.has_required_terms)), p.set("node_selector", term_json(s.node_selector));
    Value rq = Value::array(), pf = Value::array(), sp = Value::array();
    for (auto &t : s.required) rq.a.push_back(term_json(t));
    for (auto &t : s.preferred) {
        Value e = Value::object();
        e.set("weight", Value::num(t.first)), e.set("term", term_json(t.second));
        pf.a.push_back(e);
    }
    for (auto &k : s.spread) {
        Value e = Value::object();
        e.set("col", Value::num(k.col)), e.set("max_skew", Value::num(k.max_skew)), e.set("min_domains", Value::num(k.min_domains));
        e.set("hard", Value::boolean(k.hard)), e.set("self_match", Value::boolean(k.self_match)), e.set("is_hostname", Value::boolean(k.is_hostname));
        e.set("n_domains", Value::num(k.n_domains));
        e.set("node_match_count", k.node_match_count.empty() ? Value() : int_array(k.node_match_count));
        e.set("node_included", k.use_included ? int_array(k.node_included) : Value());
        sp.a.push_back(e);
    }
    p.set("required", rq), p.set("preferred", pf), p.set("spread", sp);
    p.set("soft_relaxed", Value::boolean(s.soft_relaxed));
    if (s.has_ipa) {
        const Ipa &a = s.ipa;
        Value e = Value::object();
        e.set("key_cols", int_array(a.key_cols)), e.set("key_ndom", int_array(a.key_ndom)), e.set("aff_keys", int_array(a.aff_keys));
        e.set("self_aff", Value::boolean(a.self_aff)), e.set("aff_existing", a.aff_existing.empty() ? Value() : int_array(a.aff_existing));
        e.set("anti_keys", int_array(a.anti_keys)), e.set("anti_self", int_array(a.anti_self));
        Value ae = Value::array(), ea = Value::array(), se = Value::array();
        for (auto &v : a.anti_existing) ae.a.push_back(v.empty() ? Value() : int_array(v));
        for (auto &v : a.exist_anti) ea.a.push_back(v.empty() ? Value() : int_array(v));
        for (auto &v : a.score_existing) se.a.push_back(v.empty() ? Value() : int_array(v));
        e.set("anti_existing", ae), e.set("exist_anti", ea), e.set("score_existing", se);
        e.set("score_self", int_array(a.score_self)), e.set("self_entries", int_array(a.self_entries)), e.set("entries_existing", Value::num(a.entries_existing));
        p.set("ipa", e);
    } else
        p.set("ipa", Value());
    p.set("has_host_ports", Value::boolean(s.has_host_ports));
    p.set("host_ports_conflict", s.host_ports_conflict.empty() ? Value() : int_array(s.host_ports_conflict));
    p.set("image_score", s.image_score.empty() ? Value() : int_array(s.image_score));
    p.set("volume_veto", s.volume_veto.empty() ? Value() : int_array(s.volume_veto));
    p.set("volume_exclusive", Value::boolean(s.volume_exclusive));
    p.set("prefilter_reject", s.prefilter_reject.empty() ? Value() : Value::str(s.prefilter_reject));
    p.set("rwop_capacity_one", Value::boolean(s.rwop_capacity_one));
    Value pre = Value::object();
    pre.set("priority", Value::num(s.priority)), pre.set("never", Value::boolean(s.preempt_never));
    pre.set("victim_count", s.victim_count.empty() ? Value() : int_array(s.victim_count));
    Value vr = Value::array();
    for (auto &v : s.victim_req) vr.a.push_back(int_array(v));
    pre.set("victim_req", vr);
    pre.set("ports_conflict_rest", s.ports_conflict_rest.empty() ? Value() : int_array(s.ports_conflict_rest));
    pre.set("victim_interacts", s.victim_interacts.empty() ? Value() : int_array(s.victim_interacts));
    pre.set("volume_veto_rest", s.volume_veto_rest.empty() ? Value() : int_array(s.volume_veto_rest));
    p.set("preempt", pre);
    return p;
}
inline Value snapshot_json(const Snapshot &s) {
    Value o = Value::object();
    auto strs = [](const std::vector<std::string> &v) {
        Value a = Value::array();
        for (auto &x : v) a.a.push_back(Value::str(x));
        return a;
    };
    o.set("names", strs(s.names)), o.set("res_names", strs(s.res_names)), o.set("scalar_names", strs(s.scalar_names));
    o.set("taint_reasons", strs(s.taint_reasons));
    Value alloc = Value::array(), req = Value::array(), cols = Value::array();
    for (auto &c : s.alloc) alloc.a.push_back(int_array(c));
    for (auto &c : s.req) req.a.push_back(int_array(c));
    for (auto &c : s.label_cols) cols.a.push_back(int_array(c));
    o.set("alloc", alloc), o.set("req", req), o.set("label_cols", cols), o.set("label_keys", strs(s.label_keys));
    o.set("alloc_pods", int_array(s.alloc_pods)), o.set("pod_count", int_array(s.pod_count)), o.set("taintset_id", int_array(s.taintset_id));
    o.set("nz_mcpu", int_array(s.nz_mcpu)), o.set("nz_mem", int_array(s.nz_mem)), o.set("unschedulable", int_array(s.unschedulable));
    o.set("pod", pod_side_json(s));
    if (!s.more.empty()) { // templates 1 .. P-1
        Value more = Value::array();
        for (const auto &m : s.more) more.a.push_back(pod_side_json(m));
        o.set("more_pods", more);
    }
    return o;
}

} // namespace cchost
