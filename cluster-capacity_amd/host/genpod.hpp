// genpod.hpp -- the pod a namespace's limits describe (the reference's cmd/genpod, pkg/client/nspod.go:34-126), from the
// snapshot's Namespace and LimitRange objects (the reference asks the API server for them).
//   * resources: for memory, cpu and nvdia.com/gpu (sic, nspod.go:31) the MINIMUM over all LimitRange items of type Pod of
//     max[resource] (:72-88, Quantity.Cmp); if any of them is non-zero the stub container gets them as limits AND requests;
//   * nodeSelector: the namespace's openshift.io/node-selector annotation, "k1=v1,k2=v2" (:119-126,
//     apimachinery labels.ConvertSelectorToLabelsMap, labels.go:159-183).
#pragma once
#include <map>
#include <string>
#include <vector>

#include "quantity.hpp"
#include "value.hpp"

namespace cchost {

// Quantity.Cmp on the exact values: mant * 10^exp10 * 2^exp2 (cross-scaled in 128 bits; quantities of a LimitRange are far
// from the range where that overflows)
inline int quantity_cmp(const Quantity &a, const Quantity &b) {
    __int128 x = a.mant, y = b.mant;
    const int e10 = a.exp10 < b.exp10 ? a.exp10 : b.exp10, e2 = a.exp2 < b.exp2 ? a.exp2 : b.exp2;
    for (int k = 0; k < a.exp10 - e10; k++) x *= 10;
    for (int k = 0; k < b.exp10 - e10; k++) y *= 10;
    for (int k = 0; k < a.exp2 - e2; k++) x *= 2;
    for (int k = 0; k < b.exp2 - e2; k++) y *= 2;
    return x < y ? -1 : (x > y ? 1 : 0);
}

inline std::string trim(const std::string &s) {
    size_t b = 0, e = s.size();
    while (b < e && (s[b] == ' ' || s[b] == '\t')) b++;
    while (e > b && (s[e - 1] == ' ' || s[e - 1] == '\t')) e--;
    return s.substr(b, e - b);
}

inline Value selector_to_labels(const std::string &selector) {
    Value out = Value::object();
    if (selector.empty()) return out;
    size_t pos = 0;
    while (pos <= selector.size()) {
        const size_t comma = selector.find(',', pos);
        const std::string label = selector.substr(pos, comma == std::string::npos ? std::string::npos : comma - pos);
        const size_t eq = label.find('=');
        if (eq == std::string::npos || label.find('=', eq + 1) != std::string::npos) throw std::runtime_error("invalid selector: " + label);
        out.set(trim(label.substr(0, eq)), Value::str(trim(label.substr(eq + 1))));
        if (comma == std::string::npos) break;
        pos = comma + 1;
    }
    return out;
}

inline Value namespace_pod(const std::string &ns_name, const std::vector<Value> &namespaces, const std::vector<Value> &limit_ranges) {
    const Value *ns = nullptr;
    for (const auto &n : namespaces)
        if (n["metadata"]["name"].text() == ns_name) ns = &n;
    if (!ns) throw std::runtime_error("Namespace " + ns_name + " not found");
    Value container = Value::object();
    container.set("name", Value::str("cluster-capacity-stub-container"));
    container.set("image", Value::str("gcr.io/google_containers/pause:2.0"));
    container.set("imagePullPolicy", Value::str("Always"));
    static const char *kResources[] = {"memory", "cpu", "nvdia.com/gpu"}; // nspod.go:66-70 (the typo is the reference's)
    std::map<std::string, std::string> best;
    for (const auto &lr : limit_ranges) {
        const std::string lr_ns = lr["metadata"]["namespace"].truthy() ? lr["metadata"]["namespace"].text() : "default";
        if (lr_ns != ns_name) continue;
        for (const auto &item : lr["spec"]["limits"].items()) {
            if (item["type"].text() != "Pod") continue;
            for (const char *r : kResources) {
                if (!item["max"].has(r)) continue;
                const std::string amount = item["max"][r].text();
                auto it = best.find(r);
                if (it == best.end() || quantity_cmp(parse_quantity(it->second), parse_quantity(amount)) > 0) best[r] = amount;
            }
        }
    }
    bool nonzero = false;
    for (const auto &kv : best) nonzero = nonzero || parse_quantity(kv.second).mant != 0;
    if (nonzero) {
        Value limits = Value::object(), requests = Value::object(), res = Value::object();
        for (const char *r : kResources) {
            auto it = best.find(r);
            if (it == best.end()) continue;
            // (the serializer prints the Quantity, not the text it was read from: "0.5" leaves as "500m", "1024Mi" as "1Gi")
            const std::string canon = quantity_canonical(quantity_nano(parse_quantity(it->second)), quantity_format(it->second));
            limits.set(r, Value::str(canon)), requests.set(r, Value::str(canon));
        }
        res.set("limits", limits), res.set("requests", requests);
        container.set("resources", res);
    }
    Value containers = Value::array();
    containers.a.push_back(container);
    Value spec = Value::object();
    spec.set("containers", containers);
    spec.set("restartPolicy", Value::str("OnFailure")), spec.set("dnsPolicy", Value::str("Default"));
    const Value &ann = (*ns)["metadata"]["annotations"];
    if (ann.has("openshift.io/node-selector")) {
        const std::string sel = ann["openshift.io/node-selector"].text();
        try {
            spec.set("nodeSelector", selector_to_labels(sel));
        } catch (const std::exception &e) {
            throw std::runtime_error("Unable to parse openshift.io/node-selector in " + sel + " namespace: " + e.what());
        }
    }
    Value meta = Value::object();
    meta.set("name", Value::str("cluster-capacity-stub-container")), meta.set("namespace", Value::str(ns_name));
    Value pod = Value::object();
    pod.set("apiVersion", Value::str("v1")), pod.set("kind", Value::str("Pod")), pod.set("metadata", meta), pod.set("spec", spec);
    return pod;
}

} // namespace cchost
