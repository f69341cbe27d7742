// cluster-capacity (native host) -- the reference's command restated in C++ above the C ABI of the MI355X engine.
//
//   cluster-capacity --podspec pod.yaml --snapshot cluster.json [--snapshot more.yaml]
//                    [--max-limit N] [--exclude-nodes a,b] [--verbose] [-o json|yaml]
//
// Flags mirror cmd/cluster-capacity/app/options/options.go:65-77; Run / Report mirror pkg/framework/simulator.go:356-381
// and pkg/framework/report.go.  There is no API server here, so --kubeconfig is replaced by --snapshot: files holding the
// Node and Pod objects SyncWithClient would list (simulator.go:176-295) -- `kubectl get nodes,pods -A -o json` output, a
// List, or a multi-document YAML stream.  --default-config takes a KubeSchedulerConfiguration (profile.hpp: plugin sets,
// weights, scoring resources, percentageOfNodesToScore); --percentage-of-nodes-to-score overrides its percentage.
//
// The simulation runs on the GPU through include/ccsim.h (libccsim.so, loaded at run time); there is NO CPU fallback: if
// the library or the device is missing the command fails.  Test hooks (no GPU needed): --dump-snapshot prints the integer
// snapshot the ingest produced, --fake-result renders the report for a result read from a file.
#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <climits>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>

#include "cluster_capacity.hpp"
#include "genpod.hpp"

using namespace cchost;

namespace {

std::string read_file(const std::string &path) {
    std::ifstream f(path, std::ios::binary | std::ios::ate);
    if (!f) throw std::runtime_error("cannot open " + path);
    const std::streamoff size = f.tellg();
    if (size < 0) { // not seekable (a pipe): stream it
        std::ostringstream ss;
        f.clear(), ss << f.rdbuf();
        return ss.str();
    }
    std::string text((size_t)size, '\0');
    f.seekg(0);
    f.read(text.data(), size);
    return text;
}

// A snapshot file as text without copying it: regular files are mapped (a dump of a large cluster is hundreds of MB to GBs;
// zero-filling a string of that size and copying the page cache into it costs as much as scanning it), anything else is read.
class FileText {
  public:
    explicit FileText(const std::string &path) {
        const int fd = ::open(path.c_str(), O_RDONLY);
        if (fd < 0) throw std::runtime_error("cannot open " + path);
        struct stat st;
        if (::fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 0) {
            void *p = ::mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
            if (p != MAP_FAILED) {
                ::madvise(p, (size_t)st.st_size, MADV_SEQUENTIAL);
                map_ = p, len_ = (size_t)st.st_size;
            }
        }
        ::close(fd);
        if (!map_) owned_ = read_file(path);
    }
    ~FileText() {
        if (map_) ::munmap(map_, len_);
    }
    FileText(const FileText &) = delete;
    FileText &operator=(const FileText &) = delete;
    std::string_view view() const { return map_ ? std::string_view((const char *)map_, len_) : std::string_view(owned_); }

  private:
    void *map_ = nullptr;
    size_t len_ = 0;
    std::string owned_;
};

// Every object of the snapshot files goes to `sink` exactly once, MOVED out of the parsed documents (a kubectl dump of a large
// cluster is hundreds of MB: copying the documents, as the first version did, cost more than parsing them); lists are flattened.
template <class Sink> void for_each_object(const std::vector<std::string> &paths, Sink sink, const std::vector<std::string> *wanted_images = nullptr) {
    for (const auto &path : paths) {
        const FileText file(path);
        std::vector<Value> docs = parse_documents(file.view(), /*prune_cluster_objects=*/true, wanted_images);
        for (Value &d : docs) {
            if (!d.truthy()) continue;
            const std::string kind = d["kind"].text();
            const bool is_list = kind.size() >= 4 && kind.compare(kind.size() - 4, 4, "List") == 0 && d.has("items");
            if (!is_list) {
                sink(std::move(d));
                continue;
            }
            for (auto &kv : d.o)
                if (kv.first == "items")
                    for (Value &o : kv.second.a) sink(std::move(o));
        }
    }
}

// every object of one kind in the snapshot files
std::vector<Value> load_kind(const std::vector<std::string> &paths, const std::string &want) {
    std::vector<Value> out;
    for_each_object(paths, [&](Value &&o) {
        if (o["kind"].text() == want) out.push_back(std::move(o));
    });
    return out;
}

// `templates`: the simulated pods -- the Nodes' status.images entries are kept only for the images their containers name
void load_objects(const std::vector<std::string> &paths, const std::vector<Value> &templates, std::vector<Value> &nodes, std::vector<Value> &pods,
                  std::vector<Value> &namespaces, std::vector<Value> &services, VolumeObjects &vol) {
    // spec.volumes of the dump's pods are materialised only when a template has volumes the volume plugins compare them with
    bool template_volumes = false;
    for (const auto &t : templates)
        for (const auto &v : t["spec"]["volumes"].items()) template_volumes = template_volumes || restricted(v) || !v["persistentVolumeClaim"].is_null();
    prune::keep_pod_volumes() = template_volumes;
    std::vector<std::string> wanted;
    for (const auto &t : templates)
        for (const char *list : {"initContainers", "containers"})
            for (const auto &c : t["spec"][list].items()) wanted.push_back(normalized_image_name(c["image"].text()));
    for_each_object(paths, [&](Value &&o) {
        const std::string kind = o["kind"].text();
        if (kind == "Node") nodes.push_back(std::move(o));
        else if (kind == "Pod") pods.push_back(std::move(o));
        else if (kind == "Namespace") namespaces.push_back(std::move(o));
        else if (kind == "Service" || kind == "ReplicationController" || kind == "ReplicaSet" || kind == "StatefulSet") services.push_back(std::move(o));
        else if (kind == "PersistentVolumeClaim") vol.claims.push_back(std::move(o)); // SyncWithClient copies claims and classes (simulator.go:228-295) ...
        else if (kind == "StorageClass") vol.classes.push_back(std::move(o));
        else if (kind == "PersistentVolume" && vol.sync_volumes) vol.volumes.push_back(std::move(o)); // ... not the volumes (--sync-persistent-volumes)
        else if (kind == "CSINode" && vol.sync_volumes) vol.csinodes.push_back(std::move(o));           // (... nor what NodeVolumeLimits counts against)
        else if (kind == "VolumeAttachment" && vol.sync_volumes) vol.attachments.push_back(std::move(o));
    }, &wanted);
}

// options.go:79-147 ParseAPISpec (defaults only; API validation is the apiserver's job)
Value parse_pod_spec(const std::string &path) {
    std::vector<Value> docs = parse_documents(read_file(path));
    if (docs.empty() || docs[0]["kind"].text() != "Pod") throw std::runtime_error("Failed to decode config file: not a Pod");
    Value pod = docs[0];
    auto setdefault = [](Value &obj, const std::string &k, Value v) {
        if (!obj.has(k)) obj.set(k, std::move(v));
    };
    if (!pod.has("metadata")) pod.set("metadata", Value::object());
    if (!pod.has("spec")) pod.set("spec", Value::object());
    for (auto &kv : pod.o) {
        if (kv.first == "metadata") setdefault(kv.second, "namespace", Value::str("default"));
        if (kv.first == "spec") {
            setdefault(kv.second, "schedulerName", Value::str("default-scheduler"));
            setdefault(kv.second, "dnsPolicy", Value::str("ClusterFirst"));
            setdefault(kv.second, "restartPolicy", Value::str("Always"));
            for (auto &sk : kv.second.o)
                if (sk.first == "containers")
                    for (auto &c : sk.second.a) setdefault(c, "terminationMessagePolicy", Value::str("FallbackToLogsOnError"));
        }
    }
    if (!pod["spec"]["containers"].truthy()) throw std::runtime_error("Invalid pod: spec.containers: Required value");
    return pod;
}

RunResult result_from_json(const Value &v) {
    RunResult r;
    r.placed = v["placed"].as_int(), r.stop = (int32_t)v["stop"].as_int(), r.n_code_unschedulable = v["n_code_unschedulable"].as_int();
    for (const auto &x : v["per_node_count"].items()) r.per_node_count.push_back((int32_t)x.as_int());
    for (const auto &x : v["log"].items()) r.log.push_back((int32_t)x.as_int());
    for (const auto &x : v["hist"].items()) r.hist.push_back(x.as_int());
    for (const auto &x : v["hist_taintset"].items()) r.hist_taintset.push_back(x.as_int());
    if (v.has("stop_spec")) r.stop_spec = (int32_t)v["stop_spec"].as_int(); // several templates: the one whose pod did not fit
    return r;
}

int usage(const char *msg) {
    std::fprintf(stderr, "%s\nusage: cluster-capacity --podspec FILE [--podspec FILE ...] --snapshot FILE [--snapshot FILE ...] [--max-limit N] [--exclude-nodes a,b]\n"
                         "                        [--default-config FILE] [--verbose] [-o json|yaml] [--mode batched|sequential]\n"
                         "                        [--percentage-of-nodes-to-score P] [--sync-persistent-volumes] [--device D] [--gpus N]\n"
                         "       cluster-capacity --genpod NAMESPACE --snapshot FILE [--snapshot FILE ...] [-o json|yaml]\n",
                 msg);
    return 2;
}

} // namespace

int main(int argc, char **argv) {
    std::string output, mode, dump, fake, sched_config, genpod_ns;
    std::vector<std::string> podspecs; // repeatable: the templates are cycled round-robin
    bool dump_profile = false, pct_flag = false;
    std::vector<std::string> snapshots, exclude;
    int64_t max_limit = 0;
    int percentage = 100, device = 0, gpus = 1;
    bool force_sharded = false;
    bool sync_volumes = false;
    bool verbose = false;
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i], val;
        const size_t eq = a.find('=');
        bool has_val = false;
        if (a.rfind("--", 0) == 0 && eq != std::string::npos) val = a.substr(eq + 1), a = a.substr(0, eq), has_val = true;
        auto need = [&]() -> std::string {
            if (has_val) return val;
            if (i + 1 >= argc) throw std::runtime_error("flag needs an argument: " + a);
            return argv[++i];
        };
        try {
            if (a == "--podspec") podspecs.push_back(need());
            else if (a == "--snapshot") snapshots.push_back(need());
            else if (a == "--max-limit") max_limit = std::stoll(need());
            else if (a == "--exclude-nodes") {
                std::stringstream ss(need());
                for (std::string x; std::getline(ss, x, ',');)
                    if (!x.empty()) exclude.push_back(x);
            } else if (a == "--help" || a == "-h") {
                usage("Cluster-capacity is used for simulating scheduling of one or multiple pods"); // (the reference's command description, cmd/cluster-capacity/app/server.go:56)
                return 0;
            } else if (a == "--verbose") verbose = true;
            else if (a == "-o" || a == "--output") output = need();
            else if (a == "--mode") mode = need();
            else if (a == "--percentage-of-nodes-to-score") percentage = std::stoi(need()), pct_flag = true;
            else if (a == "--default-config") sched_config = need();
            else if (a == "--sync-persistent-volumes") sync_volumes = true; // beyond the reference: PersistentVolume objects are taken too
            else if (a == "--dump-profile") dump_profile = true;
            else if (a == "--genpod") genpod_ns = need(); // cmd/genpod: the pod a namespace's LimitRanges / annotations describe
            else if (a == "--device") device = std::stoi(need());
            else if (a == "--gpus") gpus = std::stoi(need()); // node-range shards over GPUs 0 .. N-1 of this box
            else if (a == "--force-sharded") force_sharded = true; // (test hook: the sharded path with one rank)
            else if (a == "--dump-snapshot") dump = need();
            else if (a == "--fake-result") fake = need();
            else if (a == "--parse") { // test hook: the documents of a file, as JSON (one array)
                Value all = Value::array();
                for (const Value &d : parse_documents(read_file(need()))) all.a.push_back(d);
                std::string out;
                to_json(out, all);
                std::cout << out << "\n";
                return 0;
            }
            else if (a == "--kubeconfig") return usage("not supported here: --kubeconfig (see --snapshot)");
            else return usage(("unknown flag " + a).c_str());
        } catch (const std::exception &e) {
            return usage(e.what());
        }
    }
    try { // --default-config: Path to JSON or YAML file containing scheduler configuration (options.go:73)
        if (dump_profile) {
            HostProfile hp = profile_from_config(sched_config.empty() ? Value() : parse_documents(read_file(sched_config)).at(0));
            if (pct_flag) hp.c.percentage_of_nodes_to_score = percentage, hp.percentage_set = true;
            std::string out;
            to_json(out, profile_json(hp));
            std::cout << out << "\n";
            return 0;
        }
    } catch (const std::exception &e) {
        std::fprintf(stderr, "cluster-capacity: %s\n", e.what());
        return 1;
    }
    if (!genpod_ns.empty()) { // genpod --namespace NS [--output json|yaml] (cmd/genpod/app/server.go:36-104); objects from --snapshot
        if (snapshots.empty()) return usage("--snapshot is required");
        if (!output.empty() && output != "json" && output != "yaml") return usage(("Output format " + output + " not recognized: only json and yaml are allowed").c_str());
        try {
            const Value pod = namespace_pod(genpod_ns, load_kind(snapshots, "Namespace"), load_kind(snapshots, "LimitRange"));
            std::string out;
            if (output == "json") to_json(out, pod), out += "\n";
            else to_yaml(out, sorted_keys(pod)); // PrintPod (pkg/utils/utils.go:47-71): yaml unless json is asked for
            std::cout << out;
            return 0;
        } catch (const std::exception &e) {
            std::fprintf(stderr, "Error: %s\n", e.what());
            return 1;
        }
    }
    if (podspecs.empty()) return usage("Pod spec file is missing"); // options.go / server.go:71-73
    if (snapshots.empty()) return usage("--snapshot is required");
    if (!output.empty() && output != "json" && output != "yaml") return usage("output format must be json or yaml");
    try {
        HostProfile prof = profile_from_config(sched_config.empty() ? Value() : parse_documents(read_file(sched_config)).at(0));
        if (pct_flag) prof.c.percentage_of_nodes_to_score = percentage, prof.percentage_set = true;
        // runSimulator (cmd/cluster-capacity/app/server.go:163-183): New -> SyncWithClient -> Run -> Report
        std::vector<Value> templates;
        for (const auto &p : podspecs) templates.push_back(parse_pod_spec(p));
        ClusterCapacity cc = ClusterCapacity::New(prof, templates, max_limit, exclude);
        cc.device = device, cc.mode = mode, cc.gpus = gpus, cc.force_sharded = force_sharded;
        // CCHOST_TIMING=1: wall time of the host phases on stderr (read + parse, intern + integer snapshot, engine run, report)
        const bool timing = std::getenv("CCHOST_TIMING") != nullptr;
        auto now = [] { return std::chrono::steady_clock::now(); };
        auto lap = [&](const char *what, std::chrono::steady_clock::time_point &t0) {
            const auto t1 = now();
            if (timing) std::fprintf(stderr, "[timing] %-28s %9.1f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
            t0 = t1;
        };
        auto t0 = now();
        std::vector<Value> node_objs, pod_objs, ns_objs, svc_objs;
        VolumeObjects vol;
        vol.sync_volumes = sync_volumes;
        load_objects(snapshots, templates, node_objs, pod_objs, ns_objs, svc_objs, vol);
        lap("read + parse objects", t0);
        cc.SyncWithClient(node_objs, pod_objs, ns_objs, svc_objs, std::move(vol));
        if (cc.snapshot().default_spreading_unmodelled)
            std::fprintf(stderr, "warning: a Service (or its controller) selects the simulated pod and it has no topologySpreadConstraints of its own: the scheduler's "
                                 "system default spreading (hostname maxSkew 3, zone maxSkew 5, ScheduleAnyway) would score the nodes too; some node lacks one of the "
                                 "two labels (or several templates run): not modelled -- the order of the placements (and so a --max-limit result) may differ, the total does not\n");
        lap("intern + integer snapshot", t0);
        if (!dump.empty()) {
            std::string out;
            to_json(out, snapshot_json(cc.snapshot()));
            if (dump == "-") std::cout << out << "\n";
            else std::ofstream(dump) << out << "\n";
            lap("dump snapshot", t0);
            std::cout.flush(), std::fflush(nullptr);
            if (!std::getenv("CCHOST_FULL_EXIT")) std::_Exit(0); // (see the end of main)
            return 0;
        }
        if (fake.empty()) cc.Run();
        else cc.SetResult(result_from_json(parse_documents(read_file(fake)).at(0)));
        lap("marshal + engine run", t0);
        const Value review = cc.Report();
        lap("report", t0);
        cc.Close();
        std::string out;
        if (output == "json") to_json(out, review), out += "\n";
        else if (output == "yaml") to_yaml(out, sorted_keys(review));
        else out = pretty(review, verbose);
        std::cout << out;
        // the object trees of a large dump take longer to free() node by node than the OS takes to reclaim the pages
        std::cout.flush(), std::fflush(nullptr);
        if (!std::getenv("CCHOST_FULL_EXIT")) std::_Exit(0); // (CCHOST_FULL_EXIT=1: a normal exit, for gprof / sanitizers)
        return 0;
    } catch (const std::exception &e) {
        std::fprintf(stderr, "cluster-capacity: %s\n", e.what());
        return 1;
    }
}
