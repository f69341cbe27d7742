// cluster_capacity.hpp -- the reference's simulator object (pkg/framework/simulator.go) on the MI355X engine, same method
// names, same order of use (cmd/cluster-capacity/app/server.go:163-183 runSimulator):
//
//     cc := framework.New(kubeSchedulerConfig, restConfig, simulatedPod, maxPods, excludeNodes)   simulator.go:107-158
//     cc.SyncWithClient(client)    copy Nodes and non-terminal Pods into the simulator              :176-295
//     cc.Run()                     schedule clones of the pod until Unschedulable / the limit       :356-381
//     cc.Report()                  ClusterCapacityReview                                             :160-170, report.go:196-225
//     cc.Close()                                                                                     :314-325
//
// Differences, on purpose: there is no API server, so SyncWithClient takes the listed objects themselves; the scheduling
// loop is ccsim_run on the GPU (exact) instead of one fake-clientset round trip per pod; Status keeps the
// per-node counts and the placement order instead of one *v1.Pod per clone.
#pragma once
#include <string>
#include <vector>

#include "engine.hpp"

namespace cchost {

class ClusterCapacity {
  public:
    struct Status { // simulator.go:90-93
        RunResult Pods;         // which node every simulated pod landed on (counts + order)
        std::string StopReason; // "LimitReached: ..." / "Unschedulable: <FitError>"
    };
    int device = 0;
    int gpus = 1;               // > 1 (or force_sharded): node-range shards over the GPUs of this box, RCCL exchange per pass
    bool force_sharded = false; // the sharded path with ONE rank (plumbing test on a one-GPU box)
    std::string mode; // "" = batched unless the pod couples nodes or the search is sampled

    static ClusterCapacity New(const HostProfile &kubeSchedulerConfig, Value simulatedPod, int64_t maxPods, std::vector<std::string> excludeNodes) {
        return New(kubeSchedulerConfig, std::vector<Value>{std::move(simulatedPod)}, maxPods, std::move(excludeNodes));
    }
    // several templates: scheduled pod i is a clone of template i mod P (what the reference's report layer assumes, report.go:146-171)
    static ClusterCapacity New(const HostProfile &kubeSchedulerConfig, std::vector<Value> simulatedPods, int64_t maxPods, std::vector<std::string> excludeNodes) {
        ClusterCapacity c;
        c.profile_ = kubeSchedulerConfig, c.pods_ = std::move(simulatedPods), c.max_simulated_ = maxPods, c.exclude_ = std::move(excludeNodes);
        return c;
    }
    // spreading_objs: the Services and controllers of the dump (PodTopologySpread's system default constraints)
    // volume_objs: the PersistentVolumeClaims and StorageClasses of the dump (simulator.go:228-295; volumes only with sync_volumes)
    void SyncWithClient(const std::vector<Value> &nodes, const std::vector<Value> &pods, const std::vector<Value> &namespaces = {},
                        const std::vector<Value> &spreading_objs = {}, VolumeObjects volume_objs = {}) {
        volume_objs.plugins = profile_.volume_plugins, volume_objs.plugins_partial = profile_.volume_plugins_partial;
        volume_objs.dra_enabled = profile_.dra_enabled, volume_objs.dra_partial = profile_.dra_partial;
        snap_ = build_snapshot(nodes, pods, pods_, exclude_, profile_.hard_pod_affinity_weight, namespaces, spreading_objs,
                               profile_.c.w_topologyspread != 0 && profile_.system_default_spreading, &volume_objs);
        synced_ = true;
    }
    void Run() {
        if (!synced_) throw std::runtime_error("ClusterCapacity.Run before SyncWithClient");
        if (gpus > 1 || force_sharded) SetResult(simulate_sharded(snap_, max_simulated_, mode, profile_, gpus));
        else SetResult(simulate(snap_, max_simulated_, mode, profile_, device));
    }
    void SetResult(RunResult r) { // (test hook: a result that did not come from the engine)
        status_.Pods = std::move(r);
        status_.StopReason = terminal_stop_reason(snap_, status_.Pods, max_simulated_, profile_.c.filter_mask, /*warn=*/true);
        ran_ = true;
    }
    Value Report() const {
        if (!ran_) throw std::runtime_error("ClusterCapacity.Report before Run");
        return build_review(pods_, snap_, status_.Pods, max_simulated_, profile_.c.filter_mask);
    }
    const Status &GetStatus() const { return status_; }
    const Snapshot &snapshot() const { return snap_; }
    void Close() { synced_ = ran_ = false; }

  private:
    HostProfile profile_;
    std::vector<Value> pods_;
    int64_t max_simulated_ = 0;
    std::vector<std::string> exclude_;
    Snapshot snap_;
    Status status_;
    bool synced_ = false, ran_ = false;
};

} // namespace cchost
