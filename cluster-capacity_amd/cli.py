"""cluster-capacity CLI restated on top of the MI355X engine.

    ./cluster-capacity --podspec pod.yaml --snapshot cluster.yaml [--snapshot more.json]
           [--max-limit N] [--exclude-nodes a,b] [--verbose] [-o json|yaml]

Flags mirror cmd/cluster-capacity/app/options/options.go:65-77.  There is no API server to talk to here, so
`--kubeconfig` is replaced by `--snapshot`: files holding the Node and Pod objects `SyncWithClient` would list
(pkg/framework/simulator.go:176-295) -- `kubectl get nodes,pods -A -o yaml` output, a `List`, or multi-document YAML.
`--default-config` takes a KubeSchedulerConfiguration (schedconfig.py: what the engine can express of it).

Output formats mirror pkg/framework/report.go:235-317 (pretty / json / yaml).  The simulation itself runs on the
GPU through the C ABI (capi.Engine); there is no CPU fallback.
"""
from __future__ import annotations

import argparse
import copy
import datetime
import json
import math
from fractions import Fraction
import sys
from typing import List, Optional

import numpy as np
import yaml

from . import ingest, model as M, preemption, report as R, schedconfig


def load_by_kind(paths: List[str]) -> dict:
    """Every object of the snapshot files, by kind (lists are flattened; one pass over the files)."""
    out: dict = {}
    for path in paths:
        with open(path) as f:
            docs = list(yaml.safe_load_all(f))  # YAML is a superset of JSON
        for d in docs:
            if not d:
                continue
            items = d["items"] if str(d.get("kind", "")).endswith("List") and "items" in d else [d]
            for o in items:
                out.setdefault(o.get("kind"), []).append(o)
    return out


def load_all(paths: List[str]):
    """-> (nodes, pods, namespaces): what SyncWithClient lists (simulator.go:176-295)."""
    by = load_by_kind(paths)
    return by.get("Node", []), by.get("Pod", []), by.get("Namespace", [])


def load_objects(paths: List[str]):
    return load_all(paths)[:2]


def load_kind(paths: List[str], kind: str):
    """Every object of one kind in the snapshot files (lists are flattened)."""
    out = []
    for path in paths:
        with open(path) as f:
            docs = list(yaml.safe_load_all(f))
        for d in docs:
            if not d:
                continue
            items = d["items"] if str(d.get("kind", "")).endswith("List") and "items" in d else [d]
            out += [o for o in items if o.get("kind") == kind]
    return out


def parse_pod_spec(path: str, scheduler_name: str = "default-scheduler") -> dict:
    """options.go:79-147 ParseAPISpec (defaults only; API validation is the apiserver's job)."""
    with open(path) as f:
        pod = yaml.safe_load(f)
    if not pod or pod.get("kind") != "Pod":
        raise SystemExit("Failed to decode config file: not a Pod")
    md = pod.setdefault("metadata", {})
    md.setdefault("namespace", "default")
    spec = pod.setdefault("spec", {})
    spec.setdefault("schedulerName", scheduler_name)
    spec.setdefault("dnsPolicy", "ClusterFirst")
    spec.setdefault("restartPolicy", "Always")
    for c in spec.get("containers") or []:
        c.setdefault("terminationMessagePolicy", "FallbackToLogsOnError")
    if not spec.get("containers"):
        raise SystemExit("Invalid pod: spec.containers: Required value")
    return pod


def pod_requirements(pod: dict) -> dict:
    """report.go:111-144 getResourceRequest (containers only) + :182-194.  cpu and memory are sums of Quantities printed by
    Quantity.String(): the container's quantity is the receiver of Add (`rQuantity.Add(sum so far)`), so the sum carries the Format
    of the LAST container that names the resource with a non-zero amount (quantity.go:600-613), starting from DecimalSI (cpu) /
    BinarySI (memory): `memory: 512M` prints as 512M, `512Mi` as 512Mi, 1Gi + 512Mi as 1536Mi."""
    total = {"cpu": (Fraction(0), "DecimalSI"), "memory": (Fraction(0), "BinarySI")}
    scalars = {}
    for c in pod["spec"].get("containers") or []:
        for name, q in ((c.get("resources") or {}).get("requests") or {}).items():
            if name in total:
                v = Fraction(math.ceil(ingest.parse_quantity(q) * 10**9), 10**9)
                acc_v, acc_fmt = total[name]
                total[name] = (v + acc_v, ingest.quantity_format(q) if v != 0 else acc_fmt)
            elif ingest.is_scalar_resource(name):
                scalars[name] = scalars.get(name, 0) + ingest.value(q)
    return {"podName": pod["metadata"].get("name", ""),
            "resources": {"primaryResources": {"cpu": ingest.quantity_canonical(*total["cpu"]), "memory": ingest.quantity_canonical(*total["memory"]),
                                               "nvdia.com/gpu": "0"},  # sic: ResourceNvidiaGPU = "nvdia.com/gpu" (report.go:34)
                          "scalarResources": scalars or None},
            "nodeSelectors": pod["spec"].get("nodeSelector")}


def build_review(pod, snap: ingest.Snapshot, result: M.RunResult, max_limit: int, filter_mask: int = M.F_ALL) -> dict:
    """report.go:196-225 GetReport.  `pod`: the template or the list of templates; scheduled pod i is a clone of template
    i mod P (parsePodsReview, report.go:146-171), which is exactly the order the engine cycles them in."""
    pods = list(pod) if isinstance(pod, (list, tuple)) else [pod]
    P = len(pods)
    failing = result.stop_spec if P > 1 and result.stop_spec >= 0 else 0  # the FitError describes the template that did not fit
    outcome = None
    if result.stop == M.STOP_UNSCHEDULABLE:  # the PostFilter of the terminal cycle: DefaultPreemption's dry run (message tail only)
        mixed = len({(p.preempt.priority if p.preempt else 0) for p in snap.pods}) > 1  # clones of one template below another's priority
        pod_f = snap.pods[failing]
        if pod_f.rwop_capacity_one and result.placed >= 1 and not getattr(result, "prefilter_msg", None):
            # the terminal cycle saw the pod with its ReadWriteOncePod claim held by its own clone (with_rwop_in_use): a clone is no victim,
            # so the claim stays in conflict on every node whatever the dry run removes (static disk conflicts may leave with a victim)
            pod_f = with_rwop_in_use(pod_f, snap.nodes.n, result.per_node_count if P == 1 else None)
            pre = copy.copy(pod_f.preempt or M.PreemptionSide())
            rest = np.full(snap.nodes.n, M.VOL_RWOP, np.uint8)
            if pre.volume_veto_rest is not None:
                rest[np.asarray(pre.volume_veto_rest) == M.VOL_DISK_CONFLICT] = M.VOL_DISK_CONFLICT
            pre.volume_veto_rest = rest
            pod_f.preempt = pre
        outcome = preemption.dry_run(snap.nodes, pod_f, result.per_node_count, result.n_code_unschedulable, filter_mask, P, mixed)
        if outcome.kind == "unmodelled":
            print("warning: a lower-priority pod takes part in a topology-coupled filter of the simulated pod (or several templates "
                  "run): the preemption dry run is not modelled, the 'preemption:' part of the message assumes no victims",
                  file=sys.stderr)
    stop = R.stop_reason(result, len(snap.names), max_limit, taint_reasons=snap.taint_reasons_all[failing], scalar_names=snap.scalar_names,
                         preemption=outcome)
    if P == 1:
        per_template = [R.replicas_on_nodes(result.per_node_count, snap.names, result.log)]
    else:
        if result.log is None or len(result.log) < result.placed:
            raise SystemExit("several templates: the placement log does not cover the run (raise the log capacity)")
        per_template = []
        for t in range(P):
            log_t = np.asarray(result.log[t::P])
            cnt = np.bincount(log_t, minlength=len(snap.names)).astype(np.int32) if len(log_t) else np.zeros(len(snap.names), np.int32)
            per_template.append(R.replicas_on_nodes(cnt, snap.names, log_t))
    return {
        "spec": {"templates": pods, "replicas": 0, "podRequirements": [pod_requirements(p) for p in pods]},
        "status": {
            # time.Time marshals as RFC 3339 with "Z" for UTC (report.go:56, getReviewStatus :214-221)
            "creationTimestamp": datetime.datetime.now(datetime.timezone.utc).isoformat().replace("+00:00", "Z"),
            "replicas": int(result.placed),
            "failReason": R.main_fail_reason(stop),
            "pods": [{"podName": p["metadata"].get("name", ""), "replicasOnNodes": per_template[t], "failSummary": None}
                     for t, p in enumerate(pods)],
        },
    }


def pretty(review: dict, verbose: bool) -> str:
    """report.go:235-283 clusterCapacityReviewPrettyPrint."""
    out = []
    if verbose:
        for req in review["spec"]["podRequirements"]:
            out.append(f"{req['podName']} pod requirements:")
            out.append(f"\t- CPU: {req['resources']['primaryResources']['cpu']}")
            out.append(f"\t- Memory: {req['resources']['primaryResources']['memory']}")
            if req["resources"]["scalarResources"] is not None:
                # fmt's %v of a map[v1.ResourceName]int64: map[key:value key:value], keys sorted (report.go:244-246)
                out.append("\t- ScalarResources: map[" + " ".join(f"{k}:{v}" for k, v in sorted(req["resources"]["scalarResources"].items())) + "]")
            if req["nodeSelectors"] is not None:
                out.append("\t- NodeSelector: " + ",".join(f"{k}={v}" for k, v in sorted(req["nodeSelectors"].items())))
            out.append("")
    for p in review["status"]["pods"]:
        total = sum(r["replicas"] for r in p["replicasOnNodes"])
        out.append(f"The cluster can schedule {total} instance(s) of the pod {p['podName']}." if verbose else f"{total}")
    if verbose:
        fr = review["status"]["failReason"]
        out.append(f"\nTermination reason: {fr['failType']}: {fr['failMessage']}")
        if review["status"]["replicas"] > 0:
            out.append("\nPod distribution among nodes:")
            for p in review["status"]["pods"]:
                out.append(p["podName"])
                for r in p["replicasOnNodes"]:
                    out.append(f"\t- {r['nodeName']}: {r['replicas']} instance(s)")
    return "\n".join(out) + "\n"


def simulate(snap: ingest.Snapshot, max_limit: int, mode: Optional[str] = None, device: int = 0,
             percentage_of_nodes_to_score: int = 100, profile: Optional[M.Profile] = None) -> M.RunResult:
    from . import capi

    coupled = bool(snap.pod.spread) or snap.pod.ipa is not None
    # a sampled search (schedule_one.go:610-723) is order-dependent: the literal one-cycle-per-pass loop
    sampled = percentage_of_nodes_to_score != 100 and snap.nodes.n >= 100
    mode = mode or ("sequential" if coupled or sampled else "batched")
    prof = profile or M.Profile.default()
    prof.percentage_of_nodes_to_score = percentage_of_nodes_to_score
    if len(snap.pods) == 1 and snap.pod.prefilter_reject:  # a volume plugin's PreFilter rejects the pod everywhere: the first cycle ends the run
        return rejected_by_prefilter(snap.nodes, snap.pod, snap.pod.prefilter_reject)
    eng = capi.Engine(device=device)
    # several templates: ccsim_set_pods, cycled round-robin by ccsim_run (windows of pods x nodes; `mode` does not apply)
    cap = max_limit if max_limit > 0 else int(min(int(snap.nodes.alloc_pods.astype(np.int64).sum()), 1 << 26))
    try:
        try:
            if len(snap.pods) > 1 and any(q.prefilter_reject or q.rwop_capacity_one for q in snap.pods):
                # (what these templates need is host-side: nothing ccsim_set_pods could refuse)
                return simulate_specs_one_cycle_at_a_time(snap.nodes, snap.pods, prof, max_limit, eng)
            eng.load(snap.nodes, snap.pods if len(snap.pods) > 1 else snap.pod, prof)
        except capi.CcsimError as ex:
            if len(snap.pods) > 1 and ex.rc == -38:  # -ENOSYS: a set of pod specs the window engine does not take
                print("cluster-capacity: note: these templates are placed one scheduling cycle at a time (the windows of several pod specs do not "
                      f"take them: {str(ex).split(': ', 1)[-1]})", file=sys.stderr)
                return simulate_specs_one_cycle_at_a_time(snap.nodes, snap.pods, prof, max_limit, eng)
            raise
        if len(snap.pods) == 1 and snap.pod.rwop_capacity_one and max_limit != 1:
            # A ReadWriteOncePod claim nobody uses yet: the first clone takes it (volume_restrictions.go:249-264, 283-291), the second
            # cycle then fails on every node that gets as far as VolumeRestrictions: the pod set again on the SAME state with that verdict
            first = eng.run(max_limit=1, mode=mode, want_log=True, log_cap=1)
            if first.placed == 0:
                return first
            eng.set_pod(with_rwop_in_use(snap.pod, snap.nodes.n, first.per_node_count))
            last = eng.run(max_limit=0, mode=mode, want_log=True, log_cap=1)
            assert last.placed == 0 and last.stop == M.STOP_UNSCHEDULABLE
            last.placed, last.per_node_count, last.log, last.rounds = 1, first.per_node_count, first.log, first.rounds + last.rounds
            return last
        return eng.run(max_limit=max_limit, mode=mode, want_log=True, log_cap=max(1, cap))
    finally:
        eng.close()


def rejected_by_prefilter(nodes: M.NodesSoA, pod: M.PodSpec, msg: str, placed_before=None) -> M.RunResult:
    """The cycle a PreFilter plugin rejects (schedule_one.go:495-508): no node is evaluated, every node carries the plugin's status."""
    n = nodes.n
    if n == 0:  # schedulePod returns ErrNoNodesAvailable BEFORE any PreFilter plugin runs (schedule_one.go:438-440)
        return M.RunResult(placed=0, stop=M.STOP_NO_NODES, per_node_count=np.zeros(0, np.int32), log=np.zeros(0, np.int32), hist=np.zeros(M.NREASON, np.int64),
                           hist_taintset=np.zeros(max(1, len(pod.taint_filter_ok)), np.int64), n_code_unschedulable=0, rounds=1)
    return M.RunResult(placed=0, stop=M.STOP_UNSCHEDULABLE, per_node_count=np.zeros(n, np.int32), log=np.zeros(0, np.int32),
                       hist=np.zeros(M.NREASON, np.int64), hist_taintset=np.zeros(max(1, len(pod.taint_filter_ok)), np.int64),
                       n_code_unschedulable=0, rounds=1, prefilter_msg=msg)


def with_rwop_in_use(pod: M.PodSpec, n: int, clones=None) -> M.PodSpec:
    """The pod once a clone holds its ReadWriteOncePod claim: VolumeRestrictions fails every node -- after its own disk check (`clones`:
    where the pod's earlier clones sit, for a pod whose disks are exclusive: the engine counts clones from the moment a pod is set)."""
    q = copy.copy(pod)
    veto = np.zeros(n, np.uint8) if pod.volume_veto is None else np.asarray(pod.volume_veto, np.uint8).copy()
    if pod.volume_exclusive and clones is not None:
        veto[np.asarray(clones) > 0] = M.VOL_DISK_CONFLICT
    veto[veto != M.VOL_DISK_CONFLICT] = M.VOL_RWOP
    q.volume_veto, q.rwop_capacity_one = veto, False
    return q


def pod_with_clones(nodes: M.NodesSoA, pod: M.PodSpec, clones: np.ndarray) -> M.PodSpec:
    """The pod spec as the cycle after `clones[n]` of its own clones landed on node n must see it: every clone is an existing pod of the
    next cycle (types.go:345-350 AddPod), so it joins the per-node counts the spec's topology-coupled plugins were given for the
    snapshot's pods -- exactly what the oracle's workspace does with its `placed` vector (oracle/ccref.c node_match_count, ipa_build) and
    what the engine's own tables do within a run of ONE spec."""
    import copy

    if not clones.any():
        return pod
    p = copy.copy(pod)
    c32 = clones.astype(np.int32)
    p.spread = [copy.copy(k) for k in pod.spread]
    for k in p.spread:
        if k.self_match:
            k.node_match_count = c32.copy() if k.node_match_count is None else (np.asarray(k.node_match_count, np.int32) + c32)
    if pod.ipa is not None:
        a = copy.copy(pod.ipa)
        add = lambda arr, inc, dt: inc.astype(dt) if arr is None else np.asarray(arr, dt) + inc.astype(dt)  # noqa: E731
        if a.self_aff:
            a.aff_existing = add(a.aff_existing, c32, np.int32)
        a.anti_existing = [add(arr, c32, np.int32) if a.anti_self[t] else arr for t, arr in enumerate(a.anti_existing)]
        n_keys = len(a.key_cols)
        exist = list(a.exist_anti) + [None] * (n_keys - len(a.exist_anti))
        score = list(a.score_existing) + [None] * (n_keys - len(a.score_existing))
        entries = int(a.entries_existing)
        for k in range(n_keys):
            terms = sum(1 for t, kk in enumerate(a.anti_keys) if kk == k and a.anti_self[t])
            if terms:
                exist[k] = add(exist[k], c32 * terms, np.int32)
            w = int(a.score_self[k]) if k < len(a.score_self) else 0
            if w:
                score[k] = add(score[k], clones.astype(np.int64) * w, np.int64)
            se = int(a.self_entries[k]) if k < len(a.self_entries) else 0
            if se:
                entries += int(clones[np.asarray(nodes.label_cols[a.key_cols[k]]) != 0].sum()) * se
        a.exist_anti, a.score_existing, a.entries_existing = exist, score, entries
        p.ipa = a
    if pod.volume_exclusive:  # (... and of a pod whose disks conflict with a clone's: volume_restrictions.go:105-150, the first volume check)
        veto = np.zeros(len(clones), np.uint8) if pod.volume_veto is None else np.asarray(pod.volume_veto, np.uint8).copy()
        veto[clones > 0] = M.VOL_DISK_CONFLICT
        p.volume_veto = veto
    if pod.has_host_ports:  # (a node holds at most one clone of a pod with host ports: node_ports.go:164-176)
        conflict = (clones > 0).astype(np.uint8)
        p.host_ports_conflict = conflict if pod.host_ports_conflict is None else (np.asarray(pod.host_ports_conflict, np.uint8) | conflict)
    return p


def simulate_specs_one_cycle_at_a_time(nodes: M.NodesSoA, pods, prof: M.Profile, max_limit: int, eng=None, device: int = 0) -> M.RunResult:
    """Several pod specs, cycled round-robin, WITHOUT the window engine (csrc/ccsim_multi.h takes BASELINE config 5's shape; scalar
    resources, ScheduleAnyway constraints, more than two topology keys, values outside the narrow mirrors are -ENOSYS there): the literal
    loop of the reference -- cycle i schedules a clone of spec i mod P (pkg/framework/simulator.go:297-381) -- with one ccsim_set_pod
    + one scheduling cycle of the HIP engine per placement.  The node columns carry every earlier clone; what a spec's own earlier
    clones add to ITS plugin state is folded into the per-node counts it is set with (pod_with_clones; the templates' selectors are
    disjoint -- ingest._check_templates_disjoint -- so other specs' clones add nothing).  A slow path (an O(N) upload per cycle), exact:
    tests/test_multi.py holds it against the oracle's round-robin loop.  Host ports: a clone excludes clones of OTHER specs with the same
    port from its node as well, which nothing here tracks -- refused unless at most one template asks for host ports."""
    from . import capi

    P, N = len(pods), nodes.n
    if sum(1 for q in pods if q.has_host_ports) > 1:
        raise NotImplementedError("several templates with host ports")
    own = eng is None
    if own:
        eng = capi.Engine(device=device)
    prof = copy.copy(prof)  # (a shallow copy keeps the host-side notes schedconfig hangs on the object)
    prof.percentage_of_nodes_to_score = 100  # as the window engine: every node is scored
    eng.load(nodes, pods[0], prof)
    clones = np.zeros((P, N), np.int64)
    log, per_node, per_spec = [], np.zeros(N, np.int32), np.zeros(P, np.int32)
    last, stop, stop_spec = None, M.STOP_LIMIT, -1
    pods = list(pods)
    prefilter_msg = None
    try:
        while True:
            t = len(log) % P
            if pods[t].prefilter_reject:  # a volume plugin's PreFilter: this template's cycle ends the run
                last, prefilter_msg = rejected_by_prefilter(nodes, pods[t], pods[t].prefilter_reject), pods[t].prefilter_reject
                stop, stop_spec = last.stop, t  # (STOP_NO_NODES on an empty cluster: ErrNoNodesAvailable comes before any PreFilter)
                prefilter_msg = last.prefilter_msg
                break
            if pods[t].rwop_capacity_one and per_spec[t] == 1:  # its first clone holds the ReadWriteOncePod claim now
                pods[t] = with_rwop_in_use(pods[t], N)  # (pod_with_clones adds the clones' own disks below)
            eng.set_pod(pod_with_clones(nodes, pods[t], clones[t]))
            last = eng.run(max_limit=1, mode="sequential", want_log=True, log_cap=1)
            if last.placed == 0:
                stop, stop_spec = last.stop, t
                break
            w = int(last.log[0])
            log.append(w)
            clones[t, w] += 1
            per_node[w] += 1
            per_spec[t] += 1
            if max_limit > 0 and len(log) >= max_limit:
                break
    finally:
        if own:
            eng.close()
    return M.RunResult(placed=len(log), stop=stop, per_node_count=per_node, log=np.asarray(log, np.int32), hist=last.hist if stop_spec >= 0 else np.zeros(M.NREASON, np.int64),
                       hist_taintset=last.hist_taintset if stop_spec >= 0 else np.zeros_like(last.hist_taintset), n_code_unschedulable=last.n_code_unschedulable if stop_spec >= 0 else 0,
                       rounds=len(log) + (1 if stop_spec >= 0 else 0), per_spec_count=per_spec, stop_spec=stop_spec, prefilter_msg=prefilter_msg)


def ingest_volume_plugins():
    from . import volumes
    return volumes.PLUGINS


def hard_coupled(pod) -> bool:
    """A topology-coupled FILTER is active (host/snapshot.hpp PodSide::hard_coupled): a DoNotSchedule spread constraint, required
    inter-pod (anti-)affinity, or existing pods' anti-affinity terms matching the template -- the total then depends on the node sampling."""
    if any(k.hard for k in pod.spread):
        return True
    q = pod.ipa
    return q is not None and bool(q.aff_keys or q.anti_keys or any(a is not None and len(a) for a in q.exist_anti))


def main(argv: Optional[List[str]] = None, out=sys.stdout) -> int:
    ap = argparse.ArgumentParser(prog="cluster-capacity", description="Cluster-capacity is used for simulating scheduling of one or multiple pods")
    ap.add_argument("--podspec", action="append", default=[],
                    help="Path to JSON or YAML file containing pod definition.  Repeatable: the templates are cycled round-robin "
                         "(scheduled pod i is a clone of template i mod P, as the reference's report layer counts them)")
    ap.add_argument("--genpod", default="", metavar="NAMESPACE",
                    help="cmd/genpod: print the pod the namespace's LimitRanges / node-selector annotation describe (objects from --snapshot)")
    ap.add_argument("--snapshot", action="append", required=True, help="File(s) with the cluster's Node and Pod objects (replaces --kubeconfig)")
    ap.add_argument("--max-limit", type=int, default=0, help="Number of instances of pod to be scheduled after which analysis stops. By default unlimited.")
    ap.add_argument("--exclude-nodes", default="", help="Exclude nodes to be scheduled")
    ap.add_argument("--verbose", action="store_true", help="Verbose mode")
    ap.add_argument("-o", "--output", default="", choices=["", "json", "yaml"], help="Output format. One of: json|yaml")
    ap.add_argument("--mode", default=None, choices=["batched", "sequential"], help="engine mode (default: batched unless the pod couples nodes)")
    ap.add_argument("--default-config", default="", help="Path to JSON or YAML file containing scheduler configuration.")
    ap.add_argument("--sync-persistent-volumes", action="store_true",
                    help="also take the PersistentVolume objects of the snapshot: bound claims are then judged as kube-scheduler judges them on the "
                         "live cluster (VolumeBinding's node affinity, VolumeZone's labels).  The reference's SyncWithClient does not copy volumes "
                         "(a bound claim ends its run with 'persistentvolume \"x\" not found'): that is the default here too")
    ap.add_argument("--percentage-of-nodes-to-score", type=int, default=None,
                    help="KubeSchedulerConfiguration.percentageOfNodesToScore: 100 scores every node (the final capacity and "
                         "distribution do not depend on it for pods without topology constraints); 0 = the scheduler's adaptive default")
    args = ap.parse_args(argv)
    if args.genpod:
        from . import genpod
        try:
            pod = genpod.namespace_pod(args.genpod, load_kind(args.snapshot, "Namespace"), load_kind(args.snapshot, "LimitRange"))
        except genpod.GenpodError as e:
            sys.stderr.write(f"Error: {e}\n")
            return 1
        out.write(json.dumps(pod) + "\n" if args.output == "json" else yaml.safe_dump(pod, sort_keys=True))  # the YAML serializer goes through JSON: keys sorted
        return 0
    if not args.podspec:
        ap.error("Pod spec file is missing")

    cfg = None
    if args.default_config:
        with open(args.default_config) as f:
            cfg = yaml.safe_load(f)
    prof, hard_weight = schedconfig.profile_from_config(cfg)
    try:
        pods = [parse_pod_spec(p) for p in args.podspec]
        pod = pods if len(pods) > 1 else pods[0]
        by = load_by_kind(args.snapshot)
        node_objs, pod_objs, ns_objs = by.get("Node", []), by.get("Pod", []), by.get("Namespace", [])
        owners = [o for k in ("ReplicationController", "ReplicaSet", "StatefulSet") for o in by.get(k, [])]
        services = by.get("Service", [])
        snap = ingest.build_snapshot(node_objs, pod_objs, pod, [x for x in args.exclude_nodes.split(",") if x], hard_pod_affinity_weight=hard_weight,
                                     namespace_objs=ns_objs, service_objs=services, owner_objs=owners,
                                     system_default_spreading=bool(prof.w_topologyspread) and getattr(prof, "system_default_spreading", True),
                                     # SyncWithClient copies claims and classes, not the volumes (simulator.go:228-295)
                                     pvc_objs=by.get("PersistentVolumeClaim", []), class_objs=by.get("StorageClass", []),
                                     pv_objs=by.get("PersistentVolume", []) if args.sync_persistent_volumes else None,
                                     csinode_objs=by.get("CSINode", []) if args.sync_persistent_volumes else (),
                                     attachment_objs=by.get("VolumeAttachment", []) if args.sync_persistent_volumes else (),
                                     volume_plugins=getattr(prof, "volume_plugins", ingest_volume_plugins()),
                                     volume_plugins_partial=getattr(prof, "volume_plugins_partial", False),
                                     dra_enabled=getattr(prof, "dra_enabled", True), dra_partial=getattr(prof, "dra_partial", False))
    except (TypeError, AttributeError, KeyError, OverflowError) as e:
        # the objects are walked as plain dicts / lists: a string where a mapping belongs (a decode error in the reference, which reads
        # into typed structs) or a sum beyond int64 surfaces as one of these -- refused, like the native host refuses it
        sys.stderr.write(f"cluster-capacity: malformed object: {type(e).__name__}: {e}\n")
        return 1
    if snap.default_spreading_unmodelled:
        print("warning: a Service (or its controller) selects the simulated pod and it has no topologySpreadConstraints of its own: the scheduler's system default "
              "spreading (hostname maxSkew 3, zone maxSkew 5, ScheduleAnyway) would score the nodes too; some node lacks the kubernetes.io/hostname label (or several templates run): "
              "not modelled -- the order of the placements (and so a --max-limit result) may differ, the total does not", file=sys.stderr)
    if args.percentage_of_nodes_to_score is not None:
        pct = args.percentage_of_nodes_to_score
    elif schedconfig.sets_percentage(cfg):
        pct = prof.percentage_of_nodes_to_score
    else:
        # left unset: the reference's default is 0 = adaptive sampling (defaults.go:106-129).  The final capacity and distribution
        # do not depend on it when nothing observes the ORDER of the placements (no --max-limit, no topology-coupled plugin): then
        # every node is scored (the fast batched mode); otherwise the reference's default applies
        # (several templates are always searched completely: the engine's windows need every node scored)
        # Round 5: a template with a topology-coupled FILTER keeps the reference's default too -- with one, the reported
        # total itself depends on which nodes a cycle saw (host/engine.hpp simulate(), PodSide::hard_coupled); the windowed fast form
        # (csrc/ccsim_coupled.h, needs every node scored) is the caller's choice: --percentage-of-nodes-to-score 100
        coupled = bool(snap.pod.spread) or snap.pod.ipa is not None
        pct = 100 if len(pods) > 1 else (0 if args.max_limit > 0 or hard_coupled(snap.pod) else 100)
        if len(pods) == 1 and coupled and snap.nodes.n >= 100 and pct == 0:
            print("cluster-capacity: note: a template with topology spread constraints / inter-pod affinity is placed with the reference's default "
                  "adaptive node sampling (percentageOfNodesToScore 0: one node pass per placement); --percentage-of-nodes-to-score 100 scores every "
                  "node per cycle (also a valid reference configuration) and runs ~50x faster on large clusters", file=sys.stderr)
    result = simulate(snap, args.max_limit, args.mode, percentage_of_nodes_to_score=pct, profile=prof)
    review = build_review(pod, snap, result, args.max_limit, prof.filter_mask)
    if args.output == "json":
        out.write(json.dumps(review) + "\n")
    elif args.output == "yaml":
        out.write(yaml.safe_dump(review, sort_keys=True))  # sigs.k8s.io/yaml marshals through JSON: keys sorted (report.go:296-303)
    else:
        out.write(pretty(review, args.verbose))
    return 0


if __name__ == "__main__":
    sys.exit(main())
