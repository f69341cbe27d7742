#!/usr/bin/env python
"""The step before the path: how long does the C++ host take to turn a kubectl-style JSON dump into the integer snapshot?
    python tools/bench_ingest.py [nodes] [pods] [--realistic]     (no GPU involved: --dump-snapshot stops before the engine)
--realistic: objects as `kubectl get -o json` really emits them (managedFields, conditions, nodeInfo, 20 images per node,
container statuses, env, mounts, volumes, default tolerations, last-applied-configuration): ~4x the bytes, none of them read."""
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge

ge.load_package()
from cluster_capacity_amd import build as B

REAL = "--realistic" in sys.argv
argv = [a for a in sys.argv[1:] if not a.startswith("--")]
N = int(argv[0]) if len(argv) > 0 else 100_000
P = int(argv[1]) if len(argv) > 1 else 300_000
native = B.build_host()
with tempfile.TemporaryDirectory() as d:
    nodes = [{"kind": "Node", "apiVersion": "v1",
              "metadata": {"name": f"node-{i:07d}", "labels": {"kubernetes.io/hostname": f"node-{i:07d}", "topology.kubernetes.io/zone": f"zone-{i % 16:03d}",
                                                              "topology.kubernetes.io/region": "region-0", "node.kubernetes.io/instance-type": ["m5", "c5", "r5", "m6", "c6", "r6"][i % 6],
                                                              "beta.kubernetes.io/arch": "amd64", "kubernetes.io/os": "linux"},
                           "annotations": {"node.alpha.kubernetes.io/ttl": "0", "volumes.kubernetes.io/controller-managed-attach-detach": "true"}},
              "spec": {"taints": ([{"key": "dedicated", "value": "infra", "effect": "NoSchedule"}] if i % 20 == 0 else [])},
              "status": {"allocatable": {"cpu": "15890m", "memory": "64Gi", "pods": "110", "ephemeral-storage": "100Gi"},
                         "capacity": {"cpu": "16", "memory": "65Gi", "pods": "110"},
                         "images": [{"names": [f"registry.k8s.io/pause:3.{j}"], "sizeBytes": 300000 + j} for j in range(3)],
                         "conditions": [{"type": "Ready", "status": "True", "reason": "KubeletReady", "message": "kubelet is posting ready status"}]}} for i in range(N)]
    pods = [{"kind": "Pod", "metadata": {"name": f"p{i}", "namespace": "default", "labels": {"app": "x"}},
             "spec": {"nodeName": f"node-{(i * 7) % N:07d}", "containers": [{"name": "c", "resources": {"requests": {"cpu": "250m", "memory": "512Mi"}}}]},
             "status": {"phase": "Running"}} for i in range(P)]
    if REAL:
        mf = [{"manager": m, "operation": "Update", "apiVersion": "v1", "time": "2025-01-01T00:00:00Z", "fieldsType": "FieldsV1",
               "fieldsV1": {"f:metadata": {"f:labels": {".": {}, **{f"f:label-{k}": {} for k in range(8)}}, "f:annotations": {".": {}, "f:a": {}}},
                            "f:status": {"f:conditions": {f'k:{{"type":"{t}"}}': {".": {}, "f:lastHeartbeatTime": {}, "f:status": {}, "f:reason": {}} for t in
                                                          ("Ready", "MemoryPressure", "DiskPressure", "PIDPressure")}}}} for m in ("kubelet", "kube-controller-manager")]
        for i, n in enumerate(nodes):
            n["metadata"].update(managedFields=mf, uid=f"{i:08x}-0000-0000-0000-000000000000", resourceVersion=str(10_000_000 + i), creationTimestamp="2025-01-01T00:00:00Z")
            n["status"].update(
                conditions=[{"type": t, "status": "False", "lastHeartbeatTime": "2025-01-01T00:00:00Z", "lastTransitionTime": "2025-01-01T00:00:00Z",
                             "reason": "KubeletHas" + t, "message": "kubelet has sufficient " + t.lower()} for t in ("MemoryPressure", "DiskPressure", "PIDPressure", "Ready")],
                addresses=[{"type": "InternalIP", "address": f"10.{i >> 16}.{(i >> 8) & 255}.{i & 255}"}, {"type": "Hostname", "address": n["metadata"]["name"]}],
                nodeInfo={"machineID": "0" * 32, "systemUUID": "0" * 36, "bootID": "0" * 36, "kernelVersion": "6.8.0", "osImage": "Ubuntu 24.04", "containerRuntimeVersion": "containerd://2.0",
                          "kubeletVersion": "v1.34.0", "kubeProxyVersion": "v1.34.0", "operatingSystem": "linux", "architecture": "amd64"},
                daemonEndpoints={"kubeletEndpoint": {"Port": 10250}},
                images=[{"names": [f"registry.example.com/team/app-{j}@sha256:{j:064x}", f"registry.example.com/team/app-{j}:v1.{j}"], "sizeBytes": 50_000_000 + j} for j in range(20)])
        for i, p in enumerate(pods):
            p["metadata"].update(managedFields=mf[:1], uid=f"{i:08x}-1111-0000-0000-000000000000", resourceVersion=str(20_000_000 + i), creationTimestamp="2025-01-01T00:00:00Z",
                                 ownerReferences=[{"apiVersion": "apps/v1", "kind": "ReplicaSet", "name": "rs", "uid": "x", "controller": True, "blockOwnerDeletion": True}],
                                 annotations={"kubectl.kubernetes.io/last-applied-configuration": json.dumps(p)})
            c = p["spec"]["containers"][0]
            c.update(image="registry.example.com/team/app-1:v1.1", env=[{"name": f"VAR_{k}", "value": "x" * 24} for k in range(6)], imagePullPolicy="IfNotPresent",
                     volumeMounts=[{"name": "kube-api-access-abcde", "readOnly": True, "mountPath": "/var/run/secrets/kubernetes.io/serviceaccount"}],
                     livenessProbe={"httpGet": {"path": "/healthz", "port": 8080, "scheme": "HTTP"}, "timeoutSeconds": 1, "periodSeconds": 10},
                     terminationMessagePath="/dev/termination-log", terminationMessagePolicy="File")
            p["spec"].update(volumes=[{"name": "kube-api-access-abcde", "projected": {"sources": [{"serviceAccountToken": {"expirationSeconds": 3607, "path": "token"}},
                                                                                                   {"configMap": {"name": "kube-root-ca.crt", "items": [{"key": "ca.crt", "path": "ca.crt"}]}}]}}],
                             tolerations=[{"key": f"node.kubernetes.io/{k}", "operator": "Exists", "effect": "NoExecute", "tolerationSeconds": 300} for k in ("not-ready", "unreachable")],
                             restartPolicy="Always", dnsPolicy="ClusterFirst", serviceAccountName="default", schedulerName="default-scheduler", priority=0, enableServiceLinks=True)
            p["status"].update(conditions=[{"type": t, "status": "True", "lastProbeTime": None, "lastTransitionTime": "2025-01-01T00:00:00Z"} for t in ("Initialized", "Ready", "ContainersReady", "PodScheduled")],
                               hostIP="10.0.0.1", podIP="10.1.0.1", podIPs=[{"ip": "10.1.0.1"}], startTime="2025-01-01T00:00:00Z", qosClass="Burstable",
                               containerStatuses=[{"name": "c", "state": {"running": {"startedAt": "2025-01-01T00:00:00Z"}}, "lastState": {}, "ready": True, "restartCount": 0,
                                                   "image": "registry.example.com/team/app-1:v1.1", "imageID": "registry.example.com/team/app-1@sha256:" + "0" * 64,
                                                   "containerID": "containerd://" + "0" * 64, "started": True}])
    cluster, podspec, dump = os.path.join(d, "cluster.json"), os.path.join(d, "pod.json"), os.path.join(d, "snapshot.json")
    json.dump({"kind": "List", "items": nodes + pods}, open(cluster, "w"))
    json.dump({"apiVersion": "v1", "kind": "Pod", "metadata": {"name": "small-pod", "labels": {"app": "guestbook"}},
               "spec": {"containers": [{"name": "c", "image": "registry.k8s.io/pause:3.1", "resources": {"requests": {"cpu": "150m", "memory": "100Mi"}}}]}}, open(podspec, "w"))
    if os.environ.get("CCBENCH_KEEP"):  # keep the generated dump (profiling the host outside this script)
        import shutil
        shutil.copy(cluster, os.environ["CCBENCH_KEEP"]), shutil.copy(podspec, os.environ["CCBENCH_KEEP"] + ".pod.json")
    del nodes, pods  # (several GB of Python objects at --realistic sizes: do not make the host compete for memory)
    size = os.path.getsize(cluster) / 1e6
    best = None
    for _ in range(3):
        t0 = time.perf_counter()
        r = subprocess.run([native, "--podspec", podspec, "--snapshot", cluster, "--dump-snapshot", dump], check=True, capture_output=True, text=True,
                           env=dict(os.environ, CCHOST_TIMING="1"))
        dt = time.perf_counter() - t0
        if best is None or dt < best:
            best, phases = dt, r.stderr
    ms = [float(l.split()[-2]) for l in phases.splitlines() if l.startswith("[timing]")]
    before_engine = (ms[0] + ms[1]) / 1e3  # what the CLI does before ccsim_load_nodes; the dump is this script's way of stopping there
    print(f"{N} Nodes + {P} Pods{' (kubectl-realistic objects)' if REAL else ''}, {size:.0f} MB of JSON: read + parse + intern + integer snapshot in "
          f"{before_engine:.2f} s = {size / before_engine:.0f} MB/s (process incl. the snapshot dump: {best:.2f} s; {os.cpu_count()} host cores: the List's items, the per-pod and the per-node walks run on up to "
          f"that many threads)")
    print(phases, end="")
