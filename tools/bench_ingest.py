#!/usr/bin/env python
"""The step before the path: how long does the C++ host take to turn a kubectl-style JSON dump into the integer snapshot?
    python tools/bench_ingest.py [nodes] [pods]        (no GPU involved: --dump-snapshot stops before the engine)"""
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

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
P = int(sys.argv[2]) if len(sys.argv) > 2 else 300_000
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
    cluster, podspec, dump = os.path.join(d, "cluster.json"), os.path.join(d, "pod.json"), os.path.join(d, "snapshot.json")
    json.dump({"kind": "List", "items": nodes + pods}, open(cluster, "w"))
    json.dump({"apiVersion": "v1", "kind": "Pod", "metadata": {"name": "small-pod", "labels": {"app": "guestbook"}},
               "spec": {"containers": [{"name": "c", "image": "registry.k8s.io/pause:3.1", "resources": {"requests": {"cpu": "150m", "memory": "100Mi"}}}]}}, open(podspec, "w"))
    size = os.path.getsize(cluster) / 1e6
    best = None
    for _ in range(3):
        t0 = time.perf_counter()
        subprocess.run([native, "--podspec", podspec, "--snapshot", cluster, "--dump-snapshot", dump], check=True)
        dt = time.perf_counter() - t0
        best = dt if best is None or dt < best else best
    print(f"{N} Nodes + {P} Pods, {size:.0f} MB of JSON: parse + intern + integer snapshot (+ its dump) in {best:.2f} s = {size / best:.0f} MB/s "
          f"({os.cpu_count()} host cores, one thread)")
