"""Deterministic synthetic cluster snapshots and the BASELINE.json benchmark configurations
(SURVEY.md 8(d)).  Data generation only -- not on the hot path.

Node i (name node-%07d) draws from splitmix64(seed ^ i); zone = i mod Z, so the canonical node
order of the reference (zone round-robin over lexicographic names,
vendor/k8s.io/kubernetes/pkg/scheduler/backend/cache/node_tree.go:119-143) is simply i = 0..N-1.
"""
from __future__ import annotations

import numpy as np

from . import model as M

GiB = 1 << 30
MiB = 1 << 20
SEED = 0xC0FFEE

INSTANCE_TYPES = ["m5.xlarge", "m5.2xlarge", "m5.4xlarge", "m5.8xlarge", "m5.16xlarge", "m5.24xlarge"]
CPU_CHOICES = np.array([4000, 8000, 16000, 32000, 64000, 96000], np.int64)
CPU_CDF = np.cumsum([20, 25, 25, 15, 10, 5]) / 100.0

# taint sets: 0 none, 1 dedicated=infra:NoSchedule, 2 maintenance=soon:PreferNoSchedule, 3 both
TAINTSETS = [
    [],
    [("dedicated", "infra", "NoSchedule")],
    [("maintenance", "soon", "PreferNoSchedule")],
    [("dedicated", "infra", "NoSchedule"), ("maintenance", "soon", "PreferNoSchedule")],
]


def splitmix64(x: np.ndarray) -> np.ndarray:
    x = (x + np.uint64(0x9E3779B97F4A7C15)).astype(np.uint64)
    z = x
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def _uniform(seed: int, idx: np.ndarray, k: int) -> np.ndarray:
    """k-th uniform [0,1) draw of node idx."""
    with np.errstate(over="ignore"):
        h = splitmix64(splitmix64(np.uint64(seed) ^ idx.astype(np.uint64)) + np.uint64(k))
    return (h >> np.uint64(11)).astype(np.float64) / float(1 << 53)


def zones_for(n: int) -> int:
    return 3 if n <= 10_000 else (16 if n <= 100_000 else 64)


def make_nodes(n: int, seed: int = SEED, offset: int = 0, with_names: bool = False, n_total: int = 0) -> M.NodesSoA:
    """Nodes offset..offset+n-1 of the synthetic cluster (a shard is just a different offset)."""
    with np.errstate(over="ignore"):
        i = np.arange(offset, offset + n, dtype=np.int64)
        u = [_uniform(seed, i, k) for k in range(9)]
    itype = np.searchsorted(CPU_CDF, u[0], side="right").clip(0, 5)
    a_cpu = CPU_CHOICES[itype]
    cores = a_cpu // 1000
    a_mem = cores * np.array([2, 4, 8], np.int64)[(u[1] * 3).astype(np.int64).clip(0, 2)] * GiB
    a_eph = np.full(n, 100 * GiB, np.int64)
    r_cpu = ((u[2] * 0.6 * a_cpu).astype(np.int64) // 50) * 50
    r_mem = ((u[3] * 0.6 * a_mem).astype(np.int64) // (64 * MiB)) * (64 * MiB)
    pods = (u[4] * 40).astype(np.int32)
    dedicated = u[5] < 0.05
    maint = u[6] < 0.02
    unsched = (u[7] < 0.005).astype(np.uint8)
    taintset = dedicated.astype(np.int32) + 2 * maint.astype(np.int32)
    z = np.zeros(n, np.int64)
    names = [f"node-{j:07d}" for j in i] if with_names else None
    nodes = M.NodesSoA(
        alloc=[a_cpu, a_mem, a_eph], alloc_pods=np.full(n, 110, np.int32),
        req=[r_cpu, r_mem, z], nz_mcpu=r_cpu.copy(), nz_mem=r_mem.copy(), pod_count=pods,
        taintset_id=taintset, unschedulable=unsched,
        # column 0 = node.kubernetes.io/instance-type, column 1 = topology.kubernetes.io/zone (zone = i mod Z)
        label_cols=[(itype + 1).astype(np.int32), ((i % zones_for(n_total or (offset + n))) + 1).astype(np.int32)],
        names=names,
    )
    return nodes


def zone_spread(n_nodes: int, max_skew: int = 1, min_domains: int = 1) -> M.SpreadConstraint:
    """whenUnsatisfiable: DoNotSchedule on topology.kubernetes.io/zone, selector matching the clones themselves."""
    return M.SpreadConstraint(col=1, max_skew=max_skew, min_domains=min_domains, hard=True, self_match=True,
                              n_domains=zones_for(n_nodes))


def examples_pod(tolerate_infra: bool = False, prefer_types: bool = False) -> M.PodSpec:
    """examples/pod.yaml of the reference: one container, cpu 150m, memory 100Mi (req == limit)."""
    ok = np.array([1, 1 if tolerate_infra else 0, 1, 1 if tolerate_infra else 0], np.uint8)
    cnt = np.array([0, 0, 1, 1], np.int32)
    preferred = []
    if prefer_types:
        def table(type_idx):
            t = np.zeros(len(INSTANCE_TYPES) + 1, np.uint8)
            t[type_idx + 1] = 1  # operator In [INSTANCE_TYPES[type_idx]]
            return t
        preferred = [(10, [(0, table(2))]), (40, [(0, table(4))])]
    return M.PodSpec(req=np.array([150, 100 * MiB, 0], np.int64), nz_mcpu=150, nz_mem=100 * MiB,
                     taint_filter_ok=ok, taint_prefer_cnt=cnt, preferred=preferred)


POOLS = 8  # node-pool label values (config 5's second nodeSelector key)


def make_c5(n_nodes: int = 100_000, n_specs: int = 1024, seed: int = SEED):
    """BASELINE config 5: nodes of the synthetic cluster + `n_specs` pod specs in genpod shape
    (pkg/client/nspod.go:34-126: one container, request == limit): cpu in [50m, 2000m], memory in [64Mi, 8Gi], 25 % with a
    2-key nodeSelector (instance type + node pool), every spec with a DoNotSchedule zone spread over its own label
    (maxSkew 1-5, nodeAffinityPolicy Honor: only nodes matching the selector are counted) and required anti-affinity to
    its own label on kubernetes.io/hostname.  Returns (nodes, [PodSpec], profile); pods are cycled round-robin."""
    nodes = make_nodes(n_nodes, seed)
    with np.errstate(over="ignore"):
        i = np.arange(n_nodes, dtype=np.int64)
        pool = (_uniform(seed, i, 11) * POOLS).astype(np.int32).clip(0, POOLS - 1) + 1
    nodes.label_cols = list(nodes.label_cols) + [pool, (i + 1).astype(np.int32)]  # col 2 = node pool, col 3 = kubernetes.io/hostname
    Z = zones_for(n_nodes)
    ok = np.array([1, 0, 1, 0], np.uint8)   # no tolerations: dedicated=infra:NoSchedule rejects
    cnt = np.array([0, 0, 1, 1], np.int32)  # maintenance=soon:PreferNoSchedule is not tolerated either
    with np.errstate(over="ignore"):
        p = np.arange(n_specs, dtype=np.int64)
        u = [_uniform(seed ^ 0x5EC5, p, k) for k in range(6)]
    included = {}
    pods = []
    for j in range(n_specs):
        cpu = 50 * (1 + int(u[0][j] * 40))                      # 50m .. 2000m
        mem = 64 * MiB * (1 + int(u[1][j] * 128))               # 64Mi .. 8Gi
        kw = {}
        inc = None
        if u[2][j] < 0.25:
            t, q = int(u[3][j] * len(INSTANCE_TYPES)) % len(INSTANCE_TYPES), int(u[4][j] * POOLS) % POOLS
            tt = np.zeros(len(INSTANCE_TYPES) + 1, np.uint8)
            tt[t + 1] = 1
            tq = np.zeros(POOLS + 1, np.uint8)
            tq[q + 1] = 1
            kw = dict(affinity_filter_active=True, has_node_selector=True, node_selector=[(0, tt), (2, tq)])
            if (t, q) not in included:  # one array per distinct selector (the engine dedups inclusion arrays by pointer)
                included[(t, q)] = ((nodes.label_cols[0] == t + 1) & (pool == q + 1)).astype(np.uint8)
            inc = included[(t, q)]
        spread = [M.SpreadConstraint(col=1, max_skew=1 + int(u[5][j] * 5) % 5, min_domains=1, hard=True, self_match=True,
                                     n_domains=Z, node_included=inc)]
        ipa = M.InterPodAffinity(key_cols=[3], key_ndom=[n_nodes], anti_keys=[0], anti_self=[True], anti_existing=[None],
                                 exist_anti=[None], score_existing=[None], score_self=[0], self_entries=[0])
        pods.append(M.PodSpec(req=np.array([cpu, mem, 0], np.int64), nz_mcpu=cpu, nz_mem=mem, taint_filter_ok=ok,
                              taint_prefer_cnt=cnt, spread=spread, ipa=ipa, **kw))
    return nodes, pods, M.Profile.default()


def make_config(name: str, n_nodes: int | None = None, seed: int = SEED, offset: int = 0, n_total: int = 0):
    """BASELINE.json configs -> (nodes, pod, profile).
    C2: NodeResourcesFit only (Filter + LeastAllocated).  C3/C4: default plugin set, pod tolerates
    dedicated=infra and has preferred node affinity on instance type (weights 10, 40)."""
    name = name.upper()
    if name == "C2":
        n = n_nodes or 10_000
        return make_nodes(n, seed, offset, n_total=n_total), examples_pod(), M.Profile.fit_only()
    if name in ("C3", "C4"):
        n = n_nodes or (100_000 if name == "C3" else 1_000_000)
        return (make_nodes(n, seed, offset, n_total=n_total), examples_pod(tolerate_infra=True, prefer_types=True),
                M.Profile.default())
    raise ValueError(name)
