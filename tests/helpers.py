"""Hand-built tiny snapshots (the reference's own fixtures) for parity tests."""
import numpy as np
from cluster_capacity_amd import model as M

GiB = 1 << 30
MiB = 1 << 20


def simple_nodes(alloc_mcpu, alloc_mem, alloc_pods, req_mcpu=None, req_mem=None, pod_count=None, alloc_eph=None,
                 taintset_id=None, unschedulable=None, label_cols=(), names=None):
    n = len(alloc_mcpu)
    z = np.zeros(n, np.int64)
    rc = np.array(req_mcpu if req_mcpu is not None else z, np.int64)
    rm = np.array(req_mem if req_mem is not None else z, np.int64)
    return M.NodesSoA(
        alloc=[np.array(alloc_mcpu, np.int64), np.array(alloc_mem, np.int64),
               np.array(alloc_eph if alloc_eph is not None else z, np.int64)],
        alloc_pods=np.array(alloc_pods, np.int32),
        req=[rc, rm, z.copy()],
        nz_mcpu=rc.copy(), nz_mem=rm.copy(),
        pod_count=np.array(pod_count if pod_count is not None else np.zeros(n), np.int32),
        taintset_id=np.array(taintset_id if taintset_id is not None else np.zeros(n), np.int32),
        unschedulable=np.array(unschedulable if unschedulable is not None else np.zeros(n), np.uint8),
        label_cols=[np.array(c, np.int32) for c in label_cols],
        names=names or [f"node-{i}" for i in range(n)],
    )


def simple_pod(mcpu, mem, eph=0, **kw):
    """One container with explicit cpu/memory requests (so NonZero == raw requests)."""
    nz_c = mcpu if mcpu else 100            # schedutil.DefaultMilliCPURequest
    nz_m = mem if mem else 200 * MiB        # schedutil.DefaultMemoryRequest
    return M.PodSpec(req=np.array([mcpu, mem, eph], np.int64), nz_mcpu=nz_c, nz_mem=nz_m, **kw)


def test_prediction_nodes():
    """pkg/framework/simulator_test.go:103-152 setupNodes (allocatable; pods=3 each)."""
    return simple_nodes([300, 400, 1200], [int(1e9), int(2e9), int(1e9)], [3, 3, 3],
                        names=["test-node-1", "test-node-2", "test-node-3"])


def test_prediction_pod():
    """simulator_test.go:180-214: 100m / 5e6 B, plus a zero-quantity extended resource which
    makes len(ScalarResources) != 0."""
    return simple_pod(100, int(5e6), has_scalar_entries=True)


def readme_nodes(k=4):
    """README.md:44-66 demo: k nodes x (2 CPU, 4 GB, 110 pods)."""
    return simple_nodes([2000] * k, [int(4e9)] * k, [110] * k, names=[f"kube-node-{i + 1}" for i in range(k)])


def examples_pod():
    """examples/pod.yaml: cpu 150m, memory 100Mi."""
    return simple_pod(150, 100 * MiB)


def random_case(rng, n):
    """Random snapshot + pod + profile exercising every plugin of the engine (shared by CPU and GPU tests)."""
    a_cpu = rng.choice([1000, 2000, 4000, 8000], n)
    a_mem = rng.choice([2, 4, 8, 16], n) * GiB
    a_eph = rng.choice([0, 10, 50], n) * GiB
    nodes = simple_nodes(a_cpu, a_mem, rng.integers(1, 12, n), req_mcpu=rng.integers(0, 900, n),
                           req_mem=rng.integers(0, 3, n) * GiB // 2, pod_count=rng.integers(0, 4, n), alloc_eph=a_eph,
                           taintset_id=rng.integers(0, 4, n), unschedulable=(rng.random(n) < 0.05),
                           label_cols=[rng.integers(0, 5, n), rng.integers(0, 3, n)])
    nodes.nz_mcpu = nodes.req[0] + rng.integers(0, 3, n) * 100  # existing pods without cpu requests
    nodes.nz_mem = nodes.req[1] + rng.integers(0, 2, n) * 200 * MiB
    t_in = lambda size, ids: np.isin(np.arange(size), ids).astype(np.uint8)
    pod = M.PodSpec(
        req=np.array([int(rng.choice([0, 100, 250, 500])), int(rng.choice([0, 256, 512])) * MiB, int(rng.choice([0, 0, 1])) * GiB]),
        nz_mcpu=0, nz_mem=0,
        taint_filter_ok=np.array([1, rng.integers(0, 2), 1, rng.integers(0, 2)], np.uint8),
        taint_prefer_cnt=np.array([0, 0, rng.integers(0, 3), rng.integers(0, 3)], np.int32),
        tolerates_unschedulable=bool(rng.integers(0, 2)),
        affinity_filter_active=bool(rng.integers(0, 2)),
        has_node_selector=bool(rng.integers(0, 2)), node_selector=[(1, t_in(3, [1, 2]))],
        has_required_terms=bool(rng.integers(0, 2)),
        required=[[(0, t_in(5, [1, 2, 3])), (1, t_in(3, [0, 1]))], [(0, t_in(5, [4]))], []],
        preferred=[(int(rng.integers(1, 100)), [(0, t_in(5, [2]))]), (int(rng.integers(1, 100)), [(1, t_in(3, [2])), (0, t_in(5, [0, 2, 4]))])]
        if rng.integers(0, 2) else [],
    )
    pod.nz_mcpu = int(pod.req[0]) or 100
    pod.nz_mem = int(pod.req[1]) or 200 * MiB
    prof = M.Profile(fit_res_w=(int(rng.integers(1, 4)), int(rng.integers(1, 4))),
                     w_taint=int(rng.integers(0, 4)), w_nodeaffinity=int(rng.integers(0, 3)), w_fit=int(rng.integers(0, 3)),
                     w_balanced=int(rng.integers(0, 2)))
    return nodes, pod, prof


def with_ports_and_images(rng, nodes, pod, prof):
    """Random host-port conflicts / ImageLocality scores on top of a random_case() (kept apart: the golden vectors of the
    plain cases depend on random_case()'s random stream)."""
    n = nodes.n
    if rng.integers(0, 3):
        pod.has_host_ports = True
        pod.host_ports_conflict = (rng.random(n) < 0.2).astype(np.uint8) if rng.integers(0, 2) else None
    if rng.integers(0, 3):
        pod.image_score = (rng.integers(0, 101, n) * (rng.random(n) < 0.5)).astype(np.uint8)
    prof.w_imagelocality = int(rng.integers(0, 3))
    if rng.integers(0, 4) == 0:
        prof.filter_mask &= ~M.F_NODEPORTS
    return nodes, pod, prof


def random_spread(rng, nodes, n_constraints=None):
    """Random hard topology spread constraints over the two label columns of random_case() snapshots
    (value id 0 = key absent, which exercises the missing-label path)."""
    n = nodes.n
    cons = []
    for col, ndom in ((1, 2), (0, 4)):
        if n_constraints is not None and len(cons) >= n_constraints:
            break
        if rng.integers(0, 3) == 0 and n_constraints is None:
            continue
        cons.append(M.SpreadConstraint(
            col=col, max_skew=int(rng.integers(1, 4)), min_domains=int(rng.integers(1, 4)), hard=True,
            self_match=bool(rng.integers(0, 4) != 0), n_domains=ndom,
            node_match_count=rng.integers(0, 3, n).astype(np.int32) if rng.integers(0, 2) else None,
            node_included=(rng.random(n) < 0.9).astype(np.uint8) if rng.integers(0, 2) else None))
    return cons


def random_ipa(rng, nodes):
    """Random InterPodAffinity description over the two label columns of random_case() snapshots."""
    n = nodes.n
    keys, ndom = [1, 0], [2, 4]
    sparse = lambda hi, p=0.1: (rng.integers(1, hi + 1, n) * (rng.random(n) < p)).astype(np.int32)
    n_aff, n_anti = int(rng.integers(0, 3)), int(rng.integers(0, 3))
    score_existing = [(rng.integers(-5, 6, n) * (rng.random(n) < 0.15)).astype(np.int64) if rng.integers(0, 2) else None
                      for _ in keys]
    entries = 0
    for k, se in enumerate(score_existing):
        if se is not None:
            entries += int(np.count_nonzero(se[nodes.label_cols[keys[k]] != 0]))
    score_self = [int(rng.integers(-2, 4)) if rng.integers(0, 2) else 0 for _ in keys]
    return M.InterPodAffinity(
        key_cols=keys, key_ndom=ndom,
        aff_keys=[int(rng.integers(0, 2)) for _ in range(n_aff)], self_aff=bool(rng.integers(0, 2)),
        aff_existing=sparse(2, 0.05) if rng.integers(0, 2) else None,
        anti_keys=[int(rng.integers(0, 2)) for _ in range(n_anti)], anti_self=[bool(rng.integers(0, 2)) for _ in range(n_anti)],
        anti_existing=[sparse(2, 0.03) if rng.integers(0, 2) else None for _ in range(n_anti)],
        exist_anti=[sparse(2, 0.03) if rng.integers(0, 3) == 0 else None for _ in keys],
        score_existing=score_existing, score_self=score_self, entries_existing=entries,
        self_entries=[1 if w else int(rng.integers(0, 2)) for w in score_self])


# Child processes started by the tests (native host, C demo): a fresh GPU box pages the HIP runtime, libccsim's code objects and --
# on the sharded path -- a 570 MB librccl in from a cold image, which has taken more than a minute (GPUTEST r02).  The limit only
# has to catch a real hang (ccsim_dist_comm_init bounds its own wait and reports): generous, one knob.
SUBPROC_TIMEOUT = int(__import__("os").environ.get("CCSIM_TEST_SUBPROC_TIMEOUT", "600"))

