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
