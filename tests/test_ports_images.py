"""NodePorts (P/nodeports/node_ports.go:67-176) and ImageLocality (P/imagelocality/image_locality.go:54-127) -- SURVEY 8(f) row 4.

The reference vendors no tests for either plugin, so the known answers below are derived by hand from the cited lines
("parity unpinned", like every score of this path).  CPU: the oracle's unit function, both hosts' string-side evaluation,
the oracle loop.  GPU: the HIP engine (every mode) against the oracle; the CLI end to end."""
import io
import json

import numpy as np
import pytest

from helpers import SUBPROC_TIMEOUT
import yaml

import helpers as H
from cluster_capacity_amd import capi, cli, ingest, model as M, report as R
from test_native_host import CASES, _write

MB = 1024 * 1024


# ---- ImageLocality arithmetic -----------------------------------------------------------------------------------
def test_image_locality_known_answers(ccref):
    """calculatePriority / scaledImageScore (image_locality.go:84-115), 2 containers, 2 nodes:
    one 40 MB image on one of two nodes: 40M * 1/2 = 20M < 23M (minThreshold)              -> 0
    one 250 MB image on one of two nodes: 100 * (250M/2 - 23M) / (2 * 1000M - 23M) = 5.16   -> 5"""
    assert ccref.image_locality_score([40 * MB], [1], 2, 2) == 0
    assert ccref.image_locality_score([250 * MB], [1], 2, 2) == 5
    assert ccref.image_locality_score([], [], 2, 2) == 0                       # no image of the pod on the node
    assert ccref.image_locality_score([4000 * MB], [3], 3, 1) == 100           # clamped to maxContainerThreshold x containers
    assert ccref.image_locality_score([600 * MB, 600 * MB], [4, 2], 4, 2) == 100 * (600 * MB + 300 * MB - 23 * MB) // (2000 * MB - 23 * MB)
    assert ccref.image_locality_score([250 * MB], [1], 3, 1) == 100 * (int(250 * MB * (1 / 3)) - 23 * MB) // (1000 * MB - 23 * MB)


def test_python_image_score_equals_oracle():
    import ccref_py

    rng = np.random.default_rng(77)
    for _ in range(3000):
        total = int(rng.integers(1, 5000))
        k = int(rng.integers(0, 5))
        sizes = [int(rng.integers(0, 3_000_000_000)) for _ in range(k)]
        nn = [int(rng.integers(1, total + 1)) for _ in range(k)]
        nc = int(rng.integers(max(1, k), 7))
        assert ingest.image_locality_score(list(zip(sizes, nn)), total, nc) == ccref_py.image_locality_score(sizes, nn, total, nc)


def test_normalized_image_name():
    f = ingest.normalized_image_name  # image_locality.go:122-127
    assert f("busybox") == "busybox:latest" and f("busybox:1.36") == "busybox:1.36"
    assert f("localhost:5000/app") == "localhost:5000/app:latest" and f("localhost:5000/app:v2") == "localhost:5000/app:v2"
    assert f("gcr.io/x/y@sha256:abc") == "gcr.io/x/y@sha256:abc"


# ---- host ports ------------------------------------------------------------------------------------------------------
def test_host_port_conflicts():
    """HostPortInfo.CheckConflict (kube-scheduler/framework/types.go:499-528) after sanitize (:530-538)."""
    c = ingest.ports_conflict
    used = {("0.0.0.0", "TCP", 8080)}
    assert c([("10.0.0.1", "TCP", 8080)], used) and c([("0.0.0.0", "TCP", 8080)], used)
    assert not c([("10.0.0.1", "UDP", 8080)], used) and not c([("10.0.0.1", "TCP", 8081)], used)
    used = {("10.0.0.9", "TCP", 8080)}
    assert c([("0.0.0.0", "TCP", 8080)], used) and c([("10.0.0.9", "TCP", 8080)], used) and not c([("10.0.0.1", "TCP", 8080)], used)
    spec = {"containers": [{"ports": [{"containerPort": 80}, {"containerPort": 81, "hostPort": 0}, {"containerPort": 82, "hostPort": 9000, "protocol": "UDP"}]}],
            "initContainers": [{"ports": [{"hostPort": 1}]}, {"restartPolicy": "Always", "ports": [{"hostPort": 2, "hostIP": "1.2.3.4"}]}]}
    # util.GetHostPorts (S/util/utils.go:175-210): hostPort > 0; init containers only when restartable
    assert ingest.host_ports(spec) == [("1.2.3.4", "TCP", 2), ("0.0.0.0", "UDP", 9000)]


def _ports_images_snapshot():
    nodes, pods, pod, exclude = CASES["ports-images"]()
    return ingest.build_snapshot(nodes, pods, pod, exclude), pod


def test_ingest_and_oracle_known_answer(ccref):
    """tests/test_native_host.py ports_images_case, by hand (5 nodes, 2 containers):
    image states: gcr.io/40:latest = (40 MB reported by w0, the first node by name; on 2 nodes), gcr.io/250:latest = (250 MB, 1 node)
      w0: int(40M * 2/5) = 16M < 23M -> 0;  w1: int(250M / 5) + 16M = 66M -> 100 * (66M - 23M) / (2000M - 23M) = 2;  others 0
    ports: the pod wants 10.0.0.1:8080/TCP; w3's pod holds 0.0.0.0:8080/TCP (conflict), w4's 10.0.0.9:8080/TCP and w2's 8080/UDP (none)
    run: one clone per node (the clone's own port), w3 never: 4 placements, w1 first (+2 ImageLocality), then w0, then w2 / w4
    (their existing pods count 100m / 200Mi non-zero requests), all 5 nodes end without free ports."""
    snap, _ = _ports_images_snapshot()
    assert snap.pod.has_host_ports and snap.pod.host_ports_conflict.tolist() == [0, 0, 0, 1, 0]
    assert snap.pod.image_score.tolist() == [0, 2, 0, 0, 0]
    r = ccref.run(M.Profile.default(), snap.nodes, snap.pod)
    assert r.placed == 4 and r.stop == M.STOP_UNSCHEDULABLE and r.log.tolist() == [1, 0, 2, 4]
    assert r.hist[M.R_NODEPORTS] == 5 and r.hist.sum() == 5 and r.n_code_unschedulable == 5
    assert R.stop_reason(r, 5, 0, taint_reasons=snap.taint_reasons, scalar_names=snap.scalar_names) == (
        "Unschedulable: 0/5 nodes are available: 5 node(s) didn't have free ports for the requested pod ports. "
        "preemption: 0/5 nodes are available: 5 No preemption victims found for incoming pod.")
    # NodePorts disabled in the profile: the README arithmetic is back (2000m / 150m = 13 per node)
    off = M.Profile(filter_mask=M.F_ALL & ~M.F_NODEPORTS)
    assert ccref.run(off, snap.nodes, snap.pod).placed == 13 * 5
    # ImageLocality disabled: w0 (lowest index among the empty nodes) goes first
    assert ccref.run(M.Profile(w_imagelocality=0), snap.nodes, snap.pod).log.tolist() == [0, 1, 2, 4]


def test_pods_the_hosts_still_refuse():
    """(volumes: round 5 evaluates the volume plugins -- tests/test_volume_ingest.py; every such pod ends at a PreFilter, as in the reference)"""
    nodes, pods, pod, _ = CASES["readme"]()
    pod["spec"]["volumes"] = [{"name": "scratch", "emptyDir": {}}, {"name": "cfg", "configMap": {"name": "x"}}]
    ingest.build_snapshot(nodes, pods, pod)  # node-independent volumes are fine
    pod["spec"]["volumes"].append({"name": "data", "persistentVolumeClaim": {"claimName": "pvc-1"}})
    assert ingest.build_snapshot(nodes, pods, pod).pod.prefilter_reject == 'persistentvolumeclaim "pvc-1" not found'
    pod["spec"]["volumes"][-1] = {"name": "data", "ephemeral": {"volumeClaimTemplate": {}}}
    assert ingest.build_snapshot(nodes, pods, pod).pod.prefilter_reject == f'waiting for ephemeral volume controller to create the persistentvolumeclaim "{pod["metadata"]["name"]}-0-data"'
    pod["spec"]["volumes"].pop()
    # (DRA: the fake cluster holds no ResourceClaim -- the plugin's PreFilter rejects the pod; tests/test_volume_ingest.py)
    pod["spec"]["resourceClaims"] = [{"name": "gpu"}]
    assert "none of the supported fields are set" in ingest.build_snapshot(nodes, pods, pod).pod.prefilter_reject
    with pytest.raises(NotImplementedError, match="filter point of DynamicResources"):
        ingest.build_snapshot(nodes, pods, pod, dra_partial=True)


decorate = H.with_ports_and_images


@pytest.mark.parametrize("seed", range(12))
def test_oracle_ports_clamp_equivalence(ccref, seed):
    """The engine keeps 'one clone per node' as a clamped pod capacity (csrc: k_ports_clamp): on the oracle, NodePorts and
    the same snapshot with allocatable pods = min(real, pods + 1) (+ the static conflicts as an unschedulable-style veto)
    place identically -- the argument the engine relies on."""
    rng = np.random.default_rng(4400 + seed)
    nodes, pod, prof = H.random_case(rng, int(rng.integers(1, 800)))
    pod.has_host_ports = True
    prof.filter_mask |= M.F_NODEPORTS | M.F_FIT
    a = ccref.run(prof, nodes, pod)
    clamped = nodes.copy()
    clamped.alloc_pods = np.minimum(nodes.alloc_pods, nodes.pod_count + 1).astype(np.int32)
    import copy
    q = copy.copy(pod)
    q.has_host_ports = False
    b = ccref.run(prof, clamped, q)
    assert a.placed == b.placed and np.array_equal(a.log, b.log) and a.per_node_count.max(initial=0) <= 1


# ---- GPU ---------------------------------------------------------------------------------------------------------------
def _same(got, ref):
    assert got.placed == ref.placed and got.stop == ref.stop
    assert np.array_equal(got.per_node_count, ref.per_node_count) and np.array_equal(got.log, ref.log)
    if ref.stop == M.STOP_UNSCHEDULABLE:
        assert np.array_equal(got.hist, ref.hist) and got.n_code_unschedulable == ref.n_code_unschedulable
        assert np.array_equal(got.hist_taintset[: len(ref.hist_taintset)], ref.hist_taintset)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(10))
def test_gpu_random_ports_and_images_vs_oracle(ccref, monkeypatch, seed):
    rng = np.random.default_rng(5500 + seed)
    nodes, pod, prof = decorate(rng, *H.random_case(rng, int(rng.integers(1, 2500))))
    limit = int(rng.choice([0, 0, 0, 23]))
    ref = ccref.run(prof, nodes, pod, max_limit=limit)
    for mode, persist in (("sequential", "1"), ("batched", "1"), ("batched", "0")):
        monkeypatch.setenv("CCSIM_PERSIST", persist)
        e = capi.Engine(device=0)
        e.load(nodes, pod, prof)
        _same(e.run(max_limit=limit, mode=mode, log_cap=max(1, ref.placed)), ref)
        if mode == "batched":  # the blind fast path (no log), then a second run on the restored state
            e.reset_state()
            got = e.run(max_limit=limit, mode=mode, want_log=False, log_cap=0)
            assert got.placed == ref.placed and np.array_equal(got.per_node_count, ref.per_node_count)
        e.close()


@pytest.mark.gpu
def test_gpu_image_scores_on_the_c3_snapshot(ccref):
    """ImageLocality on the narrow / persistent path at a BASELINE-shaped snapshot (scores ride in the packed static word)."""
    from cluster_capacity_amd import synth

    nodes, pod, prof = synth.make_config("C3", n_nodes=20_000, seed=99)
    rng = np.random.default_rng(1)
    pod.image_score = (rng.integers(0, 101, nodes.n) * (rng.random(nodes.n) < 0.3)).astype(np.uint8)
    ref = ccref.run(prof, nodes, pod, max_limit=3000, threads=8)
    for mode in ("sequential", "batched"):
        e = capi.Engine(device=0)
        e.load(nodes, pod, prof)
        _same(e.run(max_limit=3000, mode=mode, log_cap=3000), ref)
        e.close()
    # the whole run until Unschedulable on the blind fast path, at a size the oracle finishes in seconds (the 20k-node full run
    # costs the CPU oracle two minutes: 1.1 M cycles x 20k nodes)
    nodes, pod, prof = synth.make_config("C3", n_nodes=1500, seed=98)
    pod.image_score = (rng.integers(0, 101, nodes.n) * (rng.random(nodes.n) < 0.3)).astype(np.uint8)
    ref = ccref.run(prof, nodes, pod, threads=8, want_log=False)
    e = capi.Engine(device=0)
    e.load(nodes, pod, prof)
    got = e.run(mode="batched", want_log=False, log_cap=0)
    assert got.placed == ref.placed and np.array_equal(got.per_node_count, ref.per_node_count) and np.array_equal(got.hist, ref.hist)
    e.close()


@pytest.mark.gpu
def test_gpu_ports_pod_set_after_runs_of_another_pod(ccref):
    """the NodePorts clamp is built from the pod counts as they are when the pod is set (clones of an earlier pod spec are
    ordinary pods of the node), rebuilt on a reset, and a second run of the same ports pod finds the first run's clones in the way --
    with the FitError histogram still summing to N."""
    rng = np.random.default_rng(77)
    nodes, pod_a, prof = H.random_case(rng, 800)
    prof.filter_mask |= M.F_NODEPORTS | M.F_FIT
    pod_b = H.simple_pod(100, 64 * H.MiB, has_host_ports=True, taint_filter_ok=pod_a.taint_filter_ok, taint_prefer_cnt=pod_a.taint_prefer_cnt)
    def after(nd, pod, counts):  # NodeInfo.update (types.go:409-428) x the clones per node: the snapshot a run leaves behind
        out = nd.copy()
        c = counts.astype(np.int64)
        out.req = [r + c * int(pod.req[k]) if k < len(pod.req) else r for k, r in enumerate(out.req)]
        out.nz_mcpu, out.nz_mem = out.nz_mcpu + c * int(pod.nz_mcpu), out.nz_mem + c * int(pod.nz_mem)
        out.pod_count = (out.pod_count + counts).astype(np.int32)
        return out

    ref_a = ccref.run(prof, nodes, pod_a, max_limit=300)
    after_a = after(nodes, pod_a, ref_a.per_node_count)  # what the oracle sees as "existing pods" for B
    ref_b = ccref.run(prof, after_a, pod_b, max_limit=0)
    e = capi.Engine(device=0)
    e.load(nodes, pod_a, prof)
    a = e.run(max_limit=300, mode="sequential", log_cap=300)
    assert np.array_equal(a.log, ref_a.log)
    e.set_pod(pod_b)  # no reset: A's clones stay on the nodes
    b = e.run(max_limit=0, mode="sequential", log_cap=max(1, ref_b.placed))
    _same(b, ref_b)
    assert int(np.sum(b.hist)) + int(np.sum(b.hist_taintset)) >= nodes.n  # (NodeResourcesFit may give a node several reasons)
    # a second run of B: every node that took a clone now fails NodePorts; nothing can be placed, and the reasons still cover every node
    c = e.run(max_limit=0, mode="sequential", log_cap=1)
    ref_c = ccref.run(prof, after(after_a, pod_b, ref_b.per_node_count), H.simple_pod(100, 64 * H.MiB, has_host_ports=True, taint_filter_ok=pod_a.taint_filter_ok, taint_prefer_cnt=pod_a.taint_prefer_cnt,
                                                           host_ports_conflict=(ref_b.per_node_count > 0).astype(np.uint8)), max_limit=0)
    assert c.placed == ref_c.placed == 0 and np.array_equal(c.hist, ref_c.hist) and np.array_equal(c.hist_taintset[: len(ref_c.hist_taintset)], ref_c.hist_taintset)
    # after a reset B sees the pristine snapshot again
    e.reset_state()
    _same(e.run(max_limit=0, mode="batched", log_cap=1 << 20), ccref.run(prof, nodes, pod_b, max_limit=0))
    e.close()


@pytest.mark.gpu
def test_gpu_ports_sharded_engines_and_switching_pods(ccref):
    """Two shards of one snapshot on one GPU (the multi-GPU protocol) with NodePorts; then the same engine takes a pod
    without host ports: the clamped pod capacity must be gone."""
    import test_gpu_parity as T

    rng = np.random.default_rng(31)
    nodes, pod, prof = H.random_case(rng, 1500)
    pod.has_host_ports, pod.host_ports_conflict = True, (rng.random(nodes.n) < 0.3).astype(np.uint8)
    prof.filter_mask |= M.F_NODEPORTS | M.F_FIT
    ref = ccref.run(prof, nodes, pod)
    pod.image_score = (rng.integers(0, 101, nodes.n) * (rng.random(nodes.n) < 0.5)).astype(np.uint8)
    ref = ccref.run(prof, nodes, pod)
    for mode in ("sequential", "batched"):
        res, log = T._LocalShards(nodes, pod, prof, 3).run(0, mode, max(1, ref.placed), 4)
        assert all(r.placed == ref.placed and r.stop == ref.stop for r in res)
        assert np.array_equal(np.concatenate([r.per_node_count for r in res]), ref.per_node_count)
        assert np.array_equal(log[: ref.placed], ref.log) and np.array_equal(sum(r.hist for r in res), ref.hist)
    e = capi.Engine(device=0)
    e.load(nodes, pod, prof)
    _same(e.run(mode="batched", log_cap=max(1, ref.placed)), ref)
    import copy
    q = copy.copy(pod)
    q.has_host_ports, q.host_ports_conflict = False, None
    ref2 = ccref.run(prof, nodes, q)
    e.reset_state()
    e.set_pod(q)
    _same(e.run(mode="batched", log_cap=max(1, ref2.placed)), ref2)
    e.close()


@pytest.mark.gpu
def test_gpu_cli_ports_images_both_hosts(tmp_path):
    from cluster_capacity_amd import build as B
    import subprocess

    nodes, pods, pod, _ = CASES["ports-images"]()
    podspec, snaps = _write(tmp_path, "json", nodes, pods, pod)
    args = ["--podspec", podspec, "--snapshot", snaps[0], "--verbose"]
    buf = io.StringIO()
    assert cli.main(args, out=buf) == 0
    txt = buf.getvalue()
    assert "The cluster can schedule 4 instance(s) of the pod small-pod." in txt
    assert "5 node(s) didn't have free ports for the requested pod ports" in txt
    p = subprocess.run([B.build_host()] + args, capture_output=True, text=True, timeout=SUBPROC_TIMEOUT)
    assert p.returncode == 0, p.stderr
    assert p.stdout == txt
