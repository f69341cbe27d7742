"""The volume plugins -- VolumeRestrictions, NodeVolumeLimits, VolumeBinding, VolumeZone -- at the engine boundary:
`ccsim_pod.volume_veto` (the caller's per-node verdict against the snapshot's pods: the code of the first of the four that rejects the
node) and `volume_exclusive` (the pod's own disks conflict with a clone's: one clone per node).  They run AFTER NodeResourcesFit and
before PodTopologySpread (default_plugins.go:40-45), so a node that also fails Fit reports Fit (framework.go:897-930: the first failing
plugin sets the status), and the codes 1..3 are plain Unschedulable (preemption candidates), the rest UnschedulableAndUnresolvable.

The reference vendors no test that drives these plugins through a scheduling cycle: the known answers below follow the cited lines
("parity unpinned", like the rest of the loop).  CPU: the oracle.  GPU: the HIP engine in every mode against the oracle.  The hosts'
side (which objects give which code) is tests/test_volume_ingest.py."""
import numpy as np
import pytest

import helpers as H
from cluster_capacity_amd import capi, model as M, report as R


def _slot(code):
    return M.R_VOL0 + code - 1


def test_ka_veto_codes_report_after_fit(ccref):
    """5 nodes, room for 2 clones each (cpu).  Node 0: disk conflict with an existing pod; node 1: volume zone conflict; node 2: both a
    zone conflict AND too small for the pod (Fit reports, not the volume plugin); nodes 3, 4 free."""
    nodes = H.simple_nodes([2000, 2000, 500, 2000, 2000], [8 << 30] * 5, [10] * 5)
    pod = H.simple_pod(1000, 1 << 30)
    pod.volume_veto = np.array([M.VOL_DISK_CONFLICT, M.VOL_ZONE, M.VOL_ZONE, 0, 0], np.uint8)
    r = ccref.run(M.Profile.default(), nodes, pod)
    assert r.placed == 4 and r.per_node_count.tolist() == [0, 0, 0, 2, 2] and r.stop == M.STOP_UNSCHEDULABLE
    assert r.hist[_slot(M.VOL_DISK_CONFLICT)] == 1 and r.hist[_slot(M.VOL_ZONE)] == 1
    assert r.hist[M.R_RES0] == 3  # "Insufficient cpu": node 2 (never fits) and the two full nodes
    assert r.n_code_unschedulable == 3  # the disk conflict and the two full nodes; node 2 asks for more than it has (Unresolvable)
    msg = R.stop_reason(r, 5, 0)
    assert "1 node(s) had no available disk" in msg and "1 node(s) had no available volume zone" in msg and "3 Insufficient cpu" in msg


def test_ka_exclusive_disks_one_clone_per_node(ccref):
    """An EBS volume (or a read-write GCE PD): a clone conflicts with the next one on the same node (volume_restrictions.go:105-150), so
    every node takes one; at the terminal cycle they all report the disk conflict -- except the node that is also full."""
    nodes = H.simple_nodes([4000, 4000, 1000], [8 << 30] * 3, [10] * 3)
    pod = H.simple_pod(1000, 1 << 30)
    pod.volume_exclusive = True
    r = ccref.run(M.Profile.default(), nodes, pod)
    assert r.placed == 3 and r.per_node_count.tolist() == [1, 1, 1]
    assert r.hist[_slot(M.VOL_DISK_CONFLICT)] == 2 and r.hist[M.R_RES0] == 1 and r.n_code_unschedulable == 3


def test_ka_host_ports_report_before_fit_disks_after(ccref):
    """Host ports AND exclusive disks: a node that holds a clone fails NodePorts first (default_plugins.go:38-41)."""
    nodes = H.simple_nodes([4000, 1000], [8 << 30] * 2, [10] * 2)
    pod = H.simple_pod(1000, 1 << 30)
    pod.volume_exclusive, pod.has_host_ports = True, True
    r = ccref.run(M.Profile.default(), nodes, pod)
    assert r.placed == 2 and r.hist[M.R_NODEPORTS] == 2 and r.hist[_slot(M.VOL_DISK_CONFLICT)] == 0 and r.hist[M.R_RES0] == 0


def _decorate(rng, nodes, pod):
    n = nodes.n
    kind = int(rng.integers(0, 4))
    if kind != 1:
        codes = rng.integers(1, M.VOL_CODES + 1, n).astype(np.uint8)
        pod.volume_veto = np.where(rng.random(n) < rng.choice([0.05, 0.4, 0.9]), codes, 0).astype(np.uint8)
    if kind >= 1:
        pod.volume_exclusive = True
    if kind == 3:
        pod.has_host_ports = True
        pod.host_ports_conflict = (rng.random(n) < 0.2).astype(np.uint8)
    return pod


def _same(got, ref):
    assert got.placed == ref.placed and got.stop == ref.stop
    assert np.array_equal(got.per_node_count, ref.per_node_count) and np.array_equal(got.log, ref.log)
    if ref.stop == M.STOP_UNSCHEDULABLE:
        assert np.array_equal(got.hist, ref.hist) and got.n_code_unschedulable == ref.n_code_unschedulable
        assert np.array_equal(got.hist_taintset[: len(ref.hist_taintset)], ref.hist_taintset)


@pytest.mark.gpu
def test_gpu_known_answers(ccref):
    for build in (test_ka_veto_codes_report_after_fit, test_ka_exclusive_disks_one_clone_per_node, test_ka_host_ports_report_before_fit_disks_after):
        class Engine:  # the same known answers through the HIP engine: an object with the oracle's `run`
            @staticmethod
            def run(prof, nodes, pod, max_limit=0):
                e = capi.Engine(device=0)
                try:
                    e.load(nodes, pod, prof)
                    return e.run(max_limit=max_limit, mode="sequential", log_cap=64)
                finally:
                    e.close()
        build(Engine)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(16))
def test_gpu_random_volume_verdicts_vs_oracle(ccref, monkeypatch, seed):
    rng = np.random.default_rng(8800 + seed)
    nodes, pod, prof = H.random_case(rng, int(rng.integers(1, 3000)))
    pod = _decorate(rng, nodes, pod)
    limit = int(rng.choice([0, 0, 0, 31]))
    ref = ccref.run(prof, nodes, pod, max_limit=limit)
    for mode, persist in (("sequential", "1"), ("batched", "1"), ("batched", "0")):
        monkeypatch.setenv("CCSIM_PERSIST", persist)
        e = capi.Engine(device=0)
        e.load(nodes, pod, prof)
        _same(e.run(max_limit=limit, mode=mode, log_cap=max(1, ref.placed)), ref)
        if mode == "batched":  # the blind fast path, then a second run on the restored state
            e.reset_state()
            got = e.run(max_limit=limit, mode=mode, want_log=False, log_cap=0)
            assert got.placed == ref.placed and np.array_equal(got.per_node_count, ref.per_node_count)
            if ref.stop == M.STOP_UNSCHEDULABLE:
                assert np.array_equal(got.hist, ref.hist) and got.n_code_unschedulable == ref.n_code_unschedulable
        e.close()


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(8))
def test_gpu_volume_verdicts_with_topology_coupled_plugins_vs_oracle(ccref, monkeypatch, seed):
    """The volume plugins sit between NodeResourcesFit and PodTopologySpread / InterPodAffinity: with hard spread constraints and
    (anti-)affinity terms in play, windowed and one pass per placement."""
    rng = np.random.default_rng(8900 + seed)
    nodes, pod, prof = H.random_case(rng, int(rng.integers(2, 1200)))
    pod.spread = H.random_spread(rng, nodes)
    if seed % 2:
        pod.ipa = H.random_ipa(rng, nodes)
    pod = _decorate(rng, nodes, pod)
    limit = int(rng.choice([0, 0, 90]))
    ref = ccref.run(prof, nodes, pod, max_limit=limit)
    for cw in ("1", "0"):
        monkeypatch.setenv("CCSIM_CW", cw)
        e = capi.Engine(device=0)
        e.load(nodes, pod, prof)
        _same(e.run(max_limit=limit, mode="sequential", log_cap=max(1, ref.placed)), ref)
        e.close()


@pytest.mark.gpu
def test_gpu_volume_verdicts_on_shards(ccref):
    import test_gpu_parity as T
    rng = np.random.default_rng(8999)
    nodes, pod, prof = H.random_case(rng, 1700)
    pod = _decorate(rng, nodes, pod)
    pod.volume_exclusive = True
    prof.filter_mask |= M.F_FIT
    ref = ccref.run(prof, nodes, pod)
    for shards, mode in ((2, "sequential"), (3, "batched")):
        res, log = T._LocalShards(nodes, pod, prof, shards).run(0, mode, max(1, ref.placed), 4)
        assert all(r.placed == ref.placed and r.stop == ref.stop for r in res)
        assert np.array_equal(np.concatenate([r.per_node_count for r in res]), ref.per_node_count)
        assert np.array_equal(log[: ref.placed], ref.log) and np.array_equal(sum(r.hist for r in res), ref.hist)
        assert sum(r.n_code_unschedulable for r in res) == ref.n_code_unschedulable


def test_several_pod_specs_with_volumes_are_refused_by_the_window_engine_on_the_abi(recorder_free=None):
    """(documented in include/ccsim.h: ccsim_set_pods answers -ENOSYS, the hosts then place one cycle at a time)"""
    import re
    src = open(__import__("os").path.join(__import__("os").path.dirname(__file__), "..", "cluster-capacity_amd", "csrc", "ccsim_engine.hip")).read()
    assert re.search(r'volume_exclusive \|\| q\.volume_veto\) return fail\(e, -ENOSYS', src)
