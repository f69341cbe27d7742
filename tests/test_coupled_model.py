"""The exactness argument of the windowed mode for ONE template with topology-coupled plugins (tests/coupled_model.py: classes of
nodes the coupled plugins cannot tell apart, W best per class by the node-local score, W cycles per scan from class heads +
touched nodes + domain tables) checked on the CPU: the same placement log and stop as the oracle's literal loop, whatever the
window size.  Hard and soft spread constraints over shared keys and over a unique-per-node key (hostname), inter-pod (anti)affinity
over the same keys, existing pods' counts, missing labels, node inclusion, maxSkew / minDomains, host ports, images."""
import numpy as np
import pytest

import helpers as H
from cluster_capacity_amd import model as M
from coupled_model import CoupledWindowModel, ShardedCoupledWindowModel


def coupled_case(rng, n, roomy=False):
    nodes, pod, prof = H.with_ports_and_images(rng, *H.random_case(rng, n))
    if roomy:  # long runs: several windows per case, nodes that take many clones
        nodes.alloc_pods = (nodes.alloc_pods * 6).astype(np.int32)
        nodes.alloc = [a * 6 for a in nodes.alloc]
    host = np.arange(1, n + 1, dtype=np.int32)
    host[rng.random(n) < 0.05] = 0  # a few nodes without the hostname label
    nodes.label_cols = list(nodes.label_cols) + [host]
    kind = int(rng.integers(0, 4))
    cons = []
    if kind != 3:
        for col, ndom, hostname in ((1, 2, False), (0, 4, False), (2, n, True)):
            if rng.integers(0, 2):
                cons.append(M.SpreadConstraint(
                    col=col, max_skew=int(rng.integers(1, 4)), min_domains=int(rng.integers(1, 4)), hard=bool(rng.integers(0, 2)),
                    self_match=bool(rng.integers(0, 4) != 0), n_domains=ndom, is_hostname=hostname,
                    node_match_count=rng.integers(0, 3, n).astype(np.int32) if rng.integers(0, 2) else None,
                    node_included=(rng.random(n) < 0.9).astype(np.uint8) if rng.integers(0, 2) else None))
    pod.spread = cons
    if kind >= 2 or not cons:
        ipa = H.random_ipa(rng, nodes)
        if rng.integers(0, 2):  # terms over the unique-per-node key as well (hostname anti-affinity, the C5 shape)
            ipa.key_cols, ipa.key_ndom = ipa.key_cols + [2], ipa.key_ndom + [n]
            ipa.exist_anti.append(None)
            se = (rng.integers(-3, 4, n) * (rng.random(n) < 0.1)).astype(np.int64) if rng.integers(0, 2) else None
            ipa.score_existing.append(se)
            ipa.entries_existing += int(np.count_nonzero(se[host != 0])) if se is not None else 0
            w = int(rng.integers(-2, 3))
            ipa.score_self.append(w), ipa.self_entries.append(1 if w else 0)
            if rng.integers(0, 2):
                ipa.anti_keys.append(2), ipa.anti_self.append(True), ipa.anti_existing.append(None)
        pod.ipa = ipa
    prof.w_topologyspread, prof.w_interpodaffinity = int(rng.integers(0, 3)), int(rng.integers(0, 3))
    return nodes, pod, prof


@pytest.mark.parametrize("window", [1, 4, 64])
@pytest.mark.parametrize("seed", range(40))
def test_coupled_window_model_vs_oracle(ccref, seed, window):
    rng = np.random.default_rng(7100 + seed)
    nodes, pod, prof = coupled_case(rng, int(rng.integers(12, 160)))
    limit = int(rng.choice([0, 0, 29]))
    ref = ccref.run(prof, nodes, pod, max_limit=limit or 4000)
    log, stop, scans, stats = CoupledWindowModel(prof, nodes.copy(), pod, ccref.go_log, window=window).run(limit or 4000)
    assert log == ref.log.tolist(), (seed, window)
    assert (stop == "Unschedulable") == (ref.stop == M.STOP_UNSCHEDULABLE)
    if window == 1:
        assert scans == len(log) + (stop == "Unschedulable")


@pytest.mark.parametrize("window", [7, 64])
@pytest.mark.parametrize("seed", range(24))
def test_coupled_window_model_long_runs(ccref, seed, window):
    rng = np.random.default_rng(7300 + seed)
    nodes, pod, prof = coupled_case(rng, int(rng.integers(12, 90)), roomy=True)
    ref = ccref.run(prof, nodes, pod, max_limit=1500)
    log, stop, scans, stats = CoupledWindowModel(prof, nodes.copy(), pod, ccref.go_log, window=window).run(1500)
    assert log == ref.log.tolist(), (seed, window)
    assert (stop == "Unschedulable") == (ref.stop == M.STOP_UNSCHEDULABLE)


@pytest.mark.parametrize("window,list_len", [(1, 1), (7, 2), (64, 8), (64, 64)])
@pytest.mark.parametrize("seed", range(40))
def test_device_plan_vs_oracle(ccref, seed, window, list_len):
    """The simplifications of the HIP port (no `counted` bit in the class key, truncated class lists, tracked minimum of a
    unique-key hard constraint): same log as the oracle."""
    rng = np.random.default_rng(7100 + seed)
    nodes, pod, prof = coupled_case(rng, int(rng.integers(12, 160)), roomy=seed % 3 == 0)
    limit = 1500 if seed % 3 == 0 else int(rng.choice([0, 0, 29])) or 4000
    ref = ccref.run(prof, nodes, pod, max_limit=limit)
    log, stop, scans, stats = CoupledWindowModel(prof, nodes.copy(), pod, ccref.go_log, window=window, device_plan=True, list_len=list_len).run(limit)
    assert log == ref.log.tolist(), (seed, window, list_len)
    assert (stop == "Unschedulable") == (ref.stop == M.STOP_UNSCHEDULABLE)


def test_zone_spread_with_hostname_anti_affinity(ccref):
    """The config-5 pod shape as ONE template: DoNotSchedule zone spread (maxSkew 1) + required hostname anti-affinity against its
    own clones.  One clone per node, zones filled evenly: 16 zones x heterogeneous nodes -> the classes are the zones (times the
    few distinct hostname-level states), and a 64-cycle window does 64 placements per scan."""
    rng = np.random.default_rng(5)
    n, zones = 400, 16
    nodes = H.simple_nodes(rng.choice([4000, 8000, 16000], n), rng.choice([8, 16, 32], n) * (1 << 30), np.full(n, 110),
                           req_mcpu=rng.integers(0, 2000, n), req_mem=rng.integers(0, 4, n) * (1 << 30), pod_count=rng.integers(0, 20, n),
                           label_cols=[rng.integers(1, zones + 1, n), np.arange(1, n + 1)])
    nodes.nz_mcpu, nodes.nz_mem = nodes.req[0].copy(), nodes.req[1].copy()
    pod = H.simple_pod(500, 1 << 30)
    pod.spread = [M.SpreadConstraint(col=0, max_skew=1, min_domains=1, hard=True, self_match=True, n_domains=zones)]
    pod.ipa = M.InterPodAffinity(key_cols=[1], key_ndom=[n], anti_keys=[0], anti_self=[True], anti_existing=[None], exist_anti=[None],
                                 score_existing=[None], score_self=[0], self_entries=[0])
    prof = M.Profile.default()
    ref = ccref.run(prof, nodes, pod)
    log, stop, scans, stats = CoupledWindowModel(prof, nodes.copy(), pod, ccref.go_log, window=64).run()
    assert log == ref.log.tolist() and stop == "Unschedulable"
    assert ref.placed > 200 and stats["classes_max"] <= 2 * zones and scans <= ref.placed // 64 + 3, (ref.placed, scans, stats)


def test_windows_save_scans(ccref):
    """The point of the exercise: scans per placement.  Over the random cases a 64-cycle window needs a fraction of the scans of
    the one-scan-per-placement loop (windows end early only when the assumed maxima move or nothing known is feasible)."""
    placed = scans = cuts = 0
    for seed in range(40):
        rng = np.random.default_rng(7100 + seed)
        nodes, pod, prof = coupled_case(rng, int(rng.integers(12, 160)))
        log, stop, s, stats = CoupledWindowModel(prof, nodes.copy(), pod, ccref.go_log, window=64).run(4000)
        placed, scans, cuts = placed + len(log), scans + s, cuts + stats["cut_by_maxima"]
    assert placed > 1000 and scans * 8 < placed, (placed, scans, cuts)


# ---- round 5: sweeps (csrc/ccsim_coupled.h `sweep`): the rule that predicts a round of placements, audited against the cycles --------
@pytest.mark.parametrize("seed", range(40))
def test_sweep_predictions_are_the_cycles_the_loop_runs(ccref, seed):
    """tests/coupled_model.py sweep_predict restates the kernel's rule; run(audit_sweeps=True) asserts, cycle by cycle, that every
    predicted winner is the winner the loop then picks and that no stop test of the loop fires inside a predicted stretch -- on the
    adversarial generator of tests/test_coupled.py::test_sweeps_random (unequal domain counts, maxSkew 1 ... 3, minDomains, nodes that
    do not count, few holders of the normalization maxima, winners that stay / leave, limits inside a round), log == oracle's."""
    from test_coupled import sweep_case
    rng = np.random.default_rng(8800 + seed)
    nodes, pod, prof = sweep_case(rng, int(rng.integers(40, 500)))
    limit = int(rng.choice([0, 0, 37, 333]))
    window, list_len = [(4096, 64), (64, 16), (300, 5)][seed % 3]
    ref = ccref.run(prof, nodes, pod, max_limit=limit)
    log, stop, scans, stats = CoupledWindowModel(prof, nodes.copy(), pod, ccref.go_log, window=window, device_plan=True, list_len=list_len).run(limit, audit_sweeps=True)
    assert log == ref.log.tolist(), (seed, window, list_len)


def test_sweep_predictions_cover_a_good_part_of_the_adversarial_cases(ccref):
    """(Guard against a rule that never fires: over the generator's cases a third of the placements lie inside predicted stretches.)"""
    from test_coupled import sweep_case
    swept = placed = 0
    for seed in range(0, 40, 3):
        rng = np.random.default_rng(8800 + seed)
        nodes, pod, prof = sweep_case(rng, int(rng.integers(40, 500)))
        log, stop, scans, stats = CoupledWindowModel(prof, nodes.copy(), pod, ccref.go_log, window=4096, device_plan=True, list_len=64).run(int(rng.choice([0, 0, 37, 333])), audit_sweeps=True)
        swept, placed = swept + stats["swept"], placed + len(log)
    assert placed > 500 and swept > 0.15 * placed, (swept, placed)


def test_sweeps_cover_the_config_5_pod_shape(ccref):
    """On BASELINE config 5's pod shape (zone spread maxSkew 1 + hostname anti-affinity) the rule predicts all but a handful of cycles."""
    rng = np.random.default_rng(5)
    n, zones = 600, 16
    nodes = H.simple_nodes(rng.choice([4000, 8000, 16000], n), rng.choice([8, 16, 32], n) * (1 << 30), np.full(n, 110),
                           req_mcpu=rng.integers(0, 2000, n), req_mem=rng.integers(0, 4, n) * (1 << 30), pod_count=rng.integers(0, 20, n),
                           label_cols=[rng.integers(1, zones + 1, n), np.arange(1, n + 1)])
    nodes.nz_mcpu, nodes.nz_mem = nodes.req[0].copy(), nodes.req[1].copy()
    pod = H.simple_pod(500, 1 << 30)
    pod.spread = [M.SpreadConstraint(col=0, max_skew=1, min_domains=1, hard=True, self_match=True, n_domains=zones)]
    pod.ipa = M.InterPodAffinity(key_cols=[1], key_ndom=[n], anti_keys=[0], anti_self=[True], anti_existing=[None], exist_anti=[None],
                                 score_existing=[None], score_self=[0], self_entries=[0])
    prof = M.Profile.default()
    ref = ccref.run(prof, nodes, pod)
    log, stop, scans, stats = CoupledWindowModel(prof, nodes.copy(), pod, ccref.go_log, window=1024, device_plan=True, list_len=64).run(0, audit_sweeps=True)
    assert log == ref.log.tolist() and stop == "Unschedulable"
    assert stats["swept"] >= 0.9 * len(log), (stats, len(log))


@pytest.mark.parametrize("world", [2, 3, 5])
@pytest.mark.parametrize("seed", range(24))
def test_window_records_of_node_range_shards_unify_to_the_same_windows(ccref, seed, world):
    """Round 5 (windows on shards): per-range class records -- members, list entries, maxima and their holders -- merged as
    k_cw_xunify merges them give the windows of the unsharded pass: same log as the oracle, same number of node passes."""
    rng = np.random.default_rng(3300 + seed)
    nodes, pod, prof = coupled_case(rng, int(rng.integers(5, 400)), roomy=bool(seed % 2))
    limit = int(rng.choice([0, 0, 61, 400]))
    ref = ccref.run(prof, nodes, pod, max_limit=limit or 3000)
    one = CoupledWindowModel(prof, nodes.copy(), pod, ccref.go_log, window=64, device_plan=True, list_len=8).run(limit or 3000)
    log, stop, scans, stats = ShardedCoupledWindowModel(prof, nodes.copy(), pod, ccref.go_log, window=64, device_plan=True, list_len=8, world=world).run(limit or 3000)
    assert log == ref.log.tolist() and log == one[0] and stop == one[1] and scans == one[2]
