"""Parity of the HIP path (through the C ABI) against the CPU oracle.  Bit-exact: integer work."""
import numpy as np
import pytest

import helpers as H
from cluster_capacity_amd import capi, model as M, report as R, synth

pytestmark = pytest.mark.gpu
MODES = ["sequential", "batched"]


def _engine(nodes, pod, prof, **kw):
    e = capi.Engine(device=0, **kw)
    e.load(nodes, pod, prof)
    return e


def _assert_same(got, ref, nodes, pod, check_log=True):
    assert got.placed == ref.placed
    assert got.stop == ref.stop
    assert np.array_equal(got.per_node_count, ref.per_node_count)
    if check_log and ref.log is not None and got.log is not None:
        assert np.array_equal(got.log, ref.log)
    if ref.stop == M.STOP_UNSCHEDULABLE:
        assert np.array_equal(got.hist, ref.hist)
        assert np.array_equal(got.hist_taintset[: len(ref.hist_taintset)], ref.hist_taintset)
        assert got.n_code_unschedulable == ref.n_code_unschedulable
        assert R.stop_reason(got, nodes.n, 0) == R.stop_reason(ref, nodes.n, 0)


@pytest.mark.parametrize("mode", MODES)
def test_ka1_test_prediction(ccref, mode):
    nodes, pod, prof = H.test_prediction_nodes(), H.test_prediction_pod(), M.Profile.default()
    e = _engine(nodes, pod, prof)
    for limit in (0, 6):
        got = e.run(max_limit=limit, mode=mode)
        ref = ccref.run(prof, nodes, pod, max_limit=limit)
        _assert_same(got, ref, nodes, pod)
        e.load(nodes, pod, prof)
    got = _engine(nodes, pod, prof).run(mode=mode)
    assert R.stop_reason(got, 3, 0) == ("Unschedulable: 0/3 nodes are available: 1 Insufficient cpu, 3 Too many pods. "
                                       "preemption: 0/3 nodes are available: 3 No preemption victims found for incoming pod.")


@pytest.mark.parametrize("mode", MODES)
def test_ka2_readme(ccref, mode):
    nodes, pod, prof = H.readme_nodes(4), H.examples_pod(), M.Profile.default()
    got = _engine(nodes, pod, prof).run(mode=mode)
    assert got.placed == 52 and got.per_node_count.tolist() == [13] * 4
    _assert_same(got, ccref.run(prof, nodes, pod), nodes, pod)


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("cfg,n,limit", [("C2", 1000, 0), ("C3", 1000, 0), ("C3", 4096, 700), ("C2", 5000, 300),
                                          ("C3", 777, 0), ("C3", 1, 0), ("C3", 513, 50)])
def test_synthetic_vs_oracle(ccref, mode, cfg, n, limit):
    nodes, pod, prof = synth.make_config(cfg, n_nodes=n, seed=1234 + n)
    ref = ccref.run(prof, nodes, pod, max_limit=limit)
    e = _engine(nodes, pod, prof)
    got = e.run(max_limit=limit, mode=mode)
    _assert_same(got, ref, nodes, pod)
    # final dynamic state == initial + count * pod request (NodeInfo.update)
    st = e.read_state()
    cnt = got.per_node_count.astype(np.int64)
    assert np.array_equal(st["req_mcpu"], nodes.req[0] + cnt * int(pod.req[0]))
    assert np.array_equal(st["req_mem"], nodes.req[1] + cnt * int(pod.req[1]))
    assert np.array_equal(st["nz_mcpu"], nodes.nz_mcpu + cnt * pod.nz_mcpu)
    assert np.array_equal(st["pod_count"], nodes.pod_count + got.per_node_count)


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("seed", range(12))
def test_random_plugin_mix_vs_oracle(ccref, mode, seed):
    rng = np.random.default_rng(seed)
    nodes, pod, prof = H.random_case(rng, int(rng.integers(1, 1500)))
    limit = int(rng.choice([0, 0, 37, 500]))
    ref = ccref.run(prof, nodes, pod, max_limit=limit)
    got = _engine(nodes, pod, prof).run(max_limit=limit, mode=mode)
    _assert_same(got, ref, nodes, pod)


def test_schedule_one_matches_oracle_round_by_round(ccref):
    nodes, pod, prof = synth.make_config("C3", n_nodes=600, seed=5)
    ref = ccref.run(prof, nodes, pod, max_limit=200)
    e = _engine(nodes, pod, prof)
    for r in range(200):
        node, evaluated, feasible = e.schedule_one()
        assert node == ref.log[r]
        assert evaluated == nodes.n and feasible > 0


def test_schedule_one_fit_error_and_empty_snapshot(ccref):
    nodes, pod, prof = H.readme_nodes(2), H.examples_pod(), M.Profile.default()
    e = _engine(nodes, pod, prof)
    seen = [e.schedule_one()[0] for _ in range(27)]
    assert seen[:26].count(0) == 13 and seen[:26].count(1) == 13 and seen[26] == -1
    empty = H.simple_nodes([], [], [])
    got = _engine(empty, pod, prof).run()
    assert got.placed == 0 and got.stop == M.STOP_NO_NODES
    assert R.stop_reason(got, 0, 0) == "Unschedulable: no nodes available to schedule pods"


def test_eager_and_graph_launch_agree(ccref):
    nodes, pod, prof = synth.make_config("C3", n_nodes=3000, seed=9)
    a = _engine(nodes, pod, prof, use_graph=True, rounds_per_sync=64).run(max_limit=500)
    b = _engine(nodes, pod, prof, use_graph=False, rounds_per_sync=7).run(max_limit=500)
    assert np.array_equal(a.log, b.log) and a.placed == b.placed == 500


def _numpy_total_scores(nodes, pod, prof):
    """TotalScore of every node of the initial snapshot and the feasibility vector, in numpy, for pods without topology-coupled plugins
    and without required node affinity (C3 / C4): NodeUnschedulable + TaintToleration + NodeResourcesFit filters (fit.go:564-615);
    TaintToleration (reverse-normalized, taint_toleration.go:169-199), NodeAffinity preferred terms (normalized, node_affinity.go:241-290),
    LeastAllocated (least_allocated.go:30-61), BalancedAllocation (balanced_allocation.go:146-180), weighted (framework.go:1214-1238)."""
    assert not pod.spread and pod.ipa is None and not pod.has_required_terms and not pod.has_node_selector and pod.image_score is None
    n = nodes.n
    feasible = (nodes.unschedulable == 0) | bool(pod.tolerates_unschedulable)
    feasible &= np.asarray(pod.taint_filter_ok)[nodes.taintset_id] != 0
    feasible &= nodes.pod_count + 1 <= nodes.alloc_pods
    for c in range(3):
        if pod.req[c]:
            feasible &= pod.req[c] <= nodes.alloc[c] - nodes.req[c]
    prefer = np.asarray(pod.taint_prefer_cnt)[nodes.taintset_id].astype(np.int64)
    mx = int(prefer[feasible].max())
    taint = np.full(n, 100, np.int64) if mx == 0 else 100 - 100 * prefer // mx
    aff = np.zeros(n, np.int64)
    for w, term in pod.preferred:
        m = np.ones(n, bool)
        for col, table in term:
            m &= np.asarray(table)[nodes.label_cols[col]] != 0
        aff += w * m
    mx = int(aff[feasible].max())
    aff = aff if mx == 0 else 100 * aff // mx
    nz = [nodes.nz_mcpu + pod.nz_mcpu, nodes.nz_mem + pod.nz_mem]
    least = np.zeros(n, np.int64)
    for c, w in zip(prof.fit_res, prof.fit_res_w):
        assert c < 2
        least += np.where(nz[c] > nodes.alloc[c], 0, (nodes.alloc[c] - nz[c]) * 100 // np.maximum(nodes.alloc[c], 1)) * w
    least //= sum(prof.fit_res_w)
    assert tuple(prof.bal_res) == (0, 1)
    f = [np.minimum(1.0, (nodes.req[c] + pod.req[c]).astype(np.float64) / nodes.alloc[c].astype(np.float64)) for c in (0, 1)]
    bal = ((1.0 - np.abs(f[0] - f[1]) / 2.0) * 100.0).astype(np.int64)
    total = prof.w_taint * taint + prof.w_nodeaffinity * aff + prof.w_fit * least + prof.w_balanced * bal
    return total, feasible


def test_full_size_properties_1m_nodes():
    """BASELINE full size: size-independent properties instead of the (too slow) oracle."""
    nodes, pod, prof = synth.make_config("C4", n_nodes=1_000_000)
    e = _engine(nodes, pod, prof)
    L = 1500
    got = e.run(max_limit=L, mode="sequential")
    assert got.placed == L and got.stop == M.STOP_LIMIT and int(got.per_node_count.sum()) == L
    assert np.array_equal(np.bincount(got.log, minlength=nodes.n).astype(np.int32), got.per_node_count)
    # every winner passed the static filters and never exceeds its Fit capacity
    w = np.unique(got.log)
    assert not nodes.unschedulable[w].any()
    free_cpu = nodes.alloc[0] - nodes.req[0]
    assert (got.per_node_count.astype(np.int64) * 150 <= free_cpu).all()
    assert (got.per_node_count + nodes.pod_count <= nodes.alloc_pods).all()
    # greedy property of cycle 1 (schedule_one.go:894-941 with the canonical tie-break): the first winner holds the maximum TotalScore
    # over the feasible nodes of the INITIAL snapshot, and no feasible node before it in canonical order scores as much -- recomputed here
    # in numpy from the reference's formulas (least_allocated.go:30-61, balanced_allocation.go:146-180, normalize_score.go:28-56,
    # framework.go:1214-1238), independent of the oracle and of the kernels
    total, feasible = _numpy_total_scores(nodes, pod, prof)
    first = int(got.log[0])
    assert feasible[first] and total[first] == total[feasible].max()
    assert not (feasible[:first] & (total[:first] == total[first])).any()
    st = e.read_state()
    assert np.array_equal(st["pod_count"], nodes.pod_count + got.per_node_count)


# ---- multi-shard protocol on ONE GPU: several engines own contiguous ranges of the snapshot and the
# "all-gather" is a device copy; exercises k_final/k_level_final publishing and k_decide/k_level_decide.
class _LocalShards:
    def __init__(self, nodes, pod, prof, world):
        import torch
        from cluster_capacity_amd import dist as ccdist
        self.torch, self.world = torch, world
        self.ts = torch.cuda.Stream(device=0)  # handle 0 (the default stream) would mean "create your own"
        torch.cuda.set_stream(self.ts)
        stream = self.ts.cuda_stream
        assert stream != 0
        self.engines, self.send = [], []
        self.recv = [torch.zeros(capi.XCHG_WORDS * world, dtype=torch.int64, device="cuda:0") for _ in range(world)]
        for r in range(world):
            lo, hi = ccdist.shard_bounds(nodes.n, world, r)
            e = capi.Engine(device=0, stream=stream, use_graph=False)
            e.load(nodes.slice(lo, hi), ccdist.shard_pod(pod, lo, hi), prof, global_offset=lo, n_global=nodes.n)
            self.engines.append(e)
            self.send.append(torch.zeros(capi.XCHG_WORDS, dtype=torch.int64, device="cuda:0"))
        # replicated topology tables: the in-process stand-in for dist.all_reduce over the ranks
        tabs = [ccdist.table_tensors(e, 0) for e in self.engines]
        for j in range(len(tabs[0])):
            stack = torch.stack([t[j][0] for t in tabs])
            red = stack.max(0).values if tabs[0][j][1] == "max" else stack.sum(0)
            for t in tabs:
                t[j][0].copy_(red)
        torch.cuda.synchronize()
        for e in self.engines:
            if tabs[0]:
                e.dist_tables_done()

    def run(self, limit, mode, log_cap, poll_every=1):
        from cluster_capacity_amd import dist as ccdist
        W = self.world
        for r, e in enumerate(self.engines):
            e.dist_begin(limit, mode, W, r, self.send[r].data_ptr(), self.recv[r].data_ptr(), log_cap)
        # one template, zone spread + hostname anti-affinity: windows of placements per exchange where every rank can (round 5)
        self.cw_windows = 0
        go = all([e.dist_cw_eligible() for e in self.engines])
        for e in self.engines:
            e.dist_cw_enable(go)
        if go:
            bufs = [e.dist_cw_buffers() for e in self.engines]
            nbytes = bufs[0][2]
            assert nbytes % 8 == 0 and all(b[2] == nbytes for b in bufs)
            snd = [self.torch.as_tensor(ccdist._DevArray(b[0], nbytes // 8, 8), device="cuda:0") for b in bufs]
            rcv = [self.torch.as_tensor(ccdist._DevArray(b[1], W * nbytes // 8, 8), device="cuda:0") for b in bufs]
            for _ in range(1_000_000):
                for e in self.engines:
                    e.dist_cw_scan()
                gathered = self.torch.cat(snd)
                for r in range(W):
                    rcv[r].copy_(gathered)
                for e in self.engines:
                    e.dist_cw_decide()
                self.cw_windows += 1
                done = [e.dist_poll()[0] for e in self.engines]
                fell = [e.coupled_info()["fell_back"] for e in self.engines]
                assert len(set(done)) == 1 and len(set(fell)) == 1  # (replicated decisions)
                if done[0] or fell[0]:
                    if fell[0]:
                        self.cw_windows -= 1
                    break
            if all([e.dist_poll()[0] for e in self.engines]):
                res = [e.dist_finish(log_cap > 0, log_cap) for e in self.engines]
                return res, (ccdist.merge_logs([r.log for r in res]) if log_cap > 0 else None)
        for _ in range(1_000_000):
            for _ in range(poll_every):  # (the batched mode launches its full pass only right after a poll)
                for e in self.engines:
                    e.dist_scan()
                gathered = self.torch.cat(self.send)
                for r in range(W):
                    self.recv[r].copy_(gathered)
                for e in self.engines:
                    e.dist_decide()
            if all([e.dist_poll()[0] for e in self.engines]):  # every rank polls (no short-circuit): same cadence everywhere
                break
        res = [e.dist_finish(log_cap > 0, log_cap) for e in self.engines]
        return res, (ccdist.merge_logs([r.log for r in res]) if log_cap > 0 else None)


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("world,cfg,n,limit,poll_every", [(2, "C3", 1500, 0, 1), (3, "C3", 1100, 450, 1), (2, "C2", 700, 0, 1),
                                                          (2, "C3", 1500, 0, 8), (3, "C3", 1100, 450, 5), (4, "C3", 2100, 0, 32)])
def test_sharded_protocol_matches_oracle(ccref, mode, world, cfg, n, limit, poll_every):
    nodes, pod, prof = synth.make_config(cfg, n_nodes=n, seed=77 + n)
    ref = ccref.run(prof, nodes, pod, max_limit=limit)
    if mode == "sequential" and ref.placed > 3000:
        limit = 3000
        ref = ccref.run(prof, nodes, pod, max_limit=limit)
    res, log = _LocalShards(nodes, pod, prof, world).run(limit, mode, max(1, ref.placed), poll_every)
    assert all(r.placed == ref.placed and r.stop == ref.stop for r in res)
    assert np.array_equal(np.concatenate([r.per_node_count for r in res]), ref.per_node_count)
    assert np.array_equal(log[: ref.placed], ref.log)
    if ref.stop == M.STOP_UNSCHEDULABLE:
        assert np.array_equal(sum(r.hist for r in res), ref.hist)


@pytest.mark.parametrize("kb", ["1", "3", "64"])
@pytest.mark.parametrize("world,cfg,n,limit,poll_every", [(2, "C3", 1500, 0, 1), (3, "C3", 1100, 450, 5), (2, "C2", 700, 0, 1), (4, "C3", 2100, 0, 32),
                                                          (3, "C3", 4000, 2777, 8), (2, "C3", 2500, 0, 32)])
def test_sharded_blind_level_batches_match_oracle(ccref, monkeypatch, kb, world, cfg, n, limit, poll_every):
    """No placement log: the sharded batched mode commits up to CCSIM_LEVEL_BATCH score levels per exchange, blindly, and validates
    afterwards (a normalization maximum out of holders, --max-limit crossed: roll back on every rank, half the levels, down to the
    ordered one-level commit).  Totals, per-node counts and the terminal histogram are the observables."""
    monkeypatch.setenv("CCSIM_LEVEL_BATCH", kb)
    nodes, pod, prof = synth.make_config(cfg, n_nodes=n, seed=77 + n)
    ref = ccref.run(prof, nodes, pod, max_limit=limit)
    sh = _LocalShards(nodes, pod, prof, world)
    res, _ = sh.run(limit, "batched", 0, poll_every)
    assert all(r.placed == ref.placed and r.stop == ref.stop for r in res)
    assert np.array_equal(np.concatenate([r.per_node_count for r in res]), ref.per_node_count)
    if ref.stop == M.STOP_UNSCHEDULABLE:
        assert np.array_equal(sum(r.hist for r in res), ref.hist)
    if kb == "64" and limit == 0:
        assert res[0].scans * 4 < _LocalShards(nodes, pod, prof, world).run(limit, "batched", max(1, ref.placed), poll_every)[0][0].scans  # far fewer exchanges than levels


@pytest.mark.parametrize("world,seed", [(2, 0), (3, 1), (2, 2), (4, 3)])
def test_sharded_blind_level_batches_random_plugin_mix(ccref, world, seed):
    """Taints / preferred affinity with few holders: the roll-back cases, sharded."""
    rng = np.random.default_rng(7600 + seed)
    nodes, pod, prof = H.random_case(rng, int(rng.integers(200, 3000)))
    prof.filter_mask |= M.F_FIT
    for limit in (0, int(rng.choice([37, 500]))):
        ref = ccref.run(prof, nodes, pod, max_limit=limit)
        res, _ = _LocalShards(nodes, pod, prof, world).run(limit, "batched", 0, 4)
        assert all(r.placed == ref.placed and r.stop == ref.stop for r in res)
        assert np.array_equal(np.concatenate([r.per_node_count for r in res]), ref.per_node_count)
        if ref.stop == M.STOP_UNSCHEDULABLE:
            assert np.array_equal(sum(r.hist for r in res), ref.hist)


def test_full_size_batched_equals_sequential_1m_nodes():
    """BASELINE full size: the two engine modes must produce the same placement sequence (the oracle is too slow
    here; its equality with the sequential mode is established at the sizes above)."""
    nodes, pod, prof = synth.make_config("C4", n_nodes=1_000_000)
    L = 20_000
    e = _engine(nodes, pod, prof)
    seq = e.run(max_limit=L, mode="sequential", log_cap=L)
    e.reset_state()
    bat = e.run(max_limit=L, mode="batched", log_cap=L)
    assert seq.placed == bat.placed == L and seq.stop == bat.stop == M.STOP_LIMIT
    assert np.array_equal(seq.log, bat.log)
    assert np.array_equal(seq.per_node_count, bat.per_node_count)
    assert bat.scans < seq.scans / 50  # levels, not cycles
    # exhaustive batched run: conservation and capacity properties at full size
    e.reset_state()
    full = e.run(max_limit=0, mode="batched", want_log=False)
    assert full.stop == M.STOP_UNSCHEDULABLE and int(full.per_node_count.astype(np.int64).sum()) == full.placed
    cnt = full.per_node_count.astype(np.int64)
    assert (cnt * 150 <= nodes.alloc[0] - nodes.req[0]).all() and (cnt + nodes.pod_count <= nodes.alloc_pods).all()
    ok = (nodes.unschedulable == 0) & (np.isin(nodes.taintset_id, [0, 1, 2, 3]))
    # every node that can still take one more pod must be statically infeasible (none here: pod tolerates infra)
    room = (cnt + 1) * 150 <= nodes.alloc[0] - nodes.req[0]
    room &= (cnt + 1) * (100 << 20) <= nodes.alloc[1] - nodes.req[1]
    room &= cnt + nodes.pod_count + 1 <= nodes.alloc_pods
    assert not (room & ok).any()
    assert full.hist[M.R_UNSCHEDULABLE] == int(nodes.unschedulable.sum())


@pytest.mark.parametrize("world,seed", [(2, 0), (3, 1), (2, 2), (4, 3), (2, 4), (3, 5)])
def test_sharded_protocol_with_topology_coupled_plugins(ccref, world, seed):
    """Hard spread constraints + inter-pod affinity across shards: replicated count tables (all-reduced after
    set_pod), global verification of the assumed minimum / min-max, the winner's domain ids travelling in the record."""
    rng = np.random.default_rng(300 + seed)
    nodes, pod, prof = H.random_case(rng, int(rng.integers(50, 900)))
    pod.spread = H.random_spread(rng, nodes, n_constraints=int(rng.integers(0, 3)))
    if seed % 2 == 0 or not pod.spread:
        pod.ipa = H.random_ipa(rng, nodes)
    limit = int(rng.choice([0, 60, 150]))
    ref = ccref.run(prof, nodes, pod, max_limit=limit)
    if ref.placed > 1500:
        limit = 1500
        ref = ccref.run(prof, nodes, pod, max_limit=limit)
    res, log = _LocalShards(nodes, pod, prof, world).run(limit, "sequential", max(1, ref.placed))
    assert all(r.placed == ref.placed and r.stop == ref.stop for r in res)
    assert np.array_equal(np.concatenate([r.per_node_count for r in res]), ref.per_node_count)
    assert np.array_equal(log[: ref.placed], ref.log)
    if ref.stop == M.STOP_UNSCHEDULABLE:
        assert np.array_equal(sum(r.hist for r in res), ref.hist)
        assert sum(r.n_code_unschedulable for r in res) == ref.n_code_unschedulable


def _sharded_vs_oracle(ccref, nodes, pod, prof, limit, world, cap=1500):
    ref = ccref.run(prof, nodes, pod, max_limit=limit)
    if ref.placed > cap:
        limit = cap
        ref = ccref.run(prof, nodes, pod, max_limit=limit)
    e_nodes, e_pod = M.relax_soft(nodes, pod)  # (requireAllTopologies = false: the engine form, derived on the WHOLE snapshot before it is sharded)
    res, log = _LocalShards(e_nodes, e_pod, prof, world).run(limit, "sequential", max(1, ref.placed))
    assert all(r.placed == ref.placed and r.stop == ref.stop for r in res), ([(r.placed, r.stop) for r in res], ref.placed, ref.stop)
    n = min(len(log), len(ref.log))
    first = next((i for i in range(n) if log[i] != ref.log[i]), None)
    assert first is None, ("first differing placement", first, log[max(0, first - 2): first + 3].tolist(), ref.log[max(0, first - 2): first + 3].tolist())
    assert np.array_equal(np.concatenate([r.per_node_count for r in res]), ref.per_node_count)
    if ref.stop == M.STOP_UNSCHEDULABLE:
        assert np.array_equal(sum(r.hist for r in res), ref.hist)


@pytest.mark.parametrize("world,seed", [(1, 0), (2, 1), (3, 2), (2, 3), (4, 4), (2, 5), (3, 6), (5, 7), (2, 8), (3, 9), (2, 10), (4, 11)])
def test_sharded_protocol_with_schedule_anyway_constraints(ccref, world, seed):
    """ScheduleAnyway spread constraints (and hard ones, and inter-pod terms beside them) across shards: the candidate-domain SETS
    travel as bitmaps in the exchange record, the counts of feasible non-ignored nodes add, the raw-score ranges combine; the
    decision verifies the assumed log(size + 2) weights and the assumed normalization range on the gathered records and rescans
    when they moved; every rank adds the winner's clone to its replicated per-domain tables (tests/sharded_coupled_model.py)."""
    rng = np.random.default_rng(700 + seed)  # (the cases of tests/test_spread.py::test_gpu_soft_and_hard_spread_random)
    nodes, pod, prof = H.random_case(rng, int(rng.integers(20, 900)))
    cons = H.random_spread(rng, nodes, n_constraints=2)
    for k in cons:
        k.hard = bool(rng.integers(0, 2))
    if seed % 4 == 0:
        cons[0].is_hostname, cons[0].hard = True, False  # scored per node instead of per domain
    if not any(not k.hard for k in cons):
        cons[-1].hard = False
    pod.spread = cons
    if seed % 3 == 2:
        pod.ipa = H.random_ipa(rng, nodes)
    _sharded_vs_oracle(ccref, nodes, pod, prof, int(rng.choice([0, 0, 70])), world)


@pytest.mark.parametrize("world,seed", [(2, 0), (3, 1), (4, 2), (2, 3)])
def test_sharded_system_default_spreading(ccref, world, seed):
    """The plugin's system default constraints (requireAllTopologies = false, scoring.go:140) on clusters where nodes lack the zone key,
    across shards: the extra value id that stands for the missing key is a domain like any other in the records' bitmaps and sizes."""
    from test_spread import _relaxed_case
    rng = np.random.default_rng(5100 + seed)  # (the cases of tests/test_spread.py::test_gpu_system_default_spreading_random)
    nodes, pod, prof = _relaxed_case(rng, int(rng.integers(40, 700)))
    _sharded_vs_oracle(ccref, nodes, pod, prof, int(rng.choice([0, 0, 90])), world)


@pytest.mark.parametrize("world,n,limit,hostname", [(2, 500, 300, False), (3, 1500, 0, False), (4, 700, 400, True)])
def test_sharded_soft_spread_synthetic(ccref, world, n, limit, hostname):
    nodes, pod, prof = synth.make_config("C3", n_nodes=n, seed=40 + n)  # (tests/test_spread.py::test_gpu_soft_spread_synthetic)
    pod.spread = [M.SpreadConstraint(col=1, max_skew=2, hard=False, self_match=True, n_domains=synth.zones_for(n))]
    if hostname:
        nodes.label_cols.append(np.arange(1, n + 1, dtype=np.int32))
        pod.spread.append(M.SpreadConstraint(col=2, max_skew=1, hard=False, self_match=True, is_hostname=True, n_domains=n))
    _sharded_vs_oracle(ccref, nodes, pod, prof, limit, world)


def test_sharded_soft_spread_beyond_the_record_is_refused():
    n = 800
    nodes, pod, prof = synth.make_config("C3", n_nodes=n, seed=9)
    nodes.label_cols.append((np.arange(n, dtype=np.int32) % 400) + 1)  # 400 domains: more than the record's bitmap carries
    pod.spread = [M.SpreadConstraint(col=2, max_skew=1, hard=False, self_match=True, n_domains=400)]
    with pytest.raises(capi.CcsimError):
        _LocalShards(nodes, pod, prof, 2).run(20, "sequential", 20)


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("narrow", ["0", "1"])
def test_wide_and_narrow_column_paths_agree_with_oracle(ccref, monkeypatch, mode, narrow):
    """The full-pass kernels have two storage / arithmetic paths (int64 columns + fp64, int32 mirrors + f32 estimates);
    both must reproduce the oracle.  CCSIM_NARROW=0 forces the wide path; odd byte counts force it naturally."""
    monkeypatch.setenv("CCSIM_NARROW", narrow)
    for cfg, n, seed, limit in (("C3", 3000, 91, 0), ("C2", 2000, 92, 1500)):
        nodes, pod, prof = synth.make_config(cfg, n_nodes=n, seed=seed)
        ref = ccref.run(prof, nodes, pod, max_limit=limit)
        _assert_same(_engine(nodes, pod, prof).run(max_limit=limit, mode=mode), ref, nodes, pod)
    # memory values with no common power-of-two unit and > 2^30 after the shift: the wide path even with CCSIM_NARROW=1
    rng = np.random.default_rng(5)
    n = 800
    nodes = H.simple_nodes(rng.choice([4000, 8000, 64000], n), rng.integers(1 << 34, 1 << 38, n) | 1, np.full(n, 40),
                           req_mcpu=rng.integers(0, 2000, n), req_mem=rng.integers(0, 1 << 33, n))
    pod = H.simple_pod(137, (1 << 28) + 12345)
    ref = ccref.run(M.Profile.default(), nodes, pod, max_limit=0)
    _assert_same(_engine(nodes, pod, M.Profile.default()).run(mode=mode), ref, nodes, pod)


def _scalar_case(rng, n, n_scalar):
    """Snapshots with extended resources (columns 3+): exercises the NX > 0 kernel variants and their reasons."""
    base = H.simple_nodes(rng.choice([4000, 8000], n), rng.choice([8, 16], n) * H.GiB, rng.integers(5, 30, n),
                          req_mcpu=rng.integers(0, 1000, n), alloc_eph=rng.choice([0, 20, 100], n) * H.GiB)
    alloc = list(base.alloc) + [rng.integers(0, 9, n).astype(np.int64) for _ in range(n_scalar)]
    req = list(base.req) + [np.minimum(rng.integers(0, 3, n), alloc[3 + k]).astype(np.int64) for k in range(n_scalar)]
    nodes = M.NodesSoA(alloc=alloc, alloc_pods=base.alloc_pods, req=req, nz_mcpu=base.nz_mcpu, nz_mem=base.nz_mem,
                       pod_count=base.pod_count, taintset_id=base.taintset_id, unschedulable=base.unschedulable,
                       scalar_names=[f"example.com/dev{k}" for k in range(n_scalar)])
    preq = [int(rng.choice([100, 250])), 256 * H.MiB, int(rng.choice([0, 1])) * H.GiB] + [int(rng.integers(0, 3)) for _ in range(n_scalar)]
    pod = M.PodSpec(req=np.array(preq, np.int64), nz_mcpu=preq[0], nz_mem=preq[1], has_scalar_entries=True)
    return nodes, pod


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("n_scalar,seed", [(1, 0), (2, 1), (3, 2), (8, 3)])
def test_scalar_resources_vs_oracle(ccref, mode, n_scalar, seed):
    rng = np.random.default_rng(40 + seed)
    nodes, pod = _scalar_case(rng, int(rng.integers(300, 1500)), n_scalar)
    prof = M.Profile.default()
    limit = int(rng.choice([0, 0, 200]))
    ref = ccref.run(prof, nodes, pod, max_limit=limit)
    got = _engine(nodes, pod, prof).run(max_limit=limit, mode=mode)
    _assert_same(got, ref, nodes, pod)
    if ref.stop == M.STOP_UNSCHEDULABLE:
        msg = R.stop_reason(got, nodes.n, 0, scalar_names=nodes.scalar_names)
        assert msg == R.stop_reason(ref, nodes.n, 0, scalar_names=nodes.scalar_names)


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("prof", [
    M.Profile(fit_res=(0,), fit_res_w=(3,), bal_res=(0, 1)),                       # LeastAllocated on cpu only
    M.Profile(fit_res=(1,), fit_res_w=(1,), bal_res=(1,)),                         # one Balanced resource: std = 0
    M.Profile(fit_res=(0, 1), fit_res_w=(2, 5), bal_res=(0,), w_balanced=3),       # uneven weights (weight sum 7)
    M.Profile(filter_mask=M.F_FIT | M.F_TAINT, w_nodeaffinity=0, w_taint=1),       # filters partly disabled
    M.Profile(filter_mask=M.F_FIT, w_taint=0, w_nodeaffinity=0, w_fit=0, w_balanced=1, w_topologyspread=0, w_interpodaffinity=0),
], ids=["cpu-only", "one-balanced", "weights-2-5", "mask-fit-taint", "balanced-only"])
def test_profile_variants_vs_oracle(ccref, mode, prof):
    nodes, pod, _ = synth.make_config("C3", n_nodes=1500, seed=314)
    for limit in (0, 333):
        ref = ccref.run(prof, nodes, pod, max_limit=limit)
        _assert_same(_engine(nodes, pod, prof).run(max_limit=limit, mode=mode), ref, nodes, pod)


@pytest.mark.parametrize("narrow", ["0", "1"])
def test_modes_interleave_on_one_engine(ccref, monkeypatch, narrow):
    """One engine, one snapshot, runs continued in alternating modes: the batched mode keeps its state in commit rows and a
    score cache, the sequential mode and ccsim_schedule_one in the columns -- every hand-over must be lossless."""
    monkeypatch.setenv("CCSIM_NARROW", narrow)
    nodes, pod, prof = synth.make_config("C3", n_nodes=3000, seed=31)
    ref = ccref.run(prof, nodes, pod, max_limit=0)
    e = _engine(nodes, pod, prof)
    logs = []
    a = e.run(max_limit=700, mode="batched", log_cap=700)
    logs.append(a.log)
    b = e.run(max_limit=300, mode="sequential", log_cap=300)
    logs.append(b.log)
    one = [e.schedule_one()[0] for _ in range(50)]
    logs.append(np.array(one, np.int32))
    c = e.run(max_limit=2000, mode="batched", log_cap=2000)
    logs.append(c.log)
    d = e.run(max_limit=0, mode="batched", log_cap=max(1, ref.placed))
    logs.append(d.log)
    got = np.concatenate(logs)
    assert a.placed == 700 and b.placed == 300 and c.placed == 2000
    assert len(got) == ref.placed and np.array_equal(got, ref.log)
    assert d.stop == ref.stop and np.array_equal(d.hist, ref.hist)
    st = e.read_state()
    cnt = np.bincount(ref.log, minlength=nodes.n).astype(np.int64)
    assert np.array_equal(st["req_mcpu"], nodes.req[0] + cnt * int(pod.req[0]))
    assert np.array_equal(st["pod_count"], nodes.pod_count + cnt)


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("fit_res,fit_w,bal_res,n_scalar", [
    ((0, 1, 2), (1, 1, 1), (0, 1, 2), 0),        # ephemeral-storage in both lists: three fractions -> math.Sqrt branch
    ((0, 2), (3, 2), (1, 2), 0),                 # two resources, one of them ephemeral-storage
    ((0, 1, 3, 4), (1, 1, 2, 1), (0, 1, 3, 4), 2),  # scalar resources: requested ones take part, others are bypassed
    ((3,), (1,), (0, 1, 2, 3), 1),               # LeastAllocated on a scalar only; four fractions
], ids=["eph-3way", "eph-2way", "scalars", "scalar-only-fit"])
def test_scoring_resource_lists_beyond_cpu_and_memory(ccref, mode, fit_res, fit_w, bal_res, n_scalar):
    """resource_allocation.go:97-110: ephemeral-storage and requested scalar resources take part in LeastAllocated and in
    BalancedAllocation (population standard deviation of > 2 fractions, balanced_allocation.go:168-174)."""
    rng = np.random.default_rng(sum(fit_res) * 31 + n_scalar)
    for seed in range(3):
        n = int(rng.integers(200, 1200))
        if n_scalar:
            nodes, pod = _scalar_case(rng, n, n_scalar)
            if seed == 1:
                pod.req[3] = 0  # a scalar in the list that THIS pod does not request: bypassed
        else:
            base = H.simple_nodes(rng.choice([4000, 8000, 16000], n), rng.choice([8, 16, 32], n) * H.GiB, rng.integers(5, 40, n),
                                  req_mcpu=rng.integers(0, 20, n) * 100, req_mem=rng.integers(0, 8, n) * H.GiB // 2,
                                  alloc_eph=rng.choice([0, 20, 50, 100], n) * H.GiB)
            base.req[2] = (rng.integers(0, 10, n) * H.GiB).astype(np.int64)
            nodes = base
            pod = H.simple_pod(int(rng.choice([100, 250, 500])), int(rng.choice([128, 512])) * H.MiB, eph=int(rng.choice([0, 1, 2])) * H.GiB)
        prof = M.Profile(fit_res=fit_res, fit_res_w=fit_w, bal_res=bal_res, w_balanced=2)
        limit = int(rng.choice([0, 150]))
        ref = ccref.run(prof, nodes, pod, max_limit=limit)
        got = _engine(nodes, pod, prof).run(max_limit=limit, mode=mode)
        _assert_same(got, ref, nodes, pod)
