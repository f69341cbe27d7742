"""The persistent form of the batched mode (csrc/ccsim_persist.h: one launch, node state resident in LDS, grid-wide
reduce + barrier per score level) against the oracle and against the multi-kernel batched mode (CCSIM_PERSIST=0).

The rest of the GPU suite runs the batched mode with its default (persistent where the snapshot qualifies) and mostly
WITH a placement log, i.e. on the ordered path; here the blind fast path (no log), its roll-back cases (limit crossed
inside a level, a normalization maximum losing its last feasible holder) and the fallback conditions are pinned."""
import numpy as np
import pytest

import helpers as H
from cluster_capacity_amd import capi, model as M, synth

pytestmark = pytest.mark.gpu


def _run(nodes, pod, prof, limit, want_log, **kw):
    e = capi.Engine(device=0, **kw)
    e.load(nodes, pod, prof)
    r = e.run(max_limit=limit, mode="batched", want_log=want_log, log_cap=None if want_log else 0)
    st = e.read_state()
    e.close()
    return r, st


def _same(got, ref, check_log):
    assert got.placed == ref.placed and got.stop == ref.stop
    assert np.array_equal(got.per_node_count, ref.per_node_count)
    if check_log:
        assert np.array_equal(got.log, ref.log)
    if ref.stop == M.STOP_UNSCHEDULABLE:
        assert np.array_equal(got.hist, ref.hist) and got.n_code_unschedulable == ref.n_code_unschedulable


@pytest.mark.parametrize("persist", ["1", "0"])
@pytest.mark.parametrize("cfg,n,limit", [("C3", 4096, 0), ("C3", 4096, 700), ("C2", 5000, 300), ("C3", 777, 0), ("C3", 513, 50),
                                          ("C3", 20_000, 12_345), ("C2", 3000, 0), ("C3", 1, 0)])
def test_fast_path_without_log_vs_oracle(ccref, monkeypatch, persist, cfg, n, limit):
    monkeypatch.setenv("CCSIM_PERSIST", persist)
    nodes, pod, prof = synth.make_config(cfg, n_nodes=n, seed=4321 + n)
    ref = ccref.run(prof, nodes, pod, max_limit=limit, threads=8)
    got, st = _run(nodes, pod, prof, limit, want_log=False)
    _same(got, ref, check_log=False)
    cnt = ref.per_node_count.astype(np.int64)
    assert np.array_equal(st["req_mcpu"], nodes.req[0] + cnt * int(pod.req[0]))
    assert np.array_equal(st["req_mem"], nodes.req[1] + cnt * int(pod.req[1]))
    assert np.array_equal(st["nz_mcpu"], nodes.nz_mcpu + cnt * pod.nz_mcpu)
    assert np.array_equal(st["pod_count"], nodes.pod_count + ref.per_node_count)
    got, _ = _run(nodes, pod, prof, limit, want_log=True)  # ordered path
    _same(got, ref, check_log=True)


@pytest.mark.parametrize("seed", range(24))
def test_random_plugin_mix_fast_and_ordered_paths(ccref, seed):
    """Taints / preferred affinity with few holders: normalization maxima lose their last feasible holder inside a level
    (roll-back + ordered redo with a cut), limits fall inside levels."""
    rng = np.random.default_rng(7000 + seed)
    nodes, pod, prof = H.random_case(rng, int(rng.integers(1, 3000)))
    limit = int(rng.choice([0, 0, 37, 500]))
    ref = ccref.run(prof, nodes, pod, max_limit=limit)
    for want_log in (False, True):
        got, _ = _run(nodes, pod, prof, limit, want_log)
        _same(got, ref, check_log=want_log)


@pytest.mark.parametrize("kb", ["1", "3", "64"])
@pytest.mark.parametrize("seed", range(12))
def test_multi_kernel_form_blind_level_batches(ccref, monkeypatch, seed, kb):
    """The multi-kernel form (CCSIM_PERSIST=0; also every sharded run and every snapshot beyond the persistent form's LDS) resolves
    several score levels per pass as well: blind commit on the rows, validation in the decision, roll-back by stamp."""
    monkeypatch.setenv("CCSIM_PERSIST", "0")
    monkeypatch.setenv("CCSIM_LEVEL_BATCH", kb)
    rng = np.random.default_rng(7000 + seed)
    nodes, pod, prof = H.random_case(rng, int(rng.integers(1, 3000)))
    limit = int(rng.choice([0, 0, 37, 500]))
    ref = ccref.run(prof, nodes, pod, max_limit=limit)
    for want_log in (False, True):
        got, _ = _run(nodes, pod, prof, limit, want_log)
        _same(got, ref, check_log=want_log)


def test_lost_workgroup_falls_back_to_the_multi_kernel_form(ccref, monkeypatch):
    """A grid barrier that cannot complete (here injected: workgroup 0 never arrives -- what a CU mask or a co-tenant would cause): the
    persistent launch writes nothing back, ccsim_run redoes the run on the multi-kernel path from the untouched state."""
    monkeypatch.setenv("CCSIM_PERSIST_FAULT", "1")
    nodes, pod, prof = synth.make_config("C3", n_nodes=3000, seed=17)
    for limit, want_log in ((0, False), (700, True)):
        ref = ccref.run(prof, nodes, pod, max_limit=limit, threads=8)
        got, st = _run(nodes, pod, prof, limit, want_log)
        _same(got, ref, check_log=want_log)
        assert np.array_equal(st["pod_count"], nodes.pod_count + ref.per_node_count)


def test_continued_runs_and_mode_switches(ccref):
    """The persistent launch starts from the columns and writes them back: runs continue across launches and modes."""
    nodes, pod, prof = synth.make_config("C3", n_nodes=6000, seed=99)
    ref = ccref.run(prof, nodes, pod, max_limit=0, threads=8)
    e = capi.Engine(device=0)
    e.load(nodes, pod, prof)
    a = e.run(max_limit=1000, mode="batched", want_log=False, log_cap=0)
    b = e.run(max_limit=200, mode="sequential", log_cap=200)
    c = e.run(max_limit=5000, mode="batched", log_cap=5000)
    d = e.run(max_limit=0, mode="batched", want_log=False, log_cap=0)
    assert (a.placed, b.placed, c.placed) == (1000, 200, 5000) and a.placed + b.placed + c.placed + d.placed == ref.placed
    assert np.array_equal(b.log, ref.log[1000:1200]) and np.array_equal(c.log, ref.log[1200:6200])
    total = a.per_node_count + b.per_node_count + c.per_node_count + d.per_node_count
    assert np.array_equal(total, ref.per_node_count) and d.stop == ref.stop and np.array_equal(d.hist, ref.hist)
    e.close()


def test_large_snapshot_4_nodes_per_thread(ccref):
    """> 512k nodes: 4 nodes per thread (the 1M-node BASELINE shape); oracle prefix + closed-form exhaustive vector."""
    nodes, pod, prof = synth.make_config("C4", n_nodes=600_000, seed=5)
    ref = ccref.run(prof, nodes, pod, max_limit=400, threads=8)
    for want_log in (True, False):
        got, _ = _run(nodes, pod, prof, 400, want_log)
        _same(got, ref, check_log=want_log)
    full, _ = _run(nodes, pod, prof, 0, False)
    free_c, free_m = nodes.alloc[0] - nodes.req[0], nodes.alloc[1] - nodes.req[1]
    cap = np.minimum(np.minimum(free_c // 150, free_m // (100 << 20)), (nodes.alloc_pods - nodes.pod_count).astype(np.int64)).clip(0)
    cap = np.where(nodes.unschedulable == 0, cap, 0)
    assert full.placed == int(cap.sum()) and np.array_equal(full.per_node_count.astype(np.int64), cap)


def test_fallback_when_the_snapshot_does_not_qualify(ccref):
    """Extended resources (NX > 0) and odd memory values (no narrow mirrors) take the multi-kernel path: same answers."""
    rng = np.random.default_rng(11)
    n = 900
    nodes = H.simple_nodes(rng.choice([4000, 8000, 64000], n), rng.integers(1 << 34, 1 << 38, n) | 1, np.full(n, 40),
                           req_mcpu=rng.integers(0, 2000, n), req_mem=rng.integers(0, 1 << 33, n))
    pod = H.simple_pod(137, (1 << 28) + 12345)
    ref = ccref.run(M.Profile.default(), nodes, pod, max_limit=0)
    got, _ = _run(nodes, pod, M.Profile.default(), 0, False)
    _same(got, ref, check_log=False)


_ORACLE_CACHE = {}


@pytest.mark.parametrize("vranks", ["2", "3", "4", "8"])
@pytest.mark.parametrize("cfg,n,limit", [("C3", 4096, 0), ("C3", 4096, 700), ("C2", 5000, 300), ("C3", 20_000, 12_345), ("C3", 1500, 0)])
def test_mailbox_form_on_virtual_ranks(ccref, monkeypatch, vranks, cfg, n, limit):
    """The cross-GPU form of the persistent kernel (ccsim_persist.h, MB = true: per-rank local reduce, the completing workgroup
    publishes the rank's eight words as tagged 8-byte granules into every rank's mailbox, everybody polls its own box), validated on
    ONE device: the grid's workgroups split into virtual ranks with separate sync blocks and mailboxes -- same protocol, same
    system-scope accesses, commit rows published only after every rank succeeded.  Blind and ordered paths."""
    monkeypatch.setenv("CCSIM_PERSIST_VRANKS", vranks)
    nodes, pod, prof = synth.make_config(cfg, n_nodes=n, seed=4321 + n)
    if (cfg, n, limit) not in _ORACLE_CACHE:  # (the same case for every number of ranks: one oracle run)
        _ORACLE_CACHE[(cfg, n, limit)] = ccref.run(prof, nodes, pod, max_limit=limit, threads=8)
    ref = _ORACLE_CACHE[(cfg, n, limit)]
    got, st = _run(nodes, pod, prof, limit, want_log=False)
    _same(got, ref, check_log=False)
    assert got.pass_launches == 1  # ONE persistent launch did the run: the mailbox form itself, not the multi-kernel fallback
    cnt = ref.per_node_count.astype(np.int64)
    assert np.array_equal(st["req_mcpu"], nodes.req[0] + cnt * int(pod.req[0]))
    assert np.array_equal(st["req_mem"], nodes.req[1] + cnt * int(pod.req[1]))
    assert np.array_equal(st["pod_count"], nodes.pod_count + ref.per_node_count)
    got, _ = _run(nodes, pod, prof, limit, want_log=True)  # ordered path: positions need the lower ranks' planned placements
    _same(got, ref, check_log=True)
    assert got.pass_launches == 1


@pytest.mark.parametrize("vranks", ["2", "5"])
@pytest.mark.parametrize("seed", range(12))
def test_mailbox_form_random_plugin_mix(ccref, monkeypatch, vranks, seed):
    monkeypatch.setenv("CCSIM_PERSIST_VRANKS", vranks)
    rng = np.random.default_rng(7000 + seed)
    nodes, pod, prof = H.random_case(rng, int(rng.integers(600, 6000)))
    limit = int(rng.choice([0, 0, 37, 500]))
    ref = ccref.run(prof, nodes, pod, max_limit=limit)
    for want_log in (False, True):
        got, _ = _run(nodes, pod, prof, limit, want_log)
        _same(got, ref, check_log=want_log)


def test_mailbox_form_1m_nodes_and_continued_runs(ccref, monkeypatch):
    """C4 at 1M nodes on 8 virtual ranks of 32 workgroups: oracle prefix with the log, the closed-form exhaustive vector without; and a
    run continued across launches (the commit rows of one launch are the columns the next one loads)."""
    monkeypatch.setenv("CCSIM_PERSIST_VRANKS", "8")
    nodes, pod, prof = synth.make_config("C4", n_nodes=1_000_000)
    ref = ccref.run(prof, nodes, pod, max_limit=300, threads=16)
    e = capi.Engine(device=0)
    e.load(nodes, pod, prof)
    a = e.run(max_limit=300, mode="batched", log_cap=300)
    assert a.placed == 300 and np.array_equal(a.log, ref.log) and np.array_equal(a.per_node_count, ref.per_node_count)
    b = e.run(max_limit=0, mode="batched", want_log=False, log_cap=0)  # continues from the state the first launch left
    free_c, free_m = nodes.alloc[0] - nodes.req[0], nodes.alloc[1] - nodes.req[1]
    cap = np.minimum(np.minimum(free_c // 150, free_m // (100 << 20)), (nodes.alloc_pods - nodes.pod_count).astype(np.int64)).clip(0)
    cap = np.where((nodes.unschedulable == 0), cap, 0)
    assert a.placed + b.placed == int(cap.sum())
    assert np.array_equal((a.per_node_count + b.per_node_count).astype(np.int64), cap)
    e.reset_state()
    c = e.run(max_limit=0, mode="batched", want_log=False, log_cap=0)
    assert c.placed == int(cap.sum()) and np.array_equal(c.per_node_count.astype(np.int64), cap) and np.array_equal(c.hist, b.hist)
    e.close()


def test_lazy_reset_restores_the_extra_resource_columns_too(ccref):
    """ccsim_reset_state is lazy when the next run is a persistent launch, and that launch loads cpu / memory / pod counts
    from the pristine copies itself -- the columns of the OTHER resources (ephemeral-storage, scalars), which a pod without such requests
    never reads, were left as an earlier pod's run had moved them.  Sequence: pod A (asks for ephemeral storage) runs; pod B (does not)
    is set, the state is reset, B runs a few placements on the persistent form; pod A is set again WITHOUT a reset: it must find B's
    placements and its own column pristine -- the oracle's run of A on the snapshot with B's placements applied."""
    import copy

    n = 3000
    nodes, pod_b, prof = synth.make_config("C3", n_nodes=n, seed=99)
    pod_a = copy.copy(pod_b)
    pod_a.req = pod_b.req.copy()
    pod_a.req[2] = 30 << 30  # 30 GiB of the nodes' 100 GiB ephemeral storage: three clones per node, whatever cpu / memory allow
    e = capi.Engine(device=0)
    e.load(nodes, pod_a, prof)
    first = e.run(max_limit=0, mode="batched", want_log=False)
    assert first.hist[M.R_RES0 + 2] > 0  # (ephemeral storage really was what ran out)
    e.set_pod(pod_b)
    e.reset_state()
    rb = e.run(max_limit=1000, mode="batched", want_log=False)  # the persistent launch consumes the lazy reset
    assert rb.placed == 1000 and rb.pass_launches >= 1
    e.set_pod(pod_a)
    ra = e.run(max_limit=0, mode="batched", want_log=False)
    e.close()
    after_b = nodes.copy()
    cnt = rb.per_node_count.astype(np.int64)
    for c in range(3):
        after_b.req[c] = after_b.req[c] + cnt * int(pod_b.req[c])
    after_b.nz_mcpu, after_b.nz_mem = after_b.nz_mcpu + cnt * pod_b.nz_mcpu, after_b.nz_mem + cnt * pod_b.nz_mem
    after_b.pod_count = (after_b.pod_count + rb.per_node_count).astype(np.int32)
    ref = ccref.run(prof, after_b, pod_a, max_limit=0, threads=8)
    assert ref.placed > 0
    _same(ra, ref, check_log=False)


@pytest.mark.parametrize("mode", ["batched", "sequential"])
def test_a_pod_spec_set_after_runs_of_a_finer_grained_one(ccref, mode):
    """The narrow mirrors keep memory in units of the largest power of two dividing every value of the snapshot AND of the pod.  A pod
    with a coarse unit (1 GiB) set after a run of a pod with a finer one (100 MiB = 25 x 4 MiB) meets columns that are no longer multiples
    of 64 MiB: the unit must come down with them (round 5: found by the one-cycle-at-a-time loop over several pod specs)."""
    import copy

    n = 2000
    nodes, pod_a, prof = synth.make_config("C3", n_nodes=n, seed=123)  # 150m / 100 MiB
    pod_b = copy.copy(pod_a)
    pod_b.req = pod_a.req.copy()
    pod_b.req[0], pod_b.req[1], pod_b.nz_mcpu, pod_b.nz_mem = 500, 1 << 30, 500, 1 << 30
    e = capi.Engine(device=0)
    e.load(nodes, pod_a, prof)
    ra = e.run(max_limit=3000, mode=mode, want_log=False)
    assert ra.placed == 3000
    e.set_pod(pod_b)
    rb = e.run(max_limit=0, mode=mode, want_log=False)
    e.close()
    after_a = nodes.copy()
    cnt = ra.per_node_count.astype(np.int64)
    for c in range(3):
        after_a.req[c] = after_a.req[c] + cnt * int(pod_a.req[c])
    after_a.nz_mcpu, after_a.nz_mem = after_a.nz_mcpu + cnt * pod_a.nz_mcpu, after_a.nz_mem + cnt * pod_a.nz_mem
    after_a.pod_count = (after_a.pod_count + ra.per_node_count).astype(np.int32)
    ref = ccref.run(prof, after_a, pod_b, max_limit=0, threads=8)
    _same(rb, ref, check_log=False)


@pytest.mark.parametrize("frame", ["on", "off"])
@pytest.mark.parametrize("cfg,n,limit,width", [("C3", 4096, 0, 1), ("C3", 4096, 700, 2), ("C2", 3000, 0, 1), ("C3", 777, 0, 2), ("C4", 1500, 0, 1), ("C3", 1, 0, 1)])
def test_step_frame_and_narrow_per_node_counts(ccref, monkeypatch, frame, cfg, n, limit, width):
    """Round 6: (1) state, histogram and sync block travel as ONE frame (one copy out, one copy back: CCSIM_FRAME_OFF=1 is the form of
    the rounds before, three fills and three copies); (2) ABI 5: ccsim_report.per_node_count_narrow -- the per-node counts in 1- or
    2-byte elements when the caller offers that and every count fits.  Same counts, histogram and state as the oracle either way, over
    repeated runs on one engine (reset in between: what bench.py's timed loop does)."""
    if frame == "off":
        monkeypatch.setenv("CCSIM_FRAME_OFF", "1")
    nodes, pod, prof = synth.make_config(cfg, n_nodes=n, seed=4321 + n)  # (the snapshots of test_fast_path_without_log_vs_oracle: the oracle's answers are memoized per session)
    ref = ccref.run(prof, nodes, pod, max_limit=limit, threads=8)
    e = capi.Engine(device=0)
    e.load(nodes, pod, prof)
    for it in range(3):
        e.reset_state()
        got = e.run(max_limit=limit, mode="batched", want_log=False, reuse_buffers=True, narrow_counts=width if it != 1 else 0)
        assert got.per_node_count.dtype == (np.int32 if it == 1 else (np.uint8 if width == 1 else np.uint16)), (it, got.per_node_count.dtype)
        _same(got, ref, check_log=False)
        assert np.array_equal(got.hist_taintset[: len(ref.hist_taintset)], ref.hist_taintset)
    st = e.read_state()
    assert np.array_equal(st["pod_count"], nodes.pod_count + ref.per_node_count)
    e.close()


def test_narrow_counts_are_refused_when_a_count_may_not_fit(ccref):
    """A node whose pod capacity is 300 may take more than 255 clones: one-byte counts are not filled (per_node_count is), two-byte
    ones are; a run that does not take the persistent form (a placement log: the ordered path still does; sequential mode does not)
    fills per_node_count."""
    nodes, pod, prof = synth.make_config("C2", n_nodes=2000, seed=5)
    nodes.alloc_pods[:] = 300
    nodes.alloc[0][:7] = nodes.alloc[0][:7] * 8  # a few nodes large enough to hold > 255 clones
    nodes.alloc[1][:7] = nodes.alloc[1][:7] * 8
    ref = ccref.run(prof, nodes, pod, max_limit=0, threads=8)
    assert ref.per_node_count.max() > 255
    e = capi.Engine(device=0)
    e.load(nodes, pod, prof)
    for width, dt in ((1, np.int32), (2, np.uint16)):
        e.reset_state()
        got = e.run(max_limit=0, mode="batched", want_log=False, reuse_buffers=True, narrow_counts=width)
        assert got.per_node_count.dtype == dt
        _same(got, ref, check_log=False)
    e.reset_state()
    got = e.run(max_limit=200, mode="sequential", want_log=False, reuse_buffers=True, narrow_counts=2)
    assert got.per_node_count.dtype == np.int32 and got.placed == 200
    e.close()


def test_more_taint_sets_than_the_step_frame_holds(ccref):
    """The step frame carries the FitError bins of 256 taint sets; a pod spec over more of them gets an array of its own and the launch
    its separate fills and copies (ccsim_engine::ts_in_frame).  Same histogram per taint set as the oracle -- every set rejected by the
    TaintToleration filter shows up in its own bin."""
    nodes, pod, prof = synth.make_config("C3", n_nodes=3000, seed=17)
    rng = np.random.default_rng(5)
    n_sets = 700
    nodes.taintset_id = rng.integers(0, n_sets, nodes.n).astype(np.int32)
    pod.taint_filter_ok = (rng.random(n_sets) < 0.6).astype(np.uint8)
    pod.taint_prefer_cnt = rng.integers(0, 4, n_sets).astype(np.int32)
    ref = ccref.run(prof, nodes, pod, max_limit=0, threads=8)
    assert (ref.hist_taintset > 0).sum() > 200
    e = capi.Engine(device=0)
    e.load(nodes, pod, prof)
    for narrow in (0, 1):
        e.reset_state()
        got = e.run(max_limit=0, mode="batched", want_log=False, reuse_buffers=True, narrow_counts=narrow)
        _same(got, ref, check_log=False)
        assert np.array_equal(got.hist_taintset[: n_sets], ref.hist_taintset[: n_sets])
    e.close()


@pytest.mark.parametrize("cfg,n", [("C3", 4096), ("C4", 1500), ("C2", 3000)])
def test_a_run_that_ends_at_an_event_ends_without_one_more_pass(ccref, monkeypatch, cfg, n):
    """Round 6: when the batch that ends at a normalization event also fills the last feasible node (the run's last holder of a maximum is
    the run's last node), the launch ends there -- the feasible count it tracks is exact -- instead of after a re-score pass that finds
    nothing (CCSIM_PERSIST_END=0: the rounds before).  Same result either way, one sync less where the run ends that way."""
    nodes, pod, prof = synth.make_config(cfg, n_nodes=n, seed=4321 + n)
    ref = ccref.run(prof, nodes, pod, max_limit=0, threads=8)
    scans = {}
    for knob in ("1", "0"):
        monkeypatch.setenv("CCSIM_PERSIST_END", knob)
        got, st = _run(nodes, pod, prof, 0, want_log=False)
        _same(got, ref, check_log=False)
        assert np.array_equal(st["pod_count"], nodes.pod_count + ref.per_node_count)
        scans[knob] = got.scans
        got, _ = _run(nodes, pod, prof, 0, want_log=True)  # the ordered path
        _same(got, ref, check_log=True)
    assert scans["0"] - 1 <= scans["1"] <= scans["0"], scans
