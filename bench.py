#!/usr/bin/env python
"""bench.py -- simulated pod placements/sec on a synthetic 1M-node snapshot (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--mode batched|sequential] [--nodes 1000000]

A "step" is one pass of the hot path over one batch of synthetic input: one whole cluster-capacity
simulation (filter -> score -> select -> assume per placement, pkg/framework/simulator.go:356-381)
of `--limit` placements (0 = until the scheduler reports Unschedulable) against the HBM-resident
snapshot.  The snapshot is resident in HBM before the timed region starts; every step first restores
the dynamic node columns device-to-device (ccsim_reset_state, inside the timed region).

Workload (BASELINE config 4, "C4"): 1M synthetic nodes, default plugin set, examples/pod.yaml +
toleration + preferred node affinity, percentageOfNodesToScore=100.  N=1: the 1M-node snapshot on one
GPU.  N>1 (launched by torch.distributed.run, one rank per GPU): STRONG scaling by default -- the SAME
1M-node snapshot sharded by contiguous node range over the N GPUs (BASELINE config 4 literally), one RCCL
all-gather of a 256-byte record per pass (the max-loc exchange), only owning ranks update their columns.
It is latency-bound by construction (the per-pass GPU work shrinks N-fold while the exchange does not) and
one GPU holds the whole snapshot in LDS, so sharding buys capacity, not speed.
`--scaling weak` makes it an N x 1M-node cluster (1M nodes per GPU) instead.

Modes (identical placement sequences, see tests/): `batched` resolves a whole score level (many
placement rounds) per full pods x nodes pass; `sequential` is the literal one-round-per-pass loop.
The headline `value` is the batched mode; a sequential sample is reported next to it in `config`.

Prints ONE JSON line (rank 0).  `roofline` is measured live with HIP events on the engine's stream for the
dominant kernel (batched: k_level_persist, the ONE persistent launch of a simulation; sequential: k_scan) as a
physical HBM rate (PMC bytes / duration), with the sync-latency model that actually bounds it, and, under
`full_pass`, for the kernel that streams every node column (k_level_score / k_scan).  `cpu_baseline` is the C oracle (a port of the reference algorithm -- the Go reference cannot
be built here) timed on this box's host cores on a bounded sample.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

ge.load_package()
import numpy as np  # noqa: E402,F401
from cluster_capacity_amd import capi, dist as ccdist, synth  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
READ_PEAK_MEASURED_GBPS = 6420.0  # pure streaming read of the same 60 MB on the round-1 box (tools/hbm_peak.hip, profiles/r01/hbm_peak.txt)


def lib_sha16() -> str:
    import hashlib
    from cluster_capacity_amd import build as b

    return hashlib.sha256(open(b.lib_path(), "rb").read()).hexdigest()[:16]


def src_sha16() -> str:
    from cluster_capacity_amd import build as b

    return b.source_sha16()


def cpu_baseline(nodes, pod, prof, rounds: int, engine_log, blind_counts=None, count_slice=None):
    """Oracle (port of the reference algorithm) on the host cores, bounded sample.  Its placement log must equal the
    engine's first `rounds` placements on the ORDERED path, and its per-node counts must equal what the BLIND path -- the
    path the timed steps take: no log, 64-level batches, validate / roll back -- leaves when the same limit cuts a level
    (the checker checks the thing measured before the number is printed)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import ccref_py

    threads = min(16, os.cpu_count() or 1)  # reference default Parallelism = 16
    t0 = time.perf_counter()
    r = ccref_py.run(prof, nodes, pod, max_limit=rounds, threads=threads, want_log=True)
    dt = time.perf_counter() - t0
    if engine_log is not None:
        assert np.array_equal(np.asarray(r.log[: r.placed]), np.asarray(engine_log[: r.placed])), "engine and oracle placement logs differ"
    if blind_counts is not None:
        ref_counts = np.asarray(r.per_node_count) if count_slice is None else np.asarray(r.per_node_count)[count_slice[0]:count_slice[1]]
        assert np.array_equal(ref_counts, np.asarray(blind_counts)), "blind (timed) path and oracle per-node counts differ"
    return {
        "log_equals_engine_prefix": engine_log is not None,
        "timed_path_per_node_counts_equal_oracle_at_limit": blind_counts is not None,
        "value": r.placed / dt,
        "unit": "placements/s",
        "cores": threads,
        "kind": "port",
        "sample": f"first {r.placed} placement rounds of the same {nodes.n}-node snapshot, full scan per round "
                  f"(percentageOfNodesToScore=100), OpenMP over nodes, {dt:.1f}s",
    }


def secondary_lines(device: int):
    """BASELINE configs[4] (100k nodes x 1024 pod specs, round-robin) and its pod shape as ONE template on the 1M-node / 64-zone cluster,
    measured in the SAME process as the headline line ("no driver-timed line exists for anything but C4"), outside its timed
    region.  The oracle is the checker, as in cpu_baseline(): each engine's first placements must be the oracle's before its number counts."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import ccref_py
    from cluster_capacity_amd import capi, model as M, synth

    threads = min(16, os.cpu_count() or 1)
    out = {}

    def best_of(e, run, reps=3):
        best = None
        for _ in range(reps):
            e.reset_state()
            t0 = time.perf_counter()
            r = run()
            dt = time.perf_counter() - t0
            best = dt if best is None or dt < best else best
        return r, best

    # config 5: windows of <= 64 different pod specs per pass (csrc/ccsim_multi.h)
    nodes, pods, prof = synth.make_c5(100_000, 1024)
    C5_GATE = 2200  # every one of the 1024 specs at least twice, >= 34 windows incl. the ones that end early (oracle: ~5 s on 16 threads)
    ref = ccref_py.run_multi(prof, nodes, pods, max_limit=C5_GATE, threads=threads)
    e = capi.Engine(device=device)
    e.load(nodes, pods, prof)
    head = e.run(max_limit=C5_GATE, log_cap=C5_GATE)
    assert np.array_equal(head.log, ref.log) and np.array_equal(head.per_node_count, ref.per_node_count), "config 5: engine and oracle placement logs differ"
    r, dt = best_of(e, lambda: e.run(max_limit=200_000, want_log=False, log_cap=0))
    out["c5_100k_nodes_x_1024_specs"] = {"value": r.placed / dt, "unit": "placements/s", "placements": int(r.placed), "windows": int(r.scans),
                                          "us_per_window": dt * 1e6 / max(1, r.scans), f"first_{C5_GATE}_placements_equal_oracle": True, "windows_in_checked_prefix": int(head.scans)}
    e.close()
    # config 5's pod shape as one template (zone DoNotSchedule spread + hostname anti-affinity), the generator's own 64 zones at 1M nodes:
    # windows of placements per node pass (csrc/ccsim_coupled.h)
    n = 1_000_000
    nodes, pod, prof = synth.make_config("C3", n_nodes=n, seed=5)
    nodes.label_cols.append(np.arange(1, n + 1, dtype=np.int32))  # kubernetes.io/hostname
    pod.ipa = M.InterPodAffinity(key_cols=[len(nodes.label_cols) - 1], key_ndom=[n], anti_keys=[0], anti_self=[True], anti_existing=[None])
    pod.spread = [synth.zone_spread(n, max_skew=1)]
    CW_GATE = 4300  # two full 2048-cycle windows and the start of a third (oracle: ~90 cycles/s at 1M nodes on 16 threads, ~50 s)
    ref = ccref_py.run(prof, nodes, pod, max_limit=CW_GATE, threads=threads)
    e = capi.Engine(device=device)
    e.load(nodes, pod, prof)
    head = e.run(max_limit=CW_GATE, mode="sequential", log_cap=CW_GATE)
    assert np.array_equal(head.log, ref.log) and np.array_equal(head.per_node_count, ref.per_node_count), "coupled template: engine and oracle placement logs differ"
    head_windows = e.coupled_info()["windows"]
    r, dt = best_of(e, lambda: e.run(max_limit=50_000, mode="sequential", want_log=False, log_cap=0))
    info = e.coupled_info()
    out["coupled_template_1M_nodes_64_zones"] = {"value": r.placed / dt, "unit": "placements/s", "placements": int(r.placed), "node_passes": int(r.scans),
                                                  "us_per_pass": dt * 1e6 / max(1, r.scans), "windowed": bool(info["windows"] and not info["fell_back"]),
                                                  f"first_{CW_GATE}_placements_equal_oracle": True, "windows_in_checked_prefix": int(head_windows)}
    e.close()
    # the headline snapshot and pod under the reference's DEFAULT percentageOfNodesToScore (0 = adaptive: 5 % at 1M nodes, the first K = 50 000
    # feasible nodes of the rotating visiting order per cycle, schedule_one.go:610-723; SURVEY 8(d) "mode B"): the literal sequence of
    # scheduling cycles, one placement each -- evaluated a LAP of the ring at a time (the ~19 cycles of a lap visit disjoint stretches:
    # csrc/ccsim_sampled.h k_sb_laps).  Gate: 3000 cycles = ~158 laps, each of them once round the ring (a wrap per lap): log AND nodes visited.
    import dataclasses
    nodes, pod, prof = synth.make_config("C4", n_nodes=n)
    prof = dataclasses.replace(prof, percentage_of_nodes_to_score=0)
    MB_GATE = 3000
    ref = ccref_py.run(prof, nodes, pod, max_limit=MB_GATE, threads=threads)
    e = capi.Engine(device=device)
    e.load(nodes, pod, prof)
    head = e.run(max_limit=MB_GATE, mode="sequential", log_cap=MB_GATE)
    assert np.array_equal(head.log, ref.log) and head.evaluated_total == ref.evaluated_total, "mode B: engine and oracle differ (log / nodes visited)"
    gate_info = e.sampled_info()
    r, dt = best_of(e, lambda: e.run(max_limit=100_000, mode="sequential", want_log=False, log_cap=0))
    info = e.sampled_info()
    out["mode_b_adaptive_sampling_1M_nodes"] = {"value": r.placed / dt, "unit": "placements/s", "placements": int(r.placed), "us_per_cycle": dt * 1e6 / max(1, r.placed),
                                                "nodes_visited_per_cycle": r.evaluated_total / max(1, r.placed), "resident_form": bool(r.pass_launches > 0),
                                                "lap_at_a_time": bool(info["laps_form"]), "laps": int(info["laps"]), "cycles_per_lap": r.placed / max(1, info["laps"]),
                                                "us_per_lap": dt * 1e6 / max(1, info["laps"]), "kernel_launches": int(info["launches"]),
                                                f"first_{MB_GATE}_placements_and_visited_nodes_equal_oracle": True, "laps_in_checked_prefix": int(gate_info["laps"])}
    e.close()
    # ... and config 5's pod shape as ONE template (zone DoNotSchedule spread + hostname anti-affinity) under that default percentage -- what both
    # hosts run for such a template when the flag is left unset: per-(block, zone) entries under the mask of eligible zones
    # (csrc/ccsim_sampled_zone.h; round 5: three node passes per cycle, 66 us at 1M nodes).  Gate: 700 cycles = 10 rounds of the constraint.
    nodes, pod, prof = synth.make_config("C3", n_nodes=n, seed=5)
    nodes.label_cols.append(np.arange(1, n + 1, dtype=np.int32))  # kubernetes.io/hostname
    pod.ipa = M.InterPodAffinity(key_cols=[len(nodes.label_cols) - 1], key_ndom=[n], anti_keys=[0], anti_self=[True], anti_existing=[None])
    pod.spread = [synth.zone_spread(n, max_skew=1)]
    prof = dataclasses.replace(prof, percentage_of_nodes_to_score=0)
    MZ_GATE = 700
    ref = ccref_py.run(prof, nodes, pod, max_limit=MZ_GATE, threads=threads)
    e = capi.Engine(device=device)
    e.load(nodes, pod, prof)
    head = e.run(max_limit=MZ_GATE, mode="sequential", log_cap=MZ_GATE)
    assert np.array_equal(head.log, ref.log) and head.evaluated_total == ref.evaluated_total, "coupled template, default percentage: engine and oracle differ (log / nodes visited)"
    r, dt = best_of(e, lambda: e.run(max_limit=20_000, mode="sequential", want_log=False, log_cap=0), reps=2)
    info = e.sampled_info()
    out["coupled_template_default_percentage_1M_nodes"] = {"value": r.placed / dt, "unit": "placements/s", "placements": int(r.placed), "us_per_cycle": dt * 1e6 / max(1, r.placed),
                                                           "nodes_visited_per_cycle": r.evaluated_total / max(1, r.placed), "resident_zone_form": bool(info["zone_form"]),
                                                           f"first_{MZ_GATE}_placements_and_visited_nodes_equal_oracle": True}
    e.close()
    return out


def launch_ranks(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: the same command as N ranks of one node, one per GPU, the way the contract's launcher
    would start it (python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py ...).
    Rank 0 prints the line; this process only passes the ranks' output and exit status on."""
    import socket
    import subprocess

    with socket.socket() as s:  # a free port of the loopback interface (the container's hostname may not resolve)
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None,
                    help="GPUs of this node = ranks of the job.  Under a launcher (WORLD_SIZE set: torch.distributed.run) it must equal the world size; "
                         "without one, N > 1 makes bench.py launch itself as N ranks (torch.distributed.run, 127.0.0.1).  Default: the launcher's world size, else 1")
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default: 20 in the batched mode -- a step is ~0.3 ms -- and 3 in the sequential mode)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed steps before them (default: 3 / 1)")
    ap.add_argument("--mode", default="batched", choices=["sequential", "batched"])
    ap.add_argument("--nodes", type=int, default=1_000_000, help="nodes per GPU (weak) / in the whole snapshot (strong)")
    ap.add_argument("--scaling", default="strong", choices=["weak", "strong"],
                    help="N > 1: strong = ONE --nodes snapshot sharded over the N GPUs (BASELINE config 4 literally); weak = N x --nodes")
    ap.add_argument("--counts-width", type=int, default=1, choices=[0, 1, 2],
                    help="bytes per per-node count the caller offers (ABI 5 per_node_count_narrow; 0 = int32 only)")
    ap.add_argument("--no-variants", action="store_true", help="skip the multi-kernel / wide-path comparison runs")
    ap.add_argument("--limit", type=int, default=-1, help="placements per step (0 = until Unschedulable; "
                    "-1 = mode default: 0 for batched, 2048 for sequential)")
    ap.add_argument("--seq-rounds", type=int, default=20000, help="rounds of the sequential-mode sample (0 = skip)")
    ap.add_argument("--cpu-rounds", type=int, default=1600,
                    help="placement rounds of the CPU baseline sample (the oracle scans every node per round: ~9 ms per round at 1M nodes "
                         "on 16 threads, so ~15 s); its placement log must equal the engine's first --cpu-rounds placements")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip config 5 and the coupled template (`secondary` in the line; they need the oracle: also off with --no-cpu)")
    ap.add_argument("--no-roofline", action="store_true", help="skip the per-launch timing run (PMC collection runs)")
    args = ap.parse_args()
    # --gpus is the number of ranks, one per GPU.  Either a launcher made them (the driver: python -m torch.distributed.run --nproc-per-node N
    # bench.py --gpus N) and the flag must agree with it, or bench.py makes them itself: a plain `python bench.py --gpus 8` is an 8-rank job too.
    world_env = os.environ.get("WORLD_SIZE")
    if world_env is None and args.gpus is not None and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus))
    if args.gpus is None:
        args.gpus = int(world_env or 1)
    if int(world_env or 1) != args.gpus:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but the launcher started {world_env} rank(s) (WORLD_SIZE): refusing to print a line for the wrong job\n")
        sys.exit(2)
    if args.steps is None:
        args.steps = 20 if args.mode == "batched" else 3
    if args.warmup is None:
        args.warmup = 3 if args.mode == "batched" else 1

    # ONE JSON line on stdout: RCCL announces itself on stdout (version banner, some of it at exit) -- everything but the line goes
    # to stderr: fd 1 points at stderr for the run, the line is written to the saved descriptor
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch

    have_gpu = torch.cuda.is_available()
    if not have_gpu and not os.environ.get("CCSIM_LIB"):
        # no CPU path exists in the product.  (tests/test_bench_line.py runs the launcher / rank plumbing of this file on the CPU with
        # tests/abi_recorder.c named by CCSIM_LIB -- a stand-in that schedules nothing -- and gloo in place of RCCL; never a measurement.)
        sys.stderr.write("bench.py: no GPU visible (torch.cuda.is_available() is False): the engine is HIP only\n")
        sys.exit(3)
    if have_gpu:
        torch.cuda.set_device(local_rank)
    distributed = world > 1 or os.environ.get("CCSIM_FORCE_DIST") == "1"  # world == 1 over RCCL: a plumbing self-test
    if distributed:
        import torch.distributed as dist

        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # N > 1, two forms of the same sharded run, both measured by this process:
        #   * the RCCL pass protocol (commit + reduce + ncclAllGather(256 B/rank) + decide per pass): the HEADLINE -- the library's default,
        #     the form whose every piece has run on real hardware (one-rank RCCL on the GPU box, the protocol on gloo);
        #   * the persistent level kernel ACROSS the GPUs (mailboxes over xGMI inside ONE launch per rank; csrc/ccsim_persist.h MB form,
        #     include/ccsim.h ccsim_dist_mbox_*): connected here (CCSIM_DIST_MAILBOX=1), PROBED by one untimed step, and timed as the A/B
        #     field `multi_gpu_forms.mailbox` only if that probe stood on every rank.  A launch that had to be abandoned is never tried
        #     again for the pod spec (libccsim.so: mb_go = 0), so a form that cannot run on this box costs its bounded spins once.
        os.environ.setdefault("CCSIM_DIST_MAILBOX", "1")
        if world == 1:  # CCSIM_FORCE_DIST=1 without a launcher: a one-rank job on this GPU
            for k, v in (("RANK", "0"), ("WORLD_SIZE", "1"), ("LOCAL_RANK", "0"), ("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29513")):
                os.environ.setdefault(k, v)
        if have_gpu:
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
        else:
            dist.init_process_group("gloo")
        assert dist.get_world_size() == world == args.gpus, (dist.get_world_size(), world, args.gpus)
    limit = args.limit if args.limit >= 0 else (0 if args.mode == "batched" else 2048)
    n_global = args.nodes * world if args.scaling == "weak" else args.nodes
    lo, hi = ccdist.shard_bounds(n_global, world, rank)
    nodes, pod, prof = synth.make_config("C4", n_nodes=hi - lo, offset=lo, n_total=n_global)

    def barrier():
        if distributed:
            dist.barrier()
        if have_gpu:
            torch.cuda.synchronize()

    if distributed:
        runner = ccdist.make_torch_runner(nodes, pod, prof, lo, n_global, local_rank)
        eng = runner.engine

        def step(mode, lim):
            eng.reset_state()
            if isinstance(runner, ccdist.LibraryRunner):  # (the shard's counts into the engine's page-locked array, as at N = 1)
                return runner.run(max_limit=lim, mode=mode, reuse_buffers=True)
            return runner.run(max_limit=lim, mode=mode)
    else:
        eng = capi.Engine(device=local_rank)
        eng.load(nodes, pod, prof)

        def step(mode, lim):
            # the per-node counts are delivered into the engine's page-locked result array, reused from step to step (include/ccsim.h
            # ccsim_host_alloc): a fresh pageable array per step costs 0.18 ms of page faults and staging at 1M nodes
            # (profiles/r04/step_breakdown.txt), which is the caller's allocation policy and not the simulation
            # ABI 5: the counts in ONE byte each where every count provably fits (per_node_count_narrow: no node takes more clones than its
            # pod capacity, 110 here) -- 1 MB instead of 4 MB over PCIe; the engine answers int32 where that does not hold
            eng.reset_state()
            return eng.run(max_limit=lim, mode=mode, want_log=False, reuse_buffers=True, narrow_counts=args.counts_width)

    library_driven = distributed and isinstance(runner, ccdist.LibraryRunner)
    forms = None
    if library_driven:  # the timed steps take the pass protocol; the mailbox form is probed and timed after them
        os.environ["CCSIM_DIST_FORM"] = "passes"

    def timed_steps(n_steps):
        """EXACTLY n_steps steps between two barriers: (whole time, per-step times, the last result, placements, passes)"""
        barrier()
        t_0 = time.perf_counter()
        per, placed_, scans_, res = [], 0, 0, None
        for _ in range(n_steps):
            s_0 = time.perf_counter()
            res = step(args.mode, limit)
            per.append(time.perf_counter() - s_0)
            placed_ += res.placed
            scans_ += res.scans
        barrier()
        return time.perf_counter() - t_0, per, res, placed_, scans_

    def over_ranks(x, op="max"):
        if not distributed:
            return x
        t_ = torch.tensor([x], dtype=torch.float64, device=f"cuda:{local_rank}" if have_gpu else "cpu")
        dist.all_reduce(t_, op=dist.ReduceOp.MAX if op == "max" else dist.ReduceOp.MIN)
        return float(t_.item())

    for _ in range(args.warmup):
        step(args.mode, limit)
    dt, per_step, r, placed, scans = timed_steps(args.steps)
    r.per_node_count = r.per_node_count.copy()  # (a view of the reused result array until here: later runs of the engine overwrite it)
    dt = over_ranks(dt)
    if library_driven:
        def form_stats(dt_, per_, res_, placed_):
            return {"form": eng.dist_info()["last_form"], "ms_per_step": dt_ / args.steps * 1e3, "max_step_ms": over_ranks(max(per_)) * 1e3,
                    "median_step_ms": over_ranks(float(np.median(per_))) * 1e3, "value": placed_ / dt_, "passes_per_step": int(res_.scans)}
        forms = {"passes": form_stats(dt, per_step, r, placed)}
        assert forms["passes"]["form"] == "passes", forms
        # the persistent kernel across the GPUs: one untimed probe step; timed only if it stood on every rank (all-reduced inside the library)
        os.environ["CCSIM_DIST_FORM"] = "mailbox"
        info0 = eng.dist_info()
        p0 = time.perf_counter()
        rp = step(args.mode, limit)
        probe_s = over_ranks(time.perf_counter() - p0)
        info = eng.dist_info()
        stood = over_ranks(1.0 if info["last_form"] == "mailbox" else 0.0, "min") == 1.0
        forms["mailbox"] = {"connected": bool(info["mailboxes_connected"]), "probe_step_ms": probe_s * 1e3, "probe_stood_on_every_rank": stood,
                            "launches_abandoned": int(info["mailbox_abandoned"] - info0["mailbox_abandoned"]),
                            "probe_result_equals_pass_protocol": bool(rp.placed == r.placed and np.array_equal(rp.per_node_count, r.per_node_count))}
        if stood and args.mode == "batched":
            for _ in range(args.warmup):
                step(args.mode, limit)
            mdt, mper, mr, mplaced, _ = timed_steps(args.steps)
            forms["mailbox"].update(form_stats(over_ranks(mdt), mper, mr, mplaced))
            forms["mailbox"]["result_equals_pass_protocol"] = bool(mr.placed == r.placed and np.array_equal(mr.per_node_count, r.per_node_count))
        os.environ["CCSIM_DIST_FORM"] = "passes"

    # the literal one-round-per-pass loop on the same snapshot, for comparison (untimed by the driver)
    seq = None
    if args.mode == "batched" and args.seq_rounds > 0:
        nseq = args.seq_rounds if not distributed else min(args.seq_rounds, 2048)  # (on shards a cycle is an exchange between the ranks)
        step("sequential", nseq)
        barrier()
        s0 = time.perf_counter()
        rs = step("sequential", nseq)
        barrier()
        seq = rs.placed / (time.perf_counter() - s0)

    # Roofline.  The dominant kernel of the batched mode is the persistent level kernel k_level_persist (csrc/ccsim_persist.h:
    # one launch per simulation; it reads the narrow node columns once, keeps them in LDS, and writes the state back at
    # the end), of the sequential mode k_scan.  Duration: HIP events around the launch on the engine's stream
    # (ccsim_report.kernel_ns of the LAST timed step; rocprofv3 --kernel-trace --stats of this command, profiles/r04/,
    # reports the same average).  `achieved` = HBM bytes the launch really moves (rocprofv3 PMC, profiles/r04/pmc_traffic.json,
    # accepted only if it was collected with THIS libccsim.so) / duration: a physical rate.  The persistent kernel is not
    # HBM-bound -- it is bound by its grid-wide syncs (syncs x (barrier latency + the run-downs of the slowest workgroup)) --
    # so the fraction is small by design; `sync_bound` carries that model.  The work the reference semantics imply
    # (every placement round evaluates every node) is reported separately under `algorithmic`, never as GB/s.
    # The kernels that DO stream every node column per launch (k_level_score, the multi-kernel batched mode's full pass /
    # k_scan, the sequential mode's pass) are timed as a train of back-to-back launches on the freshly restored snapshot
    # and reported under `full_pass` with both the algorithmic (60 B/node) and the physical (narrow mirrors) byte counts.
    persistent = args.mode == "batched" and not distributed and os.environ.get("CCSIM_PERSIST", "1") != "0"
    kernel = "k_level_persist" if persistent else ("k_level_commit" if args.mode == "batched" else "k_scan")
    sha = lib_sha16()
    pmc, pmc_note = {}, "no profiles/rNN/pmc_traffic.json for this workload"
    import glob
    for pmc_path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]", "pmc_traffic.json")), reverse=True):  # the newest round first
        rel = os.path.relpath(pmc_path, ROOT)
        try:
            pj = json.load(open(pmc_path))
            if hi - lo != 1_000_000:
                pmc_note = "PMC traffic was collected at 1,000,000 nodes per GPU"
            elif pj.get("lib_sha16") != sha and pj.get("src_sha16") != src_sha16():  # (the binary embeds its build path: the sources decide)
                pmc_note = f"stale: {rel} was collected with libccsim.so {pj.get('lib_sha16')} / sources {pj.get('src_sha16')}, this run uses {sha} / {src_sha16()}"
                continue
            else:
                pmc, pmc_note = pj["kernels"], f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, same libccsim.so ({rel})"
        except (OSError, KeyError, ValueError):
            continue
        break
    roofline = None
    if not args.no_roofline and not distributed:
        eng.reset_state()
        train = 200
        scan_ns, bytes_per_scan = eng.time_scan(train, mode=args.mode)
        full_s = scan_ns / train / 1e9
        full_kernel = "k_level_score" if args.mode == "batched" else "k_scan"
        n_here = hi - lo
        phys = n_here * (36 + (4 if args.mode == "batched" else 0))  # int32 mirrors (24 B) + stat, alloc_pods, pod_count (12 B) [+ the 4-byte score cache written]
        full_pass = {
            "kernel": full_kernel, "us_per_launch": full_s * 1e6, "launches_timed": train,
            "bytes_algorithmic_not_moved": bytes_per_scan,  # SURVEY 8(d)'s 60 B/node of int64 columns: what the reference semantics read, NOT what this kernel moves
            "bytes_physical": phys, "achieved": phys / full_s / 1e9, "frac": phys / full_s / 1e9 / HBM_PEAK_GBPS,
            "frac_of_measured_read_peak": phys / full_s / 1e9 / READ_PEAK_MEASURED_GBPS,
            "traffic": pmc.get(full_kernel, {}).get("hbm_bytes_largest_launch" if args.mode == "batched" else "hbm_bytes_per_launch"),
        }
        launches = max(1, r.pass_launches) if persistent else max(1, r.scans)
        dom_s = r.kernel_ns / 1e9 / (launches if persistent else 1)
        traffic = pmc.get(kernel, {}).get("hbm_bytes_per_launch")
        # without counters: what the kernel is written to move (the PMC run of round 4 says 55.6 MB read + 23.8 MB written at 1M nodes)
        model_bytes = n_here * (56 + 24)  # (round 4: read 48 B pristine columns + ~8 B static words, write the 24 B of mirrors and pod counts)
        moved = traffic if traffic else model_bytes
        roofline = {
            # (the contract's vocabulary is hbm | mfma; neither bounds this kernel -- VERDICT r2: say what does)
            "bound": "sync-latency" if persistent else "hbm", "kernel": kernel,
            "achieved": moved / dom_s / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": moved / dom_s / 1e9 / HBM_PEAK_GBPS,
            "traffic": traffic, "traffic_source": pmc_note, "bytes_per_launch": moved,
            "bytes_definition": "HBM bytes one launch moves (PMC when available, else the kernel's designed traffic): a physical rate",
            "us_per_launch": dom_s * 1e6, "launches_timed": int(launches),
            # what a timed step spends outside its dominant kernel: the per-node counts over PCIe (4 MB at 1M nodes),
            # the state upload / read-back, launch and the one stream sync, the Python binding (profiles/r04/step_breakdown.txt)
            "step_overhead_us": (dt / args.steps - dom_s * (launches if persistent else 1)) * 1e6 if persistent else None,
            "sync_bound": {
                "what": "the persistent kernel is bound by grid-wide sync latency, not HBM: one reduce+barrier per resolved batch of score levels",
                "syncs_per_launch": int(r.scans), "us_per_sync": dom_s * 1e6 / max(1, r.scans),
                "barrier_floor_us": 2.6,  # tools/barrier_bench.hip on this chip: arrive + release + poll, nothing published (profiles/r02/barrier_bench.txt)
            } if persistent else None,
            "algorithmic": {
                "definition": "SURVEY 8(d): every placement round evaluates every node = rounds x nodes x 60 B; the level-batched engine "
                              "does not do that work (an algorithmic speed-up, not bandwidth)",
                "full_scan_bytes_per_step": int(r.rounds) * n_here * 60,
                "algorithmic_speedup_vs_bytes_moved": int(r.rounds) * n_here * 60 / max(1, moved),
            },
            "full_pass": full_pass,
        }
    variants = {}
    if not distributed:  # the same step with the per-node counts as int32 (ABI <= 4's only form): same counts
        counts_dtype = str(r.per_node_count.dtype)
        eng.reset_state(); r32 = eng.run(max_limit=limit, mode=args.mode, want_log=False, reuse_buffers=True)
        assert r32.per_node_count.dtype == np.int32 and np.array_equal(r32.per_node_count, r.per_node_count) and r32.placed == r.placed
        ts32 = []
        for _ in range(min(args.steps, 10)):
            torch.cuda.synchronize() if have_gpu else None
            v0 = time.perf_counter()
            eng.reset_state(); eng.run(max_limit=limit, mode=args.mode, want_log=False, reuse_buffers=True)
            ts32.append(time.perf_counter() - v0)
        variants["int32_counts_ms_per_step"] = sorted(ts32)[len(ts32) // 2] * 1e3
        variants["int32_counts_placements_per_s"] = r32.placed / (variants["int32_counts_ms_per_step"] * 1e-3)  # the same metric with ccsim_report.per_node_count (int32)
        variants["per_node_counts"] = (f"{counts_dtype} ({'ccsim_report.per_node_count_narrow, ABI 5: every count <= the largest pod capacity of the snapshot' if counts_dtype != 'int32' else 'ccsim_report.per_node_count'}); "
                                       "equal to the int32 vector of the same run (asserted)")
    if args.mode == "batched" and not distributed and not args.no_variants:
        # the same step on the other code paths of the batched mode (untimed by the driver): the multi-kernel form
        # (one commit + one decision dispatch per level) and the wide path (int64 columns, fp64 arithmetic)
        for name, env in (("multi_kernel_ms_per_step", {"CCSIM_PERSIST": "0"}), ("wide_path_ms_per_step", {"CCSIM_NARROW": "0"})):
            old = {k: os.environ.get(k) for k in env}
            os.environ.update(env)
            ve = capi.Engine(device=local_rank)
            ve.load(nodes, pod, prof)
            ve.reset_state(); ve.run(max_limit=limit, mode="batched", want_log=False)
            torch.cuda.synchronize()
            v0 = time.perf_counter()
            ve.reset_state(); vr = ve.run(max_limit=limit, mode="batched", want_log=False)
            variants[name] = (time.perf_counter() - v0) * 1e3
            assert vr.placed == r.placed, (name, vr.placed, r.placed)
            assert np.array_equal(vr.per_node_count, r.per_node_count), name  # the whole observable of an unlogged run
            ve.close()
            for k, v in old.items():
                os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)
    out = {
        "metric": "simulated pod placements/sec at 1M nodes",
        "value": placed / dt,
        "unit": "placements/s",
        "n_gpus": world,
        # the size of the communicator libccsim.so really built (ncclCommCount), not the flag: null on one GPU without a communicator
        "rccl_ranks_seen": (eng.dist_comm_size()[0] if distributed and isinstance(runner, ccdist.LibraryRunner) else (dist.get_world_size() if distributed else None)),
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "step_ms": {"median": over_ranks(float(np.median(per_step))) * 1e3, "max": over_ranks(max(per_step)) * 1e3},  # (the slowest rank's; a step stuck in a bounded spin shows in `max`)
        "higher_is_better": True,
        "scaling": args.scaling,
        "vs_baseline": None,
        "dtype": "int32",
        "data": "synthetic",
        "config": {
            "workload": f"{n_global}-node synthetic snapshot ({hi - lo} nodes/GPU; C4: default plugin set, examples/pod.yaml + toleration + "
                        f"preferred node affinity, percentageOfNodesToScore=100), "
                        f"{'until Unschedulable' if limit == 0 else str(limit) + ' placements'} per step",
            "mode": args.mode,
            "placements_per_step": placed // max(1, args.steps),
            "passes_per_step": scans // max(1, args.steps),
            "sequential_mode_placements_per_s": seq,
            "parallelism": f"node-shard x{world}",
            # the form the TIMED steps took (`value`, `ms_per_step`); `multi_gpu_forms` holds both forms' measurements, each with the form the
            # library reports for it, the median and the maximum step time (a step that sat in a bounded spin shows there)
            "multi_gpu_form": None if not distributed else "pass protocol: commit + reduce + ncclAllGather(256 B/rank) + decide per pass",
            "multi_gpu_forms": forms,
            "arithmetic": "exact integer results: the canonical state is int64 columns; this snapshot's values fit the engine's lossless "
                          "32-bit mirrors (validated per pod spec), which the timed kernels compute in (int32 / f32 estimates with exact "
                          "fix-ups); `wide_path_ms_per_step` is the same step on the int64 / fp64 path",
            **variants,
        },
        "roofline": roofline,
    }
    if distributed and not args.no_roofline:
        # a sharded run is the multi-kernel form: per pass a sparse commit (reads the 4-byte score cache of the shard, rewrites the
        # rows of the nodes in the batch), a one-block reduction, ONE 256-byte all-gather, a one-thread decision.  What bounds it is
        # the chain of dependent dispatches + the exchange, not bytes; the physical rate of the commit's cache read is given for scale.
        passes = max(1, int(r.scans))
        pass_s = r.kernel_ns / 1e9 / passes
        moved = (hi - lo) * 4
        if out["config"]["multi_gpu_form"].startswith("persistent"):
            # one persistent launch per rank: the shard's narrow state read once (36 B/node), the commit rows written once (32 B/node);
            # bound by its grid-wide syncs, each now a local reduce + one 128-byte store burst per peer over xGMI + a poll of the own box
            moved = (hi - lo) * (36 + 32)
            out["roofline"] = {
                "bound": "sync-latency", "kernel": "k_level_persist<K, mailbox> (one launch per rank and run)",
                "achieved": moved / (r.kernel_ns / 1e9) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": moved / (r.kernel_ns / 1e9) / 1e9 / HBM_PEAK_GBPS,
                "traffic": None, "bytes_per_launch": moved,
                "bytes_definition": "bytes the launch is written to move on this rank (no counters for the sharded run)",
                "us_per_launch": r.kernel_ns / 1e3, "launches_timed": 1,
                "sync_bound": {"syncs_per_launch": passes, "us_per_sync": r.kernel_ns / 1e3 / passes,
                               "what": "HIP events around the launch of the last timed step on rank 0; the step adds two ncclAllReduce agreements (go / finished) around it"},
            }
        else:
          out["roofline"] = {
            "bound": "sync-latency", "kernel": "k_level_commit (+ k_level_final, ncclAllGather 256 B/rank, k_level_decide) per pass",
            "achieved": moved / pass_s / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": moved / pass_s / 1e9 / HBM_PEAK_GBPS,
            "traffic": None, "bytes_per_launch": moved,
            "bytes_definition": "score-cache bytes of this rank's shard one commit pass reads (4 B/node); rows of the batch's nodes come on top",
            "us_per_launch": pass_s * 1e6, "launches_timed": passes,
            "exchange_bound": {"passes_per_step": passes, "us_per_pass": pass_s * 1e6, "levels_per_pass": "up to CCSIM_LEVEL_BATCH (default 384), blind + validated",
                               "what": "HIP events around the whole pass train of the last timed step on rank 0 (kernels + all-gathers)"},
        }
    if distributed and not args.no_cpu:
        # the oracle's per-node counts at a limit inside a level against this rank's shard of a sharded blind run to the same limit
        eng.reset_state()
        blind = runner.run(max_limit=args.cpu_rounds, mode=args.mode, want_log=False)  # (collective: every rank runs it)
        if rank == 0:
            nodes_full = synth.make_config("C4", n_nodes=n_global)[0] if world > 1 else nodes
            assert blind.placed == args.cpu_rounds, blind.placed
            out["cpu_baseline"] = cpu_baseline(nodes_full, pod, prof, args.cpu_rounds, None, blind.per_node_count, (lo, hi))
        dist.barrier()
    if rank == 0 and not args.no_cpu and not distributed:
        nodes_full = nodes if world == 1 else synth.make_config("C4", n_nodes=n_global)[0]
        eng.reset_state()
        head = eng.run(max_limit=args.cpu_rounds, mode=args.mode, want_log=True, log_cap=args.cpu_rounds)
        # the path the timed steps took (want_log=False: blind multi-level batches), cut by the same limit INSIDE a score level:
        # validation must roll the batch back and redo it in canonical order -- the per-node vector is the observable
        eng.reset_state()
        blind = eng.run(max_limit=args.cpu_rounds, mode=args.mode, want_log=False)
        assert blind.placed == head.placed, (blind.placed, head.placed)
        out["cpu_baseline"] = cpu_baseline(nodes_full, pod, prof, args.cpu_rounds, head.log, blind.per_node_count)
        if args.mode == "batched" and args.seq_rounds > 0:  # the sequential sample above: the same placements, cycle by cycle
            eng.reset_state()
            sq = eng.run(max_limit=args.cpu_rounds, mode="sequential", want_log=True, log_cap=args.cpu_rounds)
            assert np.array_equal(sq.log, head.log), "the sequential mode's placement log differs from the oracle's"
            out["config"]["sequential_mode_first_placements_equal_oracle"] = int(sq.placed)
            out["config"]["sequential_mode_form"] = ("full search on resident block summaries, one wave (k_sf_cycles)" if eng.sampled_info()["full_search_form"]
                                                     else "one pass over the nodes per cycle (k_scan_fused)")
        out["timed_path_check"] = {
            "what": "the blind path (no log) run with --max-limit inside a score level; per-node counts equal the oracle's at the same limit",
            "limit": args.cpu_rounds, "placed": int(blind.placed), "passes": int(blind.scans), "ordered_path_passes": int(head.scans)}
    elif rank == 0 and "cpu_baseline" not in out:
        out["cpu_baseline"] = None
    if rank == 0 and not distributed and not args.no_cpu and not args.no_secondary and args.mode == "batched":
        eng.close()  # (the headline engine's columns are not needed any more: the memory is the next engines')
        try:
            out["secondary"] = secondary_lines(local_rank)
        except Exception as ex:  # (the headline line above is complete: a failure here is reported, not fatal)
            out["secondary"] = {"error": f"{type(ex).__name__}: {ex}"[:400]}
    if not have_gpu:  # (the CPU plumbing test: nothing was scheduled, nothing was measured -- say so in the line itself)
        out["invalid"] = f"no GPU: ABI stand-in {os.environ.get('CCSIM_LIB')} in place of libccsim.so, gloo in place of RCCL; launcher / rank plumbing only"
    if rank == 0:
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
