#!/usr/bin/env python
"""bench.py -- simulated pod placements/sec on a synthetic 1M-node snapshot (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--mode batched|sequential] [--nodes 1000000]

A "step" is one pass of the hot path over one batch of synthetic input: one whole cluster-capacity
simulation (filter -> score -> select -> assume per placement, pkg/framework/simulator.go:356-381)
of `--limit` placements (0 = until the scheduler reports Unschedulable) against the HBM-resident
snapshot.  The snapshot is resident in HBM before the timed region starts; every step first restores
the dynamic node columns device-to-device (ccsim_reset_state, inside the timed region).

Workload (BASELINE config 4, "C4"): 1M synthetic nodes per GPU, default plugin set, examples/pod.yaml +
toleration + preferred node affinity, percentageOfNodesToScore=100.  N=1: the 1M-node snapshot on one
GPU.  N>1 (launched by torch.distributed.run, one rank per GPU): WEAK scaling -- an N x 1M-node cluster
sharded by contiguous node range (1M nodes per GPU), one RCCL all-gather of a 256-byte record per pass
(the max-loc exchange), only owning ranks update their columns.  `--scaling strong` shards ONE 1M-node
snapshot over the N GPUs instead (BASELINE config 4 literally); it is latency-bound by construction (the
per-pass GPU work shrinks N-fold while the exchange does not), see DESIGN.md section 5.

Modes (identical placement sequences, see tests/): `batched` resolves a whole score level (many
placement rounds) per full pods x nodes pass; `sequential` is the literal one-round-per-pass loop.
The headline `value` is the batched mode; a sequential sample is reported next to it in `config`.

Prints ONE JSON line (rank 0).  `roofline` is measured live with HIP events on the engine's stream for the
dominant kernel (batched: k_level_commit, the one launch per pass that commits a level off the score cache;
sequential: k_scan) and, under `full_pass`, for the kernel that streams every node column (k_level_score /
k_scan, 60 B/node); k_level_final (one block) is listed in profiles/.  `cpu_baseline` is the C oracle (a port of the reference algorithm -- the Go reference cannot
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


def cpu_baseline(nodes, pod, prof, rounds: int):
    """Oracle (port of the reference algorithm) on the host cores, bounded sample."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import ccref_py

    threads = min(16, os.cpu_count() or 1)  # reference default Parallelism = 16
    t0 = time.perf_counter()
    r = ccref_py.run(prof, nodes, pod, max_limit=rounds, threads=threads, want_log=False)
    dt = time.perf_counter() - t0
    return {
        "value": r.placed / dt,
        "unit": "placements/s",
        "cores": threads,
        "kind": "port",
        "sample": f"first {r.placed} placement rounds of the same {nodes.n}-node snapshot, full scan per round "
                  f"(percentageOfNodesToScore=100), OpenMP over nodes, {dt:.1f}s",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--mode", default="batched", choices=["sequential", "batched"])
    ap.add_argument("--nodes", type=int, default=1_000_000, help="nodes per GPU (weak) / in the whole snapshot (strong)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--limit", type=int, default=-1, help="placements per step (0 = until Unschedulable; "
                    "-1 = mode default: 0 for batched, 2048 for sequential)")
    ap.add_argument("--seq-rounds", type=int, default=2048, help="rounds of the sequential-mode sample (0 = skip)")
    ap.add_argument("--cpu-rounds", type=int, default=160)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-roofline", action="store_true", help="skip the per-launch timing run (PMC collection runs)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch

    torch.cuda.set_device(local_rank)
    distributed = world > 1 or os.environ.get("CCSIM_FORCE_DIST") == "1"  # world == 1 over RCCL: a plumbing self-test
    if distributed:
        import torch.distributed as dist

        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
    limit = args.limit if args.limit >= 0 else (0 if args.mode == "batched" else 2048)
    n_global = args.nodes * world if args.scaling == "weak" else args.nodes
    lo, hi = ccdist.shard_bounds(n_global, world, rank)
    nodes, pod, prof = synth.make_config("C4", n_nodes=hi - lo, offset=lo, n_total=n_global)

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    if distributed:
        runner = ccdist.make_torch_runner(nodes, pod, prof, lo, n_global, local_rank)
        eng = runner.engine

        def step(mode, lim):
            eng.reset_state()
            return runner.run(max_limit=lim, mode=mode)
    else:
        eng = capi.Engine(device=local_rank)
        eng.load(nodes, pod, prof)

        def step(mode, lim):
            eng.reset_state()
            return eng.run(max_limit=lim, mode=mode, want_log=False)

    for _ in range(args.warmup):
        step(args.mode, limit)
    barrier()
    t0 = time.perf_counter()
    placed = scans = 0
    for _ in range(args.steps):
        r = step(args.mode, limit)
        placed += r.placed
        scans += r.scans
    barrier()
    dt = time.perf_counter() - t0
    if distributed:
        t = torch.tensor([dt], dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # the literal one-round-per-pass loop on the same snapshot, for comparison (untimed by the driver)
    seq = None
    if args.mode == "batched" and args.seq_rounds > 0:
        step("sequential", args.seq_rounds)
        barrier()
        s0 = time.perf_counter()
        rs = step("sequential", args.seq_rounds)
        barrier()
        seq = rs.placed / (time.perf_counter() - s0)

    # Roofline.  The dominant kernel of the batched mode is k_level_commit (one launch per pass: reads the 4-byte score
    # cache of every node, runs the level's nodes down on their commit rows, re-scores them, reduces the next level);
    # of the sequential mode, k_scan.  Their average launch duration is measured live: one more run of the SAME workload
    # on a second engine whose passes are launched eagerly with a stop stamp per dispatch on the engine's stream
    # (cfg.time_passes -> hipExtLaunchKernelGGL events; rocprofv3 --kernel-trace --stats of this command, profiles/,
    # reports the same average).  `achieved` uses SURVEY 8(d)'s algorithmic bytes: one pass = one evaluation of every
    # (pod, node) pair = N x B_node, the full-scan definition -- what the pass would have to stream without the score
    # cache; `traffic` is what it really moves (PMC).  The kernel that DOES stream every node column, the full pass
    # k_level_score (first pass of a run and whenever the normalization constants move), is timed as a train of
    # back-to-back launches on the freshly restored snapshot and reported next to it (`full_pass`).
    kernel = "k_level_commit" if args.mode == "batched" else "k_scan"
    pmc = {}
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "r01", "pmc_traffic.json")))["kernels"] if hi - lo == 1_000_000 else {}
    except (OSError, KeyError, ValueError):
        pass
    roofline = None
    if not args.no_roofline:
        eng.reset_state()
        train = 200
        scan_ns, bytes_per_scan = eng.time_scan(train, mode=args.mode)
        full_s = scan_ns / train / 1e9
        full_kernel = "k_level_score" if args.mode == "batched" else "k_scan"
        full_pass = {
            "kernel": full_kernel, "achieved": bytes_per_scan / full_s / 1e9, "frac": bytes_per_scan / full_s / 1e9 / HBM_PEAK_GBPS,
            "frac_of_measured_read_peak": bytes_per_scan / full_s / 1e9 / READ_PEAK_MEASURED_GBPS,
            "us_per_launch": full_s * 1e6, "launches_timed": train, "bytes_per_launch": bytes_per_scan,
            # most k_level_score launches of a profiled run are no-op graph heads: the largest launch is a real one
            "traffic": pmc.get(full_kernel, {}).get("hbm_bytes_largest_launch" if args.mode == "batched" else "hbm_bytes_per_launch"),
        }
        pe = capi.Engine(device=local_rank, time_passes=True)  # this rank's shard as a stand-alone snapshot
        pe.load(nodes, pod, prof)
        prun = pe.run(max_limit=limit, mode=args.mode, want_log=False)
        pe.close()
        dom_s = prun.pass_kernel_ns / max(1, prun.pass_launches) / 1e9
        achieved = bytes_per_scan / dom_s / 1e9
        roofline = {
            "bound": "hbm",
            "achieved": achieved,
            "peak": HBM_PEAK_GBPS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBPS,
            "traffic": pmc.get(kernel, {}).get("hbm_bytes_per_launch"),
            "kernel": kernel,
            "bytes_per_launch": bytes_per_scan,
            "bytes_definition": "algorithmic, full-scan definition (SURVEY 8(d)): nodes x enabled column bytes per pass; "
                                "`traffic` = HBM bytes the launch really moves (rocprofv3 PMC, profiles/r01/pmc_traffic.json)",
            "us_per_launch": dom_s * 1e6,
            "launches_timed": int(prun.pass_launches),
            "full_pass": full_pass,
        }
    out = {
        "metric": "simulated pod placements/sec at 1M nodes",
        "value": placed / dt,
        "unit": "placements/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": args.scaling,
        "vs_baseline": None,
        "dtype": "int64",
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
            "arithmetic": "exact integer results (int64 columns); this snapshot's values fit the engine's lossless 32-bit mirrors, "
                          "which the scan / level kernels then use",
        },
        "roofline": roofline,
    }
    if rank == 0 and not args.no_cpu and not distributed:
        nodes_full = nodes if world == 1 else synth.make_config("C4", n_nodes=n_global)[0]
        out["cpu_baseline"] = cpu_baseline(nodes_full, pod, prof, args.cpu_rounds)
    elif rank == 0:
        out["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(out))
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
