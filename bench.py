#!/usr/bin/env python
"""bench.py -- simulated pod placements/sec on a synthetic 1M-node snapshot (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--mode sequential|batched] [--nodes 1000000]

A "step" is one pass of the hot path over one batch of synthetic input: `--rounds` placement rounds
of the cluster-capacity simulation loop (filter -> score -> select -> assume per placement) against
the HBM-resident snapshot.  The snapshot is resident in HBM before the timed region starts.

N=1: the 1M-node default-plugin-set snapshot (BASELINE config "1M synthetic nodes", C4) on one GPU.
N>1 (launched by torch.distributed.run, one rank per GPU): WEAK scaling -- every rank holds its own
1M-node shard of an N x 1M-node cluster (contiguous node ranges), one RCCL all-gather of a 64-byte
(packed score/position key) record per round picks the global winner, only the owning rank updates.

Prints ONE JSON line (rank 0).  `roofline` is measured live on the dominant kernel (k_scan) with HIP
events on the engine's stream; `cpu_baseline` is the C oracle (a port of the reference algorithm --
the Go reference cannot be built here) timed on this box's host cores on a bounded sample.
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
import numpy as np  # noqa: E402
from cluster_capacity_amd import capi, dist as ccdist, synth  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


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
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--mode", default="sequential", choices=["sequential", "batched"])
    ap.add_argument("--nodes", type=int, default=1_000_000, help="nodes per GPU")
    ap.add_argument("--rounds", type=int, default=0, help="placement rounds per step (0 = mode default)")
    ap.add_argument("--cpu-rounds", type=int, default=160)
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch

    torch.cuda.set_device(local_rank)
    distributed = world > 1
    if distributed:
        import torch.distributed as dist

        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
    rounds = args.rounds or 2048
    n_global = args.nodes * world
    offset = rank * args.nodes
    nodes, pod, prof = synth.make_config("C4", n_nodes=args.nodes, offset=offset)

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    if distributed:
        runner = ccdist.make_torch_runner(nodes, pod, prof, offset, n_global, local_rank)
        eng = runner.engine
        step = lambda: runner.run(max_limit=rounds, mode=args.mode)  # noqa: E731
    else:
        eng = capi.Engine(device=local_rank)
        eng.load(nodes, pod, prof)
        step = lambda: eng.run(max_limit=rounds, mode=args.mode, want_log=False)  # noqa: E731

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    placed = scans = 0
    for _ in range(args.steps):
        r = step()
        placed += r.placed
        scans += r.scans
    barrier()
    dt = time.perf_counter() - t0
    if distributed:
        t = torch.tensor([dt], dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # roofline of the dominant kernel, measured live with HIP events on the engine's stream
    scan_ns, bytes_per_scan = eng.time_scan(50)
    scan_s = scan_ns / 50 / 1e9
    achieved = bytes_per_scan / scan_s / 1e9
    out = {
        "metric": "simulated pod placements/sec at 1M nodes",
        "value": placed / dt,
        "unit": "placements/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "int64",
        "data": "synthetic",
        "config": {
            "workload": f"{n_global}-node synthetic snapshot ({args.nodes}/GPU), default plugin set (C4), examples/pod.yaml "
                        f"+ toleration + preferred node affinity, percentageOfNodesToScore=100",
            "mode": args.mode,
            "rounds_per_step": rounds,
            "placements": placed,
            "scans": scans,
            "parallelism": f"node-shard x{world}",
        },
        "roofline": {
            "bound": "hbm",
            "achieved": achieved,
            "peak": HBM_PEAK_GBPS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBPS,
            "traffic": None,
            "kernel": "k_scan",
            "bytes_per_launch": bytes_per_scan,
            "us_per_launch": scan_s * 1e6,
        },
    }
    if rank == 0 and not args.no_cpu and not distributed:
        out["cpu_baseline"] = cpu_baseline(nodes, pod, prof, args.cpu_rounds)
    elif rank == 0:
        out["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(out))
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
