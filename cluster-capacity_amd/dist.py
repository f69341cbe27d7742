"""Multi-GPU driver: one process per GPU, node-range shards, one RCCL collective per round.

Two drivers of the same protocol (include/ccsim.h):
  * LibraryRunner (default on GPUs): the whole sharded run inside libccsim.so -- ccsim_dist_run enqueues scan ->
    ncclAllGather -> decide from C++ on the engine's stream, over the engine's own RCCL communicator; torch.distributed only
    carries the 128-byte ncclUniqueId to the ranks (and the bench's barriers).
  * DistRunner: the stepwise entry points with a caller-supplied collective (torch.distributed all_gather; the gloo CPU
    tests use it with a stand-in engine; CCSIM_DIST_DRIVER=python selects it on GPUs for A/B runs).

torch.distributed is plumbing here (process group + the collective over xGMI); the shard work is
the HIP engine.  Protocol per pass (include/ccsim.h "multi-GPU stepping"; a pass is one placement
round in sequential mode and one whole score level -- many rounds -- in batched mode):

    engine.dist_scan()                         # full pass over the shard -> 128-byte record in `send`
    dist.all_gather_into_tensor(recv, send)    # the max-loc exchange (packed key in word 0)
    engine.dist_decide()                       # identical reduction on every rank; owners commit

The engine enqueues on torch's current stream, so the collective is ordered with the kernels
without any host synchronization; the host only syncs every `rounds_per_poll` rounds to read the
done flag.
"""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np

from . import capi
from . import model as M


def shard_bounds(n_global: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous ranges of the canonical node order: rank g owns [g*ceil(N/G), (g+1)*ceil(N/G))."""
    per = -(-n_global // world)
    lo = min(n_global, rank * per)
    return lo, min(n_global, lo + per)


def shard_pod(pod: M.PodSpec, lo: int, hi: int) -> M.PodSpec:
    """The pod spec for the shard [lo, hi): per-node side arrays (existing matching pods etc.) follow the nodes."""
    import copy

    p = copy.copy(pod)
    cut = lambda a: None if a is None else a[lo:hi]  # noqa: E731
    p.host_ports_conflict, p.image_score, p.volume_veto = cut(pod.host_ports_conflict), cut(pod.image_score), cut(pod.volume_veto)
    p.spread = [copy.copy(k) for k in pod.spread]
    for k in p.spread:
        k.node_match_count, k.node_included = cut(k.node_match_count), cut(k.node_included)
    if pod.ipa is not None:
        q = copy.copy(pod.ipa)
        q.aff_existing = cut(q.aff_existing)
        q.anti_existing = [cut(a) for a in q.anti_existing]
        q.exist_anti = [cut(a) for a in q.exist_anti]
        q.score_existing = [cut(a) for a in q.score_existing]
        if lo > 0:
            q.entries_existing = 0  # a cluster-wide count: contributed once (rank 0), then all-reduced
        p.ipa = q
    return p


class DistRunner:
    """Drives one rank's engine; `collective(recv, send)` performs the all-gather."""

    def __init__(self, engine, world: int, rank: int, send, recv, collective, rounds_per_poll: int = 32,
                 buffer_arg=lambda t: t.data_ptr()):
        self.engine, self.world, self.rank, self.send, self.recv = engine, world, rank, send, recv
        self.collective = collective
        self.rounds_per_poll = rounds_per_poll
        self.buffer_arg = buffer_arg  # how the engine wants the exchange buffers (the HIP engine: device pointers)

    def run(self, max_limit: int = 0, mode: str = "sequential", want_log: bool = False, log_cap: int = 0) -> M.RunResult:
        e = self.engine
        e.dist_begin(max_limit, mode, self.world, self.rank, self.buffer_arg(self.send), self.buffer_arg(self.recv),
                     log_cap if want_log else 0)
        while True:
            for _ in range(self.rounds_per_poll):
                e.dist_scan()
                self.collective(self.recv, self.send)
                e.dist_decide()
            done, _placed = e.dist_poll()
            if done:
                break
        return e.dist_finish(want_log, log_cap)


class _DevArray:
    """Zero-copy view of an engine table for torch (__cuda_array_interface__)."""

    def __init__(self, ptr: int, n: int, elem_bytes: int):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<i4" if elem_bytes == 4 else "<i8", "data": (ptr, False),
                                         "version": 2}


def table_tensors(engine, device: int):
    """The engine's replicated topology tables as torch tensors sharing their memory, with the reduction each needs."""
    import torch

    return [(torch.as_tensor(_DevArray(p, n, eb), device=f"cuda:{device}"), "max" if op == 1 else "sum")
            for (p, n, eb, op) in engine.dist_tables()]


def sync_tables(engine, device: int):
    """Topology-coupled plugins: every rank filled its count tables from its own nodes only -- all-reduce them
    (RCCL) so that every rank holds the cluster-wide tables (include/ccsim.h ccsim_dist_table)."""
    import torch.distributed as dist

    for t, op in table_tensors(engine, device):
        dist.all_reduce(t, op=dist.ReduceOp.MAX if op == "max" else dist.ReduceOp.SUM)
    engine.dist_tables_done()


class LibraryRunner:
    """The sharded run driven inside libccsim.so (ccsim_dist_run) over the engine's own RCCL communicator."""

    def __init__(self, engine, world: int, rank: int):
        self.engine, self.world, self.rank = engine, world, rank

    def run(self, max_limit: int = 0, mode: str = "sequential", want_log: bool = False, log_cap: int = 0, reuse_buffers: bool = False) -> M.RunResult:
        return self.engine.dist_run(max_limit, mode, want_log, log_cap, reuse_buffers=reuse_buffers)


def make_library_runner(nodes_shard: M.NodesSoA, pod: M.PodSpec, profile: M.Profile, global_offset: int, n_global: int,
                        device: int) -> LibraryRunner:
    """One rank of a torch.distributed job: load the shard, rendezvous the engine's RCCL communicator (rank 0's
    ncclUniqueId travels through torch.distributed), all-reduce the replicated topology tables inside the library."""
    import torch
    import torch.distributed as dist

    if torch.cuda.is_available():  # (absent only under the CPU tests, where an ABI recorder stands in for libccsim.so)
        torch.cuda.set_device(device)
    world, rank = dist.get_world_size(), dist.get_rank()
    eng = capi.Engine(device=device, use_graph=False)
    eng.load(nodes_shard, pod, profile, global_offset=global_offset, n_global=n_global)
    ids = [capi.dist_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(ids, src=0)
    eng.dist_comm_init(ids[0], world, rank)
    if eng.dist_tables():
        eng.dist_sync_tables()
    return LibraryRunner(eng, world, rank)


def make_torch_runner(nodes_shard: M.NodesSoA, pod: M.PodSpec, profile: M.Profile, global_offset: int, n_global: int,
                      device: int, rounds_per_poll: int = 32):
    import os

    import torch
    import torch.distributed as dist

    if os.environ.get("CCSIM_DIST_DRIVER", "library") != "python":
        return make_library_runner(nodes_shard, pod, profile, global_offset, n_global, device)
    torch.cuda.set_device(device)
    # a real (non-default) torch stream: the engine enqueues on it and RCCL orders against it.  The legacy
    # default stream has handle 0, which the C ABI reads as "create your own stream".
    ts = torch.cuda.Stream(device=device)
    torch.cuda.set_stream(ts)
    eng = capi.Engine(device=device, stream=ts.cuda_stream, use_graph=False)
    eng._torch_stream = ts  # keep it alive
    eng.load(nodes_shard, pod, profile, global_offset=global_offset, n_global=n_global)
    if eng.dist_tables():
        sync_tables(eng, device)
    world = dist.get_world_size()
    send = torch.zeros(capi.XCHG_WORDS, dtype=torch.int64, device=f"cuda:{device}")
    recv = torch.zeros(capi.XCHG_WORDS * world, dtype=torch.int64, device=f"cuda:{device}")

    def collective(r, s):
        dist.all_gather_into_tensor(r, s)

    return DistRunner(eng, world, dist.get_rank(), send, recv, collective, rounds_per_poll)


def merge_logs(logs) -> np.ndarray:
    """Each shard's log holds the node index at the positions of ITS placements and -1 elsewhere."""
    out = logs[0].copy()
    for l in logs[1:]:
        np.maximum(out, l, out=out)
    return out
