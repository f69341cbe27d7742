"""TEST INFRASTRUCTURE: a CPU stand-in for one shard's engine, speaking the multi-GPU stepping protocol of
include/ccsim.h (dist_begin / dist_scan / dist_decide / dist_poll / dist_finish) on CPU tensors, so that the
host-side sharding logic (cluster-capacity_amd/dist.py: shard bounds, one all-gather per pass, identical
reduction on every rank, owner-only update, log merge) can be exercised over gloo with world_size > 1.
Sequential mode only; the per-node arithmetic is tests/level_model.py's restatement.  Record layout =
ccsim_kernels.h XRec: word0 packed key ((score+1)<<40 | (2^40-1 - global idx)), word1 mt, word2 ma, word3 nfeas."""
import numpy as np

from cluster_capacity_amd import model as M
from level_model import LevelModel

IDX_BITS = 40
IDX_MASK = (1 << IDX_BITS) - 1


class CpuShardEngine:
    def __init__(self, nodes, pod, prof, global_offset, n_global):
        self.m = LevelModel(prof, nodes, pod)
        self.off, self.n_global, self.n = global_offset, n_global, nodes.n

    def dist_begin(self, max_limit, mode, n_ranks, rank, send, recv, log_cap=0):
        assert mode == "sequential"
        self.limit, self.world, self.rank, self.send, self.recv = max_limit, n_ranks, rank, send, recv
        self.mt = self.ma = 0
        self.placed, self.done = 0, 0
        self.per_node = np.zeros(self.n, np.int32)
        self.log = np.full(max(1, log_cap), -1, np.int32)

    def dist_scan(self):
        if self.done:
            return
        m = self.m
        key = mt = ma = nf = 0
        for n in range(self.n):
            if not m.feasible(n):
                continue
            nf += 1
            mt, ma = max(mt, m.cnt[n]), max(ma, m.aff[n])
            k = ((m.stat(n, self.mt, self.ma) + m.dyn(n) + 1) << IDX_BITS) | (IDX_MASK - (self.off + n))
            key = max(key, k)
        self.send[:4] = self.send.new_tensor([key, mt, ma, nf])

    def dist_decide(self):
        if self.done:
            return
        rec = self.recv.view(self.world, -1).numpy()
        key, mt, ma, nf = int(rec[:, 0].max()), int(rec[:, 1].max()), int(rec[:, 2].max()), int(rec[:, 3].sum())
        if key == 0:
            self.done = 1
        elif (mt, ma) != (self.mt, self.ma):
            self.mt, self.ma = mt, ma  # rescan with the right normalization constants
        else:
            g = IDX_MASK - (key & IDX_MASK)
            i = g - self.off
            if 0 <= i < self.n:  # only the owning rank updates
                self.m.apply(i)
                self.per_node[i] += 1
            if self.placed < len(self.log):
                self.log[self.placed] = g
            self.placed += 1
            if self.limit > 0 and self.placed >= self.limit:
                self.done = 2

    def dist_poll(self):
        return self.done, self.placed

    def dist_finish(self, want_log=False, log_cap=0):
        return M.RunResult(placed=self.placed, stop=M.STOP_LIMIT if self.done == 2 else M.STOP_UNSCHEDULABLE,
                           per_node_count=self.per_node, log=self.log[: self.placed] if want_log else None,
                           hist=np.zeros(M.NREASON, np.int64), hist_taintset=np.zeros(1, np.int64), n_code_unschedulable=0)
