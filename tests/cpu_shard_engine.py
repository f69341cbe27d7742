"""TEST INFRASTRUCTURE: a CPU stand-in for one shard's engine, speaking the multi-GPU stepping protocol of
include/ccsim.h (dist_begin / dist_scan / dist_decide / dist_poll / dist_finish) on CPU tensors, so that the
host-side sharding logic (cluster-capacity_amd/dist.py: shard bounds, one all-gather per pass, identical
reduction on every rank, owner-only update, log merge) can be exercised over gloo with world_size > 1.
The per-node arithmetic is tests/level_model.py's restatement.  Record layout = ccsim_kernels.h XRec: word0 packed key
((score+1)<<40 | (2^40-1 - global idx)), word1 mt, word2 ma, word3 nfeas; the batched mode adds c_mt, c_ma, committed,
n_top, T, e_mt, e_ma, cut_mt, cut_ma (words 4..12) and follows the level / plan / cut state machine of
ccsim_level.h (level_decide, k_level_final, k_level_decide): one exchange per score level, every rank reduces the
gathered records identically, placements of a level are ordered by rank (shards are contiguous node ranges).
The sampled search (percentageOfNodesToScore < 100) follows the engine's two-phase form (ccsim_kernels.h DevState::smp_phase, k_decide):
a counting pass whose record carries this shard's feasible nodes (all / before the start index; words 13, 14 here, XRec::pad in the
engine), then the scoring pass over the nodes whose rank in the rotating visiting order is below K, with the visiting position of the
node of rank K (it cancels the search) riding on its owner's record (word 13)."""
import numpy as np

from cluster_capacity_amd import model as M
from level_model import LevelModel

IDX_BITS = 40
IDX_MASK = (1 << IDX_BITS) - 1
NO_CUT = 1 << 62


class CpuShardEngine:
    def __init__(self, nodes, pod, prof, global_offset, n_global):
        self.m = LevelModel(prof, nodes, pod)
        self.off, self.n_global, self.n = global_offset, n_global, nodes.n

    def dist_begin(self, max_limit, mode, n_ranks, rank, send, recv, log_cap=0):
        assert mode in ("sequential", "batched")
        self.mode, self.want_log = mode, log_cap > 0
        # batched mode (level_decide's state): the level the next pass commits / measures
        self.lvl_valid = self.lvl_plan_only = self.lvl_prefix = False
        self.lvl_M, self.lvl_cut, self.lvl_remaining, self.lvl_rank_prefix, self.lvl_c_mt, self.lvl_c_ma = 0, NO_CUT, NO_CUT, 0, 0, 0
        self.limit, self.world, self.rank, self.send, self.recv = max_limit, n_ranks, rank, send, recv
        self.mt = self.ma = 0
        self.placed, self.done = 0, 0
        # sampled search of the sequential mode (schedule_one.go:610-723)
        p = self.m.prof
        scoring = any((p.w_taint, p.w_nodeaffinity, p.w_fit, p.w_balanced))
        N, pct = self.n_global, p.percentage_of_nodes_to_score
        k = N if N < 100 else max(100, N * (pct if pct else max(5, 50 - N // 125)) // 100)
        self.smp_K = (k if scoring else 1) if (k if scoring else 1) < N and mode == "sequential" else 0
        self.smp_start = self.smp_phase = self.smp_off = self.smp_Ftotal = self.smp_Fs = 0
        self.per_node = np.zeros(self.n, np.int32)
        self.log = np.full(max(1, log_cap), -1, np.int32)

    # ---- batched mode -------------------------------------------------------------------------------------------
    def _level_nodes(self):
        m = self.m
        return [n for n in range(self.n) if m.feasible(n) and m.stat(n, self.mt, self.ma) + m.dyn(n) == self.lvl_M
                and self.off + n <= self.lvl_cut]

    def _scan_batched(self):
        m = self.m
        committed = T = e_mt = e_ma = 0
        cut_mt = cut_ma = -1
        if self.lvl_plan_only:  # measure the level: run-down lengths, exhausted holders of the normalization maxima
            for n in self._level_nodes():
                j, f = m.run_down(n, m.stat(n, self.mt, self.ma), self.lvl_M, 1 << 30)
                for _ in range(j):
                    m.apply(n, -1)
                T += j
                if not f:
                    if self.mt > 0 and m.cnt[n] == self.mt:
                        e_mt, cut_mt = e_mt + 1, max(cut_mt, self.off + n)
                    if self.ma > 0 and m.aff[n] == self.ma:
                        e_ma, cut_ma = e_ma + 1, max(cut_ma, self.off + n)
        elif self.lvl_valid:  # commit it: blindly, or (ordered) with this rank's prefix of the level's placement sequence
            pos = self.lvl_rank_prefix
            for n in self._level_nodes():
                stat = m.stat(n, self.mt, self.ma)
                j, _ = m.run_down(n, stat, self.lvl_M, 1 << 30)
                took = j
                if self.lvl_prefix:
                    took = max(0, min(j, self.lvl_remaining - pos))
                    for q in range(took):
                        if self.placed + pos + q < len(self.log):
                            self.log[self.placed + pos + q] = self.off + n
                    pos += j
                for _ in range(j - took):
                    m.apply(n, -1)
                self.per_node[n] += took
                committed += took
        rec = [0] * 16
        rec[6], rec[8], rec[9], rec[10], rec[11], rec[12] = committed, T, e_mt, e_ma, cut_mt, cut_ma
        if not self.lvl_plan_only:  # the next level, from the state the commit left
            key = mt = ma = nf = c_mt = c_ma = n_top = 0
            top = -1
            for n in range(self.n):
                if not m.feasible(n):
                    continue
                nf += 1
                if m.cnt[n] > mt:
                    mt, c_mt = m.cnt[n], 1
                elif m.cnt[n] == mt:
                    c_mt += 1
                if m.aff[n] > ma:
                    ma, c_ma = m.aff[n], 1
                elif m.aff[n] == ma:
                    c_ma += 1
                sc = m.stat(n, self.mt, self.ma) + m.dyn(n)
                if sc > top:
                    top, n_top = sc, 1
                elif sc == top:
                    n_top += 1
                key = max(key, ((sc + 1) << IDX_BITS) | (IDX_MASK - (self.off + n)))
            rec[0], rec[1], rec[2], rec[3], rec[4], rec[5], rec[7] = key, mt, ma, nf, c_mt, c_ma, n_top
        self.send[:16] = self.send.new_tensor(rec)

    def _decide_batched(self):
        rec = self.recv.view(self.world, -1).numpy()
        key = int(rec[:, 0].max())
        mt, ma = int(rec[:, 1].max()), int(rec[:, 2].max())
        c_mt = int(sum(r[4] for r in rec if r[1] == mt))
        c_ma = int(sum(r[5] for r in rec if r[2] == ma))
        top = (key >> IDX_BITS) - 1 if key else -1
        n_top = int(sum(r[7] for r in rec if r[0] and (int(r[0]) >> IDX_BITS) - 1 == top))
        committed, T = int(rec[:, 6].sum()), int(rec[:, 8].sum())
        e_mt, e_ma, cut_mt, cut_ma = int(rec[:, 9].sum()), int(rec[:, 10].sum()), int(rec[:, 11].max()), int(rec[:, 12].max())
        if self.lvl_plan_only:  # level_decide: the pass measured the level, now commit it carefully
            self.lvl_plan_only, self.lvl_valid = False, True
            cut = NO_CUT
            if self.mt > 0 and e_mt == self.lvl_c_mt:
                cut = min(cut, cut_mt)
            if self.ma > 0 and e_ma == self.lvl_c_ma:
                cut = min(cut, cut_ma)
            self.lvl_cut = cut
            self.lvl_remaining = self.limit - self.placed if self.limit > 0 else NO_CUT
            self.lvl_prefix = self.want_log or (self.limit > 0 and self.placed + T > self.limit)
            self.lvl_rank_prefix = int(rec[: self.rank, 8].sum())  # contiguous shards: lower ranks' placements come first
            return
        self.placed += committed
        self.lvl_valid = False
        if self.limit > 0 and self.placed >= self.limit:
            self.done = 2
        elif key == 0:
            self.done = 1
        elif (mt, ma) != (self.mt, self.ma):
            self.mt, self.ma = mt, ma  # the scores above used stale constants: next pass rescans
        else:
            self.lvl_M, self.lvl_c_mt, self.lvl_c_ma = top, c_mt, c_ma
            self.lvl_cut, self.lvl_remaining, self.lvl_prefix, self.lvl_rank_prefix = NO_CUT, NO_CUT, False, 0
            plan = self.want_log or self.limit > 0 or (mt > 0 and n_top >= c_mt) or (ma > 0 and n_top >= c_ma)
            self.lvl_plan_only, self.lvl_valid = plan, not plan

    def dist_scan(self):
        if self.done:
            return
        if self.mode == "batched":
            return self._scan_batched()
        m = self.m
        if self.smp_K:
            return self._scan_sampled()
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
        if self.mode == "batched":
            return self._decide_batched()
        rec = self.recv.view(self.world, -1).numpy()
        if self.smp_K:
            return self._decide_sampled(rec)
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

    # ---- sampled search on shards: counting pass, exchange, scoring pass, exchange ------------------------------------
    def _scan_sampled(self):
        m, N, S = self.m, self.n_global, self.smp_start
        feas = [n for n in range(self.n) if m.feasible(n)]
        self.send.zero_()
        if self.smp_phase == 0:
            self.send[13], self.send[14] = len(feas), sum(1 for n in feas if self.off + n < S)
            return
        key = mt = ma = nf = 0
        stop = -1
        for f_local, n in enumerate(feas):  # F(i): feasible nodes before i in INDEX order, over all shards
            g, F = self.off + n, self.smp_off + f_local
            rank = F - self.smp_Fs if g >= S else F + self.smp_Ftotal - self.smp_Fs
            vpos = g - S if g >= S else g + N - S
            if rank == self.smp_K:
                stop = vpos  # the node that cancels the search (schedule_one.go:655-662)
            if rank >= self.smp_K:
                continue
            nf += 1
            mt, ma = max(mt, m.cnt[n]), max(ma, m.aff[n])
            key = max(key, ((m.stat(n, self.mt, self.ma) + m.dyn(n) + 1) << IDX_BITS) | (IDX_MASK - vpos))
        self.send[:4] = self.send.new_tensor([key, mt, ma, nf])
        self.send[13] = stop

    def _decide_sampled(self, rec):
        N = self.n_global
        if self.smp_phase == 0:
            tot = rec[:, 13]
            self.smp_Ftotal, self.smp_Fs, self.smp_off = int(tot.sum()), int(rec[:, 14].sum()), int(tot[: self.rank].sum())
            if self.smp_Ftotal == 0:
                self.done = 1
            else:
                self.smp_phase = 1
            return
        key, mt, ma, stop = int(rec[:, 0].max()), int(rec[:, 1].max()), int(rec[:, 2].max()), int(rec[:, 13].max())
        if (mt, ma) != (self.mt, self.ma):
            self.mt, self.ma = mt, ma  # the maxima over the SELECTED nodes moved: the scoring pass again
            return
        g = (IDX_MASK - (key & IDX_MASK) + self.smp_start) % N  # the key carries the visiting position
        i = g - self.off
        if 0 <= i < self.n:
            self.m.apply(i)
            self.per_node[i] += 1
        if self.placed < len(self.log):
            self.log[self.placed] = g
        self.placed += 1
        visited = N if self.smp_Ftotal <= self.smp_K else stop
        self.smp_start = (self.smp_start + visited) % N
        self.smp_phase = 0
        if self.limit > 0 and self.placed >= self.limit:
            self.done = 2

    def dist_poll(self):
        return self.done, self.placed

    def dist_finish(self, want_log=False, log_cap=0):
        return M.RunResult(placed=self.placed, stop=M.STOP_LIMIT if self.done == 2 else M.STOP_UNSCHEDULABLE,
                           per_node_count=self.per_node, log=self.log[: self.placed] if want_log else None,
                           hist=np.zeros(M.NREASON, np.int64), hist_taintset=np.zeros(1, np.int64), n_code_unschedulable=0)
