"""Executable specification of the SAMPLED search (percentageOfNodesToScore < 100, schedule_one.go:610-723) on node-range SHARDS --
TEST INFRASTRUCTURE: the protocol specification from which the engine's sharded sampled search was built (round 3: ccsim_kernels.h
DevState::smp_phase, k_decide; tests/test_sampling.py::test_sampled_search_on_shards_*).  The engine folds exchange 2 into exchange 3 --
the maxima over the selected nodes are ASSUMED by the scoring pass and verified on the gathered winner records, as in its unsharded
mode, and the cancelling node's position rides on the same record -- so a cycle costs two all-gathers instead of three.
R ranks own contiguous node ranges; every exchange below is one fixed-size record per rank, all-gathered (the engine's 256-byte
exchange record has room for each of them), and every rank derives the same decisions from the gathered records.

One cycle = three exchanges:
  1. COUNT   each rank filters its nodes and reports how many are feasible before and after the rotating start index within its
             range.  From the gathered counts every rank knows, in visiting order (start, start+1, ... wrapping), which prefix of
             each shard's feasible nodes falls among the first K, and in which shard the (K+1)-th feasible node lies -- that node
             cancels the search (:655-662), so the number of VISITED nodes (which moves nextStartNodeIndex, :538-539) is its
             visiting position; only its owner can name that position, so it rides on exchange 2.
  2. MAXIMA  each rank reports the TaintToleration / NodeAffinity maxima over ITS selected nodes (normalize_score.go:28-56 needs the
             maxima over the K selected nodes only) and, if it owns it, the visiting position of the cancelling node.
  3. WINNER  each rank scores its selected nodes under the global maxima and reports its best (TotalScore, earliest visiting
             position); the global maximum decides, ties go to the earliest position (the canonical selectHost of the oracle); the
             owner applies the placement, everybody advances the start index by the visited count.
No rank ever needs another rank's node columns.  Topology-coupled plugins are out of this model (their PreFilter state covers all
nodes and is already replicated by the engine's table all-reduce; Filter / Score would then run on the selected nodes only).

Checked against the oracle's literal visiting loop in tests/test_sharded_sampled_model.py: same log, same stop, same start index
trajectory, for 1 .. 5 shards."""
from __future__ import annotations

from coupled_model import CoupledWindowModel


def num_feasible_nodes_to_find(pct, n):  # schedule_one.go:697-723
    if n < 100:
        return n
    if pct == 0:
        pct = max(5, 50 - n // 125)
    return max(100, n * pct // 100)


class ShardedSampledModel:
    def __init__(self, prof, nodes, pod, ranks):
        assert not pod.spread and pod.ipa is None
        self.m = CoupledWindowModel(prof, nodes, pod, go_log=None, every_node_scored=False)
        self.N, self.R = nodes.n, ranks
        per = -(-self.N // ranks)
        self.bounds = [(min(self.N, r * per), min(self.N, r * per + per)) for r in range(ranks)]
        self.K = num_feasible_nodes_to_find(prof.percentage_of_nodes_to_score, self.N)
        self.start = 0
        self.exchanges = 0

    def _position(self, n):  # place of node n in this cycle's visiting order
        return (n - self.start) % self.N

    def run(self, limit=0):
        m, N, K = self.m, self.N, self.K
        log, starts = [], []
        while True:
            starts.append(self.start)
            # ---- exchange 1: feasible counts per shard, split at the start index ----
            feas = [[n for n in range(lo, hi) if m.node_feasible(n)] for lo, hi in self.bounds]  # (each rank: its own list)
            rec1 = [(sum(1 for n in f if n >= self.start), sum(1 for n in f if n < self.start)) for f in feas]
            self.exchanges += 1
            # every rank, from rec1 alone: the visiting order crosses the shards as  [start-part of r0.. rR-1] then [wrapped part of r0 ..]
            segments = [(r, True) for r in range(self.R)] + [(r, False) for r in range(self.R)]  # (shard, after-start part?)
            seg_count = [rec1[r][0] if after else rec1[r][1] for r, after in segments]
            total = sum(seg_count)
            if total == 0:
                return log, "Unschedulable", starts
            take, left = [], K
            cancel_seg, cancel_rank_in_seg = None, 0  # which segment holds the (K+1)-th feasible node, and its rank within the segment
            for s, c in enumerate(seg_count):
                t = min(c, left)
                take.append(t)
                left -= t
                if left == 0 and cancel_seg is None and c > t:
                    cancel_seg, cancel_rank_in_seg = s, t
                elif left == 0 and cancel_seg is None and c == t:
                    # the K-th feasible node ended this segment exactly: the (K+1)-th is the first feasible node of a later segment
                    for s2 in range(s + 1, len(seg_count)):
                        if seg_count[s2] > 0:
                            cancel_seg, cancel_rank_in_seg = s2, 0
                            break
                    if cancel_seg is None:
                        cancel_seg = -1  # fewer than K+1 feasible nodes: every node is visited
            if cancel_seg is None:
                cancel_seg = -1
            # ---- each rank: its selected nodes (a prefix of each of its two segments, in index order) ----
            selected = []
            for r in range(self.R):
                after = [n for n in feas[r] if n >= self.start][: take[r]]
                before = [n for n in feas[r] if n < self.start][: take[self.R + r]]
                selected.append(after + before)
            # ---- exchange 2: maxima over the selected nodes + the cancelling node's position from its owner ----
            rec2 = []
            for r in range(self.R):
                mt = max((m.cnt[n] for n in selected[r]), default=0)
                ma = max((m.aff[n] for n in selected[r]), default=0)
                pos = -1
                if cancel_seg >= 0 and segments[cancel_seg][0] == r:
                    part = [n for n in feas[r] if (n >= self.start) == segments[cancel_seg][1]]
                    pos = self._position(part[cancel_rank_in_seg])
                rec2.append((mt, ma, pos))
            self.exchanges += 1
            mt, ma = max(x[0] for x in rec2), max(x[1] for x in rec2)
            cancel_pos = max(x[2] for x in rec2)
            visited = cancel_pos if cancel_seg >= 0 else N  # nodes processed before the search was cancelled (:538-539: nf + failed)
            # ---- exchange 3: each rank's best (score, earliest position) ----
            rec3 = []
            for r in range(self.R):
                best = None
                for n in selected[r]:
                    key = (m.local_score(n, mt, ma), -self._position(n))
                    if best is None or key > best[0]:
                        best = (key, n)
                rec3.append(best)
            self.exchanges += 1
            winner = max((b for b in rec3 if b is not None), key=lambda b: b[0])[1]
            log.append(winner)
            m.place(winner)  # (its owner only)
            self.start = (self.start + visited) % N
            if limit and len(log) >= limit:
                return log, "LimitReached", starts
