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


class ShardedSampledCoupledModel:
    """Round 6: the same protocol for a template WITH topology-coupled plugins (hard / soft spread constraints, inter-pod affinity) --
    TEST INFRASTRUCTURE, the specification of what csrc/ccsim_kernels.h does on shards since the refusal of `percentageOfNodesToScore <
    100 + coupled plugins + several GPUs` was lifted (k_scan<PTS, SMP>, final_body, k_decide):

      * the plugins' PreFilter state (match counts per domain, inter-pod topology-pair counts) covers ALL nodes of the cluster and is
        REPLICATED -- every rank applies every winner's clone to its own copy (`add_clone_to_tables`), so Filter verdicts need no exchange;
      * COUNT: a rank's feasible nodes are those that pass the node-local filters AND the coupled Filter under the replicated tables;
      * the PreScore facts cover the SELECTED nodes only (the reference's PreScore sees `filteredNodes`, schedule_one.go:757-790):
        candidate domain sets and the count of non-ignored nodes per soft constraint (-> the log weights), the range of the raw spread
        scores under those weights, the range of the raw inter-pod scores -- each rank reports its own selected nodes' share, sets unite,
        counts add, ranges combine (the engine ASSUMES them in the scoring pass and verifies them on the gathered records; here they
        are explicit exchanges);
      * WINNER as before.

    `all_gather(record) -> [records in rank order]`: None = every rank's share is computed in this process; a callable = one real
    exchange per call (tests/test_dist_gloo.py: gloo process groups, each rank computing its own share only).  Every rank keeps the whole
    cluster's state, as tests/coupled_model.py::ShardedCoupledWindowModel does: what is checked is the protocol."""

    def __init__(self, prof, nodes, pod, go_log, ranks, rank=0, all_gather=None):
        self.m = CoupledWindowModel(prof, nodes, pod, go_log=go_log, every_node_scored=False)
        self.p = prof
        self.N, self.R, self.rank = nodes.n, ranks, rank
        per = -(-self.N // ranks)
        self.bounds = [(min(self.N, r * per), min(self.N, r * per + per)) for r in range(ranks)]
        has_score = any((prof.w_taint, prof.w_nodeaffinity, prof.w_fit, prof.w_balanced, prof.w_topologyspread, prof.w_interpodaffinity, prof.w_imagelocality))
        self.K = num_feasible_nodes_to_find(prof.percentage_of_nodes_to_score, self.N) if has_score else 1
        self.start = 0
        self.exchanges = 0
        self._all_gather = all_gather

    def exchange(self, share):
        """share(r) -> rank r's record; returns the records of all ranks in rank order"""
        self.exchanges += 1
        if self._all_gather is None:
            return [share(r) for r in range(self.R)]
        return self._all_gather(share(self.rank))

    def run(self, limit=0):
        m, N, K, p = self.m, self.N, self.K, self.p
        log, visited_per_cycle = [], []
        pos = lambda n: (n - self.start) % N
        while True:
            T = m.build_tables()
            minima = m.hard_minima(T)
            mine = lambda r: [n for n in range(*self.bounds[r]) if m.node_feasible(n) and m.coupled_filter(T, minima, n)]
            # ---- exchange 1: feasible counts per shard, split at the start index
            rec1 = self.exchange(lambda r: (sum(1 for n in mine(r) if n >= self.start), sum(1 for n in mine(r) if n < self.start)))
            segments = [(r, True) for r in range(self.R)] + [(r, False) for r in range(self.R)]
            seg_count = [rec1[r][0] if after else rec1[r][1] for r, after in segments]
            if sum(seg_count) == 0:
                visited_per_cycle.append(N)
                return log, "Unschedulable", visited_per_cycle
            take, left, cancel = [], K, None  # cancel: (segment, rank within the segment) of the (K+1)-th feasible node
            for s, c in enumerate(seg_count):
                t = min(c, left)
                take.append(t)
                left -= t
                if left == 0 and cancel is None:
                    if c > t:
                        cancel = (s, t)
                    else:
                        nxt = [s2 for s2 in range(s + 1, len(seg_count)) if seg_count[s2] > 0]
                        cancel = (nxt[0], 0) if nxt else (-1, 0)
            if cancel is None:
                cancel = (-1, 0)

            def selected(r):
                f = mine(r)
                return [n for n in f if n >= self.start][: take[r]] + [n for n in f if n < self.start][: take[self.R + r]]

            # ---- exchange 2: over each rank's selected nodes -- normalization maxima, the cancelling node's position, the PreScore sets
            def facts(r):
                sel = selected(r)
                cp = -1
                if cancel[0] >= 0 and segments[cancel[0]][0] == r:
                    part = [n for n in mine(r) if (n >= self.start) == segments[cancel[0]][1]]
                    cp = pos(part[cancel[1]])
                doms = {i: sorted({m.sdom[i][n] for n in sel if m.soft_keys[n]}) for i in m.soft if not m.spread[i].is_hostname}
                return (max((m.cnt[n] for n in sel), default=0), max((m.aff[n] for n in sel), default=0), cp, len(sel), sum(1 for n in sel if not m.soft_keys[n]), doms)
            rec2 = self.exchange(facts)
            mt, ma = max(x[0] for x in rec2), max(x[1] for x in rec2)
            cancel_pos = max(x[2] for x in rec2)
            visited = cancel_pos if cancel[0] >= 0 else N
            nf, n_ignored = sum(x[3] for x in rec2), sum(x[4] for x in rec2)
            weights = {}
            for i in m.soft:
                sz = nf - n_ignored if m.spread[i].is_hostname else len(set().union(*[set(x[5][i]) for x in rec2]))
                weights[i] = m.go_log(float(sz + 2))
            # ---- exchange 2b: the ranges of the raw coupled scores over the selected nodes (scoring.go:238-265, interpodaffinity/scoring.go:258-289)
            ipa_on = m.ipa is not None and p.w_interpodaffinity and T["entries"] > 0

            def ranges(r):
                sel = selected(r)
                rp = [m.raw_pts(T, weights, n) for n in sel if m.soft_keys[n]] if m.soft else []
                ri = [m.raw_ipa(T, n) for n in sel] if ipa_on else []
                return (min(rp, default=None), max(rp, default=None), min(ri, default=None), max(ri, default=None))
            rec2b = self.exchange(ranges)
            cmb = lambda k, f: (lambda v: f(v) if v else None)([x[k] for x in rec2b if x[k] is not None])
            p_lo, p_hi, i_lo, i_hi = cmb(0, min), cmb(1, max), cmb(2, min), cmb(3, max)

            def total(n):
                t = m.local_score(n, mt, ma)
                if m.soft:
                    if not m.soft_keys[n]:
                        s = 0
                    elif p_hi == 0:
                        s = 100
                    else:
                        s = 100 * (p_hi + p_lo - m.raw_pts(T, weights, n)) // p_hi
                    t += s * p.w_topologyspread
                if ipa_on:
                    t += (int(100.0 * (float(m.raw_ipa(T, n) - i_lo) / float(i_hi - i_lo))) if i_hi > i_lo else 0) * p.w_interpodaffinity
                return t

            # ---- exchange 3: each rank's best (score, earliest visiting position)
            def best(r):
                keys = [((total(n), -pos(n)), n) for n in selected(r)]
                return max(keys) if keys else None
            rec3 = self.exchange(best)
            winner = max(b for b in rec3 if b is not None)[1]
            log.append(winner)
            visited_per_cycle.append(visited)
            m.place(winner)  # (its owner: the node columns; every rank: the replicated tables, rebuilt from the clones at the next cycle)
            self.start = (self.start + visited) % N
            if limit and len(log) >= limit:
                return log, "LimitReached", visited_per_cycle
