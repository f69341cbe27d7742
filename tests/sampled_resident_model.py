"""TEST INFRASTRUCTURE: the argument behind csrc/ccsim_sampled.h (round 5) in plain Python -- the sampled search of
findNodesThatPassFilters (vendor/k8s.io/kubernetes/pkg/scheduler/schedule_one.go:610-723) for a template WITHOUT topology-coupled
plugins, answered from resident per-node words and per-block summaries instead of node passes.

What is resident (k_sb_build): memo[n] = TotalScore of node n under the ASSUMED normalization maxima (mt_a, ma_a), -1 = infeasible; per
block of B nodes: feasible count, best (score, lowest index) key, maxima of the two raw scores over the feasible nodes.
A cycle (k_sb_cycles), in the kernel's own steps:
  1. ring prefix of the blocks' feasible counts from the start block: which segment holds the (K+1)-th feasible node of the visiting
     order -- the start block behind the start index (mode 1), a whole block (mode 2), the start block before the start index (mode 3),
     or none (mode 0: fewer than K+1 feasible nodes, every node is visited);
  2. argmax and maxima over exactly the kept nodes: the two cut blocks node by node, every block in between by its summary (ties: the
     earliest visiting position -- inside a whole block that is its lowest index);
  3. maxima differ from the assumed ones -> everything is rebuilt under the true ones, the cycle is repeated;
  4. the winner's placement changes ONE memo word and its block's summary; the start index moves to the node the search stopped at.
Checked against the oracle's literal visiting loop in tests/test_sampled_resident_model.py: same log, same nodes visited per cycle, same
start index trajectory, and (the invariant the kernel rests on) every summary equals its recomputation from the memo at every use."""
from __future__ import annotations

from coupled_model import CoupledWindowModel
from sharded_sampled_model import num_feasible_nodes_to_find


class ResidentSampledModel:
    def __init__(self, prof, nodes, pod, block=256, check=True):
        assert not pod.spread and pod.ipa is None
        self.m = CoupledWindowModel(prof, nodes, pod, go_log=None, every_node_scored=False)
        self.N, self.B, self.check = nodes.n, block, check
        self.nb = -(-self.N // block)
        self.K = num_feasible_nodes_to_find(prof.percentage_of_nodes_to_score, self.N)
        # (a profile without any Score plugin searches with K = 1, schedule_one.go:619-621: the node model underneath wants a Score
        # plugin, so that case is the GPU suite's -- tests/test_sampling.py::test_profile_without_score_plugins_*)
        self.start, self.mt_a, self.ma_a = 0, 0, 0
        self.builds = 0
        self.build()

    # ---- k_sb_build
    def _word(self, n):
        return self.m.local_score(n, self.mt_a, self.ma_a) if self.m.node_feasible(n) else -1

    def _summary(self, b):
        fc, key, mt, ma = 0, None, 0, 0
        for n in range(b * self.B, min(self.N, (b + 1) * self.B)):
            if self.memo[n] >= 0:
                fc += 1
                k = (self.memo[n], -n)
                key = k if key is None or k > key else key
                mt, ma = max(mt, self.m.cnt[n]), max(ma, self.m.aff[n])
        return [fc, key, mt, ma]

    def build(self):
        self.builds += 1
        self.memo = [self._word(n) for n in range(self.N)]
        self.sm = [self._summary(b) for b in range(self.nb)]
        self.Ftotal = sum(s[0] for s in self.sm)

    def _ringpos(self, n):
        return n - self.start if n >= self.start else n + self.N - self.start

    # ---- one cycle of k_sb_cycles; returns (winner, visited) or None when no node is feasible; "rebuild" when the maxima moved
    def cycle(self):
        N, B, K, nb, S = self.N, self.B, self.K, self.nb, self.start
        if self.Ftotal == 0:
            return None
        sb = S // B
        if self.check:
            for b in range(nb):
                assert self.sm[b] == self._summary(b), ("stale summary", b)
            assert self.Ftotal == sum(s[0] for s in self.sm)
        blk = range(sb * B, min(N, (sb + 1) * B))
        tail = [n for n in blk if n >= S and self.memo[n] >= 0]
        head = [n for n in blk if n < S and self.memo[n] >= 0]
        ring = [(sb + r) % nb for r in range(1, nb)]
        fullF = sum(self.sm[b][0] for b in ring)
        everything = self.Ftotal <= K
        mode = 0 if everything else (1 if len(tail) >= K + 1 else (2 if len(tail) + fullF >= K + 1 else 3))
        best, mt, ma, stop = None, 0, 0, None

        def take_node(n):
            nonlocal best, mt, ma
            k = (self.memo[n], -self._ringpos(n))
            best = (k, n) if best is None or k > best[0] else best
            mt, ma = max(mt, self.m.cnt[n]), max(ma, self.m.aff[n])

        for rank, n in enumerate(tail):
            if mode != 1 or rank < K:
                take_node(n)
            elif rank == K:
                stop = n
        if mode in (0, 3):
            need = K - (len(tail) + fullF)
            for rank, n in enumerate(head):
                if mode == 0 or rank < need:
                    take_node(n)
                elif rank == need:
                    stop = n
        if mode != 1:
            run = len(tail)
            for b in ring:
                fc, key, bmt, bma = self.sm[b]
                if mode in (0, 3) or run + fc <= K:
                    if key is not None:  # the block's best node: highest score, lowest index = earliest position of a whole block
                        n = -key[1]
                        k = (key[0], -self._ringpos(n))
                        best = (k, n) if best is None or k > best[0] else best
                        mt, ma = max(mt, bmt), max(ma, bma)
                elif run <= K:  # the block the stretch ends in, node by node
                    need = K - run
                    feas = [n for n in range(b * B, min(N, (b + 1) * B)) if self.memo[n] >= 0]
                    for rank, n in enumerate(feas):
                        if rank < need:
                            take_node(n)
                        elif rank == need:
                            stop = n
                run += fc
        if (mt, ma) != (self.mt_a, self.ma_a):
            self.mt_a, self.ma_a = mt, ma
            self.build()
            return "rebuild"
        g = best[1]
        visited = N if everything else self._ringpos(stop)
        if not everything:
            self.start = stop
        self.m.place(g)
        self.memo[g] = self._word(g)
        if self.memo[g] < 0:
            self.Ftotal -= 1
        self.sm[g // B] = self._summary(g // B)
        return g, visited

    def run(self, limit=0):
        log, visited_total, starts = [], 0, []
        while True:
            starts.append(self.start)
            r = self.cycle()
            if r == "rebuild":
                starts.pop()
                continue
            if r is None:
                return log, "Unschedulable", visited_total + self.N, starts
            log.append(r[0])
            visited_total += r[1]
            if limit and len(log) >= limit:
                return log, "LimitReached", visited_total, starts
