"""TEST INFRASTRUCTURE: the argument behind k_sb_laps (csrc/ccsim_sampled.h, round 6) in plain Python -- the sampled search of
findNodesThatPassFilters (vendor/k8s.io/kubernetes/pkg/scheduler/schedule_one.go:610-723) for a template WITHOUT topology-coupled
plugins, a LAP of the ring at a time.

A cycle keeps the first K feasible nodes of the visiting order from nextStartNodeIndex, stops at the (K+1)-th, and the NEXT cycle starts
exactly there (:538-539 with :655-662).  So with F feasible nodes at ring ranks 0 .. F-1 from the start index S, cycle j of the lap
keeps ranks [jK, (j+1)K) and stops at rank (j+1)K -- as long as that rank exists before the ring is back at S, i.e. for
j < J = (F-1) // K.  The winner of a cycle lies inside its own stretch, a placement changes that one node only, and the two
normalization maxima are taken over the own K nodes: the J cycles of a lap read disjoint nodes that no earlier cycle of the lap has
written.  They are independent and are evaluated side by side from the state at the start of the lap:
  1. ring prefix of the blocks' feasible counts from the start block: every multiple of K falls into one segment (a whole block, or the
     start block's part before the start index) -- that block is CUT by a stretch boundary; all others lie wholly inside one stretch;
  2. whole blocks contribute their summary to their stretch; cut blocks are read node by node: the feasible nodes before the boundary go
     to the stretch that ends there, the boundary node and the ones behind it to the next (ties: earliest visiting position);
  3. a stretch whose kept nodes' maxima differ from the assumed ones is re-evaluated node by node under its own maxima (the score under
     other maxima follows from the memo word and the static word); when such stretches cover more than a quarter of the ring the
     first of them ends the lap instead: the stretches before it are committed, everything is rebuilt under its maxima (as the
     one-cycle form does on every mismatch); the limit ends a lap likewise;
  4. commits: one node each, disjoint; the winners' block summaries are recomputed; the start index moves to the last committed
     stretch's stop node.
F <= K is the degenerate lap of ONE stretch without a boundary: every node is visited, the start index stays.
Needs K >= block size (a segment then holds at most one boundary); the engine picks the block size accordingly (K >= 100 whenever the
search samples with a Score plugin, schedule_one.go:697-723) and keeps the one-cycle form for K < 64 (a profile without Score plugins).
Checked against the oracle's literal visiting loop in tests/test_sampled_lap_model.py."""
from __future__ import annotations

from sampled_resident_model import ResidentSampledModel


class LapSampledModel(ResidentSampledModel):
    def __init__(self, prof, nodes, pod, block=64, check=True, max_stretches=63, slow_floor=65536):
        super().__init__(prof, nodes, pod, block=block, check=check)
        self.slow_floor, self.slow_stretches = slow_floor, 0
        assert self.K >= block, "a segment may hold one stretch boundary at most"
        self.JMAX = max_stretches
        self.laps = 0
        self.stretches_per_lap = []

    def _static(self, n, mt, ma):
        """the part of TotalScore that depends on the normalization maxima (csrc/ccsim_kernels.h static_score, without the image term)"""
        p, t = self.m.prof, 0
        if p.w_taint:
            t += (100 if mt == 0 else 100 - (100 * self.m.cnt[n]) // mt) * p.w_taint
        if p.w_nodeaffinity and self.m.pod.preferred:
            t += (0 if ma == 0 else (100 * self.m.aff[n]) // ma) * p.w_nodeaffinity
        return t

    def lap(self, limit_left):
        """-> list of (winner, visited) of the committed cycles, flag in {"ok", "rebuild", None (no feasible node)}"""
        N, B, K, nb, S = self.N, self.B, self.K, self.nb, self.start
        F = self.Ftotal
        if F == 0:
            return [], None
        if self.check:
            for b in range(nb):
                assert self.sm[b] == self._summary(b), ("stale summary", b)
            assert F == sum(s[0] for s in self.sm)
        sb = S // B
        everything = F <= K
        J = 1 if everything else min(self.JMAX, (F - 1) // K)
        # ---- segments in ring order: the start block behind the start index, the other blocks, the start block before the start index
        start_blk = range(sb * B, min(N, (sb + 1) * B))
        tailF = sum(1 for n in start_blk if n >= S and self.memo[n] >= 0)
        headF = self.sm[sb][0] - tailF
        best = [None] * J  # per stretch: ((score, -ringpos), node)
        mx = [[0, 0] for _ in range(J)]
        starts = [S] + [None] * J  # starts[j] = first node of stretch j; starts[J] = where the lap's last stretch stopped

        def take(j, n):
            k = (self.memo[n], -self._ringpos(n))
            if best[j] is None or k > best[j][0]:
                best[j] = (k, n)
            mx[j][0], mx[j][1] = max(mx[j][0], self.m.cnt[n]), max(mx[j][1], self.m.aff[n])

        def cut(nodes_in_segment, j, need):
            """a segment holding the boundary between stretch j - 1 and stretch j: `need` feasible nodes of it come before the boundary"""
            rank = 0
            for n in nodes_in_segment:
                if self.memo[n] < 0:
                    continue
                if rank < need:
                    take(j - 1, n)
                else:
                    if rank == need:
                        starts[j] = n
                    if j < J:
                        take(j, n)
                rank += 1

        for n in start_blk:  # cut 0: the start index itself
            if n >= S and self.memo[n] >= 0:
                take(0, n)
        run = tailF
        assert everything or tailF <= K
        for r in range(1, nb):
            b = (sb + r) % nb
            fc, key, bmt, bma = self.sm[b]
            if fc == 0:
                continue
            jt = 0 if everything else max(1, -(-run // K))  # the first boundary at or behind this block's first feasible node
            if not everything and jt * K < run + fc:  # rank jt * K lies in this block: it is cut (K >= B: by one boundary only)
                if jt <= J:
                    cut(range(b * B, min(N, (b + 1) * B)), jt, jt * K - run)
            else:
                j = 0 if everything else run // K
                if j < J:
                    n = -key[1]
                    k = (key[0], -self._ringpos(n))
                    if best[j] is None or k > best[j][0]:
                        best[j] = (k, n)
                    mx[j][0], mx[j][1] = max(mx[j][0], bmt), max(mx[j][1], bma)
            run += fc
        head = [n for n in start_blk if n < S]
        if everything:
            for n in head:
                if self.memo[n] >= 0:
                    take(0, n)
        elif run <= J * K < run + headF:
            cut(head, J, J * K - run)
        # ---- which stretches stand.  A stretch whose kept nodes' maxima differ from the assumed ones has been scored with the wrong
        # normalization: it is re-evaluated node by node under its own maxima (TotalScore = static part + state-dependent part, so
        # the score under other maxima follows from the memo word and the node's static word) -- unless such stretches cover so much
        # of the ring that rebuilding everything under the first one's maxima is cheaper: then the stretches before it are committed
        # and the lap ends there (what the one-cycle form does on every mismatch).
        jc, flag, scans = J, "ok", J
        if limit_left and jc >= limit_left:  # the limit ends the run before the stretch behind it is looked at
            jc = scans = limit_left
        mism = [j for j in range(jc) if tuple(mx[j]) != (self.mt_a, self.ma_a)]

        def span(j):
            return N if everything else self._ringpos(starts[j + 1]) - self._ringpos(starts[j])

        if mism and sum(span(j) for j in mism) > max(self.slow_floor, N // 4):
            jc, flag = mism[0], "rebuild"
            scans = jc + 1
            new_maxima = tuple(mx[jc])
        else:
            for j in mism:
                mt, ma = mx[j]
                best[j] = None
                first = starts[j]
                for d in range(span(j)):
                    n = (first + d) % N
                    if self.memo[n] < 0:
                        continue
                    sc = self.memo[n] - self._static(n, self.mt_a, self.ma_a) + self._static(n, mt, ma)
                    assert sc == self.m.local_score(n, mt, ma)
                    k = (sc, -self._ringpos(n))
                    if best[j] is None or k > best[j][0]:
                        best[j] = (k, n)
                self.slow_stretches += 1
        out = []
        for j in range(jc):
            g = best[j][1]
            if everything:
                visited = N
            else:
                assert starts[j + 1] is not None
                visited = self._ringpos(starts[j + 1]) - self._ringpos(starts[j])
            out.append((g, visited))
        for g, _ in out:
            self.m.place(g)
            self.memo[g] = self._word(g)
            if self.memo[g] < 0:
                self.Ftotal -= 1
        for g, _ in out:
            self.sm[g // B] = self._summary(g // B)
        if not everything and jc > 0:
            self.start = starts[jc]
        if flag == "rebuild":
            self.mt_a, self.ma_a = new_maxima
            self.build()
        self.laps += 1
        self.stretches_per_lap.append(len(out))
        self.scans = getattr(self, "scans", 0) + scans
        return out, flag

    def run(self, limit=0):
        log, visited_total = [], 0
        while True:
            out, flag = self.lap(limit - len(log) if limit else 0)
            if flag is None:
                return log, "Unschedulable", visited_total + self.N, []
            for g, v in out:
                log.append(g)
                visited_total += v
            if limit and len(log) >= limit:
                return log, "LimitReached", visited_total, []
