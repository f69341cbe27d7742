"""TEST INFRASTRUCTURE: the argument behind k_sf_cycles (csrc/ccsim_search_full.h, round 6) in plain Python -- the FULL search
(percentageOfNodesToScore = 100: every node filtered and scored each cycle, vendor/k8s.io/kubernetes/pkg/scheduler/schedule_one.go:430-478,
:697-723) of a template WITHOUT topology-coupled plugins, answered from resident per-node words and per-block summaries by ONE sequential
agent (the kernel: one wave) that never passes over the nodes.

Resident (k_sb_build): memo[n] = TotalScore of node n under the ASSUMED normalization maxima (mt_a, ma_a), -1 = infeasible; per block of B
nodes the best (score, lowest index) key and the maxima of the two raw scores over its feasible nodes; per GROUP of G blocks the maximum
of the blocks' keys and maxima.  With every node kept, the maxima over the kept nodes are those over all feasible nodes (= the maximum
of the groups' maxima) and the winner is the greatest key (all N nodes are processed: the start index stays, ties go to the lowest index).

The agent HOLDS the node that won last (its row and memo word are not in memory yet) together with
    rest = the best key among everything else = max(ob: the block's other nodes, og: the group's other blocks, orr: the other groups),
which does not move while nothing but the held node changes.  So
  * the held node wins the next cycle too while its key stays above `rest`: its next `lanes` states are scored at once, the first whose
    key falls below `rest` (or that is infeasible) ends the STREAK and is the state the node is left in;
  * only when another node wins: the held node's word goes to memory (and into the block's words), the block's key to the summaries if
    the winner lies in another block, the winner's block is fetched, and ob / og / orr are reduced anew;
  * a node that leaves the feasible ones may have held a maximum: block, group and global maxima are recomputed; global maxima other
    than the assumed ones -> everything is rebuilt under the true ones ("stale maxima: rescan").
Checked against the oracle's literal loop in tests/test_full_search_model.py: same log, same stop, same number of nodes visited -- and
(check=True) at every change of the winner: every summary equals its recomputation from the memo, `rest` equals the maximum over all
other nodes' true words."""
from __future__ import annotations

from coupled_model import CoupledWindowModel

IDX_BITS = 40
IDX_MASK = (1 << IDX_BITS) - 1


def make_key(score, idx):  # csrc/ccsim_kernels.h make_key: greater score first, then the LOWER index
    return ((score + 1) << IDX_BITS) | (IDX_MASK - idx)


def key_index(key):
    return IDX_MASK - (key & IDX_MASK)


class FullSearchModel:
    def __init__(self, prof, nodes, pod, block=64, group=8, lanes=64, check=True, start=0, node_model=None, assumed=(0, 0)):
        """start / node_model / assumed: the SAMPLED search hands over to this form once fewer feasible nodes are left than it wants to keep
        (schedule_one.go:538: every node is visited from then on and nextStartNodeIndex stays where it is): the visiting order -- and with
        it the tie-break -- starts at that index, so every key carries the node's RING position behind it instead of its index."""
        assert not pod.spread and pod.ipa is None
        self.m = node_model if node_model is not None else CoupledWindowModel(prof, nodes, pod, go_log=None, every_node_scored=False)
        self.N, self.B, self.G, self.W, self.check = nodes.n, block, group, lanes, check
        self.S = start
        self.nb = -(-self.N // block)
        self.ng = -(-self.nb // group)
        self.mt_a, self.ma_a = assumed
        self.builds = self.node_changes = self.block_changes = self.evaluations = self.gone = 0
        self.build()

    # ---- k_sb_build
    def _word(self, n):
        return self.m.local_score(n, self.mt_a, self.ma_a) if self.m.node_feasible(n) else -1

    def _key(self, w, n):  # greater score first, then the earlier ring position
        return make_key(w, n - self.S if n >= self.S else n + self.N - self.S)

    def _node(self, key):
        n = key_index(key) + self.S
        return n - self.N if n >= self.N else n

    def _block(self, b):
        return range(b * self.B, min(self.N, (b + 1) * self.B))

    def _summary(self, b, words=None):
        key, mt, ma = 0, 0, 0
        for n in self._block(b):
            w = self.memo[n] if words is None else words[n - b * self.B]
            if w >= 0:
                key = max(key, self._key(w, n))
                mt, ma = max(mt, self.m.cnt[n]), max(ma, self.m.aff[n])
        return key, (mt, ma)

    def build(self):
        self.builds += 1
        self.memo = [self._word(n) for n in range(self.N)]
        sm = [self._summary(b) for b in range(self.nb)]
        self.key, self.mx = [s[0] for s in sm], [s[1] for s in sm]
        self.Ftotal = sum(1 for w in self.memo if w >= 0)
        self.gk = [max(self.key[g * self.G:(g + 1) * self.G]) for g in range(self.ng)]
        self.gm = [self._pkmax(self.mx[g * self.G:(g + 1) * self.G]) for g in range(self.ng)]
        self.root_mx = self._pkmax(self.gm)

    @staticmethod
    def _pkmax(pairs):
        return (max((p[0] for p in pairs), default=0), max((p[1] for p in pairs), default=0))

    def _unplace(self, n):
        m = self.m
        for c in range(m.ncol):
            m.req[c][n] -= m.preq[c]
        m.z0[n] -= int(m.pod.nz_mcpu)
        m.z1[n] -= int(m.pod.nz_mem)
        m.npods[n] -= 1
        m.clones[n] -= 1

    # ---- k_sf_cycles
    def run(self, limit=0):
        N, B, G = self.N, self.B, self.G
        log, visited = [], 0
        while True:  # one launch per iteration: a rebuild ends it
            pg = pb = -1
            cur_m, cur_key, ob, og, orr = -1, 0, 0, 0, 0
            bm = []
            top = rest = max(self.gk, default=0)
            stop = None

            def close_block():
                if pb < 0:
                    return
                leaf = max(cur_key, ob)
                self.key[pb] = leaf
                self.gk[pb // G] = max(leaf, og)

            def store_row():
                if pg >= 0:
                    self.memo[pg] = cur_m

            while True:
                if limit and len(log) >= limit:
                    stop = "LimitReached"
                    break
                if top == 0:
                    stop, visited = "Unschedulable", visited + N
                    break
                if self.root_mx != (self.mt_a, self.ma_a):
                    stop = "rebuild"
                    break
                g = self._node(top)
                if g != pg:
                    self.node_changes += 1
                    b, grp = g // B, g // B // G
                    other = b != pb
                    if other:
                        self.block_changes += 1
                        close_block()
                    if pg >= 0:
                        bm[pg - pb * B] = cur_m  # the held node's slot of its block's words
                    store_row()
                    if other:
                        bm = [self.memo[n] for n in self._block(b)]
                        og = max((self.key[x] for x in range(grp * G, min(self.nb, (grp + 1) * G)) if x != b), default=0)
                        orr = max((self.gk[x] for x in range(self.ng) if x != grp), default=0)
                    ob = max((self._key(w, b * B + i) for i, w in enumerate(bm) if w >= 0 and b * B + i != g), default=0)
                    rest = max(ob, og, orr)
                    pg, pb = g, b
                    if self.check:
                        for x in range(self.nb):
                            if x != pb:
                                assert (self.key[x], self.mx[x]) == self._summary(x), ("stale summary", x)
                        for x in range(self.ng):
                            if x != pb // G:
                                assert self.gk[x] == max(self.key[x * G:(x + 1) * G]), ("stale group key", x)
                        assert rest == max((self._key(w, n) for n, w in enumerate(self.memo) if w >= 0 and n != g), default=0), "rest"
                        assert self.memo[g] >= 0 and self._key(self.memo[g], g) == top
                # ---- the streak: the held node's next W states at once
                self.evaluations += 1
                cap = limit - len(log) if limit else 1 << 62
                words = []
                for _ in range(self.W):
                    self.m.place(g)
                    words.append(self._word(g))
                r = self.W
                for j, w in enumerate(words):
                    if not (w >= 0 and self._key(w, g) > rest):
                        r = j + 1
                        break
                r = min(r, cap)
                for _ in range(self.W - r):
                    self._unplace(g)
                cur_m = words[r - 1]
                cur_key = self._key(cur_m, g) if cur_m >= 0 else 0
                log += [g] * r
                visited += r * N
                if cur_m < 0:  # the node left the feasible ones
                    self.gone += 1
                    self.Ftotal -= 1
                    mt = max((self.m.cnt[pb * B + i] for i, w in enumerate(bm) if w >= 0 and pb * B + i != g), default=0)
                    ma = max((self.m.aff[pb * B + i] for i, w in enumerate(bm) if w >= 0 and pb * B + i != g), default=0)
                    self.mx[pb] = (mt, ma)
                    self.gm[pb // G] = self._pkmax(self.mx[(pb // G) * G:(pb // G + 1) * G])
                    self.root_mx = self._pkmax(self.gm)
                top = max(cur_key, rest)
            close_block()
            store_row()
            if stop != "rebuild":
                if self.check:
                    assert self.memo == [self._word(n) for n in range(N)], "memo words after the launch"
                return log, stop, visited
            self.mt_a, self.ma_a = self.root_mx
            self.build()
