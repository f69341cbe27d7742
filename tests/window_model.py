"""Executable specification of the several-pod-specs path (cluster-capacity_amd/csrc/ccsim_multi.h) in plain Python --
TEST INFRASTRUCTURE.  It restates the WINDOW argument so that it can be checked against the sequential oracle
(oracle/ccref.c ccref_run_multi) on the CPU, without a GPU:

  scan      every pod of the window against every node in the state S0 at the start of the window: feasibility, TotalScore
            under the pod's normalization maxima over ITS feasible set, per node tile the best TWO keys + the third as a bound
            on what the tile hides, the number of holders of each maximum;
  select    the pod's top-K candidate list + the best key left out of it;
  commit    pods in order.  Pod j's argmax over S_j (= S0 + the placements of pods 0..j-1 of the window) is
            max(best UNTOUCHED node, best TOUCHED node): an untouched node has the state the scan saw -- the pod's own spread /
            anti-affinity state only changes through its OWN clones and it appears once per window, and the normalization
            maxima cannot move while an untouched holder remains -- so the first untouched entry of the candidate list is the
            best untouched node PROVIDED nothing hidden can beat it (tiles whose two recorded nodes are both touched hide nodes
            bounded by their third key; an exhausted list hides nodes bounded by the best key left out); touched nodes are
            re-evaluated exactly.  Whenever that argument does not cover a pod the window ENDS before it.
  assign + verify (the parallel commit): pod j takes the first list entry no earlier pod took; it stands iff none of the nodes
            the earlier pods took, with one more pod on it, beats it for pod j, and the maxima keep an untouched holder.

Arithmetic restated from the reference: fit.go:564-660, least_allocated.go:30-61, balanced_allocation.go:146-180,
normalize_score.go:28-56, podtopologyspread/filtering.go:235-356, interpodaffinity/filtering.go:352-379."""
from __future__ import annotations

import numpy as np

F_UNSCHEDULABLE, F_TAINT, F_NODEAFFINITY, F_FIT, F_TOPOLOGYSPREAD, F_INTERPODAFFINITY = 1, 4, 8, 16, 32, 64
IDX_BITS = 40


def _term(nodes, reqs, n, empty_matches):
    if not reqs:
        return empty_matches
    return all(table[nodes.label_cols[col][n]] for col, table in reqs)


class WindowModel:
    def __init__(self, prof, nodes, pods, tile=16, topk=8, window=64):
        self.prof, self.nd, self.pods = prof, nodes, pods
        self.N, self.P = nodes.n, len(pods)
        self.tile, self.topk, self.window = tile, topk, min(window, len(pods))
        N = self.N
        self.a0, self.a1 = [int(x) for x in nodes.alloc[0]], [int(x) for x in nodes.alloc[1]]
        self.r0, self.r1 = [int(x) for x in nodes.req[0]], [int(x) for x in nodes.req[1]]
        self.z0, self.z1 = [int(x) for x in nodes.nz_mcpu], [int(x) for x in nodes.nz_mem]
        self.npods, self.apods = [int(x) for x in nodes.pod_count], [int(x) for x in nodes.alloc_pods]
        self.placed = [[0] * N for _ in pods]  # clones of spec s on node n
        fm = prof.filter_mask
        self.ok, self.cnt, self.aff = [], [], []
        for q in pods:
            ok, cnt, aff = [], [], []
            for n in range(N):
                ts = int(nodes.taintset_id[n])
                o = True
                if (fm & F_UNSCHEDULABLE) and nodes.unschedulable[n] and not q.tolerates_unschedulable:
                    o = False
                if o and (fm & F_TAINT) and not q.taint_filter_ok[ts]:
                    o = False
                if o and (fm & F_NODEAFFINITY) and q.affinity_filter_active:
                    m = _term(nodes, q.node_selector, n, True) if q.has_node_selector else True
                    if m and q.has_required_terms:
                        m = any(_term(nodes, t, n, False) for t in q.required)
                    o = m
                ok.append(o)
                cnt.append(int(q.taint_prefer_cnt[ts]) if prof.w_taint else 0)
                aff.append(sum(w for (w, t) in q.preferred if _term(nodes, t, n, False)) if (q.preferred and prof.w_nodeaffinity) else 0)
            self.ok.append(ok), self.cnt.append(cnt), self.aff.append(aff)

    # ---- one pod against one node in the CURRENT state -------------------------------------------------------------
    def spread_state(self, s):
        """TpValueToMatchNum, the minimum over present domains, the number of present domains (filtering.go:235-308)."""
        q, out = self.pods[s], []
        hard = [k for k in q.spread if k.hard] if (self.prof.filter_mask & F_TOPOLOGYSPREAD) else []
        for k in hard:
            match = {}
            for n in range(self.N):
                if any(self.nd.label_cols[kk.col][n] == 0 for kk in hard):
                    continue
                if k.node_included is not None and not k.node_included[n]:
                    continue
                v = int(self.nd.label_cols[k.col][n])
                c = (int(k.node_match_count[n]) if k.node_match_count is not None else 0) + (self.placed[s][n] if k.self_match else 0)
                match[v] = match.get(v, 0) + c
            mn = min(match.values()) if match else 2147483647
            if len(match) < k.min_domains:
                mn = 0
            out.append((k, match, mn))
        return out

    def feasible(self, s, n, sp):
        q = self.pods[s]
        if not self.ok[s][n]:
            return False
        if self.prof.filter_mask & F_FIT:
            if self.npods[n] + 1 > self.apods[n]:
                return False
            if int(q.req[0]) > 0 and int(q.req[0]) > self.a0[n] - self.r0[n]:
                return False
            if int(q.req[1]) > 0 and int(q.req[1]) > self.a1[n] - self.r1[n]:
                return False
        for k, match, mn in sp:
            v = int(self.nd.label_cols[k.col][n])
            if v == 0:
                return False
            if match.get(v, 0) + (1 if k.self_match else 0) - mn > k.max_skew:
                return False
        if q.ipa is not None and (self.prof.filter_mask & F_INTERPODAFFINITY):
            a = q.ipa  # required anti-affinity to its own clones on a one-node-per-domain key (what P > 1 supports)
            for t in range(len(a.anti_keys)):
                ex = int(a.anti_existing[t][n]) if a.anti_existing and a.anti_existing[t] is not None else 0
                if ex + (self.placed[s][n] if a.anti_self[t] else 0) > 0:
                    return False
        return True

    def score(self, s, n, mt, ma):
        p, q = self.prof, self.pods[s]
        t = 0
        if p.w_taint:
            t += (100 if mt == 0 else 100 - (100 * self.cnt[s][n]) // mt) * p.w_taint
        if p.w_nodeaffinity and q.preferred:
            t += (0 if ma == 0 else (100 * self.aff[s][n]) // ma) * p.w_nodeaffinity
        if p.w_fit:
            sc = ws = 0
            for c, w in zip(p.fit_res, p.fit_res_w):
                a = (self.a0, self.a1)[c][n]
                if a == 0:
                    continue
                r = (self.z0[n] + int(q.nz_mcpu)) if c == 0 else (self.z1[n] + int(q.nz_mem))
                sc += (0 if r > a else ((a - r) * 100) // a) * w
                ws += w
            t += (sc // ws if ws else 0) * p.w_fit
        if p.w_balanced and not all(int(q.req[c]) == 0 for c in p.bal_res):
            fr = []
            for c in p.bal_res:
                a = (self.a0, self.a1)[c][n]
                if a == 0:
                    continue
                f = float((self.r0, self.r1)[c][n] + int(q.req[c])) / float(a)
                fr.append(1.0 if f > 1 else f)
            std = abs((fr[0] - fr[1]) / 2) if len(fr) == 2 else 0.0
            t += int((1 - std) * 100.0) * p.w_balanced
        return t

    def key(self, total, n):
        return ((total + 1) << IDX_BITS) | ((1 << IDX_BITS) - 1 - n)

    def place(self, s, n):
        q = self.pods[s]
        self.r0[n] += int(q.req[0]); self.r1[n] += int(q.req[1])
        self.z0[n] += int(q.nz_mcpu); self.z1[n] += int(q.nz_mem)
        self.npods[n] += 1
        self.placed[s][n] += 1

    # ---- scan + select for one pod on the current state ---------------------------------------------------------------
    def scan(self, s):
        sp = self.spread_state(s)
        feas = [n for n in range(self.N) if self.feasible(s, n, sp)]
        if not feas:
            return dict(nfeas=0)
        mt, ma = max(self.cnt[s][n] for n in feas), max(self.aff[s][n] for n in feas)
        keys = {n: self.key(self.score(s, n, mt, ma), n) for n in feas}
        tiles = {}
        for n in feas:
            tiles.setdefault(n // self.tile, []).append(keys[n])
        top2, third = [], {}
        for b, ks in tiles.items():
            ks.sort(reverse=True)
            top2 += ks[:2]
            third[b] = ks[2] if len(ks) > 2 else 0
        top2.sort(reverse=True)
        return dict(nfeas=len(feas), mt=mt, ma=ma, c_mt=sum(1 for n in feas if self.cnt[s][n] == mt), c_ma=sum(1 for n in feas if self.aff[s][n] == ma),
                    cand=top2[: self.topk], bound=top2[self.topk] if len(top2) > self.topk else 0, third=third, sp=sp)

    def node_of(self, key):
        return (1 << IDX_BITS) - 1 - (key & ((1 << IDX_BITS) - 1))

    def _untouched_choice(self, cd, touched):
        """The first untouched entry of the list and the bound on what the list cannot show (0 = nothing hidden)."""
        bound, seen = 0, {}
        for k in cd["cand"]:
            n = self.node_of(k)
            if n not in touched:
                return k, bound
            b = n // self.tile
            seen[b] = seen.get(b, 0) + 1
            if seen[b] == 2:  # both recorded nodes of the tile are touched: its hidden nodes are bounded by its third key
                bound = max(bound, cd["third"].get(b, 0))
        return 0, max(bound, cd["bound"])  # the list ran out

    # ---- the run: windows with the in-order commit (multi_commit_inorder) or assign + verify (k_multi_commit_par) ---------------
    def run(self, limit=0, parallel=False):
        log, windows, early = [], 0, 0
        nxt = 0
        while True:
            W = self.window if limit <= 0 else min(self.window, limit - len(log))
            specs = [(nxt + j) % self.P for j in range(W)]
            cds = [self.scan(s) for s in specs]  # all against S0
            windows += 1
            done = self._commit_par(specs, cds, log) if parallel else self._commit_seq(specs, cds, log)
            nxt = (nxt + done["committed"]) % self.P
            early += done["committed"] < W and not done.get("stop")
            if done.get("stop") == "unschedulable":
                return dict(placed=len(log), stop=0, stop_spec=done["spec"], log=np.array(log, np.int32), windows=windows, early=early)
            if limit > 0 and len(log) >= limit:
                return dict(placed=len(log), stop=1, stop_spec=-1, log=np.array(log, np.int32), windows=windows, early=early)
            assert done["committed"] > 0, "pod 0 of a window always commits"

    def _holders_left(self, s, cd, touched):
        th_mt = sum(1 for n in touched if self.cnt[s][n] == cd["mt"])
        th_ma = sum(1 for n in touched if self.aff[s][n] == cd["ma"])
        w_aff = self.prof.w_nodeaffinity and self.pods[s].preferred
        return not ((cd["mt"] > 0 and cd["c_mt"] <= th_mt) or (cd["ma"] > 0 and w_aff and cd["c_ma"] <= th_ma))

    def _commit_seq(self, specs, cds, log):
        touched, committed = set(), 0
        for s, cd in zip(specs, cds):
            if cd["nfeas"] == 0:
                return dict(committed=committed, stop="unschedulable", spec=s)  # touched nodes only lost room
            if not self._holders_left(s, cd, touched):
                break
            tkey = 0
            for n in touched:  # re-evaluated in their current state (the pod's own plugin state is what the scan saw)
                if self.feasible(s, n, cd["sp"]):
                    tkey = max(tkey, self.key(self.score(s, n, cd["mt"], cd["ma"]), n))
            ukey, bound = self._untouched_choice(cd, touched)
            win = max(tkey, ukey)
            if (bound and win < bound) or not win:
                break
            n = self.node_of(win)
            self.place(s, n)
            touched.add(n)
            log.append(n)
            committed += 1
        return dict(committed=committed)

    def _commit_par(self, specs, cds, log):
        # A: assignment in order (what the lane-parallel fixed point of the kernel converges to)
        taken, picks = [], []
        for s, cd in zip(specs, cds):
            if cd["nfeas"] == 0:
                picks.append(("unsched", s))
                break
            ukey, bound = self._untouched_choice(cd, set(taken))
            if not ukey or (bound and ukey < bound):
                break
            picks.append((ukey, s))
            taken.append(self.node_of(ukey))
        # B: verification of every pod against the nodes the EARLIER pods took, each carrying one more pod
        ok_n = 0
        for j, (ukey, s) in enumerate(picks):
            if ukey == "unsched":
                break
            cd, before = cds[j], taken[:j]
            if not self._holders_left(s, cd, before):
                break
            beaten = False
            for t, n in enumerate(before):
                self.place(specs[t], n)  # node n as pod t left it
                if self.feasible(s, n, cd["sp"]) and self.key(self.score(s, n, cd["mt"], cd["ma"]), n) > ukey:
                    beaten = True
                self._unplace(specs[t], n)
                if beaten:
                    break
            if beaten:
                break
            ok_n += 1
        # C: apply
        for j in range(ok_n):
            self.place(specs[j], taken[j])
            log.append(taken[j])
        if ok_n < len(picks) and picks[ok_n][0] == "unsched" and ok_n == len(picks) - 1:
            return dict(committed=ok_n, stop="unschedulable", spec=picks[ok_n][1])
        return dict(committed=ok_n)

    def _unplace(self, s, n):
        q = self.pods[s]
        self.r0[n] -= int(q.req[0]); self.r1[n] -= int(q.req[1])
        self.z0[n] -= int(q.nz_mcpu); self.z1[n] -= int(q.nz_mem)
        self.npods[n] -= 1
        self.placed[s][n] -= 1


class MemoWindowModel(WindowModel):
    """Round 4's bookkeeping on top of the window argument (csrc/ccsim_multi.h: the score memo, the per-domain spread masks, the
    ASSUMED normalization maxima), restated so that its invariants can be asserted on the CPU at every use:

      memo      memo[s][n] = 0 if the static verdict / NodeResourcesFit / the spec's anti-affinity reject the pair, else
                TotalScore(s, n) + 1 under the maxima the row is STAMPED with.  A scan whose spec's stamp equals the maxima it
                assumes reads the row (here: asserts word == recomputation); any other scan fills the row; `select` stamps it.
      refresh   after a commit, the words of the nodes the window placed pods on are recomputed for every spec whose stamp
                equals its assumed maxima AFTER the commit's repairs (k_multi_refresh); a row whose spec's maxima were
                re-derived keeps its old stamp and is refilled by the spec's next scan.
      masks     allow[s][c] = the value ids a node may carry to pass constraint c's skew test; rebuilt for the specs a window
                placed (their tables moved), asserted current at every scan.
      maxima    scores are computed under the spec's ASSUMED maxima; a scan that finds other true maxima over its feasible set
                ends the window before that pod and repairs the assumption of it and of every later pod of the window.
      flags     (round 5, `flags=True`: the 16-bit memo word) a scan that READS its row sees, per node, only how the node's taint count /
                affinity sum stands to the maxima the row was computed under -- equal, above, below -- not the values.  It reports a
                true maximum that differs from the assumed one as DIFFERENT (assumed + 1 when some feasible node lies above, assumed - 1
                when none holds it), not exactly (csrc/ccsim_multi.h m_max_from_level).  The window still ends before such a pod; the
                next scan of the spec computes (its stamp no longer matches) and reports the exact maxima, so at most TWO windows in a
                row make no progress, and every score a commit uses was computed under maxima checked against the true ones."""

    def __init__(self, prof, nodes, pods, tile=16, topk=8, window=64, refresh=True, flags=False):
        super().__init__(prof, nodes, pods, tile, topk, window)
        self.flags = flags
        self.mt_a, self.ma_a = [0] * self.P, [0] * self.P
        self.stamp = [None] * self.P
        self.memo = [[0] * self.N for _ in pods]
        self.allow = [self._masks(s) for s in range(self.P)]
        self.refresh_on = refresh
        self.stats = dict(memo_scans=0, full_scans=0, refreshed=0, repairs=0, words_checked=0)

    def _hard(self, s):
        return [k for k in self.pods[s].spread if k.hard] if (self.prof.filter_mask & F_TOPOLOGYSPREAD) else []

    def _masks(self, s):
        out = []
        for k, match, mn in self.spread_state(s):
            vals = {int(self.nd.label_cols[k.col][n]) for n in range(self.N)} - {0}
            out.append({v for v in vals if match.get(v, 0) + (1 if k.self_match else 0) - mn <= k.max_skew})
        return out

    def _word(self, s, n):
        return self.score(s, n, self.mt_a[s], self.ma_a[s]) + 1 if self.feasible(s, n, []) else 0  # (no spread state: everything but that filter)

    def scan(self, s):
        assumed = (self.mt_a[s], self.ma_a[s])
        lean = self.stamp[s] == assumed
        if lean:
            self.stats["memo_scans"] += 1
            for n in range(self.N):  # THE invariant: what the scan reads is what it would compute
                assert self.memo[s][n] == self._word(s, n), ("stale memo word", s, n, self.memo[s][n], self._word(s, n))
            self.stats["words_checked"] += self.N
        else:
            self.stats["full_scans"] += 1
            self.memo[s] = [self._word(s, n) for n in range(self.N)]
        self.stamp[s] = assumed  # (k_multi_select, after the scan)
        sp = self.spread_state(s)
        assert self.allow[s] == self._masks(s), ("stale spread masks", s)
        hard = self._hard(s)
        feas = [n for n in range(self.N)
                if self.memo[s][n] and all(int(self.nd.label_cols[k.col][n]) in self.allow[s][c] for c, k in enumerate(hard))]
        assert feas == [n for n in range(self.N) if self.feasible(s, n, sp)]  # (word + masks == the filters)
        mt = max((self.cnt[s][n] for n in feas), default=0)
        ma = max((self.aff[s][n] for n in feas), default=0)
        if lean and self.flags:  # what the flag bits of the row let the scan say (m_max_from_level)
            def from_flags(vals, a):
                level = 2 if any(v > a for v in vals) else 1 if any(v == a for v in vals) else 0
                return a + 1 if level == 2 else a if level == 1 else (0 if a == 0 or not vals else a - 1)
            rep_t, rep_a = from_flags([self.cnt[s][n] for n in feas], assumed[0]), from_flags([self.aff[s][n] for n in feas], assumed[1])
            assert (rep_t == assumed[0]) == (mt == assumed[0]) and (rep_a == assumed[1]) == (ma == assumed[1])  # "differs" is exact, the value is not
            self.stats["inexact"] = self.stats.get("inexact", 0) + ((rep_t, rep_a) != (mt, ma))
            mt, ma = rep_t, rep_a
        cd = dict(nfeas=len(feas), mt=mt, ma=ma, wrong=(mt, ma) != assumed, sp=sp)
        if not feas:
            return cd
        keys = {n: self.key(self.memo[s][n] - 1, n) for n in feas}  # scores under the ASSUMED maxima
        tiles = {}
        for n in feas:
            tiles.setdefault(n // self.tile, []).append(keys[n])
        top2, third = [], {}
        for b, ks in tiles.items():
            ks.sort(reverse=True)
            top2 += ks[:2]
            third[b] = ks[2] if len(ks) > 2 else 0
        top2.sort(reverse=True)
        cd.update(c_mt=sum(1 for n in feas if self.cnt[s][n] == mt), c_ma=sum(1 for n in feas if self.aff[s][n] == ma),
                  cand=top2[: self.topk], bound=top2[self.topk] if len(top2) > self.topk else 0, third=third)
        return cd

    def run(self, limit=0, parallel=False):
        log, windows, nxt, idle = [], 0, 0, 0
        while True:
            W = self.window if limit <= 0 else min(self.window, limit - len(log))
            specs = [(nxt + j) % self.P for j in range(W)]
            cds = [self.scan(s) for s in specs]  # all against S0
            windows += 1
            cut = next((j for j, cd in enumerate(cds) if cd["wrong"]), W)  # a wrong assumption ends the window before that pod ...
            for j in range(cut, W):  # ... and one window repairs every later pod of it
                if (self.mt_a[specs[j]], self.ma_a[specs[j]]) != (cds[j]["mt"], cds[j]["ma"]):
                    self.stats["repairs"] += 1
                self.mt_a[specs[j]], self.ma_a[specs[j]] = cds[j]["mt"], cds[j]["ma"]
            before = len(log)
            done = (self._commit_par if parallel else self._commit_seq)(specs[:cut], cds[:cut], log)
            placed = list(zip(specs, log[before:]))
            for s, _ in placed:  # the placed specs' tables moved: their masks (k_multi_refresh, block (0, t))
                self.allow[s] = self._masks(s)
            if self.refresh_on:
                for _, n in placed:  # the touched nodes' words, for every spec whose row stands under today's maxima
                    for s2 in range(self.P):
                        if self.stamp[s2] == (self.mt_a[s2], self.ma_a[s2]):
                            self.memo[s2][n] = self._word(s2, n)
                            self.stats["refreshed"] += 1
            nxt = (nxt + done["committed"]) % self.P
            if done.get("stop") == "unschedulable":
                return dict(placed=len(log), stop=0, stop_spec=done["spec"], log=np.array(log, np.int32), windows=windows, stats=self.stats)
            if limit > 0 and len(log) >= limit:
                return dict(placed=len(log), stop=1, stop_spec=-1, log=np.array(log, np.int32), windows=windows, stats=self.stats)
            idle = 0 if done["committed"] else idle + 1
            assert idle <= (2 if self.flags else 1), "a window that only repaired maxima is followed by one whose pod 0 commits (two with the flag words)"
