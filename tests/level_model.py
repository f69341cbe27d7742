"""Executable specification of CCSIM_MODE_BATCHED (cluster-capacity_amd/csrc/ccsim_level.h) in plain
Python -- TEST INFRASTRUCTURE.  It mirrors the decision logic of k_level / level_decide (levels,
run-downs, the normalization "cut", limit truncation) so that the exactness argument of the batched
mode can be checked against the sequential oracle on the CPU, without a GPU.

Arithmetic restated from the reference (S/ = vendor/k8s.io/kubernetes/pkg/scheduler):
  fitsRequest            S/framework/plugins/noderesources/fit.go:564-660
  LeastAllocated         S/framework/plugins/noderesources/least_allocated.go:30-61
  BalancedAllocation     S/framework/plugins/noderesources/balanced_allocation.go:146-180
  DefaultNormalizeScore  S/framework/plugins/helper/normalize_score.go:28-56
"""
from __future__ import annotations

import numpy as np

F_UNSCHEDULABLE, F_TAINT, F_NODEAFFINITY, F_FIT = 1, 4, 8, 16


def _term(nodes, reqs, n, empty_matches):
    if not reqs:
        return empty_matches
    return all(table[nodes.label_cols[col][n]] for col, table in reqs)


class LevelModel:
    def __init__(self, prof, nodes, pod):
        self.prof, self.nd, self.pod = prof, nodes, pod
        N = nodes.n
        self.N = N
        self.ncol = len(nodes.alloc)
        self.alloc = [a.astype(object) for a in nodes.alloc]
        self.req = [np.array(r, dtype=object) for r in nodes.req]
        self.z_cpu = np.array(nodes.nz_mcpu, dtype=object)
        self.z_mem = np.array(nodes.nz_mem, dtype=object)
        self.npods = [int(x) for x in nodes.pod_count]
        self.a_pods = [int(x) for x in nodes.alloc_pods]
        self.preq = [int(x) for x in pod.req] + [0] * (self.ncol - len(pod.req))
        fm = prof.filter_mask
        self.fit_on = bool(fm & F_FIT)
        self.all_zero = self.preq[0] == 0 and self.preq[1] == 0 and self.preq[2] == 0 and not pod.has_scalar_entries
        self.xcols = [c for c in range(2, self.ncol) if self.preq[c] != 0]
        w_taint = prof.w_taint
        self.w_aff = prof.w_nodeaffinity if pod.preferred else 0
        self.w_taint = w_taint
        best_effort = all(self.preq[c] == 0 for c in prof.bal_res)
        self.w_bal = 0 if best_effort else prof.w_balanced
        # static part (k_static)
        self.ok, self.cnt, self.aff = [], [], []
        for n in range(N):
            ts = int(nodes.taintset_id[n])
            ok = True
            if (fm & F_UNSCHEDULABLE) and nodes.unschedulable[n] and not pod.tolerates_unschedulable:
                ok = False
            if ok and (fm & F_TAINT) and not pod.taint_filter_ok[ts]:
                ok = False
            if ok and (fm & F_NODEAFFINITY) and pod.affinity_filter_active:
                m = True
                if pod.has_node_selector:
                    m = _term(nodes, pod.node_selector, n, True)
                if m and pod.has_required_terms:
                    m = any(_term(nodes, t, n, False) for t in pod.required)
                ok = m
            self.ok.append(ok)
            self.cnt.append(int(pod.taint_prefer_cnt[ts]) if w_taint else 0)
            self.aff.append(sum(w for (w, t) in pod.preferred if _term(nodes, t, n, False)) if self.w_aff else 0)

    # ---- per-node functions on the current state ----
    def feasible(self, n):
        if not self.ok[n]:
            return False
        if not self.fit_on:
            return True
        if self.npods[n] + 1 > self.a_pods[n]:
            return False
        if not self.all_zero:
            for c in (0, 1):
                if self.preq[c] > 0 and self.preq[c] > self.alloc[c][n] - self.req[c][n]:
                    return False
            for c in self.xcols:
                if self.preq[c] > self.alloc[c][n] - self.req[c][n]:
                    return False
        return True

    def dyn(self, n):
        p, t = self.prof, 0
        if p.w_fit:
            score = wsum = 0
            for c, w in zip(p.fit_res, p.fit_res_w):
                a = int(self.alloc[c][n])
                if a == 0:
                    continue
                r = int(self.z_cpu[n] + self.pod.nz_mcpu) if c == 0 else int(self.z_mem[n] + self.pod.nz_mem)
                s = 0 if r > a else ((a - r) * 100) // a
                score += s * w
                wsum += w
            t += (score // wsum if wsum else 0) * p.w_fit
        if self.w_bal:
            fr = []
            for c in p.bal_res:
                a = int(self.alloc[c][n])
                if a == 0:
                    continue
                f = float(int(self.req[c][n]) + self.preq[c]) / float(a)
                fr.append(1.0 if f > 1 else f)
            std = abs((fr[0] - fr[1]) / 2) if len(fr) == 2 else 0.0
            t += int((1 - std) * 100.0) * self.w_bal
        return t

    def stat(self, n, mt, ma):
        t = 0
        if self.w_taint:
            t += (100 if mt == 0 else 100 - (100 * self.cnt[n]) // mt) * self.w_taint
        if self.w_aff:
            t += (0 if ma == 0 else (100 * self.aff[n]) // ma) * self.w_aff
        return t

    def apply(self, n, sign=1):
        for c in range(self.ncol):
            self.req[c][n] += sign * self.preq[c]
        self.z_cpu[n] += sign * self.pod.nz_mcpu
        self.z_mem[n] += sign * self.pod.nz_mem
        self.npods[n] += sign

    def run_down(self, n, stat, M, jmax):
        j = 0
        while True:
            self.apply(n)
            j += 1
            f = self.feasible(n)
            s = stat + self.dyn(n) if f else -1
            if not (f and s >= M and j < jmax):
                return j, f

    # ---- the batched loop (k_level + level_decide) ----
    def run(self, limit=0):
        placed, log, levels = 0, [], 0
        per_node = np.zeros(self.N, np.int32)
        while True:
            feas = [n for n in range(self.N) if self.feasible(n)]
            if not feas:
                return dict(placed=placed, stop=0, log=np.array(log, np.int32), per_node_count=per_node, levels=levels)
            mt = max(self.cnt[n] for n in feas)
            ma = max(self.aff[n] for n in feas)
            c_mt = sum(1 for n in feas if self.cnt[n] == mt)
            c_ma = sum(1 for n in feas if self.aff[n] == ma)
            sc = {n: self.stat(n, mt, ma) + self.dyn(n) for n in feas}
            M = max(sc.values())
            level = [n for n in feas if sc[n] == M]
            # plan: full run-downs on scratch state
            e_mt = e_ma = 0
            cut_mt = cut_ma = -1
            for n in level:
                j, f = self.run_down(n, self.stat(n, mt, ma), M, 1 << 30)
                for _ in range(j):
                    self.apply(n, -1)
                if not f:
                    if mt > 0 and self.cnt[n] == mt:
                        e_mt, cut_mt = e_mt + 1, max(cut_mt, n)
                    if ma > 0 and self.aff[n] == ma:
                        e_ma, cut_ma = e_ma + 1, max(cut_ma, n)
            cut = 1 << 62
            if mt > 0 and e_mt == c_mt:
                cut = min(cut, cut_mt)
            if ma > 0 and e_ma == c_ma:
                cut = min(cut, cut_ma)
            levels += 1
            for n in level:
                if n > cut:
                    break
                allowed = (limit - placed) if limit > 0 else (1 << 30)
                if allowed <= 0:
                    break
                j, _ = self.run_down(n, self.stat(n, mt, ma), M, allowed)
                placed += j
                per_node[n] += j
                log += [n] * j
            if limit > 0 and placed >= limit:
                return dict(placed=placed, stop=1, log=np.array(log, np.int32), per_node_count=per_node, levels=levels)

    # ---- the incremental pass structure (k_level_score only while the cache is invalid; otherwise k_level_commit
    #      re-scores the nodes it changed and the next level comes from the score cache) ----
    def run_incremental(self, limit=0):
        """Mirrors the state machine of level_decide + k_level_commit's cache maintenance: cached TotalScore per node
        (-1 infeasible) under the constants (mt, ma), running feasible / holder counts, a full pass only when a
        normalization maximum lost its last feasible holder.  Returns also how many full passes were needed."""
        placed, log, passes, full_passes = 0, [], 0, 0
        per_node = np.zeros(self.N, np.int32)
        cs = [-1] * self.N
        full, mt, ma, nfeas, c_mt, c_ma = True, 0, 0, 0, 0, 0
        while True:
            passes += 1
            if full:  # k_level_score: every node from its columns
                full_passes += 1
                feas = [n for n in range(self.N) if self.feasible(n)]
                if not feas:
                    return dict(placed=placed, stop=0, log=np.array(log, np.int32), per_node_count=per_node, levels=passes,
                                full_passes=full_passes)
                t_mt, t_ma = max(self.cnt[n] for n in feas), max(self.aff[n] for n in feas)
                if (t_mt, t_ma) != (mt, ma):  # scores would use stale constants: adopt them, rescan
                    mt, ma = t_mt, t_ma
                    continue
                cs = [self.stat(n, mt, ma) + self.dyn(n) if self.feasible(n) else -1 for n in range(self.N)]
                nfeas = len(feas)
                c_mt = sum(1 for n in feas if self.cnt[n] == mt)
                c_ma = sum(1 for n in feas if self.aff[n] == ma)
                full = False
            M = max(cs)
            if M < 0:
                return dict(placed=placed, stop=0, log=np.array(log, np.int32), per_node_count=per_node, levels=passes,
                            full_passes=full_passes)
            level = [n for n in range(self.N) if cs[n] == M]
            # plan (when the level could exhaust every holder of a maximum): where is the cut?
            cut = 1 << 62
            if (mt > 0 and len(level) >= c_mt) or (ma > 0 and len(level) >= c_ma) or limit > 0:
                e_mt = e_ma = 0
                cut_mt = cut_ma = -1
                for n in level:
                    j, f = self.run_down(n, self.stat(n, mt, ma), M, 1 << 30)
                    for _ in range(j):
                        self.apply(n, -1)
                    if not f:
                        if mt > 0 and self.cnt[n] == mt:
                            e_mt, cut_mt = e_mt + 1, max(cut_mt, n)
                        if ma > 0 and self.aff[n] == ma:
                            e_ma, cut_ma = e_ma + 1, max(cut_ma, n)
                if mt > 0 and e_mt == c_mt:
                    cut = min(cut, cut_mt)
                if ma > 0 and e_ma == c_ma:
                    cut = min(cut, cut_ma)
            # commit pass: run the level down, re-score exactly its nodes, count what became infeasible
            for n in level:
                if n > cut:
                    break
                allowed = (limit - placed) if limit > 0 else (1 << 30)
                if allowed <= 0:
                    break
                j, _ = self.run_down(n, self.stat(n, mt, ma), M, allowed)
                placed += j
                per_node[n] += j
                log += [n] * j
                f = self.feasible(n)
                cs[n] = self.stat(n, mt, ma) + self.dyn(n) if f else -1
                if not f:
                    nfeas -= 1
                    c_mt -= self.cnt[n] == mt
                    c_ma -= self.aff[n] == ma
            if limit > 0 and placed >= limit:
                return dict(placed=placed, stop=1, log=np.array(log, np.int32), per_node_count=per_node, levels=passes,
                            full_passes=full_passes)
            if nfeas > 0 and ((mt > 0 and c_mt == 0) or (ma > 0 and c_ma == 0)):
                full = True  # every cached score used a maximum that no feasible node holds any more

    # ---- the persistent kernel's fast path (ccsim_persist.h): several score levels per grid-wide sync, committed blindly,
    #      validated afterwards, rolled back and redone with fewer levels (down to ONE level in canonical order) ----
    @staticmethod
    def _pick_event(k_mt, k_ma):
        """Event keys are (level, node) of the holder of a maximum that fills up LAST: the lowest level, of its nodes the highest.
        Of the two maxima the event that comes first in canonical order counts: the higher level, then the lower node."""
        ev = None
        for k in (k_mt, k_ma):
            if k is not None and (ev is None or k[0] > ev[0] or (k[0] == ev[0] and k[1] < ev[1])):
                ev = k
        return ev if ev is not None else (-1, -1)

    def run_persistent(self, limit=0, level_batch=8, spec=True):
        """Mirrors k_level_persist without a placement log.  Per sync: every feasible node scoring >= Lo = M - kb + 1 runs down
        until it scores < Lo (or stops fitting) -- for one node exactly the sequence of its run-downs at the levels in between,
        and without a log or a limit the interleaving across nodes is unobservable.  If that exhausted every feasible holder
        of a normalization maximum, or crossed the limit, the whole batch is undone and retried.  Round 4: the re-score predicts
        WHERE the constants end -- (level, node) of the holder that fills up last -- and a batch may END exactly there: the levels
        above the event, and of the event's level the nodes up to that node (a per-node threshold), which is what the reference
        places before its normalization constants change.  The batch stands iff every maximum that ran out of holders did so at
        exactly that (level, node), as reported by the holders that filled up; anything else is rolled back (and the report says
        where the event really was), else with half the levels; a single level that still trips is redone ORDERED (plan, cut,
        canonical commit: run()'s level step).  `spec=False`: round 3's form (stop above the event, that level ordered).
        Returns the counters a log-less run reports (placed, per-node counts, stop) + how many syncs / roll-backs it took."""
        placed, syncs, rollbacks, spec_ok = 0, 0, 0, 0
        per_node = np.zeros(self.N, np.int32)
        kb = level_batch
        rescore = True
        ev_level, ev_cut = -1, -1  # where the current constants end: score level, node (-1: unknown)
        mt = ma = c_mt = c_ma = 0

        def filled_key(n):  # the score the (full) node had before its last clone, and the node
            self.apply(n, -1)
            sp = self.stat(n, mt, ma) + self.dyn(n)
            self.apply(n, +1)
            return (sp, n)

        def later(a, b):  # the later of two fill-ups in canonical order: lower level, then higher node
            if a is None:
                return b
            return b if (b[0] < a[0] or (b[0] == a[0] and b[1] > a[1])) else a

        while True:
            feas = [n for n in range(self.N) if self.feasible(n)]
            if not feas:
                return dict(placed=placed, stop=0, per_node_count=per_node, syncs=syncs, rollbacks=rollbacks, spec_ok=spec_ok)
            if rescore:  # exact maxima over the feasible set, then every node's TotalScore
                mt, ma = max(self.cnt[n] for n in feas), max(self.aff[n] for n in feas)
                rescore = False
                # where these constants will end: every holder's score before the clone that fills it (its run-down depends on nothing
                # but the node); per maximum the LAST holder to go, of the two maxima the first event
                keys = {"mt": None, "ma": None}
                for which, top, arr in (("mt", mt, self.cnt), ("ma", ma, self.aff)):
                    for n in feas:
                        if top > 0 and arr[n] == top:
                            j = 0
                            while self.feasible(n):
                                self.apply(n, +1)
                                j += 1
                            key = filled_key(n)
                            for _ in range(j):
                                self.apply(n, -1)
                            keys[which] = later(keys[which], key)
                ev_level, ev_cut = self._pick_event(keys["mt"], keys["ma"])
                if not spec:
                    ev_cut = -1
            c_mt = sum(1 for n in feas if self.cnt[n] == mt)
            c_ma = sum(1 for n in feas if self.aff[n] == ma)
            sc = {n: self.stat(n, mt, ma) + self.dyn(n) for n in feas}
            M = max(sc.values())
            ordered = False
            while True:  # one sync (retried with fewer levels after a roll-back)
                syncs += 1
                spec_batch = False
                if not ordered and ev_level >= 0 and M <= ev_level:
                    if ev_cut >= 0 and M == ev_level:
                        spec_batch = True
                    else:
                        ordered, ev_level, ev_cut = True, -1, -1
                kcap = kb
                if not ordered and not spec_batch and ev_level >= 0:
                    if ev_cut >= 0:
                        spec_batch = M - ev_level + 1 <= kb
                    elif M - ev_level < kcap:
                        kcap = M - ev_level
                Lo = M if ordered else (ev_level if spec_batch else max(M - (kcap - 1), 0))
                thr = (lambda n: Lo + (1 if n > ev_cut else 0)) if spec_batch else (lambda n: Lo)
                work = [n for n in feas if sc[n] >= thr(n)]
                if ordered:  # the level step of run(): plan, cut, canonical order, limit clamp
                    e_mt = e_ma = 0
                    cut_mt = cut_ma = -1
                    for n in work:
                        j, f = self.run_down(n, self.stat(n, mt, ma), M, 1 << 30)
                        for _ in range(j):
                            self.apply(n, -1)
                        if not f:
                            if mt > 0 and self.cnt[n] == mt:
                                e_mt, cut_mt = e_mt + 1, max(cut_mt, n)
                            if ma > 0 and self.aff[n] == ma:
                                e_ma, cut_ma = e_ma + 1, max(cut_ma, n)
                    cut = 1 << 62
                    if mt > 0 and e_mt == c_mt:
                        cut = min(cut, cut_mt)
                    if ma > 0 and e_ma == c_ma:
                        cut = min(cut, cut_ma)
                    for n in work:
                        if n > cut:
                            break
                        allowed = (limit - placed) if limit > 0 else (1 << 30)
                        if allowed <= 0:
                            break
                        j, _ = self.run_down(n, self.stat(n, mt, ma), M, allowed)
                        placed += j
                        per_node[n] += j
                    if limit > 0 and placed >= limit:
                        return dict(placed=placed, stop=1, per_node_count=per_node, syncs=syncs, rollbacks=rollbacks, spec_ok=spec_ok)
                    rescore = cut != 1 << 62  # a maximum lost its last feasible holder: new constants
                    break
                took, x_mt, x_ma = {}, 0, 0
                k_mt = k_ma = None  # the LAST holder that filled up, per maximum: (score before its last clone, node)
                for n in work:  # blind: any order
                    j, f = self.run_down(n, self.stat(n, mt, ma), thr(n), 1 << 30)
                    took[n] = j
                    if not f:
                        x_mt += self.cnt[n] == mt
                        x_ma += self.aff[n] == ma
                        if j > 0 and ((mt > 0 and self.cnt[n] == mt) or (ma > 0 and self.aff[n] == ma)):
                            key = filled_key(n)
                            if mt > 0 and self.cnt[n] == mt:
                                k_mt = later(k_mt, key)
                            if ma > 0 and self.aff[n] == ma:
                                k_ma = later(k_ma, key)
                total = sum(took.values())
                ex_mt, ex_ma = mt > 0 and x_mt == c_mt, ma > 0 and x_ma == c_ma
                cut_event = ex_mt or ex_ma
                over = limit > 0 and placed + total > limit
                event_done = False
                if spec_batch and cut_event and not over:  # the batch stands iff every exhausted maximum went exactly at the predicted place
                    want = (ev_level, ev_cut)
                    event_done = (not ex_mt or k_mt == want) and (not ex_ma or k_ma == want)
                if (cut_event and not event_done) or over:  # undo the whole batch
                    rollbacks += 1
                    for n, j in took.items():
                        for _ in range(j):
                            self.apply(n, -1)
                    if Lo < M or spec_batch:
                        kb = max(1, (M - Lo + 1) >> 1)
                        if cut_event and not over:
                            old = (ev_level, ev_cut if spec_batch else -1)
                            ev_level, ev_cut = -1, -1
                            ev, ec = self._pick_event(k_mt if ex_mt else None, k_ma if ex_ma else None)
                            if not spec:
                                ec = -1
                            if Lo <= ev <= M:
                                ev_level, ev_cut, kb = ev, ec, level_batch
                                if spec_batch and (ev, ec) == old:
                                    ev_cut = -1
                        if spec_batch and Lo == M and (ev_level < 0 or over):
                            ordered = True
                    else:
                        ordered = True
                    continue
                if spec_batch:
                    ev_level, ev_cut = -1, -1
                    spec_ok += event_done
                placed += total
                for n, j in took.items():
                    per_node[n] += j
                kb = min(2 * kb, level_batch)
                if limit > 0 and placed >= limit:
                    return dict(placed=placed, stop=1, per_node_count=per_node, syncs=syncs, rollbacks=rollbacks, spec_ok=spec_ok)
                if event_done:
                    rescore = True
                break
