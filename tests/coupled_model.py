"""Executable specification of a WINDOWED exact mode for ONE template with topology-coupled plugins (PodTopologySpread,
InterPodAffinity) -- TEST INFRASTRUCTURE, and the design the next engine kernel for SURVEY 8(a) rows a11/a12 follows (DESIGN.md
8.3).  Today the engine runs such pods one full node scan per placement (`ccsim_run` sequential mode); this model shows that
one scan can serve a whole window of W placements, exactly, and is checked placement by placement against the oracle's literal
loop (oracle/ccref.c) in tests/test_coupled_model.py.

The argument.  For the template, a node's verdict and score split into
  * a NODE-LOCAL part: static filters, NodePorts, NodeResourcesFit, and the scores of TaintToleration / NodeAffinity (under
    normalization maxima Mt / Ma), LeastAllocated, BalancedAllocation, ImageLocality -- it changes only when a clone lands on
    that node;
  * a COUPLED part: the PodTopologySpread filter and score and the InterPodAffinity filter and score.  They read per-domain
    tables (matching pods per topology value, affinity / anti-affinity counts, score sums) at the node's own topology values,
    plus cycle-wide quantities (the global minimum of a constraint, the number of candidate domains, the feasible-node count,
    the min / max of the raw scores) -- nothing else of the node.
So nodes that agree on the values the coupled part reads form a CLASS: same topology value for every key shared by several
nodes, same table ENTRIES for every key whose values are unique per node (kubernetes.io/hostname: the "domain" is the node, its
table entries are node state).  Inside a class every node that has not received a clone since the scan has the same coupled
filter verdict and the same raw coupled scores in every later cycle; they differ by the node-local part A(n) only, which is
frozen until a clone lands.  Hence, per cycle, the best node of a class is the first not-yet-taken entry of the class's list
sorted by (A descending, index ascending), and the cycle's winner is the best over the classes' heads and the nodes touched in
this window (re-evaluated individually).  A window of W cycles consumes at most W entries, so the scan keeps the W best per
class -- the per-cycle work no longer depends on the number of nodes.

What a cycle recomputes exactly from (class heads, per-class counters, touched nodes, domain tables): the hard constraints'
domain minima, which classes pass the coupled filters, the feasible-node count and the candidate-domain counts behind the
PodTopologySpread weights (scoring.go:294-296), the min / max of the raw PodTopologySpread and InterPodAffinity scores.
What it ASSUMES and verifies: the TaintToleration / NodeAffinity maxima Mt / Ma over the feasible set equal the ones the scan
normalized A with; a class whose last holder of its own maximum was taken, or a cycle whose maxima differ, ENDS the window
before that cycle (the next scan starts from the exact state, so every executed cycle is exact; the first cycle of a window
always executes).

Arithmetic: fit.go:564-660, least_allocated.go:30-61, balanced_allocation.go:146-180, normalize_score.go:28-56,
podtopologyspread/filtering.go:235-356 + scoring.go:61-265, interpodaffinity/filtering.go:204-432 + scoring.go:81-290 -- restated
here the way oracle/ccref.c states them (math.Log through the oracle's ccref_go_log)."""
from __future__ import annotations

import numpy as np

F_UNSCHEDULABLE, F_TAINT, F_NODEAFFINITY, F_FIT, F_TOPOLOGYSPREAD, F_INTERPODAFFINITY, F_NODEPORTS = 1, 4, 8, 16, 32, 64, 128
MAXINT32 = 2147483647


def _term(nodes, reqs, n, empty_matches):
    if not reqs:
        return empty_matches
    return all(table[nodes.label_cols[col][n]] for col, table in reqs)


def _go_round(x: float) -> int:  # C round(): half away from zero
    import math
    return int(math.floor(x + 0.5)) if x >= 0 else -int(math.floor(-x + 0.5))


class CoupledWindowModel:
    def __init__(self, prof, nodes, pod, go_log, window=64, every_node_scored=True, device_plan=False, list_len=None):
        """device_plan: the three simplifications the HIP port (csrc/ccsim_coupled.h) makes, so that they are checked against the
        oracle here first -- (1) the class key carries no `counted` bit (nothing an untouched node's verdict or score reads);
        (2) a class keeps only its `list_len` best members: a cycle that would need an unknown head ends the window; (3) the
        minimum of a hard constraint over a unique-per-node key is not recomputed from N table entries per cycle: the scan's
        (minimum, nodes at the minimum) pair is tracked, and the window ends when the last node at the minimum is taken."""
        self.prof, self.nd, self.pod, self.go_log, self.W = prof, nodes, pod, go_log, window
        self.device_plan, self.L = device_plan, (list_len if list_len else window)
        N = self.N = nodes.n
        fm = self.fm = prof.filter_mask
        self.ncol = len(nodes.alloc)
        self.alloc = [[int(x) for x in a] for a in nodes.alloc]
        self.req = [[int(x) for x in a] for a in nodes.req]
        self.z0, self.z1 = [int(x) for x in nodes.nz_mcpu], [int(x) for x in nodes.nz_mem]
        self.npods, self.apods = [int(x) for x in nodes.pod_count], [int(x) for x in nodes.alloc_pods]
        self.clones = [0] * N
        self.preq = [int(x) for x in pod.req]
        self.all_zero = not any(self.preq[c] > 0 for c in range(3)) and not pod.has_scalar_entries
        assert not every_node_scored or prof.percentage_of_nodes_to_score == 100 or N < 100  # (the window argument needs every node scored)
        assert any((prof.w_taint, prof.w_nodeaffinity, prof.w_fit, prof.w_balanced, prof.w_topologyspread, prof.w_interpodaffinity, prof.w_imagelocality))
        # ---- static per node ----
        self.ok, self.cnt, self.aff = [], [], []
        for n in range(N):
            ts = int(nodes.taintset_id[n])
            o = not ((fm & F_UNSCHEDULABLE) and nodes.unschedulable[n] and not pod.tolerates_unschedulable)
            if o and (fm & F_TAINT) and not pod.taint_filter_ok[ts]:
                o = False
            if o and (fm & F_NODEAFFINITY) and pod.affinity_filter_active:
                m = _term(nodes, pod.node_selector, n, True) if pod.has_node_selector else True
                if m and pod.has_required_terms:
                    m = any(_term(nodes, t, n, False) for t in pod.required)
                o = m
            self.ok.append(o)
            self.cnt.append(int(pod.taint_prefer_cnt[ts]) if prof.w_taint else 0)
            self.aff.append(sum(w for (w, t) in pod.preferred if _term(nodes, t, n, False)) if (prof.w_nodeaffinity and pod.preferred) else 0)
        self.ports_on = bool((fm & F_NODEPORTS) and pod.has_host_ports)
        self.img = [int(x) for x in pod.image_score] if (prof.w_imagelocality and pod.image_score is not None) else None
        self.bal_on = bool(prof.w_balanced) and any(self.preq[c] != 0 for c in prof.bal_res)
        # ---- coupled plugins: which are on, label columns, uniqueness of a key's values ----
        self.spread = list(pod.spread)
        self.hard = [i for i, c in enumerate(self.spread) if c.hard] if (fm & F_TOPOLOGYSPREAD) else []
        self.soft = [i for i, c in enumerate(self.spread) if not c.hard] if prof.w_topologyspread else []
        self.sdom = [[int(x) for x in nodes.label_cols[c.col]] for c in self.spread]
        self.sexist = [[int(x) for x in c.node_match_count] if c.node_match_count is not None else [0] * N for c in self.spread]
        self.sincl = [[int(x) for x in c.node_included] if c.node_included is not None else [1] * N for c in self.spread]
        self.hard_keys = [all(self.sdom[i][n] != 0 for i, c in enumerate(self.spread) if c.hard) for n in range(N)]
        self.soft_keys = [all(self.sdom[i][n] != 0 for i, c in enumerate(self.spread) if not c.hard) for n in range(N)]
        self.ipa = pod.ipa
        if self.ipa is not None:
            a = self.ipa
            self.kdom = [[int(x) for x in nodes.label_cols[col]] for col in a.key_cols]
            z = [0] * N
            self.i_aff = [int(x) for x in a.aff_existing] if a.aff_existing is not None else z
            self.i_anti = [[int(x) for x in v] if v is not None else z for v in a.anti_existing]
            self.i_exist = [[int(x) for x in v] if v is not None else z for v in a.exist_anti]
            self.i_score = [[int(x) for x in v] if v is not None else z for v in a.score_existing]

        def unique(dom):
            seen = set()
            for v in dom:
                if v and v in seen:
                    return False
                seen.add(v)
            return True
        self.s_unique = [unique(d) for d in self.sdom]
        self.k_unique = [unique(d) for d in self.kdom] if self.ipa is not None else []
        # n_dom of a hard constraint (filtering.go:105-117): domains holding a counted node -- static
        self.n_dom = {}
        for i in self.hard:
            self.n_dom[i] = len({self.sdom[i][n] for n in range(N) if self.hard_keys[n] and self.sincl[i][n]})

    # ---- node-local ----
    def node_feasible(self, n):
        if not self.ok[n]:
            return False
        if self.ports_on and ((self.pod.host_ports_conflict is not None and self.pod.host_ports_conflict[n]) or self.clones[n] > 0):
            return False
        if not self.fm & F_FIT:
            return True
        if self.npods[n] + 1 > self.apods[n]:
            return False
        if not self.all_zero:
            for c in range(self.ncol):
                rq = self.preq[c]
                if rq != 0 and rq > self.alloc[c][n] - self.req[c][n]:
                    return False
        return True

    def local_score(self, n, mt, ma):
        p, t = self.prof, 0
        if p.w_taint:
            t += (100 if mt == 0 else 100 - (100 * self.cnt[n]) // mt) * p.w_taint
        if p.w_nodeaffinity and self.pod.preferred:
            t += (0 if ma == 0 else (100 * self.aff[n]) // ma) * p.w_nodeaffinity
        if p.w_fit:
            score = wsum = 0
            for c, w in zip(p.fit_res, p.fit_res_w):
                pr = self.pod.nz_mcpu if c == 0 else self.pod.nz_mem if c == 1 else self.preq[c]
                if c >= 3 and pr == 0:
                    continue
                a = self.alloc[c][n]
                if a == 0:
                    continue
                r = (self.z0[n] if c == 0 else self.z1[n] if c == 1 else self.req[c][n]) + int(pr)
                score += (0 if r > a else ((a - r) * 100) // a) * int(w)
                wsum += int(w)
            t += (score // wsum if wsum else 0) * p.w_fit
        if self.bal_on:
            fr = []
            for c in p.bal_res:
                if c >= 3 and self.preq[c] == 0:
                    continue
                a = self.alloc[c][n]
                if a == 0:
                    continue
                f = float(self.req[c][n] + self.preq[c]) / float(a)
                fr.append(1.0 if f > 1 else f)
            if len(fr) == 2:
                std = abs((fr[0] - fr[1]) / 2)
            elif len(fr) > 2:
                mean = sum(fr) / float(len(fr))  # (left-to-right sums, as the reference accumulates them)
                acc = 0.0
                for f in fr:
                    acc = acc + (f - mean) * (f - mean)
                std = float(np.sqrt(acc / float(len(fr))))
            else:
                std = 0.0
            t += int((1 - std) * 100.0) * p.w_balanced
        if self.img is not None:
            t += self.img[n] * p.w_imagelocality
        return t

    def place(self, n):
        for c in range(self.ncol):
            self.req[c][n] += self.preq[c]
        self.z0[n] += int(self.pod.nz_mcpu)
        self.z1[n] += int(self.pod.nz_mem)
        self.npods[n] += 1
        self.clones[n] += 1

    # ---- domain tables from the whole state (what one scan pass accumulates) ----
    def build_tables(self):
        N = self.N
        T = {"hard": {}, "soft": {}, "aff": {}, "anti": {}, "exist": {}, "score": {}}
        for i in self.hard:
            t = {}
            for n in range(N):
                if self.hard_keys[n] and self.sincl[i][n]:
                    v = self.sdom[i][n]
                    t[v] = t.get(v, 0) + self.sexist[i][n] + (self.clones[n] if self.spread[i].self_match else 0)
            T["hard"][i] = t
        for i in self.soft:
            t = {}
            if not self.spread[i].is_hostname:
                for n in range(N):
                    if self.soft_keys[n] and self.sincl[i][n]:
                        v = self.sdom[i][n]
                        t[v] = t.get(v, 0) + self.sexist[i][n] + (self.clones[n] if self.spread[i].self_match else 0)
            T["soft"][i] = t
        T["aff_total"] = T["exist_total"] = 0
        T["entries"] = 0
        if self.ipa is not None:
            a = self.ipa
            T["entries"] = int(a.entries_existing)
            for k in range(len(a.key_cols)):
                for name in ("aff", "anti", "exist", "score"):
                    T[name][k] = {}
            for n in range(N):
                self._ipa_contrib(T, n, self.clones[n], with_existing=True)
        return T

    def _ipa_contrib(self, T, n, clones, with_existing):
        """ipa_build's loop body for one node (oracle/ccref.c): `clones` clones, optionally the node's existing pods."""
        a = self.ipa
        am = (self.i_aff[n] if with_existing else 0) + (clones if a.self_aff else 0)
        if a.aff_keys and am:
            for k in a.aff_keys:
                v = self.kdom[k][n]
                if v:
                    T["aff"][k][v] = T["aff"][k].get(v, 0) + am
                    T["aff_total"] += am
        for t, k in enumerate(a.anti_keys):
            m = (self.i_anti[t][n] if with_existing else 0) + (clones if a.anti_self[t] else 0)
            v = self.kdom[k][n]
            if m and v:
                T["anti"][k][v] = T["anti"][k].get(v, 0) + m
        for k in range(len(a.key_cols)):
            m = self.i_exist[k][n] if with_existing else 0
            for t, kk in enumerate(a.anti_keys):
                if kk == k and a.anti_self[t]:
                    m += clones
            v = self.kdom[k][n]
            if m and v:
                T["exist"][k][v] = T["exist"][k].get(v, 0) + m
                T["exist_total"] += m
            w = (self.i_score[k][n] if with_existing else 0) + clones * int(a.score_self[k])
            if v:
                T["score"][k][v] = T["score"][k].get(v, 0) + w
                T["entries"] += clones * int(a.self_entries[k])

    def add_clone_to_tables(self, T, n):
        for i in self.hard:
            if self.hard_keys[n] and self.sincl[i][n] and self.spread[i].self_match:
                v = self.sdom[i][n]
                T["hard"][i][v] = T["hard"][i].get(v, 0) + 1
        for i in self.soft:
            if not self.spread[i].is_hostname and self.soft_keys[n] and self.sincl[i][n] and self.spread[i].self_match:
                v = self.sdom[i][n]
                T["soft"][i][v] = T["soft"][i].get(v, 0) + 1
        if self.ipa is not None:
            self._ipa_contrib(T, n, 1, with_existing=False)

    # ---- the coupled part of one node against the tables ----
    def class_key(self, T, n):
        key = []
        for i, c in enumerate(self.spread):
            v = self.sdom[i][n]
            if i in self.hard:
                counted = self.hard_keys[n] and bool(self.sincl[i][n])
                if self.device_plan:
                    key.append(("h", v != 0, T["hard"][i].get(v, 0)) if self.s_unique[i] else ("h", v))
                else:
                    key.append(("h", v != 0, counted, T["hard"][i].get(v, 0)) if self.s_unique[i] else ("h", v, counted))
            elif i in self.soft:
                if c.is_hostname:
                    key.append(("s", v != 0, self.sexist[i][n] + (self.clones[n] if c.self_match else 0)))
                else:
                    key.append(("s", v))
        if self.ipa is not None:
            for k in range(len(self.ipa.key_cols)):
                v = self.kdom[k][n]
                if self.k_unique[k]:
                    key.append(("k", v != 0) + tuple(T[name][k].get(v, 0) for name in ("aff", "anti", "exist", "score")))
                else:
                    key.append(("k", v))
        return tuple(key)

    def hard_minima(self, T):
        out = {}
        for i in self.hard:
            vals = T["hard"][i]
            mn = min(vals.values()) if vals else MAXINT32
            out[i] = 0 if self.n_dom[i] < self.spread[i].min_domains else mn
        return out

    def coupled_filter(self, T, minima, n):
        for i in self.hard:
            c = self.spread[i]
            v = self.sdom[i][n]
            if v == 0:
                return False
            if T["hard"][i].get(v, 0) + (1 if c.self_match else 0) - minima[i] > c.max_skew:
                return False
        if self.ipa is not None and (self.fm & F_INTERPODAFFINITY):
            a = self.ipa
            if not (T["exist_total"] == 0 and not a.aff_keys and not a.anti_keys):
                pods_exist = True
                for k in a.aff_keys:
                    v = self.kdom[k][n]
                    if not v:
                        return False
                    if T["aff"][k].get(v, 0) <= 0:
                        pods_exist = False
                if not pods_exist and not (T["aff_total"] == 0 and a.self_aff):
                    return False
                for k in a.anti_keys:
                    v = self.kdom[k][n]
                    if v and T["anti"][k].get(v, 0) > 0:
                        return False
                if T["exist_total"] > 0:
                    for k in range(len(a.key_cols)):
                        v = self.kdom[k][n]
                        if v and T["exist"][k].get(v, 0) > 0:
                            return False
        return True

    def raw_pts(self, T, weights, n):
        score = 0.0
        for i in self.soft:
            c = self.spread[i]
            v = self.sdom[i][n]
            if v == 0:
                continue
            ct = (self.sexist[i][n] + (self.clones[n] if c.self_match else 0)) if c.is_hostname else T["soft"][i].get(v, 0)
            score += float(ct) * weights[i] + float(c.max_skew - 1)
        return _go_round(score)

    def raw_ipa(self, T, n):
        return sum(T["score"][k].get(self.kdom[k][n], 0) for k in range(len(self.ipa.key_cols)) if self.kdom[k][n])

    # ---- sweeps (round 5, csrc/ccsim_coupled.h `sweep`): the next cycles of a window PREDICTED from the cycle at hand ----
    def sweep_predict(self, T, minima, cand, mt, ma, budget):
        """The kernel's rule, restated: one hard constraint over a shared key, at most one inter-pod key and that one unique per node,
        no plugin score beside the node-local one; every candidate is a class head whose domain sits AT THE CAP of the skew test and
        whose head counts for the constraint.  Then cycle p is won by the p-th best of the domains' best heads, for as long as the
        loop's own stop tests -- as positions in that order -- allow.  Returns the predicted winners (possibly fewer than 2: no sweep).
        run(audit_sweeps=True) holds every prediction against the cycles the loop then really runs."""
        p = self.prof
        if len(self.hard) != 1 or self.soft:
            return []
        i = self.hard[0]
        c0 = self.spread[i]
        if self.s_unique[i] or not c0.self_match:
            return []
        a = self.ipa
        danti = 0
        if a is not None:
            if len(a.key_cols) != 1 or not self.k_unique[0] or (p.w_interpodaffinity and (T["entries"] > 0 or any(a.self_entries))):
                return []
            danti = sum(1 for t, k in enumerate(a.anti_keys) if k == 0 and a.anti_self[t])
            if (self.fm & F_INTERPODAFFINITY) and ((T["aff_total"] == 0 and a.self_aff and a.aff_keys) or (T["exist_total"] == 0 and danti)):
                return []  # the first clone of a run moves the totals from "none" to "some": the loop's business
        if any(c is None for _, c in cand):
            return []  # a node that already took a clone is a candidate
        cnt_of = lambda n: T["hard"][i].get(self.sdom[i][n], 0)
        for n, _ in cand:
            if not (self.hard_keys[n] and self.sincl[i][n]) or cnt_of(n) + 1 - minima[i] != c0.max_skew:
                return []
        key = lambda n, c: (-c["list"][c["head"]][0], -n)
        best = {}
        for n, c in cand:
            d = self.sdom[i][n]
            if d not in best or key(n, c) > key(*best[d]):
                best[d] = (n, c)
        order = sorted(best.values(), key=lambda nc: key(*nc), reverse=True)
        pos = {self.sdom[i][n]: q for q, (n, _) in enumerate(order)}
        length = min(len(order), budget)

        def stays(n):  # would the head stay a candidate after its clone?  (commit's `dead`)
            if self.ports_on or self.npods[n] + 2 > self.apods[n]:
                return False
            if (self.fm & F_FIT) and not self.all_zero and any(self.preq[c] != 0 and 2 * self.preq[c] > self.alloc[c][n] - self.req[c][n] for c in range(self.ncol)):
                return False
            return not (a is not None and (self.fm & F_INTERPODAFFINITY) and danti and self.kdom[0][n])

        for q, (n, _) in enumerate(order):
            if stays(n):
                length = min(length, q)
                break
        # the assumed maxima must be held by a candidate that is still feasible: the last position with a holder
        length = min(length, 1 + max((pos[self.sdom[i][n]] for n, c in cand if c["mt"] == mt), default=-1),
                     1 + max((pos[self.sdom[i][n]] for n, c in cand if c["ma"] == ma), default=-1))
        # the candidate that takes the last domain at the minimum ends the round
        vals = T["hard"][i]
        mn = min(vals.values()) if vals else MAXINT32
        n_at_min = sum(v == mn for v in vals.values())
        at_min = [q for q, (n, _) in enumerate(order) if cnt_of(n) == mn]
        if n_at_min and len(at_min) >= n_at_min:
            length = min(length, at_min[n_at_min - 1] + 1)
        return [n for n, _ in order[:length]] if length >= 2 else []

    # ---- the windowed loop ----
    # ---- the node pass, per node range (the whole cluster here; a rank's shard in ShardedCoupledWindowModel) ------------------------------
    node_range = None  # (lo, hi): set in run()

    def exchange(self, scan):
        """What the pass `scan(lo, hi)` finds on every rank's node range, in rank order (one rank: its own)."""
        return [scan(*self.node_range)]

    def scan_facts(self, T, minima, lo, hi):
        feas0 = [n for n in range(lo, hi) if self.node_feasible(n) and self.coupled_filter(T, minima, n)]
        return (bool(feas0), max((self.cnt[n] for n in feas0), default=0), max((self.aff[n] for n in feas0), default=0))

    def scan_classes(self, T, mt, ma, lo, hi):
        """key -> dict(list = the best node-feasible members of the range by (A desc, index asc), nf, max / holders of cnt and aff)"""
        classes = {}
        for n in range(lo, hi):
            if not self.node_feasible(n):
                continue
            c = classes.setdefault(self.class_key(T, n), {"all": [], "nf": 0})
            c["all"].append((-self.local_score(n, mt, ma), n))
            c["nf"] += 1
        for c in classes.values():
            c["all"].sort()
            c["list"] = c["all"][: (self.L if self.device_plan else self.W)]
            members = [n for _, n in c["all"]]
            c["mt"], c["ma"] = max(self.cnt[n] for n in members), max(self.aff[n] for n in members)
            c["ht"], c["ha"] = sum(self.cnt[n] == c["mt"] for n in members), sum(self.aff[n] == c["ma"] for n in members)
            del c["all"]
        return classes

    def merge_classes(self, parts):
        """The ranges' class records unified (csrc/ccsim_coupled.h k_cw_xunify): members add, lists merge and keep the best, a maximum is
        the largest of the ranges' and its holders are those of the ranges that reach it."""
        out = {}
        for part in parts:
            for k, c in part.items():
                o = out.get(k)
                if o is None:
                    out[k] = dict(c, list=list(c["list"]))
                    continue
                o["nf"] += c["nf"]
                o["list"] = sorted(o["list"] + c["list"])[: (self.L if self.device_plan else self.W)]
                for m, h, vals in (("mt", "ht", None), ("ma", "ha", None)):
                    if c[m] > o[m]:
                        o[m], o[h] = c[m], c[h]
                    elif c[m] == o[m]:
                        o[h] += c[h]
        for c in out.values():
            c["head"] = 0
        return out

    def run(self, limit=0, audit_sweeps=False):
        if self.node_range is None:
            self.node_range = (0, self.N)
        N, p = self.N, self.prof
        log, scans = [], 0
        stats = {"windows": 0, "classes_max": 0, "cut_by_maxima": 0, "swept": 0}
        pred = []  # audit_sweeps: winners a sweep predicted for the coming cycles
        while True:
            # ===== scan: everything below is one pass over the nodes in the state at the start of the window =====
            scans += 1
            T = self.build_tables()
            minima = self.hard_minima(T)
            # the pass over the nodes, in two steps so that it can run on node-range shards (ShardedCoupledWindowModel): the global
            # facts the local scores need (is anything feasible, the normalization maxima), then the classes of the range
            facts = self.exchange(lambda lo, hi: self.scan_facts(T, minima, lo, hi))
            if not any(f[0] for f in facts):
                return log, "Unschedulable", scans, stats
            mt, ma = max(f[1] for f in facts), max(f[2] for f in facts)
            classes = self.merge_classes(self.exchange(lambda lo, hi: self.scan_classes(T, mt, ma, lo, hi)))
            stats["windows"] += 1
            stats["classes_max"] = max(stats["classes_max"], len(classes))
            touched = []  # nodes that received a clone in this window
            done = 0
            umin = {}  # device plan: unique-key hard constraints -> [minimum, counted nodes at the minimum] as the scan saw them
            if self.device_plan:
                for i in self.hard:
                    if self.s_unique[i]:
                        vals = list(T["hard"][i].values())
                        mn = min(vals) if vals else MAXINT32
                        umin[i] = [mn, sum(v == mn for v in vals)]
            end_window = False
            # ===== decide: up to W cycles from (class heads, class counters, touched nodes, tables) only =====
            while done < self.W and not end_window:
                minima = self.hard_minima(T)
                for i, (mn, _) in umin.items():  # (the tracked value IS the minimum while a node at it remains)
                    assert minima[i] == (0 if self.n_dom[i] < self.spread[i].min_domains else mn)
                if self.device_plan and done > 0 and any(c["nf"] > 0 and c["head"] >= len(c["list"]) for c in classes.values()):
                    stats["cut_by_list"] = stats.get("cut_by_list", 0) + 1
                    pred = []  # (this model ends the window for ANY class with an exhausted list, the kernel only for a feasible one: a prediction may outlive it)
                    break  # a class's next head is not among the members the scan kept
                cand = []  # (node, class or None)
                for c in classes.values():
                    if c["nf"] > 0:
                        n = c["list"][c["head"]][1]  # an untouched member: what holds for it holds for the class
                        if self.coupled_filter(T, minima, n):
                            cand.append((n, c))
                for n in touched:
                    if self.node_feasible(n) and self.coupled_filter(T, minima, n):
                        cand.append((n, None))
                if not cand:
                    assert not pred, "a sweep predicted a cycle the loop does not run (nothing feasible)"
                    break  # nothing feasible among what the window knows: the next scan decides (it sees every node)
                # the normalization maxima the scan assumed must be the maxima of THIS cycle's feasible set
                if done > 0:
                    known = all(c is None or (c["ht"] > 0 and c["ha"] > 0) for _, c in cand)
                    mt_now = max(c["mt"] if c is not None else self.cnt[n] for n, c in cand)
                    ma_now = max(c["ma"] if c is not None else self.aff[n] for n, c in cand)
                    if not known or mt_now != mt or ma_now != ma:
                        stats["cut_by_maxima"] += 1
                        assert not pred, "a sweep predicted a cycle the loop does not run (the maxima moved)"
                        break
                nf = sum(c["nf"] if c is not None else 1 for _, c in cand)
                # PodTopologySpread score: weights from the candidate domains / feasible count, then min / max of the raw scores
                pts = None
                if self.soft:
                    n_ignored = sum((c["nf"] if c is not None else 1) for n, c in cand if not self.soft_keys[n])
                    weights = {}
                    for i in self.soft:
                        if self.spread[i].is_hostname:
                            sz = nf - n_ignored
                        else:
                            sz = len({self.sdom[i][n] for n, _ in cand if self.soft_keys[n]})
                        weights[i] = self.go_log(float(sz + 2))
                    raws = {n: self.raw_pts(T, weights, n) for n, _ in cand if self.soft_keys[n]}
                    if raws:
                        lo, hi = min(raws.values()), max(raws.values())
                    pts = {}
                    for n, _ in cand:
                        if n not in raws:
                            pts[n] = 0
                        elif hi == 0:
                            pts[n] = 100
                        else:
                            pts[n] = 100 * (hi + lo - raws[n]) // hi
                ipa = None
                if self.ipa is not None and p.w_interpodaffinity and T["entries"] > 0:
                    raws = {n: self.raw_ipa(T, n) for n, _ in cand}
                    lo, hi = min(raws.values()), max(raws.values())
                    ipa = {n: (int(100.0 * (float(raws[n] - lo) / float(hi - lo))) if hi > lo else 0) for n, _ in cand}
                best = None
                for n, c in cand:
                    a_n = -c["list"][c["head"]][0] if c is not None else self.local_score(n, mt, ma)
                    total = a_n + (pts[n] * p.w_topologyspread if pts is not None else 0) + (ipa[n] * p.w_interpodaffinity if ipa is not None else 0)
                    if best is None or total > best[0] or (total == best[0] and n < best[1]):
                        best = (total, n, c)
                _, w, c = best
                if audit_sweeps and self.device_plan:
                    if not pred and pts is None and ipa is None:
                        pred = self.sweep_predict(T, minima, cand, mt, ma, min(self.W - done, (limit - len(log)) if limit else self.W))
                        stats["swept"] += len(pred)
                    if pred:
                        assert pred[0] == w, ("a sweep predicted another winner", pred[:4], w, len(log))
                        pred = pred[1:]
                for i in umin:  # the winner leaves the minimum of a unique-key hard constraint
                    if self.hard_keys[w] and self.sincl[i][w] and self.spread[i].self_match and T["hard"][i].get(self.sdom[i][w], 0) == umin[i][0]:
                        umin[i][1] -= 1
                        if umin[i][1] == 0:
                            end_window = True  # the new minimum is not known without a pass over the nodes
                log.append(w)
                if c is not None:  # the head of a class leaves it
                    c["head"] += 1
                    c["nf"] -= 1
                    c["ht"] -= self.cnt[w] == c["mt"]
                    c["ha"] -= self.aff[w] == c["ma"]
                    touched.append(w)
                self.place(w)
                self.add_clone_to_tables(T, w)
                done += 1
                if limit and len(log) >= limit:
                    return log, "LimitReached", scans, stats


class ShardedCoupledWindowModel(CoupledWindowModel):
    """The windowed mode on node-range shards (round 5, csrc/ccsim_coupled.h k_cw_xpack / k_cw_xunify, ccsim_dist_cw_*): every rank passes
    over ITS nodes only, the ranks exchange what the pass found -- once for the global facts the local scores need, once for the window
    records (classes, their statistics, their staged list entries) -- every rank unifies the records identically and runs the deciding
    loop replicated.  (This stand-in keeps the whole cluster's state on every rank -- the real engine applies a placement on its owner and
    keeps only the domain tables replicated -- so what it checks is the exchange and the unification, on real process groups.)"""

    def __init__(self, *a, world=1, rank=0, all_gather=None, **kw):
        super().__init__(*a, **kw)
        per = -(-self.N // world)
        self.ranges = [(min(self.N, r * per), min(self.N, r * per + per)) for r in range(world)]
        self.node_range = self.ranges[rank]
        self._all_gather = all_gather  # None: every range is passed over in THIS process (the unification alone, no process group)

    def exchange(self, scan):
        if self._all_gather is None:
            return [scan(lo, hi) for lo, hi in self.ranges]
        return self._all_gather(scan(*self.node_range))
