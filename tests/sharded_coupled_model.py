"""Executable specification of ONE template with topology-coupled SCORES (PodTopologySpread ScheduleAnyway constraints, InterPodAffinity
preferred terms) on node-range SHARDS -- TEST INFRASTRUCTURE: the protocol specification from which the engine's sharded ScheduleAnyway
scoring was built (round 3: the candidate-domain sets as bitmaps, the counts and the raw-score range in the max-loc record, XRec /
xrec_soft in ccsim_kernels.h; tests/test_gpu_parity.py::test_sharded_protocol_with_schedule_anyway_constraints).  The engine needs ONE
exchange per pass instead of the three below: the scan scores under ASSUMED weights and ranges, the decision verifies them on the
gathered records and rescans when they moved, as its unsharded mode does.  The per-domain tables are replicated on every rank (the engine already all-reduces them
at load and every rank applies the winner's contribution); what a rank cannot know alone are the cycle-wide quantities the
normalizations need.  One cycle = three exchanges of one small record per rank:
  A  feasible count, ignored count, the SET of candidate domains per soft constraint (a bitmap over the constraint's domains), the
     TaintToleration / NodeAffinity maxima over the rank's feasible nodes
        -> everybody: the PodTopologySpread weights log(size + 2) (scoring.go:294-296: size = |union of the candidate sets|, or
           feasible - ignored for hostname constraints), the global maxima;
  B  min / max of the rank's raw PodTopologySpread scores and of its raw InterPodAffinity scores
        -> everybody: the two NormalizeScore ranges (scoring.go:226-265, interpodaffinity/scoring.go:259-290);
  C  the rank's best (TotalScore, lowest index) -> the winner; its owner updates the node, every rank the replicated tables.
Checked against the oracle in tests/test_sharded_coupled_model.py for 1 .. 4 shards (hard and soft constraints, inter-pod terms)."""
from __future__ import annotations

from coupled_model import CoupledWindowModel


class ShardedCoupledModel:
    def __init__(self, prof, nodes, pod, go_log, ranks):
        self.m = CoupledWindowModel(prof, nodes, pod, go_log)
        self.N, self.R = nodes.n, ranks
        per = -(-self.N // ranks)
        self.bounds = [(min(self.N, r * per), min(self.N, r * per + per)) for r in range(ranks)]
        self.exchanges = 0

    def run(self, limit=0):
        m, p = self.m, self.m.prof
        T = m.build_tables()  # replicated (all-reduced once at load)
        log = []
        while True:
            minima = m.hard_minima(T)  # from the replicated tables: every rank computes the same
            feas = [[n for n in range(lo, hi) if m.node_feasible(n) and m.coupled_filter(T, minima, n)] for lo, hi in self.bounds]
            # ---- exchange A ----
            recA = []
            for f in feas:
                live = [n for n in f if m.soft_keys[n]]
                recA.append({"nf": len(f), "ignored": len(f) - len(live), "doms": {i: {m.sdom[i][n] for n in live} for i in m.soft if not m.spread[i].is_hostname},
                             "mt": max((m.cnt[n] for n in f), default=0), "ma": max((m.aff[n] for n in f), default=0)})
            self.exchanges += 1
            nf = sum(r["nf"] for r in recA)
            if nf == 0:
                return log, "Unschedulable"
            mt, ma = max(r["mt"] for r in recA), max(r["ma"] for r in recA)
            weights = {}
            for i in m.soft:
                sz = nf - sum(r["ignored"] for r in recA) if m.spread[i].is_hostname else len(set().union(*[r["doms"][i] for r in recA]))
                weights[i] = m.go_log(float(sz + 2))
            # ---- exchange B ----
            ipa_on = m.ipa is not None and p.w_interpodaffinity and T["entries"] > 0
            raws_p = [{n: m.raw_pts(T, weights, n) for n in f if m.soft_keys[n]} if m.soft else {} for f in feas]
            raws_i = [{n: m.raw_ipa(T, n) for n in f} if ipa_on else {} for f in feas]
            self.exchanges += 1
            allp = [v for d in raws_p for v in d.values()]
            alli = [v for d in raws_i for v in d.values()]
            plo, phi = (min(allp), max(allp)) if allp else (0, 0)
            ilo, ihi = (min(alli), max(alli)) if alli else (0, 0)
            # ---- exchange C ----
            best = None
            for r, f in enumerate(feas):
                for n in f:
                    total = m.local_score(n, mt, ma)
                    if m.soft:
                        total += (0 if n not in raws_p[r] else 100 if phi == 0 else 100 * (phi + plo - raws_p[r][n]) // phi) * p.w_topologyspread
                    if ipa_on:
                        total += (int(100.0 * (float(raws_i[r][n] - ilo) / float(ihi - ilo))) if ihi > ilo else 0) * p.w_interpodaffinity
                    if best is None or total > best[0] or (total == best[0] and n < best[1]):
                        best = (total, n)
            self.exchanges += 1
            w = best[1]
            log.append(w)
            m.place(w)                 # the owner
            m.add_clone_to_tables(T, w)  # every rank
            if limit and len(log) >= limit:
                return log, "LimitReached"
