/*
 * ccref.c -- CPU ORACLE (test infrastructure only; see ccref.h header comment).
 *
 * Sequential restatement of the reference scheduling cycle.  Every function cites the
 * reference file:line it follows (S/ = vendor/k8s.io/kubernetes/pkg/scheduler, P/ = S/framework/plugins).
 * Build with -ffp-contract=off: Go on amd64 never fuses a*b+c, and BalancedAllocation /
 * PodTopologySpread scores are fp64.
 */
#include "ccref.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define MAX_NODE_SCORE 100 /* S/framework/interface.go MaxNodeScore */

/* ------------------------------------------------------------------------------------------
 * Go math.Log, pure-Go path (go1.24 src/math/log.go; FreeBSD /usr/src/lib/msun/src/e_log.c).
 * Third-party arithmetic not under /root/reference (Go standard library) -- restated from the
 * published algorithm.  Used by P/podtopologyspread/scoring.go:294-296.
 * ------------------------------------------------------------------------------------------ */
double ccref_go_log(double x) {
    const double Ln2Hi = 6.93147180369123816490e-01, Ln2Lo = 1.90821492927058770002e-10;
    const double L1 = 6.666666666666735130e-01, L2 = 3.999999999940941908e-01, L3 = 2.857142874366239149e-01,
                 L4 = 2.222219843214978396e-01, L5 = 1.818357216161805012e-01, L6 = 1.531383769920937332e-01,
                 L7 = 1.479819860511658591e-01;
    if (isnan(x) || (isinf(x) && x > 0)) return x;
    if (x < 0) return NAN;
    if (x == 0) return -INFINITY;
    int ki;
    double f1 = frexp(x, &ki); /* Go Frexp: f1 in [0.5,1) */
    if (f1 < 0.70710678118654752440 /* Sqrt2/2 */) {
        f1 *= 2;
        ki--;
    }
    double f = f1 - 1;
    double k = (double)ki;
    double s = f / (2 + f);
    double s2 = s * s;
    double s4 = s2 * s2;
    double t1 = s2 * (L1 + s4 * (L3 + s4 * (L5 + s4 * L7)));
    double t2 = s4 * (L2 + s4 * (L4 + s4 * L6));
    double R = t1 + t2;
    double hfsq = 0.5 * f * f;
    return k * Ln2Hi - ((hfsq - (s * (hfsq + R) + k * Ln2Lo)) - f);
}

/* Go math.Round: half away from zero (C99 round() has the same definition). */
static double go_round(double x) { return round(x); }

/* P/noderesources/least_allocated.go:30-61 */
static int64_t least_requested_score(int64_t requested, int64_t capacity) {
    if (capacity == 0) return 0;
    if (requested > capacity) return 0;
    return ((capacity - requested) * MAX_NODE_SCORE) / capacity;
}

int64_t ccref_least_allocated(const int64_t *requested, const int64_t *allocatable, const int64_t *weights, int n) {
    int64_t node_score = 0, weight_sum = 0;
    for (int i = 0; i < n; i++) {
        if (allocatable[i] == 0) continue;
        node_score += least_requested_score(requested[i], allocatable[i]) * weights[i];
        weight_sum += weights[i];
    }
    if (weight_sum == 0) return 0;
    return node_score / weight_sum;
}

/* P/noderesources/balanced_allocation.go:146-180 */
int64_t ccref_balanced_allocation(const int64_t *requested, const int64_t *allocatable, int n) {
    double fr[CCREF_MAX_RES];
    int m = 0;
    double total = 0;
    for (int i = 0; i < n; i++) {
        if (allocatable[i] == 0) continue;
        double fraction = (double)requested[i] / (double)allocatable[i];
        if (fraction > 1) fraction = 1;
        total += fraction;
        fr[m++] = fraction;
    }
    double std = 0.0;
    if (m == 2) {
        std = fabs((fr[0] - fr[1]) / 2);
    } else if (m > 2) {
        double mean = total / (double)m;
        double sum = 0;
        for (int i = 0; i < m; i++) sum = sum + (fr[i] - mean) * (fr[i] - mean);
        std = sqrt(sum / (double)m);
    }
    return (int64_t)((1 - std) * (double)MAX_NODE_SCORE);
}

/* P/helper/normalize_score.go:28-56 */
void ccref_default_normalize(int64_t max_priority, int reverse, int64_t *scores, int64_t n) {
    int64_t max_count = 0;
    for (int64_t i = 0; i < n; i++)
        if (scores[i] > max_count) max_count = scores[i];
    if (max_count == 0) {
        if (reverse)
            for (int64_t i = 0; i < n; i++) scores[i] = max_priority;
        return;
    }
    for (int64_t i = 0; i < n; i++) {
        int64_t score = max_priority * scores[i] / max_count;
        if (reverse) score = max_priority - score;
        scores[i] = score;
    }
}

/* S/schedule_one.go:697-723 */
int32_t ccref_num_feasible_nodes_to_find(int32_t percentage, int32_t num_all_nodes) {
    const int32_t min_feasible = 100, min_pct = 5;
    if (num_all_nodes < min_feasible) return num_all_nodes;
    if (percentage == 0) {
        percentage = 50 - num_all_nodes / 125;
        if (percentage < min_pct) percentage = min_pct;
    }
    int32_t num = (int32_t)((int64_t)num_all_nodes * percentage / 100);
    if (num < min_feasible) return min_feasible;
    return num;
}

/* P/imagelocality/image_locality.go:84-115: calculatePriority(sumImageScores(...), numContainers) */
int64_t ccref_image_locality_score(const int64_t *size, const int32_t *num_nodes, int n_present, int32_t total_nodes,
                                   int n_containers) {
    const int64_t mb = 1024 * 1024, min_threshold = 23 * mb, max_container_threshold = 1000 * mb;
    int64_t sum = 0;
    for (int i = 0; i < n_present; i++) { /* scaledImageScore :105-108 */
        double spread = (double)num_nodes[i] / (double)total_nodes;
        sum += (int64_t)((double)size[i] * spread);
    }
    int64_t max_threshold = max_container_threshold * (int64_t)n_containers;
    if (sum < min_threshold)
        sum = min_threshold;
    else if (sum > max_threshold)
        sum = max_threshold;
    return MAX_NODE_SCORE * (sum - min_threshold) / (max_threshold - min_threshold);
}

static int has_scoring(const ccref_profile *p) {
    return p->w_taint || p->w_nodeaffinity || p->w_fit || p->w_balanced || p->w_topologyspread || p->w_interpodaffinity ||
           p->w_imagelocality;
}

/* component-helpers nodeaffinity.go term.match: AND over requirements; empty term matches nothing */
static int term_matches(const ccref_nodes *nd, const ccref_pod *pod, const ccref_term *t, int64_t n) {
    if (t->n_req == 0) return 0;
    for (int i = 0; i < t->n_req; i++) {
        const ccref_requirement *r = &pod->reqs[t->first_req + i];
        if (!pod->req_tables[r->table_off + nd->label_cols[r->col][n]]) return 0;
    }
    return 1;
}

/* nodeaffinity.go:84-103 RequiredNodeAffinity.Match: nodeSelector AND (OR over required terms) */
static int required_affinity_matches(const ccref_nodes *nd, const ccref_pod *pod, int64_t n) {
    if (pod->has_node_selector) {
        /* SelectorFromSet of an empty map matches everything (nodeaffinity.go:306-310) */
        const ccref_term *t = &pod->node_selector;
        for (int i = 0; i < t->n_req; i++) {
            const ccref_requirement *r = &pod->reqs[t->first_req + i];
            if (!pod->req_tables[r->table_off + nd->label_cols[r->col][n]]) return 0;
        }
    }
    if (pod->has_required_terms) {
        int any = 0;
        for (int i = 0; i < pod->n_required && !any; i++) any = term_matches(nd, pod, &pod->required[i], n);
        if (!any) return 0;
    }
    return 1;
}

/* ------------------------------------------------------------------------------------------
 * PodTopologySpread PreFilter state (P/podtopologyspread/filtering.go:235-308)
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    int64_t *match_num[CCREF_MAX_TSC]; /* TpValueToMatchNum[c][value id]; -1 = domain absent from the map */
    int64_t min_match[CCREF_MAX_TSC];  /* CriticalPaths[c][0].MatchNum */
    int64_t n_dom[CCREF_MAX_TSC];      /* len(TpValueToMatchNum[c]) */
} pts_state;

static int node_has_all_keys(const ccref_nodes *nd, const ccref_pod *pod, int hard, int64_t n) {
    for (int c = 0; c < pod->n_spread; c++) {
        if (pod->spread[c].hard != hard) continue;
        if (nd->label_cols[pod->spread[c].col][n] == 0) return 0;
    }
    return 1;
}

/* number of pods on node n matching constraint c: existing + simulated clones (types.go:345-350 AddPod) */
static int64_t node_match_count(const ccref_spread_constraint *c, const int32_t *placed, int64_t n) {
    int64_t cnt = c->node_match_count ? c->node_match_count[n] : 0;
    if (c->self_match) cnt += placed[n];
    return cnt;
}

static void pts_prefilter(const ccref_nodes *nd, const ccref_pod *pod, const int32_t *placed, pts_state *s) {
    for (int c = 0; c < pod->n_spread; c++) {
        const ccref_spread_constraint *k = &pod->spread[c];
        if (!k->hard) continue;
        for (int64_t v = 0; v <= k->n_domains; v++) s->match_num[c][v] = -1;
    }
    for (int64_t n = 0; n < nd->n; n++) {
        if (!node_has_all_keys(nd, pod, 1, n)) continue; /* filtering.go:267-270 */
        for (int c = 0; c < pod->n_spread; c++) {
            const ccref_spread_constraint *k = &pod->spread[c];
            if (!k->hard) continue;
            if (k->node_included && !k->node_included[n]) continue; /* :274-277 */
            int32_t v = nd->label_cols[k->col][n];
            if (s->match_num[c][v] < 0) s->match_num[c][v] = 0;
            s->match_num[c][v] += node_match_count(k, placed, n);
        }
    }
    for (int c = 0; c < pod->n_spread; c++) {
        const ccref_spread_constraint *k = &pod->spread[c];
        if (!k->hard) continue;
        int64_t mn = 2147483647LL /* math.MaxInt32, filtering.go:105 */, nd_ = 0;
        for (int64_t v = 1; v <= k->n_domains; v++)
            if (s->match_num[c][v] >= 0) {
                nd_++;
                if (s->match_num[c][v] < mn) mn = s->match_num[c][v];
            }
        s->min_match[c] = mn;
        s->n_dom[c] = nd_;
    }
}

/* filtering.go:311-356; returns 0 ok, 1 missing label (Unresolvable), 2 skew (Unschedulable) */
static int pts_filter(const ccref_nodes *nd, const ccref_pod *pod, const pts_state *s, int64_t n) {
    for (int c = 0; c < pod->n_spread; c++) {
        const ccref_spread_constraint *k = &pod->spread[c];
        if (!k->hard) continue;
        int32_t v = nd->label_cols[k->col][n];
        if (v == 0) return 1;
        int64_t min_match = s->min_match[c];
        if (s->n_dom[c] < k->min_domains) min_match = 0; /* filtering.go:56-69 */
        int64_t match_num = s->match_num[c][v] < 0 ? 0 : s->match_num[c][v];
        int64_t skew = match_num + (k->self_match ? 1 : 0) - min_match;
        if (skew > k->max_skew) return 2;
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * InterPodAffinity PreFilter / PreScore state (P/interpodaffinity/filtering.go:204-309, scoring.go:128-221),
 * rebuilt every cycle like the reference does.  Count/score maps are keyed by topology pair, i.e. per KEY.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    int64_t *aff[CCREF_MAX_IPA_KEYS], *anti[CCREF_MAX_IPA_KEYS], *exist[CCREF_MAX_IPA_KEYS], *score[CCREF_MAX_IPA_KEYS];
    int64_t aff_total;   /* entries of affinityCounts (len == 0 enables the first-pod exception, filtering.go:396-405) */
    int64_t exist_total; /* entries of existingAntiAffinityCounts */
    int64_t entries;     /* processTerm hits of PreScore (0 -> Skip, scoring.go:199-201) */
    int filter_active;   /* PreFilter did not return Skip (filtering.go:299-301) */
} ipa_state;

static void ipa_build(const ccref_nodes *nd, const ccref_pod *pod, const int32_t *placed, ipa_state *s) {
    const ccref_ipa *a = &pod->ipa;
    for (int k = 0; k < a->n_keys; k++) {
        size_t len = sizeof(int64_t) * (size_t)(a->key_ndom[k] + 1);
        memset(s->aff[k], 0, len);
        memset(s->anti[k], 0, len);
        memset(s->exist[k], 0, len);
        memset(s->score[k], 0, len);
    }
    s->aff_total = s->exist_total = 0;
    s->entries = a->entries_existing;
    for (int64_t n = 0; n < nd->n; n++) {
        const int64_t clones = placed[n]; /* every simulated clone is an existing pod of the next cycle */
        /* incoming affinity terms: a pod counts only if it matches ALL terms, once per term (filtering.go:165-173) */
        int64_t am = (a->aff_existing ? a->aff_existing[n] : 0) + (a->self_aff ? clones : 0);
        if (a->n_aff_terms && am)
            for (int t = 0; t < a->n_aff_terms; t++) {
                int k = a->aff_key[t];
                int32_t v = nd->label_cols[a->key_col[k]][n];
                if (v) s->aff[k][v] += am, s->aff_total += am;
            }
        /* incoming anti-affinity terms, per term (filtering.go:177-184) */
        for (int t = 0; t < a->n_anti_terms; t++) {
            int64_t m = (a->anti_existing[t] ? a->anti_existing[t][n] : 0) + (a->anti_self[t] ? clones : 0);
            int k = a->anti_key[t];
            int32_t v = nd->label_cols[a->key_col[k]][n];
            if (m && v) s->anti[k][v] += m;
        }
        /* existing pods' anti-affinity terms matching the incoming pod (filtering.go:204-232); a clone carries the
           incoming pod's own terms */
        for (int k = 0; k < a->n_keys; k++) {
            int64_t m = a->exist_anti[k] ? a->exist_anti[k][n] : 0;
            for (int t = 0; t < a->n_anti_terms; t++)
                if (a->anti_key[t] == k && a->anti_self[t]) m += clones;
            int32_t v = nd->label_cols[a->key_col[k]][n];
            if (m && v) s->exist[k][v] += m, s->exist_total += m;
            /* score map (scoring.go:81-125) */
            int64_t w = (a->score_existing[k] ? a->score_existing[k][n] : 0) + clones * a->score_self[k];
            if (v) {
                s->score[k][v] += w;
                s->entries += clones * a->self_entries[k];
            }
        }
    }
    s->filter_active = !(s->exist_total == 0 && a->n_aff_terms == 0 && a->n_anti_terms == 0);
}

/* InterPodAffinity.Score (scoring.go:235-257): the sum of the score map's entries at the node's own topology values */
static void ipa_raw_scores(const ccref_nodes *nd, const ccref_pod *pod, const ipa_state *s, const int64_t *feas, int64_t nf, int64_t *out) {
    const ccref_ipa *a = &pod->ipa;
    for (int64_t i = 0; i < nf; i++) {
        int64_t v = 0;
        for (int k = 0; k < a->n_keys; k++) {
            int32_t d = nd->label_cols[a->key_col[k]][feas[i]];
            if (d) v += s->score[k][d];
        }
        out[i] = v;
    }
}

/* filtering.go:352-432; returns 0 ok, 1 affinity (Unresolvable), 2 anti-affinity, 3 existing pods' anti-affinity */
static int ipa_filter(const ccref_nodes *nd, const ccref_pod *pod, const ipa_state *s, int64_t n) {
    const ccref_ipa *a = &pod->ipa;
    int pods_exist = 1;
    for (int t = 0; t < a->n_aff_terms; t++) { /* satisfyPodAffinity :382-408 */
        int k = a->aff_key[t];
        int32_t v = nd->label_cols[a->key_col[k]][n];
        if (!v) return 1; /* all topology labels must exist on the node */
        if (s->aff[k][v] <= 0) pods_exist = 0;
    }
    if (!pods_exist && !(s->aff_total == 0 && a->self_aff)) return 1;
    for (int t = 0; t < a->n_anti_terms; t++) { /* satisfyPodAntiAffinity :367-379 */
        int k = a->anti_key[t];
        int32_t v = nd->label_cols[a->key_col[k]][n];
        if (v && s->anti[k][v] > 0) return 2;
    }
    if (s->exist_total > 0) /* satisfyExistingPodsAntiAffinity :352-364 */
        for (int k = 0; k < a->n_keys; k++) {
            int32_t v = nd->label_cols[a->key_col[k]][n];
            if (v && s->exist[k][v] > 0) return 3;
        }
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * Filter chain for one node: S/framework/runtime/framework.go:897-930 (short-circuit AND in the
 * configured order; first failing plugin sets the status).  Returns 0 if feasible, else a
 * negative code: -1 Unschedulable, -2 UnschedulableAndUnresolvable; *first_plugin receives the
 * failing plugin bit; fit_reasons receives a bitmask over {pods, col0, col1, ...} for Fit.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    uint32_t plugin;
    uint32_t fit_mask; /* bit0 Too many pods, bit 1+col Insufficient <col> */
    int pts_code;
    int ipa_code;
    int vol_code; /* 1..CCREF_VOL_CODES */
} fail_info;

static int filter_node(const ccref_profile *prof, const ccref_nodes *nd, const ccref_pod *pod, const pts_state *pts,
                       const ipa_state *ipa, const int32_t *placed, int64_t n, fail_info *fi) {
    uint32_t fm = prof->filter_mask;
    /* P/nodeunschedulable/node_unschedulable.go:133-150 */
    if ((fm & CCREF_F_UNSCHEDULABLE) && nd->unschedulable && nd->unschedulable[n] && !pod->tolerates_unschedulable) {
        fi->plugin = CCREF_F_UNSCHEDULABLE;
        return -2;
    }
    /* P/nodename: pod.Spec.NodeName is always "" for generated pods (podgenerator.go:27-46) */
    /* P/tainttoleration/taint_toleration.go:111-121 */
    if ((fm & CCREF_F_TAINT) && nd->taintset_id && !pod->taint_filter_ok[nd->taintset_id[n]]) {
        fi->plugin = CCREF_F_TAINT;
        return -2;
    }
    /* P/nodeaffinity/node_affinity.go:206-227 (skipped when PreFilter returned Skip :149-155) */
    if ((fm & CCREF_F_NODEAFFINITY) && pod->affinity_filter_active && !required_affinity_matches(nd, pod, n)) {
        fi->plugin = CCREF_F_NODEAFFINITY;
        return -2;
    }
    /* P/nodeports/node_ports.go:148-176 (PreFilter Skip for pods without host ports :67-76): the node's UsedPorts are
     * those of its existing pods plus those of the clones placed on it */
    if ((fm & CCREF_F_NODEPORTS) && pod->has_host_ports &&
        ((pod->host_ports_conflict && pod->host_ports_conflict[n]) || (placed && placed[n] > 0))) {
        fi->plugin = CCREF_F_NODEPORTS;
        return -1;
    }
    /* P/noderesources/fit.go:564-660 fitsRequest: ALL insufficient resources are kept */
    if (fm & CCREF_F_FIT) {
        uint32_t mask = 0;
        int unresolvable = 0;
        if ((int64_t)nd->pod_count[n] + 1 > (int64_t)nd->alloc_pods[n]) mask |= 1u;
        int all_zero = pod->req[0] == 0 && pod->req[1] == 0 && pod->req[2] == 0 && !pod->has_scalar_entries;
        if (!all_zero) {
            int ncol = 3 + nd->n_scalar;
            for (int c = 0; c < ncol; c++) {
                int64_t rq = pod->req[c];
                if (c < 3 ? !(rq > 0) : rq == 0) continue;
                int64_t alloc = nd->alloc[c] ? nd->alloc[c][n] : 0;
                int64_t used = nd->req[c] ? nd->req[c][n] : 0;
                if (rq > alloc - used) {
                    mask |= 1u << (1 + c);
                    if (rq > alloc) unresolvable = 1;
                }
            }
        }
        if (mask) {
            fi->plugin = CCREF_F_FIT;
            fi->fit_mask = mask;
            return unresolvable ? -2 : -1;
        }
    }
    /* VolumeRestrictions, NodeVolumeLimits, VolumeBinding, VolumeZone (default_plugins.go:41-44), evaluated by the caller against the
     * snapshot's pods (volume_veto: the first of them that rejects the node); a clone's own disks (volume_restrictions.go:105-150,
     * 310-313: the first check of the first of the four) */
    if ((pod->volume_exclusive && placed && placed[n] > 0) || (pod->volume_veto && pod->volume_veto[n])) {
        fi->plugin = CCREF_F_VOLUMES;
        fi->vol_code = (pod->volume_exclusive && placed && placed[n] > 0) ? 1 : (int)pod->volume_veto[n];
        return fi->vol_code <= CCREF_VOL_LAST_UNSCHEDULABLE ? -1 : -2;
    }
    /* P/podtopologyspread/filtering.go:311-356 */
    if ((fm & CCREF_F_TOPOLOGYSPREAD) && pts) {
        int r = pts_filter(nd, pod, pts, n);
        if (r) {
            fi->plugin = CCREF_F_TOPOLOGYSPREAD;
            fi->pts_code = r;
            return r == 1 ? -2 : -1;
        }
    }
    /* P/interpodaffinity/filtering.go:410-432 */
    if ((fm & CCREF_F_INTERPODAFFINITY) && ipa && ipa->filter_active) {
        int r = ipa_filter(nd, pod, ipa, n);
        if (r) {
            fi->plugin = CCREF_F_INTERPODAFFINITY;
            fi->ipa_code = r;
            return r == 1 ? -2 : -1;
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * Score plugins (S/framework/runtime/framework.go:1137-1244) over the feasible list.
 * ------------------------------------------------------------------------------------------ */
static int64_t col_alloc(const ccref_nodes *nd, int c, int64_t n) { return nd->alloc[c] ? nd->alloc[c][n] : 0; }
static int64_t col_req(const ccref_nodes *nd, int c, int64_t n) { return nd->req[c] ? nd->req[c][n] : 0; }

/* P/noderesources/resource_allocation.go:48-114 with useRequested=false (NodeResourcesFit scorer) */
static int64_t fit_score(const ccref_profile *prof, const ccref_nodes *nd, const ccref_pod *pod, int64_t n) {
    int64_t requested[CCREF_MAX_RES], allocatable[CCREF_MAX_RES];
    for (int i = 0; i < prof->n_fit_res; i++) {
        int c = prof->fit_res[i];
        /* pod request for this resource: cpu/mem use the NonZero defaults (:118-148) */
        int64_t pod_req = c == 0 ? pod->nz_mcpu : c == 1 ? pod->nz_mem : pod->req[c];
        int64_t alloc, req;
        if (c >= 3 && pod_req == 0) { /* :97-99 scalar the pod does not request is bypassed */
            alloc = 0;
            req = 0;
        } else if (c == 0) {
            alloc = col_alloc(nd, 0, n);
            req = nd->nz_mcpu[n] + pod_req;
        } else if (c == 1) {
            alloc = col_alloc(nd, 1, n);
            req = nd->nz_mem[n] + pod_req;
        } else {
            alloc = col_alloc(nd, c, n);
            req = col_req(nd, c, n) + pod_req;
        }
        allocatable[i] = 0;
        requested[i] = 0;
        if (alloc == 0) continue; /* :66-69 */
        allocatable[i] = alloc;
        requested[i] = req;
    }
    return ccref_least_allocated(requested, allocatable, prof->fit_res_w, prof->n_fit_res);
}

/* same with useRequested=true (balanced_allocation.go:140): raw requests on both sides */
static int64_t balanced_score(const ccref_profile *prof, const ccref_nodes *nd, const ccref_pod *pod, int64_t n) {
    int64_t requested[CCREF_MAX_RES], allocatable[CCREF_MAX_RES];
    for (int i = 0; i < prof->n_bal_res; i++) {
        int c = prof->bal_res[i];
        int64_t pod_req = pod->req[c];
        int64_t alloc, req;
        if (c >= 3 && pod_req == 0) {
            alloc = 0;
            req = 0;
        } else {
            alloc = col_alloc(nd, c, n);
            req = col_req(nd, c, n) + pod_req;
        }
        allocatable[i] = 0;
        requested[i] = 0;
        if (alloc == 0) continue;
        allocatable[i] = alloc;
        requested[i] = req;
    }
    return ccref_balanced_allocation(requested, allocatable, prof->n_bal_res);
}

/* balanced_allocation.go:66-79 isBestEffortPod over the plugin's resource list */
static int balanced_skipped(const ccref_profile *prof, const ccref_pod *pod) {
    for (int i = 0; i < prof->n_bal_res; i++)
        if (pod->req[prof->bal_res[i]] != 0) return 0;
    return 1;
}

static int has_soft_spread(const ccref_pod *pod) {
    for (int c = 0; c < pod->n_spread; c++)
        if (!pod->spread[c].hard) return 1;
    return 0;
}
static int has_hard_spread(const ccref_pod *pod) {
    for (int c = 0; c < pod->n_spread; c++)
        if (pod->spread[c].hard) return 1;
    return 0;
}

/* PodTopologySpread.NormalizeScore (P/podtopologyspread/scoring.go:226-265): `ignored` = the node is in IgnoredNodes (NULL: none) */
void ccref_pts_normalize(int64_t *scores, const uint8_t *ignored, int64_t n) {
    int64_t min_score = INT64_MAX, max_score = 0;
    for (int64_t i = 0; i < n; i++) {
        if (ignored && ignored[i]) continue;
        if (scores[i] < min_score) min_score = scores[i];
        if (scores[i] > max_score) max_score = scores[i];
    }
    for (int64_t i = 0; i < n; i++) {
        if (ignored && ignored[i]) {
            scores[i] = 0;
            continue;
        }
        if (max_score == 0) {
            scores[i] = MAX_NODE_SCORE;
            continue;
        }
        scores[i] = MAX_NODE_SCORE * (max_score + min_score - scores[i]) / max_score;
    }
}

/* InterPodAffinity.NormalizeScore (P/interpodaffinity/scoring.go:259-290) */
void ccref_ipa_normalize(int64_t *scores, int64_t n) {
    int64_t mn = INT64_MAX, mx = INT64_MIN;
    for (int64_t i = 0; i < n; i++) {
        if (scores[i] > mx) mx = scores[i];
        if (scores[i] < mn) mn = scores[i];
    }
    const int64_t diff = mx - mn;
    for (int64_t i = 0; i < n; i++) {
        double f = 0;
        if (diff > 0) f = (double)MAX_NODE_SCORE * ((double)(scores[i] - mn) / (double)diff);
        scores[i] = (int64_t)f;
    }
}

/* P/podtopologyspread/scoring.go:61-265 PreScore + Score + NormalizeScore over the feasible list */
static void pts_scores_ex(const ccref_nodes *nd, const ccref_pod *pod, const int32_t *placed, const int64_t *feas,
                          int64_t nf, int64_t *out, int64_t *raw_out, double *weight_out) {
    int64_t *cnt[CCREF_MAX_TSC];
    double weight[CCREF_MAX_TSC];
    int64_t topo_size[CCREF_MAX_TSC];
    uint8_t *ignored = (uint8_t *)calloc((size_t)nf, 1);
    int64_t n_ignored = 0;
    for (int c = 0; c < pod->n_spread; c++) {
        cnt[c] = NULL;
        topo_size[c] = 0;
        if (pod->spread[c].hard) continue;
        cnt[c] = (int64_t *)malloc(sizeof(int64_t) * (size_t)(pod->spread[c].n_domains + 1));
        for (int64_t v = 0; v <= pod->spread[c].n_domains; v++) cnt[c][v] = -1; /* nil */
    }
    /* initPreScoreState :61-115.  requireAllTopologies (:140) is true when the pod carries its own constraints; with the plugin's
     * system defaults (pod->soft_relaxed) no node is ignored and a missing key is the value "" (id 0 here) */
    const int require_all = !pod->soft_relaxed;
    for (int64_t i = 0; i < nf; i++) {
        int64_t n = feas[i];
        if (require_all && !node_has_all_keys(nd, pod, 0, n)) {
            ignored[i] = 1;
            n_ignored++;
            continue;
        }
        for (int c = 0; c < pod->n_spread; c++) {
            const ccref_spread_constraint *k = &pod->spread[c];
            if (k->hard || k->is_hostname) continue;
            int32_t v = nd->label_cols[k->col][n];
            if (cnt[c][v] < 0) {
                cnt[c][v] = 0;
                topo_size[c]++;
            }
        }
    }
    for (int c = 0; c < pod->n_spread; c++) {
        const ccref_spread_constraint *k = &pod->spread[c];
        if (k->hard) continue;
        int64_t sz = k->is_hostname ? nf - n_ignored : topo_size[c];
        weight[c] = ccref_go_log((double)(sz + 2)); /* :294-296 */
    }
    /* PreScore :147-178: count matching pods over ALL nodes into candidate domains */
    for (int64_t n = 0; n < nd->n; n++) {
        if (require_all && !node_has_all_keys(nd, pod, 0, n)) continue; /* :161-164 */
        for (int c = 0; c < pod->n_spread; c++) {
            const ccref_spread_constraint *k = &pod->spread[c];
            if (k->hard || k->is_hostname) continue;
            if (k->node_included && !k->node_included[n]) continue;
            int32_t v = nd->label_cols[k->col][n];
            if (cnt[c][v] < 0) continue;
            cnt[c][v] += node_match_count(k, placed, n);
        }
    }
    /* Score :196-223 */
    for (int64_t i = 0; i < nf; i++) {
        int64_t n = feas[i];
        if (ignored[i]) {
            out[i] = 0;
            continue;
        }
        double score = 0;
        for (int c = 0; c < pod->n_spread; c++) {
            const ccref_spread_constraint *k = &pod->spread[c];
            if (k->hard) continue;
            int32_t v = nd->label_cols[k->col][n];
            if (v == 0) continue; /* `if tpVal, ok := node.Labels[c.TopologyKey]; ok` :210 -- only reachable with soft_relaxed */
            int64_t ct = k->is_hostname ? node_match_count(k, placed, n) : cnt[c][v];
            score += (double)ct * weight[c] + (double)(k->max_skew - 1); /* scoreForCount :302-304 */
        }
        out[i] = (int64_t)go_round(score);
    }
    if (raw_out) memcpy(raw_out, out, sizeof(int64_t) * (size_t)nf);
    if (weight_out)
        for (int c = 0; c < pod->n_spread; c++) weight_out[c] = pod->spread[c].hard ? 0.0 : weight[c];
    ccref_pts_normalize(out, ignored, nf);
    for (int c = 0; c < pod->n_spread; c++) free(cnt[c]);
    free(ignored);
}

static void pts_scores(const ccref_nodes *nd, const ccref_pod *pod, const int32_t *placed, const int64_t *feas, int64_t nf, int64_t *out) {
    pts_scores_ex(nd, pod, placed, feas, nf, out, NULL, NULL);
}

/* ------------------------------------------------------------------------------------------
 * One cycle: schedulePod (S/schedule_one.go:430-478), findNodesThatFitPod (:482-564),
 * findNodesThatPassFilters (:610-693) in CANONICAL mode = one worker visiting
 * (start+i)%N in order and stopping when the (K+1)-th feasible node is met, prioritizeNodes
 * (:776-890), selectHost (:894-941) with tie-break = lowest position in the feasible list
 * (the reference draws uniformly from the same set), assume (:967-984 -> types.go:409-428).
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    int64_t *feas;    /* feasible node indices in visit order */
    int64_t *total;   /* TotalScore */
    int64_t *scratch; /* per-plugin scores */
    int8_t *status;   /* per node: 0 unvisited/feasible, -1, -2 */
    fail_info *fails;
    int32_t *placed;  /* simulated pods per node (== res->per_node_count) */
    pts_state pts;
    ipa_state ipa;
    int threads;
} workspace;

/* RunScorePlugins' last block (S/framework/runtime/framework.go:1214-1238): plugin weight x normalized score, summed per node */
void ccref_weigh(int64_t *total, const int64_t *scores, int64_t weight, int64_t n) {
    for (int64_t i = 0; i < n; i++) total[i] += scores[i] * weight;
}

/* selectHost (S/schedule_one.go:894-941) under the canonical tie-break: the reference samples uniformly among the nodes
 * of the maximum TotalScore; the FIRST of them in feasible-list order is the outcome chosen here */
int64_t ccref_select_host(const int64_t *total, int64_t n) {
    int64_t best = 0;
    for (int64_t i = 1; i < n; i++)
        if (total[i] > total[best]) best = i;
    return best;
}

static int64_t schedule_one_ws(const ccref_profile *prof, ccref_nodes *nd, const ccref_pod *pod, ccref_sched_state *st,
                               ccref_result *res, workspace *ws) {
    const int64_t N = nd->n;
    int32_t num_to_find = ccref_num_feasible_nodes_to_find(prof->percentage_of_nodes_to_score, (int32_t)N);
    if (!has_scoring(prof)) num_to_find = 1; /* :619-621 */

    /* PreFilter: PodTopologySpread state is rebuilt every cycle (filtering.go:140-150,235-308) */
    const pts_state *pts = NULL;
    if ((prof->filter_mask & CCREF_F_TOPOLOGYSPREAD) && has_hard_spread(pod)) {
        pts_prefilter(nd, pod, ws->placed, &ws->pts);
        pts = &ws->pts;
    }

    /* InterPodAffinity PreFilter + PreScore state */
    const ipa_state *ipa = NULL;
    if (pod->has_ipa) {
        ipa_build(nd, pod, ws->placed, &ws->ipa);
        ipa = &ws->ipa;
    }

    int64_t nf = 0, visited = 0, failed = 0;
    const int64_t start = st->next_start_node_index;
    if (num_to_find >= N) {
        /* every node is visited: order-independent, so the node loop may run in parallel */
#pragma omp parallel for schedule(static) num_threads(ws->threads) if (ws->threads > 1)
        for (int64_t i = 0; i < N; i++) {
            int64_t n = (start + i) % N;
            fail_info fi = {0, 0, 0, 0};
            int r = filter_node(prof, nd, pod, pts, ipa, ws->placed, n, &fi);
            ws->status[n] = (int8_t)r;
            if (r) ws->fails[n] = fi;
        }
        for (int64_t i = 0; i < N; i++) {
            int64_t n = (start + i) % N;
            if (ws->status[n] == 0)
                ws->feas[nf++] = n;
            else
                failed++;
        }
        visited = N;
    } else {
        for (int64_t i = 0; i < N; i++) {
            int64_t n = (start + i) % N;
            fail_info fi = {0, 0, 0, 0};
            int r = filter_node(prof, nd, pod, pts, ipa, ws->placed, n, &fi);
            if (r == 0) {
                if (nf == num_to_find) break; /* :655-662 the (K+1)-th feasible node cancels the search */
                ws->feas[nf++] = n;
                ws->status[n] = 0;
            } else {
                ws->status[n] = (int8_t)r;
                ws->fails[n] = fi;
                failed++;
            }
            visited++;
        }
    }
    /* :538-539 */
    st->next_start_node_index = (st->next_start_node_index + nf + failed) % N;
    if (res) {
        res->evaluated_total += nf + failed;
        res->last_evaluated = (int32_t)(nf + failed);
        res->last_feasible = (int32_t)nf;
    }

    if (nf == 0) {
        /* FitError diagnosis (types.go:787-836): histogram of every reason of every node */
        if (res) {
            memset(res->hist, 0, sizeof(res->hist));
            res->n_code_unschedulable = 0;
            if (res->hist_taintset) memset(res->hist_taintset, 0, sizeof(int64_t) * (size_t)pod->n_taintsets);
            for (int64_t n = 0; n < N; n++) {
                const fail_info *fi = &ws->fails[n];
                if (ws->status[n] == -1) res->n_code_unschedulable++;
                switch (fi->plugin) {
                case CCREF_F_UNSCHEDULABLE: res->hist[CCREF_R_UNSCHEDULABLE]++; break;
                case CCREF_F_TAINT:
                    if (res->hist_taintset) res->hist_taintset[nd->taintset_id[n]]++;
                    break;
                case CCREF_F_NODEAFFINITY: res->hist[CCREF_R_NODEAFFINITY]++; break;
                case CCREF_F_NODEPORTS: res->hist[CCREF_R_NODEPORTS]++; break;
                case CCREF_F_VOLUMES: res->hist[CCREF_R_VOL0 + fi->vol_code - 1]++; break;
                case CCREF_F_FIT:
                    if (fi->fit_mask & 1u) res->hist[CCREF_R_TOO_MANY_PODS]++;
                    for (int c = 0; c < CCREF_MAX_RES; c++)
                        if (fi->fit_mask & (1u << (1 + c))) res->hist[CCREF_R_RES0 + c]++;
                    break;
                case CCREF_F_TOPOLOGYSPREAD:
                    res->hist[fi->pts_code == 1 ? CCREF_R_PTS_MISSING_LABEL : CCREF_R_PTS_SKEW]++;
                    break;
                case CCREF_F_INTERPODAFFINITY:
                    res->hist[fi->ipa_code == 1 ? CCREF_R_IPA_AFFINITY : fi->ipa_code == 2 ? CCREF_R_IPA_ANTI : CCREF_R_IPA_EXISTING_ANTI]++;
                    break;
                default: break;
                }
            }
        }
        return -1;
    }

    int64_t winner;
    if (nf == 1) {
        winner = ws->feas[0]; /* :457-463 */
    } else if (!has_scoring(prof)) {
        winner = ws->feas[0]; /* prioritizeNodes :787-796: all TotalScore 1 -> first in list */
    } else {
        memset(ws->total, 0, sizeof(int64_t) * (size_t)nf);
        int64_t *sc = ws->scratch;
        /* TaintToleration Score + NormalizeScore(reverse) (taint_toleration.go:184-199) */
        if (prof->w_taint) {
            for (int64_t i = 0; i < nf; i++)
                sc[i] = nd->taintset_id ? pod->taint_prefer_cnt[nd->taintset_id[ws->feas[i]]] : 0;
            ccref_default_normalize(MAX_NODE_SCORE, 1, sc, nf);
            ccref_weigh(ws->total, sc, prof->w_taint, nf);
        }
        /* NodeAffinity Score + NormalizeScore (node_affinity.go:260-290); Skip without preferred terms */
        if (prof->w_nodeaffinity && pod->n_preferred > 0) {
#pragma omp parallel for schedule(static) num_threads(ws->threads) if (ws->threads > 1)
            for (int64_t i = 0; i < nf; i++) {
                int64_t count = 0;
                for (int t = 0; t < pod->n_preferred; t++)
                    if (term_matches(nd, pod, &pod->preferred[t], ws->feas[i])) count += pod->preferred[t].weight;
                sc[i] = count;
            }
            ccref_default_normalize(MAX_NODE_SCORE, 0, sc, nf);
            ccref_weigh(ws->total, sc, prof->w_nodeaffinity, nf);
        }
        /* NodeResourcesFit LeastAllocated (fit.go:663-672); no normalization */
        if (prof->w_fit) {
#pragma omp parallel for schedule(static) num_threads(ws->threads) if (ws->threads > 1)
            for (int64_t i = 0; i < nf; i++) ws->total[i] += fit_score(prof, nd, pod, ws->feas[i]) * prof->w_fit;
        }
        /* PodTopologySpread soft constraints */
        if (prof->w_topologyspread && has_soft_spread(pod)) {
            pts_scores(nd, pod, ws->placed, ws->feas, nf, sc);
            ccref_weigh(ws->total, sc, prof->w_topologyspread, nf);
        }
        /* InterPodAffinity Score + NormalizeScore (scoring.go:226-290); PreScore Skip without any term hit */
        if (prof->w_interpodaffinity && ipa && ipa->entries > 0) {
            ipa_raw_scores(nd, pod, ipa, ws->feas, nf, sc);
            ccref_ipa_normalize(sc, nf);
            ccref_weigh(ws->total, sc, prof->w_interpodaffinity, nf);
        }
        /* NodeResourcesBalancedAllocation (balanced_allocation.go:100-115); Skip for best-effort */
        if (prof->w_balanced && !balanced_skipped(prof, pod)) {
#pragma omp parallel for schedule(static) num_threads(ws->threads) if (ws->threads > 1)
            for (int64_t i = 0; i < nf; i++)
                ws->total[i] += balanced_score(prof, nd, pod, ws->feas[i]) * prof->w_balanced;
        }
        /* ImageLocality (image_locality.go:54-66): no PreScore, no NormalizeScore */
        if (prof->w_imagelocality && pod->image_score)
            for (int64_t i = 0; i < nf; i++) ws->total[i] += (int64_t)pod->image_score[ws->feas[i]] * prof->w_imagelocality;
        /* selectHost, canonical tie-break: first maximum in feasible-list order */
        winner = ws->feas[ccref_select_host(ws->total, nf)];
    }

    /* assume -> NodeInfo.AddPod -> update (types.go:409-428) */
    int ncol = 3 + nd->n_scalar;
    for (int c = 0; c < ncol; c++)
        if (nd->req[c]) nd->req[c][winner] += pod->req[c];
    nd->nz_mcpu[winner] += pod->nz_mcpu;
    nd->nz_mem[winner] += pod->nz_mem;
    nd->pod_count[winner] += 1;
    ws->placed[winner] += 1;
    return winner;
}

static int ws_init(workspace *ws, const ccref_nodes *nd, const ccref_pod *pod, int32_t *placed, int threads) {
    memset(ws, 0, sizeof(*ws));
    size_t n = (size_t)(nd->n > 0 ? nd->n : 1);
    ws->feas = (int64_t *)malloc(sizeof(int64_t) * n);
    ws->total = (int64_t *)malloc(sizeof(int64_t) * n);
    ws->scratch = (int64_t *)malloc(sizeof(int64_t) * n);
    ws->status = (int8_t *)calloc(n, 1);
    ws->fails = (fail_info *)calloc(n, sizeof(fail_info));
    ws->placed = placed;
    ws->threads = threads > 0 ? threads : 1;
    for (int c = 0; c < pod->n_spread; c++)
        if (pod->spread[c].hard)
            ws->pts.match_num[c] = (int64_t *)malloc(sizeof(int64_t) * (size_t)(pod->spread[c].n_domains + 1));
    if (pod->has_ipa)
        for (int k = 0; k < pod->ipa.n_keys; k++) {
            size_t len = sizeof(int64_t) * (size_t)(pod->ipa.key_ndom[k] + 1);
            ws->ipa.aff[k] = (int64_t *)malloc(len);
            ws->ipa.anti[k] = (int64_t *)malloc(len);
            ws->ipa.exist[k] = (int64_t *)malloc(len);
            ws->ipa.score[k] = (int64_t *)malloc(len);
        }
    return ws->feas && ws->total && ws->scratch && ws->status && ws->fails ? 0 : -1;
}

static void ws_free(workspace *ws) {
    free(ws->feas);
    free(ws->total);
    free(ws->scratch);
    free(ws->status);
    free(ws->fails);
    for (int c = 0; c < CCREF_MAX_TSC; c++) free(ws->pts.match_num[c]);
    for (int k = 0; k < CCREF_MAX_IPA_KEYS; k++) {
        free(ws->ipa.aff[k]);
        free(ws->ipa.anti[k]);
        free(ws->ipa.exist[k]);
        free(ws->ipa.score[k]);
    }
}

/* Test hooks (tests/test_reference_vectors.py): the PreFilter states of the two topology-coupled plugins for a cluster state, so that the
 * counting loops above can be held against the reference's own (calPreFilterState; updateWithAffinityTerms / updateWithAntiAffinityTerms). */
int ccref_unit_pts_prefilter(const ccref_nodes *nd, const ccref_pod *pod, const int32_t *placed, int c, int64_t *match_num, int64_t *min_match,
                             int64_t *n_dom) {
    if (c < 0 || c >= pod->n_spread || !pod->spread[c].hard) return -1;
    pts_state s;
    memset(&s, 0, sizeof s);
    for (int j = 0; j < pod->n_spread; j++)
        if (pod->spread[j].hard) s.match_num[j] = (int64_t *)malloc(sizeof(int64_t) * (size_t)(pod->spread[j].n_domains + 1));
    pts_prefilter(nd, pod, placed, &s);
    memcpy(match_num, s.match_num[c], sizeof(int64_t) * (size_t)(pod->spread[c].n_domains + 1));
    *min_match = s.min_match[c], *n_dom = s.n_dom[c];
    for (int j = 0; j < CCREF_MAX_TSC; j++) free(s.match_num[j]);
    return 0;
}

/* PodTopologySpread PreScore + Score + NormalizeScore (scoring.go:61-265) over the feasible list `feas`: raw scores (math.Round'ed), the
 * normalized ones, the per-constraint weights -- what tests/test_reference_vectors.py holds against the reference's own functions */
int ccref_unit_pts_scores(const ccref_nodes *nd, const ccref_pod *pod, const int32_t *placed, const int64_t *feas, int64_t nf, int64_t *raw,
                          int64_t *norm, double *weights) {
    if (nf < 0) return -1;
    pts_scores_ex(nd, pod, placed, feas, nf, norm, raw, weights);
    return 0;
}

int ccref_unit_ipa_build(const ccref_nodes *nd, const ccref_pod *pod, const int32_t *placed, int k, int64_t *aff, int64_t *anti, int64_t *exist,
                         int64_t *score, int64_t *totals) {
    if (!pod->has_ipa || k < 0 || k >= pod->ipa.n_keys) return -1;
    ipa_state s;
    memset(&s, 0, sizeof s);
    for (int j = 0; j < pod->ipa.n_keys; j++) {
        size_t len = sizeof(int64_t) * (size_t)(pod->ipa.key_ndom[j] + 1);
        s.aff[j] = (int64_t *)malloc(len), s.anti[j] = (int64_t *)malloc(len), s.exist[j] = (int64_t *)malloc(len), s.score[j] = (int64_t *)malloc(len);
    }
    ipa_build(nd, pod, placed, &s);
    size_t len = sizeof(int64_t) * (size_t)(pod->ipa.key_ndom[k] + 1);
    memcpy(aff, s.aff[k], len), memcpy(anti, s.anti[k], len), memcpy(exist, s.exist[k], len), memcpy(score, s.score[k], len);
    totals[0] = s.aff_total, totals[1] = s.exist_total, totals[2] = s.entries;
    for (int j = 0; j < CCREF_MAX_IPA_KEYS; j++) free(s.aff[j]), free(s.anti[j]), free(s.exist[j]), free(s.score[j]);
    return 0;
}

/* InterPodAffinity PreScore + Score + NormalizeScore (scoring.go:128-290) over the feasible list: raw and normalized scores, and whether
 * PreScore skips the plugin (no term hit anywhere: *skipped = 1, the scores are then all 0) */
int ccref_unit_ipa_scores(const ccref_nodes *nd, const ccref_pod *pod, const int32_t *placed, const int64_t *feas, int64_t nf, int64_t *raw,
                          int64_t *norm, int32_t *skipped) {
    if (!pod->has_ipa || nf < 0) return -1;
    ipa_state s;
    memset(&s, 0, sizeof s);
    for (int j = 0; j < pod->ipa.n_keys; j++) {
        size_t len = sizeof(int64_t) * (size_t)(pod->ipa.key_ndom[j] + 1);
        s.aff[j] = (int64_t *)malloc(len), s.anti[j] = (int64_t *)malloc(len), s.exist[j] = (int64_t *)malloc(len), s.score[j] = (int64_t *)malloc(len);
    }
    ipa_build(nd, pod, placed, &s);
    *skipped = s.entries > 0 ? 0 : 1;
    memset(raw, 0, sizeof(int64_t) * (size_t)nf), memset(norm, 0, sizeof(int64_t) * (size_t)nf);
    if (s.entries > 0) {
        ipa_raw_scores(nd, pod, &s, feas, nf, raw);
        memcpy(norm, raw, sizeof(int64_t) * (size_t)nf);
        ccref_ipa_normalize(norm, nf);
    }
    for (int j = 0; j < CCREF_MAX_IPA_KEYS; j++) free(s.aff[j]), free(s.anti[j]), free(s.exist[j]), free(s.score[j]);
    return 0;
}

int64_t ccref_schedule_one(const ccref_profile *prof, ccref_nodes *nodes, const ccref_pod *pod, ccref_sched_state *st,
                           ccref_result *res) {
    if (nodes->n == 0) return -1;
    workspace ws;
    int32_t *placed = res && res->per_node_count ? res->per_node_count : (int32_t *)calloc((size_t)nodes->n, 4);
    if (ws_init(&ws, nodes, pod, placed, 1)) return -2;
    int64_t w = schedule_one_ws(prof, nodes, pod, st, res, &ws);
    ws_free(&ws);
    if (!(res && res->per_node_count)) free(placed);
    return w;
}

/* pkg/framework/simulator.go:297-381: createNextPod / postBindHook / Update */
int ccref_run(const ccref_profile *prof, ccref_nodes *nodes, const ccref_pod *pod, int64_t max_limit, int threads,
              ccref_result *res) {
    res->placed = 0;
    res->rounds = 0;
    res->evaluated_total = 0;
    res->last_evaluated = res->last_feasible = 0;
    memset(res->hist, 0, sizeof(res->hist));
    res->n_code_unschedulable = 0;
    if (nodes->n == 0) { /* schedule_one.go:438-440 ErrNoNodesAvailable */
        res->stop = CCREF_STOP_NO_NODES;
        return 0;
    }
    memset(res->per_node_count, 0, sizeof(int32_t) * (size_t)nodes->n);
    workspace ws;
    if (ws_init(&ws, nodes, pod, res->per_node_count, threads)) return -1;
    ccref_sched_state st = {0};
    for (;;) {
        res->rounds++;
        int64_t w = schedule_one_ws(prof, nodes, pod, &st, res, &ws);
        if (w < 0) {
            res->stop = CCREF_STOP_UNSCHEDULABLE; /* simulator.go:327-342 */
            break;
        }
        if (res->log && res->placed < res->log_cap) res->log[res->placed] = (int32_t)w;
        res->placed++;
        /* simulator.go:298-305: append, then test simulated >= maxSimulated */
        if (max_limit > 0 && res->placed >= max_limit) {
            res->stop = CCREF_STOP_LIMIT;
            break;
        }
    }
    ws_free(&ws);
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * Several pod specs against one snapshot (BASELINE.json configs[4]: "100k nodes x 1024 genpod pod specs").
 * The reference simulates ONE template (pkg/framework/simulator.go:107); this is the same loop
 * (simulator.go:297-381: createNextPod -> schedule -> postBindHook) with the next pod taken from P
 * specs ROUND-ROBIN: placement i is a clone of spec i mod P, every cycle is the reference's
 * schedulePod for that pod against everything placed so far, and the run ends like the reference's:
 * at the first pod the scheduler reports Unschedulable, or at max_limit placements.
 *
 * What a clone contributes to the plugin state of LATER cycles is per spec, exactly as in the
 * single-spec loop: placed[p][n] = clones of spec p on node n feeds spec p's own selectors
 * (self_match / anti_self / self_aff of ITS pod struct).  Callers must not hand over specs whose
 * selectors match the clones of another spec (the harness generates disjoint labels; config 5's
 * pods select their own label only).
 * ------------------------------------------------------------------------------------------ */
int ccref_run_multi(const ccref_profile *prof, ccref_nodes *nodes, const ccref_pod *pods, int32_t n_pods, int64_t max_limit,
                    int threads, ccref_multi_result *res) {
    res->placed = 0;
    res->rounds = 0;
    res->stop_spec = -1;
    memset(res->hist, 0, sizeof(res->hist));
    res->n_code_unschedulable = 0;
    if (n_pods <= 0) return -1;
    if (nodes->n == 0) {
        res->stop = CCREF_STOP_NO_NODES;
        return 0;
    }
    const size_t N = (size_t)nodes->n;
    int32_t *placed = (int32_t *)calloc(N * (size_t)n_pods, sizeof(int32_t)); /* clones of spec p on node n */
    if (!placed) return -1;
    if (res->per_node_count) memset(res->per_node_count, 0, sizeof(int32_t) * N);
    if (res->per_spec_count) memset(res->per_spec_count, 0, sizeof(int32_t) * (size_t)n_pods);
    ccref_sched_state st = {0};
    ccref_result one;
    int rc = 0;
    for (;;) {
        const int32_t p = (int32_t)(res->placed % n_pods);
        workspace ws;
        if (ws_init(&ws, nodes, &pods[p], placed + (size_t)p * N, threads)) {
            rc = -1;
            break;
        }
        memset(&one, 0, sizeof one);
        one.hist_taintset = res->hist_taintset;
        res->rounds++;
        const int64_t w = schedule_one_ws(prof, nodes, &pods[p], &st, &one, &ws);
        ws_free(&ws);
        if (w < 0) {
            res->stop = CCREF_STOP_UNSCHEDULABLE;
            res->stop_spec = p;
            memcpy(res->hist, one.hist, sizeof(res->hist));
            res->n_code_unschedulable = one.n_code_unschedulable;
            break;
        }
        if (res->log && res->placed < res->log_cap) res->log[res->placed] = (int32_t)w;
        if (res->per_node_count) res->per_node_count[w] += 1;
        if (res->per_spec_count) res->per_spec_count[p] += 1;
        res->placed++;
        if (max_limit > 0 && res->placed >= max_limit) {
            res->stop = CCREF_STOP_LIMIT;
            break;
        }
    }
    free(placed);
    return rc;
}


/* ------------------------------------------------------------------------------------------
 * DefaultPreemption dry run of the terminal cycle (see ccref.h).  Preempt: preemption.go:234-303; findCandidates :306-331
 * (potential nodes = status code Unschedulable, NodesForStatusCode); DryRunPreemption :741-794; SelectVictimsOnNode:
 * default_preemption.go:217-310 (remove every lower-priority pod :245-252, none -> :255-257, Filter again :265-267).
 * ------------------------------------------------------------------------------------------ */
static void add_reasons(const fail_info *fi, int64_t *hist) {
    switch (fi->plugin) {
    case CCREF_F_NODEPORTS: hist[CCREF_R_NODEPORTS]++; break;
    case CCREF_F_VOLUMES: hist[CCREF_R_VOL0 + fi->vol_code - 1]++; break;
    case CCREF_F_FIT:
        if (fi->fit_mask & 1u) hist[CCREF_R_TOO_MANY_PODS]++;
        for (int c = 0; c < CCREF_MAX_RES; c++)
            if (fi->fit_mask & (1u << (1 + c))) hist[CCREF_R_RES0 + c]++;
        break;
    case CCREF_F_TOPOLOGYSPREAD: hist[fi->pts_code == 1 ? CCREF_R_PTS_MISSING_LABEL : CCREF_R_PTS_SKEW]++; break;
    case CCREF_F_INTERPODAFFINITY:
        hist[fi->ipa_code == 1 ? CCREF_R_IPA_AFFINITY : fi->ipa_code == 2 ? CCREF_R_IPA_ANTI : CCREF_R_IPA_EXISTING_ANTI]++;
        break;
    default: break; /* the plugins before NodePorts do not look at the node's pods: they passed before, they pass again */
    }
}

int ccref_preemption_dry_run(const ccref_profile *prof, const ccref_nodes *nodes, const ccref_pod *pod, const int32_t *placed,
                             const ccref_victims *victims, ccref_preemption *out) {
    const int64_t N = nodes->n;
    const int ncol = 3 + nodes->n_scalar;
    memset(out, 0, sizeof *out);

    /* the terminal NodeInfo: Requested and len(Pods) after the clones (types.go:409-428) */
    ccref_nodes t = *nodes;
    int64_t *cols[CCREF_MAX_RES] = {0};
    for (int c = 0; c < ncol; c++) {
        cols[c] = (int64_t *)malloc(sizeof(int64_t) * (size_t)(N > 0 ? N : 1));
        for (int64_t n = 0; n < N; n++) cols[c][n] = (nodes->req[c] ? nodes->req[c][n] : 0) + (int64_t)placed[n] * pod->req[c];
        t.req[c] = cols[c];
    }
    int32_t *pc = (int32_t *)malloc(sizeof(int32_t) * (size_t)(N > 0 ? N : 1));
    for (int64_t n = 0; n < N; n++) pc[n] = nodes->pod_count[n] + placed[n];
    t.pod_count = pc;
    ccref_pod without = *pod; /* the node's used ports once the victims are gone */
    without.host_ports_conflict = victims ? victims->ports_conflict_rest : NULL;
    without.volume_veto = victims ? victims->volume_veto_rest : NULL; /* (the clones' own disks stay: volume_exclusive + placed) */

    /* the cycle's PreFilter state of the topology-coupled plugins, as schedule_one_ws builds it (the second Filter run of the dry
     * run reads this state; removing a victim that takes no part in it leaves it as it is) */
    workspace ws;
    int32_t *placed_rw = (int32_t *)malloc(sizeof(int32_t) * (size_t)(N > 0 ? N : 1));
    memcpy(placed_rw, placed, sizeof(int32_t) * (size_t)N);
    if (ws_init(&ws, &t, pod, placed_rw, 1)) return -12;
    const pts_state *pts = NULL;
    if ((prof->filter_mask & CCREF_F_TOPOLOGYSPREAD) && has_hard_spread(pod)) {
        pts_prefilter(&t, pod, placed_rw, &ws.pts);
        pts = &ws.pts;
    }
    const ipa_state *ipa = NULL;
    if (pod->has_ipa) {
        ipa_build(&t, pod, placed_rw, &ws.ipa);
        ipa = &ws.ipa;
    }
    const int coupled = pts != NULL || (ipa != NULL && (prof->filter_mask & CCREF_F_INTERPODAFFINITY) && ipa->filter_active);

    int rc = 0;
    for (int64_t n = 0; n < N; n++) {
        fail_info fi = {0, 0, 0, 0};
        const int code = filter_node(prof, &t, pod, pts, ipa, placed, n, &fi);
        if (code != -1) { /* (0 cannot happen at a terminal cycle; counted as not helpful if the caller asks anyway) */
            out->not_helpful++;
            continue;
        }
        const int32_t vc = victims && victims->victim_count ? victims->victim_count[n] : 0;
        if (vc == 0) {
            out->no_victims++;
            continue;
        }
        if (coupled && victims->victim_interacts && victims->victim_interacts[n]) {
            rc = -38; /* RunPreFilterExtensionRemovePod would change the coupled state: not restated */
            break;
        }
        for (int c = 0; c < ncol; c++)
            if (victims->victim_req[c]) cols[c][n] -= victims->victim_req[c][n];
        pc[n] -= vc;
        fail_info fj = {0, 0, 0, 0};
        if (filter_node(prof, &t, &without, pts, ipa, placed, n, &fj) == 0) out->nominated = 1;
        else add_reasons(&fj, out->hist);
        for (int c = 0; c < ncol; c++)
            if (victims->victim_req[c]) cols[c][n] += victims->victim_req[c][n];
        pc[n] += vc;
    }
    ws_free(&ws);
    free(placed_rw);
    for (int c = 0; c < ncol; c++) free(cols[c]);
    free(pc);
    return rc;
}
