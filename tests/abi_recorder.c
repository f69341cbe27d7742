/* abi_recorder.c -- TEST INFRASTRUCTURE.  A stand-in for libccsim.so that RECORDS what a host passes through the C ABI
 * (include/ccsim.h) and schedules nothing: ccsim_run returns a canned, obviously artificial result (one pod on every
 * node, LimitReached).  tests/test_native_host.py points the native host at it (CCSIM_LIB) to check, without a GPU, that
 * the structs the C++ host marshals equal the ones the Python binding marshals from the same snapshot.
 * The recording goes to the file named by $CCSIM_RECORD as JSON; with $CCSIM_RECORD_PER_DEVICE set, engine d writes
 * "$CCSIM_RECORD.d" (the sharded host path creates one engine per device, concurrently).
 * The ccsim_dist_* driver entry points are recorded the same way: ccsim_dist_run returns a canned sharded result (one pod on
 * every node of the shard, global placement i = global node i, logged by the rank that owns the node, -1 elsewhere). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/ccsim.h"

struct ccsim_engine {
    FILE *f;
    int64_t n;
    int n_taintsets;
    int first;
    int64_t global_offset, n_global;
    int world, rank;
    int64_t one_cycle_runs; /* ccsim_dist_run calls with max_limit == 1 so far (the hosts' one-cycle-at-a-time loop) */
};

static void arr64(FILE *f, const char *k, const int64_t *p, int64_t n) {
    fprintf(f, "\"%s\": ", k);
    if (!p) { fprintf(f, "null"); return; }
    fprintf(f, "[");
    for (int64_t i = 0; i < n; i++) fprintf(f, "%s%lld", i ? ", " : "", (long long)p[i]);
    fprintf(f, "]");
}
static void arr32(FILE *f, const char *k, const int32_t *p, int64_t n) {
    fprintf(f, "\"%s\": ", k);
    if (!p) { fprintf(f, "null"); return; }
    fprintf(f, "[");
    for (int64_t i = 0; i < n; i++) fprintf(f, "%s%d", i ? ", " : "", p[i]);
    fprintf(f, "]");
}
static void arr8(FILE *f, const char *k, const uint8_t *p, int64_t n) {
    fprintf(f, "\"%s\": ", k);
    if (!p) { fprintf(f, "null"); return; }
    fprintf(f, "[");
    for (int64_t i = 0; i < n; i++) fprintf(f, "%s%d", i ? ", " : "", (int)p[i]);
    fprintf(f, "]");
}
static void sep(ccsim_engine *e) { fprintf(e->f, e->first ? "" : ",\n"), e->first = 0; }

int32_t ccsim_abi_version(void) { return CCSIM_ABI_VERSION; }
const char *ccsim_last_error(const ccsim_engine *e) { (void)e; return "abi recorder"; }

int ccsim_create(const ccsim_config *cfg, ccsim_engine **out) {
    const char *path = getenv("CCSIM_RECORD");
    if (!path || !cfg || cfg->abi_version != CCSIM_ABI_VERSION) return -22;
    ccsim_engine *e = (ccsim_engine *)calloc(1, sizeof *e);
    char full[4096];
    if (getenv("CCSIM_RECORD_PER_DEVICE")) snprintf(full, sizeof full, "%s.%d", path, cfg->device);
    else snprintf(full, sizeof full, "%s", path);
    e->f = fopen(full, "w");
    if (!e->f) return -5;
    e->first = 1;
    fprintf(e->f, "{\n");
    sep(e);
    fprintf(e->f, "\"config\": {\"device\": %d, \"use_graph\": %d, \"time_passes\": %d, \"stream_is_null\": %d}", cfg->device, cfg->use_graph,
            cfg->time_passes, cfg->stream == NULL);
    *out = e;
    return 0;
}

void ccsim_destroy(ccsim_engine *e) {
    if (!e) return;
    fprintf(e->f, "\n}\n");
    fclose(e->f);
    free(e);
}

int ccsim_load_nodes(ccsim_engine *e, const ccsim_nodes *n) {
    FILE *f = e->f;
    e->n = n->n_nodes, e->global_offset = n->global_offset, e->n_global = n->n_global;
    sep(e);
    fprintf(f, "\"nodes\": {\"n_nodes\": %lld, \"global_offset\": %lld, \"n_global\": %lld, \"n_scalar\": %d, \"n_label_cols\": %d, ", (long long)n->n_nodes,
            (long long)n->global_offset, (long long)n->n_global, n->n_scalar, n->n_label_cols);
    fprintf(f, "\"alloc\": [");
    for (int c = 0; c < 3 + n->n_scalar; c++) { fprintf(f, "%s{", c ? ", " : ""); arr64(f, "v", n->alloc[c], n->n_nodes); fprintf(f, "}"); }
    fprintf(f, "], \"req\": [");
    for (int c = 0; c < 3 + n->n_scalar; c++) { fprintf(f, "%s{", c ? ", " : ""); arr64(f, "v", n->req[c], n->n_nodes); fprintf(f, "}"); }
    fprintf(f, "], \"label_cols\": [");
    for (int c = 0; c < n->n_label_cols; c++) { fprintf(f, "%s{", c ? ", " : ""); arr32(f, "v", n->label_cols[c], n->n_nodes); fprintf(f, "}"); }
    fprintf(f, "], ");
    arr32(f, "alloc_pods", n->alloc_pods, n->n_nodes), fprintf(f, ", ");
    arr64(f, "nz_mcpu", n->nz_mcpu, n->n_nodes), fprintf(f, ", ");
    arr64(f, "nz_mem", n->nz_mem, n->n_nodes), fprintf(f, ", ");
    arr32(f, "pod_count", n->pod_count, n->n_nodes), fprintf(f, ", ");
    arr32(f, "taintset_id", n->taintset_id, n->n_nodes), fprintf(f, ", ");
    arr8(f, "unschedulable", n->unschedulable, n->n_nodes);
    fprintf(f, "}");
    return 0;
}

int ccsim_set_profile(ccsim_engine *e, const ccsim_profile *p) {
    FILE *f = e->f;
    sep(e);
    fprintf(f, "\"profile\": {\"filter_mask\": %u, \"w\": [%d, %d, %d, %d, %d, %d, %d], \"pct\": %d, ", p->filter_mask, p->w_taint, p->w_nodeaffinity, p->w_fit,
            p->w_balanced, p->w_topologyspread, p->w_interpodaffinity, p->w_imagelocality, p->percentage_of_nodes_to_score);
    arr32(f, "fit_res", p->fit_res, p->n_fit_res), fprintf(f, ", ");
    arr64(f, "fit_res_w", p->fit_res_w, p->n_fit_res), fprintf(f, ", ");
    arr32(f, "bal_res", p->bal_res, p->n_bal_res);
    fprintf(f, "}");
    return 0;
}

static void term(FILE *f, const ccsim_pod *p, const ccsim_term *t) { /* resolved: [weight, [[col, table...], ...]] */
    fprintf(f, "[%d, [", t->weight);
    for (int i = 0; i < t->n_req; i++) {
        const ccsim_requirement *r = &p->reqs[t->first_req + i];
        /* the table's length is not part of the ABI: it ends where the next requirement's begins (or at req_tables_len) */
        int64_t end = p->req_tables_len;
        for (int j = 0; j < p->n_reqs; j++)
            if (p->reqs[j].table_off > r->table_off && p->reqs[j].table_off < end) end = p->reqs[j].table_off;
        fprintf(f, "%s[%d", i ? ", " : "", r->col);
        for (int64_t k = r->table_off; k < end; k++) fprintf(f, ", %d", (int)p->req_tables[k]);
        fprintf(f, "]");
    }
    fprintf(f, "]]");
}

static void record_pod(ccsim_engine *e, const ccsim_pod *p);

int ccsim_set_pod(ccsim_engine *e, const ccsim_pod *p) {
    e->n_taintsets = p->n_taintsets;
    sep(e);
    fprintf(e->f, "\"pod\": ");
    record_pod(e, p);
    return 0;
}

/* several templates, cycled round-robin by ccsim_run */
int ccsim_set_pods(ccsim_engine *e, const ccsim_pod *pods, int32_t n_pods) {
    e->n_taintsets = pods[0].n_taintsets;
    sep(e);
    fprintf(e->f, "\"pods\": [");
    for (int32_t i = 0; i < n_pods; i++) {
        if (i) fprintf(e->f, ", ");
        record_pod(e, &pods[i]);
    }
    fprintf(e->f, "]");
    return 0;
}

static void record_pod(ccsim_engine *e, const ccsim_pod *p) {
    FILE *f = e->f;
    const int64_t N = e->n;
    fprintf(f, "{");
    arr64(f, "req", p->req, CCSIM_MAX_RES);
    fprintf(f, ", \"has_scalar_entries\": %d, \"nz_mcpu\": %lld, \"nz_mem\": %lld, \"tolerates_unschedulable\": %d, \"affinity_filter_active\": %d, "
               "\"has_node_selector\": %d, \"has_required_terms\": %d, \"n_reqs\": %d, ",
            p->has_scalar_entries, (long long)p->nz_mcpu, (long long)p->nz_mem, p->tolerates_unschedulable, p->affinity_filter_active, p->has_node_selector,
            p->has_required_terms, p->n_reqs);
    arr8(f, "taint_filter_ok", p->taint_filter_ok, p->n_taintsets), fprintf(f, ", ");
    arr32(f, "taint_prefer_cnt", p->taint_prefer_cnt, p->n_taintsets);
    fprintf(f, ", \"node_selector\": ");
    term(f, p, &p->node_selector);
    fprintf(f, ", \"required\": [");
    for (int i = 0; i < p->n_required; i++) fprintf(f, "%s", i ? ", " : ""), term(f, p, &p->required[i]);
    fprintf(f, "], \"preferred\": [");
    for (int i = 0; i < p->n_preferred; i++) fprintf(f, "%s", i ? ", " : ""), term(f, p, &p->preferred[i]);
    fprintf(f, "], \"spread\": [");
    for (int i = 0; i < p->n_spread; i++) {
        const ccsim_spread_constraint *c = &p->spread[i];
        fprintf(f, "%s{\"k\": [%d, %d, %d, %d, %d, %d, %d, %d], ", i ? ", " : "", c->col, c->max_skew, c->min_domains, c->hard, c->self_match, c->n_domains, c->is_hostname, c->missing_value);
        arr32(f, "node_match_count", c->node_match_count, N), fprintf(f, ", ");
        arr8(f, "node_included", c->node_included, N);
        fprintf(f, "}");
    }
    fprintf(f, "], \"has_ipa\": %d", p->has_ipa);
    if (p->has_ipa) {
        const ccsim_ipa *a = &p->ipa;
        fprintf(f, ", \"ipa\": {\"self_aff\": %d, \"entries_existing\": %lld, ", a->self_aff, (long long)a->entries_existing);
        arr32(f, "key_col", a->key_col, a->n_keys), fprintf(f, ", ");
        arr32(f, "key_ndom", a->key_ndom, a->n_keys), fprintf(f, ", ");
        arr32(f, "aff_key", a->aff_key, a->n_aff_terms), fprintf(f, ", ");
        arr32(f, "anti_key", a->anti_key, a->n_anti_terms), fprintf(f, ", ");
        arr32(f, "anti_self", a->anti_self, a->n_anti_terms), fprintf(f, ", ");
        arr64(f, "score_self", a->score_self, a->n_keys), fprintf(f, ", ");
        arr32(f, "self_entries", a->self_entries, a->n_keys), fprintf(f, ", ");
        arr32(f, "aff_existing", a->aff_existing, N);
        fprintf(f, ", \"anti_existing\": [");
        for (int t = 0; t < a->n_anti_terms; t++) { fprintf(f, "%s{", t ? ", " : ""); arr32(f, "v", a->anti_existing[t], N); fprintf(f, "}"); }
        fprintf(f, "], \"exist_anti\": [");
        for (int k = 0; k < a->n_keys; k++) { fprintf(f, "%s{", k ? ", " : ""); arr32(f, "v", a->exist_anti[k], N); fprintf(f, "}"); }
        fprintf(f, "], \"score_existing\": [");
        for (int k = 0; k < a->n_keys; k++) { fprintf(f, "%s{", k ? ", " : ""); arr64(f, "v", a->score_existing[k], N); fprintf(f, "}"); }
        fprintf(f, "]}");
    }
    fprintf(f, ", \"has_host_ports\": %d, ", p->has_host_ports);
    arr8(f, "host_ports_conflict", p->host_ports_conflict, N), fprintf(f, ", ");
    arr8(f, "image_score", p->image_score, N);
    fprintf(f, ", \"volume_exclusive\": %d, ", p->volume_exclusive);
    arr8(f, "volume_veto", p->volume_veto, N);
    fprintf(f, "}");
}

int ccsim_run(ccsim_engine *e, int64_t max_limit, int32_t mode, ccsim_report *out) {
    sep(e);
    fprintf(e->f, "\"run\": {\"max_limit\": %lld, \"mode\": %d, \"per_node_cap\": %lld, \"log_cap\": %lld, \"hist_taintset_cap\": %d}", (long long)max_limit, mode,
            (long long)out->per_node_cap, (long long)out->log_cap, out->hist_taintset_cap);
    if (out->per_node_cap < e->n || (out->hist_taintset && out->hist_taintset_cap < e->n_taintsets)) return -22;
    /* the canned result: one pod on every node, in index order */
    out->placed = e->n, out->stop = CCSIM_STOP_LIMIT, out->log_len = 0;
    for (int64_t i = 0; i < e->n; i++) {
        out->per_node_count[i] = 1;
        if (out->log && i < out->log_cap) out->log[i] = (int32_t)i, out->log_len = i + 1;
    }
    memset(out->hist, 0, sizeof out->hist);
    out->n_code_unschedulable = 0;
    return 0;
}

/* ---- the library-driven multi-GPU entry points ---- */
int ccsim_dist_unique_id(uint8_t *id_out) {
    for (int i = 0; i < CCSIM_DIST_ID_BYTES; i++) id_out[i] = (uint8_t)(i * 7 + 3);
    return 0;
}

int ccsim_dist_comm_init(ccsim_engine *e, const uint8_t *id, int32_t n_ranks, int32_t rank) {
    int id_ok = 1;
    for (int i = 0; i < CCSIM_DIST_ID_BYTES; i++) id_ok &= id[i] == (uint8_t)(i * 7 + 3);
    e->world = n_ranks, e->rank = rank;
    sep(e);
    fprintf(e->f, "\"dist_comm_init\": {\"n_ranks\": %d, \"rank\": %d, \"id_ok\": %d}", n_ranks, rank, id_ok);
    return id_ok ? 0 : -22;
}

int ccsim_dist_comm_size(ccsim_engine *e, int32_t *n_ranks_out, int32_t *rank_out) {
    if (e->world < 1) return -22;
    *n_ranks_out = e->world;
    if (rank_out) *rank_out = e->rank;
    return 0;
}

int ccsim_dist_cw_eligible(ccsim_engine *e) { (void)e; return 0; }
int ccsim_dist_cw_enable(ccsim_engine *e, int32_t all_ok) { (void)e; return all_ok ? -38 : 0; }
int ccsim_dist_cw_buffers(ccsim_engine *e, void **s, void **r, int64_t *b) { (void)e, (void)s, (void)r, (void)b; return -38; }
int ccsim_dist_cw_scan(ccsim_engine *e) { (void)e; return -38; }
int ccsim_dist_cw_decide(ccsim_engine *e) { (void)e; return -38; }

int ccsim_dist_sync_tables(ccsim_engine *e) {
    sep(e);
    fprintf(e->f, "\"dist_sync_tables\": %d", e->world);
    return e->world > 0 ? 0 : -22;
}

int ccsim_dist_run(ccsim_engine *e, int64_t max_limit, int32_t mode, ccsim_report *out) {
    sep(e);
    fprintf(e->f, "\"dist_run\": {\"max_limit\": %lld, \"mode\": %d, \"per_node_cap\": %lld, \"log_cap\": %lld, \"hist_taintset_cap\": %d}", (long long)max_limit, mode,
            (long long)out->per_node_cap, (long long)out->log_cap, out->hist_taintset_cap);
    if (e->world <= 0 || out->per_node_cap < e->n || (out->hist_taintset && out->hist_taintset_cap < e->n_taintsets)) return -22;
    if (max_limit == 1) { /* the one-cycle-at-a-time loop: cycle c lands on global node c (logged by its owner), cycle n_global finds nothing */
        const int64_t c = e->one_cycle_runs++;
        memset(out->hist, 0, sizeof out->hist);
        for (int64_t i = 0; i < e->n; i++) out->per_node_count[i] = 0;
        out->log_len = 0, out->n_code_unschedulable = 0;
        if (c < e->n_global) {
            const int mine = c >= e->global_offset && c < e->global_offset + e->n;
            out->placed = 1, out->stop = CCSIM_STOP_LIMIT;
            if (mine) out->per_node_count[c - e->global_offset] = 1;
            if (out->log && out->log_cap > 0) out->log[0] = mine ? (int32_t)c : -1, out->log_len = 1;
        } else {
            out->placed = 0, out->stop = CCSIM_STOP_UNSCHEDULABLE;
            out->hist[CCSIM_R_TOO_MANY_PODS] = e->n;
            if (out->hist_taintset)
                for (int i = 0; i < e->n_taintsets; i++) out->hist_taintset[i] = 0;
            out->n_code_unschedulable = e->n;
        }
        return 0;
    }
    out->placed = e->n_global, out->stop = CCSIM_STOP_UNSCHEDULABLE, out->log_len = 0;
    for (int64_t i = 0; i < e->n; i++) out->per_node_count[i] = 1;
    if (out->log)
        for (int64_t g = 0; g < e->n_global && g < out->log_cap; g++) {
            const int mine = g >= e->global_offset && g < e->global_offset + e->n;
            out->log[g] = mine ? (int32_t)g : -1, out->log_len = g + 1;
        }
    /* FitError pieces of THIS shard: every node of the shard under one reason, the first taint set */
    memset(out->hist, 0, sizeof out->hist);
    out->hist[CCSIM_R_TOO_MANY_PODS] = e->n;
    if (out->hist_taintset)
        for (int i = 0; i < e->n_taintsets; i++) out->hist_taintset[i] = i == 0 ? e->rank + 1 : 0;
    out->n_code_unschedulable = e->rank == 0 ? 1 : 0;
    return 0;
}

/* ---- the rest of the ABI, so that the Python binding (capi.load binds EVERY declared symbol) can run on this library:
 * entry points this stand-in has no use for answer -ENOSYS ---- */
int ccsim_schedule_pod(ccsim_engine *e, int32_t pod_idx, ccsim_cycle *out) { (void)e, (void)pod_idx, (void)out; return -38; }
int ccsim_schedule_one(ccsim_engine *e, ccsim_cycle *out) { (void)e, (void)out; return -38; }
int ccsim_read_state(ccsim_engine *e, int64_t *a, int64_t *b, int64_t *c, int64_t *d, int32_t *f) { (void)e, (void)a, (void)b, (void)c, (void)d, (void)f; return -38; }
int ccsim_dist_begin(ccsim_engine *e, int64_t max_limit, int32_t mode, int32_t n_ranks, int32_t rank, void *s, void *r, int64_t log_cap) {
    (void)e, (void)max_limit, (void)mode, (void)n_ranks, (void)rank, (void)s, (void)r, (void)log_cap;
    return -38;
}
int ccsim_dist_scan(ccsim_engine *e) { (void)e; return -38; }
int ccsim_dist_decide(ccsim_engine *e) { (void)e; return -38; }
int ccsim_dist_poll(ccsim_engine *e, int32_t *done, int64_t *placed) { (void)e, (void)done, (void)placed; return -38; }
int ccsim_dist_finish(ccsim_engine *e, ccsim_report *out) { (void)e, (void)out; return -38; }
/* one replicated table is announced, so that hosts which ask "are there tables?" go on to ccsim_dist_sync_tables */
int ccsim_dist_table_count(ccsim_engine *e) { (void)e; return 1; }
int ccsim_dist_table(ccsim_engine *e, int32_t idx, void **ptr, int64_t *len, int32_t *elem_bytes, int32_t *op) {
    static int64_t dummy[1];
    (void)e;
    if (idx != 0) return -22;
    *ptr = dummy, *len = 1, *elem_bytes = 8, *op = 0;
    return 0;
}
int ccsim_dist_tables_done(ccsim_engine *e) { (void)e; return 0; }
int ccsim_reset_state(ccsim_engine *e) { (void)e; return 0; } /* (nothing to restore: bench.py's step frame calls it before every run) */
void *ccsim_host_alloc(ccsim_engine *e, size_t bytes) { (void)e; return calloc(1, bytes); }
void ccsim_host_free(ccsim_engine *e, void *p) { (void)e; free(p); }
int ccsim_time_scan(ccsim_engine *e, int32_t a, int32_t b, int64_t *c, int64_t *d) { (void)e, (void)a, (void)b, (void)c, (void)d; return -38; }
/* the mailbox form (ccsim_dist_mbox_*): not recorded -- "not eligible", so that a host falls back to the pass protocol */
int ccsim_dist_mbox_info(ccsim_engine *e, uint8_t *o) { (void)e, (void)o; return -38; }
int ccsim_dist_mbox_connect(ccsim_engine *e, const uint8_t *a, int32_t n, int32_t r) { (void)e, (void)a, (void)n, (void)r; return -38; }
int ccsim_dist_mbox_eligible(ccsim_engine *e) { (void)e; return 0; }
int ccsim_dist_mbox_launch(ccsim_engine *e) { (void)e; return -38; }
int ccsim_dist_mbox_status(ccsim_engine *e, int32_t *ok) { (void)e, (void)ok; return -38; }
int ccsim_dist_mbox_finish(ccsim_engine *e, int32_t all_ok) { (void)e, (void)all_ok; return -38; }
int ccsim_debug_persist_prof(ccsim_engine *e, int64_t *out) { (void)e, (void)out; return -38; }
int ccsim_debug_multi_stops(ccsim_engine *e, int64_t *out) { (void)e, (void)out; return -38; }
int ccsim_debug_multi_memo(ccsim_engine *e, int64_t *out) { (void)e, (void)out; return -38; }
int ccsim_debug_coupled(ccsim_engine *e, int64_t *out) { (void)e, (void)out; return -38; }
int ccsim_debug_sampled(ccsim_engine *e, int64_t *out) { (void)e, (void)out; return -38; }
int ccsim_debug_dist(ccsim_engine *e, int64_t *out) { /* no mailboxes here: every run is the pass protocol */
    for (int i = 0; i < 8; i++) out[i] = 0;
    out[1] = -1, out[3] = 2, out[5] = e->world;
    return 0;
}
