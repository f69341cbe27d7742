/* ccsim_demo.c -- the C ABI used from plain C, no Python: the README demo of kubernetes-sigs/cluster-capacity
 * (README.md:44-66: 4 nodes x (2 CPU, 4 GB, 110 pods), examples/pod.yaml = 150m / 100Mi) -> 52 = 13 x 4,
 * "Insufficient cpu" on all four nodes.  Build + run (on a box with an MI355X):
 *     gcc -std=c11 -Iinclude examples/ccsim_demo.c -Lcluster-capacity_amd/csrc -lccsim \
 *         -Wl,-rpath,$PWD/cluster-capacity_amd/csrc -Wl,-rpath,/opt/rocm/lib -o ccsim_demo && ./ccsim_demo
 * This is what a cgo shim does, minus the Go. */
#include <stdio.h>
#include <string.h>

#include "ccsim.h"

#define N 4

int main(void) {
    int64_t alloc_cpu[N], alloc_mem[N], alloc_eph[N], zero64[N];
    int32_t alloc_pods[N], pod_count[N], taintset[N];
    uint8_t unsched[N];
    for (int i = 0; i < N; i++) {
        alloc_cpu[i] = 2000, alloc_mem[i] = 4000000000LL, alloc_eph[i] = 0, zero64[i] = 0;
        alloc_pods[i] = 110, pod_count[i] = 0, taintset[i] = 0, unsched[i] = 0;
    }
    ccsim_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.abi_version = CCSIM_ABI_VERSION, cfg.device = 0, cfg.use_graph = 1;
    ccsim_engine *e = NULL;
    int rc = ccsim_create(&cfg, &e);
    if (rc) { fprintf(stderr, "ccsim_create: %d (no HIP device?)\n", rc); return 2; }

    ccsim_nodes nd;
    memset(&nd, 0, sizeof nd);
    nd.n_nodes = N, nd.n_global = N;
    nd.alloc[0] = alloc_cpu, nd.alloc[1] = alloc_mem, nd.alloc[2] = alloc_eph;
    nd.req[0] = zero64, nd.req[1] = zero64, nd.req[2] = zero64;
    nd.nz_mcpu = zero64, nd.nz_mem = zero64;
    nd.alloc_pods = alloc_pods, nd.pod_count = pod_count, nd.taintset_id = taintset, nd.unschedulable = unsched;
    if ((rc = ccsim_load_nodes(e, &nd))) { fprintf(stderr, "load_nodes: %s\n", ccsim_last_error(e)); return 1; }

    ccsim_profile pf; /* the default profile: S/apis/config/v1/default_plugins.go:30-58 */
    memset(&pf, 0, sizeof pf);
    pf.filter_mask = CCSIM_F_UNSCHEDULABLE | CCSIM_F_NODENAME | CCSIM_F_TAINT | CCSIM_F_NODEAFFINITY | CCSIM_F_FIT |
                     CCSIM_F_TOPOLOGYSPREAD | CCSIM_F_INTERPODAFFINITY;
    pf.w_taint = 3, pf.w_nodeaffinity = 2, pf.w_fit = 1, pf.w_balanced = 1, pf.w_topologyspread = 2, pf.w_interpodaffinity = 2;
    pf.n_fit_res = 2, pf.fit_res[0] = 0, pf.fit_res[1] = 1, pf.fit_res_w[0] = 1, pf.fit_res_w[1] = 1;
    pf.n_bal_res = 2, pf.bal_res[0] = 0, pf.bal_res[1] = 1;
    pf.percentage_of_nodes_to_score = 100;
    if ((rc = ccsim_set_profile(e, &pf))) { fprintf(stderr, "set_profile: %s\n", ccsim_last_error(e)); return 1; }

    const uint8_t taint_ok[1] = {1};
    const int32_t taint_cnt[1] = {0};
    ccsim_pod pod;
    memset(&pod, 0, sizeof pod);
    pod.req[0] = 150, pod.req[1] = 100LL << 20, pod.nz_mcpu = 150, pod.nz_mem = 100LL << 20;
    pod.n_taintsets = 1, pod.taint_filter_ok = taint_ok, pod.taint_prefer_cnt = taint_cnt;
    if ((rc = ccsim_set_pod(e, &pod))) { fprintf(stderr, "set_pod: %s\n", ccsim_last_error(e)); return 1; }

    int32_t per_node[N], log[64];
    ccsim_report rep;
    memset(&rep, 0, sizeof rep);
    rep.per_node_count = per_node, rep.per_node_cap = N, rep.log = log, rep.log_cap = 64;
    if ((rc = ccsim_run(e, 0, CCSIM_MODE_BATCHED, &rep))) { fprintf(stderr, "run: %s\n", ccsim_last_error(e)); return 1; }
    printf("The cluster can schedule %lld instance(s) of the pod small-pod.\n", (long long)rep.placed);
    printf("Termination reason: %s: 0/%d nodes are available: %lld Insufficient cpu.\n",
           rep.stop == CCSIM_STOP_UNSCHEDULABLE ? "Unschedulable" : "LimitReached", N, (long long)rep.hist[CCSIM_R_RES0 + 0]);
    for (int i = 0; i < N; i++) printf("\t- kube-node-%d: %d instance(s)\n", i + 1, per_node[i]);
    /* ABI 5: the same run once more without the placement log, offering ONE-BYTE per-node counts next to the int32 array (no node
     * holds more clones than its pod capacity, 110 here).  The engine fills one of the two and says which in per_node_filled_width. */
    uint8_t per_node8[N];
    ccsim_report rep2;
    memset(&rep2, 0, sizeof rep2);
    memset(per_node, 0, sizeof per_node), memset(per_node8, 0, sizeof per_node8);
    rep2.per_node_count = per_node, rep2.per_node_cap = N, rep2.per_node_count_narrow = per_node8, rep2.per_node_narrow_width = 1;
    if ((rc = ccsim_reset_state(e)) || (rc = ccsim_run(e, 0, CCSIM_MODE_BATCHED, &rep2))) { fprintf(stderr, "run 2: %s\n", ccsim_last_error(e)); return 1; }
    int narrow_ok = rep2.placed == 52 && (rep2.per_node_filled_width == 1 || rep2.per_node_filled_width == 4);
    for (int i = 0; i < N; i++) narrow_ok = narrow_ok && (rep2.per_node_filled_width == 1 ? per_node8[i] : per_node[i]) == 13;
    printf("per-node counts of the second run came back in %d-byte elements\n", rep2.per_node_filled_width);
    ccsim_destroy(e);
    return rep.placed == 52 && rep.hist[CCSIM_R_RES0] == 4 && narrow_ok ? 0 : 1;
}
