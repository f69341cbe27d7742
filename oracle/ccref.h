/*
 * ccref.h -- CPU ORACLE for the cluster-capacity placement path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is a plain-C, sequential restatement of the reference's one-pod-at-a-time scheduling
 * loop (kubernetes-sigs/cluster-capacity driving the vendored kube-scheduler v1.34.1).  It is
 * NOT part of the product: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load it.  The product path (cluster-capacity_amd/csrc) never links or calls it.
 *
 * Parity status: the reference cannot be executed here (no Go toolchain) and none of its tests
 * pins an instance count, so numeric parity is pinned only by the prose known answers of the
 * reference's README (52 = 4x13, 52 = 2x26) and the FailType assertions of
 * pkg/framework/simulator_test.go -- see tests/test_oracle_known_answers.py.  "parity unpinned"
 * beyond those for the loop as a whole.  What can be anchored on the reference's SOURCES without running Go is: the unit arithmetic
 * (ccref_least_allocated, ccref_balanced_allocation, ccref_default_normalize, ccref_num_feasible_nodes_to_find,
 * ccref_image_locality_score, ccref_pts_normalize, ccref_ipa_normalize) and three filters as the loop applies them -- NodeResourcesFit's fitsRequest (reason
 * set + the Unresolvable status rule), PodTopologySpread's Filter with minMatchNum, InterPodAffinity's Filter with its satisfy* functions, each checked node by
 * node through ccref_run -- and, since round 3, the loop-level pieces: PodTopologySpread's calPreFilterState (ccref_unit_pts_prefilter) and its
 * PreScore + Score (ccref_unit_pts_scores), InterPodAffinity's count maps (ccref_unit_ipa_build) and its PreScore + Score + Skip
 * (ccref_unit_ipa_scores), the node search of a cycle from a given start index (ccref_schedule_one: nodes visited, feasible nodes kept, the next
 * start index), RunScorePlugins' weight-and-sum (ccref_weigh), selectHost (ccref_select_host), topologyNormalizingWeight
 * (ccref_go_log) -- equal the output of a mechanical line-by-line transliteration of the reference's own Go functions
 * (tests/golden/reference_vectors.json, tests/test_reference_vectors.py); every message string, status code, default and the
 * filter order equal what the sources say (tests/golden/reference_pins.json, tests/test_reference_pins.py).
 *
 * Paths below are relative to the reference root; S/ = vendor/k8s.io/kubernetes/pkg/scheduler,
 * P/ = S/framework/plugins.
 *
 * Strings never appear here: taints, tolerations, labels and selectors are interned by the
 * caller (tests/objmodel.py mirrors the string-level matching rules for that step).
 */
#ifndef CCREF_H
#define CCREF_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CCREF_MAX_SCALAR 8
#define CCREF_MAX_RES (3 + CCREF_MAX_SCALAR) /* resource "columns": 0 cpu(milli) 1 memory 2 ephemeral 3+k scalar k */
#define CCREF_MAX_LABEL_COLS 32
#define CCREF_MAX_TSC 8 /* topology spread constraints per pod */
#define CCREF_MAX_IPA_KEYS 4  /* distinct topology keys of inter-pod (anti)affinity terms */
#define CCREF_MAX_IPA_TERMS 8 /* required (anti)affinity terms of the incoming pod */

/* filter plugins, in the default profile's order (S/apis/config/v1/default_plugins.go:30-58) */
enum {
    CCREF_F_UNSCHEDULABLE = 1u << 0, /* P/nodeunschedulable */
    CCREF_F_NODENAME = 1u << 1,      /* P/nodename (always passes: podgenerator clears Spec.NodeName) */
    CCREF_F_TAINT = 1u << 2,         /* P/tainttoleration */
    CCREF_F_NODEAFFINITY = 1u << 3,  /* P/nodeaffinity */
    CCREF_F_FIT = 1u << 4,           /* P/noderesources/fit.go */
    CCREF_F_TOPOLOGYSPREAD = 1u << 5, /* P/podtopologyspread */
    CCREF_F_INTERPODAFFINITY = 1u << 6, /* P/interpodaffinity */
    CCREF_F_NODEPORTS = 1u << 7, /* P/nodeports (runs between NodeAffinity and NodeResourcesFit, default_plugins.go:34-40) */
    CCREF_F_VOLUMES = 1u << 8    /* (a tag for fail_info only, never part of a profile's filter_mask: which of the four volume plugins
                                  * run is the caller's business -- it evaluates them into ccref_pod.volume_veto / volume_exclusive) */
};

/* reason slots of the terminal-round histogram (S/framework/types.go:787-836) */
enum {
    CCREF_R_UNSCHEDULABLE = 0, /* "node(s) were unschedulable" */
    CCREF_R_NODENAME = 1,      /* "node(s) didn't match the requested node name" */
    CCREF_R_NODEAFFINITY = 2,  /* "node(s) didn't match Pod's node affinity/selector" */
    CCREF_R_TOO_MANY_PODS = 3, /* "Too many pods" */
    CCREF_R_RES0 = 4,          /* + column: "Insufficient cpu|memory|ephemeral-storage|<scalar name>" */
    CCREF_R_PTS_MISSING_LABEL = CCREF_R_RES0 + CCREF_MAX_RES, /* "node(s) didn't match pod topology spread constraints (missing required label)" */
    CCREF_R_PTS_SKEW,                                         /* "node(s) didn't match pod topology spread constraints" */
    CCREF_R_IPA_AFFINITY,      /* "node(s) didn't match pod affinity rules" (UnschedulableAndUnresolvable) */
    CCREF_R_IPA_ANTI,          /* "node(s) didn't match pod anti-affinity rules" */
    CCREF_R_IPA_EXISTING_ANTI, /* "node(s) didn't satisfy existing pods anti-affinity rules" */
    CCREF_R_NODEPORTS,         /* "node(s) didn't have free ports for the requested pod ports" (P/nodeports/node_ports.go:39) */
    /* the volume plugins in filter order (default_plugins.go:41-44): slot = CCREF_R_VOL0 + (volume_veto code - 1) */
    CCREF_R_VOL0,
    CCREF_R_VOL_DISK_CONFLICT = CCREF_R_VOL0, /* "node(s) had no available disk" (P/volumerestrictions/volume_restrictions.go:55) */
    CCREF_R_VOL_RWOP,          /* "node(s) unavailable due to PersistentVolumeClaim with ReadWriteOncePod access mode already in-use by another pod" (:59) */
    CCREF_R_VOL_MAX_COUNT,     /* "node(s) exceed max volume count" (P/nodevolumelimits/csi.go:44) */
    CCREF_R_VOL_NODE_AFFINITY, /* "node(s) didn't match PersistentVolume's node affinity" (P/volumebinding/binder.go:67, UnschedulableAndUnresolvable) */
    CCREF_R_VOL_NO_PV,         /* "node(s) didn't find available persistent volumes to bind" (binder.go:65, UnschedulableAndUnresolvable) */
    CCREF_R_VOL_PV_NOT_EXIST,  /* "node(s) unavailable due to one or more pvc(s) bound to non-existent pv(s)" (binder.go:71, UnschedulableAndUnresolvable) */
    CCREF_R_VOL_ZONE,          /* "node(s) had no available volume zone" (P/volumezone/volume_zone.go:61, UnschedulableAndUnresolvable) */
    CCREF_NREASON
};
#define CCREF_VOL_CODES 7
#define CCREF_VOL_LAST_UNSCHEDULABLE 3 /* codes 1..3 plain Unschedulable, 4..6 UnschedulableAndUnresolvable */

enum { CCREF_STOP_UNSCHEDULABLE = 0, CCREF_STOP_LIMIT = 1, CCREF_STOP_NO_NODES = 2 };

typedef struct {
    int64_t n;
    /* NodeInfo.Allocatable (S/framework/types.go:940-950) */
    const int64_t *alloc[CCREF_MAX_RES]; /* [col][n]; unused cols may be NULL */
    const int32_t *alloc_pods;
    /* NodeInfo.Requested / NonZeroRequested / len(Pods): MUTATED by placements (types.go:409-428) */
    int64_t *req[CCREF_MAX_RES];
    int64_t *nz_mcpu, *nz_mem;
    int32_t *pod_count;
    int32_t n_scalar;
    /* interned static attributes */
    const int32_t *taintset_id;   /* distinct Spec.Taints list id */
    const uint8_t *unschedulable; /* Spec.Unschedulable */
    int32_t n_label_cols;
    const int32_t *label_cols[CCREF_MAX_LABEL_COLS]; /* value id per node, 0 = label absent */
} ccref_nodes;

/* one matchExpression/matchField evaluated by table lookup on the node's value id for `col` */
typedef struct {
    int32_t col;
    int32_t table_off; /* match iff req_tables[table_off + label_cols[col][node]] != 0 */
} ccref_requirement;

typedef struct {
    int32_t first_req, n_req; /* AND over requirements; n_req==0 never matches (nodeaffinity.go:60-64) */
    int32_t weight;           /* preferred terms only */
} ccref_term;

/* one topologySpreadConstraint (P/podtopologyspread/common.go:42-56) */
typedef struct {
    int32_t col;          /* label column of the topologyKey */
    int32_t max_skew;     /* >=1 */
    int32_t min_domains;  /* >=1 (nil -> 1) */
    int32_t hard;         /* 1 = DoNotSchedule (filter), 0 = ScheduleAnyway (score) */
    int32_t self_match;   /* 1 if the pod's own labels match the constraint's selector */
    int32_t is_hostname;  /* topologyKey == kubernetes.io/hostname (scoring.go:214-215) */
    int32_t n_domains;    /* number of value ids of `col` (ids are 1..n_domains) */
    /* per-node count of EXISTING pods (same namespace, selector match) on the node, or NULL = 0;
       simulated clones add self_match per placement */
    const int32_t *node_match_count;
    /* node inclusion policy (NodeAffinityPolicy=Honor, NodeTaintsPolicy=Ignore by default,
       common.go:107-122): precomputed per node, NULL = all included */
    const uint8_t *node_included;
} ccref_spread_constraint;

/* InterPodAffinity in the integer world (P/interpodaffinity/{filtering.go:204-432, scoring.go:81-290}).
 * The reference keys its count/score maps by topology PAIR (key, value), so terms sharing a topology key share
 * entries; hence everything here is per distinct KEY.  The caller evaluates label selectors / namespaces of
 * existing pods once (strings) and hands over per-node counts; simulated clones are identical to the incoming
 * pod, so what one clone adds is a per-pod constant ("self" fields). */
typedef struct {
    int32_t n_keys;
    int32_t key_col[CCREF_MAX_IPA_KEYS];   /* label column of the topology key */
    int32_t key_ndom[CCREF_MAX_IPA_KEYS];  /* value ids 1..n */
    /* incoming pod's REQUIRED affinity terms (filtering.go:187-199: an existing pod counts only if it matches ALL terms) */
    int32_t n_aff_terms;
    int32_t aff_key[CCREF_MAX_IPA_TERMS];  /* index into key_col */
    int32_t self_aff;                      /* podMatchesAllAffinityTerms(own terms, own pod) */
    const int32_t *aff_existing;           /* [n] existing pods on the node matching all terms, NULL = 0 */
    /* incoming pod's REQUIRED anti-affinity terms (counted per term) */
    int32_t n_anti_terms;
    int32_t anti_key[CCREF_MAX_IPA_TERMS];
    int32_t anti_self[CCREF_MAX_IPA_TERMS];            /* the term's selector matches the pod itself */
    const int32_t *anti_existing[CCREF_MAX_IPA_TERMS]; /* [n] existing pods matching term t, NULL = 0 */
    /* existing pods' required anti-affinity terms that match the incoming pod, per topology key (filtering.go:204-232) */
    const int32_t *exist_anti[CCREF_MAX_IPA_KEYS]; /* [n] (pod, term) pairs with that key on the node, NULL = 0 */
    /* Score (scoring.go:81-125): net weight existing pods put on the node's (key, value) pair, what one clone adds,
     * and how many processTerm hits exist (PreScore returns Skip without any, scoring.go:199-201) */
    const int64_t *score_existing[CCREF_MAX_IPA_KEYS]; /* [n], NULL = 0 */
    int64_t score_self[CCREF_MAX_IPA_KEYS];
    int64_t entries_existing;                          /* hits among existing pods (node has the key) */
    int32_t self_entries[CCREF_MAX_IPA_KEYS];          /* hits one clone adds on a node that has the key */
} ccref_ipa;

typedef struct {
    /* fit.go:224-233 computePodResourceRequest; col order as above */
    int64_t req[CCREF_MAX_RES];
    int32_t has_scalar_entries; /* len(ScalarResources) != 0 even if all zero (fit.go:578-583) */
    /* NonZero requests: resource_allocation.go:118-148 == types.go:700-734 Non0CPU/Non0Mem */
    int64_t nz_mcpu, nz_mem;
    /* TaintToleration per distinct taint set */
    int32_t n_taintsets;
    const uint8_t *taint_filter_ok;  /* 1 iff every NoSchedule/NoExecute taint is tolerated */
    const int32_t *taint_prefer_cnt; /* # PreferNoSchedule taints not tolerated (taint_toleration.go:169-181) */
    int32_t tolerates_unschedulable; /* node_unschedulable.go:141-147 */
    /* NodeAffinity (P/nodeaffinity/node_affinity.go) */
    int32_t affinity_filter_active; /* 0 = PreFilter returned Skip (:149-155) */
    int32_t has_node_selector;      /* spec.nodeSelector != nil */
    ccref_term node_selector;       /* AND of equalities */
    int32_t has_required_terms;     /* requiredDuringScheduling... != nil */
    int32_t n_required;             /* ORed; 0 terms => nothing matches */
    const ccref_term *required;
    int32_t n_preferred; /* 0 => PreScore Skip (:243-246) */
    const ccref_term *preferred;
    const ccref_requirement *reqs;
    const uint8_t *req_tables;
    /* PodTopologySpread */
    int32_t n_spread;
    ccref_spread_constraint spread[CCREF_MAX_TSC];
    int32_t has_ipa; /* 0 = no inter-pod (anti)affinity anywhere: PreFilter and PreScore return Skip */
    ccref_ipa ipa;
    /* NodePorts (P/nodeports/node_ports.go:67-76,148-176).  has_host_ports = len(util.GetHostPorts(pod)) != 0
     * (S/util/utils.go:175-210: hostPort > 0 of containers and restartable init containers; 0 => PreFilter Skip).
     * host_ports_conflict[n] = 1 iff a port an EXISTING pod of node n holds conflicts with one of the pod's
     * (HostPortInfo.CheckConflict, kube-scheduler/framework/types.go:499-528 -- ip / protocol strings: evaluated by the
     * caller).  A simulated clone holds the same ports, so a node that took one clone conflicts with the next
     * (NodeInfo.updateUsedPorts, S/framework/types.go:431-439). */
    int32_t has_host_ports;
    const uint8_t *host_ports_conflict; /* [n], NULL = none */
    /* ImageLocality (P/imagelocality/image_locality.go:54-115): the node's score 0..100 for this pod's images -- image
     * names are strings, so the caller evaluates ccref_image_locality_score per node.  NULL = 0 everywhere. */
    const uint8_t *image_score; /* [n] */
    /* VolumeRestrictions / NodeVolumeLimits / VolumeBinding / VolumeZone: after NodeResourcesFit, before PodTopologySpread
     * (default_plugins.go:40-45).  volume_veto[n] = 0 or the code (1..CCREF_VOL_CODES) of the first of them that rejects node n
     * against the snapshot's pods -- string and object work, evaluated by the caller; volume_exclusive: the pod's own disks conflict
     * with a clone's (isVolumeConflict, volume_restrictions.go:105-150), so a node that took a clone fails with the disk-conflict
     * reason (:310-313). */
    int32_t volume_exclusive;
    const uint8_t *volume_veto; /* [n], NULL = none */
    /* PodTopologySpread scoring with requireAllTopologies = false (scoring.go:140: the pod has no constraints of its own and the
     * plugin's SYSTEM DEFAULT constraints apply -- "this allows nodes that don't have a zone label to still have hostname
     * spreading"): no node is ignored, a missing key counts as the value "" when the domains are sized and counted, and scores
     * nothing for that constraint (scoring.go:61-115, 147-178, 205-219).  0 = the pod's own constraints: every key required. */
    int32_t soft_relaxed;
} ccref_pod;

typedef struct {
    uint32_t filter_mask;
    /* score plugin weights; 0 = plugin not enabled (default_plugins.go:38-50) */
    int32_t w_taint, w_nodeaffinity, w_fit, w_balanced, w_topologyspread;
    int32_t w_interpodaffinity; /* default 2 */
    /* NodeResourcesFit scoringStrategy LeastAllocated resources (defaults.go:33-36) */
    int32_t n_fit_res;
    int32_t fit_res[CCREF_MAX_RES];
    int64_t fit_res_w[CCREF_MAX_RES];
    /* NodeResourcesBalancedAllocation resources (defaults.go:229-245) */
    int32_t n_bal_res;
    int32_t bal_res[CCREF_MAX_RES];
    int32_t percentage_of_nodes_to_score; /* 0 = adaptive (schedule_one.go:697-723) */
    int32_t w_imagelocality; /* default 1 (default_plugins.go:49); no NormalizeScore */
} ccref_profile;

typedef struct {
    int64_t placed;
    int32_t stop; /* CCREF_STOP_* */
    int32_t *per_node_count; /* caller-allocated [n], zeroed by ccref_run */
    int32_t *log;            /* optional caller-allocated placement log (node idx per placement) */
    int64_t log_cap;
    int64_t hist[CCREF_NREASON]; /* terminal round only */
    int64_t *hist_taintset;      /* optional caller-allocated [n_taintsets]: nodes rejected by TaintToleration */
    int64_t n_code_unschedulable; /* nodes whose terminal status code is plain Unschedulable (preemption msg) */
    int64_t rounds;
    int64_t evaluated_total; /* sum over rounds of nodes the filters visited */
    int32_t last_evaluated, last_feasible;
} ccref_result;

/* scheduler state that survives rounds (S/scheduler.go nextStartNodeIndex) */
typedef struct {
    int64_t next_start_node_index;
} ccref_sched_state;

/* One scheduling cycle (schedule_one.go:430-478 schedulePod + :967-984 assume).
 * Returns the winning node index (and applies the placement), or -1 when no node fits, in which
 * case res->hist etc. describe the FitError.  res may be NULL. */
int64_t ccref_schedule_one(const ccref_profile *prof, ccref_nodes *nodes, const ccref_pod *pod, ccref_sched_state *st,
                           ccref_result *res);

/* The simulator loop (pkg/framework/simulator.go:297-381): place clones until Unschedulable or
 * max_limit placements (max_limit <= 0: unlimited). threads>1 evaluates the node loop with OpenMP
 * (same results; mirrors Parallelism=16 for timing only). */
int ccref_run(const ccref_profile *prof, ccref_nodes *nodes, const ccref_pod *pod, int64_t max_limit, int threads,
              ccref_result *res);

/* Several pod specs cycled round-robin against one snapshot (BASELINE.json configs[4]); see ccref.c. */
typedef struct {
    int64_t placed;
    int32_t stop;      /* CCREF_STOP_* */
    int32_t stop_spec; /* the spec whose pod was Unschedulable, -1 otherwise */
    int32_t *per_node_count; /* optional caller-allocated [n]: simulated pods per node (all specs) */
    int32_t *per_spec_count; /* optional caller-allocated [n_pods] */
    int32_t *log;            /* optional: node index per placement (placement i is spec i mod n_pods) */
    int64_t log_cap;
    int64_t hist[CCREF_NREASON]; /* terminal cycle (the failing pod) */
    int64_t *hist_taintset;      /* optional caller-allocated [n_taintsets of the failing pod] */
    int64_t n_code_unschedulable;
    int64_t rounds;
} ccref_multi_result;
int ccref_run_multi(const ccref_profile *prof, ccref_nodes *nodes, const ccref_pod *pods, int32_t n_pods, int64_t max_limit,
                    int threads, ccref_multi_result *res);

/* plugin score unit functions, exported for the known-answer vectors */
int64_t ccref_least_allocated(const int64_t *requested, const int64_t *allocatable, const int64_t *weights, int n);
int64_t ccref_balanced_allocation(const int64_t *requested, const int64_t *allocatable, int n);
void ccref_default_normalize(int64_t max_priority, int reverse, int64_t *scores, int64_t n);
int32_t ccref_num_feasible_nodes_to_find(int32_t percentage, int32_t num_all_nodes);
double ccref_go_log(double x); /* restatement of Go's math.Log (pure-Go path) */
/* RunScorePlugins' weight-and-sum block and selectHost under the canonical tie-break: the functions the cycle itself calls */
void ccref_weigh(int64_t *total, const int64_t *scores, int64_t weight, int64_t n);
int64_t ccref_select_host(const int64_t *total, int64_t n);
/* test hooks: PodTopologySpread's / InterPodAffinity's PreFilter state of a cluster (match_num etc. hold n_domains + 1 entries; -1 = the domain is
 * absent from TpValueToMatchNum; totals = len(affinityCounts) entries, existing anti-affinity entries, PreScore hits) */
int ccref_unit_pts_prefilter(const ccref_nodes *nd, const ccref_pod *pod, const int32_t *placed, int c, int64_t *match_num, int64_t *min_match,
                             int64_t *n_dom);
int ccref_unit_pts_scores(const ccref_nodes *nd, const ccref_pod *pod, const int32_t *placed, const int64_t *feas, int64_t nf, int64_t *raw,
                          int64_t *norm, double *weights);
int ccref_unit_ipa_scores(const ccref_nodes *nd, const ccref_pod *pod, const int32_t *placed, const int64_t *feas, int64_t nf, int64_t *raw,
                          int64_t *norm, int32_t *skipped);
int ccref_unit_ipa_build(const ccref_nodes *nd, const ccref_pod *pod, const int32_t *placed, int k, int64_t *aff, int64_t *anti, int64_t *exist,
                         int64_t *score, int64_t *totals);
/* the NormalizeScore steps of PodTopologySpread (scoring.go:226-265; ignored: in IgnoredNodes, NULL = none) and InterPodAffinity
 * (scoring.go:259-290), in place */
void ccref_pts_normalize(int64_t *scores, const uint8_t *ignored, int64_t n);
void ccref_ipa_normalize(int64_t *scores, int64_t n);
/* ImageLocality score of one node (image_locality.go:54-115): size[i] / num_nodes[i] = ImageStateSummary of the i-th
 * pod container (incl. init containers) whose image the node holds; n_containers = len(InitContainers) + len(Containers) */
int64_t ccref_image_locality_score(const int64_t *size, const int32_t *num_nodes, int n_present, int32_t total_nodes,
                                   int n_containers);

/* DefaultPreemption's dry run for the terminal cycle (S/framework/preemption/preemption.go:234-303,741-794;
 * P/defaultpreemption/default_preemption.go:217-310 SelectVictimsOnNode) -- it decides the tail of the FitError message,
 * never a placement (the reference stops on the Unschedulable condition either way, pkg/framework/simulator.go:327-342).
 * `nodes` is the snapshot as loaded, `placed` the clones per node at the terminal cycle (the terminal NodeInfo is rebuilt
 * here).  A victim is an existing pod of lower priority than the incoming one (default_preemption.go:392-396); strings and
 * priorities are the caller's, so the caller hands over per node: how many victims, what they request per column, and
 * whether a REMAINING pod still holds a conflicting host port.  Every node whose filter status is plain Unschedulable is
 * tried: no victims -> no_victims++; all victims removed and the Filter plugins run again -> feasible: nominated = 1
 * (PostFilter returns Success, empty message), else the failing plugin's reasons go to hist.  Other nodes: not_helpful++.
 * Topology-coupled filters (hard spread constraints, inter-pod affinity) are evaluated in the second run against the cycle's
 * PreFilter state, which is the terminal cycle's as long as the removed victims take no part in it; a potential node whose victims
 * do (victim_interacts) is refused with -38: RunPreFilterExtensionRemovePod is not restated. */
typedef struct {
    const int32_t *victim_count;              /* [n], NULL = no victims anywhere */
    const int64_t *victim_req[CCREF_MAX_RES]; /* [n] per column, NULL = 0 */
    const uint8_t *ports_conflict_rest;       /* [n], NULL = none */
    /* [n], NULL = none: a victim of the node takes part in the PreFilter state of a topology-coupled filter of the pod (matches a
     * hard spread selector / a required (anti)affinity term, or carries an anti-affinity term matching the pod) */
    const uint8_t *victim_interacts;
    /* [n], NULL = none: ccref_pod.volume_veto evaluated against the node's REMAINING pods (a victim's disks and claims leave with it) */
    const uint8_t *volume_veto_rest;
} ccref_victims;
typedef struct {
    int32_t nominated;
    int64_t hist[CCREF_NREASON];
    int64_t no_victims, not_helpful;
} ccref_preemption;
int ccref_preemption_dry_run(const ccref_profile *prof, const ccref_nodes *nodes, const ccref_pod *pod, const int32_t *placed,
                             const ccref_victims *victims, ccref_preemption *out);

#ifdef __cplusplus
}
#endif
#endif
