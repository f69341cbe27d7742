/*
 * ccsim.h -- C ABI of libccsim.so, the MI355X-native batched placement engine that replaces the
 * sequential schedule-one-pod loop of kubernetes-sigs/cluster-capacity.
 *
 * This is the drop-in boundary: what a cgo shim in pkg/framework, a Python ctypes
 * harness and the C++ CLI all bind.  Each entry point names the reference interface it replaces.
 * Paths are relative to the reference root; S/ = vendor/k8s.io/kubernetes/pkg/scheduler,
 * P/ = S/framework/plugins.
 *
 * Conventions
 *   - plain C types only; no C++/torch types; no exceptions cross the boundary.
 *   - return value: 0 = OK; < 0 = error (-EINVAL bad argument, -ENOMEM, -EIO HIP failure, -ENOSYS
 *     unsupported configuration).  ccsim_last_error() returns the text.
 *   - the caller owns every input array; the library copies during the call and never retains a
 *     caller pointer (cgo rule).  Output arrays are caller-allocated with explicit capacities.
 *   - strings never cross the boundary: taints/labels/selectors are interned ids and small lookup
 *     tables built by the host layer (cluster-capacity_amd/host).
 *   - a handle is single-caller (the reference calls schedulePod strictly serially,
 *     S/schedule_one.go:65); every entry point sets the HIP device itself.
 *
 * Resource "columns": 0 = cpu (milli), 1 = memory, 2 = ephemeral-storage, 3+k = scalar resource k
 * (S/framework/types.go:940-950 Resource).
 */
#ifndef CCSIM_H
#define CCSIM_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CCSIM_ABI_VERSION 5
#define CCSIM_MAX_SCALAR 8
#define CCSIM_MAX_RES (3 + CCSIM_MAX_SCALAR)
#define CCSIM_MAX_LABEL_COLS 32
#define CCSIM_MAX_TSC 8
#define CCSIM_MAX_IPA_KEYS 4
#define CCSIM_MAX_IPA_TERMS 8

/* filter plugins, default profile order (S/apis/config/v1/default_plugins.go:30-58) */
enum {
    CCSIM_F_UNSCHEDULABLE = 1u << 0,
    CCSIM_F_NODENAME = 1u << 1,
    CCSIM_F_TAINT = 1u << 2,
    CCSIM_F_NODEAFFINITY = 1u << 3,
    CCSIM_F_FIT = 1u << 4,
    CCSIM_F_TOPOLOGYSPREAD = 1u << 5,
    CCSIM_F_INTERPODAFFINITY = 1u << 6,
    CCSIM_F_NODEPORTS = 1u << 7 /* P/nodeports; runs between NodeAffinity and NodeResourcesFit (the bit order is not the plugin order) */
};

/* reason slots of the terminal-round histogram (FitError.Error, S/framework/types.go:787-836) */
enum {
    CCSIM_R_UNSCHEDULABLE = 0,
    CCSIM_R_NODENAME = 1,
    CCSIM_R_NODEAFFINITY = 2,
    CCSIM_R_TOO_MANY_PODS = 3,
    CCSIM_R_RES0 = 4, /* + column */
    CCSIM_R_PTS_MISSING_LABEL = CCSIM_R_RES0 + CCSIM_MAX_RES,
    CCSIM_R_PTS_SKEW,
    CCSIM_R_IPA_AFFINITY,      /* "node(s) didn't match pod affinity rules" */
    CCSIM_R_IPA_ANTI,          /* "node(s) didn't match pod anti-affinity rules" */
    CCSIM_R_IPA_EXISTING_ANTI, /* "node(s) didn't satisfy existing pods anti-affinity rules" */
    CCSIM_R_NODEPORTS,         /* "node(s) didn't have free ports for the requested pod ports" (P/nodeports/node_ports.go:39) */
    /* the volume plugins, in their filter order (default_plugins.go:41-44): slot = CCSIM_R_VOL0 + (ccsim_pod.volume_veto code - 1) */
    CCSIM_R_VOL0,
    CCSIM_R_VOL_DISK_CONFLICT = CCSIM_R_VOL0, /* "node(s) had no available disk" (P/volumerestrictions/volume_restrictions.go:55), Unschedulable */
    CCSIM_R_VOL_RWOP,          /* "node(s) unavailable due to PersistentVolumeClaim with ReadWriteOncePod access mode already in-use by another pod" (:59), Unschedulable */
    CCSIM_R_VOL_MAX_COUNT,     /* "node(s) exceed max volume count" (P/nodevolumelimits/csi.go:44), Unschedulable */
    CCSIM_R_VOL_NODE_AFFINITY, /* "node(s) didn't match PersistentVolume's node affinity" (P/volumebinding/binder.go:67), UnschedulableAndUnresolvable */
    CCSIM_R_VOL_NO_PV,         /* "node(s) didn't find available persistent volumes to bind" (binder.go:65), UnschedulableAndUnresolvable */
    CCSIM_R_VOL_PV_NOT_EXIST,  /* "node(s) unavailable due to one or more pvc(s) bound to non-existent pv(s)" (binder.go:71), UnschedulableAndUnresolvable */
    CCSIM_R_VOL_ZONE,          /* "node(s) had no available volume zone" (P/volumezone/volume_zone.go:61), UnschedulableAndUnresolvable */
    CCSIM_NREASON
};
#define CCSIM_VOL_CODES 7           /* volume_veto codes 1 .. CCSIM_VOL_CODES */
#define CCSIM_VOL_LAST_UNSCHEDULABLE 3 /* codes 1..3 are plain Unschedulable (preemption dry-run candidates), the rest UnschedulableAndUnresolvable */

enum { CCSIM_STOP_UNSCHEDULABLE = 0, CCSIM_STOP_LIMIT = 1, CCSIM_STOP_NO_NODES = 2 };

/* ccsim_run modes */
enum {
    CCSIM_MODE_SEQUENTIAL = 0, /* one full pods x nodes scan per placement round (the literal reference loop) */
    CCSIM_MODE_BATCHED = 1     /* exact level-batched resolution: many rounds per scan, identical placement sequence */
};

typedef struct ccsim_engine ccsim_engine;

typedef struct {
    int32_t abi_version; /* CCSIM_ABI_VERSION */
    int32_t device;      /* HIP device ordinal */
    void *stream;        /* hipStream_t to enqueue on, or NULL: the engine creates its own */
    int32_t rounds_per_sync; /* placement rounds enqueued between host checks of the done flag; 0 = default */
    int32_t use_graph;       /* replay rounds from a captured hipGraph (1) or launch eagerly (0) */
    int32_t time_passes;     /* measurement runs: launch eagerly with HIP event stamps on every launch of the pass's
                                dominant kernel (sequential: k_scan; batched: k_level_commit) and report their summed
                                duration in ccsim_report.pass_kernel_ns */
} ccsim_config;

/* Node snapshot, structure-of-arrays, canonical node order (S/backend/cache/node_tree.go:119-143).
 * Replaces the []NodeInfo the reference's Cache.UpdateSnapshot hands to schedulePod
 * (S/backend/cache/cache.go:194-288; NodeInfo: S/framework/types.go:160-200). */
typedef struct {
    int64_t n_nodes;       /* nodes in THIS shard */
    int64_t global_offset; /* canonical index of this shard's first node (0 on one GPU) */
    int64_t n_global;      /* nodes in the whole snapshot (== n_nodes on one GPU) */
    int32_t n_scalar;
    const int64_t *alloc[CCSIM_MAX_RES]; /* Allocatable per column, [n_nodes]; NULL = all zero */
    const int32_t *alloc_pods;           /* Allocatable.AllowedPodNumber */
    const int64_t *req[CCSIM_MAX_RES];   /* Requested per column */
    const int64_t *nz_mcpu, *nz_mem;     /* NonZeroRequested */
    const int32_t *pod_count;            /* len(NodeInfo.Pods) */
    const int32_t *taintset_id;          /* id of the node's distinct Spec.Taints list */
    const uint8_t *unschedulable;        /* Spec.Unschedulable */
    int32_t n_label_cols;
    const int32_t *label_cols[CCSIM_MAX_LABEL_COLS]; /* value id of label key k on each node; 0 = absent */
} ccsim_nodes;

/* one matchExpression / matchField, pre-evaluated by the host against every distinct value of the
 * label column (AM/labels/selector.go:246-293 semantics incl. Gt/Lt ParseInt): a table lookup */
typedef struct {
    int32_t col;
    int32_t table_off; /* matches iff req_tables[table_off + label_cols[col][node]] != 0 */
} ccsim_requirement;

typedef struct {
    int32_t first_req, n_req; /* AND; n_req == 0 matches nothing (component-helpers nodeaffinity.go:60-64) */
    int32_t weight;           /* preferred terms only */
} ccsim_term;

/* One topologySpreadConstraint after interning (P/podtopologyspread/common.go:42-56).  hard = DoNotSchedule ->
 * Filter (filtering.go:235-356); soft = ScheduleAnyway -> Score (scoring.go:61-265).  The engine keeps
 * TpValueToMatchNum / TopologyPairToPodCounts as per-domain count tables in HBM, updated at every placement. */
typedef struct {
    int32_t col;         /* label column of the topologyKey (value id 0 = node lacks the key) */
    int32_t max_skew;    /* >= 1 */
    int32_t min_domains; /* >= 1 (nil -> 1) */
    int32_t hard;        /* 1 = DoNotSchedule (filter), 0 = ScheduleAnyway (score) */
    int32_t self_match;  /* 1 if the pod's own labels match the constraint's selector (common.go:144-159) */
    int32_t n_domains;   /* value ids of `col` are 1..n_domains */
    const int32_t *node_match_count; /* [n_nodes] existing pods on the node matching the selector, NULL = 0 */
    const uint8_t *node_included;    /* [n_nodes] node inclusion policies (common.go:107-122), NULL = all */
    int32_t is_hostname; /* topologyKey == kubernetes.io/hostname: scored per node, not per domain (scoring.go:214-215) */
    /* ScheduleAnyway constraints scored with requireAllTopologies = false (scoring.go:140: the plugin's system default constraints on a
     * pod without constraints of its own): the caller gives the nodes that lack the key ONE MORE value id of `col` (counted in
     * n_domains) and names it here.  It is sized and counted like any domain (the reference's "" value, scoring.go:96-103,166-173) but
     * scores nothing for this constraint (scoring.go:210), and since every node then carries every key nobody is ignored.
     * 0 = none (the pod's own constraints: a node without the key is ignored, scoring.go:84-88).  (Occupies what was padding: v3 layout.) */
    int32_t missing_value;
} ccsim_spread_constraint;

/* InterPodAffinity in the integer world (P/interpodaffinity/{filtering.go:204-432, scoring.go:81-290}).  The
 * reference keys its count / score maps by topology PAIR (key, value): terms sharing a topology key share
 * entries, so everything is per distinct KEY.  The caller evaluates selectors / namespaces of the snapshot's pods
 * once (strings) and hands over per-node counts; simulated clones are identical to the incoming pod, so what one
 * clone adds is a per-pod constant (the "self" fields).  The engine keeps one table per key per map in HBM. */
typedef struct {
    int32_t n_keys;
    int32_t key_col[CCSIM_MAX_IPA_KEYS];  /* label column of the topology key */
    int32_t key_ndom[CCSIM_MAX_IPA_KEYS]; /* value ids 1..n */
    int32_t n_aff_terms;                  /* REQUIRED affinity terms (an existing pod counts iff it matches ALL) */
    int32_t aff_key[CCSIM_MAX_IPA_TERMS]; /* index into key_col */
    int32_t self_aff;                     /* podMatchesAllAffinityTerms(own terms, own pod) */
    const int32_t *aff_existing;          /* [n_nodes] existing pods matching all terms, NULL = 0 */
    int32_t n_anti_terms;                 /* REQUIRED anti-affinity terms (counted per term) */
    int32_t anti_key[CCSIM_MAX_IPA_TERMS];
    int32_t anti_self[CCSIM_MAX_IPA_TERMS];
    const int32_t *anti_existing[CCSIM_MAX_IPA_TERMS]; /* [n_nodes] existing pods matching term t, NULL = 0 */
    const int32_t *exist_anti[CCSIM_MAX_IPA_KEYS];     /* [n_nodes] (existing pod, anti term) pairs with that key
                                                          matching the incoming pod, NULL = 0 */
    const int64_t *score_existing[CCSIM_MAX_IPA_KEYS]; /* [n_nodes] net weight put on the node's pair, NULL = 0 */
    int64_t score_self[CCSIM_MAX_IPA_KEYS];            /* net weight one clone adds (both directions) */
    int64_t entries_existing;                          /* processTerm hits among existing pods (0 hits -> Skip) */
    int32_t self_entries[CCSIM_MAX_IPA_KEYS];          /* hits one clone adds on a node that has the key */
} ccsim_ipa;

/* Pod-spec constants.  Replaces the per-cycle PreFilter/PreScore state of the plugins:
 * fit.go:224-233 (computePodResourceRequest), resource_allocation.go:118-148,
 * taint_toleration.go:111-121,146-153, node_affinity.go:147-197,241-258. */
typedef struct {
    int64_t req[CCSIM_MAX_RES];
    int32_t has_scalar_entries; /* len(ScalarResources) != 0 (fit.go:578-583) */
    int64_t nz_mcpu, nz_mem;    /* non-zero requests (100m / 200Mi container defaults applied) */
    int32_t n_taintsets;
    const uint8_t *taint_filter_ok;  /* [n_taintsets] every NoSchedule/NoExecute taint tolerated */
    const int32_t *taint_prefer_cnt; /* [n_taintsets] untolerated PreferNoSchedule taints (<= 2047) */
    int32_t tolerates_unschedulable;
    int32_t affinity_filter_active; /* 0 = NodeAffinity PreFilter returned Skip */
    int32_t has_node_selector;
    ccsim_term node_selector;
    int32_t has_required_terms;
    int32_t n_required;
    const ccsim_term *required;
    int32_t n_preferred;
    const ccsim_term *preferred; /* sum of weights must stay < 2^13 (the packed static word) */
    int32_t n_reqs;
    const ccsim_requirement *reqs;
    int64_t req_tables_len;
    const uint8_t *req_tables;
    int32_t n_spread; /* <= CCSIM_MAX_TSC */
    ccsim_spread_constraint spread[CCSIM_MAX_TSC];
    int32_t has_ipa; /* 0 = no inter-pod (anti)affinity anywhere (PreFilter / PreScore Skip) */
    ccsim_ipa ipa;
    /* NodePorts (P/nodeports/node_ports.go:67-76 PreFilter, :148-176 Filter).  has_host_ports: the pod asks for at least
     * one host port (util.GetHostPorts, S/util/utils.go:175-210; 0 = PreFilter Skip).  host_ports_conflict[n] = 1 iff a
     * port held by an EXISTING pod of node n conflicts with one of them (HostPortInfo.CheckConflict,
     * kube-scheduler/framework/types.go:499-528: ip / protocol strings, evaluated by the caller).  A clone holds the
     * same ports as the next one, so a node takes at most one (NodeInfo.updateUsedPorts, S/framework/types.go:431-439):
     * the engine keeps UsedPorts as one more resource column (allocatable 1). */
    int32_t has_host_ports;
    const uint8_t *host_ports_conflict; /* [n_nodes], NULL = no existing pod conflicts */
    /* ImageLocality (P/imagelocality/image_locality.go:54-115): the node's score 0..100 for the pod's container images
     * (sizes x spread over the snapshot's nodes: strings, evaluated by the caller); folded into the static word. */
    const uint8_t *image_score; /* [n_nodes], NULL = 0 */
    /* The volume plugins -- VolumeRestrictions, NodeVolumeLimits, VolumeBinding, VolumeZone -- run after NodeResourcesFit and before
     * PodTopologySpread (default_plugins.go:40-45).  Their verdicts are string / object-graph work (volume ids, PV labels and node
     * affinity, CSINode limits) that does not change while clones are placed, EXCEPT a clone's own disks: the caller evaluates them.
     * volume_veto[n] = 0, or the code 1..CCSIM_VOL_CODES of the FIRST volume plugin that rejects node n against the snapshot's pods
     * (1 disk conflict volume_restrictions.go:310-313, 2 ReadWriteOncePod :314-318, 3 max volume count nodevolumelimits/csi.go:255-345,
     * 4 PersistentVolume node affinity, 5 no persistent volume to bind, 6 bound to a non-existent PV volumebinding/volume_binding.go:417-445,
     * 7 volume zone volumezone/volume_zone.go:191-240); reported in slot CCSIM_R_VOL0 + code - 1 for nodes that pass every filter up to and
     * including NodeResourcesFit (first failing plugin reports, framework.go:897-930).
     * volume_exclusive = 1: two clones conflict on the same node (isVolumeConflict of the pod's volumes with themselves,
     * volume_restrictions.go:105-150: an EBS volume, a GCE PD / ISCSI / RBD mount that is not read-only), so a node takes at most
     * one -- the NodePorts construction, with the disk-conflict reason attributed AFTER NodeResourcesFit. */
    int32_t volume_exclusive;
    const uint8_t *volume_veto; /* [n_nodes], NULL = no node is rejected */
} ccsim_pod;

/* Scheduler profile: which plugins run and their weights/args.  Replaces
 * KubeSchedulerConfiguration.Profiles[0] (pkg/utils/utils.go:90-143; defaults
 * S/apis/config/v1/default_plugins.go:30-58, defaults.go:33-36,229-245). */
typedef struct {
    uint32_t filter_mask;
    int32_t w_taint, w_nodeaffinity, w_fit, w_balanced, w_topologyspread; /* 0 = score plugin disabled */
    int32_t w_interpodaffinity;
    int32_t n_fit_res;
    int32_t fit_res[CCSIM_MAX_RES];
    int64_t fit_res_w[CCSIM_MAX_RES];
    int32_t n_bal_res;
    int32_t bal_res[CCSIM_MAX_RES];
    /* KubeSchedulerConfiguration.percentageOfNodesToScore (schedule_one.go:697-723): 100 = every node is scored;
     * 0 = adaptive (50 - N/125, at least 5 %); below 100 the search keeps the first numFeasibleNodesToFind
     * feasible nodes of a rotating visiting order (schedule_one.go:610-680).  The sampled search is order-dependent:
     * CCSIM_MODE_SEQUENTIAL only (ccsim_run / ccsim_schedule_one; on shards two exchanges per cycle, ccsim_dist_* -- since
     * round 6 for pods with topology-coupled plugins too); snapshots with fewer than 100 nodes are always searched completely. */
    int32_t percentage_of_nodes_to_score;
    int32_t w_imagelocality; /* default 1 (default_plugins.go:49); 0 = disabled.  No NormalizeScore. */
} ccsim_profile;

/* Result of ccsim_run.  Replaces ClusterCapacity.Status{Pods, StopReason}
 * (pkg/framework/simulator.go:90-93) + the inputs of parsePodsReview (report.go:146-180). */
typedef struct {
    int64_t placed;
    int32_t stop; /* CCSIM_STOP_* */
    int32_t *per_node_count; /* caller-allocated [per_node_cap]: simulated pods per node of this shard */
    int64_t per_node_cap;
    int32_t *log; /* optional caller-allocated placement log: global node index per placement, in order */
    int64_t log_cap;
    int64_t log_len;
    int64_t hist[CCSIM_NREASON]; /* terminal round: nodes per failure reason (this shard) */
    int64_t *hist_taintset;      /* optional caller-allocated [hist_taintset_cap] */
    int32_t hist_taintset_cap;
    int64_t n_code_unschedulable; /* nodes whose terminal status is plain Unschedulable */
    /* counters */
    int64_t rounds;          /* scheduling cycles simulated (placements + the terminal one) */
    int64_t scans;           /* passes executed (sequential: one full pods x nodes scan per attempt of a round; batched: one per
                                score level -- a commit pass off the score cache, or the rare full pass) */
    int64_t evaluated_total; /* (pod, node) evaluations the reference semantics imply = rounds * n */
    int32_t last_feasible;   /* FeasibleNodes of the last cycle */
    int64_t kernel_ns;       /* GPU time of all launches of the run (HIP events on the engine's stream) */
    int64_t pass_kernel_ns;  /* cfg.time_passes: summed duration of the dominant kernel's launches alone ... */
    int64_t pass_launches;   /* ... and how many were launched (incl. early-exit launches after the done flag) */
    int64_t bytes_per_scan;  /* algorithmic bytes one scan reads: n_nodes * sum of enabled column widths */
    /* several pod specs (ccsim_set_pods): */
    int32_t *per_spec_count; /* optional caller-allocated [per_spec_cap]: placements per pod spec */
    int32_t per_spec_cap;
    int32_t stop_spec;       /* the spec whose pod was Unschedulable (hist describes ITS FitError), -1 otherwise */
    /* ABI 5: the per-node counts in the narrowest element that holds them.  At 1M nodes the int32 vector is 4 MB -- 72 us over PCIe, a
     * quarter of a whole batched run -- while no node of a cluster holds more simulated pods than its pod capacity (110 by default,
     * `Allocatable.AllowedPodNumber`).  A caller that can read 1- or 2-byte counts offers an array of per_node_cap elements of that
     * width; when the run's form supports it (the persistent batched launch) and every count provably fits (the snapshot's largest
     * pod capacity < 2^(8 width)) the engine fills THAT array instead of per_node_count and says so in per_node_filled_width.
     * Otherwise per_node_count is filled as before (offer both).  Same values either way (tests/test_persist.py). */
    void *per_node_count_narrow;    /* optional caller-allocated [per_node_cap] elements of per_node_narrow_width bytes */
    int32_t per_node_narrow_width;  /* in: 1 (uint8_t) or 2 (uint16_t); 0 = not offered */
    int32_t per_node_filled_width;  /* out: 4 = per_node_count was filled, 1 / 2 = per_node_count_narrow was, 0 = neither (none offered / usable) */
} ccsim_report;

/* Result of one scheduling cycle.  Replaces ScheduleResult (S/scheduler.go:154-164) as returned by
 * Scheduler.SchedulePod (S/scheduler.go:88-91, S/schedule_one.go:430-478). */
typedef struct {
    int64_t node; /* global index of the suggested host, -1 = FitError */
    int32_t evaluated_nodes;
    int32_t feasible_nodes;
} ccsim_cycle;

int32_t ccsim_abi_version(void);

/* framework.New / createScheduler (pkg/framework/simulator.go:107-158,383-431) */
int ccsim_create(const ccsim_config *cfg, ccsim_engine **out);
void ccsim_destroy(ccsim_engine *e); /* ClusterCapacity.Close (simulator.go:314-325) */
const char *ccsim_last_error(const ccsim_engine *e);

/* SyncWithClient's end product (simulator.go:176-295): the node snapshot, copied into HBM */
int ccsim_load_nodes(ccsim_engine *e, const ccsim_nodes *nodes);
int ccsim_set_profile(ccsim_engine *e, const ccsim_profile *profile);
/* the simulated pod (New(..., simulatedPod, ...) simulator.go:107): uploads tables, runs the static
 * (unschedulable / taint / node-affinity) kernel once */
int ccsim_set_pod(ccsim_engine *e, const ccsim_pod *pod);

/* Several pod specs against one snapshot (BASELINE.json configs[4]: "100k nodes x 1024 genpod pod specs").  The
 * reference simulates ONE template (New(..., simulatedPod, ...) simulator.go:107); with P specs the same loop
 * (simulator.go:297-381) takes the next pod ROUND-ROBIN: placement i is a clone of spec i mod P, every cycle is the
 * reference's schedulePod for that pod against everything placed so far, and the run ends like the reference's -- at the
 * first pod reported Unschedulable (ccsim_report.stop_spec) or at max_limit.  ccsim_set_pods(e, pods, 1) == ccsim_set_pod.
 * With P > 1, ccsim_run evaluates WINDOWS of consecutive pods against the HBM-resident node columns in one pass (each
 * node's columns are read once per window and shared by the window's pods) and commits them in order with an exact
 * validation of every pod's choice against the placements of the pods before it (csrc/ccsim_multi.h).
 * What P > 1 supports: the Filter/Score plugins of ccsim_set_pod except ScheduleAnyway spread constraints; at most two
 * DoNotSchedule constraints per spec (<= 62 domains each) over at most two label columns in total; inter-pod
 * affinity only as REQUIRED ANTI-affinity of a spec to its own clones on a one-node-per-domain key (kubernetes.io/hostname);
 * no selector of one spec may match the clones of another (the caller's labels are disjoint); requests over cpu / memory
 * only; no host ports; ImageLocality scores per spec (round 4); percentageOfNodesToScore 100; plugin weights whose total score stays below
 * 2^21 (the scan packs (score, node) into 32 bits).  Anything else: -ENOSYS. */
int ccsim_set_pods(ccsim_engine *e, const ccsim_pod *pods, int32_t n_pods);
/* Scheduler.SchedulePod + assume for one pod of spec pod_idx (the B2 seam with several templates) */
int ccsim_schedule_pod(ccsim_engine *e, int32_t pod_idx, ccsim_cycle *out);

/* ClusterCapacity.Run (simulator.go:356-381): place clones until Unschedulable or max_limit
 * (<= 0: unlimited).  After ccsim_set_pods with P > 1 the specs are cycled round-robin (`mode` is ignored). */
int ccsim_run(ccsim_engine *e, int64_t max_limit, int32_t mode, ccsim_report *out);

/* Scheduler.SchedulePod + assume for one pod (S/schedule_one.go:430-478,967-984).  For a template without topology-coupled plugins the
 * call is ONE launch on the resident block summaries (csrc/ccsim_search_full.h when every node is scored, csrc/ccsim_sampled.h under
 * percentageOfNodesToScore < 100), not a pass over the nodes: the first call (and the first after any entry point that changes the
 * snapshot or the pod spec) builds them. */
int ccsim_schedule_one(ccsim_engine *e, ccsim_cycle *out);

/* Read back the dynamic node columns (NodeInfo.Requested etc.) of this shard; any pointer may be NULL. */
int ccsim_read_state(ccsim_engine *e, int64_t *req_mcpu, int64_t *req_mem, int64_t *nz_mcpu, int64_t *nz_mem,
                     int32_t *pod_count);

/* ---- multi-GPU stepping: one rank per GPU, node-range shards, ONE collective per round ----
 * (no reference counterpart: the reference is one process).
 * sendbuf / recvbuf are device buffers the caller owns (e.g. torch tensors): int64[CCSIM_XCHG_WORDS]
 * and int64[n_ranks * CCSIM_XCHG_WORDS].  Per pass (sequential: one placement round; batched: one
 * score level = many rounds) every rank:
 *     ccsim_dist_scan()    full pass over the shard + reduce -> this rank's record in sendbuf
 *     all-gather sendbuf -> recvbuf over RCCL/xGMI on the same stream (the max-loc exchange:
 *                          record word 0 is the packed (score, position) key combined with MAX)
 *     ccsim_dist_decide()  every rank reduces the gathered records identically; only the rank that
 *                          owns a winning node applies NodeInfo.update to its HBM columns
 * ccsim_dist_poll() synchronizes and reads the done flag; call it every few passes, after the SAME number of passes on
 * every rank (the batched mode launches its rare full pass only in the first two passes after a begin / poll: a full
 * pass that falls due in between makes the passes up to the next poll no-ops on all ranks alike).
 * With a placement log each rank fills the positions of ITS placements in its own log copy and leaves
 * -1 elsewhere: the element-wise maximum over ranks is the global log. */
#define CCSIM_XCHG_WORDS 32
int ccsim_dist_begin(ccsim_engine *e, int64_t max_limit, int32_t mode, int32_t n_ranks, int32_t rank, void *sendbuf,
                     void *recvbuf, int64_t log_cap);
int ccsim_dist_scan(ccsim_engine *e);
int ccsim_dist_decide(ccsim_engine *e);
int ccsim_dist_poll(ccsim_engine *e, int32_t *done, int64_t *placed);
int ccsim_dist_finish(ccsim_engine *e, ccsim_report *out);

/* ---- multi-GPU, driven by the library: the same protocol with the loop and the collective inside libccsim.so ----
 * The engine owns an RCCL communicator (librccl.so.1 is bound with dlopen the first time one of these is called: no
 * link-time dependency) and its exchange buffers; ccsim_dist_run enqueues scan -> ncclAllGather (256 B per rank over
 * xGMI, on the engine's stream) -> decide for 32 passes per host poll, until every rank's replicated state says done.
 * Rendezvous is the caller's: rank 0 calls ccsim_dist_unique_id and distributes the 128 bytes by any means (the Python
 * host: torch.distributed.broadcast_object_list; a cgo host: whatever the job launcher offers), then every rank calls
 * ccsim_dist_comm_init (collective) after ccsim_load_nodes of ITS shard.  ccsim_dist_sync_tables replaces the caller-side
 * all-reduce of ccsim_dist_table (collective; call it after every ccsim_set_pod). */
#define CCSIM_DIST_ID_BYTES 128
int ccsim_dist_unique_id(uint8_t *id_out /* [CCSIM_DIST_ID_BYTES] */);
int ccsim_dist_comm_init(ccsim_engine *e, const uint8_t *id /* [CCSIM_DIST_ID_BYTES] */, int32_t n_ranks, int32_t rank);
int ccsim_dist_sync_tables(ccsim_engine *e);
/* What the communicator itself says (ncclCommCount / ncclCommUserRank), not what the caller passed: the number of ranks that really
 * joined and this engine's rank among them.  bench.py prints it as `rccl_ranks_seen` beside `n_gpus`.  -EINVAL before
 * ccsim_dist_comm_init. */
int ccsim_dist_comm_size(ccsim_engine *e, int32_t *n_ranks_out, int32_t *rank_out);
/* ClusterCapacity.Run on the sharded snapshot: every rank calls it with the same max_limit / mode; out describes THIS
 * shard (per_node_count, hist) plus the global totals (placed, stop, rounds); with a log each rank fills its own
 * placements (-1 elsewhere), as ccsim_dist_finish does. */
int ccsim_dist_run(ccsim_engine *e, int64_t max_limit, int32_t mode, ccsim_report *out);

/* ---- multi-GPU, one template with topology-coupled plugins: WINDOWS of placements per exchange (round 5; csrc/ccsim_coupled.h "windows
 * on shards").  For the shape ONE hard spread constraint over a shared key + ONE unique-per-node inter-pod key (zone
 * spread + hostname anti-affinity: BASELINE config 5's pod shape), percentageOfNodesToScore = 100, sequential mode.  After
 * ccsim_dist_begin on every rank:
 *   ccsim_dist_cw_eligible  1 if this rank can take part (collect the minimum over the ranks)
 *   ccsim_dist_cw_enable    the minimum; 0 = the pass protocol only.  With 1, per window and on every rank:
 *   ccsim_dist_cw_scan      the pass over the shard -> this rank's window record (classes: tuple, statistics, staged list entries)
 *   -- all-gather `bytes_per_rank` bytes per rank from `send` into `recv` (ccsim_dist_cw_buffers: engine-owned device buffers) --
 *   ccsim_dist_cw_decide    classes unified and lists merged over the ranks, up to 4096 scheduling cycles decided identically on every
 *                           rank, owners apply.  ccsim_dist_poll as usual; a window no rank can take sets the run to one pass per
 *                           placement on every rank alike (ccsim_dist_scan / _decide continue it from the current state).
 * ccsim_dist_run does all of this over the engine's communicator (ncclAllGather of 73 KB per rank and window).  CCSIM_CW_SHARDS=0
 * turns it off. */
int ccsim_dist_cw_eligible(ccsim_engine *e);
int ccsim_dist_cw_enable(ccsim_engine *e, int32_t all_ok);
int ccsim_dist_cw_buffers(ccsim_engine *e, void **send, void **recv, int64_t *bytes_per_rank);
int ccsim_dist_cw_scan(ccsim_engine *e);
int ccsim_dist_cw_decide(ccsim_engine *e);

/* ---- multi-GPU, the persistent level kernel ACROSS the GPUs (csrc/ccsim_persist.h, mailbox form; SURVEY.md 8(e) "if RCCL latency
 * dominates: persistent kernels + P2P-mapped flag/mailbox buffers over xGMI implementing the 8-way max-loc in-kernel") ----
 * Every rank keeps its shard in LDS for the whole batched run; the grid-wide reduce of the one-GPU form is extended over the ranks:
 * the workgroup that completes a rank's local reduction writes the rank's eight words as tagged 8-byte granules into a mailbox
 * on every GPU (one 128-byte store burst per peer per sync over xGMI), every workgroup polls its own GPU's box.  No collective
 * call, no kernel boundary per exchange.  A rank that cannot take part (snapshot not eligible, a peer's box not mappable, a
 * bounded spin that expired) makes ALL ranks fall back to the pass protocol above: nothing is published before every rank agrees.
 *   ccsim_dist_mbox_info     this rank's addressing record (process, device, pointer, IPC handle): allocate the box, CCSIM_MBOX_INFO_BYTES out
 *   ccsim_dist_mbox_connect  all ranks' records, in rank order: map every peer's box (same process: directly / peer access; else IPC);
 *                            zeroes this rank's box and restarts its launch sequence, so NO rank may launch before EVERY rank has
 *                            connected (any agreement step does: ccsim_dist_comm_init all-reduces "connected" behind it).  A box that
 *                            could only be allocated as ordinary (coarse-grained) memory is refused between devices (-ENOTSUP on every rank)
 *   ccsim_dist_mbox_eligible 1 if this rank's shard and pod qualify for the persistent form (collect the minimum over the ranks)
 *   -- then, after ccsim_dist_begin(mode = CCSIM_MODE_BATCHED) on every rank and only if EVERY rank is eligible:
 *   ccsim_dist_mbox_launch   enqueue the persistent launch (asynchronous)
 *   ccsim_dist_mbox_status   wait for it; *ok = 1 if this rank finished the run cleanly (collect the minimum over the ranks)
 *   ccsim_dist_mbox_finish   all_ok = 1: publish the result (then ccsim_dist_finish as usual), returns 0.  all_ok = 0: restore the
 *                            run state of ccsim_dist_begin and return 1: continue with ccsim_dist_scan / _decide from the untouched columns.
 * ccsim_dist_comm_init + ccsim_dist_run do all of this over the engine's RCCL communicator when CCSIM_DIST_MAILBOX=1 (default 0:
 * the form has been validated on one GPU -- virtual ranks inside one grid, two engines and two processes sharing the device --
 * never on a multi-GPU box). */
#define CCSIM_MBOX_INFO_BYTES 96
int ccsim_dist_mbox_info(ccsim_engine *e, uint8_t *info_out /* [CCSIM_MBOX_INFO_BYTES] */);
int ccsim_dist_mbox_connect(ccsim_engine *e, const uint8_t *all_infos /* [n_ranks][CCSIM_MBOX_INFO_BYTES] */, int32_t n_ranks, int32_t rank);
int ccsim_dist_mbox_eligible(ccsim_engine *e);
int ccsim_dist_mbox_launch(ccsim_engine *e);
int ccsim_dist_mbox_status(ccsim_engine *e, int32_t *ok);
int ccsim_dist_mbox_finish(ccsim_engine *e, int32_t all_ok);

/* Topology-coupled plugins on several GPUs (hard PodTopologySpread constraints, InterPodAffinity): the per-domain
 * count / score tables are replicated on every rank, but ccsim_set_pod can only fill them from the rank's own nodes.
 * After ccsim_set_pod on every rank the caller all-reduces each table in place across ranks (table i:
 * ccsim_dist_table -> device pointer, element count, element size 4 / 8, op 0 = SUM / 1 = MAX), then calls
 * ccsim_dist_tables_done on every rank.  No-op for pods without such plugins (count 0). */
int ccsim_dist_table_count(ccsim_engine *e);
int ccsim_dist_table(ccsim_engine *e, int32_t idx, void **ptr, int64_t *len, int32_t *elem_bytes, int32_t *op);
int ccsim_dist_tables_done(ccsim_engine *e);

/* Restore the dynamic node columns (Requested / NonZeroRequested / pod count) to the loaded snapshot,
 * device-to-device from pristine copies kept in HBM: the next ccsim_run starts from the same cluster
 * (what a fresh framework.New + SyncWithClient would give, simulator.go:107-295) without a host upload. */
int ccsim_reset_state(ccsim_engine *e);

/* Page-locked host memory for the result arrays of ccsim_report (per_node_count, log, hist_taintset).  Optional -- any host
 * pointer is accepted there -- but a device-to-host copy into ordinary (pageable) memory is staged by the runtime and, into pages
 * that were never touched, takes a page fault per 4 KiB: at 1M nodes the 4 MB of per-node counts cost more than the simulation
 * itself ("what a step costs around the kernel").  A caller that runs many simulations allocates its result
 * arrays here once and reuses them (cgo: C memory wrapped with unsafe.Slice).  The device is the engine's.  ccsim_host_alloc
 * returns NULL on failure; ccsim_host_free(NULL) is a no-op. */
void *ccsim_host_alloc(ccsim_engine *e, size_t bytes);
void ccsim_host_free(ccsim_engine *e, void *p);

/* Measurement aid (bench.py roofline): time `iters` back-to-back launches of the dominant kernel of
 * `mode` (the full pods x nodes pass: k_scan or k_level_score) with HIP events on the engine's stream;
 * simulation state is not advanced. */
int ccsim_time_scan(ccsim_engine *e, int32_t mode, int32_t iters, int64_t *total_ns, int64_t *bytes_per_scan);

/* Measurement aid: where the last persistent batched launch (csrc/ccsim_persist.h) spent its time.
 * out16: 10 ns ticks of workgroup 0 in [0] level scan + work list, [1] run-down planning (+ the blind batch's apply), [2] ordered commit +
 * re-score, [3] block reduction, [4] grid-wide reduce + barrier, [5] re-score: scores + event prediction + reduce; [6] = level passes;
 * [7] load HBM -> LDS, [8] re-score: normalization maxima (+ reduce), [9] (unused), [10] write-back + FitError diagnosis; [11..15] unused. */
int ccsim_debug_persist_prof(ccsim_engine *e, int64_t *out16);
/* ... and why the windows of the last multi-spec run (ccsim_set_pods, P > 1) ended: out8[r] = windows ended by reason r
 * (0 complete, 1 a normalization maximum was re-derived, 2 Unschedulable, 3 too few holders of a maximum left untouched,
 * 4 the candidate bounds could not prove the choice, 5 every candidate touched and full, 6 touched-node table full, 7 limit). */
int ccsim_debug_multi_stops(ccsim_engine *e, int64_t *out8);
/* ... and how its scans were served (csrc/ccsim_multi.h, the score memo: one resident 32-bit word per (pod spec, node) holding what the
 * scan computes from the node's columns and the spec alone; CCSIM_MULTI_MEMO_MB caps its size, 0 turns it off): out4[0] = 1 if the
 * memo exists, [1] = pods of the last run's windows whose scan READ its memo row, [2] = pods whose scan computed (and filled) it,
 * [3] = bytes of the memo; [4..6] = 10 ns ticks scan workgroup (0, 0) spent issuing its loads + staging the pods' tables, evaluating
 * the pods, merging; [7] = scans.  `out` holds 8 values. */
int ccsim_debug_multi_memo(ccsim_engine *e, int64_t *out8);
/* ... and how the last run of ONE template with topology-coupled plugins (PodTopologySpread, InterPodAffinity) was resolved
 * (csrc/ccsim_coupled.h: windows of placements per node pass): out8[0] = 1 if the pod spec has a windowed plan, [1] = windows
 * of the last run, [2] = 1 if that run fell back to one pass per placement (more classes / plugin inputs than the mode
 * represents), [3] = window length W, [4] = class-list length L, [5] = windows the lane-per-candidate kernel took, [6] = windows its
 * 64-class form took (the others ran the general decide kernel), [7] = placements that kernel resolved in whole rounds (sweeps: CCSIM_CW_SWEEP=0 turns them off); out[8..15] (with CCSIM_CW_PROF=1): 10 ns ticks the deciding
 * wave spent in [8] staging, [9] minima set-up, [10] candidates' verdicts, [11] raw scores, [12] totals + argmax, [13] commit, [14] write-back;
 * [15] = cycles.  `out` holds 16 values.  Knobs (read by ccsim_set_pod): CCSIM_CW=0 disables the mode, CCSIM_CW_WINDOW, CCSIM_CW_LIST. */
int ccsim_debug_coupled(ccsim_engine *e, int64_t *out16);
/* ... and how the last sampled search (percentageOfNodesToScore < 100; S/schedule_one.go:610-723) of a template without topology-coupled
 * plugins ran (csrc/ccsim_sampled.h): out8[0] = 1 if it ran on the resident block summaries, [1] = 1 if a lap of the ring at a time
 * (k_sb_laps; 0: a cycle at a time, k_sb_cycles; 2: a template with a hard spread constraint over zones, csrc/ccsim_sampled_zone.h: [3] = cycles;
 * 3: the FULL search, percentageOfNodesToScore = 100, on the same summaries, csrc/ccsim_search_full.h: [3] = cycles; 4: laps, then -- fewer
 * feasible nodes left than the search keeps: every node visited -- that kernel to the end of the run), [2] = launches of that kernel, [3] = laps evaluated, [4] = stretches re-evaluated node
 * by node under their own normalization maxima, [5] = log2 of the block size, [6] = blocks, [7] = K (numFeasibleNodesToFind);
 * out[8..14] (with CCSIM_SB_PROF=1): 10 ns ticks k_sb_laps spent [8] on the cut blocks (wave 1: the tree, the range queries), [9] on
 * the decision (+ stretches re-evaluated), [10] waiting for the placements, [11] on the winners' leaves (wave 0: the next lap's cuts);
 * [14] the committing wave's own time inside [10].  `out` holds 16 values.
 * Knobs (read when a run begins; every value gives the same results): CCSIM_SB=0 three node passes per cycle, =2 a cycle at a time;
 * CCSIM_SB_CYCLES cycles per launch; CCSIM_SB_SHIFT block size; CCSIM_SB_SLOW_FLOOR nodes below which differing maxima never rebuild;
 * CCSIM_SB_HANDOVER=0 the lap kernel's one-stretch laps to the end of the run instead of handing over to k_sf_cycles;
 * CCSIM_SF=0 the full search as one pass over the nodes per cycle (k_scan_fused) instead of k_sf_cycles; CCSIM_SF_SHIFT=10 blocks of 1024. */
int ccsim_debug_sampled(ccsim_engine *e, int64_t *out16);
/* ... and which form this engine's library-driven sharded runs (ccsim_dist_run) took: out8[0] = 1 if the ranks' mailboxes are connected
 * (ccsim_dist_comm_init under CCSIM_DIST_MAILBOX=1), [1] = the ranks' agreement on the persistent kernel across the GPUs for the current
 * pod spec (-1 not asked yet, 1 go, 0 no: not eligible, or a launch had to be abandoned -- one attempt per pod spec), [2] = launches
 * abandoned so far, [3] = the form of the last run (1 that kernel, 2 the RCCL pass protocol, 3 windows of placements per exchange),
 * [4] = launches of that kernel so far, [5] = ranks of the communicator.  CCSIM_DIST_FORM=passes (read per run, the same on every
 * rank) keeps a run on the pass protocol although the mailboxes are connected. */
int ccsim_debug_dist(ccsim_engine *e, int64_t *out8);

#ifdef __cplusplus
}
#endif
#endif
