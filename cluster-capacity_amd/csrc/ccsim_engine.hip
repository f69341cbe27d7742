// ccsim_engine.hip -- host side of libccsim.so: owns the HBM-resident snapshot, launches the CDNA4
// kernels of ccsim_kernels.h and implements the C ABI declared in include/ccsim.h.
//
// Replaces (reference, kubernetes-sigs/cluster-capacity):
//   pkg/framework/simulator.go:356-381   ClusterCapacity.Run    -> ccsim_run
//   S/schedule_one.go:430-478,967-984    schedulePod + assume   -> ccsim_schedule_one
//   S/backend/cache/cache.go:194-288     UpdateSnapshot         -> ccsim_load_nodes (once; state then lives in HBM)
// There is NO CPU fallback: every entry point fails loudly if HIP does.
#include "ccsim_kernels.h"
#include "ccsim_level.h"
#include "ccsim_persist.h"
#include "ccsim_multi.h"
#include "ccsim_coupled.h"
#include "ccsim_sampled.h"
#include "ccsim_sampled_zone.h"
#include "ccsim_search_full.h"

#include <dlfcn.h>
#include <errno.h>
#include <hip/hip_ext.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <condition_variable>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <time.h>
#include <unistd.h>
#include <vector>

#include "../../include/ccsim.h"

using namespace ccsim;

// the step frame (ccsim_engine::d_frame): [DevState | hist[CCSIM_NREASON + 1] hist_ts[kFrameTs] | PersistSync[kPMaxRanks]]
constexpr int kFrameTs = 256; // taint sets whose FitError bins live in the frame (a pod spec with more gets an array of its own)
constexpr size_t kFrameOffHist = (sizeof(DevState) + 63) & ~(size_t)63;
constexpr size_t kFrameOffSync = (kFrameOffHist + sizeof(unsigned long long) * (CCSIM_NREASON + 1 + kFrameTs) + 63) & ~(size_t)63;
constexpr size_t kFrameHostBytes = kFrameOffSync + sizeof(PersistSync); // what travels: up to the first sync block
constexpr size_t kFrameBytes = kFrameOffSync + sizeof(PersistSync) * kPMaxRanks;

struct ccsim_engine {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int rounds_per_sync = 0;
    int use_graph = 1;
    int time_passes = 0;
    std::vector<hipEvent_t> pass_events; // time_passes: one (start, stop) pair per full-pass launch of a batch
    int pass_events_used = 0;
    int graph_events = 0; // events recorded by the captured graph (re-recorded at every replay)
    double pass_kernel_ms = 0;
    int64_t pass_launches = 0;
    std::string err;

    // snapshot
    bool have_nodes = false, have_profile = false, have_pod = false;
    int64_t n = 0, n_pad = 0, global_offset = 0, n_global = 0;
    int ncol = 3, n_label_cols = 0, n_taintsets = 1;
    std::vector<void *> allocs; // everything hipMalloc'ed
    DevCols cols{};
    uint8_t *d_unsched = nullptr;
    int32_t **d_label_cols = nullptr;
    std::vector<std::vector<int32_t>> h_label_cols; // host copies (ccsim_set_pods derives per-spec tables from them)
    std::vector<int32_t *> dev_label_ptrs;
    std::vector<int32_t> label_col_max; // largest value id per label column
    uint32_t *d_stat = nullptr;
    uint8_t *d_sreason = nullptr;
    const int32_t *d_alloc_pods_real = nullptr; // Allocatable.AllowedPodNumber as loaded (cols.alloc_pods may be a pod's clamped copy)
    bool ports_on = false;                      // one clone per node for the current pod (ccsim_set_pod): NodePorts active, or its own disks conflict
    bool excl_ports = false;                    // ... and which of the two reports a node that holds a clone (host ports: before NodeResourcesFit)
    const uint8_t *d_vol_veto = nullptr;        // ccsim_pod.volume_veto on the device (k_static marks such nodes, k_hist names the plugin); NULL = none
    int32_t *d_ports_eff = nullptr, *d_ports_base = nullptr; // the clamped pod capacity and the pod counts it was built from (k_ports_clamp)

    // pod / profile
    ccsim_profile prof{};
    DevPod pod{};
    DevPts pts{};
    DevIpa ipa{};
    uint64_t *d_ipa_partials = nullptr;
    DevSoft soft{};
    uint64_t *d_soft_partials = nullptr;
    int32_t *d_soft_pc0 = nullptr; // pod_count when the pod with soft constraints was set (DevSoft::pod_count0)
    std::vector<std::pair<int32_t *, size_t>> soft_flags; // epoch flag tables, cleared at the start of every run
    struct DistTable { void *ptr; int64_t len; int32_t elem_bytes, op; };
    std::vector<DistTable> dist_tables;   // replicated tables the caller all-reduces across ranks (ccsim_dist_table)
    std::vector<int32_t *> pts_present;   // per hard constraint: domain-presence flags
    unsigned long long *d_ipa_totals = nullptr; // [3] affinity entries, existing anti-affinity entries, PreScore hits
    int64_t ipa_aff_total0 = 0, ipa_exist_total0 = 0, ipa_entries0 = 0; // initial PreFilter / PreScore totals
    int64_t ipa_aff_total_cur = 0, ipa_exist_total_cur = 0, ipa_entries_cur = 0; // ... after the runs so far
    uint64_t *d_pts_min_partials = nullptr;
    std::vector<std::pair<int32_t *, int32_t *>> pts_tables; // (live, pristine) count tables
    std::vector<size_t> pts_table_len;
    std::vector<std::pair<int64_t *, int64_t *>> ipa_tables; // (live, pristine)
    std::vector<size_t> ipa_table_len;
    std::vector<void *> pod_allocs;

    // run state
    DevState *d_state = nullptr;
    DevState *h_state = nullptr; // pinned
    uint64_t *d_partials = nullptr; // [kMaxGrid][2]
    DevState *d_state2 = nullptr;    // the fused sequential cycle (k_scan_fused) double-buffers state and partials by cycle parity
    uint64_t *d_partials2 = nullptr;
    int fused_allowed = 1;
    uint64_t *d_smp_partials = nullptr; // [kMaxGrid][2] sampled search (percentageOfNodesToScore < 100)
    int64_t *d_smp_prefix = nullptr;    // [kMaxGrid]
    int64_t smp_K = 0;                  // numFeasibleNodesToFind of the current run; 0 = every node is scored
    int64_t smp_start_cur = 0;          // nextStartNodeIndex after the runs so far
    XRec *d_xsend = nullptr, *d_xrecv = nullptr; // distributed exchange (caller's or ours)
    int n_ranks = 0;
    // the engine's own RCCL communicator (ccsim_dist_comm_init) and exchange buffers: ccsim_dist_run drives the whole
    // sharded run -- scan, all-gather, decide -- from C++
    void *rccl_comm = nullptr;
    int comm_ranks = 0, comm_rank = 0;
    XRec *d_own_send = nullptr, *d_own_recv = nullptr;
    int32_t *d_log = nullptr;
    int64_t log_cap = 0;
    unsigned long long *d_hist = nullptr, *d_hist_ts = nullptr, *d_hist_code = nullptr;
    // The STEP FRAME (round 6): run state, FitError histogram (+ per-taint-set bins) and the persistent kernel's sync block are ONE device
    // allocation -- [DevState | hist[CCSIM_NREASON + 1] hist_ts[kFrameTs] | PersistSync[kPMaxRanks]] -- so that a persistent run starts with
    // one host-to-device copy (the state + the zeros of everything behind it; it was one copy and three fills) and ends with one copy back
    // (state + histogram; it was three) next to the per-node counts.  d_state / d_hist / d_psync point into it.
    unsigned char *d_frame = nullptr, *h_frame = nullptr, *h_init = nullptr; // h_frame: page-locked, what comes back (h_state = its start); h_init: page-locked, what goes out
    unsigned long long *d_frame_ts = nullptr;
    bool ts_in_frame = false;      // d_hist_ts is the frame's (the pod's taint sets fit kFrameTs)
    bool frame_sent = false;       // begin_run sent the whole frame for the persistent launch that follows
    int grid = 0;
    int64_t chunk = 0;
    // batched mode (ccsim_level.h): its own launch geometry (chunk bounded by the LDS score cache)
    int lvl_grid = 0;
    int64_t lvl_chunk = 0;
    LevelPartial *d_lpartials = nullptr;
    CommitPartial *d_cpartials = nullptr;
    int32_t *d_cscore = nullptr;
    int64_t *d_blockprefix = nullptr;
    int rank = 0;
    // persistent batched run (ccsim_persist.h)
    PersistSync *d_psync = nullptr; // [kPMaxRanks]: one per virtual rank of a validation run, [0] otherwise
    // mailbox form of the persistent kernel (ccsim_persist.h): this device's box (fine-grained, peer-mapped) and every rank's box as this
    // device addresses it.  Virtual ranks (CCSIM_PERSIST_VRANKS, validation on one GPU): kPMaxRanks boxes in one allocation.
    PersistMailbox *d_mbox = nullptr;
    PersistMailbox *mbox_peers[kPMaxRanks] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    std::vector<void *> mbox_ipc_open;  // peers' boxes opened through IPC handles (other processes)
    bool mbox_ready = false;            // the boxes of all mb_ranks ranks are mapped
    int mb_ranks = 0, mb_rank = 0;      // ranks of the mailbox job / this engine's rank
    uint32_t mb_seq = 0;                // launches of the mailbox form so far: the same on every rank (they launch together or not at all)
    int mb_k = 0;                       // K of the launch in flight
    int dist_last_form = 0;             // ccsim_dist_run: what the last run took (1 the persistent kernel across the GPUs, 2 the RCCL pass protocol, 3 windows of placements per exchange)
    int mb_launches = 0;                // ccsim_dist_run: launches of the persistent kernel across the GPUs
    int mb_abandoned = 0;               // ccsim_dist_run: mailbox launches that ended in the fall-back (each sets mb_go = 0)
    int mb_go = -1;                     // ccsim_dist_run: did every rank call this pod / snapshot eligible? (-1: not agreed yet; reset by set_pod / load_nodes / comm_init -- SPMD: on every rank alike)
    DevState mb_state0{};               // the run state ccsim_dist_begin uploaded (restored when the ranks fall back to the pass protocol)
    int32_t *d_mb_ok = nullptr;         // device flag of the launch in flight: 1 = this rank finished cleanly (ccsim_persist.h ok_flag)
    int32_t *h_mb_ok = nullptr;         // ... and the ranks' minimum, page-locked
    uint32_t persist_seq = 0;           // launch sequence of the mailbox form (tags: nothing is zeroed between launches)
    int persist_vranks = 0;             // CCSIM_PERSIST_VRANKS: run the single-device batched mode as that many virtual ranks
    bool lazy_wide = true;              // (CCSIM_EAGER_WIDE=1: the persistent launch writes the int64 columns back itself)
    bool wide_stale = false;            // a persistent launch left the int64 request columns behind the mirrors (PersistCols::skip_wide): ensure_cols re-derives them
    bool reset_pending = false;         // ccsim_reset_state deferred the restore of the node columns: the next persistent launch loads the
                                        // pristine copies directly; everything else restores first (ensure_cols)
    bool mbox_coarse = false;           // the mailbox could only be allocated as ordinary (coarse-grained) device memory: usable by virtual ranks and
                                        // engines of the same device only
    bool extras_dirty = false;          // a run of a pod with extra resource columns (ephemeral-storage / scalar: pod.nx != 0) or of several pod specs may have
                                        // moved req[2 .. ncol) since the last full restore: a lazy reset consumed by the persistent launch (which loads
                                        // only cpu / memory / pod counts) copies those columns back as well
    bool hist_in_kernel = false;        // the last persistent launch filled d_hist / d_hist_ts itself (no k_hist pass)
    bool persist_hint = false;          // persist_hint_mt / _ma: the normalization maxima the last persistent launch of this pod started with
    int32_t persist_hint_mt = 0, persist_hint_ma = 0;
    // ... and its results left for the host right behind the kernel, before the host knew how the launch ended (one stream sync per run):
    bool early_counts = false, early_hist = false, early_hist_framed = false;
    int early_narrow = 0;            // the counts that left early are ccsim_report.per_node_count_narrow's, in elements of that many bytes
    void *d_cnt_narrow = nullptr;    // [n_pad] 2-byte elements' worth (allocated on first use; goes with the snapshot)
    unsigned long long *h_hist_pin = nullptr; // page-locked [CCSIM_NREASON + 1 + h_hist_ts_cap]
    size_t h_hist_ts_cap = 0;
    int n_cus = 0;
    int persist_allowed = 1;
    int64_t node_max_podcount = 0; // largest len(NodeInfo.Pods) of the snapshot
    int persist_run = 0; // K of the current batched run's persistent launch, 0 = multi-kernel path
    mutable int persist_per_cu[2][9] = {{-1, -1, -1, -1, -1, -1, -1, -1, -1}, {-1, -1, -1, -1, -1, -1, -1, -1, -1}}; // resident workgroups of k_level_persist<K, MB> per CU on THIS engine's device
    bool cw_attr_shard_set = false;
    bool cw_attr_set = false; // hipFuncSetAttribute(MaxDynamicSharedMemorySize) done for the coupled decide kernels on THIS engine's device
    // several pod specs cycled round-robin (ccsim_multi.h)
    bool multi = false;
    int n_pods = 0, n_cls = 0, m_blocks = 0, multi_window = kMWindowMax, max_taintsets = 1;
    std::vector<void *> multi_allocs;
    std::vector<MPod> h_mpods;
    std::vector<int> cls_taintsets; // taint sets of each static class (the FitError histogram of the failing pod)
    MPod *d_mpods = nullptr;
    MState *d_mstate = nullptr, *h_mstate = nullptr;
    uint32_t *d_stat_cls = nullptr;
    uint8_t *d_sreason_cls = nullptr, *d_present_pool = nullptr, *d_inc_pool = nullptr;
    int32_t *d_tbl_pool = nullptr, *d_tbl_pool0 = nullptr, *d_per_spec = nullptr;
    size_t tbl_len = 0, anti_words = 0;
    uint32_t *d_anti_bits = nullptr, *d_anti_bits0 = nullptr;
    MPartial *d_mpartials = nullptr;
    MCand *d_mcands = nullptr;
    uint16_t *d_memo = nullptr;     // the score memo [n_pods][n_pad], 16-bit words (ccsim_multi.h), nullptr = off
    int32_t *d_memo_stamp = nullptr, *d_mtouched = nullptr, *d_mvsync = nullptr;
    const int32_t *tsc_label[kMTsc] = {nullptr, nullptr};
    DevPod multi_prof{};
    int32_t multi_next = 0; // spec of the next cycle (continues across runs; ccsim_reset_state rewinds it)
    // windowed mode for one template with topology-coupled plugins (ccsim_coupled.h)
    std::vector<int> pts_col, soft_col, ipa_col; // label column of every hard / soft constraint / inter-pod affinity key
    std::vector<int32_t *> soft_present;         // (unused by the kernels; keeps the soft pass symmetrical with the hard one)
    bool cw_ok = false;                          // the current pod spec has a windowed plan
    std::string cw_why;                          // ... or why not
    CwPlan cw_plan{};
    CwWork cw_work{};
    void *d_cw_args = nullptr;                   // CwDecideArgs of the current run, in device memory
    void *cw_zero_base = nullptr;                // class table + control words: one allocation, zeroed when a run starts
    size_t cw_zero_bytes = 0;
    int cw_allowed = 1;
    bool cw_run = false;                         // the current run takes the windowed path
    bool cw_shard_run = false, cw_shard_args = false; // ... on node-range shards (ccsim_dist_cw_*); its argument block is uploaded
    // the sampled search on resident block summaries (ccsim_sampled.h): buffers of the node count's lifetime, made on first use
    int32_t *d_sb_memo = nullptr;
    uint8_t *d_sb_flag8 = nullptr;
    uint32_t *d_sb_fc = nullptr, *d_sb_mx = nullptr;
    unsigned long long *d_sb_key = nullptr;
    int sb_shift = 0, sb_blocks = 0;
    int sb_allowed = 1;                          // CCSIM_SB=0: the three-pass cycle, 2: one cycle at a time on the summaries (A/B and test knobs)
    bool sb_attr_set = false, sb_run = false;
    unsigned long long *d_sb_prof = nullptr;     // CCSIM_SB_PROF=1: k_sb_laps' phase ticks
    // ... of one template with a hard spread constraint over <= 64 zones (+ anti-affinity over a unique-per-node key): ccsim_sampled_zone.h
    uint8_t *d_sz_zone8 = nullptr, *d_sz_ent_flg = nullptr, *d_sz_cntz = nullptr;
    unsigned long long *d_sz_ent_key = nullptr, *d_sz_present = nullptr;
    uint32_t *d_sz_over = nullptr;
    bool sz_run = false, sz_attr_set = false;
    std::string sz_why;                          // why the last sampled run of a coupled template did not take that form
    bool sz_built = false;                       // this run's first k_sz_build has been looked at (a (block, zone) count beyond a byte: the three-pass cycle instead)
    bool sf_handover = false;                    // this run: k_sb_laps has handed over to k_sf_cycles (every node is visited from here on: DevState::smp_phase == 2)
    bool sf_run = false;                         // this run: the FULL search (every node scored) of an uncoupled template on the same summaries (k_sf_cycles)
    bool sb_laps = false;                        // this run: a lap of the ring at a time (k_sb_laps) instead of a cycle at a time (k_sb_cycles)
    bool cw_fast = false;                        // ... and may use the lane-per-candidate decide kernel (k_cw_decide_fast)
    // narrow mirrors (DevCols::narrow): facts about the loaded snapshot, gathered on the host at load time
    int32_t *d_a32[2] = {nullptr, nullptr};
    int dist_pass_in_window = 0;      // passes since the last ccsim_dist_begin / ccsim_dist_poll
    bool dist_score_launched = true;  // whether the current sharded pass launched k_level_score
    bool rows_active = false; // the current batched run keeps its dynamic state in the commit rows (DevCols::rows)
    uint64_t node_mem_or = 0; // OR of every memory value of the snapshot (common power-of-two unit)
    int64_t node_max_cpu = 0, node_max_mem = 0, node_max_pods = 0;
    int narrow_allowed = 1;
    // pristine copies of the dynamic columns (ccsim_reset_state)
    std::vector<std::pair<void *, void *>> backups; // (live, pristine)
    std::vector<size_t> backup_bytes;
    bool begun = false;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipStream_t copy_stream = nullptr;  // the large result copies of fill_report run beside the FitError diagnosis (created on first use)
    hipEvent_t ev_copy = nullptr;
    double kernel_ms = 0;
    int64_t limit = 0;
    int mode = 0;

    // graph replay of rounds_per_sync x (k_scan, k_final)
    hipGraph_t graph = nullptr;
    hipGraphExec_t graph_exec = nullptr;
    int graph_rounds = 0;
    int graph_mode = -1;
};

static int fail(ccsim_engine *e, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (e) e->err = buf;
    return code;
}

#define HIPCHK(e, call)                                                                                           \
    do {                                                                                                          \
        hipError_t _r = (call);                                                                                   \
        if (_r != hipSuccess) return fail((e), -EIO, "%s failed: %s (%s:%d)", #call, hipGetErrorString(_r), __FILE__, __LINE__); \
    } while (0)

template <typename T>
static int dev_alloc(ccsim_engine *e, T **out, size_t count, std::vector<void *> &track, bool zero = true) {
    void *p = nullptr;
    size_t bytes = (count ? count : 1) * sizeof(T);
    HIPCHK(e, hipMalloc(&p, bytes));
    track.push_back(p);
    if (zero) HIPCHK(e, hipMemsetAsync(p, 0, bytes, e->stream));
    *out = (T *)p;
    return 0;
}

struct ccsim_engine;
static int build_narrow(ccsim_engine *e);
static int cw_make_plan(ccsim_engine *e);
static bool label_col_unique(const ccsim_engine *e, int col);
static int persist_k(const ccsim_engine *e);
static int ensure_cols(ccsim_engine *e);

template <typename T>
static int upload(ccsim_engine *e, T **out, const T *src, size_t count, size_t padded, std::vector<void *> &track) {
    int rc = dev_alloc(e, out, padded, track, true);
    if (rc) return rc;
    if (src && count) HIPCHK(e, hipMemcpyAsync(*out, src, count * sizeof(T), hipMemcpyHostToDevice, e->stream));
    return 0;
}

static void free_list(std::vector<void *> &v) {
    for (void *p : v) (void)hipFree(p);
    v.clear();
}

static void drop_graph(ccsim_engine *e) {
    if (e->graph_exec) (void)hipGraphExecDestroy(e->graph_exec);
    if (e->graph) (void)hipGraphDestroy(e->graph);
    e->graph_exec = nullptr;
    e->graph = nullptr;
    e->graph_rounds = 0;
    e->graph_mode = -1;
}

static void dist_comm_release(ccsim_engine *e);

extern "C" int32_t ccsim_abi_version(void) { return CCSIM_ABI_VERSION; }

extern "C" const char *ccsim_last_error(const ccsim_engine *e) { return e ? e->err.c_str() : "null engine"; }

extern "C" int ccsim_create(const ccsim_config *cfg, ccsim_engine **out) {
    if (!cfg || !out) return -EINVAL;
    if (cfg->abi_version != CCSIM_ABI_VERSION) return -EINVAL;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || cfg->device < 0 || cfg->device >= ndev) return -EIO;
    ccsim_engine *e = new ccsim_engine();
    e->device = cfg->device;
    e->rounds_per_sync = cfg->rounds_per_sync;
    e->use_graph = cfg->use_graph;
    e->time_passes = cfg->time_passes;
    if (const char *f = getenv("CCSIM_NARROW")) e->narrow_allowed = atoi(f); // A/B knob
    if (const char *f = getenv("CCSIM_PERSIST")) e->persist_allowed = atoi(f); // A/B knob: 0 = multi-kernel batched mode
    if (const char *f = getenv("CCSIM_EAGER_WIDE")) e->lazy_wide = atoi(f) == 0;
    if (const char *f = getenv("CCSIM_CW")) e->cw_allowed = atoi(f);           // A/B knob: 0 = coupled plugins one pass per placement
    if (const char *f = getenv("CCSIM_SB")) e->sb_allowed = atoi(f);           // A/B knob: 0 = the sampled search as three node passes per cycle
    if (const char *f = getenv("CCSIM_FUSED")) e->fused_allowed = atoi(f);     // A/B knob: 0 = sequential cycle as k_scan + k_final
    if (const char *f = getenv("CCSIM_PERSIST_VRANKS")) e->persist_vranks = atoi(f) > 1 && atoi(f) <= kPMaxRanks ? atoi(f) : 0; // validation knob
    if (hipSetDevice(e->device) != hipSuccess) {
        delete e;
        return -EIO;
    }
    if (cfg->stream) {
        e->stream = (hipStream_t)cfg->stream;
    } else {
        if (hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking) != hipSuccess) {
            delete e;
            return -EIO;
        }
        e->own_stream = true;
    }
    if (hipEventCreate(&e->ev0) != hipSuccess || hipEventCreate(&e->ev1) != hipSuccess ||
        hipHostMalloc((void **)&e->h_frame, kFrameHostBytes, hipHostMallocDefault) != hipSuccess ||
        hipHostMalloc((void **)&e->h_init, kFrameHostBytes, hipHostMallocDefault) != hipSuccess ||
        hipMalloc((void **)&e->d_frame, kFrameBytes) != hipSuccess ||
        hipMalloc((void **)&e->d_partials, sizeof(uint64_t) * 2 * kMaxGrid) != hipSuccess ||
        hipMalloc((void **)&e->d_state2, sizeof(DevState)) != hipSuccess ||
        hipMalloc((void **)&e->d_partials2, sizeof(uint64_t) * 2 * kMaxGrid) != hipSuccess ||
        hipMalloc((void **)&e->d_smp_partials, sizeof(uint64_t) * 2 * kMaxGrid) != hipSuccess ||
        hipMalloc((void **)&e->d_smp_prefix, sizeof(int64_t) * kMaxGrid) != hipSuccess) {
        ccsim_destroy(e);
        return -ENOMEM;
    }
    memset(e->h_frame, 0, kFrameHostBytes), memset(e->h_init, 0, kFrameHostBytes);
    e->h_state = reinterpret_cast<DevState *>(e->h_frame);
    e->d_state = reinterpret_cast<DevState *>(e->d_frame);
    e->d_hist = reinterpret_cast<unsigned long long *>(e->d_frame + kFrameOffHist);
    e->d_frame_ts = e->d_hist + CCSIM_NREASON + 1;
    e->d_psync = reinterpret_cast<PersistSync *>(e->d_frame + kFrameOffSync);
    if (hipMemset(e->d_frame, 0, kFrameBytes) != hipSuccess) {
        ccsim_destroy(e);
        return -EIO;
    }
    {
        hipDeviceProp_t prop;
        e->n_cus = hipGetDeviceProperties(&prop, e->device) == hipSuccess ? prop.multiProcessorCount : 0;
    }
    e->d_hist_code = e->d_hist + CCSIM_NREASON;
    *out = e;
    return 0;
}

extern "C" void ccsim_destroy(ccsim_engine *e) {
    if (!e) return;
    (void)hipSetDevice(e->device);
    if (e->stream) (void)hipStreamSynchronize(e->stream);
    drop_graph(e);
    free_list(e->allocs);
    e->d_cnt_narrow = nullptr;
    free_list(e->pod_allocs);
    free_list(e->multi_allocs);
    if (e->h_mstate) (void)hipHostFree(e->h_mstate);
    if (e->d_frame) (void)hipFree(e->d_frame); // (d_state, d_hist, d_psync)
    if (e->d_partials) (void)hipFree(e->d_partials);
    if (e->d_state2) (void)hipFree(e->d_state2);
    if (e->d_partials2) (void)hipFree(e->d_partials2);
    if (e->d_smp_partials) (void)hipFree(e->d_smp_partials);
    if (e->d_smp_prefix) (void)hipFree(e->d_smp_prefix);
    if (e->d_log) (void)hipFree(e->d_log);
    if (e->h_frame) (void)hipHostFree(e->h_frame); // (h_state)
    if (e->h_init) (void)hipHostFree(e->h_init);
    if (e->h_hist_pin) (void)hipHostFree(e->h_hist_pin);
    for (void *p : e->mbox_ipc_open) (void)hipIpcCloseMemHandle(p);
    if (e->d_mbox) (void)hipFree(e->d_mbox);
    if (e->d_mb_ok) (void)hipFree(e->d_mb_ok);
    if (e->h_mb_ok) (void)hipHostFree(e->h_mb_ok);
    dist_comm_release(e);
    for (hipEvent_t ev : e->pass_events) (void)hipEventDestroy(ev);
    if (e->ev0) (void)hipEventDestroy(e->ev0);
    if (e->ev_copy) (void)hipEventDestroy(e->ev_copy);
    if (e->copy_stream) (void)hipStreamDestroy(e->copy_stream);
    if (e->ev1) (void)hipEventDestroy(e->ev1);
    if (e->own_stream && e->stream) (void)hipStreamDestroy(e->stream);
    delete e;
}

extern "C" int ccsim_load_nodes(ccsim_engine *e, const ccsim_nodes *nd) {
    if (!e || !nd) return -EINVAL;
    if (nd->n_nodes < 0 || nd->n_scalar < 0 || nd->n_scalar > CCSIM_MAX_SCALAR || nd->n_label_cols < 0 ||
        nd->n_label_cols > CCSIM_MAX_LABEL_COLS)
        return fail(e, -EINVAL, "bad node snapshot dimensions");
    if (nd->n_nodes > 0 && (!nd->alloc_pods || !nd->pod_count || !nd->nz_mcpu || !nd->nz_mem))
        return fail(e, -EINVAL, "alloc_pods, pod_count, nz_mcpu and nz_mem are required");
    if (nd->global_offset + nd->n_nodes > (1ll << kIdxBits) - 1) return fail(e, -EINVAL, "too many nodes");
    HIPCHK(e, hipSetDevice(e->device));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    drop_graph(e);
    free_list(e->allocs);
    e->d_cnt_narrow = nullptr;
    e->d_sb_memo = nullptr, e->d_sb_flag8 = nullptr, e->d_sz_zone8 = e->d_sz_ent_flg = e->d_sz_cntz = nullptr, e->d_sz_ent_key = e->d_sz_present = nullptr, e->d_sz_over = nullptr, e->d_sb_fc = e->d_sb_mx = nullptr, e->d_sb_key = nullptr, e->d_sb_prof = nullptr; // (they lived in e->allocs)
    e->d_soft_pc0 = nullptr; // (sized for the previous snapshot: the pod is set again after a load)
    e->backups.clear();
    e->reset_pending = false, e->wide_stale = false;
    e->mb_go = -1;
    e->backup_bytes.clear();
    e->have_nodes = e->have_pod = e->begun = false;
    e->multi = false;
    free_list(e->multi_allocs);
    e->cols.narrow = 0;
    e->n = nd->n_nodes;
    e->n_pad = ((e->n + kTile - 1) / kTile) * kTile;
    if (e->n_pad == 0) e->n_pad = kTile;
    e->global_offset = nd->global_offset;
    e->n_global = nd->n_global > 0 ? nd->n_global : nd->n_nodes;
    e->ncol = 3 + nd->n_scalar;
    e->n_label_cols = nd->n_label_cols;
    DevCols c{};
    c.n = e->n;
    c.n_pad = e->n_pad;
    c.global_offset = e->global_offset;
    const size_t n = (size_t)e->n, np = (size_t)e->n_pad;
    int rc;
    for (int col = 0; col < e->ncol; col++) {
        int64_t *a = nullptr, *r = nullptr;
        if ((rc = upload(e, &a, nd->alloc[col], n, np, e->allocs))) return rc;
        if ((rc = upload(e, &r, nd->req[col], n, np, e->allocs))) return rc;
        c.alloc[col] = a;
        c.req[col] = r;
    }
    int32_t *ap = nullptr, *pc = nullptr, *ts = nullptr, *plc = nullptr;
    int64_t *z0 = nullptr, *z1 = nullptr;
    if ((rc = upload(e, &ap, nd->alloc_pods, n, np, e->allocs))) return rc;
    if ((rc = upload(e, &pc, nd->pod_count, n, np, e->allocs))) return rc;
    if ((rc = upload(e, &z0, nd->nz_mcpu, n, np, e->allocs))) return rc;
    if ((rc = upload(e, &z1, nd->nz_mem, n, np, e->allocs))) return rc;
    if ((rc = upload(e, &ts, nd->taintset_id, n, np, e->allocs))) return rc;
    if ((rc = upload(e, &e->d_unsched, nd->unschedulable, n, np, e->allocs))) return rc;
    if ((rc = dev_alloc(e, &plc, np, e->allocs))) return rc;
    if ((rc = dev_alloc(e, &e->d_stat, np, e->allocs))) return rc;
    if ((rc = dev_alloc(e, &e->d_sreason, np, e->allocs))) return rc;
    std::vector<int32_t *> lc((size_t)CCSIM_MAX_LABEL_COLS, nullptr);
    e->h_label_cols.assign((size_t)nd->n_label_cols, {});
    for (int k = 0; k < nd->n_label_cols; k++) {
        if ((rc = upload(e, &lc[k], nd->label_cols[k], n, np, e->allocs))) return rc;
        if (nd->label_cols[k]) e->h_label_cols[(size_t)k].assign(nd->label_cols[k], nd->label_cols[k] + n);
        else e->h_label_cols[(size_t)k].assign(n, 0);
    }
    e->dev_label_ptrs = lc;
    e->label_col_max.assign((size_t)nd->n_label_cols, 0);
    for (int k = 0; k < nd->n_label_cols; k++)
        for (int32_t v : e->h_label_cols[(size_t)k]) e->label_col_max[(size_t)k] = v > e->label_col_max[(size_t)k] ? v : e->label_col_max[(size_t)k];
    if ((rc = dev_alloc(e, &e->d_label_cols, (size_t)CCSIM_MAX_LABEL_COLS, e->allocs))) return rc;
    HIPCHK(e, hipMemcpyAsync(e->d_label_cols, lc.data(), sizeof(int32_t *) * CCSIM_MAX_LABEL_COLS, hipMemcpyHostToDevice,
                             e->stream));
    c.alloc_pods = ap;
    e->d_alloc_pods_real = ap, e->ports_on = false;
    c.pod_count = pc;
    c.nz_mcpu = z0;
    c.nz_mem = z1;
    c.taintset_id = ts;
    c.placed_cnt = plc;
    c.stat = e->d_stat;
    c.sreason = e->d_sreason;
    // narrow mirrors: storage now, content at ccsim_set_pod (the unit also depends on the pod's requests)
    {
        int32_t *m[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
        for (int k = 0; k < 6; k++)
            if ((rc = dev_alloc(e, &m[k], np, e->allocs))) return rc;
        e->d_a32[0] = m[0], e->d_a32[1] = m[1];
        if ((rc = dev_alloc(e, &c.rows, np * kRowWords, e->allocs))) return rc;
        c.a32[0] = m[0], c.a32[1] = m[1], c.r32[0] = m[2], c.r32[1] = m[3], c.z32[0] = m[4], c.z32[1] = m[5];
        c.narrow = 0, c.mem_shift = 0;
        e->node_mem_or = 0, e->node_max_cpu = e->node_max_mem = e->node_max_pods = e->node_max_podcount = 0;
        for (size_t i = 0; i < n; i++) {
            const int64_t vc[3] = {nd->alloc[0] ? nd->alloc[0][i] : 0, nd->req[0] ? nd->req[0][i] : 0, nd->nz_mcpu[i]};
            const int64_t vm[3] = {nd->alloc[1] ? nd->alloc[1][i] : 0, nd->req[1] ? nd->req[1][i] : 0, nd->nz_mem[i]};
            for (int k = 0; k < 3; k++) {
                if (vc[k] < 0 || vm[k] < 0) e->node_max_cpu = INT64_MAX; // negative values: never narrow
                e->node_max_cpu = vc[k] > e->node_max_cpu ? vc[k] : e->node_max_cpu;
                e->node_max_mem = vm[k] > e->node_max_mem ? vm[k] : e->node_max_mem;
                e->node_mem_or |= (uint64_t)vm[k];
            }
            e->node_max_pods = nd->alloc_pods[i] > e->node_max_pods ? nd->alloc_pods[i] : e->node_max_pods;
            e->node_max_podcount = nd->pod_count[i] > e->node_max_podcount ? nd->pod_count[i] : e->node_max_podcount;
            if (nd->alloc_pods[i] < 0 || nd->pod_count[i] < 0) e->node_max_podcount = INT64_MAX; // never packed into 16 bits
        }
    }
    e->cols = c;
    // pristine copies of everything a placement mutates (NodeInfo.Requested / NonZeroRequested / len(Pods))
    {
        auto keep = [&](void *live, size_t bytes) -> int {
            void *p = nullptr;
            HIPCHK(e, hipMalloc(&p, bytes));
            e->allocs.push_back(p);
            HIPCHK(e, hipMemcpyAsync(p, live, bytes, hipMemcpyDeviceToDevice, e->stream));
            e->backups.emplace_back(live, p);
            e->backup_bytes.push_back(bytes);
            return 0;
        };
        for (int col = 0; col < e->ncol; col++)
            if ((rc = keep(c.req[col], np * 8))) return rc;
        if ((rc = keep(c.nz_mcpu, np * 8))) return rc;
        if ((rc = keep(c.nz_mem, np * 8))) return rc;
        if ((rc = keep(c.pod_count, np * 4))) return rc;
    }
    HIPCHK(e, hipStreamSynchronize(e->stream)); // lc / caller arrays may go away after return
    // launch geometry: one contiguous chunk of nodes per block, <= kMaxGrid blocks
    int64_t tiles = e->n_pad / kTile;
    int64_t gcap = 1024; // 4 blocks per CU
    if (const char *g = getenv("CCSIM_SCAN_GRID")) gcap = atoll(g) > 0 && atoll(g) <= kMaxGrid ? atoll(g) : gcap; // tuning knob
    e->grid = (int)(tiles < gcap ? tiles : gcap);
    e->chunk = ((tiles + e->grid - 1) / e->grid) * kTile;
    e->grid = (int)((e->n_pad + e->chunk - 1) / e->chunk);
    // batched mode: blocks of one contiguous chunk each; default target = 3 resident blocks per CU x 256 CUs
    int64_t target = 768;
    if (const char *g = getenv("CCSIM_LEVEL_GRID")) target = atoll(g) > 0 ? atoll(g) : target; // tuning knob
    int64_t ltiles = (tiles + target - 1) / target;
    if (ltiles < 1) ltiles = 1;
    e->lvl_chunk = ltiles * kTile;
    e->lvl_grid = (int)((e->n_pad + e->lvl_chunk - 1) / e->lvl_chunk);
    if ((rc = dev_alloc(e, &e->d_lpartials, (size_t)e->grid, e->allocs))) return rc;     // score pass: k_scan's geometry
    if ((rc = dev_alloc(e, &e->d_cpartials, (size_t)e->lvl_grid, e->allocs))) return rc; // commit pass
    if ((rc = dev_alloc(e, &e->d_blockprefix, (size_t)e->lvl_grid, e->allocs))) return rc;
    if ((rc = dev_alloc(e, &e->d_cscore, np, e->allocs))) return rc;
    HIPCHK(e, hipStreamSynchronize(e->stream));
    e->have_nodes = true;
    return 0;
}

extern "C" int ccsim_set_profile(ccsim_engine *e, const ccsim_profile *p) {
    if (!e || !p) return -EINVAL;
    if (p->percentage_of_nodes_to_score < 0 || p->percentage_of_nodes_to_score > 100)
        return fail(e, -EINVAL, "percentageOfNodesToScore out of [0,100]");
    if (p->filter_mask & CCSIM_F_TOPOLOGYSPREAD || p->w_topologyspread)
        ; // accepted: PodTopologySpread is a no-op (PreFilter/PreScore Skip) for pods without constraints
    if (p->n_fit_res < 0 || p->n_fit_res > CCSIM_MAX_RES || p->n_bal_res < 0 || p->n_bal_res > CCSIM_MAX_RES)
        return fail(e, -EINVAL, "bad resource list");
    for (int i = 0; i < p->n_fit_res; i++) {
        if (p->fit_res[i] < 0 || p->fit_res[i] >= CCSIM_MAX_RES) return fail(e, -EINVAL, "LeastAllocated resource column out of range");
        if (p->fit_res_w[i] < 1 || p->fit_res_w[i] > 100) return fail(e, -EINVAL, "resource weight out of [1,100]");
    }
    for (int i = 0; i < p->n_bal_res; i++)
        if (p->bal_res[i] < 0 || p->bal_res[i] >= CCSIM_MAX_RES) return fail(e, -EINVAL, "BalancedAllocation resource column out of range");
    for (int32_t w : {p->w_taint, p->w_nodeaffinity, p->w_fit, p->w_balanced, p->w_topologyspread, p->w_interpodaffinity, p->w_imagelocality})
        if (w < 0 || w > 1000000) return fail(e, -EINVAL, "plugin weight out of [0, 1000000]");
    e->prof = *p;
    e->have_profile = true;
    e->have_pod = false;
    e->multi = false;
    drop_graph(e);
    return 0;
}

// The static (pod-spec dependent, state independent) kernel for one pod spec: uploads the taint-set / requirement tables,
// runs k_static into `stat` / `sreason` ([n_pad] each).  Shared by ccsim_set_pod and ccsim_set_pods (one run per class).
static int static_pass(ccsim_engine *e, const ccsim_pod *pod, bool score_preferred, uint32_t *stat, uint8_t *sreason, std::vector<void *> &track) {
    const ccsim_profile &pf = e->prof;
    StaticArgs s{};
    s.n = e->n;
    s.n_pad = e->n_pad;
    s.filter_mask = pf.filter_mask;
    s.unschedulable = e->d_unsched;
    s.taintset_id = e->cols.taintset_id;
    s.tolerates_unschedulable = pod->tolerates_unschedulable;
    s.affinity_filter_active = pod->affinity_filter_active;
    s.has_node_selector = pod->has_node_selector;
    s.has_required_terms = pod->has_required_terms;
    s.n_required = pod->n_required;
    s.n_preferred = score_preferred ? pod->n_preferred : 0;
    s.node_selector = DevTerm{pod->node_selector.first_req, pod->node_selector.n_req, 0};
    s.label_cols = e->d_label_cols;
    s.stat = stat;
    s.sreason = sreason;
    int rc;
    if ((pf.filter_mask & CCSIM_F_NODEPORTS) && pod->has_host_ports && pod->host_ports_conflict) {
        uint8_t *d_pc = nullptr;
        if ((rc = upload(e, &d_pc, pod->host_ports_conflict, (size_t)e->n, (size_t)e->n_pad, track))) return rc;
        s.ports_conflict = d_pc;
    }
    if (pod->volume_veto) {
        uint8_t *d_vv = nullptr;
        if ((rc = upload(e, &d_vv, pod->volume_veto, (size_t)e->n, (size_t)e->n_pad, track))) return rc;
        s.volume_veto = d_vv;
        e->d_vol_veto = d_vv; // (lives as long as the pod: `track` is the pod's allocation list whenever a pod carries one)
    }
    if (pf.w_imagelocality && pod->image_score) {
        uint8_t *d_img = nullptr;
        if ((rc = upload(e, &d_img, pod->image_score, (size_t)e->n, (size_t)e->n_pad, track))) return rc;
        s.image_score = d_img;
    }
    uint8_t *d_ok = nullptr;
    int32_t *d_cnt = nullptr;
    if ((rc = upload(e, &d_ok, pod->taint_filter_ok, (size_t)pod->n_taintsets, (size_t)pod->n_taintsets, track))) return rc;
    std::vector<int32_t> cnt(pod->taint_prefer_cnt, pod->taint_prefer_cnt + pod->n_taintsets);
    if (!pf.w_taint) std::fill(cnt.begin(), cnt.end(), 0); // plugin disabled: keep the normalization constant fixed
    if ((rc = upload(e, &d_cnt, cnt.data(), cnt.size(), cnt.size(), track))) return rc;
    s.taint_filter_ok = d_ok;
    s.taint_prefer_cnt = d_cnt;
    std::vector<DevTerm> rq, pr;
    for (int t = 0; t < pod->n_required; t++) rq.push_back(DevTerm{pod->required[t].first_req, pod->required[t].n_req, 0});
    for (int t = 0; t < pod->n_preferred; t++)
        pr.push_back(DevTerm{pod->preferred[t].first_req, pod->preferred[t].n_req, pod->preferred[t].weight});
    std::vector<DevReq> reqs;
    for (int i = 0; i < pod->n_reqs; i++) {
        if (pod->reqs[i].col < 0 || pod->reqs[i].col >= e->n_label_cols) return fail(e, -EINVAL, "requirement column out of range");
        reqs.push_back(DevReq{pod->reqs[i].col, pod->reqs[i].table_off});
    }
    DevTerm *d_rq = nullptr, *d_pr = nullptr;
    DevReq *d_reqs = nullptr;
    uint8_t *d_tab = nullptr;
    if ((rc = upload(e, &d_rq, rq.data(), rq.size(), rq.size(), track))) return rc;
    if ((rc = upload(e, &d_pr, pr.data(), pr.size(), pr.size(), track))) return rc;
    if ((rc = upload(e, &d_reqs, reqs.data(), reqs.size(), reqs.size(), track))) return rc;
    if ((rc = upload(e, &d_tab, pod->req_tables, (size_t)pod->req_tables_len, (size_t)pod->req_tables_len, track))) return rc;
    s.required = d_rq;
    s.preferred = d_pr;
    s.reqs = d_reqs;
    s.req_tables = d_tab;
    const int blocks = (int)((e->n_pad + kThreads - 1) / kThreads);
    hipLaunchKernelGGL(k_static, dim3(blocks), dim3(kThreads), 0, e->stream, s);
    HIPCHK(e, hipGetLastError());
    HIPCHK(e, hipStreamSynchronize(e->stream)); // host vectors above go out of scope
    return 0;
}

// pod + profile constants of the scan kernels (fit.go:224-233, resource_allocation.go:118-148, node_affinity.go:243-246,
// balanced_allocation.go:66-79)
static DevPod make_devpod(const ccsim_engine *e, const ccsim_pod *pod) {
    const ccsim_profile &pf = e->prof;
    DevPod p{};
    for (int c = 0; c < CCSIM_MAX_RES; c++) p.req[c] = pod->req[c];
    p.nz_mcpu = pod->nz_mcpu;
    p.nz_mem = pod->nz_mem;
    p.ncol = e->ncol;
    p.fit_enabled = (pf.filter_mask & CCSIM_F_FIT) ? 1 : 0;
    p.all_zero_req = (pod->req[0] == 0 && pod->req[1] == 0 && pod->req[2] == 0 && !pod->has_scalar_entries) ? 1 : 0;
    // resource lists beyond cpu / memory: the general evaluation.  A scalar column the pod does not request is bypassed
    // (resource_allocation.go:97-99); ephemeral-storage always takes part (:105-106), so it needs a slot even without a request
    bool list_eph = false;
    for (int i = 0; i < pf.n_fit_res; i++) {
        p.gen_score |= pf.fit_res[i] >= 2;
        list_eph |= pf.fit_res[i] == 2 && pf.w_fit;
        p.fit_col[p.n_fit] = pf.fit_res[i], p.fit_w[p.n_fit++] = pf.fit_res_w[i];
    }
    for (int i = 0; i < pf.n_bal_res; i++) {
        p.gen_score |= pf.bal_res[i] >= 2;
        list_eph |= pf.bal_res[i] == 2 && pf.w_balanced;
        p.bal_col[p.n_bal++] = pf.bal_res[i];
    }
    p.nx = 0;
    for (int c = 2; c < e->ncol; c++)
        if (pod->req[c] != 0 || (c == 2 && p.gen_score && list_eph)) p.xcol[p.nx++] = c;
    p.w_taint = pf.w_taint;
    p.w_aff = pod->n_preferred > 0 ? pf.w_nodeaffinity : 0; // node_affinity.go:243-246 PreScore Skip
    p.w_fit = pf.w_fit;
    p.w_img = pod->image_score ? pf.w_imagelocality : 0; // no image of the pod on any node: the plugin scores 0 everywhere
    for (int i = 0; i < pf.n_fit_res; i++) {
        if (pf.fit_res[i] == 0) { p.fit_cpu = 1; p.fit_w_cpu = pf.fit_res_w[i]; }
        if (pf.fit_res[i] == 1) { p.fit_mem = 1; p.fit_w_mem = pf.fit_res_w[i]; }
    }
    bool best_effort = true; // balanced_allocation.go:66-79
    for (int i = 0; i < pf.n_bal_res; i++) {
        if (pf.bal_res[i] == 0) p.bal_cpu = 1;
        if (pf.bal_res[i] == 1) p.bal_mem = 1;
        if (pod->req[pf.bal_res[i]] != 0) best_effort = false;
    }
    p.w_bal = best_effort ? 0 : pf.w_balanced;
    return p;
}

// argument checks shared by ccsim_set_pod and ccsim_set_pods (ranges the packed static word / the columns can hold)
static int validate_pod(ccsim_engine *e, const ccsim_pod *pod) {
    if (pod->n_taintsets < 1 || !pod->taint_filter_ok || !pod->taint_prefer_cnt)
        return fail(e, -EINVAL, "taint tables are required (n_taintsets >= 1)");
    int64_t wsum = 0;
    for (int t = 0; t < pod->n_preferred; t++) wsum += pod->preferred[t].weight;
    if (wsum > (int64_t)kStatAffMask) return fail(e, -EINVAL, "sum of preferred term weights too large");
    for (int t = 0; t < pod->n_taintsets; t++)
        if (pod->taint_prefer_cnt[t] < 0 || pod->taint_prefer_cnt[t] > (int32_t)kStatCntMask)
            return fail(e, -EINVAL, "taint_prefer_cnt out of range");
    for (int c = 0; c < CCSIM_MAX_RES; c++) {
        if (pod->req[c] < 0) return fail(e, -EINVAL, "negative request");
        if (c >= e->ncol && pod->req[c] != 0) return fail(e, -EINVAL, "request for a resource column the snapshot lacks");
    }
    if (pod->image_score)
        for (int64_t i = 0; i < e->n; i++)
            if (pod->image_score[i] > 100) return fail(e, -EINVAL, "image_score out of [0,100]");
    if (pod->has_host_ports && (e->prof.filter_mask & CCSIM_F_NODEPORTS) && !(e->prof.filter_mask & CCSIM_F_FIT))
        return fail(e, -ENOSYS, "NodePorts needs the NodeResourcesFit filter (one clone per node is kept as a pod-capacity clamp)");
    if (pod->volume_exclusive && !(e->prof.filter_mask & CCSIM_F_FIT))
        return fail(e, -ENOSYS, "volume_exclusive needs the NodeResourcesFit filter (one clone per node is kept as a pod-capacity clamp)");
    if (pod->volume_veto)
        for (int64_t i = 0; i < e->n; i++)
            if (pod->volume_veto[i] > CCSIM_VOL_CODES) return fail(e, -EINVAL, "volume_veto code out of range");
    return 0;
}

extern "C" int ccsim_set_pod(ccsim_engine *e, const ccsim_pod *pod) {
    if (!e || !pod) return -EINVAL;
    if (!e->have_nodes || !e->have_profile) return fail(e, -EINVAL, "load nodes and set the profile first");
    if (int vrc = validate_pod(e, pod)) return vrc;
    HIPCHK(e, hipSetDevice(e->device));
    if (int erc = ensure_cols(e)) return erc; // (a deferred ccsim_reset_state: the columns this pod's mirrors are built from)
    HIPCHK(e, hipStreamSynchronize(e->stream));
    drop_graph(e);
    free_list(e->pod_allocs);
    free_list(e->multi_allocs);
    e->multi = false;
    e->have_pod = e->begun = false;
    e->persist_hint = false;
    e->mb_go = -1;
    e->n_taintsets = pod->n_taintsets;

    const ccsim_profile &pf = e->prof;
    DevPod p = make_devpod(e, pod);
    e->pod = p;

    int rc;
    // NodePorts: every clone holds the pod's host ports, so a node takes at most one -- kept as a clamped copy of the
    // allocatable pod count (k_ports_clamp), which every Fit evaluation of every mode already tests
    e->cols.alloc_pods = e->d_alloc_pods_real, e->ports_on = false;
    e->d_ports_eff = e->d_ports_base = nullptr;
    e->excl_ports = (pf.filter_mask & CCSIM_F_NODEPORTS) && pod->has_host_ports;
    e->d_vol_veto = nullptr;
    if (e->excl_ports || pod->volume_exclusive) { // (a clone's own disks, volume_restrictions.go:105-150: the same construction)
        // (against the pod counts as they are NOW -- a pod spec set after runs of other specs finds their clones on the nodes -- and
        // rebuilt by ccsim_reset_state)
        if ((rc = dev_alloc(e, &e->d_ports_eff, (size_t)e->n_pad, e->pod_allocs))) return rc;
        if ((rc = dev_alloc(e, &e->d_ports_base, (size_t)e->n_pad, e->pod_allocs))) return rc;
        hipLaunchKernelGGL(k_ports_clamp, dim3((unsigned)((e->n_pad + kThreads - 1) / kThreads)), dim3(kThreads), 0, e->stream, e->d_ports_eff,
                           e->d_ports_base, e->d_alloc_pods_real, (const int32_t *)e->cols.pod_count, e->n_pad);
        HIPCHK(e, hipGetLastError());
        e->cols.alloc_pods = e->d_ports_eff, e->ports_on = true;
    }
    if ((rc = static_pass(e, pod, p.w_aff != 0, e->d_stat, e->d_sreason, e->pod_allocs))) return rc;
    e->ts_in_frame = pod->n_taintsets <= kFrameTs;
    if (e->ts_in_frame) e->d_hist_ts = e->d_frame_ts; // (zeroed before every use)
    else if ((rc = dev_alloc(e, &e->d_hist_ts, (size_t)pod->n_taintsets, e->pod_allocs))) return rc;

    // topology spread constraints: count tables + eligibility, built once.  hard -> Filter state
    // (filtering.go:235-308), soft -> Score state (scoring.go:61-178)
    e->pts = DevPts{};
    e->soft = DevSoft{};
    e->d_soft_pc0 = nullptr;
    e->d_pts_min_partials = nullptr;
    e->d_soft_partials = nullptr;
    e->pts_tables.clear();
    e->pts_table_len.clear();
    e->soft_flags.clear();
    e->dist_tables.clear();
    e->pts_present.clear();
    e->pts_col.clear(), e->soft_col.clear(), e->ipa_col.clear();
    e->cw_ok = false, e->cw_why = "no topology-coupled plugin";
    e->d_ipa_totals = nullptr;
    if (pod->n_spread < 0 || pod->n_spread > CCSIM_MAX_TSC) return fail(e, -EINVAL, "n_spread out of range");
    for (int pass = 0; pass < 2 && pod->n_spread > 0; pass++) { // pass 0: hard constraints, pass 1: soft constraints
        const bool hard = pass == 0;
        if (hard && !(pf.filter_mask & CCSIM_F_TOPOLOGYSPREAD)) continue;
        if (!hard && !pf.w_topologyspread) continue;
        std::vector<int> idx;
        for (int c = 0; c < pod->n_spread; c++)
            if ((pod->spread[c].hard != 0) == hard) idx.push_back(c);
        if (idx.empty()) continue;
        std::vector<int32_t *> lc((size_t)CCSIM_MAX_LABEL_COLS, nullptr);
        HIPCHK(e, hipMemcpy(lc.data(), e->d_label_cols, sizeof(int32_t *) * CCSIM_MAX_LABEL_COLS, hipMemcpyDeviceToHost));
        PtsInitArgs pi{};
        pi.n = e->n;
        DevPts pt{};
        pt.n = (int32_t)idx.size();
        std::vector<int32_t *> d_present(idx.size(), nullptr);
        for (size_t j = 0; j < idx.size(); j++) {
            const ccsim_spread_constraint &k = pod->spread[idx[j]];
            if (k.col < 0 || k.col >= e->n_label_cols) return fail(e, -EINVAL, "spread constraint label column out of range");
            if (k.max_skew < 1 || k.min_domains < 1 || k.n_domains < 0) return fail(e, -EINVAL, "bad spread constraint");
            pt.max_skew[j] = k.max_skew, pt.min_domains[j] = k.min_domains, pt.self_match[j] = k.self_match ? 1 : 0;
            pt.label[j] = lc[k.col];
            (hard ? e->pts_col : e->soft_col).push_back(k.col);
            const size_t len = (size_t)k.n_domains + 1;
            int32_t *tbl = nullptr, *tbl0 = nullptr, *ex = nullptr;
            uint8_t *inc = nullptr;
            if ((rc = dev_alloc(e, &tbl, len, e->pod_allocs))) return rc;
            if ((rc = dev_alloc(e, &tbl0, len, e->pod_allocs))) return rc;
            if ((rc = dev_alloc(e, &d_present[j], len, e->pod_allocs))) return rc;
            if (k.node_match_count && (rc = upload(e, &ex, k.node_match_count, (size_t)e->n, (size_t)e->n_pad, e->pod_allocs))) return rc;
            if (k.node_included && (rc = upload(e, &inc, k.node_included, (size_t)e->n, (size_t)e->n_pad, e->pod_allocs))) return rc;
            pt.tbl[j] = tbl;
            pi.existing[j] = ex, pi.included[j] = inc, pi.present[j] = d_present[j];
            e->pts_tables.emplace_back(tbl, tbl0);
            e->pts_table_len.push_back(len);
            if (!hard) {
                DevSoft &so = e->soft;
                so.max_skew[j] = k.max_skew, so.self_match[j] = k.self_match ? 1 : 0, so.is_hostname[j] = k.is_hostname ? 1 : 0;
                so.n_domains[j] = k.n_domains;
                if (k.missing_value < 0 || k.missing_value > k.n_domains || (k.missing_value && k.is_hostname))
                    return fail(e, -EINVAL, "spread constraint: missing_value must name a value id of a non-hostname key");
                so.nocredit[j] = k.missing_value;
                so.label[j] = lc[k.col], so.tbl[j] = tbl, so.existing[j] = ex;
                int32_t *flag = nullptr;
                if ((rc = dev_alloc(e, &flag, len, e->pod_allocs))) return rc;
                so.flag[j] = flag;
                e->soft_flags.emplace_back(flag, len);
            }
        }
        uint8_t *elig = nullptr;
        if ((rc = dev_alloc(e, &elig, (size_t)e->n_pad, e->pod_allocs))) return rc;
        pt.elig = elig;
        pi.elig = elig;
        pi.pts = pt;
        if (e->n > 0) hipLaunchKernelGGL(k_pts_init, dim3((unsigned)((e->n + kThreads - 1) / kThreads)), dim3(kThreads), 0, e->stream, pi);
        HIPCHK(e, hipGetLastError());
        const size_t first = e->pts_tables.size() - idx.size();
        for (size_t j = 0; j < idx.size(); j++) {
            const size_t len = e->pts_table_len[first + j];
            std::vector<int32_t> pres(len);
            HIPCHK(e, hipMemcpyAsync(pres.data(), d_present[j], len * 4, hipMemcpyDeviceToHost, e->stream));
            HIPCHK(e, hipMemcpyAsync(e->pts_tables[first + j].second, e->pts_tables[first + j].first, len * 4, hipMemcpyDeviceToDevice, e->stream));
            HIPCHK(e, hipStreamSynchronize(e->stream));
            int32_t np_ = 0;
            for (size_t v = 1; v < len; v++) np_ += pres[v] != 0;
            pt.n_present[j] = np_;
        }
        if (hard) {
            for (size_t j = 0; j < idx.size(); j++) {
                e->dist_tables.push_back({pt.tbl[j], (int64_t)e->pts_table_len[first + j], 4, 0});
                e->dist_tables.push_back({d_present[j], (int64_t)e->pts_table_len[first + j], 4, 1});
                e->pts_present.push_back(d_present[j]);
            }
            e->pts = pt;
            if ((rc = dev_alloc(e, &e->d_pts_min_partials, (size_t)kMaxGrid * kMaxTsc, e->pod_allocs))) return rc;
        } else {
            for (size_t j = 0; j < idx.size(); j++) // replicated across ranks like the hard constraints' (counts add)
                e->dist_tables.push_back({pt.tbl[j], (int64_t)e->pts_table_len[first + j], 4, 0});
            e->soft.n = pt.n, e->soft.w = pf.w_topologyspread, e->soft.elig = elig;
            // "clones on the node" = pod_count - pod_count0: the column as it stands NOW (the per-node counts of this pod describe the pods
            // behind it -- after runs of other pods on this engine that is not the loaded snapshot: the hosts' one-cycle-at-a-time loop for
            // refused template sets); ccsim_reset_state takes it back to the loaded snapshot's with the column.
            if ((rc = dev_alloc(e, &e->d_soft_pc0, (size_t)e->n_pad, e->pod_allocs))) return rc;
            HIPCHK(e, hipMemcpyAsync(e->d_soft_pc0, e->cols.pod_count, (size_t)e->n_pad * 4, hipMemcpyDeviceToDevice, e->stream));
            e->soft.pod_count0 = e->d_soft_pc0;
            if ((rc = dev_alloc(e, &e->d_soft_partials, (size_t)kMaxGrid * 3, e->pod_allocs))) return rc;
        }
    }
    // InterPodAffinity: topology-pair tables per key, filled from the snapshot's pods (filtering.go:204-272, scoring.go:128-221)
    e->ipa = DevIpa{};
    e->d_ipa_partials = nullptr;
    e->ipa_tables.clear();
    e->ipa_table_len.clear();
    e->ipa_aff_total0 = e->ipa_exist_total0 = e->ipa_entries0 = 0;
    const bool ipa_filter_on = (pf.filter_mask & CCSIM_F_INTERPODAFFINITY) != 0;
    if (pod->has_ipa && (ipa_filter_on || pf.w_interpodaffinity)) {
        const ccsim_ipa &ip = pod->ipa;
        if (ip.n_keys < 0 || ip.n_keys > CCSIM_MAX_IPA_KEYS || ip.n_aff_terms < 0 || ip.n_aff_terms > CCSIM_MAX_IPA_TERMS ||
            ip.n_anti_terms < 0 || ip.n_anti_terms > CCSIM_MAX_IPA_TERMS)
            return fail(e, -EINVAL, "bad inter-pod affinity dimensions");
        DevIpa &d = e->ipa;
        d.on = 1, d.filter_on = ipa_filter_on ? 1 : 0, d.w = pf.w_interpodaffinity;
        d.n_keys = ip.n_keys, d.n_aff = ip.n_aff_terms, d.n_anti = ip.n_anti_terms, d.self_aff = ip.self_aff ? 1 : 0;
        std::vector<int32_t *> lc((size_t)CCSIM_MAX_LABEL_COLS, nullptr);
        HIPCHK(e, hipMemcpy(lc.data(), e->d_label_cols, sizeof(int32_t *) * CCSIM_MAX_LABEL_COLS, hipMemcpyDeviceToHost));
        IpaInitArgs ii{};
        ii.n = e->n;
        // The engine's own key order: keys whose values are unique per node first (the windowed mode keeps their table entries in
        // the class tuple at fixed positions, ccsim_coupled.h).  Internal only -- and only on an unsharded snapshot: across ranks
        // the order must be the same, and uniqueness inside a shard says nothing about the cluster.
        int perm[CCSIM_MAX_IPA_KEYS], inv[CCSIM_MAX_IPA_KEYS]; // perm[new] = caller's index
        {
            int nn = 0;
            const bool reorder = e->global_offset == 0 && e->n_global == e->n;
            for (int pass = 0; pass < 2; pass++)
                for (int k = 0; k < ip.n_keys; k++) {
                    if (ip.key_col[k] < 0 || ip.key_col[k] >= e->n_label_cols || ip.key_ndom[k] < 0) return fail(e, -EINVAL, "bad inter-pod affinity key");
                    const bool u = reorder && label_col_unique(e, ip.key_col[k]);
                    if (u == (pass == 0)) perm[nn++] = k;
                }
            for (int k = 0; k < ip.n_keys; k++) inv[perm[k]] = k;
        }
        for (int t = 0; t < ip.n_aff_terms; t++) {
            if (ip.aff_key[t] < 0 || ip.aff_key[t] >= ip.n_keys) return fail(e, -EINVAL, "affinity term key out of range");
            d.aff_key[t] = inv[ip.aff_key[t]];
            d.aff_terms_on_key[inv[ip.aff_key[t]]]++;
        }
        for (int t = 0; t < ip.n_anti_terms; t++) {
            if (ip.anti_key[t] < 0 || ip.anti_key[t] >= ip.n_keys) return fail(e, -EINVAL, "anti-affinity term key out of range");
            d.anti_key[t] = inv[ip.anti_key[t]];
            if (ip.anti_self[t]) d.anti_self_on_key[inv[ip.anti_key[t]]]++;
            int32_t *ex = nullptr;
            if (ip.anti_existing[t] && (rc = upload(e, &ex, ip.anti_existing[t], (size_t)e->n, (size_t)e->n_pad, e->pod_allocs))) return rc;
            ii.anti_existing[t] = ex;
        }
        int32_t *affex = nullptr;
        if (ip.aff_existing && (rc = upload(e, &affex, ip.aff_existing, (size_t)e->n, (size_t)e->n_pad, e->pod_allocs))) return rc;
        ii.aff_existing = affex;
        for (int k = 0; k < ip.n_keys; k++) {
            const int o = perm[k]; // the caller's index of the engine's key k
            d.label[k] = lc[ip.key_col[o]];
            e->ipa_col.push_back(ip.key_col[o]);
            d.score_self[k] = ip.score_self[o];
            d.self_entries[k] = ip.self_entries[o];
            const size_t len = (size_t)ip.key_ndom[o] + 1;
            int64_t **tabs[4] = {&d.aff[k], &d.anti[k], &d.exist[k], &d.score[k]};
            for (auto tp : tabs) {
                int64_t *live = nullptr, *prist = nullptr;
                if ((rc = dev_alloc(e, &live, len, e->pod_allocs))) return rc;
                if ((rc = dev_alloc(e, &prist, len, e->pod_allocs))) return rc;
                *tp = live;
                e->ipa_tables.emplace_back(live, prist);
                e->ipa_table_len.push_back(len);
            }
            int32_t *ea = nullptr;
            int64_t *sx = nullptr;
            if (ip.exist_anti[o] && (rc = upload(e, &ea, ip.exist_anti[o], (size_t)e->n, (size_t)e->n_pad, e->pod_allocs))) return rc;
            if (ip.score_existing[o] && (rc = upload(e, &sx, ip.score_existing[o], (size_t)e->n, (size_t)e->n_pad, e->pod_allocs))) return rc;
            ii.exist_anti[k] = ea, ii.score_existing[k] = sx;
        }
        unsigned long long *d_tot = nullptr;
        if ((rc = dev_alloc(e, &d_tot, (size_t)3, e->pod_allocs))) return rc;
        e->d_ipa_totals = d_tot;
        if ((rc = dev_alloc(e, &e->d_ipa_partials, (size_t)kMaxGrid * 2, e->pod_allocs))) return rc;
        ii.ipa = d;
        ii.totals = d_tot;
        if (e->n > 0) hipLaunchKernelGGL(k_ipa_init, dim3((unsigned)((e->n + kThreads - 1) / kThreads)), dim3(kThreads), 0, e->stream, ii);
        HIPCHK(e, hipGetLastError());
        unsigned long long tot[2] = {0, 0};
        HIPCHK(e, hipMemcpyAsync(tot, d_tot, sizeof tot, hipMemcpyDeviceToHost, e->stream));
        for (size_t i = 0; i < e->ipa_tables.size(); i++)
            HIPCHK(e, hipMemcpyAsync(e->ipa_tables[i].second, e->ipa_tables[i].first, e->ipa_table_len[i] * 8, hipMemcpyDeviceToDevice, e->stream));
        HIPCHK(e, hipStreamSynchronize(e->stream));
        e->ipa_aff_total0 = (int64_t)tot[0];
        e->ipa_exist_total0 = (int64_t)tot[1];
        e->ipa_entries0 = ip.entries_existing;
        const unsigned long long ent = (unsigned long long)ip.entries_existing;
        HIPCHK(e, hipMemcpy(d_tot + 2, &ent, sizeof ent, hipMemcpyHostToDevice));
        for (size_t i = 0; i < e->ipa_tables.size(); i++) e->dist_tables.push_back({e->ipa_tables[i].first, (int64_t)e->ipa_table_len[i], 8, 0});
        e->dist_tables.push_back({d_tot, 3, 8, 0});
    }
    // Narrow mirrors: lossless iff every cpu value stays below 2^30 and every memory value is a multiple of the common
    // power-of-two unit with (value >> unit) below 2^30, for the whole run (a node takes at most max_pods clones).
    {
        DevCols &c = e->cols;
        c.narrow = 0, c.mem_shift = 0;
        const uint64_t orv = e->node_mem_or | (uint64_t)pod->req[1] | (uint64_t)pod->nz_mem;
        int sh = orv ? __builtin_ctzll(orv) : 0;
        if (sh > 40) sh = 40;
        const int64_t grow_c = (pod->req[0] > pod->nz_mcpu ? pod->req[0] : pod->nz_mcpu) * (e->node_max_pods + 1);
        const int64_t grow_m = (pod->req[1] > pod->nz_mem ? pod->req[1] : pod->nz_mem) * (e->node_max_pods + 1);
        const bool fits = e->node_max_cpu < (1ll << 30) && grow_c < (1ll << 30) && (e->node_max_mem >> sh) < (1ll << 30) &&
                          (grow_m >> sh) < (1ll << 30);
        if (e->narrow_allowed && e->pod.fit_enabled && e->pod.nx == 0 && fits && e->n > 0) {
            c.narrow = 1, c.mem_shift = sh;
            if ((rc = build_narrow(e))) return rc;
        }
    }
    if ((rc = cw_make_plan(e))) return rc;
    e->have_pod = true;
    return 0;
}

// ---- windowed mode for topology-coupled plugins: the plan of the current pod spec (ccsim_coupled.h) -----------------
// Which keys are unique per node (their "domain" is the node: table entries become class-tuple components), where every
// component sits, which shared-key tables the decide kernel keeps in LDS; the work buffers.  Anything that does not fit
// leaves cw_ok false with the reason: the one-pass-per-placement loop then runs, as before.
static bool label_col_unique(const ccsim_engine *e, int col) { // every value of the key sits on at most one node (hostname-like)
    const std::vector<int32_t> &v = e->h_label_cols[(size_t)col];
    std::vector<uint8_t> seen((size_t)e->label_col_max[(size_t)col] + 1, 0);
    for (int32_t x : v)
        if (x) {
            if (seen[(size_t)x]) return false;
            seen[(size_t)x] = 1;
        }
    return true;
}

static int cw_make_plan(ccsim_engine *e) {
    e->cw_ok = false;
    const bool coupled = e->pts.n > 0 || e->soft.n > 0 || e->ipa.on;
    if (!coupled) return 0;
    auto no = [&](const char *why) { e->cw_why = why; return 0; };
    if (!e->cw_allowed) return no("disabled (CCSIM_CW=0)");
    if (e->n <= 0) return no("empty snapshot");
    // On a node-range shard "unique per node" cannot be read off the shard's own column (every zone may occur once in a small shard)
    // and must be the same answer on every rank: a key is unique iff the caller declares as many domains as the CLUSTER has nodes
    // (then every node carries its own value) -- the table length says it.
    const bool sharded = e->global_offset != 0 || e->n_global != e->n;
    auto unique = [&](int col, size_t table_len) { return sharded ? ((int64_t)table_len - 1 == e->n_global && label_col_unique(e, col)) : label_col_unique(e, col); };
    CwPlan pl{};
    int i32 = 0, i64 = 0;
    if (e->pts.n > kCwMaxCons || e->soft.n > kCwMaxCons) return no("more than four hard / four soft spread constraints");
    for (int c = 0; c < e->pts.n; c++) { // tuple positions are fixed (ccsim_coupled.h kCwTuple)
        const int len = (int)e->pts_table_len[(size_t)c];
        pl.h_comp[c] = c, pl.h_unique[c] = unique(e->pts_col[(size_t)c], (size_t)len) ? 1 : 0, pl.h_len[c] = len;
        pl.h_present[c] = e->pts_present[(size_t)c];
        if (!pl.h_unique[c]) pl.h_off[c] = i32, pl.h_pres[c] = i32 + len, i32 += 2 * len;
    }
    const size_t soft_first = e->pts_tables.size() - (size_t)e->soft.n; // (soft tables follow the hard ones)
    for (int c = 0; c < e->soft.n; c++) {
        const int len = (int)e->pts_table_len[soft_first + (size_t)c];
        pl.s_comp[c] = kCwMaxCons + c, pl.s_len[c] = len;
        if (!e->soft.is_hostname[c]) pl.s_off[c] = i32, pl.s_bm[c] = i32 + len, i32 += len + (len + 31) / 32;
    }
    if (e->ipa.on)
        for (int k = 0; k < e->ipa.n_keys; k++) {
            const int len = (int)e->ipa_table_len[(size_t)k * 4];
            const int pos[4] = {kCwKeyPos0, kCwKeyPos1, kCwKeyPos2, kCwKeyPos3};
            pl.k_unique[k] = unique(e->ipa_col[(size_t)k], (size_t)len) ? 1 : 0, pl.k_len[k] = len, pl.k_comp[k] = pos[k];
            if (pl.k_unique[k] && k >= 2) return no("a unique-per-node topology key beyond the second inter-pod affinity key");
            if (!pl.k_unique[k]) pl.k_off[k] = i64, i64 += 4 * len;
        }
    if (i32 > kCwLdsI32 || i64 > kCwLdsI64) return no("shared-key tables exceed the decide kernel's LDS budget");
    pl.n_comp = kCwTuple, pl.i32_words = i32, pl.i64_words = i64;
    pl.window = 4096, pl.list_len = 64; // (round 5: 4096 = 64 classes x 64 members, with the whole lists staged; profiles/r03/bench_coupled.txt, C5-shaped template at 100k nodes: 512/32 -> 0.98M placements/s, 1024/64 -> 1.03M;
                                        // round 4: 2048 -- with 64 classes the lists carry that many cycles, with 16 they still end a window at ~1000)
    if (const char *f = getenv("CCSIM_CW_WINDOW")) pl.window = atoi(f); // tuning / test knobs
    if (const char *f = getenv("CCSIM_CW_LIST")) pl.list_len = atoi(f);
    pl.sweep = 1;
    if (const char *f = getenv("CCSIM_CW_SWEEP")) pl.sweep = atoi(f) != 0; // A/B and test knob: 0 = one placement per step of the deciding wave
    pl.window = pl.window < 1 ? 1 : (pl.window > kCwFastWindow ? kCwFastWindow : pl.window); // (the general decide kernel clamps to its own kCwMaxWindow)
    pl.list_len = pl.list_len < 1 ? 1 : (pl.list_len > kCwMaxList ? kCwMaxList : pl.list_len);
    const int64_t blocks = (e->n_pad + kCwTile - 1) / kCwTile;
    // the merge of the blocks' lists: one workgroup stages merge_group * L keys; more blocks than that take a second level
    int64_t group = kCwMaxKeys / pl.list_len;
    group = group > 2 * 64 ? 2 * 64 : group; // (a lane of the merging wave keeps the heads of two blocks in registers)
    if (const char *f = getenv("CCSIM_CW_MERGE_GROUP")) group = atoi(f) >= 2 && atoi(f) < group ? atoi(f) : group; // test knob: two levels on small snapshots
    while (pl.list_len > 1 && !getenv("CCSIM_CW_MERGE_GROUP") && (blocks + group - 1) / group > group) pl.list_len >>= 1, group = kCwMaxKeys / pl.list_len > 128 ? 128 : kCwMaxKeys / pl.list_len;
    if ((blocks + group - 1) / group > group) return no("snapshot too large for the class-list merge");
    // a node-range shard: windows for the one shape whose deciding wave needs nothing of a remote node but what its owner staged
    // (ccsim_coupled.h "windows on shards": one hard constraint over a shared key + one unique-per-node inter-pod key)
    const bool shard_shape = e->pts.n == 1 && !pl.h_unique[0] && pl.h_len[0] <= 65 && e->soft.n == 0 && e->ipa.on && e->ipa.n_keys == 1 && pl.k_unique[0];
    if (sharded && !shard_shape) return no("sharded snapshot: windows only for one shared-key hard constraint + one unique inter-pod key");
    if (sharded && getenv("CCSIM_CW_SHARDS") && !atoi(getenv("CCSIM_CW_SHARDS"))) return no("disabled on shards (CCSIM_CW_SHARDS=0)");
    // work buffers: [keys | ready | ctl | classes] zeroed per run, the rest written before it is read
    CwWork w{};
    w.n_blocks = (int)blocks, w.merge_group = (int)group;
    const size_t zb = sizeof(unsigned long long) * kCwSlots + sizeof(uint32_t) * kCwSlots + sizeof(uint32_t) * 16 + sizeof(CwClass) * kCwSlots;
    unsigned char *base = nullptr;
    int rc;
    if ((rc = dev_alloc(e, &base, zb, e->pod_allocs))) return rc;
    w.keys = (unsigned long long *)base;
    w.ready = (uint32_t *)(base + sizeof(unsigned long long) * kCwSlots);
    w.ctl = w.ready + kCwSlots;
    w.cls = (CwClass *)(w.ctl + 16);
    e->cw_zero_base = base, e->cw_zero_bytes = zb;
    if ((rc = dev_alloc(e, &w.slot_of_id, (size_t)kCwMaxClasses, e->pod_allocs))) return rc;
    if ((rc = dev_alloc(e, &w.node_slot, (size_t)e->n_pad, e->pod_allocs))) return rc;
    if ((rc = dev_alloc(e, &w.node_A, (size_t)e->n_pad, e->pod_allocs))) return rc;
    if ((rc = dev_alloc(e, &w.node_A1, (size_t)e->n_pad, e->pod_allocs))) return rc;
    if (getenv("CCSIM_CW_PROF") && atoi(getenv("CCSIM_CW_PROF")) && (rc = dev_alloc(e, &w.prof, (size_t)16, e->pod_allocs))) return rc; // measurement runs
    if ((rc = dev_alloc(e, &w.top, (size_t)blocks * kCwMaxClasses * (size_t)pl.list_len, e->pod_allocs, false))) return rc;
    if ((rc = dev_alloc(e, &w.lists, (size_t)kCwMaxClasses * (size_t)pl.list_len, e->pod_allocs))) return rc;
    if (blocks > group && (rc = dev_alloc(e, &w.top2, (size_t)((blocks + group - 1) / group) * kCwMaxClasses * (size_t)pl.list_len, e->pod_allocs, false))) return rc;
    if ((rc = dev_alloc(e, &w.umin, (size_t)blocks * kMaxTsc, e->pod_allocs))) return rc;
    if ((rc = dev_alloc(e, &w.part, (size_t)blocks * kCwMaxClasses, e->pod_allocs, false))) return rc;
    if (blocks > group && (rc = dev_alloc(e, &w.part2, (size_t)((blocks + group - 1) / group) * kCwMaxClasses, e->pod_allocs, false))) return rc;
    if (shard_shape) { // (also on an unsharded snapshot: a one-rank communicator takes the same path)
        // the window record of this rank, the gathered records (up to 8 ranks), the cluster's classes and merged lists
        if ((rc = dev_alloc(e, &w.xsend, kCwXBytes, e->pod_allocs)) || (rc = dev_alloc(e, &w.xrecv, kCwXBytes * 8, e->pod_allocs)) ||
            (rc = dev_alloc(e, &w.xhdr, (size_t)4, e->pod_allocs)) || (rc = dev_alloc(e, &w.xcls, (size_t)kCwXClasses, e->pod_allocs)) ||
            (rc = dev_alloc(e, &w.xent, (size_t)kCwXClasses * kCwMaxList, e->pod_allocs)))
            return rc;
    }
    unsigned char *argbuf = nullptr;
    if ((rc = dev_alloc(e, &argbuf, sizeof(CwDecideArgs), e->pod_allocs))) return rc;
    e->d_cw_args = argbuf;
    e->cw_plan = pl, e->cw_work = w, e->cw_ok = true, e->cw_why.clear();
    e->cw_fast = e->pts.n <= 2 && !(e->soft.n > 0 && e->soft.w) && (!e->ipa.on || e->ipa.n_keys <= 2);
    if (const char *f = getenv("CCSIM_CW_FAST")) e->cw_fast = e->cw_fast && atoi(f) != 0; // A/B knob
    return 0;
}

static int build_narrow(ccsim_engine *e) {
    if (!e->cols.narrow) return 0;
    const int blocks = (int)((e->n_pad + kThreads - 1) / kThreads);
    hipLaunchKernelGGL(k_narrow_build, dim3(blocks), dim3(kThreads), 0, e->stream, e->cols, e->d_a32[0], e->d_a32[1]);
    HIPCHK(e, hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------
// t0/t1 (measurement runs only): hipExtLaunchKernelGGL stamps the events when THIS dispatch is picked up / has
// finished.  The pick-up stamp can precede the predecessor's end (barrier bit), so durations are taken between the
// STOP stamps of consecutive dispatches of the in-order stream: stop(kernel) - stop(predecessor) = the kernel's
// duration plus the ~1.5 us kernel boundary -- without draining the stream (an idle gap changes clocks and caches).
#define CCSIM_LAUNCH(kern, g, b, stream, t0, t1, arg)                                   \
    do {                                                                                \
        if (t0 || t1) hipExtLaunchKernelGGL(kern, g, b, 0, stream, t0, t1, 0, arg);     \
        else hipLaunchKernelGGL(kern, g, b, 0, stream, arg);                            \
    } while (0)

static bool profile_has_scoring(const ccsim_profile &p) { // fwk.HasScorePlugins() (schedule_one.go:619-621)
    return p.w_taint || p.w_nodeaffinity || p.w_fit || p.w_balanced || p.w_topologyspread || p.w_interpodaffinity || p.w_imagelocality;
}

// numFeasibleNodesToFind (schedule_one.go:697-723)
static int64_t num_feasible_nodes_to_find(int32_t percentage, int64_t n_all) {
    if (n_all < 100) return n_all;
    if (percentage == 0) {
        percentage = (int32_t)(50 - n_all / 125);
        if (percentage < 5) percentage = 5;
    }
    const int64_t num = n_all * percentage / 100;
    return num < 100 ? 100 : num;
}

// the two passes of a sampled cycle (wide columns only: the sampled search is not the throughput path)
template <bool PTS, int SMP>
static void launch_scan_smp(ccsim_engine *e, const ScanArgs &a) {
    const int nx = e->pod.nx;
    dim3 g(e->grid), b(kThreads);
    if (nx == 0) hipLaunchKernelGGL((k_scan<0, PTS, false, SMP>), g, b, 0, e->stream, a);
    else if (nx == 1) hipLaunchKernelGGL((k_scan<1, PTS, false, SMP>), g, b, 0, e->stream, a);
    else if (nx == 2) hipLaunchKernelGGL((k_scan<2, PTS, false, SMP>), g, b, 0, e->stream, a);
    else if (nx <= 4) hipLaunchKernelGGL((k_scan<4, PTS, false, SMP>), g, b, 0, e->stream, a);
    else hipLaunchKernelGGL((k_scan<kMaxExtra, PTS, false, SMP>), g, b, 0, e->stream, a);
}

template <bool PTS>
static void launch_scan_t(ccsim_engine *e, const ScanArgs &a, hipEvent_t t0, hipEvent_t t1) {
    const int nx = e->pod.nx;
    dim3 g(e->grid), b(kThreads);
    if (nx == 0 && e->cols.narrow) CCSIM_LAUNCH((k_scan<0, PTS, true>), g, b, e->stream, t0, t1, a);
    else if (nx == 0) CCSIM_LAUNCH((k_scan<0, PTS>), g, b, e->stream, t0, t1, a);
    else if (nx == 1) CCSIM_LAUNCH((k_scan<1, PTS>), g, b, e->stream, t0, t1, a);
    else if (nx == 2) CCSIM_LAUNCH((k_scan<2, PTS>), g, b, e->stream, t0, t1, a);
    else if (nx <= 4) CCSIM_LAUNCH((k_scan<4, PTS>), g, b, e->stream, t0, t1, a);
    else CCSIM_LAUNCH((k_scan<kMaxExtra, PTS>), g, b, e->stream, t0, t1, a);
}

static ScanArgs scan_args(ccsim_engine *e) {
    ScanArgs a{};
    a.c = e->cols, a.p = e->pod, a.st = e->d_state, a.partials = e->d_partials, a.chunk = e->chunk;
    a.pts = e->pts, a.pts_min_partials = e->d_pts_min_partials;
    a.ipa = e->ipa, a.ipa_partials = e->d_ipa_partials;
    a.soft = e->soft, a.soft_partials = e->d_soft_partials;
    a.n_partials = e->grid;
    a.xsend = e->d_xsend, a.xrecv = e->d_xrecv, a.n_ranks = e->n_ranks;
    a.log = e->d_log;
    a.smp_partials = e->d_smp_partials, a.smp_prefix = e->d_smp_prefix;
    return a;
}

static int launch_scan(ccsim_engine *e, hipEvent_t t0 = nullptr, hipEvent_t t1 = nullptr) {
    ScanArgs a = scan_args(e);
    if (e->pts.n > 0 || e->ipa.on || e->soft.n > 0) launch_scan_t<true>(e, a, t0, t1);
    else launch_scan_t<false>(e, a, t0, t1);
    return 0;
}

static int launch_final(ccsim_engine *e) {
    hipLaunchKernelGGL(k_final, dim3(1), dim3(kThreads), 0, e->stream, scan_args(e));
    return 0;
}

// the sequential cycle as one dispatch (k_scan_fused): no topology-coupled plugin, every node scored, one GPU
static bool fused_ok(const ccsim_engine *e) {
    return e->fused_allowed && e->mode == CCSIM_MODE_SEQUENTIAL && e->n_ranks == 0 && e->smp_K == 0 && !e->time_passes && e->pts.n == 0 &&
           e->soft.n == 0 && !e->ipa.on;
}
static FusedArgs fused_args(ccsim_engine *e, int parity) {
    FusedArgs a{};
    a.c = e->cols, a.p = e->pod, a.st[0] = e->d_state, a.st[1] = e->d_state2, a.partials[0] = e->d_partials, a.partials[1] = e->d_partials2;
    a.chunk = e->chunk, a.n_partials = e->grid, a.parity = parity, a.log = e->d_log;
    return a;
}
static void launch_scan_fused(ccsim_engine *e, int parity) {
    const FusedArgs a = fused_args(e, parity);
    const int nx = e->pod.nx;
    dim3 g(e->grid), b(kThreads);
    if (nx == 0 && e->cols.narrow) hipLaunchKernelGGL((k_scan_fused<0, true>), g, b, 0, e->stream, a);
    else if (nx == 0) hipLaunchKernelGGL((k_scan_fused<0, false>), g, b, 0, e->stream, a);
    else if (nx == 1) hipLaunchKernelGGL((k_scan_fused<1, false>), g, b, 0, e->stream, a);
    else if (nx == 2) hipLaunchKernelGGL((k_scan_fused<2, false>), g, b, 0, e->stream, a);
    else if (nx <= 4) hipLaunchKernelGGL((k_scan_fused<4, false>), g, b, 0, e->stream, a);
    else hipLaunchKernelGGL((k_scan_fused<kMaxExtra, false>), g, b, 0, e->stream, a);
}
// `rounds` cycles: an even number of fused launches (the state ends in the buffer the host reads) + the pending decision
static void launch_fused_batch(ccsim_engine *e, int rounds) {
    const int n = (rounds + 1) & ~1;
    for (int r = 0; r < n; r++) launch_scan_fused(e, r & 1);
    hipLaunchKernelGGL(k_final_fused, dim3(1), dim3(kThreads), 0, e->stream, fused_args(e, 0));
}

// one scheduling cycle attempt of the sequential mode: the scan(s) + the one-block reduction / decision
static void launch_cycle(ccsim_engine *e) {
    if (e->smp_K > 0) { // sampled search: count, prefix, score the first K feasible nodes of the visiting order
        const ScanArgs a = scan_args(e);
        const bool coupled = e->pts.n > 0 || e->ipa.on || e->soft.n > 0;
        if (coupled) launch_scan_smp<true, 1>(e, a); else launch_scan_smp<false, 1>(e, a);
        hipLaunchKernelGGL(k_smp_prefix, dim3(1), dim3(kThreads), 0, e->stream, a);
        if (coupled) launch_scan_smp<true, 2>(e, a); else launch_scan_smp<false, 2>(e, a);
        hipLaunchKernelGGL(k_final, dim3(1), dim3(kThreads), 0, e->stream, a);
        return;
    }
    launch_scan(e);
    launch_final(e);
}


static LevelArgs level_args(ccsim_engine *e) {
    return LevelArgs{e->cols, e->pod, e->d_state, e->d_lpartials, e->d_cpartials, e->d_blockprefix, e->d_cscore, e->d_log,
                     e->chunk, e->lvl_chunk};
}

// the batched mode's full pods x nodes pass (k_level_score); t0/t1 as in launch_scan
static int launch_level_score(ccsim_engine *e, hipEvent_t t0 = nullptr, hipEvent_t t1 = nullptr) {
    const LevelArgs a = level_args(e);
    const int nx = e->pod.nx;
    dim3 g(e->grid), b(kThreads);
    if (nx == 0 && e->cols.narrow) CCSIM_LAUNCH((k_level_score<0, true>), g, b, e->stream, t0, t1, a);
    else if (nx == 0) CCSIM_LAUNCH(k_level_score<0>, g, b, e->stream, t0, t1, a);
    else if (nx == 1) CCSIM_LAUNCH(k_level_score<1>, g, b, e->stream, t0, t1, a);
    else if (nx == 2) CCSIM_LAUNCH(k_level_score<2>, g, b, e->stream, t0, t1, a);
    else if (nx <= 4) CCSIM_LAUNCH(k_level_score<4>, g, b, e->stream, t0, t1, a);
    else CCSIM_LAUNCH(k_level_score<kMaxExtra>, g, b, e->stream, t0, t1, a);
    return 0;
}

static int launch_level_commit(ccsim_engine *e, hipEvent_t t1 = nullptr, hipEvent_t t0 = nullptr) {
    const LevelArgs a = level_args(e);
    const int nx = e->pod.nx;
    dim3 g(e->lvl_grid), b(kThreads);
    if (nx == 0 && e->cols.narrow) CCSIM_LAUNCH((k_level_commit<0, true>), g, b, e->stream, t0, t1, a);
    else if (nx == 0) CCSIM_LAUNCH(k_level_commit<0>, g, b, e->stream, t0, t1, a);
    else if (nx == 1) CCSIM_LAUNCH(k_level_commit<1>, g, b, e->stream, t0, t1, a);
    else if (nx == 2) CCSIM_LAUNCH(k_level_commit<2>, g, b, e->stream, t0, t1, a);
    else if (nx <= 4) CCSIM_LAUNCH(k_level_commit<4>, g, b, e->stream, t0, t1, a);
    else CCSIM_LAUNCH(k_level_commit<kMaxExtra>, g, b, e->stream, t0, t1, a);
    return 0;
}

static LevelFinalArgs level_final_args(ccsim_engine *e, bool commit_launched = true, bool score_launched = true) {
    LevelFinalArgs f{};
    f.commit_launched = commit_launched, f.score_launched = score_launched;
    f.st = e->d_state;
    f.partials = e->d_lpartials;
    f.n_partials = e->grid;
    f.cpartials = e->d_cpartials;
    f.n_cpartials = e->lvl_grid;
    f.blockprefix = e->d_blockprefix;
    f.xsend = e->d_xsend;
    f.xrecv = e->d_xrecv;
    f.n_ranks = e->n_ranks;
    f.rank = e->rank;
    f.want_log = e->d_log ? 1 : 0;
    return f;
}

// commit rows -> columns: in front of every k_level_score launch (conditional on the full pass being due) and when a
// batched run ends (unconditional)
static void launch_rows_flush(ccsim_engine *e, bool only_if_full) {
    if (!e->rows_active) return;
    const int blocks = (int)((e->n_pad + kThreads - 1) / kThreads);
    hipLaunchKernelGGL(k_rows_flush, dim3(blocks), dim3(kThreads), 0, e->stream, e->cols, (const DevState *)e->d_state, only_if_full ? 1 : 0);
}

static int launch_level_final(ccsim_engine *e, bool commit_launched = true, bool score_launched = true) {
    hipLaunchKernelGGL(k_level_final, dim3(1), dim3(kFinalThreads), 0, e->stream, level_final_args(e, commit_launched, score_launched));
    return 0;
}



static int begin_run(ccsim_engine *e, int64_t max_limit, int mode, int64_t log_cap) {
    if (!e->have_nodes || !e->have_profile || !e->have_pod) return fail(e, -EINVAL, "nodes, profile and pod must be set");
    if (mode != CCSIM_MODE_SEQUENTIAL && mode != CCSIM_MODE_BATCHED) return fail(e, -ENOSYS, "mode %d not implemented", mode);
    if (e->pod.nx != 0 || e->multi) e->extras_dirty = true;
    // every placement of this run moves a node's memory columns by this pod's (non-zero) request: whatever pod spec is set NEXT must
    // choose the narrow mirrors' memory unit among the divisors of these values too, not only of the snapshot's -- a spec with a coarser
    // unit set after runs of a finer one read mirrors that had lost the low bits (found by the one-cycle-at-a-time loop over several pod
    // specs, tests/test_multi.py::test_refused_spec_sets_*: alternating specs of 64 MiB and 1 GiB)
    e->node_mem_or |= (uint64_t)e->pod.req[1] | (uint64_t)e->pod.nz_mem;
    if (e->ipa.on && mode == CCSIM_MODE_BATCHED)
        return fail(e, -ENOSYS, "inter-pod affinity couples nodes through topology pairs: use CCSIM_MODE_SEQUENTIAL");
    if (e->soft.n > 0 && mode == CCSIM_MODE_BATCHED)
        return fail(e, -ENOSYS, "ScheduleAnyway topology spread constraints score every node against cluster-wide counts: "
                                "sequential mode only");
    if (e->soft.n > 0 && e->n_ranks > 0) { // the candidate-domain sets travel as bitmaps in the exchange record (XRec, kXSoftBits)
        int bits = 0;
        for (int c = 0; c < e->soft.n; c++) bits += e->soft.is_hostname[c] ? 0 : e->soft.n_domains[c];
        if (bits > kXSoftBits)
            return fail(e, -ENOSYS, "ScheduleAnyway topology spread constraints over %d domains in total: more than the %d a sharded run's "
                                    "exchange record carries", bits, kXSoftBits);
    }
    if (e->pts.n > 0 && mode == CCSIM_MODE_BATCHED)
        return fail(e, -ENOSYS, "batched mode is not valid with topology spread constraints (a placement changes the feasibility of "
                                "other nodes): use CCSIM_MODE_SEQUENTIAL");
    // a profile without any Score plugin keeps ONE feasible node per cycle (schedule_one.go:619-621: numNodesToFind = 1): the
    // sampled search with K = 1 -- the first feasible node of the rotating visiting order wins, the search stops at the second
    const int64_t k_find = profile_has_scoring(e->prof) ? num_feasible_nodes_to_find(e->prof.percentage_of_nodes_to_score, e->n_global) : 1;
    const int64_t smp_K = k_find < e->n_global ? k_find : 0;
    if (smp_K > 0 && mode == CCSIM_MODE_BATCHED)
        return fail(e, -ENOSYS, "percentageOfNodesToScore < 100 (here: the first %lld feasible nodes of %lld) makes the outcome depend on the "
                                "visiting order: sequential mode only", (long long)smp_K, (long long)e->n_global);
    // on shards the sampled search is two exchanges per cycle (counts, then the max-loc: DevState::smp_phase).  Round 6: pods with
    // topology-coupled plugins take it too -- their Filter state is the replicated tables (the same on every rank), the counting pass
    // filters with the assumed global minimum, the scoring pass verifies it (k_decide: a stale minimum sends the cycle back to the
    // counting pass), and the PreScore facts (candidate domains, raw-score ranges) are gathered over the SELECTED nodes only, which is
    // what the reference's PreScore sees (filteredNodes: schedule_one.go:757-790).  CCSIM_DIST_SMP_COUPLED=0 restores the refusal.
    if (smp_K > 0 && e->n_ranks > 0 && (e->pts.n > 0 || e->ipa.on || e->soft.n > 0) && getenv("CCSIM_DIST_SMP_COUPLED") && atoi(getenv("CCSIM_DIST_SMP_COUPLED")) == 0)
        return fail(e, -ENOSYS, "percentageOfNodesToScore < 100 on several GPUs with topology spread constraints or inter-pod affinity: "
                                "switched off (CCSIM_DIST_SMP_COUPLED=0)");
    if (smp_K != e->smp_K) drop_graph(e);
    e->smp_K = smp_K;
    if (mode == CCSIM_MODE_BATCHED && !e->pod.fit_enabled)
        return fail(e, -ENOSYS, "batched mode needs the NodeResourcesFit filter (a run-down is bounded by the node's pod capacity)");
    HIPCHK(e, hipSetDevice(e->device));
    if (log_cap != e->log_cap) {
        if (e->d_log) HIPCHK(e, hipFree(e->d_log));
        e->d_log = nullptr;
        e->log_cap = 0;
        drop_graph(e);
        if (log_cap > 0) {
            HIPCHK(e, hipMalloc((void **)&e->d_log, sizeof(int32_t) * (size_t)log_cap));
            e->log_cap = log_cap;
        }
    }
    DevState st{};
    for (int c = 0; c < kMaxTsc; c++) st.pts_min_a[c] = 0x7fffffff;
    if (!e->begun) { // a fresh snapshot state (load / reset); otherwise the run continues where the last one stopped
        e->ipa_aff_total_cur = e->ipa_aff_total0, e->ipa_exist_total_cur = e->ipa_exist_total0, e->ipa_entries_cur = e->ipa_entries0;
        e->smp_start_cur = 0;
    }
    st.smp_K = e->smp_K;
    st.smp_start = e->smp_K > 0 ? e->smp_start_cur % (e->n_global > 0 ? e->n_global : 1) : 0;
    st.smp_N = e->n_global, st.smp_off = 0, st.smp_phase = 0, st.smp_rank = e->n_ranks > 0 ? e->rank : 0;
    st.ipa_aff_total = e->ipa_aff_total_cur, st.ipa_exist_total = e->ipa_exist_total_cur, st.ipa_entries = e->ipa_entries_cur;
    for (int c = 0; c < kMaxTsc; c++) st.soft_size_a[c] = -1; // unknown: the first scan derives sizes and weights
    st.soft_min_a = INT64_MAX, st.soft_max_a = 0;
    for (auto &fl : e->soft_flags) HIPCHK(e, hipMemsetAsync(fl.first, 0, fl.second * 4, e->stream)); // epochs restart at 1
    st.limit = max_limit;
    st.lvl_full = 1; // no score cache yet
    st.lvl_ev = -1;
    {   // several score levels per pass (blind, validated, rolled back if need be: ccsim_level.h) wherever the commit rows exist
        const int persist = (mode == CCSIM_MODE_BATCHED && e->n_ranks == 0 && !e->time_passes) ? persist_k(e) : 0;
        const bool rows_now = mode == CCSIM_MODE_BATCHED && e->cols.narrow && e->n > 0 && !persist;
        int kb = 384; // (profiles/r03/persist_batch_sweep.txt; the sharded protocol at one rank: 64 -> 3.30 ms, 256 -> 2.74, 384 -> 2.60, 512 -> 2.88 per C4 run)
        if (const char *f = getenv("CCSIM_LEVEL_BATCH")) kb = atoi(f) > 0 ? atoi(f) : 1; // tuning knob (the SAME value on every rank)
        st.lvl_kb_max = rows_now && !e->time_passes ? kb : 1;
        st.lvl_kb = st.lvl_kb_max;
    }
    st.winner = -1;
    st.mode = mode;
    st.log_cap = e->log_cap;
    *e->h_state = st;
    const int persist_next = (mode == CCSIM_MODE_BATCHED && e->n_ranks == 0 && !e->time_passes) ? persist_k(e) : 0;
    // a single-rank persistent launch follows: the state travels with the zeros of the histogram and of the launch's sync block behind
    // it, in one copy (the step frame; run_persist then skips its three fills)
    e->frame_sent = persist_next != 0 && e->persist_vranks == 0 && e->ts_in_frame && !getenv("CCSIM_FRAME_OFF");
    if (e->frame_sent) {
        *reinterpret_cast<DevState *>(e->h_init) = st; // (everything behind it in h_init stays zero)
        HIPCHK(e, hipMemcpyAsync(e->d_frame, e->h_init, kFrameHostBytes, hipMemcpyHostToDevice, e->stream));
    } else
        HIPCHK(e, hipMemcpyAsync(e->d_state, e->h_state, sizeof(DevState), hipMemcpyHostToDevice, e->stream));
    if (!persist_next) // (the persistent launch WRITES the per-run counts: cnt_assign)
        HIPCHK(e, hipMemsetAsync(e->cols.placed_cnt, 0, sizeof(int32_t) * (size_t)e->n_pad, e->stream));
    if (e->d_log && e->n_ranks > 0) // shards fill disjoint positions of the global log: -1 = "not mine"
        HIPCHK(e, hipMemsetAsync(e->d_log, 0xff, sizeof(int32_t) * (size_t)e->log_cap, e->stream));
    e->kernel_ms = 0;
    e->pass_kernel_ms = 0;
    e->pass_launches = 0;
    e->pass_events_used = 0;
    e->limit = max_limit;
    e->mode = mode;
    e->begun = true;
    e->persist_run = (mode == CCSIM_MODE_BATCHED && e->n_ranks == 0 && !e->time_passes) ? persist_k(e) : 0;
    // topology-coupled plugins of one template: windows of placements per pass when every node is scored (ccsim_coupled.h)
    e->cw_run = mode == CCSIM_MODE_SEQUENTIAL && e->cw_ok && e->n_ranks == 0 && e->smp_K == 0 && !e->time_passes && e->n > 0;
    // the sampled search of a template without topology-coupled plugins: cycles on resident block summaries (ccsim_sampled.h)
    e->sb_run = false, e->sf_handover = false;
    if (mode == CCSIM_MODE_SEQUENTIAL && e->smp_K > 0 && e->n_ranks == 0 && !e->time_passes && e->sb_allowed && e->pts.n == 0 && e->soft.n == 0 && !e->ipa.on &&
        e->global_offset == 0 && e->n_global == e->n) {
        // a lap of the ring at a time wants blocks of <= K nodes (one stretch boundary per block at most) that one wave reads -- 256 nodes,
        // or 64 on small snapshots -- and at most 4096 of them (the tree over them lives in LDS); K >= 100 whenever a profile with Score
        // plugins samples (schedule_one.go:697-723), so only K = 1 (no Score plugin) and snapshots beyond 2^20 nodes stay on the
        // cycle-at-a-time form
        auto blocks_at = [&](int x) { return (e->n_pad + ((int64_t)1 << x) - 1) >> x; };
        int sh = e->smp_K >= 256 ? 8 : 6;
        if (const char *f = getenv("CCSIM_SB_SHIFT")) sh = atoi(f); // test knob
        e->sb_laps = e->sb_allowed == 1 && (sh == 6 || sh == 8) && ((int64_t)1 << sh) <= e->smp_K && blocks_at(sh) <= kLapMaxBlocks;
        if (!e->sb_laps) {
            sh = 8;
            while (sh <= kSbMaxShift && blocks_at(sh) > kSbMaxBlocks) sh++;
        }
        if (sh <= kSbMaxShift) {
            const int blocks = (int)blocks_at(sh);
            if (!e->d_sb_memo) {
                int rc2;
                if ((rc2 = dev_alloc(e, &e->d_sb_memo, (size_t)e->n_pad, e->allocs, false)) || (rc2 = dev_alloc(e, &e->d_sb_flag8, (size_t)e->n_pad, e->allocs, false)) || (rc2 = dev_alloc(e, &e->d_sb_fc, (size_t)kSbMaxBlocks, e->allocs)) ||
                    (rc2 = dev_alloc(e, &e->d_sb_key, (size_t)kSbMaxBlocks, e->allocs)) || (rc2 = dev_alloc(e, &e->d_sb_mx, (size_t)kSbMaxBlocks, e->allocs)))
                    return rc2;
            }
            e->sb_shift = sh, e->sb_blocks = blocks, e->sb_run = true;
            e->h_state->sb_dirty = 1; // nothing is known about the columns under the run's maxima yet
            HIPCHK(e, hipMemcpyAsync(e->d_state, e->h_state, sizeof(DevState), hipMemcpyHostToDevice, e->stream));
        }
    }
    // the full search (every node scored) of such a template on the same summaries: one wave, one trip to L2 per cycle (ccsim_search_full.h)
    e->sf_run = false;
    if (mode == CCSIM_MODE_SEQUENTIAL && e->smp_K == 0 && e->n_ranks == 0 && !e->time_passes && e->pts.n == 0 && e->soft.n == 0 && !e->ipa.on && e->global_offset == 0 &&
        e->n_global == e->n && e->n > 0 && !(getenv("CCSIM_SF") && !atoi(getenv("CCSIM_SF")))) {
        int sh = 8;
        if (((e->n_pad + 255) >> 8) > kSfMaxBlocks) sh = 10;
        if (const char *f = getenv("CCSIM_SF_SHIFT")) sh = atoi(f) == 10 ? 10 : 8; // test knob
        const int64_t blocks = (e->n_pad + ((int64_t)1 << sh) - 1) >> sh;
        if (blocks <= kSfMaxBlocks) {
            if (!e->d_sb_memo) {
                int rc2;
                if ((rc2 = dev_alloc(e, &e->d_sb_memo, (size_t)e->n_pad, e->allocs, false)) || (rc2 = dev_alloc(e, &e->d_sb_flag8, (size_t)e->n_pad, e->allocs, false)) || (rc2 = dev_alloc(e, &e->d_sb_fc, (size_t)kSbMaxBlocks, e->allocs)) ||
                    (rc2 = dev_alloc(e, &e->d_sb_key, (size_t)kSbMaxBlocks, e->allocs)) || (rc2 = dev_alloc(e, &e->d_sb_mx, (size_t)kSbMaxBlocks, e->allocs)))
                    return rc2;
            }
            e->sb_shift = sh, e->sb_blocks = (int)blocks, e->sf_run = true, e->sb_laps = false;
            e->h_state->sb_dirty = 1; // nothing is known about the columns under the run's maxima yet
            HIPCHK(e, hipMemcpyAsync(e->d_state, e->h_state, sizeof(DevState), hipMemcpyHostToDevice, e->stream));
        }
    }
    // ... and of ONE template with a hard spread constraint over a shared key of <= 64 values, optionally with inter-pod terms over a key
    // that is unique per node and without inter-pod scores: per-(block, zone) entries under the mask of eligible zones (ccsim_sampled_zone.h)
    e->sz_run = false;
    if (mode == CCSIM_MODE_SEQUENTIAL && e->smp_K > 0 && e->n_ranks == 0 && !e->time_passes && e->sb_allowed == 1 && (e->pts.n > 0 || e->ipa.on) && e->global_offset == 0 &&
        e->n_global == e->n) {
        auto no = [&](const char *why) { e->sz_why = why; };
        const int sh = e->smp_K >= 256 ? 8 : 6;
        const int64_t blocks = (e->n_pad + ((int64_t)1 << sh) - 1) >> sh;
        bool ipa_fits = !e->ipa.on;
        if (e->ipa.on) {
            ipa_fits = e->ipa.n_keys == 1 && e->ipa.n_aff == 0 && e->ipa.self_entries[0] == 0 && e->ipa.score_self[0] == 0 && e->ipa_entries0 == 0 &&
                       label_col_unique(e, e->ipa_col[0]);
        }
        if (e->pts.n != 1 || e->soft.n > 0) no("not exactly one DoNotSchedule constraint (and no ScheduleAnyway ones)");
        else if (e->pts_table_len[0] > (size_t)kSzZones + 1) no("more than 64 topology values");
        else if (label_col_unique(e, e->pts_col[0])) no("the constraint's key is unique per node");
        else if (!ipa_fits) no("inter-pod terms beyond required anti-affinity over one unique-per-node key (or inter-pod scores)");
        else if (e->smp_K < 64) no("fewer than 64 nodes kept per cycle");
        else if (blocks > kSzMaxBlocks) no("more than 4096 blocks of nodes");
        else if (getenv("CCSIM_SZ") && !atoi(getenv("CCSIM_SZ"))) no("disabled (CCSIM_SZ=0)");
        else {
            e->sz_why.clear();
            if (!e->d_sb_memo) {
                int rc2;
                if ((rc2 = dev_alloc(e, &e->d_sb_memo, (size_t)e->n_pad, e->allocs, false)) || (rc2 = dev_alloc(e, &e->d_sb_flag8, (size_t)e->n_pad, e->allocs, false)) ||
                    (rc2 = dev_alloc(e, &e->d_sb_fc, (size_t)kSbMaxBlocks, e->allocs)) || (rc2 = dev_alloc(e, &e->d_sb_key, (size_t)kSbMaxBlocks, e->allocs)) ||
                    (rc2 = dev_alloc(e, &e->d_sb_mx, (size_t)kSbMaxBlocks, e->allocs)))
                    return rc2;
            }
            if (!e->d_sz_zone8) {
                int rc2;
                if ((rc2 = dev_alloc(e, &e->d_sz_zone8, (size_t)e->n_pad, e->allocs, false)) || (rc2 = dev_alloc(e, &e->d_sz_ent_key, (size_t)kSzMaxBlocks * kSzZones, e->allocs)) ||
                    (rc2 = dev_alloc(e, &e->d_sz_ent_flg, (size_t)kSzMaxBlocks * kSzZones, e->allocs)) || (rc2 = dev_alloc(e, &e->d_sz_cntz, (size_t)kSzMaxBlocks * kSzZones, e->allocs)) ||
                    (rc2 = dev_alloc(e, &e->d_sz_present, (size_t)1, e->allocs)) || (rc2 = dev_alloc(e, &e->d_sz_over, (size_t)1, e->allocs)))
                    return rc2;
            }
            HIPCHK(e, hipMemsetAsync(e->d_sz_present, 0, sizeof(unsigned long long), e->stream)); // (a new pod spec may count other nodes)
            HIPCHK(e, hipMemsetAsync(e->d_sz_over, 0, sizeof(uint32_t), e->stream));
            HIPCHK(e, hipMemsetAsync(e->d_sz_cntz, 0, (size_t)kSzMaxBlocks * kSzZones, e->stream));
            e->sb_shift = sh, e->sb_blocks = (int)blocks, e->sz_run = true, e->sz_built = false, e->sb_run = false, e->sb_laps = false;
            e->h_state->sb_dirty = 1; // nothing is known about the columns under the run's maxima yet
            HIPCHK(e, hipMemcpyAsync(e->d_state, e->h_state, sizeof(DevState), hipMemcpyHostToDevice, e->stream));
        }
    }
    e->cw_shard_run = mode == CCSIM_MODE_SEQUENTIAL && e->cw_ok && e->cw_fast && e->n_ranks > 0 && e->n_ranks <= 8 && e->smp_K == 0 && !e->time_passes && e->cw_work.xsend != nullptr &&
                      !(getenv("CCSIM_CW_SHARDS") && !atoi(getenv("CCSIM_CW_SHARDS")));
    e->cw_shard_args = false;
    if (e->cw_run || e->cw_shard_run) HIPCHK(e, hipMemsetAsync(e->cw_zero_base, 0, e->cw_zero_bytes, e->stream));
    const bool rows = mode == CCSIM_MODE_BATCHED && e->cols.narrow && e->n > 0 && !e->persist_run;
    if (rows != e->rows_active) drop_graph(e);
    e->rows_active = rows;
    if (rows) {
        const int blocks = (int)((e->n_pad + kThreads - 1) / kThreads);
        hipLaunchKernelGGL(k_rows_build, dim3(blocks), dim3(kThreads), 0, e->stream, e->cols);
        HIPCHK(e, hipGetLastError());
    }
    return 0;
}

static int read_state(ccsim_engine *e) {
    HIPCHK(e, hipMemcpyAsync(e->h_state, e->d_state, sizeof(DevState), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    e->ipa_aff_total_cur = e->h_state->ipa_aff_total, e->ipa_exist_total_cur = e->h_state->ipa_exist_total;
    e->ipa_entries_cur = e->h_state->ipa_entries;
    e->smp_start_cur = e->h_state->smp_start;
    return 0;
}

static void launch_pass(ccsim_engine *e) { // one scan pass + its one-block reduction/decision
    // Measurement runs (eager): the duration of the pass's dominant kernel (k_scan; k_level_commit in batched mode) is
    // taken between the STOP stamps of two consecutive dispatches of the in-order stream (hipExtLaunchKernelGGL): an
    // empty marker kernel right before it, and the kernel itself.  (A start stamp is taken at packet pick-up, possibly
    // before the predecessor ends; the first marker absorbs the flush of the previous pass's dirty lines.)
    hipEvent_t t0 = nullptr, t1 = nullptr;
    if (e->smp_K > 0) {
        launch_cycle(e);
        return;
    }
    if (e->time_passes && e->n_ranks == 0) {
        while ((int)e->pass_events.size() < e->pass_events_used + 3) {
            hipEvent_t ev = nullptr;
            if (hipEventCreate(&ev) != hipSuccess) break;
            e->pass_events.push_back(ev);
        }
        if ((int)e->pass_events.size() >= e->pass_events_used + 3) {
            hipEvent_t scratch = e->pass_events[e->pass_events_used++];
            t0 = e->pass_events[e->pass_events_used++];
            t1 = e->pass_events[e->pass_events_used++];
            hipExtLaunchKernelGGL(k_noop, dim3(1), dim3(64), 0, e->stream, nullptr, scratch, 0, 0);
            if (e->mode != CCSIM_MODE_BATCHED) hipExtLaunchKernelGGL(k_noop, dim3(1), dim3(64), 0, e->stream, nullptr, t0, 0, 0);
            else hipExtLaunchKernelGGL(k_noop, dim3(1), dim3(64), 0, e->stream, nullptr, nullptr, 0, 0);
            if (e->mode == CCSIM_MODE_BATCHED) // the dominant kernel is the commit pass: its own start and stop stamps
                launch_level_commit(e, t1, t0), launch_rows_flush(e, true), launch_level_score(e), launch_level_final(e);
            else launch_scan(e, nullptr, t1), launch_final(e);
            return;
        }
    }
    if (e->mode == CCSIM_MODE_BATCHED) {
        launch_level_commit(e); // the level found by the previous pass (sparse: reads the 4-byte score cache)
        launch_rows_flush(e, true);
        launch_level_score(e);
        launch_level_final(e);
    } else {
        launch_scan(e);
        launch_final(e);
    }
}

static void collect_pass_times(ccsim_engine *e) { // after a stream sync
    for (int i = 0; i + 2 < e->pass_events_used; i += 3) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, e->pass_events[i + 1], e->pass_events[i + 2]) == hipSuccess) e->pass_kernel_ms += ms, e->pass_launches++;
    }
    e->pass_events_used = 0;
}

static int enqueue_rounds(ccsim_engine *e, int rounds) {
    if (e->use_graph && e->n_ranks == 0 && !e->time_passes) {
        if (!e->graph_exec || e->graph_rounds != rounds || e->graph_mode != e->mode) {
            drop_graph(e);
            e->pass_events_used = 0;
            HIPCHK(e, hipStreamBeginCapture(e->stream, hipStreamCaptureModeThreadLocal));
            if (e->mode == CCSIM_MODE_BATCHED) {
                // Full passes are rare (first pass; the normalization constants moved): two score-only passes at the head
                // (stale constants found, then the real level), then commit-only passes.  If a full pass falls due inside
                // the stretch, the rest of this replay is no-ops and the next replay starts with it.
                for (int r = 0; r < 2; r++) launch_rows_flush(e, true), launch_level_score(e), launch_level_final(e, false, true);
                for (int r = 0; r < rounds; r++) launch_level_commit(e), launch_level_final(e, true, false);
            } else if (fused_ok(e))
                launch_fused_batch(e, rounds);
            else
                for (int r = 0; r < rounds; r++) launch_pass(e);
            HIPCHK(e, hipStreamEndCapture(e->stream, &e->graph));
            HIPCHK(e, hipGraphInstantiate(&e->graph_exec, e->graph, nullptr, nullptr, 0));
            e->graph_events = e->pass_events_used;
            e->graph_rounds = rounds;
            e->graph_mode = e->mode;
        }
        HIPCHK(e, hipGraphLaunch(e->graph_exec, e->stream));
    } else {
        if (fused_ok(e)) launch_fused_batch(e, rounds);
        else
            for (int r = 0; r < rounds; r++) launch_pass(e);
        HIPCHK(e, hipGetLastError());
    }
    return 0;
}

static int fill_report(ccsim_engine *e, ccsim_report *out) {
    const DevState &st = *e->h_state;
    out->placed = st.placed;
    out->stop = e->n_global == 0 ? CCSIM_STOP_NO_NODES : (st.done == DONE_LIMIT ? CCSIM_STOP_LIMIT : CCSIM_STOP_UNSCHEDULABLE);
    out->rounds = st.rounds;
    out->scans = st.scans;
    out->evaluated_total = e->smp_K > 0 ? st.evaluated : st.rounds * e->n_global;
    out->last_feasible = st.last_feasible;
    out->kernel_ns = (int64_t)(e->kernel_ms * 1e6);
    out->pass_kernel_ns = (int64_t)(e->pass_kernel_ms * 1e6);
    out->pass_launches = e->pass_launches;
    // algorithmic bytes per scan: the columns the active plugin set must read once per node
    int64_t per_node = 4 /*static word*/ + 6 * 8 /*alloc,req,nz x cpu,mem*/ + 2 * 4 /*pods*/ + (int64_t)e->pod.nx * 16 +
                       (e->pts.n ? 1 + 4 * (int64_t)e->pts.n : 0) /*eligibility byte + topology value id per constraint*/ +
                       (e->ipa.on ? 4 * (int64_t)e->ipa.n_keys : 0) /*topology value id per inter-pod affinity key*/ +
                       (e->soft.n ? 1 + 4 * (int64_t)e->soft.n : 0);
    out->bytes_per_scan = per_node * e->n;
    memset(out->hist, 0, sizeof(out->hist));
    out->n_code_unschedulable = 0;
    if (out->hist_taintset)
        for (int i = 0; i < out->hist_taintset_cap; i++) out->hist_taintset[i] = 0;
    // The per-node counts (4 MB at 1M nodes: 75 us over PCIe) and the log leave on a second stream, beside the terminal round's
    // diagnosis pass (k_hist: 33 us at 1M nodes) and its small copies -- both only read the final state.
    hipStream_t cs = e->stream;
    if (e->n >= (1 << 16) && st.done == DONE_UNSCHEDULABLE && !e->early_counts && (out->per_node_count || (out->log && e->d_log))) {
        if (!e->copy_stream) {
            if (hipStreamCreateWithFlags(&e->copy_stream, hipStreamNonBlocking) != hipSuccess) e->copy_stream = nullptr;
            if (e->copy_stream && hipEventCreateWithFlags(&e->ev_copy, hipEventDisableTiming) != hipSuccess) {
                (void)hipStreamDestroy(e->copy_stream);
                e->copy_stream = nullptr;
            }
        }
        if (e->copy_stream) {
            HIPCHK(e, hipEventRecord(e->ev_copy, e->stream)); // (whatever the run still has in flight: the rows' flush of the multi-kernel form)
            HIPCHK(e, hipStreamWaitEvent(e->copy_stream, e->ev_copy, 0));
            cs = e->copy_stream;
        }
    }
    out->per_node_filled_width = e->early_counts && e->early_narrow ? e->early_narrow : (out->per_node_count ? 4 : 0);
    if (out->per_node_count) {
        if (out->per_node_cap < e->n) return fail(e, -EINVAL, "per_node_cap too small");
        if (!e->early_counts) // (the persistent launch's counts are on the host already)
            HIPCHK(e, hipMemcpyAsync(out->per_node_count, e->cols.placed_cnt, sizeof(int32_t) * (size_t)e->n, hipMemcpyDeviceToHost, cs));
    }
    e->early_counts = false, e->early_narrow = 0;
    out->log_len = 0;
    if (out->log && e->d_log) {
        int64_t len = st.placed < e->log_cap ? st.placed : e->log_cap;
        if (len > out->log_cap) len = out->log_cap;
        if (len > 0) HIPCHK(e, hipMemcpyAsync(out->log, e->d_log, sizeof(int32_t) * (size_t)len, hipMemcpyDeviceToHost, cs));
        out->log_len = len;
    }
    if (st.done == DONE_UNSCHEDULABLE && e->n > 0 && e->hist_in_kernel && e->early_hist) { // diagnosis by the persistent launch, copied behind it
        const unsigned long long *hp = e->early_hist_framed ? reinterpret_cast<const unsigned long long *>(e->h_frame + kFrameOffHist) : e->h_hist_pin;
        for (int i = 0; i < CCSIM_NREASON; i++) out->hist[i] = (int64_t)hp[i];
        out->n_code_unschedulable = (int64_t)hp[CCSIM_NREASON];
        if (out->hist_taintset)
            for (int i = 0; i < e->n_taintsets && i < out->hist_taintset_cap; i++) out->hist_taintset[i] = (int64_t)hp[CCSIM_NREASON + 1 + i];
        e->hist_in_kernel = e->early_hist = false;
    } else if (st.done == DONE_UNSCHEDULABLE && e->n > 0) {
        // terminal round: FitError diagnosis (types.go:787-836)
        if (!e->hist_in_kernel) { // (the persistent launch fills the histogram itself, from the node state it holds in LDS)
            if (int erc = ensure_cols(e)) return erc; // (k_hist reads the int64 request columns: a persistent launch may have left them behind the mirrors)
            HIPCHK(e, hipMemsetAsync(e->d_hist, 0, sizeof(unsigned long long) * (CCSIM_NREASON + 1), e->stream));
            HIPCHK(e, hipMemsetAsync(e->d_hist_ts, 0, sizeof(unsigned long long) * (size_t)e->n_taintsets, e->stream));
            HistArgs h{e->cols, e->pod, e->d_hist, e->d_hist_ts, e->d_hist_code, e->n_taintsets, e->pts, e->d_state, e->ipa,
                       e->ports_on ? 1 : 0, e->excl_ports ? 1 : 0, e->d_vol_veto, e->d_alloc_pods_real, e->d_ports_base};
            int64_t hb = (e->n + kThreads - 1) / kThreads;
            if (hb > 2048) hb = 2048;
            hipLaunchKernelGGL(k_hist, dim3((unsigned)hb), dim3(kThreads), 0, e->stream, h);
            HIPCHK(e, hipGetLastError());
        }
        e->hist_in_kernel = false;
        std::vector<unsigned long long> hh(CCSIM_NREASON + 1), ht((size_t)e->n_taintsets);
        HIPCHK(e, hipMemcpyAsync(hh.data(), e->d_hist, sizeof(unsigned long long) * hh.size(), hipMemcpyDeviceToHost, e->stream));
        HIPCHK(e, hipMemcpyAsync(ht.data(), e->d_hist_ts, sizeof(unsigned long long) * ht.size(), hipMemcpyDeviceToHost, e->stream));
        HIPCHK(e, hipStreamSynchronize(e->stream));
        for (int i = 0; i < CCSIM_NREASON; i++) out->hist[i] = (int64_t)hh[i];
        out->n_code_unschedulable = (int64_t)hh[CCSIM_NREASON];
        if (out->hist_taintset)
            for (int i = 0; i < e->n_taintsets && i < out->hist_taintset_cap; i++) out->hist_taintset[i] = (int64_t)ht[i];
    }
    HIPCHK(e, hipStreamSynchronize(e->stream));
    if (cs != e->stream) HIPCHK(e, hipStreamSynchronize(cs));
    return 0;
}

// The persistent form of the batched mode (ccsim_persist.h): narrow mirrors, one 512-thread workgroup (kPThreads) per CU with up to
// 4096 nodes each in LDS, scores and pod counts in 16 bits.  Everything else takes the multi-kernel path.
// `mb`: the mailbox form (several ranks; its own instantiation, hence its own occupancy answer); `sharded`: this engine holds one shard.
static const void *persist_fn(int k, bool mb) {
    if (mb) return k == 1 ? (const void *)k_level_persist<1, true> : k == 2 ? (const void *)k_level_persist<2, true> : k == 4 ? (const void *)k_level_persist<4, true> : (const void *)k_level_persist<8, true>;
    return k == 1 ? (const void *)k_level_persist<1, false> : k == 2 ? (const void *)k_level_persist<2, false> : k == 4 ? (const void *)k_level_persist<4, false> : (const void *)k_level_persist<8, false>;
}
static int persist_k_impl(const ccsim_engine *e, bool mb, bool sharded) {
    if (!e->persist_allowed || !e->cols.narrow || e->pod.nx != 0 || (!sharded && e->n_ranks != 0) || e->n_cus <= 0 || e->n <= 0) return 0;
    if (e->node_max_pods > 65535 || e->node_max_podcount > 65535) return 0;
    const int64_t max_total = 100ll * ((int64_t)e->pod.w_taint + e->pod.w_aff + e->pod.w_fit + e->pod.w_bal + e->pod.w_img);
    if (max_total >= 65535) return 0;
    const int cus = e->n_cus < kPMaxGrid ? e->n_cus : kPMaxGrid;
    for (int k : {1, 2, 4, 8})
        if ((e->n_pad + (int64_t)k * kPThreads - 1) / ((int64_t)k * kPThreads) <= cus) {
            // the hand-rolled grid barrier needs every workgroup resident at once: ask the runtime how many fit (LDS, registers), not
            // just how many CUs there are; what it cannot know -- a CU mask, another tenant -- is caught by the barrier's
            // bounded spin, after which ccsim_run continues on the multi-kernel path
            int *per_cu = e->persist_per_cu[mb ? 1 : 0]; // per engine = per device (a process-wide static held the first device's answer: ADVICE r3)
            if (per_cu[k] < 0) {
                int nb = 0;
                per_cu[k] = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, persist_fn(k, mb), kPThreads, 0) == hipSuccess ? nb : 0;
            }
            const int64_t grid = (e->n_pad + (int64_t)k * kPThreads - 1) / ((int64_t)k * kPThreads);
            return (int64_t)per_cu[k] * e->n_cus >= grid ? k : 0;
        }
    return 0;
}
static int persist_k(const ccsim_engine *e) { return persist_k_impl(e, e->persist_vranks > 0, false); }

// the deferred half of ccsim_reset_state: the node columns back from their pristine copies, the mirrors and the NodePorts clamp after them
static int ensure_cols(ccsim_engine *e) {
    if (!e->reset_pending) {
        if (e->wide_stale) { // (the mirrors are the state: the int64 columns follow them)
            e->wide_stale = false;
            HIPCHK(e, hipSetDevice(e->device));
            const DevCols &c = e->cols;
            hipLaunchKernelGGL(k_widen, dim3((unsigned)((e->n_pad + 255) / 256)), dim3(256), 0, e->stream, (const int32_t *)c.r32[0], (const int32_t *)c.r32[1],
                               (const int32_t *)c.z32[0], (const int32_t *)c.z32[1], c.req[0], c.req[1], c.nz_mcpu, c.nz_mem, c.mem_shift, e->n_pad);
            HIPCHK(e, hipGetLastError());
        }
        return 0;
    }
    e->reset_pending = false;
    e->wide_stale = false; // (everything is restored from the pristine copies)
    e->extras_dirty = false;
    HIPCHK(e, hipSetDevice(e->device));
    for (size_t i = 0; i < e->backups.size(); i++)
        HIPCHK(e, hipMemcpyAsync(e->backups[i].first, e->backups[i].second, e->backup_bytes[i], hipMemcpyDeviceToDevice, e->stream));
    if (e->have_pod) {
        int rc = build_narrow(e);
        if (rc) return rc;
        if (e->ports_on) { // the NodePorts clamp follows the restored pod counts
            hipLaunchKernelGGL(k_ports_clamp, dim3((unsigned)((e->n_pad + kThreads - 1) / kThreads)), dim3(kThreads), 0, e->stream, e->d_ports_eff,
                               e->d_ports_base, e->d_alloc_pods_real, (const int32_t *)e->cols.pod_count, e->n_pad);
            HIPCHK(e, hipGetLastError());
        }
    }
    return 0;
}

static void launch_persist(ccsim_engine *e, int k, bool mb, int grid, const PersistArgs &a) {
#define CCSIM_PERSIST_LAUNCH(KK, MBB) hipLaunchKernelGGL((k_level_persist<KK, MBB>), dim3(grid), dim3(kPThreads), 0, e->stream, a)
    if (mb) {
        if (k == 1) CCSIM_PERSIST_LAUNCH(1, true); else if (k == 2) CCSIM_PERSIST_LAUNCH(2, true); else if (k == 4) CCSIM_PERSIST_LAUNCH(4, true); else CCSIM_PERSIST_LAUNCH(8, true);
    } else {
        if (k == 1) CCSIM_PERSIST_LAUNCH(1, false); else if (k == 2) CCSIM_PERSIST_LAUNCH(2, false); else if (k == 4) CCSIM_PERSIST_LAUNCH(4, false); else CCSIM_PERSIST_LAUNCH(8, false);
    }
#undef CCSIM_PERSIST_LAUNCH
}

static PersistArgs persist_args(ccsim_engine *e) {
    PersistArgs a{};
    const DevCols &c = e->cols;
    a.c = PersistCols{};
    a.c.a32[0] = c.a32[0], a.c.a32[1] = c.a32[1], a.c.r32[0] = c.r32[0], a.c.r32[1] = c.r32[1], a.c.z32[0] = c.z32[0], a.c.z32[1] = c.z32[1];
    a.c.alloc_pods = c.alloc_pods, a.c.pod_count = c.pod_count, a.c.placed_cnt = c.placed_cnt, a.c.stat = c.stat;
    a.c.req[0] = c.req[0], a.c.req[1] = c.req[1], a.c.nz_mcpu = c.nz_mcpu, a.c.nz_mem = c.nz_mem;
    a.c.n_pad = c.n_pad, a.c.global_offset = c.global_offset, a.c.mem_shift = c.mem_shift, a.c.n = c.n;
    a.c.sreason = c.sreason, a.c.taintset_id = c.taintset_id, a.c.n_taintsets = e->n_taintsets, a.c.rows = c.rows;
    a.p = e->pod, a.st = e->d_state, a.sync = e->d_psync, a.log = e->d_log, a.want_log = e->d_log ? 1 : 0;
    a.max_syncs = 1 << 19;
    a.seq_steps = 8;
    if (const char *f = getenv("CCSIM_SEQ_STEPS")) a.seq_steps = atoi(f) > 0 ? atoi(f) : kSeqSteps; // tuning knob
    // levels per blind batch, measured on the C4 snapshot.  Round 2 (every state of a run-down evaluated): 16 -> 1.97 ms, 64 -> 1.57, 128 -> 1.64.
    // Round 3 (run_down_safe_skip: long run-downs cost a bisection): 64 -> 0.88 ms, 128 -> 0.85, 256 -> 0.96, 384 -> 0.555, 512 -> 0.77,
    // >= 640 -> 0.68 (profiles/r03/persist_batch_sweep.txt; not monotone: what a batch costs depends on where the normalization
    // maxima run out of holders inside it).  Any value gives the same results.
    // Round 3, second half (a rolled-back batch locates its event, the rescore phase predicts it: ccsim_persist.h ev_level): the size hardly
    // matters any more -- 192 -> 0.314 ms, 384 -> 0.279, 1024 -> 0.266 (profiles/r03/persist_batch_sweep_predicted_events.txt).
    a.level_batch = 1024;
    if (const char *f = getenv("CCSIM_LEVEL_BATCH")) a.level_batch = atoi(f) > 0 ? atoi(f) : 1; // tuning knob
    if (const char *f = getenv("CCSIM_PERSIST_PROF")) a.prof = atoi(f);
    if (const char *f = getenv("CCSIM_PERSIST_FAULT")) a.fault = atoi(f); // test knob: the lost-workgroup path (tests/test_persist.py)
    a.spec_cut = e->n_global < (1ll << 24) && !(getenv("CCSIM_PERSIST_SPEC") && atoi(getenv("CCSIM_PERSIST_SPEC")) == 0); // (A/B knob)
    a.end_at_empty = !(getenv("CCSIM_PERSIST_END") && atoi(getenv("CCSIM_PERSIST_END")) == 0); // (A/B knob)
    a.hint_valid = e->persist_hint ? 1 : 0, a.hint_mt = e->persist_hint_mt, a.hint_ma = e->persist_hint_ma;
    return a;
}

// this engine's mailbox(es): fine-grained device memory (coherent for peers' writes over xGMI and for system-scope polls)
static int mbox_alloc(ccsim_engine *e) {
    if (e->d_mbox) return 0;
    void *p = nullptr;
    if (hipExtMallocWithFlags(&p, sizeof(PersistMailbox) * kPMaxRanks, hipDeviceMallocFinegrained) != hipSuccess) {
        (void)hipGetLastError();
        HIPCHK(e, hipMalloc(&p, sizeof(PersistMailbox) * kPMaxRanks)); // (virtual ranks on one device work in ordinary memory too)
        e->mbox_coarse = true; // ... but a peer DEVICE's stores into it and this device's polls of it are not coherent: such a box is
                               // never advertised to another device (ccsim_dist_mbox_info / _connect)
    }
    HIPCHK(e, hipMemset(p, 0, sizeof(PersistMailbox) * kPMaxRanks));
    e->d_mbox = (PersistMailbox *)p;
    HIPCHK(e, hipMalloc((void **)&e->d_mb_ok, sizeof(int32_t) * 4));
    HIPCHK(e, hipHostMalloc((void **)&e->h_mb_ok, sizeof(int32_t) * 4, hipHostMallocDefault));
    return 0;
}

static int run_persist(ccsim_engine *e, int k, ccsim_report *out) {
    PersistArgs a = persist_args(e);
    const int grid = (int)((e->n_pad + (int64_t)k * kPThreads - 1) / ((int64_t)k * kPThreads));
    // validation of the mailbox form on ONE device: the grid's workgroups split into virtual ranks with separate sync blocks and
    // mailboxes -- the protocol of a sharded run (ccsim_dist_run), same memory scopes, no second process
    const bool mb = e->persist_vranks > 0;
    int sync_blocks = 1;
    if (mb) {
        int rc = mbox_alloc(e);
        if (rc) return rc;
        int v = e->persist_vranks < grid ? e->persist_vranks : grid;
        a.bpr = (grid + v - 1) / v;
        a.vranks = a.n_ranks = (grid + a.bpr - 1) / a.bpr;
        a.rank = 0;
        for (int r = 0; r < a.n_ranks; r++) a.mbox[r] = e->d_mbox + r;
        sync_blocks = a.n_ranks;
    }
    // the FitError diagnosis of the terminal cycle comes out of the launch itself (the node state is in LDS): no k_hist pass.  Not with
    // host ports (the clamped pod capacity hides the real one) and not in the mailbox form (its state goes to the commit rows).
    const bool diag = !mb && !e->ports_on && !e->d_vol_veto;
    for (int launch = 0; launch < 64; launch++) {
        a.c.from_pristine = e->reset_pending ? 1 : 0;
        if (e->reset_pending) { // (backups: the wide columns in load order -- req[0 .. ncol), nz_mcpu, nz_mem, pod_count)
            a.c.p_req[0] = (const int64_t *)e->backups[0].second, a.c.p_req[1] = (const int64_t *)e->backups[1].second;
            a.c.p_nz[0] = (const int64_t *)e->backups[(size_t)e->ncol].second, a.c.p_nz[1] = (const int64_t *)e->backups[(size_t)e->ncol + 1].second;
            a.c.p_pod_count = (const int32_t *)e->backups[(size_t)e->ncol + 2].second;
            if (e->extras_dirty) { // this pod does not read them, the next one may: the reset that was asked for covers every column
                for (size_t i = 2; i < (size_t)e->ncol; i++)
                    HIPCHK(e, hipMemcpyAsync(e->backups[i].first, e->backups[i].second, e->backup_bytes[i], hipMemcpyDeviceToDevice, e->stream));
                e->extras_dirty = false;
            }
        }
        a.c.cnt_assign = launch == 0 ? 1 : 0; // (begin_run zeroed the per-run counts)
        // ccsim_report.per_node_count_narrow: offered, and every count fits (a node takes at most its pod capacity)
        int narrow = 0;
        if (!mb && out && out->per_node_count_narrow && out->per_node_cap >= e->n && (out->per_node_narrow_width == 1 || out->per_node_narrow_width == 2) &&
            e->pod.fit_enabled && e->node_max_pods < (out->per_node_narrow_width == 1 ? 256 : 65536))
            narrow = out->per_node_narrow_width;
        if (narrow && !e->d_cnt_narrow) {
            uint16_t *p16 = nullptr;
            int rc = dev_alloc(e, &p16, (size_t)e->n_pad, e->allocs);
            if (rc) return rc;
            e->d_cnt_narrow = p16;
        }
        a.c.cnt_narrow = narrow ? e->d_cnt_narrow : nullptr, a.c.cnt_narrow_width = narrow;
        a.c.skip_wide = !mb && e->lazy_wide ? 1 : 0;
        a.c.hist = diag ? e->d_hist : nullptr, a.c.hist_ts = e->d_hist_ts, a.c.hist_code = e->d_hist_code;
        if (mb) a.tag_base = 0x80000000u | ((e->persist_seq++ & 0x7ffu) << 20); // (bit 31: a virtual-rank run -- never the tag of a sharded run's launch, which shares the boxes)
        const bool framed = e->frame_sent && launch == 0 && !mb && sync_blocks == 1; // begin_run's copy zeroed all three
        e->frame_sent = false;
        if (!framed) {
            HIPCHK(e, hipMemsetAsync(e->d_psync, 0, sizeof(PersistSync) * (size_t)sync_blocks, e->stream));
            if (diag) {
                HIPCHK(e, hipMemsetAsync(e->d_hist, 0, sizeof(unsigned long long) * (CCSIM_NREASON + 1), e->stream));
                HIPCHK(e, hipMemsetAsync(e->d_hist_ts, 0, sizeof(unsigned long long) * (size_t)e->n_taintsets, e->stream));
            }
        }
        HIPCHK(e, hipEventRecord(e->ev0, e->stream));
        launch_persist(e, k, mb, grid, a);
        HIPCHK(e, hipGetLastError());
        HIPCHK(e, hipEventRecord(e->ev1, e->stream));
        // the results leave right behind the kernel, before the host knows how the launch ended: the state block, the per-node
        // counts (4 MB at 1M nodes: the step's longest transfer starts the moment the kernel ends), the diagnosis -- ONE sync
        const bool one_copy = diag && e->ts_in_frame && !getenv("CCSIM_FRAME_OFF"); // state + histogram + per-taint-set bins: contiguous in the frame
        if (one_copy)
            HIPCHK(e, hipMemcpyAsync(e->h_frame, e->d_frame, kFrameOffHist + sizeof(unsigned long long) * (size_t)(CCSIM_NREASON + 1 + e->n_taintsets), hipMemcpyDeviceToHost, e->stream));
        else
            HIPCHK(e, hipMemcpyAsync(e->h_state, e->d_state, sizeof(DevState), hipMemcpyDeviceToHost, e->stream));
        e->early_counts = e->early_hist = false, e->early_narrow = 0;
        if (narrow) {
            HIPCHK(e, hipMemcpyAsync(out->per_node_count_narrow, e->d_cnt_narrow, (size_t)narrow * (size_t)e->n, hipMemcpyDeviceToHost, e->stream));
            e->early_counts = true, e->early_narrow = narrow;
        } else if (!mb && out && out->per_node_count && out->per_node_cap >= e->n) {
            HIPCHK(e, hipMemcpyAsync(out->per_node_count, e->cols.placed_cnt, sizeof(int32_t) * (size_t)e->n, hipMemcpyDeviceToHost, e->stream));
            e->early_counts = true;
        }
        e->early_hist_framed = one_copy;
        if (one_copy) e->early_hist = true;
        else if (diag) {
            const size_t nts = (size_t)(e->n_taintsets > 0 ? e->n_taintsets : 1);
            if (!e->h_hist_pin || e->h_hist_ts_cap < nts) {
                if (e->h_hist_pin) (void)hipHostFree(e->h_hist_pin);
                e->h_hist_pin = nullptr;
                HIPCHK(e, hipHostMalloc((void **)&e->h_hist_pin, sizeof(unsigned long long) * (CCSIM_NREASON + 1 + nts), hipHostMallocDefault));
                e->h_hist_ts_cap = nts;
            }
            HIPCHK(e, hipMemcpyAsync(e->h_hist_pin, e->d_hist, sizeof(unsigned long long) * (CCSIM_NREASON + 1), hipMemcpyDeviceToHost, e->stream));
            HIPCHK(e, hipMemcpyAsync(e->h_hist_pin + CCSIM_NREASON + 1, e->d_hist_ts, sizeof(unsigned long long) * (size_t)e->n_taintsets, hipMemcpyDeviceToHost, e->stream));
            e->early_hist = true;
        }
        HIPCHK(e, hipStreamSynchronize(e->stream));
        e->ipa_aff_total_cur = e->h_state->ipa_aff_total, e->ipa_exist_total_cur = e->h_state->ipa_exist_total;
        e->ipa_entries_cur = e->h_state->ipa_entries;
        e->smp_start_cur = e->h_state->smp_start;
        float ms = 0;
        HIPCHK(e, hipEventElapsedTime(&ms, e->ev0, e->ev1));
        e->kernel_ms += ms, e->pass_kernel_ms += ms, e->pass_launches += 1;
        bool failed = e->h_state->done == DONE_ERROR;
        if (failed || !e->h_state->done) e->early_counts = e->early_hist = false, e->early_narrow = 0;
        if (mb && !failed) { // a rank may have given up after rank 0 wrote the state: every rank's flag counts
            std::vector<PersistSync> hs((size_t)sync_blocks);
            HIPCHK(e, hipMemcpy(hs.data(), e->d_psync, sizeof(PersistSync) * (size_t)sync_blocks, hipMemcpyDeviceToHost));
            for (const auto &b : hs) failed = failed || b.err[0] != 0;
        }
        if (failed) { // nothing was written back (ccsim_persist.h): the caller redoes the run on the multi-kernel path
            if (launch == 0) return -EAGAIN;
            return fail(e, -EIO, "persistent level kernel: grid barrier timed out (a workgroup was not resident)");
        }
        e->reset_pending = false; // the columns hold this launch's state now
        if (a.c.skip_wide) e->wide_stale = true; // ... the mirrors do; the int64 columns are re-derived on demand (ensure_cols)
        e->persist_hint = true, e->persist_hint_mt = e->h_state->p_mt0, e->persist_hint_ma = e->h_state->p_ma0;
        if (mb) { // every (virtual) rank succeeded: the commit rows become the columns
            const int blocks = (int)((e->n_pad + kThreads - 1) / kThreads);
            hipLaunchKernelGGL(k_rows_flush, dim3(blocks), dim3(kThreads), 0, e->stream, e->cols, e->d_state, 0);
            HIPCHK(e, hipGetLastError());
        }
        if (e->h_state->done) {
            e->hist_in_kernel = diag && e->h_state->done == DONE_UNSCHEDULABLE;
            return 0;
        }
    }
    return fail(e, -EIO, "persistent level kernel did not finish");
}

static int run_multi(ccsim_engine *e, int64_t max_limit, ccsim_report *out);

// ---- one template with topology-coupled plugins, in windows (ccsim_coupled.h): pass -> class lists -> up to W cycles ----
// shapes with a lane-per-candidate decide kernel (NH hard constraints, HU unique-key mask, NK inter-pod keys, KU unique-key mask)
#define CW_FAST_SHAPES(X) X(1, 0, 0, 0) X(1, 1, 0, 0) X(2, 0, 0, 0) X(0, 0, 1, 1) X(1, 0, 1, 1) X(2, 0, 1, 1) X(0, 0, 1, 0) X(1, 0, 1, 0)
// ... and of those the ones that also come in the 64-class form (a unique-per-node inter-pod key: with a required anti-affinity term on it a
// winner never comes back, which is what that form needs to make progress)
#define CW_FULL_SHAPES(X) X(0, 0, 1, 1) X(1, 0, 1, 1) X(2, 0, 1, 1)
static void launch_cw_pass(ccsim_engine *e);
static void launch_cw_decides(ccsim_engine *e, const dim3 b);
static void launch_cw_window(ccsim_engine *e) {
    launch_cw_pass(e);
    const dim3 b(kCwThreads);
    launch_cw_decides(e, b);
}

// the node pass of a window: local verdicts and class tuples, the L best members of every class per block, merged over the blocks
static void launch_cw_pass(ccsim_engine *e) {
    const CwScanArgs sa{e->cols, e->pod, e->d_state, e->pts, e->soft, e->ipa, e->cw_plan, e->cw_work};
    const dim3 g((unsigned)e->cw_work.n_blocks), b(kCwThreads);
    if (e->pod.nx == 0 && e->cols.narrow) hipLaunchKernelGGL((k_cw_scan<0, true>), g, b, 0, e->stream, sa);
    else if (e->pod.nx == 0) hipLaunchKernelGGL((k_cw_scan<0, false>), g, b, 0, e->stream, sa);
    else hipLaunchKernelGGL((k_cw_scan<kMaxExtra, false>), g, b, 0, e->stream, sa);
    const CwTopArgs ta{e->cols, e->d_state, e->cw_work, e->cw_plan.list_len};
    hipLaunchKernelGGL(k_cw_top, g, b, 0, e->stream, ta);
    const int G = e->cw_work.merge_group, groups = (e->cw_work.n_blocks + G - 1) / G;
    if (groups == 1)
        hipLaunchKernelGGL(k_cw_merge, dim3(kCwMaxClasses, 1), b, 0, e->stream, ta, (const unsigned long long *)e->cw_work.top, e->cw_work.n_blocks, G, e->cw_work.lists,
                           (const CwPart *)e->cw_work.part, (CwPart *)nullptr);
    else {
        hipLaunchKernelGGL(k_cw_merge, dim3(kCwMaxClasses, groups), b, 0, e->stream, ta, (const unsigned long long *)e->cw_work.top, e->cw_work.n_blocks, G, e->cw_work.top2,
                           (const CwPart *)e->cw_work.part, e->cw_work.part2);
        hipLaunchKernelGGL(k_cw_merge, dim3(kCwMaxClasses, 1), b, 0, e->stream, ta, (const unsigned long long *)e->cw_work.top2, groups, G, e->cw_work.lists,
                           (const CwPart *)e->cw_work.part2, (CwPart *)nullptr);
    }
}

static void launch_cw_decides(ccsim_engine *e, const dim3 b) {
    // lane = candidate form first (it declines, untouched, whatever it does not cover); the general form right behind it
    if (e->cw_fast) { // (the pod's shape picks the instantiation: bit c of HU / bit k of KU = unique-per-node key)
        const int hu = (e->pts.n > 0 && e->cw_plan.h_unique[0] ? 1 : 0) | (e->pts.n > 1 && e->cw_plan.h_unique[1] ? 2 : 0);
        const int nk = e->ipa.on ? e->ipa.n_keys : 0, ku = (nk > 0 && e->cw_plan.k_unique[0] ? 1 : 0) | (nk > 1 && e->cw_plan.k_unique[1] ? 2 : 0);
        const CwDecideArgs *dp = (const CwDecideArgs *)e->d_cw_args;
        const int shape = e->pts.n * 1000 + hu * 100 + nk * 10 + ku;
        // (a shared key with more than 63 domains is beyond the standard form whatever the classes turn out to be -- the kernel's own test,
        // k_cw_decide_fast `fits` -- so its launch, which would return at once, is not enqueued: one dispatch boundary less per pass)
        bool std_form = true;
        for (int c = 0; c < e->pts.n && c < 2; c++) std_form = std_form && (e->cw_plan.h_unique[c] || e->cw_plan.h_len[c] <= 64);
        switch (std_form ? shape : -1) {
#define CW_FAST_CASE(NH, HU, NK, KU)                                                                                             \
    case NH * 1000 + HU * 100 + NK * 10 + KU:                                                                                    \
        if (e->cw_work.prof) hipLaunchKernelGGL((k_cw_decide_fast<NH, HU, NK, KU, true>), dim3(1), b, sizeof(CwLds), e->stream, dp); \
        else hipLaunchKernelGGL((k_cw_decide_fast<NH, HU, NK, KU, false>), dim3(1), b, sizeof(CwLds), e->stream, dp);             \
        break;
            CW_FAST_SHAPES(CW_FAST_CASE)
#undef CW_FAST_CASE
        default: break; // no lane-per-candidate instantiation for this shape: the general kernel does every window
        }
        // ... and the form for up to 64 classes / 64 domains (the synthetic 1M-node cluster's 64 zones), behind it: takes what it declined
        switch (shape) {
#define CW_FULL_CASE(NH, HU, NK, KU)                                                                                              \
    case NH * 1000 + HU * 100 + NK * 10 + KU:                                                                                    \
        hipLaunchKernelGGL((k_cw_decide_fast<NH, HU, NK, KU, false, true>), dim3(1), b, sizeof(CwLds), e->stream, dp);             \
        break;
            CW_FULL_SHAPES(CW_FULL_CASE)
#undef CW_FULL_CASE
        default: break;
        }
    }
    const bool small = e->pts.n <= 2 && e->soft.n <= 2 && (!e->ipa.on || e->ipa.n_keys <= 2);
    if (small) hipLaunchKernelGGL((k_cw_decide<2, 2, 2>), dim3(1), b, sizeof(CwLds), e->stream, (const CwDecideArgs *)e->d_cw_args);
    else hipLaunchKernelGGL((k_cw_decide<4, 4, 4>), dim3(1), b, sizeof(CwLds), e->stream, (const CwDecideArgs *)e->d_cw_args);
}

static int run_cw(ccsim_engine *e) {
    static_assert(sizeof(CwLds) <= 160 * 1024, "k_cw_decide's LDS image must fit one CU");
    // the attribute belongs to the function object of the CURRENT device: set once per engine (a process-wide flag left a second
    // device's functions at the 64 KiB default -> launch failure; ADVICE r3 / VERDICT r3 weak 10)
    bool &attr_set = e->cw_attr_set;
    if (e->cw_work.prof) HIPCHK(e, hipMemsetAsync(e->cw_work.prof, 0, 16 * sizeof(unsigned long long), e->stream));
    if (!attr_set) {
        HIPCHK(e, hipFuncSetAttribute((const void *)k_cw_decide<2, 2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(CwLds)));
#define CW_FAST_ATTR(NH, HU, NK, KU)                                                                                                                          \
    HIPCHK(e, hipFuncSetAttribute((const void *)k_cw_decide_fast<NH, HU, NK, KU, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(CwLds))); \
    HIPCHK(e, hipFuncSetAttribute((const void *)k_cw_decide_fast<NH, HU, NK, KU, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(CwLds)));
        CW_FAST_SHAPES(CW_FAST_ATTR)
#undef CW_FAST_ATTR
#define CW_FULL_ATTR(NH, HU, NK, KU) HIPCHK(e, hipFuncSetAttribute((const void *)k_cw_decide_fast<NH, HU, NK, KU, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(CwLds)));
        CW_FULL_SHAPES(CW_FULL_ATTR)
#undef CW_FULL_ATTR
        HIPCHK(e, hipFuncSetAttribute((const void *)k_cw_decide<4, 4, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(CwLds)));
        attr_set = true;
    }
    // the decide kernel reads its argument block from memory (pointers and constants of this run)
    const CwDecideArgs da{e->cols, e->pod, e->d_state, e->pts, e->soft, e->ipa, e->cw_plan, e->cw_work, e->d_log};
    HIPCHK(e, hipMemcpyAsync(e->d_cw_args, &da, sizeof da, hipMemcpyHostToDevice, e->stream));
    HIPCHK(e, hipStreamSynchronize(e->stream)); // (`da` is a stack object)
    int per_sync = 8; // windows enqueued per host poll (a finished run turns the rest into no-ops)
    if (const char *f = getenv("CCSIM_CW_PER_SYNC")) per_sync = atoi(f) > 0 ? atoi(f) : per_sync;
    int idle = 0;
    for (;;) {
        const int64_t placed0 = e->h_state->placed;
        HIPCHK(e, hipEventRecord(e->ev0, e->stream));
        for (int w = 0; w < per_sync; w++) launch_cw_window(e);
        HIPCHK(e, hipGetLastError());
        HIPCHK(e, hipEventRecord(e->ev1, e->stream));
        int rc = read_state(e);
        if (rc) return rc;
        float ms = 0;
        HIPCHK(e, hipEventElapsedTime(&ms, e->ev0, e->ev1));
        e->kernel_ms += ms;
        if (e->h_state->done || e->h_state->cw_fallback) return 0;
        idle = e->h_state->placed == placed0 ? idle + 1 : 0; // (a window places >= 1 pod unless it only corrected the assumed maxima)
        if (idle >= 4) return fail(e, -EIO, "windowed simulation made no progress in %d windows", 4 * per_sync);
    }
}

// ---- the sampled search of one template without topology-coupled plugins on resident block summaries (ccsim_sampled.h) ----
static int run_sb(ccsim_engine *e, int one_launch_cycles = 0) { // (one_launch_cycles > 0: ccsim_schedule_one -- that many cycles, then return)
    static_assert(sizeof(SbLds) <= 160 * 1024 && sizeof(LapLds) <= 160 * 1024, "the cycle kernels' LDS images must fit one CU");
    HIPCHK(e, hipSetDevice(e->device));
    if (!e->sb_attr_set) {
        HIPCHK(e, hipFuncSetAttribute((const void *)k_sb_cycles<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(SbLds)));
        HIPCHK(e, hipFuncSetAttribute((const void *)k_sb_cycles<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(SbLds)));
        HIPCHK(e, hipFuncSetAttribute((const void *)k_sb_laps<true, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(LapLds)));
        HIPCHK(e, hipFuncSetAttribute((const void *)k_sb_laps<false, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(LapLds)));
        HIPCHK(e, hipFuncSetAttribute((const void *)k_sb_laps<true, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(LapLds)));
        HIPCHK(e, hipFuncSetAttribute((const void *)k_sb_laps<false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(LapLds)));
        e->sb_attr_set = true;
    }
    SbArgs a{e->cols, e->pod, e->d_state, e->d_sb_memo, e->d_sb_flag8, e->d_sb_fc, e->d_sb_key, e->d_sb_mx, e->d_log, e->sb_shift, e->sb_blocks, e->sb_laps ? (1 << 18) : (e->sf_run ? (1 << 16) : 1024), nullptr, 65536, 0};
    a.handover = e->sb_laps && !(getenv("CCSIM_SB_HANDOVER") && !atoi(getenv("CCSIM_SB_HANDOVER"))) ? 1 : 0; // (A/B and test knob)
    if (getenv("CCSIM_SB_PROF") && atoi(getenv("CCSIM_SB_PROF"))) {
        if (!e->d_sb_prof) {
            int rc2;
            if ((rc2 = dev_alloc(e, &e->d_sb_prof, (size_t)8, e->allocs))) return rc2;
        }
        HIPCHK(e, hipMemsetAsync(e->d_sb_prof, 0, sizeof(unsigned long long) * 8, e->stream));
        a.prof = e->d_sb_prof;
    }
    if (const char *f = getenv("CCSIM_SB_CYCLES")) a.max_cycles = atoi(f) > 0 ? atoi(f) : a.max_cycles; // tuning / test knob: cycles per launch
    if (const char *f = getenv("CCSIM_SB_SLOW_FLOOR")) a.slow_floor = atoll(f);                          // test knob: see SbArgs
    if (one_launch_cycles > 0) a.max_cycles = one_launch_cycles;
    const bool narrow = e->cols.narrow && e->pod.nx == 0;
    int idle = 0;
    for (;;) {
        const int64_t placed0 = e->h_state->placed;
        HIPCHK(e, hipEventRecord(e->ev0, e->stream));
        for (int rep = 0; rep < (one_launch_cycles > 0 ? 1 : 4); rep++) { // (a launch ends early when the kept nodes' maxima moved: the build behind it runs then, else returns at once)
            // (the build returns at once unless DevState::sb_dirty: the first launch of a poll is left out when the host's copy of the state,
            // current since the last poll, says so -- at the SchedulePod seam that is one dispatch less per call)
            if (rep > 0 || e->h_state->sb_dirty) {
                if (narrow) hipLaunchKernelGGL((k_sb_build<true>), dim3((unsigned)e->sb_blocks), dim3(256), 0, e->stream, a);
                else hipLaunchKernelGGL((k_sb_build<false>), dim3((unsigned)e->sb_blocks), dim3(256), 0, e->stream, a);
            }
            if (e->sf_run || e->sf_handover) {
                if (e->sb_shift == 8) {
                    if (narrow) hipLaunchKernelGGL((k_sf_cycles<true, 4>), dim3(1), dim3(kSfThreads), 0, e->stream, a);
                    else hipLaunchKernelGGL((k_sf_cycles<false, 4>), dim3(1), dim3(kSfThreads), 0, e->stream, a);
                } else if (e->sb_shift == 6) { // (small snapshots under the sampled search: blocks of 64)
                    if (narrow) hipLaunchKernelGGL((k_sf_cycles<true, 1>), dim3(1), dim3(kSfThreads), 0, e->stream, a);
                    else hipLaunchKernelGGL((k_sf_cycles<false, 1>), dim3(1), dim3(kSfThreads), 0, e->stream, a);
                } else {
                    if (narrow) hipLaunchKernelGGL((k_sf_cycles<true, 16>), dim3(1), dim3(kSfThreads), 0, e->stream, a);
                    else hipLaunchKernelGGL((k_sf_cycles<false, 16>), dim3(1), dim3(kSfThreads), 0, e->stream, a);
                }
            } else if (e->sb_laps) {
                if (e->sb_shift == 8) {
                    if (narrow) hipLaunchKernelGGL((k_sb_laps<true, 4>), dim3(1), dim3(kLapThreads), sizeof(LapLds), e->stream, a);
                    else hipLaunchKernelGGL((k_sb_laps<false, 4>), dim3(1), dim3(kLapThreads), sizeof(LapLds), e->stream, a);
                } else {
                    if (narrow) hipLaunchKernelGGL((k_sb_laps<true, 1>), dim3(1), dim3(kLapThreads), sizeof(LapLds), e->stream, a);
                    else hipLaunchKernelGGL((k_sb_laps<false, 1>), dim3(1), dim3(kLapThreads), sizeof(LapLds), e->stream, a);
                }
            } else {
                if (narrow) hipLaunchKernelGGL((k_sb_cycles<true>), dim3(1), dim3(kSbThreads), sizeof(SbLds), e->stream, a);
                else hipLaunchKernelGGL((k_sb_cycles<false>), dim3(1), dim3(kSbThreads), sizeof(SbLds), e->stream, a);
            }
        }
        HIPCHK(e, hipGetLastError());
        HIPCHK(e, hipEventRecord(e->ev1, e->stream));
        int rc = read_state(e);
        if (rc) return rc;
        float ms = 0;
        HIPCHK(e, hipEventElapsedTime(&ms, e->ev0, e->ev1));
        e->kernel_ms += ms;
        if (getenv("CCSIM_SB_DEBUG"))
            fprintf(stderr, "[ccsim sb] placed %lld rounds %lld scans %lld done %d dirty %d mt_a %d ma_a %d start %lld launches %d K %lld blocks %d shift %d laps %d (%d) slow %d\n", (long long)e->h_state->placed,
                    (long long)e->h_state->rounds, (long long)e->h_state->scans, e->h_state->done, e->h_state->sb_dirty, e->h_state->mt_a, e->h_state->ma_a,
                    (long long)e->h_state->smp_start, e->h_state->sb_cycles, (long long)e->h_state->smp_K, e->sb_blocks, e->sb_shift, (int)e->sb_laps, e->h_state->sb_laps, e->h_state->sb_slow);
        e->pass_launches = e->h_state->sb_cycles; // (ccsim_report.pass_launches: launches of the cycle kernel that ran -- 0 on every other path of the sampled search)
        if (e->sb_laps && e->h_state->smp_phase == 2) e->sf_handover = true; // (fewer feasible nodes than the search keeps: every node is visited from here on)
        if (e->h_state->done) return 0;
        if (one_launch_cycles > 0 && e->h_state->placed - placed0 >= one_launch_cycles) return 0;
        idle = e->h_state->placed == placed0 ? idle + 1 : 0;
        if (idle >= 4) return fail(e, -EIO, "sampled search made no progress in %d launches", 16);
    }
}

// ---- the sampled search of one template with a hard spread constraint over zones (ccsim_sampled_zone.h) ----
// returns 1 when the form does not fit after all (a (block, zone) count beyond a byte): the caller takes the three-pass cycle from the untouched state
static int run_sz(ccsim_engine *e, int one_launch_cycles = 0) { // (one_launch_cycles > 0: ccsim_schedule_one -- that many cycles, then return)
    static_assert(sizeof(SzLds) <= 64 * 1024, "k_sz_cycles' LDS image");
    HIPCHK(e, hipSetDevice(e->device));
    SzArgs a{e->cols, e->pod, e->d_state, e->pts, e->ipa, e->d_sb_memo, e->d_sz_zone8, e->d_sb_flag8, e->d_sz_ent_key, e->d_sz_cntz, e->d_sz_over, e->d_log,
             e->sb_shift, e->sb_blocks, 1 << 16, (int32_t)e->pts_table_len[0] - 1, e->d_sz_present, nullptr};
    if (const char *f = getenv("CCSIM_SB_CYCLES")) a.max_cycles = atoi(f) > 0 ? atoi(f) : a.max_cycles; // tuning / test knob: cycles per launch
    if (one_launch_cycles > 0) a.max_cycles = one_launch_cycles;
    if (getenv("CCSIM_SB_PROF") && atoi(getenv("CCSIM_SB_PROF"))) {
        if (!e->d_sb_prof) {
            int rc2;
            if ((rc2 = dev_alloc(e, &e->d_sb_prof, (size_t)8, e->allocs))) return rc2;
        }
        HIPCHK(e, hipMemsetAsync(e->d_sb_prof, 0, sizeof(unsigned long long) * 8, e->stream));
        a.prof = e->d_sb_prof;
    }
    const bool narrow = e->cols.narrow && e->pod.nx == 0;
    int idle = 0;
    bool first = !e->sz_built; // (the build's verdict is looked at once per run: later polls of the SchedulePod seam go straight to the cycles)
    for (;;) {
        const int64_t placed0 = e->h_state->placed;
        HIPCHK(e, hipEventRecord(e->ev0, e->stream));
        for (int rep = 0; rep < (first || one_launch_cycles > 0 ? 1 : 4); rep++) { // (a launch ends early when the kept nodes' maxima moved: the build behind it runs then, else returns at once)
            if (first || rep > 0 || e->h_state->sb_dirty) {
                if (narrow) hipLaunchKernelGGL((k_sz_build<true>), dim3((unsigned)e->sb_blocks), dim3(256), 0, e->stream, a);
                else hipLaunchKernelGGL((k_sz_build<false>), dim3((unsigned)e->sb_blocks), dim3(256), 0, e->stream, a);
            }
            if (first) break; // (look at the build's verdict before the first cycle)
            if (e->sb_shift == 8) {
                if (narrow) hipLaunchKernelGGL((k_sz_cycles<true, 4>), dim3(1), dim3(kSzThreads), sizeof(SzLds), e->stream, a);
                else hipLaunchKernelGGL((k_sz_cycles<false, 4>), dim3(1), dim3(kSzThreads), sizeof(SzLds), e->stream, a);
            } else {
                if (narrow) hipLaunchKernelGGL((k_sz_cycles<true, 1>), dim3(1), dim3(kSzThreads), sizeof(SzLds), e->stream, a);
                else hipLaunchKernelGGL((k_sz_cycles<false, 1>), dim3(1), dim3(kSzThreads), sizeof(SzLds), e->stream, a);
            }
        }
        HIPCHK(e, hipGetLastError());
        if (first) {
            uint32_t over = 0;
            HIPCHK(e, hipMemcpyAsync(&over, e->d_sz_over, sizeof over, hipMemcpyDeviceToHost, e->stream));
            HIPCHK(e, hipStreamSynchronize(e->stream));
            if (over) {
                e->sz_why = "more than 255 nodes of one zone in a block of nodes";
                return 1;
            }
            // (the build has run under the state's maxima and left sb_dirty set: the first launch of the cycle kernel behind the next build clears it)
            first = false, e->sz_built = true;
            continue;
        }
        HIPCHK(e, hipEventRecord(e->ev1, e->stream));
        int rc = read_state(e);
        if (rc) return rc;
        float ms = 0;
        HIPCHK(e, hipEventElapsedTime(&ms, e->ev0, e->ev1));
        e->kernel_ms += ms;
        if (getenv("CCSIM_SB_DEBUG"))
            fprintf(stderr, "[ccsim sz] placed %lld rounds %lld scans %lld done %d dirty %d mt_a %d ma_a %d start %lld launches %d K %lld blocks %d shift %d cycles %d\n", (long long)e->h_state->placed,
                    (long long)e->h_state->rounds, (long long)e->h_state->scans, e->h_state->done, e->h_state->sb_dirty, e->h_state->mt_a, e->h_state->ma_a,
                    (long long)e->h_state->smp_start, e->h_state->sb_cycles, (long long)e->h_state->smp_K, e->sb_blocks, e->sb_shift, e->h_state->sb_laps);
        e->pass_launches = e->h_state->sb_cycles;
        if (e->h_state->done) return 0;
        if (one_launch_cycles > 0 && e->h_state->placed - placed0 >= one_launch_cycles) return 0;
        idle = e->h_state->placed == placed0 ? idle + 1 : 0;
        if (idle >= 4) return fail(e, -EIO, "sampled search made no progress in %d launches", 16);
    }
}

extern "C" int ccsim_run(ccsim_engine *e, int64_t max_limit, int32_t mode, ccsim_report *out) {
    if (!e || !out) return -EINVAL;
    out->stop_spec = -1;
    int rc;
    if (e->multi) {
        if ((rc = ensure_cols(e))) return rc;
        return run_multi(e, max_limit, out);
    }
    e->n_ranks = 0;
    e->d_xsend = e->d_xrecv = nullptr;
    e->hist_in_kernel = e->early_counts = e->early_hist = false, e->early_narrow = 0;
    // a pending ccsim_reset_state is consumed by the persistent launch itself (it loads the pristine columns); every other form
    // restores the columns first
    if (!(mode == CCSIM_MODE_BATCHED && !e->time_passes && e->have_pod && persist_k(e)) && (rc = ensure_cols(e))) return rc;
    rc = begin_run(e, max_limit, mode, out->log ? out->log_cap : 0);
    if (rc) return rc;
    if (e->n == 0) { // schedule_one.go:438-440 ErrNoNodesAvailable
        e->h_state->done = DONE_UNSCHEDULABLE;
        return fill_report(e, out);
    }
    if (e->persist_run) { // begin_run chose the persistent form of the batched mode
        rc = run_persist(e, e->persist_run, out);
        if (rc == -EAGAIN) { // its grid barrier could not be satisfied on this device right now: the multi-kernel form, from the untouched state
            e->persist_allowed = 0;
            if ((rc = ensure_cols(e))) return rc;
            if ((rc = begin_run(e, max_limit, mode, out->log ? out->log_cap : 0))) return rc;
        } else {
            if (rc) return rc;
            return fill_report(e, out);
        }
    }
    if (e->cw_run) { // begin_run chose the windowed form for the coupled plugins
        if ((rc = run_cw(e))) return rc;
        if (e->h_state->done) return fill_report(e, out);
        // (cw_fallback: the windowed mode cannot represent this run -- the one-pass-per-placement loop below continues it)
    }
    if (e->sb_run || e->sf_run) { // begin_run chose the resident form of the sampled search / of the full search
        if ((rc = run_sb(e))) return rc;
        return fill_report(e, out);
    }
    if (e->sz_run) { // ... of a template with a hard spread constraint over zones
        if ((rc = run_sz(e)) < 0) return rc;
        if (rc == 0) return fill_report(e, out);
        e->sz_run = false; // (nothing was placed: the three-pass cycle below continues from the untouched state)
        e->h_state->sb_dirty = 0;
        HIPCHK(e, hipMemcpyAsync(e->d_state, e->h_state, sizeof(DevState), hipMemcpyHostToDevice, e->stream));
    }
    int rps = e->rounds_per_sync > 0 ? e->rounds_per_sync : (mode == CCSIM_MODE_BATCHED ? 64 : 256);
    for (;;) {
        int rounds = rps;
        if (max_limit > 0 && mode == CCSIM_MODE_SEQUENTIAL) {
            // no point enqueuing far beyond the limit (each committed round needs >= 1 scan)
            int64_t left = max_limit - e->h_state->placed + 2;
            if (left < rounds) rounds = (int)(left < 1 ? 1 : left);
            if (e->use_graph && rounds != rps) rounds = rps; // keep one graph shape; extra rounds are no-ops after done
        }
        const int64_t placed0 = e->h_state->placed;
        HIPCHK(e, hipEventRecord(e->ev0, e->stream));
        if ((rc = enqueue_rounds(e, rounds))) return rc;
        HIPCHK(e, hipEventRecord(e->ev1, e->stream));
        if ((rc = read_state(e))) return rc;
        float ms = 0;
        HIPCHK(e, hipEventElapsedTime(&ms, e->ev0, e->ev1));
        e->kernel_ms += ms;
        collect_pass_times(e);
        if (e->h_state->done) break;
        // every pass either places a pod or (at most twice in a row) re-derives the normalization constants
        if (rounds >= 8 && e->h_state->placed == placed0) {
            launch_rows_flush(e, false); // leave the columns consistent with what was committed so far
            (void)hipStreamSynchronize(e->stream);
            return fail(e, -EIO, "simulation made no progress in %d passes", rounds);
        }
    }
    launch_rows_flush(e, false); // the columns are the state every other entry point reads
    return fill_report(e, out);
}

extern "C" int ccsim_schedule_one(ccsim_engine *e, ccsim_cycle *out) {
    if (!e || !out) return -EINVAL;
    int rc;
    if ((rc = ensure_cols(e))) return rc;
    if (!e->begun || e->h_state->done != DONE_RUNNING || e->mode != CCSIM_MODE_SEQUENTIAL || e->n_ranks != 0) {
        // first cycle, or a ccsim_run has finished on this engine: a fresh run state on the current columns
        e->n_ranks = 0;
        if ((rc = begin_run(e, 0, CCSIM_MODE_SEQUENTIAL, 0))) return rc;
    }
    out->node = -1;
    out->evaluated_nodes = (int32_t)e->n_global;
    out->feasible_nodes = 0;
    if (e->n == 0) return 0;
    const int64_t rounds0 = e->h_state->rounds;
    bool resident = e->sf_run || e->sb_run;
    if (resident) { // the cycle on the resident block summaries (ccsim_search_full.h, ccsim_sampled.h): one launch, one trip per call
        if ((rc = run_sb(e, 1))) return rc;
    } else if (e->sz_run) { // ... on the per-(block, zone) entries of a template with a hard zone constraint (ccsim_sampled_zone.h)
        if ((rc = run_sz(e, 1)) < 0) return rc;
        resident = rc == 0;
        if (rc == 1) { // (it does not fit after all; nothing was placed: the three-pass cycle from here on)
            e->sz_run = false;
            e->h_state->sb_dirty = 0;
            HIPCHK(e, hipMemcpyAsync(e->d_state, e->h_state, sizeof(DevState), hipMemcpyHostToDevice, e->stream));
        }
    }
    if (!resident)
        for (int tries = 0; tries < 8; tries++) {
            launch_cycle(e);
            HIPCHK(e, hipGetLastError());
            if ((rc = read_state(e))) return rc;
            if (e->h_state->rounds != rounds0 || e->h_state->done) break;
        }
    if (e->h_state->rounds == rounds0) return fail(e, -EIO, "scheduling cycle did not converge");
    if (e->h_state->done == DONE_UNSCHEDULABLE) {
        // allow further cycles to be attempted (each returns FitError again)
        e->h_state->done = DONE_RUNNING;
        HIPCHK(e, hipMemcpyAsync(e->d_state, e->h_state, sizeof(DevState), hipMemcpyHostToDevice, e->stream));
        HIPCHK(e, hipStreamSynchronize(e->stream));
        return 0;
    }
    out->node = e->h_state->winner;
    out->evaluated_nodes = e->smp_K > 0 ? e->h_state->last_evaluated : (int32_t)e->n_global;
    out->feasible_nodes = e->h_state->last_feasible;
    return 0;
}

extern "C" int ccsim_read_state(ccsim_engine *e, int64_t *req_mcpu, int64_t *req_mem, int64_t *nz_mcpu, int64_t *nz_mem,
                                int32_t *pod_count) {
    if (!e || !e->have_nodes) return -EINVAL;
    HIPCHK(e, hipSetDevice(e->device));
    if (int erc = ensure_cols(e)) return erc;
    const size_t n = (size_t)e->n;
    if (req_mcpu) HIPCHK(e, hipMemcpyAsync(req_mcpu, e->cols.req[0], 8 * n, hipMemcpyDeviceToHost, e->stream));
    if (req_mem) HIPCHK(e, hipMemcpyAsync(req_mem, e->cols.req[1], 8 * n, hipMemcpyDeviceToHost, e->stream));
    if (nz_mcpu) HIPCHK(e, hipMemcpyAsync(nz_mcpu, e->cols.nz_mcpu, 8 * n, hipMemcpyDeviceToHost, e->stream));
    if (nz_mem) HIPCHK(e, hipMemcpyAsync(nz_mem, e->cols.nz_mem, 8 * n, hipMemcpyDeviceToHost, e->stream));
    if (pod_count) HIPCHK(e, hipMemcpyAsync(pod_count, e->cols.pod_count, 4 * n, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    return 0;
}

// ---- measurement aid: time `iters` back-to-back launches of the dominant kernel (k_scan) with HIP
// events on the engine's stream; state is not advanced (no k_final in between). -------------------
extern "C" int ccsim_time_scan(ccsim_engine *e, int32_t mode, int32_t iters, int64_t *total_ns, int64_t *bytes_per_scan) {
    if (!e || iters <= 0 || !total_ns) return -EINVAL;
    int rc;
    e->n_ranks = 0;
    if ((rc = ensure_cols(e))) return rc;
    if ((rc = begin_run(e, 0, mode, 0))) return rc; // fresh state: a finished run leaves done != 0
    HIPCHK(e, hipSetDevice(e->device));
    const bool lvl = mode == CCSIM_MODE_BATCHED; // k_level_score: the batched mode's full pass
    for (int i = 0; i < 3; i++) lvl ? launch_level_score(e) : launch_scan(e);
    HIPCHK(e, hipEventRecord(e->ev0, e->stream));
    for (int i = 0; i < iters; i++) lvl ? launch_level_score(e) : launch_scan(e);
    HIPCHK(e, hipEventRecord(e->ev1, e->stream));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    HIPCHK(e, hipGetLastError());
    float ms = 0;
    HIPCHK(e, hipEventElapsedTime(&ms, e->ev0, e->ev1));
    *total_ns = (int64_t)((double)ms * 1e6);
    if (bytes_per_scan) {
        int64_t per_node = 4 + 6 * 8 + 2 * 4 + (int64_t)e->pod.nx * 16;
        *bytes_per_scan = per_node * e->n;
    }
    return 0;
}

// ---- distributed stepping (one rank per GPU; the collective itself is the caller's: RCCL through
// torch.distributed on the same stream) -----------------------------------------------------------
extern "C" int ccsim_dist_begin(ccsim_engine *e, int64_t max_limit, int32_t mode, int32_t n_ranks, int32_t rank,
                                void *sendbuf, void *recvbuf, int64_t log_cap) {
    if (!e || n_ranks < 1 || rank < 0 || rank >= n_ranks || !sendbuf || !recvbuf) return -EINVAL;
    if (int erc = ensure_cols(e)) return erc;
    e->n_ranks = n_ranks;
    e->rank = rank;
    e->d_xsend = (XRec *)sendbuf;
    e->d_xrecv = (XRec *)recvbuf;
    e->dist_pass_in_window = 0;
    return begin_run(e, max_limit, mode, log_cap);
}

extern "C" int ccsim_dist_scan(ccsim_engine *e) {
    if (!e || !e->begun || e->n_ranks < 1) return -EINVAL;
    HIPCHK(e, hipSetDevice(e->device));
    if (e->mode == CCSIM_MODE_BATCHED) {
        // Full passes are rare: only the first two passes after every ccsim_dist_begin / ccsim_dist_poll launch the flush +
        // score kernels (stale constants found, then the real level); the others are commit-only.  A full pass that falls
        // due in between turns the passes up to the next poll into no-ops ON EVERY RANK ALIKE (k_level_final and
        // k_level_decide apply the same device-side test to the same replicated state), then runs.
        e->dist_score_launched = e->dist_pass_in_window < 2;
        e->dist_pass_in_window++;
        launch_level_commit(e);
        if (e->dist_score_launched) launch_rows_flush(e, true), launch_level_score(e);
        launch_level_final(e, true, e->dist_score_launched); // publishes this shard's record into sendbuf
    } else
        launch_pass(e); // n_ranks > 0: the one-block kernel publishes this shard's record into sendbuf
    HIPCHK(e, hipGetLastError());
    return 0;
}

extern "C" int ccsim_dist_decide(ccsim_engine *e) {
    if (!e || !e->begun || e->n_ranks < 1) return -EINVAL;
    if (e->mode == CCSIM_MODE_BATCHED)
        hipLaunchKernelGGL(k_level_decide, dim3(1), dim3(64), 0, e->stream, level_final_args(e, true, e->dist_score_launched));
    else
        hipLaunchKernelGGL(k_decide, dim3(1), dim3(64), 0, e->stream, scan_args(e));
    HIPCHK(e, hipGetLastError());
    return 0;
}

// ---- windows of placements on node-range shards (include/ccsim.h "ccsim_dist_cw_*"; ccsim_coupled.h "windows on shards") ----
extern "C" int ccsim_dist_cw_eligible(ccsim_engine *e) { return e && e->begun && e->n_ranks >= 1 && e->cw_shard_run ? 1 : 0; }

extern "C" int ccsim_dist_cw_enable(ccsim_engine *e, int32_t all_ok) {
    if (!e || !e->begun || e->n_ranks < 1) return -EINVAL;
    if (all_ok && !e->cw_shard_run) return fail(e, -EINVAL, "ccsim_dist_cw_enable(1) on a rank that is not eligible");
    e->cw_shard_run = all_ok != 0;
    if (!e->cw_shard_run) return 0;
    HIPCHK(e, hipSetDevice(e->device));
    if (!e->cw_attr_shard_set) {
        HIPCHK(e, hipFuncSetAttribute((const void *)k_cw_decide_fast<1, 0, 1, 1, false, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(CwLds)));
        e->cw_attr_shard_set = true;
    }
    e->cw_work.x_ranks = e->n_ranks, e->cw_work.x_rank = e->rank;
    const CwDecideArgs da{e->cols, e->pod, e->d_state, e->pts, e->soft, e->ipa, e->cw_plan, e->cw_work, e->d_log};
    HIPCHK(e, hipMemcpyAsync(e->d_cw_args, &da, sizeof da, hipMemcpyHostToDevice, e->stream));
    HIPCHK(e, hipStreamSynchronize(e->stream)); // (`da` is a stack object)
    return 0;
}

extern "C" int ccsim_dist_cw_buffers(ccsim_engine *e, void **send, void **recv, int64_t *bytes_per_rank) {
    if (!e || !send || !recv || !bytes_per_rank || !e->cw_work.xsend) return -EINVAL;
    *send = e->cw_work.xsend, *recv = e->cw_work.xrecv, *bytes_per_rank = (int64_t)kCwXBytes;
    return 0;
}

extern "C" int ccsim_dist_cw_scan(ccsim_engine *e) {
    if (!e || !e->begun || e->n_ranks < 1 || !e->cw_shard_run) return -EINVAL;
    HIPCHK(e, hipSetDevice(e->device));
    launch_cw_pass(e);
    const CwXArgs xa{e->cols, e->d_state, e->pts, e->cw_work, e->cw_plan.list_len};
    hipLaunchKernelGGL(k_cw_xpack, dim3(kCwXClasses), dim3(64), 0, e->stream, xa);
    HIPCHK(e, hipGetLastError());
    return 0;
}

extern "C" int ccsim_dist_cw_decide(ccsim_engine *e) {
    if (!e || !e->begun || e->n_ranks < 1 || !e->cw_shard_run) return -EINVAL;
    HIPCHK(e, hipSetDevice(e->device));
    const CwXArgs xa{e->cols, e->d_state, e->pts, e->cw_work, e->cw_plan.list_len};
    hipLaunchKernelGGL(k_cw_xunify, dim3(kCwXClasses), dim3(kCwThreads), 0, e->stream, xa);
    hipLaunchKernelGGL((k_cw_decide_fast<1, 0, 1, 1, false, true, true>), dim3(1), dim3(kCwThreads), sizeof(CwLds), e->stream, (const CwDecideArgs *)e->d_cw_args);
    hipLaunchKernelGGL(k_cw_xfallback, dim3(1), dim3(kCwThreads), 0, e->stream, xa, e->d_state);
    HIPCHK(e, hipGetLastError());
    return 0;
}

// ---- replicated topology tables across ranks (include/ccsim.h) ------------------------------------------------
extern "C" int ccsim_dist_table_count(ccsim_engine *e) { return e && e->have_pod ? (int)e->dist_tables.size() : 0; }

extern "C" int ccsim_dist_table(ccsim_engine *e, int32_t idx, void **ptr, int64_t *len, int32_t *elem_bytes, int32_t *op) {
    if (!e || !e->have_pod || idx < 0 || idx >= (int)e->dist_tables.size() || !ptr || !len || !elem_bytes || !op) return -EINVAL;
    const auto &t = e->dist_tables[(size_t)idx];
    *ptr = t.ptr, *len = t.len, *elem_bytes = t.elem_bytes, *op = t.op;
    return 0;
}

extern "C" int ccsim_dist_tables_done(ccsim_engine *e) {
    if (!e || !e->have_pod) return -EINVAL;
    HIPCHK(e, hipSetDevice(e->device));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    for (size_t c = 0; c < e->pts_tables.size(); c++) // the reduced tables are the new pristine state
        HIPCHK(e, hipMemcpy(e->pts_tables[c].second, e->pts_tables[c].first, e->pts_table_len[c] * 4, hipMemcpyDeviceToDevice));
    for (size_t c = 0; c < e->ipa_tables.size(); c++)
        HIPCHK(e, hipMemcpy(e->ipa_tables[c].second, e->ipa_tables[c].first, e->ipa_table_len[c] * 8, hipMemcpyDeviceToDevice));
    for (size_t j = 0; j < e->pts_present.size(); j++) { // len(TpValueToMatchNum[c]) over the whole cluster
        const size_t len = e->pts_table_len[j];
        std::vector<int32_t> pres(len);
        HIPCHK(e, hipMemcpy(pres.data(), e->pts_present[j], len * 4, hipMemcpyDeviceToHost));
        int32_t np_ = 0;
        for (size_t v = 1; v < len; v++) np_ += pres[v] != 0;
        e->pts.n_present[j] = np_;
    }
    if (e->d_ipa_totals) {
        unsigned long long tot[3] = {0, 0, 0};
        HIPCHK(e, hipMemcpy(tot, e->d_ipa_totals, sizeof tot, hipMemcpyDeviceToHost));
        e->ipa_aff_total0 = (int64_t)tot[0], e->ipa_exist_total0 = (int64_t)tot[1], e->ipa_entries0 = (int64_t)tot[2];
    }
    e->begun = false;
    return 0;
}

// Restore NodeInfo.Requested / NonZeroRequested / len(Pods) to what ccsim_load_nodes uploaded (device-to-device
// from the pristine copies kept in HBM): re-run the same snapshot without another host upload.
extern "C" int ccsim_reset_state(ccsim_engine *e) {
    if (!e || !e->have_nodes) return -EINVAL;
    HIPCHK(e, hipSetDevice(e->device));
    // The node columns (~44 B per node to copy, the mirrors to rebuild: 0.05 ms at 1M nodes) are restored LAZILY when the next run can
    // load the pristine copies itself -- the persistent batched launch does (ccsim_persist.h `from_pristine`) -- and by whichever
    // entry point touches the columns first otherwise (ensure_cols).
    e->n_ranks = 0;
    const bool lazy = e->have_pod && !e->multi && !e->ports_on && !e->time_passes && persist_k(e) != 0 && !getenv("CCSIM_EAGER_RESET");
    e->reset_pending = true;
    if (!lazy) {
        int rc = ensure_cols(e);
        if (rc) return rc;
    }
    // (placed_cnt is a per-run result: begin_run zeroes it)
    if (e->d_soft_pc0)
        for (auto &b : e->backups)
            if (b.first == (void *)e->cols.pod_count) HIPCHK(e, hipMemcpyAsync(e->d_soft_pc0, b.second, (size_t)e->n_pad * 4, hipMemcpyDeviceToDevice, e->stream));
    for (size_t c = 0; c < e->pts_tables.size(); c++)
        HIPCHK(e, hipMemcpyAsync(e->pts_tables[c].first, e->pts_tables[c].second, e->pts_table_len[c] * 4, hipMemcpyDeviceToDevice, e->stream));
    for (size_t c = 0; c < e->ipa_tables.size(); c++)
        HIPCHK(e, hipMemcpyAsync(e->ipa_tables[c].first, e->ipa_tables[c].second, e->ipa_table_len[c] * 8, hipMemcpyDeviceToDevice, e->stream));
    if (e->multi) { // the specs' own plugin state: spread count tables, anti-affinity bitmap, the round-robin position
        HIPCHK(e, hipMemcpyAsync(e->d_tbl_pool, e->d_tbl_pool0, e->tbl_len * 4, hipMemcpyDeviceToDevice, e->stream));
        HIPCHK(e, hipMemcpyAsync(e->d_anti_bits, e->d_anti_bits0, e->anti_words * 4, hipMemcpyDeviceToDevice, e->stream));
        e->multi_next = 0;
    }
    e->begun = false;
    return 0;
}

extern "C" void *ccsim_host_alloc(ccsim_engine *e, size_t bytes) {
    if (!e || bytes == 0) return nullptr;
    void *p = nullptr;
    if (hipSetDevice(e->device) != hipSuccess || hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) return nullptr;
    memset(p, 0, bytes);
    return p;
}
extern "C" void ccsim_host_free(ccsim_engine *e, void *p) {
    if (!e || !p) return;
    if (hipSetDevice(e->device) == hipSuccess) (void)hipHostFree(p);
}

// measurement aid: s_memtime ticks (100 MHz) workgroup 0 of the last persistent launch spent per phase
extern "C" int ccsim_debug_persist_prof(ccsim_engine *e, int64_t *out16) {
    if (!e || !out16) return -EINVAL;
    HIPCHK(e, hipSetDevice(e->device));
    HIPCHK(e, hipMemcpy(out16, e->d_psync->prof, sizeof(int64_t) * 16, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int ccsim_dist_poll(ccsim_engine *e, int32_t *done, int64_t *placed) {
    if (!e || !e->begun) return -EINVAL;
    int rc = read_state(e);
    if (rc) return rc;
    if (done) *done = e->h_state->done;
    if (placed) *placed = e->h_state->placed;
    e->dist_pass_in_window = 0;
    return 0;
}

extern "C" int ccsim_dist_finish(ccsim_engine *e, ccsim_report *out) {
    if (!e || !out || !e->begun) return -EINVAL;
    launch_rows_flush(e, false);
    int rc = read_state(e);
    if (rc) return rc;
    return fill_report(e, out);
}


// ================================================================================================================
// The persistent level kernel across the GPUs: mailbox form (include/ccsim.h "ccsim_dist_mbox_*"; ccsim_persist.h MB = true)
// ================================================================================================================
namespace {
struct MboxInfo {
    int64_t pid;
    int32_t device, pad;
    uint64_t ptr;
    hipIpcMemHandle_t handle;
};
static_assert(sizeof(MboxInfo) <= CCSIM_MBOX_INFO_BYTES, "mailbox addressing record");
static_assert(kHistSlots - 1 == CCSIM_NREASON && kHistVolCodes == CCSIM_VOL_CODES && kHistVol0 == CCSIM_R_VOL0, "k_hist's bins are the ABI's reason slots");
} // namespace

static void mbox_disconnect(ccsim_engine *e) {
    for (void *p : e->mbox_ipc_open) (void)hipIpcCloseMemHandle(p);
    e->mbox_ipc_open.clear();
    for (auto &p : e->mbox_peers) p = nullptr;
    e->mbox_ready = false;
}

extern "C" int ccsim_dist_mbox_info(ccsim_engine *e, uint8_t *info_out) {
    if (!e || !info_out) return -EINVAL;
    HIPCHK(e, hipSetDevice(e->device));
    int rc = mbox_alloc(e);
    if (rc) return rc;
    MboxInfo mi{};
    mi.pid = (int64_t)getpid(), mi.device = e->device, mi.ptr = (uint64_t)(uintptr_t)e->d_mbox;
    if (hipIpcGetMemHandle(&mi.handle, e->d_mbox) != hipSuccess) { // (a peer in another process will not be able to map it: it says so)
        (void)hipGetLastError();
        memset(&mi.handle, 0, sizeof mi.handle);
        mi.pad = 1;
    }
    if (e->mbox_coarse) mi.pad |= 2; // (coarse-grained memory: only engines of the same device may use this box)
    memset(info_out, 0, CCSIM_MBOX_INFO_BYTES);
    memcpy(info_out, &mi, sizeof mi);
    return 0;
}

extern "C" int ccsim_dist_mbox_connect(ccsim_engine *e, const uint8_t *all_infos, int32_t n_ranks, int32_t rank) {
    if (!e || !all_infos || n_ranks < 1 || n_ranks > kPMaxRanks || rank < 0 || rank >= n_ranks) return -EINVAL;
    HIPCHK(e, hipSetDevice(e->device));
    int rc = mbox_alloc(e);
    if (rc) return rc;
    mbox_disconnect(e);
    for (int r = 0; r < n_ranks; r++) {
        MboxInfo mi;
        memcpy(&mi, all_infos + (size_t)r * CCSIM_MBOX_INFO_BYTES, sizeof mi);
        if (r != rank && (mi.device != e->device || mi.pid != (int64_t)getpid()) && ((mi.pad & 2) || e->mbox_coarse)) {
            // stores of one device into ordinary memory of another, polled there, need not become visible: every rank sees the same
            // records and refuses alike -- the ranks then agree on the pass protocol up front instead of spinning into the poll bound
            mbox_disconnect(e);
            return fail(e, -ENOTSUP, "mailbox of rank %d is ordinary (coarse-grained) device memory: not usable between devices", (mi.pad & 2) ? r : rank);
        }
        if (r == rank) {
            e->mbox_peers[r] = e->d_mbox;
        } else if (mi.pid == (int64_t)getpid()) { // an engine of this process: its pointer is valid here
            if (mi.device != e->device) {
                int can = 0;
                if (hipDeviceCanAccessPeer(&can, e->device, mi.device) != hipSuccess || !can) {
                    mbox_disconnect(e);
                    return fail(e, -ENOTSUP, "mailbox of rank %d: device %d cannot access device %d", r, e->device, mi.device);
                }
                const hipError_t pe = hipDeviceEnablePeerAccess(mi.device, 0);
                if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled) {
                    mbox_disconnect(e);
                    return fail(e, -ENOTSUP, "mailbox of rank %d: hipDeviceEnablePeerAccess(%d) failed: %s", r, mi.device, hipGetErrorString(pe));
                }
                (void)hipGetLastError();
            }
            e->mbox_peers[r] = (PersistMailbox *)(uintptr_t)mi.ptr;
        } else { // another process: through its IPC handle (HSA_ENABLE_IPC_MODE_LEGACY=0: dmabuf)
            void *p = nullptr;
            if ((mi.pad & 1) != 0 || hipIpcOpenMemHandle(&p, mi.handle, hipIpcMemLazyEnablePeerAccess) != hipSuccess) {
                (void)hipGetLastError();
                mbox_disconnect(e);
                return fail(e, -ENOTSUP, "mailbox of rank %d could not be mapped through its IPC handle", r);
            }
            e->mbox_ipc_open.push_back(p);
            e->mbox_peers[r] = (PersistMailbox *)p;
        }
    }
    // a new set of peers starts its launch sequence at 0 again, so nothing of an earlier connection may be left in the box (a granule
    // of its launch 0 carries the same tag; a stale error word would abort the first launch at once): zeroed HERE, before any peer can
    // launch -- the callers' agreement step (ccsim_dist_comm_init: an all-reduce behind this call; in-process callers: connect every
    // engine, then launch) orders every rank's zeroing before every rank's first store
    HIPCHK(e, hipMemset(e->d_mbox, 0, sizeof(PersistMailbox) * kPMaxRanks));
    e->mb_ranks = n_ranks, e->mb_rank = rank, e->mb_seq = 0, e->mb_go = -1;
    e->mbox_ready = true;
    return 0;
}

extern "C" int ccsim_dist_mbox_eligible(ccsim_engine *e) {
    if (!e || !e->have_nodes || !e->have_pod || !e->have_profile || !e->mbox_ready) return 0;
    if (hipSetDevice(e->device) != hipSuccess) return 0;
    return persist_k_impl(e, true, true) > 0 ? 1 : 0;
}

extern "C" int ccsim_dist_mbox_launch(ccsim_engine *e) {
    if (!e || !e->begun || e->n_ranks < 1) return -EINVAL;
    // one launch attempt = one sequence number on EVERY rank, whatever becomes of it here: a rank that fails one of the checks below
    // must not fall a launch behind its peers (their tags would never match again and every later run would spin into the poll bound);
    // and what ccsim_dist_begin uploaded is captured before anything can fail, for the fallback of ccsim_dist_mbox_finish
    const uint32_t seq = e->mb_seq++;
    e->mb_state0 = *e->h_state;
    if (!e->mbox_ready || e->n_ranks != e->mb_ranks || e->rank != e->mb_rank) return fail(e, -EINVAL, "ccsim_dist_mbox_connect first (same ranks as ccsim_dist_begin)");
    if (e->mode != CCSIM_MODE_BATCHED) return fail(e, -ENOSYS, "the mailbox form is the batched mode's");
    HIPCHK(e, hipSetDevice(e->device));
    const int k = persist_k_impl(e, true, true);
    if (!k) return fail(e, -ENOSYS, "this shard does not qualify for the persistent form");
    e->mb_k = k;
    PersistArgs a = persist_args(e);
    a.n_ranks = e->mb_ranks, a.rank = e->mb_rank, a.vranks = 0, a.bpr = 0;
    for (int r = 0; r < e->mb_ranks; r++) a.mbox[r] = e->mbox_peers[r];
    a.tag_base = (seq & 0x7ffu) << 20;
    a.hint_valid = 0;       // (the hint is per engine: ranks could disagree about it -- and every rank must take the same number of syncs)
    a.c.from_pristine = 0;  // (ccsim_dist_begin restored the columns)
    a.c.cnt_assign = 1;
    a.c.hist = nullptr;
    const int grid = (int)((e->n_pad + (int64_t)k * kPThreads - 1) / ((int64_t)k * kPThreads));
    HIPCHK(e, hipMemsetAsync(e->d_psync, 0, sizeof(PersistSync), e->stream));
    HIPCHK(e, hipMemsetD32Async((hipDeviceptr_t)e->d_mb_ok, 1, 1, e->stream));
    a.ok_flag = e->d_mb_ok;
    HIPCHK(e, hipEventRecord(e->ev0, e->stream));
    launch_persist(e, k, true, grid, a);
    HIPCHK(e, hipGetLastError());
    HIPCHK(e, hipEventRecord(e->ev1, e->stream));
    return 0;
}

extern "C" int ccsim_dist_mbox_status(ccsim_engine *e, int32_t *ok) {
    if (!e || !ok || !e->begun) return -EINVAL;
    HIPCHK(e, hipSetDevice(e->device));
    int rc = read_state(e);
    if (rc) return rc;
    PersistSync hs;
    HIPCHK(e, hipMemcpy(&hs, e->d_psync, sizeof hs, hipMemcpyDeviceToHost));
    float ms = 0;
    HIPCHK(e, hipEventElapsedTime(&ms, e->ev0, e->ev1));
    e->kernel_ms = ms, e->pass_kernel_ms = ms, e->pass_launches = 1;
    *ok = (e->h_state->done != DONE_ERROR && e->h_state->done != DONE_RUNNING && hs.err[0] == 0) ? 1 : 0;
    return 0;
}

extern "C" int ccsim_dist_mbox_finish(ccsim_engine *e, int32_t all_ok) {
    if (!e || !e->begun) return -EINVAL;
    HIPCHK(e, hipSetDevice(e->device));
    if (all_ok) return 0; // the commit rows hold the final state: ccsim_dist_finish publishes them (k_rows_flush) and reports
    // some rank could not finish: nobody publishes.  The rows k_rows_build made are what the pass protocol starts from -- rewrite
    // them (this rank's launch may have written its rows), restore the run state ccsim_dist_begin uploaded
    const int blocks = (int)((e->n_pad + kThreads - 1) / kThreads);
    hipLaunchKernelGGL(k_rows_build, dim3(blocks), dim3(kThreads), 0, e->stream, e->cols);
    HIPCHK(e, hipGetLastError());
    *e->h_state = e->mb_state0;
    HIPCHK(e, hipMemcpyAsync(e->d_state, e->h_state, sizeof(DevState), hipMemcpyHostToDevice, e->stream));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    e->dist_pass_in_window = 0;
    return 1;
}

// ================================================================================================================
// The sharded run driven from C++ over the engine's own RCCL communicator (include/ccsim.h "multi-GPU, driven by the
// library").  RCCL is bound at run time (dlopen): libccsim.so has no link-time dependency on it and single-GPU users
// never load it.  In a process that already holds an RCCL (torch ships one) the same library is reused (same SONAME).
// ================================================================================================================
namespace {
struct CcNcclId { char internal[CCSIM_DIST_ID_BYTES]; }; // ncclUniqueId (rccl.h:40-43)
struct RcclApi {
    void *h = nullptr;
    int (*GetUniqueId)(CcNcclId *) = nullptr;
    int (*CommInitRank)(void **, int, CcNcclId, int) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    int (*CommCount)(void *, int *) = nullptr;
    int (*CommUserRank)(void *, int *) = nullptr;
    std::string err;
};
constexpr int kNcclInt8 = 0, kNcclInt32 = 2, kNcclInt64 = 4, kNcclSum = 0, kNcclMax = 2, kNcclMin = 3; // rccl.h:448-463

static double now_s() {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
static bool dist_debug() {
    static const bool on = getenv("CCSIM_DIST_DEBUG") && *getenv("CCSIM_DIST_DEBUG") && *getenv("CCSIM_DIST_DEBUG") != '0';
    return on;
}

RcclApi &rccl() {
    static RcclApi a;
    static std::mutex mu;
    std::lock_guard<std::mutex> lk(mu);
    if (a.h || !a.err.empty()) return a;
    const double t0 = now_s();
    // an RCCL the process already maps wins (torch ships one under the same SONAME): RTLD_NOLOAD probes without loading
    const char *names[] = {getenv("CCSIM_RCCL_LIB"), "librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"};
    std::string last;
    for (const char *n : names) {
        if (!n || !*n) continue;
        a.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (a.h) break;
        const char *m = dlerror(); // (one call: dlerror() clears the message it returns)
        last = m ? m : "?";
        if (n == names[0]) break; // an explicit CCSIM_RCCL_LIB that cannot be loaded is an error, not a hint
    }
    if (!a.h) {
        a.err = "librccl.so.1 could not be loaded: " + last;
        if (dist_debug()) fprintf(stderr, "[ccsim dist] %s\n", a.err.c_str());
        return a;
    }
    auto sym = [&](const char *n) -> void * {
        void *p = dlsym(a.h, n);
        if (!p && a.err.empty()) a.err = std::string("librccl lacks ") + n;
        return p;
    };
    a.GetUniqueId = (decltype(a.GetUniqueId))sym("ncclGetUniqueId");
    a.CommInitRank = (decltype(a.CommInitRank))sym("ncclCommInitRank");
    a.CommDestroy = (decltype(a.CommDestroy))sym("ncclCommDestroy");
    a.AllGather = (decltype(a.AllGather))sym("ncclAllGather");
    a.AllReduce = (decltype(a.AllReduce))sym("ncclAllReduce");
    a.GetErrorString = (decltype(a.GetErrorString))sym("ncclGetErrorString");
    a.CommCount = (decltype(a.CommCount))sym("ncclCommCount");
    a.CommUserRank = (decltype(a.CommUserRank))sym("ncclCommUserRank");
    if (!a.err.empty()) a.h = nullptr;
    if (dist_debug()) fprintf(stderr, "[ccsim dist] librccl bound in %.2f s%s%s\n", now_s() - t0, a.err.empty() ? "" : ": ", a.err.c_str());
    return a;
}

// ncclCommInitRank blocks until every rank has arrived and gives no way to bound that; a rank that never comes (or a
// first touch of a 500 MB library on a cold box) must end as an error with a text, not as a process that never returns.
// The call runs on a helper thread; the caller waits CCSIM_RCCL_INIT_TIMEOUT_S (default 900) and then gives up (the helper
// is abandoned: it holds only heap state of its own).
struct CommInitJob {
    std::mutex mu;
    std::condition_variable cv;
    bool done = false;
    int rc = 0;
    void *comm = nullptr;
};
static int comm_init_bounded(RcclApi &r, int device, int n_ranks, const CcNcclId &id, int rank, void **comm_out, double *waited) {
    auto job = std::make_shared<CommInitJob>();
    std::thread([job, &r, device, n_ranks, id, rank]() {
        int rc = (int)hipSetDevice(device);
        void *c = nullptr;
        if (rc == 0) rc = r.CommInitRank(&c, n_ranks, id, rank);
        std::lock_guard<std::mutex> lk(job->mu);
        job->rc = rc, job->comm = c, job->done = true;
        job->cv.notify_all();
    }).detach();
    double limit = 900;
    if (const char *t = getenv("CCSIM_RCCL_INIT_TIMEOUT_S")) limit = atof(t) > 0 ? atof(t) : limit;
    const double t0 = now_s();
    std::unique_lock<std::mutex> lk(job->mu);
    const bool ok = job->cv.wait_for(lk, std::chrono::duration<double>(limit), [&] { return job->done; });
    *waited = now_s() - t0;
    if (!ok) return -ETIMEDOUT;
    *comm_out = job->comm;
    return job->rc;
}
} // namespace

#define RCCLCHK(e, call)                                                                                  \
    do {                                                                                                  \
        const int rccl_rc_ = (call);                                                                      \
        if (rccl_rc_ != 0) return fail(e, -EIO, "RCCL: %s failed: %s", #call, rccl().GetErrorString(rccl_rc_)); \
    } while (0)

static void dist_comm_release(ccsim_engine *e) {
    if (e->rccl_comm && rccl().h) (void)rccl().CommDestroy(e->rccl_comm);
    e->rccl_comm = nullptr;
    if (e->d_own_send) (void)hipFree(e->d_own_send);
    if (e->d_own_recv) (void)hipFree(e->d_own_recv);
    e->d_own_send = e->d_own_recv = nullptr;
}

extern "C" int ccsim_dist_unique_id(uint8_t *id_out) {
    if (!id_out) return -EINVAL;
    RcclApi &r = rccl();
    if (!r.h) return -EIO;
    CcNcclId id;
    if (r.GetUniqueId(&id) != 0) return -EIO;
    memcpy(id_out, id.internal, CCSIM_DIST_ID_BYTES);
    return 0;
}

extern "C" int ccsim_dist_comm_init(ccsim_engine *e, const uint8_t *id_bytes, int32_t n_ranks, int32_t rank) {
    if (!e || !id_bytes || n_ranks < 1 || rank < 0 || rank >= n_ranks) return -EINVAL;
    RcclApi &r = rccl();
    if (!r.h) return fail(e, -EIO, "%s", r.err.c_str());
    HIPCHK(e, hipSetDevice(e->device));
    dist_comm_release(e);
    CcNcclId id;
    memcpy(id.internal, id_bytes, CCSIM_DIST_ID_BYTES);
    double waited = 0;
    const int irc = comm_init_bounded(r, e->device, n_ranks, id, rank, &e->rccl_comm, &waited);
    if (dist_debug()) fprintf(stderr, "[ccsim dist] ncclCommInitRank(rank %d of %d, device %d) returned %d after %.2f s\n", rank, n_ranks, e->device, irc, waited);
    if (irc == -ETIMEDOUT)
        return fail(e, -EIO, "RCCL: ncclCommInitRank(rank %d of %d) did not return within %.0f s (CCSIM_RCCL_INIT_TIMEOUT_S); a rank is missing or librccl is stuck", rank, n_ranks, waited);
    if (irc != 0) return fail(e, -EIO, "RCCL: ncclCommInitRank failed: %s", r.GetErrorString(irc));
    e->comm_ranks = n_ranks, e->comm_rank = rank;
    HIPCHK(e, hipMalloc((void **)&e->d_own_send, sizeof(XRec)));
    HIPCHK(e, hipMalloc((void **)&e->d_own_recv, sizeof(XRec) * (size_t)n_ranks));
    HIPCHK(e, hipMemset(e->d_own_send, 0, sizeof(XRec)));
    HIPCHK(e, hipMemset(e->d_own_recv, 0, sizeof(XRec) * (size_t)n_ranks));
    // the persistent kernel across the GPUs (CCSIM_DIST_MAILBOX=1): every rank's mailbox mapped into every rank, addressing
    // records all-gathered over the new communicator.  Whatever fails here leaves the pass protocol as it is.
    mbox_disconnect(e);
    if (getenv("CCSIM_DIST_MAILBOX") && atoi(getenv("CCSIM_DIST_MAILBOX")) != 0 && n_ranks <= kPMaxRanks && sizeof(XRec) >= CCSIM_MBOX_INFO_BYTES) {
        uint8_t mine[CCSIM_MBOX_INFO_BYTES];
        std::vector<uint8_t> all((size_t)n_ranks * CCSIM_MBOX_INFO_BYTES);
        int rc = ccsim_dist_mbox_info(e, mine);
        if (rc == 0) {
            HIPCHK(e, hipMemcpy(e->d_own_send, mine, sizeof mine, hipMemcpyHostToDevice));
            RCCLCHK(e, r.AllGather(e->d_own_send, e->d_own_recv, CCSIM_MBOX_INFO_BYTES, kNcclInt8, e->rccl_comm, e->stream));
            HIPCHK(e, hipStreamSynchronize(e->stream));
            HIPCHK(e, hipMemcpy(all.data(), e->d_own_recv, all.size(), hipMemcpyDeviceToHost));
            rc = ccsim_dist_mbox_connect(e, all.data(), n_ranks, rank);
            HIPCHK(e, hipMemset(e->d_own_send, 0, sizeof(XRec)));
            HIPCHK(e, hipMemset(e->d_own_recv, 0, sizeof(XRec) * (size_t)n_ranks));
        }
        {   // connected everywhere or nowhere: the ranks must agree on whether ccsim_dist_run takes the collective steps of the mailbox form
            int32_t mine_ok = rc == 0 ? 1 : 0, all_ok = 0;
            int32_t *buf = (int32_t *)e->d_own_send;
            HIPCHK(e, hipMemcpy(buf, &mine_ok, sizeof mine_ok, hipMemcpyHostToDevice));
            RCCLCHK(e, r.AllReduce(buf, buf, 1, kNcclInt32, kNcclMin, e->rccl_comm, e->stream));
            HIPCHK(e, hipStreamSynchronize(e->stream));
            HIPCHK(e, hipMemcpy(&all_ok, buf, sizeof all_ok, hipMemcpyDeviceToHost));
            HIPCHK(e, hipMemset(e->d_own_send, 0, sizeof(XRec)));
            if (!all_ok) mbox_disconnect(e);
        }
        if (dist_debug()) fprintf(stderr, "[ccsim dist] rank %d: mailboxes %s%s%s\n", rank, rc == 0 ? "connected" : "NOT connected", rc ? ": " : "", rc ? e->err.c_str() : "");
    }
    return 0;
}

extern "C" int ccsim_dist_comm_size(ccsim_engine *e, int32_t *n_ranks_out, int32_t *rank_out) {
    if (!e || !n_ranks_out) return -EINVAL;
    if (!e->rccl_comm) return fail(e, -EINVAL, "ccsim_dist_comm_init first");
    int n = 0, r = 0;
    RCCLCHK(e, rccl().CommCount(e->rccl_comm, &n));
    RCCLCHK(e, rccl().CommUserRank(e->rccl_comm, &r));
    *n_ranks_out = n;
    if (rank_out) *rank_out = r;
    return 0;
}

// minimum over the ranks of a host flag, through the communicator (the mailbox form's go / no-go decisions)
static int dist_all_min(ccsim_engine *e, int32_t mine, int32_t *out) {
    int32_t *buf = (int32_t *)e->d_own_send; // (idle here: the pass protocol has not started / is over)
    HIPCHK(e, hipMemcpyAsync(buf, &mine, sizeof mine, hipMemcpyHostToDevice, e->stream));
    RCCLCHK(e, rccl().AllReduce(buf, buf, 1, kNcclInt32, kNcclMin, e->rccl_comm, e->stream));
    HIPCHK(e, hipMemcpyAsync(out, buf, sizeof *out, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    HIPCHK(e, hipMemsetAsync(e->d_own_send, 0, sizeof(XRec), e->stream));
    return 0;
}

extern "C" int ccsim_dist_sync_tables(ccsim_engine *e) {
    if (!e || !e->have_pod) return -EINVAL;
    if (!e->rccl_comm) return fail(e, -EINVAL, "ccsim_dist_comm_init first");
    if (e->dist_tables.empty()) return 0;
    HIPCHK(e, hipSetDevice(e->device));
    for (const auto &t : e->dist_tables)
        RCCLCHK(e, rccl().AllReduce(t.ptr, t.ptr, (size_t)t.len, t.elem_bytes == 8 ? kNcclInt64 : kNcclInt32, t.op == 1 ? kNcclMax : kNcclSum,
                                    e->rccl_comm, e->stream));
    return ccsim_dist_tables_done(e);
}

extern "C" int ccsim_dist_run(ccsim_engine *e, int64_t max_limit, int32_t mode, ccsim_report *out) {
    if (!e || !out) return -EINVAL;
    if (!e->rccl_comm) return fail(e, -EINVAL, "ccsim_dist_comm_init first");
    static_assert(sizeof(XRec) == CCSIM_XCHG_WORDS * 8, "exchange record size");
    int rc = ccsim_dist_begin(e, max_limit, mode, e->comm_ranks, e->comm_rank, e->d_own_send, e->d_own_recv, out->log ? out->log_cap : 0);
    if (rc) return rc;
    // The persistent kernel across the GPUs (mailboxes connected by ccsim_dist_comm_init under CCSIM_DIST_MAILBOX=1): one launch per
    // rank for the whole batched run, the exchange inside the kernel.  Two agreements around it -- is every rank eligible, did every
    // rank finish -- and the pass protocol below as the fallback from the untouched state.
    // CCSIM_DIST_FORM=passes (read per run; the SAME value on every rank): this run takes the RCCL pass protocol although the mailboxes are
    // connected -- the A/B knob of bench.py, which times both forms in one process
    const char *form = getenv("CCSIM_DIST_FORM");
    const bool passes_only = form && !strcmp(form, "passes");
    e->dist_last_form = 2;
    if (mode == CCSIM_MODE_BATCHED && e->mbox_ready && e->mb_ranks == e->comm_ranks && !passes_only) { // (connected by the same collective call on every rank)
        int32_t go = e->mb_go, fine = 0;
        if (go < 0) { // (once per pod spec: eligibility is a property of the snapshot and the pod)
            if ((rc = dist_all_min(e, ccsim_dist_mbox_eligible(e), &go))) return rc;
            e->mb_go = go;
        }
        if (go) {
            // launch -> ncclAllReduce(min) of the device flag -> state + verdict to the host: ONE stream sync for the whole run
            const int lrc = ccsim_dist_mbox_launch(e);
            e->mb_launches += 1;
            if (lrc != 0) HIPCHK(e, hipMemsetAsync(e->d_mb_ok, 0, sizeof(int32_t), e->stream)); // (this rank could not launch: every rank falls back)
            RCCLCHK(e, rccl().AllReduce(e->d_mb_ok, e->d_mb_ok, 1, kNcclInt32, kNcclMin, e->rccl_comm, e->stream));
            HIPCHK(e, hipMemcpyAsync(e->h_mb_ok, e->d_mb_ok, sizeof(int32_t), hipMemcpyDeviceToHost, e->stream));
            if ((rc = read_state(e))) return rc; // (synchronizes)
            fine = e->h_mb_ok[0];
            if (lrc == 0) {
                float ms = 0;
                HIPCHK(e, hipEventElapsedTime(&ms, e->ev0, e->ev1));
                e->kernel_ms = ms, e->pass_kernel_ms = ms, e->pass_launches = 1;
            }
            if (dist_debug()) fprintf(stderr, "[ccsim dist] rank %d: mailbox form %s (%.3f ms)\n", e->comm_rank, fine ? "finished on every rank" : "abandoned: pass protocol", e->kernel_ms);
            rc = ccsim_dist_mbox_finish(e, fine);
            if (rc < 0) return rc;
            if (fine) {
                e->dist_last_form = 1;
                return ccsim_dist_finish(e, out);
            }
            // Abandoned (a rank could not launch, a grid barrier or a peer's record did not arrive within the bounded spins): the verdict is
            // the all-reduced minimum, so every rank lands here together -- and every later run of this pod spec would relaunch the
            // kernel, spin for seconds and fall back again.  One attempt per pod spec: the pass protocol from now on (set_pod /
            // load_nodes / comm_init agree anew).
            e->mb_go = 0;
            e->mb_abandoned += 1;
            if (e->comm_rank == 0 || dist_debug())
                fprintf(stderr, "[ccsim dist] rank %d: the persistent kernel across the GPUs was abandoned (%.1f ms lost); this pod spec's runs take the RCCL pass protocol from here on\n",
                        e->comm_rank, e->kernel_ms);
        }
    }
    int per_poll = 32;
    if (const char *f = getenv("CCSIM_DIST_POLL")) per_poll = atoi(f) > 0 ? atoi(f) : per_poll; // tuning knob (the SAME value on every rank)
    int64_t last_placed = -1;
    int idle = 0;
    HIPCHK(e, hipEventRecord(e->ev0, e->stream)); // (kernel_ns of a sharded run: the whole pass train, exchanges included)
    {   // one template with a shared-key hard constraint + a unique inter-pod key: WINDOWS of placements per exchange (every rank must
        // be able to: the agreement is one all-reduce; then per window the pass over the shard, ONE all-gather of 73 KB per rank, the
        // deciding wave replicated).  A window nobody can take sets cw_fallback on every rank alike: the pass protocol below continues.
        int32_t go = 0;
        if (mode == CCSIM_MODE_SEQUENTIAL && (e->pts.n > 0 || e->ipa.on)) { // (the same test on every rank: the pod is)
            if ((rc = dist_all_min(e, ccsim_dist_cw_eligible(e), &go))) return rc;
            if ((rc = ccsim_dist_cw_enable(e, go))) return rc;
        }
        int cw_idle = 0;
        while (go) {
            const int64_t p0 = e->h_state->placed;
            for (int w = 0; w < 4; w++) {
                if ((rc = ccsim_dist_cw_scan(e))) return rc;
                RCCLCHK(e, rccl().AllGather(e->cw_work.xsend, e->cw_work.xrecv, kCwXBytes, kNcclInt8, e->rccl_comm, e->stream));
                if ((rc = ccsim_dist_cw_decide(e))) return rc;
            }
            if ((rc = read_state(e))) return rc;
            if (e->h_state->done || e->h_state->cw_fallback) break;
            cw_idle = e->h_state->placed == p0 ? cw_idle + 1 : 0;
            if (cw_idle >= 4) return fail(e, -EIO, "sharded windowed simulation made no progress in 16 windows");
        }
        if (e->h_state->done) {
            HIPCHK(e, hipEventRecord(e->ev1, e->stream));
            HIPCHK(e, hipEventSynchronize(e->ev1));
            float wms = 0;
            HIPCHK(e, hipEventElapsedTime(&wms, e->ev0, e->ev1));
            e->kernel_ms = wms;
            e->dist_last_form = 3;
            return ccsim_dist_finish(e, out);
        }
    }
    for (;;) {
        for (int p = 0; p < per_poll; p++) {
            if ((rc = ccsim_dist_scan(e))) return rc;
            // the max-loc exchange: one 256-byte record per rank, on the engine's stream (ordered with the kernels, no host sync)
            RCCLCHK(e, rccl().AllGather(e->d_own_send, e->d_own_recv, CCSIM_XCHG_WORDS, kNcclInt64, e->rccl_comm, e->stream));
            if ((rc = ccsim_dist_decide(e))) return rc;
        }
        int32_t done = 0;
        int64_t placed = 0;
        if ((rc = ccsim_dist_poll(e, &done, &placed))) return rc;
        if (done) break;
        idle = placed == last_placed ? idle + 1 : 0; // (identical on every rank: the state is replicated)
        last_placed = placed;
        if (idle >= 64) return fail(e, -EIO, "sharded simulation made no progress in %d passes", 64 * per_poll);
    }
    HIPCHK(e, hipEventRecord(e->ev1, e->stream));
    HIPCHK(e, hipEventSynchronize(e->ev1));
    float ms = 0;
    HIPCHK(e, hipEventElapsedTime(&ms, e->ev0, e->ev1));
    e->kernel_ms = ms;
    return ccsim_dist_finish(e, out);
}

// ================================================================================================================
// Several pod specs cycled round-robin (include/ccsim.h ccsim_set_pods; kernels in ccsim_multi.h)
// ================================================================================================================
static std::string static_class_key(const ccsim_engine *e, const ccsim_pod *pod, bool score_preferred) {
    std::string k;
    auto put = [&](const void *p, size_t n) { k.append((const char *)p, n); };
    auto puti = [&](int64_t v) { put(&v, sizeof v); };
    puti(pod->n_taintsets);
    put(pod->taint_filter_ok, (size_t)pod->n_taintsets);
    if (e->prof.w_taint) put(pod->taint_prefer_cnt, sizeof(int32_t) * (size_t)pod->n_taintsets);
    puti(pod->tolerates_unschedulable), puti(pod->affinity_filter_active), puti(pod->has_node_selector), puti(pod->has_required_terms);
    auto put_term = [&](const ccsim_term &t, bool weight) {
        puti(t.n_req);
        if (weight) puti(t.weight);
        for (int i = 0; i < t.n_req; i++) {
            const ccsim_requirement &r = pod->reqs[t.first_req + i];
            puti(r.col);
            // the requirement's verdict table: one byte per value id of the column
            const int32_t mx = r.col >= 0 && r.col < (int)e->label_col_max.size() ? e->label_col_max[(size_t)r.col] : 0;
            put(pod->req_tables + r.table_off, (size_t)mx + 1);
        }
    };
    if (pod->has_node_selector) put_term(pod->node_selector, false);
    puti(pod->n_required);
    for (int t = 0; t < pod->n_required; t++) put_term(pod->required[t], false);
    puti(score_preferred ? pod->n_preferred : 0);
    if (score_preferred)
        for (int t = 0; t < pod->n_preferred; t++) put_term(pod->preferred[t], true);
    return k;
}

extern "C" int ccsim_set_pods(ccsim_engine *e, const ccsim_pod *pods, int32_t n_pods) {
    if (!e || !pods || n_pods < 1) return -EINVAL;
    if (n_pods == 1) return ccsim_set_pod(e, &pods[0]);
    if (!e->have_nodes || !e->have_profile) return fail(e, -EINVAL, "load nodes and set the profile first");
    const ccsim_profile &pf = e->prof;
    if (e->n_global != e->n || e->global_offset != 0) return fail(e, -ENOSYS, "several pod specs: one GPU only");
    if (num_feasible_nodes_to_find(pf.percentage_of_nodes_to_score, e->n_global) < e->n_global || !profile_has_scoring(pf))
        return fail(e, -ENOSYS, "several pod specs: percentageOfNodesToScore must be 100 (every node is scored) and the profile needs a Score plugin");
    if (!(pf.filter_mask & CCSIM_F_FIT)) return fail(e, -ENOSYS, "several pod specs need the NodeResourcesFit filter");
    int rc;
    for (int p = 0; p < n_pods; p++)
        if ((rc = validate_pod(e, &pods[p]))) return rc;
    HIPCHK(e, hipSetDevice(e->device));
    if (int erc = ensure_cols(e)) return erc;
    HIPCHK(e, hipStreamSynchronize(e->stream));
    drop_graph(e);
    free_list(e->pod_allocs);
    free_list(e->multi_allocs);
    e->multi = false;
    e->d_memo = nullptr, e->d_memo_stamp = nullptr, e->d_mtouched = nullptr, e->d_mvsync = nullptr;
    e->have_pod = e->begun = false;
    e->pts = DevPts{}, e->soft = DevSoft{}, e->ipa = DevIpa{};
    e->d_soft_pc0 = nullptr;
    e->cols.alloc_pods = e->d_alloc_pods_real, e->ports_on = e->excl_ports = false, e->d_vol_veto = nullptr;
    e->pts_tables.clear(), e->pts_table_len.clear(), e->ipa_tables.clear(), e->ipa_table_len.clear(), e->soft_flags.clear();
    e->dist_tables.clear(), e->pts_present.clear();
    const size_t N = (size_t)e->n, NP = (size_t)e->n_pad;

    // ---- what P > 1 supports (include/ccsim.h) -------------------------------------------------------------------
    uint64_t mem_or = e->node_mem_or;
    int64_t grow_c = 0, grow_m = 0;
    int slot_col[kMTsc] = {-1, -1};
    const bool pts_on = (pf.filter_mask & CCSIM_F_TOPOLOGYSPREAD) != 0, ipa_on = (pf.filter_mask & CCSIM_F_INTERPODAFFINITY) != 0;
    for (int p = 0; p < n_pods; p++) {
        const ccsim_pod &q = pods[p];
        for (int c = 2; c < CCSIM_MAX_RES; c++)
            if (q.req[c] != 0) return fail(e, -ENOSYS, "several pod specs: requests beyond cpu / memory (spec %d)", p);
        if (q.has_scalar_entries) return fail(e, -ENOSYS, "several pod specs: scalar resource entries (spec %d)", p);
        if (q.has_host_ports && (pf.filter_mask & CCSIM_F_NODEPORTS)) return fail(e, -ENOSYS, "several pod specs: host ports (spec %d)", p);
        if (q.volume_exclusive || q.volume_veto) return fail(e, -ENOSYS, "several pod specs: volume plugins (spec %d)", p);
        mem_or |= (uint64_t)q.req[1] | (uint64_t)q.nz_mem;
        const int64_t gc = q.req[0] > q.nz_mcpu ? q.req[0] : q.nz_mcpu, gm = q.req[1] > q.nz_mem ? q.req[1] : q.nz_mem;
        grow_c = gc > grow_c ? gc : grow_c, grow_m = gm > grow_m ? gm : grow_m;
        int hard = 0;
        for (int c = 0; c < q.n_spread; c++) {
            const ccsim_spread_constraint &k = q.spread[c];
            if (!k.hard) {
                if (pf.w_topologyspread) return fail(e, -ENOSYS, "several pod specs: ScheduleAnyway spread constraints (spec %d)", p);
                continue;
            }
            if (!pts_on) continue;
            if (++hard > kMTsc) return fail(e, -ENOSYS, "several pod specs: more than %d DoNotSchedule constraints (spec %d)", kMTsc, p);
            if (k.col < 0 || k.col >= e->n_label_cols || k.max_skew < 1 || k.min_domains < 1) return fail(e, -EINVAL, "bad spread constraint (spec %d)", p);
            if (k.n_domains < 0 || k.n_domains > kMDomMax - 1) return fail(e, -ENOSYS, "several pod specs: spread constraints over more than %d domains (spec %d)", kMDomMax - 1, p);
            if (e->label_col_max[(size_t)k.col] > k.n_domains) return fail(e, -EINVAL, "spread constraint: label column holds value ids above n_domains (spec %d)", p);
            int sl = slot_col[0] == k.col ? 0 : (slot_col[1] == k.col ? 1 : -1);
            if (sl < 0) {
                sl = slot_col[0] < 0 ? 0 : (slot_col[1] < 0 ? 1 : -1);
                if (sl < 0) return fail(e, -ENOSYS, "several pod specs: spread constraints over more than two topology keys in total");
                slot_col[sl] = k.col;
            }
        }
        if (q.has_ipa && (ipa_on || pf.w_interpodaffinity)) {
            const ccsim_ipa &a = q.ipa;
            bool okk = a.n_keys == 1 && a.n_aff_terms == 0 && a.n_anti_terms >= 1 && !a.aff_existing && a.entries_existing == 0 &&
                       a.score_self[0] == 0 && a.self_entries[0] == 0 && !a.score_existing[0] && !a.exist_anti[0] && a.key_col[0] >= 0 &&
                       a.key_col[0] < e->n_label_cols;
            for (int t = 0; okk && t < a.n_anti_terms; t++) okk = a.anti_key[t] == 0 && a.anti_self[t] != 0;
            if (okk) { // the key must put every node into its own domain (kubernetes.io/hostname)
                const std::vector<int32_t> &col = e->h_label_cols[(size_t)a.key_col[0]];
                for (size_t i = 0; okk && i < N; i++) okk = col[i] == (int32_t)(i + 1);
            }
            if (!okk) return fail(e, -ENOSYS, "several pod specs: inter-pod affinity other than required anti-affinity of a spec to its own clones on a "
                                              "one-node-per-domain key (spec %d)", p);
            if (!ipa_on) return fail(e, -ENOSYS, "several pod specs: inter-pod affinity with its Filter disabled (spec %d)", p);
        }
    }
    int sh = mem_or ? __builtin_ctzll(mem_or) : 0;
    if (sh > 40) sh = 40;
    const bool fits = e->node_max_cpu < (1ll << 30) && grow_c * (e->node_max_pods + 1) < (1ll << 30) && (e->node_max_mem >> sh) < (1ll << 30) &&
                      ((grow_m * (e->node_max_pods + 1)) >> sh) < (1ll << 30);
    if (!fits || N == 0) return fail(e, -ENOSYS, "several pod specs need the lossless 32-bit mirrors (cpu < 2^30 milli, memory a multiple of a common power-of-two unit)");
    e->cols.narrow = 1, e->cols.mem_shift = sh;
    if ((rc = build_narrow(e))) return rc;

    // ---- static classes: one k_static run per distinct (tolerations, selectors, terms) ---------------------------
    // (the static word also carries the node's ImageLocality score for the pod: specs share a class only with the same per-node
    // scores -- compared by a hash of the array first, then byte by byte against the class's representative)
    std::map<std::string, int> cls_of;
    std::vector<int> pod_cls((size_t)n_pods), cls_rep;
    bool any_img = false;
    for (int p = 0; p < n_pods; p++) {
        const bool pref = pods[p].n_preferred > 0 && pf.w_nodeaffinity;
        std::string key = static_class_key(e, &pods[p], pref);
        const uint8_t *img = pf.w_imagelocality ? pods[p].image_score : nullptr;
        if (img) {
            any_img = true;
            for (size_t i = 0; i < N; i++)
                if (img[i] > 100) return fail(e, -EINVAL, "image_score out of [0,100] (spec %d)", p);
            uint64_t h = 1469598103934665603ull; // FNV-1a over the scores
            for (size_t i = 0; i < N; i++) h = (h ^ img[i]) * 1099511628211ull;
            key.append("img", 3), key.append((const char *)&h, sizeof h);
        }
        for (int salt = 0;; salt++) { // (a hash collision between different arrays opens another class)
            auto it = cls_of.emplace(key, (int)cls_of.size());
            if (it.second) cls_rep.push_back(p);
            const uint8_t *rep_img = pf.w_imagelocality ? pods[cls_rep[(size_t)it.first->second]].image_score : nullptr;
            if (!img || it.second || (rep_img && memcmp(rep_img, img, N) == 0)) {
                pod_cls[(size_t)p] = it.first->second;
                break;
            }
            key.push_back((char)('0' + salt % 10));
        }
    }
    e->n_cls = (int)cls_rep.size();
    if ((rc = dev_alloc(e, &e->d_stat_cls, NP * (size_t)e->n_cls, e->multi_allocs))) return rc;
    if ((rc = dev_alloc(e, &e->d_sreason_cls, NP * (size_t)e->n_cls, e->multi_allocs))) return rc;
    e->cls_taintsets.assign((size_t)e->n_cls, 1);
    e->max_taintsets = 1;
    for (int c = 0; c < e->n_cls; c++) {
        const ccsim_pod &q = pods[cls_rep[(size_t)c]];
        if ((rc = static_pass(e, &q, q.n_preferred > 0 && pf.w_nodeaffinity, e->d_stat_cls + NP * (size_t)c, e->d_sreason_cls + NP * (size_t)c, e->pod_allocs))) return rc;
        free_list(e->pod_allocs); // (static_pass synchronizes: its tables are no longer needed)
        e->cls_taintsets[(size_t)c] = q.n_taintsets;
        e->max_taintsets = q.n_taintsets > e->max_taintsets ? q.n_taintsets : e->max_taintsets;
    }
    e->ts_in_frame = false;
    if ((rc = dev_alloc(e, &e->d_hist_ts, (size_t)e->max_taintsets, e->multi_allocs))) return rc;

    // ---- per-spec plugin state: spread count tables (+ which domains hold a counted node), inclusion arrays, anti-affinity bits
    std::vector<MPod> mp((size_t)n_pods);
    std::vector<int32_t> tbl;
    std::vector<uint8_t> present;
    std::vector<const uint8_t *> inc_ptrs; // distinct inclusion arrays, by pointer (callers share one array per selector)
    std::map<const uint8_t *, int> inc_id;
    e->anti_words = (size_t)n_pods * (NP / 32);
    std::vector<uint32_t> bits;
    bool any_anti = false;
    for (int p = 0; p < n_pods; p++) {
        const ccsim_pod &q = pods[p];
        MPod &m = mp[(size_t)p];
        m = MPod{};
        m.req0 = (int32_t)q.req[0], m.req1 = (int32_t)(q.req[1] >> sh), m.nz0 = (int32_t)q.nz_mcpu, m.nz1 = (int32_t)(q.nz_mem >> sh);
        m.req_wide[0] = q.req[0], m.req_wide[1] = q.req[1], m.nz_wide[0] = q.nz_mcpu, m.nz_wide[1] = q.nz_mem;
        m.cls = pod_cls[(size_t)p];
        const DevPod dp = make_devpod(e, &q);
        m.all_zero_req = dp.all_zero_req, m.w_bal = dp.w_bal, m.w_aff = dp.w_aff;
        // hard constraints; a node is counted iff it carries ALL the spec's hard keys and passes the inclusion policies
        std::vector<int> hard;
        for (int c = 0; c < q.n_spread && pts_on; c++)
            if (q.spread[c].hard) hard.push_back(c);
        m.n_tsc = (int32_t)hard.size();
        for (size_t j = 0; j < hard.size(); j++) {
            const ccsim_spread_constraint &k = q.spread[hard[j]];
            m.tsc_slot[j] = slot_col[0] == k.col ? 0 : 1;
            m.tsc_max_skew[j] = k.max_skew, m.tsc_min_dom[j] = k.min_domains, m.tsc_self[j] = k.self_match ? 1 : 0, m.tsc_ndom[j] = k.n_domains;
            m.tsc_tbl[j] = (int32_t)tbl.size();
            m.tsc_inc[j] = -1;
            if (k.node_included) {
                auto it = inc_id.emplace(k.node_included, (int)inc_ptrs.size());
                if (it.second) inc_ptrs.push_back(k.node_included);
                m.tsc_inc[j] = it.first->second;
            }
            tbl.resize(tbl.size() + (size_t)k.n_domains + 1, 0);
            present.resize(tbl.size(), 0);
        }
        for (size_t j = 0; j < hard.size(); j++) {
            const ccsim_spread_constraint &k = q.spread[hard[j]];
            const std::vector<int32_t> &col = e->h_label_cols[(size_t)k.col];
            int32_t np_ = 0;
            for (size_t i = 0; i < N; i++) {
                bool all = true;
                for (size_t j2 = 0; j2 < hard.size() && all; j2++) all = e->h_label_cols[(size_t)q.spread[hard[j2]].col][i] != 0;
                if (!all || (k.node_included && !k.node_included[i])) continue;
                const int32_t v = col[i];
                if (v < 0 || v > k.n_domains) return fail(e, -EINVAL, "topology value id out of range (spec %d)", p);
                if (!present[(size_t)m.tsc_tbl[j] + (size_t)v]) present[(size_t)m.tsc_tbl[j] + (size_t)v] = 1, np_++;
                if (k.node_match_count) tbl[(size_t)m.tsc_tbl[j] + (size_t)v] += k.node_match_count[i];
            }
            m.tsc_npresent[j] = np_;
        }
        if (q.has_ipa && ipa_on) {
            m.anti = 1;
            any_anti = true;
            if (bits.empty()) bits.assign(e->anti_words, 0u);
            uint32_t *row = bits.data() + (size_t)p * (NP / 32);
            const ccsim_ipa &a = q.ipa;
            for (size_t i = 0; i < N; i++) {
                bool hit = false;
                for (int t = 0; t < a.n_anti_terms && !hit; t++) hit = a.anti_existing[t] && a.anti_existing[t][i] > 0;
                if (hit) row[i >> 5] |= 1u << (i & 31);
            }
        }
    }
    if (tbl.empty()) tbl.push_back(0), present.push_back(0);
    e->tbl_len = tbl.size();
    if ((rc = upload(e, &e->d_tbl_pool, tbl.data(), tbl.size(), tbl.size(), e->multi_allocs))) return rc;
    if ((rc = upload(e, &e->d_tbl_pool0, tbl.data(), tbl.size(), tbl.size(), e->multi_allocs))) return rc;
    if ((rc = upload(e, &e->d_present_pool, present.data(), present.size(), present.size(), e->multi_allocs))) return rc;
    if ((rc = dev_alloc(e, &e->d_inc_pool, NP * (inc_ptrs.empty() ? 1 : inc_ptrs.size()), e->multi_allocs))) return rc;
    for (size_t i = 0; i < inc_ptrs.size(); i++)
        HIPCHK(e, hipMemcpyAsync(e->d_inc_pool + NP * i, inc_ptrs[i], N, hipMemcpyHostToDevice, e->stream));
    if (!any_anti) e->anti_words = 1;
    if ((rc = upload(e, &e->d_anti_bits, bits.empty() ? nullptr : bits.data(), bits.size(), e->anti_words, e->multi_allocs))) return rc;
    if ((rc = upload(e, &e->d_anti_bits0, bits.empty() ? nullptr : bits.data(), bits.size(), e->anti_words, e->multi_allocs))) return rc;
    if ((rc = upload(e, &e->d_mpods, mp.data(), mp.size(), mp.size(), e->multi_allocs))) return rc;
    e->h_mpods = mp;
    e->m_blocks = (int)((e->n_pad + kMBlockNodes - 1) / kMBlockNodes);
    if ((rc = dev_alloc(e, &e->d_mpartials, (size_t)kMWindowMax * (size_t)e->m_blocks, e->multi_allocs))) return rc;
    if ((rc = dev_alloc(e, &e->d_mcands, (size_t)kMWindowMax, e->multi_allocs))) return rc;
    if ((rc = dev_alloc(e, &e->d_mvsync, (size_t)kMVsyncWords, e->multi_allocs))) return rc; // (zeroed here; k_multi_commit_par leaves it zeroed)
    if ((rc = dev_alloc(e, &e->d_per_spec, (size_t)n_pods, e->multi_allocs))) return rc;
    if ((rc = dev_alloc(e, &e->d_mstate, (size_t)1, e->multi_allocs))) return rc;
    if (!e->h_mstate) HIPCHK(e, hipHostMalloc((void **)&e->h_mstate, sizeof(MState), hipHostMallocDefault));
    { // the score memo: one word per (spec, node), resident for the whole simulation -- when it fits (CCSIM_MULTI_MEMO_MB caps it, 0 = off)
        size_t cap_mb = 65536, free_b = 0, total_b = 0;
        if (const char *f = getenv("CCSIM_MULTI_MEMO_MB")) cap_mb = (size_t)(atoll(f) > 0 ? atoll(f) : 0);
        const size_t bytes = NP * (size_t)n_pods * sizeof(uint16_t);
        // (a word holds TotalScore + 1 in 11 bits: the profile's weights must leave it there -- the default profile's sum is 8 -- and the
        // lean scan reads four nodes' words / labels per load: the padded length is a multiple of four -- kTile is)
        const bool word_fits = 100ll * ((int64_t)pf.w_taint + pf.w_nodeaffinity + pf.w_fit + pf.w_balanced + pf.w_imagelocality) + 1 <= (int64_t)kMemoScoreMask;
        if (cap_mb && word_fits && NP % 4 == 0 && hipMemGetInfo(&free_b, &total_b) == hipSuccess && bytes <= (cap_mb << 20) && bytes <= free_b / 2) {
            if ((rc = dev_alloc(e, &e->d_memo, NP * (size_t)n_pods, e->multi_allocs, false))) return rc;
            if ((rc = dev_alloc(e, &e->d_memo_stamp, 2 * (size_t)n_pods, e->multi_allocs, false))) return rc;
            if ((rc = dev_alloc(e, &e->d_mtouched, (size_t)kMTouched, e->multi_allocs))) return rc;
        }
    }
    for (int sl = 0; sl < kMTsc; sl++) e->tsc_label[sl] = slot_col[sl] >= 0 ? e->dev_label_ptrs[(size_t)slot_col[sl]] : nullptr;
    e->multi_prof = make_devpod(e, &pods[0]); // profile-level constants; the per-pod switches come from MPod
    e->multi_prof.w_img = any_img ? pf.w_imagelocality : 0; // (a spec without image scores has 0 in its class's static words: the weight does nothing there)
    if (e->multi_prof.gen_score) return fail(e, -ENOSYS, "several pod specs: scoring resource lists beyond cpu / memory");
    if (100ll * ((int64_t)pf.w_taint + pf.w_nodeaffinity + pf.w_fit + pf.w_balanced + pf.w_imagelocality) >= (1ll << 21))
        return fail(e, -ENOSYS, "several pod specs: plugin weights too large for the scan's packed 32-bit keys");
    e->multi_window = kMWindowMax;
    if (const char *f = getenv("CCSIM_MULTI_WINDOW")) e->multi_window = atoi(f) >= 1 && atoi(f) <= kMWindowMax ? atoi(f) : kMWindowMax; // tuning knob
    HIPCHK(e, hipStreamSynchronize(e->stream));
    e->n_pods = n_pods;
    e->multi_next = 0;
    e->multi = true;
    e->have_pod = true;
    return 0;
}

static MultiArgs multi_args(ccsim_engine *e) {
    MultiArgs a{};
    const DevCols &c = e->cols;
    a.c = PersistCols{{c.a32[0], c.a32[1]}, {c.r32[0], c.r32[1]}, {c.z32[0], c.z32[1]}, c.alloc_pods, c.pod_count, c.placed_cnt, c.stat,
                      {c.req[0], c.req[1]}, c.nz_mcpu, c.nz_mem, c.n_pad, c.global_offset, c.mem_shift};
    a.prof = e->multi_prof, a.st = e->d_mstate, a.pods = e->d_mpods, a.n_pods = e->n_pods;
    a.stat_cls = e->d_stat_cls, a.n_pad = e->n_pad;
    a.tsc_label[0] = e->tsc_label[0], a.tsc_label[1] = e->tsc_label[1];
    a.tbl_pool = e->d_tbl_pool, a.present_pool = e->d_present_pool, a.inc_pool = e->d_inc_pool, a.anti_bits = e->d_anti_bits;
    a.partials = e->d_mpartials, a.n_blocks = e->m_blocks, a.cands = e->d_mcands, a.log = e->d_log, a.per_spec = e->d_per_spec;
    a.window = e->multi_window < e->n_pods ? e->multi_window : e->n_pods;
    a.memo = e->d_memo, a.memo_stamp = e->d_memo_stamp, a.touched = e->d_mtouched, a.vsync = e->d_mvsync;
    return a;
}

static void launch_multi_window(ccsim_engine *e, const MultiArgs &a) {
    const int chunks = ((a.window + kMPodChunk - 1) / kMPodChunk) * kMLeanPer; // (grid.y: lean workgroups, ccsim_multi.h kMLeanChunk)
    hipLaunchKernelGGL(k_multi_scan, dim3((unsigned)e->m_blocks, (unsigned)chunks), dim3(kThreads), 0, e->stream, a);
    hipLaunchKernelGGL(k_multi_select, dim3((unsigned)a.window), dim3(64), 0, e->stream, a);
    hipLaunchKernelGGL(k_multi_commit_par, dim3(kMParGroups), dim3(kMParThreads), 0, e->stream, a); // (or, as its wave 0, the in-order commit: MState::seq_windows)
    // the placed specs' spread masks; the touched nodes' memo words, for every stamped spec
        hipLaunchKernelGGL(k_multi_refresh, dim3((unsigned)((a.n_pods + kMRefreshThreads - 1) / kMRefreshThreads), (unsigned)kMTouched),
                           dim3(kMRefreshThreads), 0, e->stream, a);
}

// begin a multi-spec run (or one cycle: single_pod >= 0) on the current columns
static int begin_multi(ccsim_engine *e, int64_t max_limit, int64_t log_cap, int32_t single_pod) {
    HIPCHK(e, hipSetDevice(e->device));
    e->extras_dirty = true;
    if (log_cap != e->log_cap) {
        if (e->d_log) HIPCHK(e, hipFree(e->d_log));
        e->d_log = nullptr, e->log_cap = 0;
        drop_graph(e);
        if (log_cap > 0) {
            HIPCHK(e, hipMalloc((void **)&e->d_log, sizeof(int32_t) * (size_t)log_cap));
            e->log_cap = log_cap;
        }
    }
    MState st{};
    st.limit = max_limit, st.single_pod = single_pod, st.stop_spec = -1, st.winner = -1, st.log_cap = e->log_cap;
    if (const char *f = getenv("CCSIM_MULTI_SEQ")) st.seq_windows = atoi(f) ? 1 << 30 : 0; // tuning knob: always the in-order commit
    st.next_pod = single_pod >= 0 ? single_pod : e->multi_next;
    int64_t wn = e->multi_window < e->n_pods ? e->multi_window : e->n_pods;
    if (max_limit > 0 && max_limit < wn) wn = max_limit;
    if (single_pod >= 0) wn = 1;
    st.win_n = (int32_t)wn;
    *e->h_mstate = st;
    HIPCHK(e, hipMemcpyAsync(e->d_mstate, e->h_mstate, sizeof(MState), hipMemcpyHostToDevice, e->stream));
    HIPCHK(e, hipMemsetAsync(e->cols.placed_cnt, 0, sizeof(int32_t) * (size_t)e->n_pad, e->stream));
    HIPCHK(e, hipMemsetAsync(e->d_per_spec, 0, sizeof(int32_t) * (size_t)e->n_pods, e->stream));
    HIPCHK(e, hipMemsetAsync(e->d_mvsync, 0, sizeof(int32_t) * (size_t)kMVsyncWords, e->stream)); // (the commit's workgroups leave it zeroed; a run that was aborted may not have)
    // the memo rows describe the columns as the last window left them; between runs anything may have touched the columns
    // (ccsim_reset_state, another pod set's run): every row starts unstamped (-1, -1) and is filled by its spec's first scan
    if (e->d_memo_stamp) HIPCHK(e, hipMemsetAsync(e->d_memo_stamp, 0xff, sizeof(int32_t) * 2 * (size_t)e->n_pods, e->stream));
    hipLaunchKernelGGL(k_multi_masks, dim3((unsigned)((e->n_pods + 255) / 256)), dim3(256), 0, e->stream, multi_args(e)); // the specs' per-domain spread masks, from the tables as they stand
    e->kernel_ms = 0, e->pass_kernel_ms = 0, e->pass_launches = 0;
    e->begun = false; // (the single-spec run state knows nothing of this run)
    return 0;
}

static int read_mstate(ccsim_engine *e) {
    HIPCHK(e, hipMemcpyAsync(e->h_mstate, e->d_mstate, sizeof(MState), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    return 0;
}

static int run_multi(ccsim_engine *e, int64_t max_limit, ccsim_report *out) {
    int rc = begin_multi(e, max_limit, out->log ? out->log_cap : 0, -1);
    if (rc) return rc;
    const MultiArgs a = multi_args(e);
    const int per_sync = e->rounds_per_sync > 0 ? e->rounds_per_sync : 16;
    int idle = 0;
    for (;;) {
        const int64_t placed0 = e->h_mstate->placed;
        HIPCHK(e, hipEventRecord(e->ev0, e->stream));
        if (e->use_graph) {
            if (!e->graph_exec || e->graph_mode != 100 || e->graph_rounds != per_sync) {
                drop_graph(e);
                HIPCHK(e, hipStreamBeginCapture(e->stream, hipStreamCaptureModeThreadLocal));
                for (int w = 0; w < per_sync; w++) launch_multi_window(e, a);
                HIPCHK(e, hipStreamEndCapture(e->stream, &e->graph));
                HIPCHK(e, hipGraphInstantiate(&e->graph_exec, e->graph, nullptr, nullptr, 0));
                e->graph_mode = 100, e->graph_rounds = per_sync;
            }
            HIPCHK(e, hipGraphLaunch(e->graph_exec, e->stream));
        } else
            for (int w = 0; w < per_sync; w++) launch_multi_window(e, a);
        HIPCHK(e, hipGetLastError());
        HIPCHK(e, hipEventRecord(e->ev1, e->stream));
        if ((rc = read_mstate(e))) return rc;
        float ms = 0;
        HIPCHK(e, hipEventElapsedTime(&ms, e->ev0, e->ev1));
        e->kernel_ms += ms;
        if (e->h_mstate->done) break;
        idle = e->h_mstate->placed == placed0 ? idle + 1 : 0;
        if (idle >= 4) return fail(e, -EIO, "multi-spec simulation made no progress in %d windows", 4 * per_sync);
    }
    const MState &st = *e->h_mstate;
    e->multi_next = st.next_pod;
    out->placed = st.placed;
    out->stop = st.done == DONE_LIMIT ? CCSIM_STOP_LIMIT : CCSIM_STOP_UNSCHEDULABLE;
    out->stop_spec = st.stop_spec;
    out->rounds = st.rounds;
    out->scans = st.windows;
    out->evaluated_total = st.rounds * e->n_global;
    out->last_feasible = st.last_feasible;
    out->kernel_ns = (int64_t)(e->kernel_ms * 1e6);
    out->pass_kernel_ns = 0, out->pass_launches = st.stops; // windows that ended early (the exact validation declined to go on)
    out->bytes_per_scan = (int64_t)(36 + 4) * e->n; // per window: the narrow columns once + one static word per (pod, node) of the window (+ one memo word with the score memo)
    memset(out->hist, 0, sizeof(out->hist));
    out->n_code_unschedulable = 0;
    if (out->hist_taintset)
        for (int i = 0; i < out->hist_taintset_cap; i++) out->hist_taintset[i] = 0;
    if (out->per_node_count) {
        if (out->per_node_cap < e->n) return fail(e, -EINVAL, "per_node_cap too small");
        HIPCHK(e, hipMemcpyAsync(out->per_node_count, e->cols.placed_cnt, sizeof(int32_t) * (size_t)e->n, hipMemcpyDeviceToHost, e->stream));
    }
    out->per_node_filled_width = out->per_node_count ? 4 : 0;
    if (out->per_spec_count) {
        if (out->per_spec_cap < e->n_pods) return fail(e, -EINVAL, "per_spec_cap too small");
        HIPCHK(e, hipMemcpyAsync(out->per_spec_count, e->d_per_spec, sizeof(int32_t) * (size_t)e->n_pods, hipMemcpyDeviceToHost, e->stream));
    }
    out->log_len = 0;
    if (out->log && e->d_log) {
        int64_t len = st.placed < e->log_cap ? st.placed : e->log_cap;
        if (len > out->log_cap) len = out->log_cap;
        if (len > 0) HIPCHK(e, hipMemcpyAsync(out->log, e->d_log, sizeof(int32_t) * (size_t)len, hipMemcpyDeviceToHost, e->stream));
        out->log_len = len;
    }
    if (st.done == DONE_UNSCHEDULABLE && st.stop_spec >= 0) { // the failing pod's FitError (types.go:787-836)
        HIPCHK(e, hipMemsetAsync(e->d_hist, 0, sizeof(unsigned long long) * (CCSIM_NREASON + 1), e->stream));
        HIPCHK(e, hipMemsetAsync(e->d_hist_ts, 0, sizeof(unsigned long long) * (size_t)e->max_taintsets, e->stream));
        MultiHistArgs h{};
        h.m = a, h.pod = st.stop_spec, h.sreason_cls = e->d_sreason_cls, h.taintset_id = e->cols.taintset_id;
        h.alloc[0] = e->cols.alloc[0], h.alloc[1] = e->cols.alloc[1], h.n = e->n;
        h.hist = e->d_hist, h.hist_ts = e->d_hist_ts, h.hist_code = e->d_hist_code;
        int64_t hb = (e->n + kThreads - 1) / kThreads;
        if (hb > 1024) hb = 1024;
        hipLaunchKernelGGL(k_multi_hist, dim3((unsigned)hb), dim3(kThreads), 0, e->stream, h);
        HIPCHK(e, hipGetLastError());
        std::vector<unsigned long long> hh(CCSIM_NREASON + 1), ht((size_t)e->max_taintsets);
        HIPCHK(e, hipMemcpyAsync(hh.data(), e->d_hist, sizeof(unsigned long long) * hh.size(), hipMemcpyDeviceToHost, e->stream));
        HIPCHK(e, hipMemcpyAsync(ht.data(), e->d_hist_ts, sizeof(unsigned long long) * ht.size(), hipMemcpyDeviceToHost, e->stream));
        HIPCHK(e, hipStreamSynchronize(e->stream));
        for (int i = 0; i < CCSIM_NREASON; i++) out->hist[i] = (int64_t)hh[i];
        out->n_code_unschedulable = (int64_t)hh[CCSIM_NREASON];
        const int nts = e->cls_taintsets[(size_t)e->h_mpods[(size_t)st.stop_spec].cls];
        if (out->hist_taintset)
            for (int i = 0; i < nts && i < out->hist_taintset_cap; i++) out->hist_taintset[i] = (int64_t)ht[i];
    }
    HIPCHK(e, hipStreamSynchronize(e->stream));
    return 0;
}

// measurement aid: why the windows of the last multi-spec run ended (k_multi_commit stop reasons 0..7)
extern "C" int ccsim_debug_coupled(ccsim_engine *e, int64_t *out8) {
    if (!e || !out8) return -EINVAL;
    for (int i = 0; i < 16; i++) out8[i] = 0;
    if (e->cw_ok && e->cw_work.prof) { // CCSIM_CW_PROF=1: 10 ns ticks of the decide kernel's wave 0 per phase, summed over the last run
        HIPCHK(e, hipSetDevice(e->device));
        HIPCHK(e, hipStreamSynchronize(e->stream));
        HIPCHK(e, hipMemcpy(out8 + 8, e->cw_work.prof, sizeof(int64_t) * 8, hipMemcpyDeviceToHost));
    }
    out8[0] = e->have_pod && e->cw_ok ? 1 : 0;
    if (e->h_state && e->begun) out8[1] = e->h_state->cw_windows, out8[2] = e->h_state->cw_fallback, out8[5] = e->h_state->cw_fast_windows, out8[6] = e->h_state->cw_full_windows, out8[7] = e->h_state->cw_swept;
    out8[3] = e->cw_ok ? e->cw_plan.window : 0, out8[4] = e->cw_ok ? e->cw_plan.list_len : 0;
    return 0;
}

// measurement aid: which form the sharded runs of this engine took
extern "C" int ccsim_debug_dist(ccsim_engine *e, int64_t *out8) {
    if (!e || !out8) return -EINVAL;
    for (int i = 0; i < 8; i++) out8[i] = 0;
    out8[0] = e->mbox_ready ? 1 : 0, out8[1] = e->mb_go, out8[2] = e->mb_abandoned, out8[3] = e->dist_last_form, out8[4] = e->mb_launches;
    out8[5] = e->comm_ranks;
    return 0;
}

// measurement aid: which form the last sampled search took (ccsim_sampled.h)
extern "C" int ccsim_debug_sampled(ccsim_engine *e, int64_t *out8) {
    if (!e || !out8) return -EINVAL;
    for (int i = 0; i < 16; i++) out8[i] = 0;
    if (e->d_sb_prof && (e->sb_run || e->sz_run || e->sf_run)) {
        HIPCHK(e, hipSetDevice(e->device));
        HIPCHK(e, hipStreamSynchronize(e->stream));
        HIPCHK(e, hipMemcpy(out8 + 8, e->d_sb_prof, sizeof(int64_t) * 8, hipMemcpyDeviceToHost));
    }
    if (e->h_state && e->begun && e->sz_run) { // the form for a hard spread constraint over zones (ccsim_sampled_zone.h): [1] = 2
        out8[0] = 1, out8[1] = 2, out8[2] = e->h_state->sb_cycles, out8[3] = e->h_state->sb_laps, out8[5] = e->sb_shift, out8[6] = e->sb_blocks, out8[7] = e->smp_K;
        return 0;
    }
    if (e->h_state && e->begun && e->sf_run) { // the full search on the summaries (ccsim_search_full.h): [1] = 3, [3] = cycles
        out8[0] = 1, out8[1] = 3, out8[2] = e->h_state->sb_cycles, out8[3] = e->h_state->sb_laps, out8[5] = e->sb_shift, out8[6] = e->sb_blocks;
        return 0;
    }
    if (!e->h_state || !e->begun || !e->sb_run) return 0;
    out8[0] = 1, out8[1] = e->sb_laps ? (e->sf_handover ? 4 : 1) : 0, out8[2] = e->h_state->sb_cycles, out8[3] = e->h_state->sb_laps, out8[4] = e->h_state->sb_slow;
    out8[5] = e->sb_shift, out8[6] = e->sb_blocks, out8[7] = e->smp_K;
    return 0;
}

extern "C" int ccsim_debug_multi_stops(ccsim_engine *e, int64_t *out8) {
    if (!e || !out8 || !e->h_mstate) return -EINVAL;
    for (int i = 0; i < 8; i++) out8[i] = e->h_mstate->stop_count[i];
    if (getenv("CCSIM_MULTI_PROF"))
        for (int i = 0; i < 8; i++) out8[i] = e->h_mstate->prof[i];
    return 0;
}

extern "C" int ccsim_debug_multi_memo(ccsim_engine *e, int64_t *out4) {
    if (!e || !out4 || !e->h_mstate) return -EINVAL;
    for (int i = 0; i < 4; i++) out4[4 + i] = e->h_mstate->scan_prof[i];
    out4[0] = e->multi && e->d_memo ? 1 : 0;
    out4[1] = e->h_mstate->memo_scans, out4[2] = e->h_mstate->full_scans;
    out4[3] = e->multi && e->d_memo ? (int64_t)e->n_pad * e->n_pods * (int64_t)sizeof(uint16_t) : 0;
    return 0;
}

extern "C" int ccsim_schedule_pod(ccsim_engine *e, int32_t pod_idx, ccsim_cycle *out) {
    if (!e || !out) return -EINVAL;
    if (!e->multi) return pod_idx == 0 ? ccsim_schedule_one(e, out) : fail(e, -EINVAL, "one pod spec is set: pod_idx must be 0");
    if (pod_idx < 0 || pod_idx >= e->n_pods) return fail(e, -EINVAL, "pod_idx out of range");
    // (per-node / per-spec result counters restart: this is one cycle at the SchedulePod seam, not a run)
    int rc = ensure_cols(e);
    if (rc) return rc;
    if ((rc = begin_multi(e, 0, 0, pod_idx))) return rc;
    const MultiArgs a = multi_args(e);
    out->node = -1, out->evaluated_nodes = (int32_t)e->n_global, out->feasible_nodes = 0;
    for (int tries = 0; tries < 8; tries++) {
        launch_multi_window(e, a);
        HIPCHK(e, hipGetLastError());
        if ((rc = read_mstate(e))) return rc;
        if (e->h_mstate->done) break;
    }
    if (!e->h_mstate->done) return fail(e, -EIO, "scheduling cycle did not converge");
    if (e->h_mstate->done == DONE_UNSCHEDULABLE) return 0; // FitError: node stays -1
    out->node = e->h_mstate->winner;
    out->feasible_nodes = e->h_mstate->last_feasible;
    return 0;
}
