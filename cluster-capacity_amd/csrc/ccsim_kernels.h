// ccsim_kernels.h -- CDNA4 (gfx950) kernels of the batched placement engine.
//
// Integer / bitmask / fp64 work over structure-of-arrays node columns resident in HBM; no MFMA
// (nothing here is a contraction).  Layout and launch shape are chosen for the memory system:
//   * every column is a dense array in canonical node order, padded to a multiple of TILE nodes;
//     a thread owns 2 consecutive nodes so int64 columns are read with 16-byte loads and int32
//     columns with 8-byte loads: one 1 KiB / 512 B fully coalesced request per wave instruction;
//   * a block owns one CONTIGUOUS chunk of nodes for the whole run, so the same XCD (block b ->
//     XCD b % 8) re-reads the same lines every round and finds them in its own L2 / the MALL;
//   * feasibility is a wave64 __ballot (popcount = feasible count), the argmax is a packed
//     (score, position) 64-bit key reduced with DPP/shuffle max per wave, LDS across the 4 waves,
//     one 16-byte partial record per block -- no atomics on the hot path;
//   * cross-block agreement happens at kernel boundaries (cheaper than an in-kernel grid barrier
//     on this chip): k_scan (grid) -> k_final (1 block: reduce partials, decide, commit).
//
// Reference arithmetic restated here (file:line under vendor/k8s.io/kubernetes/pkg/scheduler):
//   fitsRequest                 framework/plugins/noderesources/fit.go:564-660
//   leastRequestedScore         framework/plugins/noderesources/least_allocated.go:30-61
//   balancedResourceScorer      framework/plugins/noderesources/balanced_allocation.go:146-180
//   DefaultNormalizeScore       framework/plugins/helper/normalize_score.go:28-56
//   weight & sum                framework/runtime/framework.go:1214-1238
//   selectHost                  schedule_one.go:894-941 (canonical tie-break: lowest position)
//   NodeInfo.update             framework/types.go:409-428
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <limits.h>

namespace ccsim {

constexpr int kMaxRes = 11;
constexpr int kMaxExtra = 9;       // ephemeral + 8 scalars
constexpr int kMaxTsc = 8;         // hard topology spread constraints per pod
constexpr int kThreads = 256;      // 4 waves
constexpr int kNodesPerThread = 2; // 16-byte loads on int64 columns
constexpr int kTile = kThreads * kNodesPerThread;
constexpr int kMaxGrid = 2048;     // upper bound of the scan grid (partial arrays are sized for it)
constexpr int kIdxBits = 40;
constexpr uint64_t kIdxMask = (1ull << kIdxBits) - 1;
// the packed static word of a node for the current pod spec (k_static): bit 31 static filters passed | bits 30..20
// untolerated PreferNoSchedule taints | bits 19..13 ImageLocality score (0..100) | bits 12..0 sum of matching preferred weights
constexpr int kStatOkBit = 31, kStatCntShift = 20, kStatImgShift = 13;
constexpr uint32_t kStatAffMask = (1u << kStatImgShift) - 1, kStatCntMask = 0x7ffu, kStatImgMask = 0x7fu;

enum { DONE_RUNNING = 0, DONE_UNSCHEDULABLE = 1, DONE_LIMIT = 2 };

// node columns (device pointers), passed by value
struct DevCols {
    const int64_t *alloc[kMaxRes];
    int64_t *req[kMaxRes];
    const int32_t *alloc_pods;
    int32_t *pod_count;
    int64_t *nz_mcpu, *nz_mem;
    const uint32_t *stat;  // static word per node for the current pod spec
    const uint8_t *sreason; // which static filter rejected the node (0 none 1 unschedulable 2 taint 3 affinity 4 host ports 5 a volume plugin)
    const int32_t *taintset_id;
    int32_t *placed_cnt;   // simulated pods per node
    int64_t n;             // real node count of this shard
    int64_t n_pad;         // padded to kTile
    int64_t global_offset; // canonical index of node 0 of this shard
    // Narrow mirrors of the six int64 columns the full pass streams (lossless: enabled only when every cpu value
    // stays < 2^31 and every memory value is a multiple of 2^mem_shift with value >> mem_shift < 2^31 for the whole
    // run).  The full-pass kernels read 36 B per node instead of 60; every writer of the wide columns also updates
    // the mirror.  Wide columns stay the canonical state (commit gathers, histogram, read-back, reset).
    int32_t narrow, mem_shift;
    const int32_t *a32[2]; // alloc cpu, alloc mem >> mem_shift
    int32_t *r32[2];       // requested
    int32_t *z32[2];       // non-zero requested
    // Commit rows (batched mode on the narrow mirrors): one 64-byte row per node holding everything a run-down needs --
    // a level node costs the commit pass one row read and half a row written instead of nine column gathers and eleven
    // scattered stores (each a full cache line of traffic).  Rows are built at the start of a batched run and are the
    // ONLY state k_level_commit<.., NARROW> touches; the columns are brought up to date (k_rows_flush) before a full pass
    // reads them and when the run ends.   int32[16]: a0 a1 alloc_pods stat | r0 r1 z0 z1 | pods placed - - | - - - -
    int32_t *rows;
};
constexpr int kRowWords = 16;

// the six streamed columns of a thread's node pair, widened back to the exact int64 values
struct Cols6 {
    longlong2 A0, A1, R0, R1, Z0, Z1;
};
template <bool NARROW>
__device__ __forceinline__ Cols6 load_cols6(const DevCols &c, int64_t i0) {
    Cols6 o;
    if (NARROW) {
        const int2 a0 = *reinterpret_cast<const int2 *>(c.a32[0] + i0), a1 = *reinterpret_cast<const int2 *>(c.a32[1] + i0);
        const int2 r0 = *reinterpret_cast<const int2 *>(c.r32[0] + i0), r1 = *reinterpret_cast<const int2 *>(c.r32[1] + i0);
        const int2 z0 = *reinterpret_cast<const int2 *>(c.z32[0] + i0), z1 = *reinterpret_cast<const int2 *>(c.z32[1] + i0);
        const int sh = c.mem_shift;
        o.A0 = {a0.x, a0.y}, o.R0 = {r0.x, r0.y}, o.Z0 = {z0.x, z0.y};
        o.A1 = {(int64_t)a1.x << sh, (int64_t)a1.y << sh};
        o.R1 = {(int64_t)r1.x << sh, (int64_t)r1.y << sh};
        o.Z1 = {(int64_t)z1.x << sh, (int64_t)z1.y << sh};
    } else {
        o.A0 = *reinterpret_cast<const longlong2 *>(c.alloc[0] + i0);
        o.A1 = *reinterpret_cast<const longlong2 *>(c.alloc[1] + i0);
        o.R0 = *reinterpret_cast<const longlong2 *>(c.req[0] + i0);
        o.R1 = *reinterpret_cast<const longlong2 *>(c.req[1] + i0);
        o.Z0 = *reinterpret_cast<const longlong2 *>(c.nz_mcpu + i0);
        o.Z1 = *reinterpret_cast<const longlong2 *>(c.nz_mem + i0);
    }
    return o;
}
// keep the mirror of one node's dynamic columns in step with the wide columns (sparse writers: commits)
__device__ __forceinline__ void store_mirror(const DevCols &c, int64_t i, int64_t r_cpu, int64_t r_mem, int64_t z_cpu, int64_t z_mem) {
    if (!c.narrow) return;
    c.r32[0][i] = (int32_t)r_cpu, c.r32[1][i] = (int32_t)(r_mem >> c.mem_shift);
    c.z32[0][i] = (int32_t)z_cpu, c.z32[1][i] = (int32_t)(z_mem >> c.mem_shift);
}

// pod + profile constants, passed by value
struct DevPod {
    int64_t req[kMaxRes];
    int64_t nz_mcpu, nz_mem;
    int32_t fit_enabled;
    int32_t all_zero_req; // fit.go:578-583 early return
    int32_t nx;           // active extra columns (checked by the Fit filter)
    int32_t xcol[kMaxExtra];
    int32_t w_taint, w_aff, w_fit, w_bal;
    int32_t w_img;            // ImageLocality weight (0 when the pod's images are on no node: the plugin scores 0 everywhere)
    int32_t fit_cpu, fit_mem; // resource present in the LeastAllocated list
    int64_t fit_w_cpu, fit_w_mem;
    int32_t bal_cpu, bal_mem; // resource present in the BalancedAllocation list
    int32_t ncol;
    // Resource lists beyond cpu / memory (ephemeral-storage, scalar resources: resource_allocation.go:97-110) take the
    // general evaluation (dynamic_score_gen, int64 columns only).  A column >= 2 is read through the extra-column slot
    // (xcol) that holds it; a scalar the pod does not request has no slot: bypassed, as the reference does.
    int32_t gen_score;
    int32_t n_fit, fit_col[kMaxRes];
    int64_t fit_w[kMaxRes];
    int32_t n_bal, bal_col[kMaxRes];
};

// Hard PodTopologySpread constraints (P/podtopologyspread/filtering.go:235-356): TpValueToMatchNum lives as
// one count table per constraint in HBM (L2-resident: #domains x 4 B), updated by the commit.
struct DevPts {
    int32_t n;
    int32_t max_skew[kMaxTsc], min_domains[kMaxTsc], self_match[kMaxTsc];
    int32_t n_present[kMaxTsc];   // len(TpValueToMatchNum[c]): domains holding >= 1 counted node (static)
    const int32_t *label[kMaxTsc]; // topology value id per node (0 = key absent)
    int32_t *tbl[kMaxTsc];         // match count per value id
    const uint8_t *elig;           // per node: bit0 has all hard keys (filtering.go:267-270), bit 1+c counted for c (:274-277)
};

// Soft (ScheduleAnyway) PodTopologySpread constraints -> Score (P/podtopologyspread/scoring.go:61-265).
// TopologyValueToPodCounts lives as one count table per constraint in HBM; which domains are "candidates" this
// cycle (hold a feasible, non-ignored node) is recorded by the scan in an epoch-stamped flag table.
struct DevSoft {
    int32_t n, w; // soft constraints; plugin weight (0 = Score off)
    int32_t max_skew[kMaxTsc], self_match[kMaxTsc], is_hostname[kMaxTsc], n_domains[kMaxTsc];
    int32_t nocredit[kMaxTsc];        // value id that stands for "the node lacks the key" under requireAllTopologies = false (ccsim.h missing_value): a
                                      // domain like any other when sizes and counts are taken, no score for this constraint (scoring.go:210); 0 = none
    const int32_t *label[kMaxTsc];
    int32_t *tbl[kMaxTsc];            // matching pods per domain, over counted nodes
    int32_t *flag[kMaxTsc];           // epoch of the last scan that saw a feasible non-ignored node in the domain
    const int32_t *existing[kMaxTsc]; // hostname constraints: matching pods already on the node (NULL = 0)
    const int32_t *pod_count0;        // len(NodeInfo.Pods) of the loaded snapshot: pod_count - pod_count0 = clones on the node
    const uint8_t *elig;              // per node: bit0 has all soft keys (scoring.go:84-88), bit 1+c counted for c
};

// InterPodAffinity (P/interpodaffinity/filtering.go:204-432, scoring.go:81-290): the topology-pair maps live as
// one int64 table per distinct topology key in HBM, updated by the commit (a clone is an existing pod next cycle).
constexpr int kMaxIpaKeys = 4, kMaxIpaTerms = 8;
struct DevIpa {
    int32_t on, filter_on, w; // plugin present; Filter enabled in the profile; score weight (0 = off)
    int32_t n_keys, n_aff, n_anti, self_aff;
    int32_t aff_key[kMaxIpaTerms], anti_key[kMaxIpaTerms];
    int32_t aff_terms_on_key[kMaxIpaKeys];  // # required affinity terms using the key
    int32_t anti_self_on_key[kMaxIpaKeys];  // # required anti-affinity terms using the key that match the pod itself
    int32_t self_entries[kMaxIpaKeys];
    int64_t score_self[kMaxIpaKeys];
    const int32_t *label[kMaxIpaKeys];
    int64_t *aff[kMaxIpaKeys], *anti[kMaxIpaKeys], *exist[kMaxIpaKeys], *score[kMaxIpaKeys];
};

struct DevState {
    int32_t pts_min_a[kMaxTsc]; // CriticalPaths[c][0].MatchNum the pending scan assumed (filtering.go:298-305)
    int64_t ipa_aff_total, ipa_exist_total, ipa_entries; // len(affinityCounts), len(existingAntiAffinityCounts), PreScore hits
    int64_t ipa_min_a, ipa_max_a;                        // NormalizeScore min / max the pending scan assumed (scoring.go:258-290)
    // soft PodTopologySpread: what the pending scan assumed
    int64_t soft_size_a[kMaxTsc]; // topoSize per constraint (hostname: feasible minus ignored nodes), scoring.go:96-113
    double soft_w[kMaxTsc];       // TopologyNormalizingWeight = math.Log(size + 2)
    int64_t soft_min_a, soft_max_a;
    int64_t placed, limit, rounds, scans;
    int64_t winner; // global index committed by the last decide (-1 none)
    int32_t done, have_prev;
    int32_t mt_a, ma_a; // normalization maxima the pending scan result was computed with
    int32_t last_feasible, mode;
    int64_t log_cap;
    // CCSIM_MODE_BATCHED (ccsim_level.h): the level the next pass commits
    int64_t lvl_M;           // TotalScore of the level
    int64_t lvl_cut;         // commit only nodes with global index <= cut (normalization change inside the level)
    int64_t lvl_remaining;   // placements still allowed (limit - placed at level start)
    int64_t lvl_rank_prefix; // placements of this level that belong to lower-ranked shards
    int64_t lvl_c_mt, lvl_c_ma; // feasible holders of the normalization maxima when the level was found
    int32_t lvl_valid;       // 1 = the next pass commits level lvl_M
    int32_t lvl_prefix;      // 1 = ordered commit (limit inside the level, or placement log wanted)
    int32_t lvl_plan_only;   // 1 = the next pass only measures level lvl_M (no commit)
    int32_t last_evaluated;  // nodes the last cycle visited (sampled search; otherwise N)
    // batched mode: the score cache (LevelArgs::cscore).  lvl_full = 1: it is invalid, the next pass is a full one
    // (k_level_score); otherwise k_level_commit keeps it current and these counts track the feasible set:
    int64_t cur_nfeas;       // feasible nodes of this shard
    int64_t cur_c_mt, cur_c_ma; // this shard's feasible holders of the (global) normalization maxima mt_a / ma_a
    int32_t lvl_full, sb_laps; // (sb_laps: k_sb_laps -- laps of the ring evaluated; diagnostics, in what was padding)
    // percentageOfNodesToScore < 100 (schedule_one.go:610-723): the sampled search of the sequential mode
    int64_t smp_K;           // numFeasibleNodesToFind; 0 = every node is scored
    int64_t smp_start;       // nextStartNodeIndex
    int64_t smp_Fs;          // feasible nodes with index < smp_start (this cycle)
    int64_t smp_Ftotal;      // feasible nodes of the whole snapshot (this cycle)
    int64_t smp_stop;        // rotated position of the (K+1)-th feasible node = nodes visited; -1: all N were visited
    int64_t evaluated;       // sum of visited nodes over the cycles
    // ... on node-range SHARDS (n_ranks > 0; tests/sharded_sampled_model.py): a cycle is two exchanges.  Phase 0: every shard
    // counts its feasible nodes (all / those before the start index), the gathered counts give the cluster totals and the number
    // of feasible nodes in the shards before this one -- with that offset the rank-in-visiting-order formula of the scoring pass
    // holds unchanged.  Phase 1: the scoring pass on the selected nodes, max-loc exchange, decision (normalization maxima over
    // the selected nodes: assumed, verified on the gathered records, the scoring pass redone if they moved).
    int64_t smp_N;           // nodes of the whole snapshot (the visiting order wraps around at it)
    int64_t smp_off;         // feasible nodes in the shards before this one (this cycle)
    int32_t smp_phase;       // sharded runs: 0 = the next pass counts, 1 = the next pass scores
    int32_t smp_rank;        // this shard's rank
    // multi-kernel batched mode: several score levels per pass, committed blindly and validated afterwards (ccsim_level.h)
    int64_t lvl_Lo;          // the pending commit takes every node scoring >= lvl_Lo down to < lvl_Lo (== lvl_M: one level)
    int64_t prev_nfeas, prev_c_mt, prev_c_ma; // this shard's counts before the pending blind batch (restored by a roll-back)
    int32_t lvl_kb, lvl_kb_max; // levels per blind batch: now / at most (1 = the one-level-per-pass protocol)
    int32_t lvl_blind;       // the commit of this pass is a blind batch: validate it before its placements count
    int32_t lvl_rollback;    // the next pass undoes the batch of pass lvl_pass (a normalization maximum ran out of holders, or --max-limit was crossed inside it)
    int32_t lvl_pass, sb_slow; // stamp of the last committing pass; (sb_slow: k_sb_laps -- stretches re-evaluated node by node under their own maxima)  // (commit rows carry it next to the clones they took in it)
    int64_t lvl_ev;          // score level at which a rolled-back batch located its normalization event (-1: none): ccsim_level.h level_decide
    // windowed mode for topology-coupled plugins (ccsim_coupled.h)
    int32_t cw_fallback;     // 1 = the windowed mode gave up on this run: the one-pass-per-placement loop continues from the current state
    int32_t cw_windows;      // windows resolved so far
    int32_t cw_fast_windows, cw_full_windows; // ... of them by the lane-per-candidate kernel / by its 64-class form
    int32_t cw_sweeps, cw_swept;              // rounds that kernel resolved at once, and the placements in them (ccsim_coupled.h `sweep`)
    // the sampled search on resident block summaries (ccsim_sampled.h)
    int32_t sb_dirty;        // 1 = memo and summaries do not describe the columns under (mt_a, ma_a): k_sb_build runs before the next cycle
    int32_t sb_cycles;       // launches of k_sb_cycles / k_sb_laps that ran (diagnostics: did this path take the run)
    // persistent batched launch (ccsim_persist.h): the normalization maxima the launch started with (the next launch's hint)
    int32_t p_mt0, p_ma0;
};

// per-block result of one scan: 16 bytes
struct __attribute__((aligned(16))) Partial {
    uint64_t key;   // ((total+1) << 40) | (2^40-1 - global index); 0 = no feasible node
    uint32_t norm;  // (max prefer-count << 20) | max affinity-sum, over this block's feasible nodes
    uint32_t nfeas;
};

// exchange record (int64[32], see CCSIM_XCHG_WORDS).  Sequential mode uses the first four words (key, mt,
// ma combine with MAX, nfeas with SUM); batched mode adds the plan of the shard's own top level.
struct XRec {
    int64_t key, mt, ma, nfeas;
    int64_t c_mt, c_ma, committed, n_top;
    int64_t T, e_mt, e_ma, cut_mt;
    int64_t cut_ma;
    // sequential mode, topology-coupled plugins: what the global verification needs, and the topology value ids of this
    // shard's best node ("the winner's domain ids travel with it": every rank updates its replicated count tables)
    int64_t pts_min[kMaxTsc];
    int64_t ipa_mn, ipa_mx;
    int64_t win_elig;   // PodTopologySpread eligibility bits of the node
    int32_t win_pts_v[kMaxTsc];
    int32_t win_ipa_v[4];
    int64_t pad[2];     // sampled search on shards: counting pass [0] feasible nodes of the shard, [1] those before the start index
                        // (scoring pass: the visiting position + 1 of the node that cancels the search rides in the upper half of win_elig);
                        // ScheduleAnyway constraints: the best node's topology value ids (xrec_soft_ids)
};
static_assert(sizeof(XRec) == 32 * 8, "XRec must be CCSIM_XCHG_WORDS int64");
// Sequential mode with ScheduleAnyway spread constraints on shards: the nine words the batched mode uses for its level plan
// (c_mt .. cut_ma) carry what the PodTopologySpread score needs cluster-wide (scoring.go:118-265; tests/sharded_coupled_model.py):
//   [0] feasible nodes that have every soft key (the rest are ignored: score 0)   [1], [2] min / max raw score over them
//   [3] eligibility bits of the shard's best node                                 [4..8] the SET of candidate domains per
//   constraint as a bitmap (constraint c's domain v at bit soft_bit_offset(c) + v - 1; hostname constraints use [0] instead)
// and pad[0..1] hold that node's topology value ids of the soft constraints.  The decision verifies the assumed weights
// (log(size + 2), size = |union of the candidate sets|) and the assumed normalization range on the gathered records, exactly as
// the unsharded mode does on its own partials, and rescans if they moved.
constexpr int kXSoftWords = 9, kXSoftBitWords = 5, kXSoftBits = kXSoftBitWords * 64;
static_assert(offsetof(XRec, cut_ma) - offsetof(XRec, c_mt) == (kXSoftWords - 1) * 8, "nine contiguous words");
__host__ __device__ inline int soft_bit_offset(const DevSoft &p, int c) { // first bit of constraint c's candidate-domain set
    int off = 0;
    for (int q = 0; q < c; q++) off += p.is_hostname[q] ? 0 : p.n_domains[q];
    return off;
}
__host__ __device__ inline int64_t *xrec_soft(XRec &r) { return &r.c_mt; }
__host__ __device__ inline const int64_t *xrec_soft(const XRec &r) { return &r.c_mt; }
__host__ __device__ inline int32_t *xrec_soft_ids(XRec &r) { return reinterpret_cast<int32_t *>(&r.pad[0]); }
__host__ __device__ inline const int32_t *xrec_soft_ids(const XRec &r) { return reinterpret_cast<const int32_t *>(&r.pad[0]); }

// ------------------------------------------------------------------------------------------------
// Wave-wide reductions on the DPP network (row_shr 1/2/4/8 inside each row of 16 lanes, then row_bcast:15 / :31 across
// the four rows -- the gfx9 sequence LLVM's atomic optimizer emits): six VALU steps, result in lane 63, broadcast with
// v_readlane.  The ds_bpermute form (__shfl_xor) costs one LDS round trip (~100 cycles) per step and the steps are
// dependent: in the latency-bound kernels (one-block decisions, the persistent level kernel) that was microseconds.
#define CCSIM_DPP_STEP32(v, ident, op, ctrl, rmask)                                                   \
    {                                                                                                  \
        const uint32_t o_ = (uint32_t)__builtin_amdgcn_update_dpp((int)(ident), (int)(v), ctrl, rmask, 0xf, false); \
        v = op(v, o_);                                                                                \
    }
__device__ __forceinline__ uint32_t op_max_u32(uint32_t a, uint32_t b) { return a > b ? a : b; }
__device__ __forceinline__ uint32_t op_add_u32(uint32_t a, uint32_t b) { return a + b; }
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
    CCSIM_DPP_STEP32(v, 0u, op_max_u32, 0x111, 0xf) CCSIM_DPP_STEP32(v, 0u, op_max_u32, 0x112, 0xf)
    CCSIM_DPP_STEP32(v, 0u, op_max_u32, 0x114, 0xf) CCSIM_DPP_STEP32(v, 0u, op_max_u32, 0x118, 0xf)
    CCSIM_DPP_STEP32(v, 0u, op_max_u32, 0x142, 0xa) CCSIM_DPP_STEP32(v, 0u, op_max_u32, 0x143, 0xc)
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ uint32_t wave_sum_u32_dpp(uint32_t v) {
    CCSIM_DPP_STEP32(v, 0u, op_add_u32, 0x111, 0xf) CCSIM_DPP_STEP32(v, 0u, op_add_u32, 0x112, 0xf)
    CCSIM_DPP_STEP32(v, 0u, op_add_u32, 0x114, 0xf) CCSIM_DPP_STEP32(v, 0u, op_add_u32, 0x118, 0xf)
    CCSIM_DPP_STEP32(v, 0u, op_add_u32, 0x142, 0xa) CCSIM_DPP_STEP32(v, 0u, op_add_u32, 0x143, 0xc)
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
// 64-bit values travel as two DPP moves per step; the combine is a full 64-bit operation
__device__ __forceinline__ uint64_t dpp_move_u64(uint64_t ident, uint64_t v, const int ctrl_sel) {
    uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    const uint32_t ilo = (uint32_t)ident, ihi = (uint32_t)(ident >> 32);
    switch (ctrl_sel) { // (the control word must be an immediate)
    case 0: lo = __builtin_amdgcn_update_dpp((int)ilo, (int)lo, 0x111, 0xf, 0xf, false), hi = __builtin_amdgcn_update_dpp((int)ihi, (int)hi, 0x111, 0xf, 0xf, false); break;
    case 1: lo = __builtin_amdgcn_update_dpp((int)ilo, (int)lo, 0x112, 0xf, 0xf, false), hi = __builtin_amdgcn_update_dpp((int)ihi, (int)hi, 0x112, 0xf, 0xf, false); break;
    case 2: lo = __builtin_amdgcn_update_dpp((int)ilo, (int)lo, 0x114, 0xf, 0xf, false), hi = __builtin_amdgcn_update_dpp((int)ihi, (int)hi, 0x114, 0xf, 0xf, false); break;
    case 3: lo = __builtin_amdgcn_update_dpp((int)ilo, (int)lo, 0x118, 0xf, 0xf, false), hi = __builtin_amdgcn_update_dpp((int)ihi, (int)hi, 0x118, 0xf, 0xf, false); break;
    case 4: lo = __builtin_amdgcn_update_dpp((int)ilo, (int)lo, 0x142, 0xa, 0xf, false), hi = __builtin_amdgcn_update_dpp((int)ihi, (int)hi, 0x142, 0xa, 0xf, false); break;
    default: lo = __builtin_amdgcn_update_dpp((int)ilo, (int)lo, 0x143, 0xc, 0xf, false), hi = __builtin_amdgcn_update_dpp((int)ihi, (int)hi, 0x143, 0xc, 0xf, false); break;
    }
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t bcast63_u64(uint64_t v) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, 63), hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), 63);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t wave_max_u64(uint64_t v) {
#pragma unroll
    for (int s = 0; s < 6; s++) {
        const uint64_t o = dpp_move_u64(0ull, v, s);
        v = o > v ? o : v;
    }
    return bcast63_u64(v);
}
__device__ __forceinline__ int64_t wave_sum_i64(int64_t v) {
    uint64_t u = (uint64_t)v;
#pragma unroll
    for (int s = 0; s < 6; s++) u += dpp_move_u64(0ull, u, s);
    return (int64_t)bcast63_u64(u);
}
// value of lane `src` (wave-uniform index) in every lane: v_readlane, not an LDS permute
__device__ __forceinline__ int32_t lane_bcast_i32(int32_t v, int src) { return __builtin_amdgcn_readlane(v, src); }
__device__ __forceinline__ int64_t lane_bcast_i64(int64_t v, int src) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, src), hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((uint64_t)v >> 32), src);
    return (int64_t)(((uint64_t)hi << 32) | lo);
}

// ---- per-block partials ---------------------------------------------------------------------------------------
// Per-block partial results are naturally aligned 8-byte words written and read with AGENT-scope relaxed atomics
// (write-through sc1 stores, L1-bypassing sc1 loads): correct across a kernel boundary and also inside one launch
// ("8-B agent atomics both sides", MI355X_MICROARCH.md).
__device__ __forceinline__ void st_agent(uint64_t *p, uint64_t v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint64_t ld_agent(const uint64_t *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- exact scores without IEEE divides on the common path -------------------------------------------
// A pass is as much VALU- as HBM-bound if every node pays five fp64 divisions (~40 instructions each on
// CDNA: v_div_scale x2, v_rcp_f64, Newton FMAs, v_div_fmas, v_div_fixup).  Both scores divide by the
// node's allocatable cpu / memory, so ONE refined reciprocal per resource (v_rcp_f64 + one Newton step,
// relative error < 2^-45) serves all of them, and exactness is restored by construction:
//   * LeastAllocated is an integer floor: estimate with the reciprocal, fix up with an exact int64 remainder;
//   * BalancedAllocation truncates an fp64 value to an integer: the reciprocal-based value differs from the
//     IEEE one by < 1e-10, so unless it lies within 1e-9 of an integer the truncation is the same; in that
//     (probability ~2e-9) case the IEEE sequence is evaluated.
__device__ __forceinline__ double refined_rcp(double d) {
    const double r0 = __builtin_amdgcn_rcp(d);
    const double e = __builtin_fma(-d, r0, 1.0);
    return __builtin_fma(r0, e, r0);
}

struct NodeRcp {
    double cpu, mem; // ~1/alloc_cpu, ~1/alloc_mem (0 when the resource is absent)
};
__device__ __forceinline__ NodeRcp make_rcp(int64_t a_cpu, int64_t a_mem) {
    NodeRcp r;
    r.cpu = a_cpu != 0 ? refined_rcp((double)a_cpu) : 0.0;
    r.mem = a_mem != 0 ? refined_rcp((double)a_mem) : 0.0;
    return r;
}

// floor(num / cap) for 0 <= num <= 100 * cap, exact: reciprocal estimate (off by at most one) + integer fix-up
__device__ __forceinline__ int64_t div_small_quotient(int64_t num, int64_t cap, double rcp_cap) {
    int64_t q = (int64_t)((double)num * rcp_cap);
    const int64_t r = num - q * cap;
    if (r < 0) q -= 1;
    else if (r >= cap) q += 1;
    return q;
}

// least_allocated.go:52-61
__device__ __forceinline__ int64_t least_requested_score(int64_t requested, int64_t capacity, double rcp_cap) {
    if (capacity == 0) return 0;
    if (requested > capacity) return 0;
    return div_small_quotient((capacity - requested) * 100, capacity, rcp_cap);
}

// balanced_allocation.go:146-180, the IEEE sequence exactly as the reference evaluates it (slow path)
__device__ __noinline__ int64_t balanced_exact(int64_t x0, int64_t a0, int64_t x1, int64_t a1) {
    double f0 = (double)x0 / (double)a0, f1 = (double)x1 / (double)a1;
    f0 = f0 > 1 ? 1 : f0;
    f1 = f1 > 1 ? 1 : f1;
    const double std = fabs((f0 - f1) / 2);
    return (int64_t)((1 - std) * 100.0);
}

// Score pair for one node in a given dynamic state: returns LeastAllocated*w_fit + Balanced*w_bal.
__device__ __forceinline__ int64_t dynamic_score(const DevPod &p, const NodeRcp &rc, int64_t a_cpu, int64_t a_mem, int64_t r_cpu,
                                                 int64_t r_mem, int64_t z_cpu, int64_t z_mem) {
    int64_t total = 0;
    if (p.w_fit) {
        // resource_allocation.go:48-114 with useRequested=false: NonZeroRequested + non-zero pod request
        int64_t node_score = 0, weight_sum = 0;
        if (p.fit_cpu && a_cpu != 0) {
            node_score += least_requested_score(z_cpu + p.nz_mcpu, a_cpu, rc.cpu) * p.fit_w_cpu;
            weight_sum += p.fit_w_cpu;
        }
        if (p.fit_mem && a_mem != 0) {
            node_score += least_requested_score(z_mem + p.nz_mem, a_mem, rc.mem) * p.fit_w_mem;
            weight_sum += p.fit_w_mem;
        }
        // weights are small positive ints; 0 <= node_score <= 100 * weight_sum
        int64_t s = 0;
        if (weight_sum == 2) s = node_score >> 1;
        else if (weight_sum == 1) s = node_score;
        else if (weight_sum != 0) s = (int64_t)((uint32_t)node_score / (uint32_t)weight_sum);
        total += s * p.w_fit;
    }
    if (p.w_bal) {
        // balanced_allocation.go:146-180 with useRequested=true: Requested + raw pod request
        const bool c = p.bal_cpu && a_cpu != 0, m = p.bal_mem && a_mem != 0;
        int64_t score = 100; // fewer than two fractions: std = 0
        if (c && m) {
            const int64_t x0 = r_cpu + p.req[0], x1 = r_mem + p.req[1];
            double f0 = (double)x0 * rc.cpu, f1 = (double)x1 * rc.mem;
            f0 = f0 > 1 ? 1 : f0;
            f1 = f1 > 1 ? 1 : f1;
            const double y = (1 - fabs((f0 - f1) / 2)) * 100.0; // in [50, 100]
            const double t = floor(y), fr = y - t;
            if (fr > 1e-9 && fr < 1 - 1e-9) score = (int64_t)t;
            else score = balanced_exact(x0, a_cpu, x1, a_mem);
        }
        total += score * p.w_bal;
    }
    return total;
}

// The general form of both scores: any resource list (least_allocated.go:30-61, balanced_allocation.go:146-180 incl. the
// population standard deviation of MORE than two fractions: IEEE divide / multiply / add / sqrt in the reference's order,
// no contraction).  xa / xr: allocatable / requested of the extra-column slots p.xcol[0..nx).
template <int NX>
__device__ __forceinline__ bool gen_resource(const DevPod &p, int col, bool nonzero, int64_t a_cpu, int64_t a_mem, int64_t r_cpu, int64_t r_mem,
                                             int64_t z_cpu, int64_t z_mem, const int64_t *xa, const int64_t *xr, int64_t &alloc, int64_t &req) {
    alloc = 0, req = 0;
    if (col == 0) alloc = a_cpu, req = nonzero ? z_cpu + p.nz_mcpu : r_cpu + p.req[0];
    else if (col == 1) alloc = a_mem, req = nonzero ? z_mem + p.nz_mem : r_mem + p.req[1];
    else {
#pragma unroll
        for (int x = 0; x < (NX > 0 ? NX : 1); x++)
            if (NX > 0 && x < p.nx && p.xcol[x] == col) alloc = xa[x], req = xr[x] + p.req[col];
    }
    return alloc != 0; // resource_allocation.go:66-69: an absent resource does not take part
}

template <int NX>
__device__ __forceinline__ int64_t dynamic_score_gen(const DevPod &p, int64_t a_cpu, int64_t a_mem, int64_t r_cpu, int64_t r_mem, int64_t z_cpu,
                                                  int64_t z_mem, const int64_t *xa, const int64_t *xr) {
    int64_t total = 0;
    if (p.w_fit) {
        int64_t node_score = 0, weight_sum = 0;
        for (int i = 0; i < p.n_fit; i++) {
            int64_t alloc, req;
            if (!gen_resource<NX>(p, p.fit_col[i], true, a_cpu, a_mem, r_cpu, r_mem, z_cpu, z_mem, xa, xr, alloc, req)) continue;
            node_score += (req > alloc ? 0 : ((alloc - req) * 100) / alloc) * p.fit_w[i];
            weight_sum += p.fit_w[i];
        }
        total += (weight_sum ? node_score / weight_sum : 0) * p.w_fit;
    }
    if (p.w_bal) {
        double f0 = 0, f1 = 0, sum_f = 0; // (no per-lane array of fractions: it would be a scratch frame; the > 2 branch recomputes them)
        int m = 0;
        for (int i = 0; i < p.n_bal; i++) {
            int64_t alloc, req;
            if (!gen_resource<NX>(p, p.bal_col[i], false, a_cpu, a_mem, r_cpu, r_mem, z_cpu, z_mem, xa, xr, alloc, req)) continue;
            double f = (double)req / (double)alloc;
            f = f > 1 ? 1 : f;
            sum_f += f;
            if (m == 0) f0 = f; else if (m == 1) f1 = f;
            m++;
        }
        double std = 0;
        if (m == 2) std = fabs((f0 - f1) / 2);
        else if (m > 2) {
            const double mean = sum_f / (double)m;
            double sum = 0;
            for (int i = 0; i < p.n_bal; i++) {
                int64_t alloc, req;
                if (!gen_resource<NX>(p, p.bal_col[i], false, a_cpu, a_mem, r_cpu, r_mem, z_cpu, z_mem, xa, xr, alloc, req)) continue;
                double f = (double)req / (double)alloc;
                f = f > 1 ? 1 : f;
                sum = sum + (f - mean) * (f - mean);
            }
            std = sqrt(sum / (double)m);
        }
        total += (int64_t)((1 - std) * 100.0) * p.w_bal;
    }
    return total;
}

// a value every lane holds identically: tell the compiler, so that it lives in an SGPR
__device__ __forceinline__ int uni32(int v) { return __builtin_amdgcn_readfirstlane(v); }

// floor(100 * c / m) for c <= m, 0 < m < 2^13 (DefaultNormalizeScore, normalize_score.go:28-56) without a runtime division per
// node: m is a normalization maximum, uniform over the launch, so the magic below is loop-invariant (hoisted by the compiler).
// floor(n / m) == mulhi(n, ceil(2^32 / m)) whenever n * m < 2^32 (Granlund-Montgomery: the error term n (M m - 2^32) / (m 2^32)
// stays below 1 / m); with n = 100 c <= 100 m that holds for m <= 6553.  (c > m only happens under a stale ASSUMED maximum,
// whose pass is thrown away by the verification that follows.)
// The magic is derived with an fp64 division, not an integer one: it has no trap, so the compiler may hoist it out of loops and
// conditionals (an integer division by a value it cannot prove non-zero stays where it is -- measured: 63 VALU instructions
// per evaluation).  For 2 <= m < 2^13 the fp64 quotient 2^32 / m truncates to floor(2^32 / m) exactly (a non-integral
// quotient is at least 2^-13 away from the next integer, the rounding error is below 2^-20), so magic >= ceil(2^32 / m) and
// magic * m - 2^32 <= m: the exactness bound above.
__device__ __forceinline__ uint32_t div_magic(uint32_t m) { return (uint32_t)(4294967296.0 / (double)(m < 2u ? 2u : m)) + 1u; }
__device__ __forceinline__ uint32_t norm100(uint32_t c, uint32_t m, uint32_t magic) {
    const uint32_t n = __umul24(100u, c); // (c < 2^13)
    if (m > 6553u) return n / m; // (uniform; beyond the exactness bound)
    return m == 1u ? n : __umulhi(n, magic);
}
__device__ __forceinline__ uint32_t norm100(uint32_t c, uint32_t m) { return norm100(c, m, div_magic(m)); }

// ---- NARROW arithmetic ---------------------------------------------------------------------------------------
// With the narrow mirrors every operand is a non-negative integer below 2^30 (cpu in milli-cores, memory in units
// of 2^mem_shift bytes).  Both scores are functions of RATIOS, which do not depend on the unit, so they are evaluated
// directly on the 32-bit values, with f32 estimates made exact the same way as above:
//   * LeastAllocated: floor((A - x) * 100 / A) from an f32 estimate (relative error < 2^-20, i.e. < 1e-4 absolute on a
//     quotient <= 100) + an exact 64-bit remainder fix-up (the estimate is off by at most one);
//   * BalancedAllocation: the f32 value of (1 - |f0 - f1| / 2) * 100 is within 5e-5 of the reference's fp64 value;
//     unless it lies within 3e-4 of an integer the truncation is the same, otherwise the IEEE fp64 sequence is
//     evaluated (scaling both operands of a division by 2^k does not change its correctly rounded quotient).
struct NarrowPod {
    int32_t req0, req1, nz0, nz1; // the pod's requests / non-zero requests in narrow units
};
__device__ __forceinline__ NarrowPod narrow_pod(const DevPod &p, int sh) {
    return NarrowPod{(int32_t)p.req[0], (int32_t)(p.req[1] >> sh), (int32_t)p.nz_mcpu, (int32_t)(p.nz_mem >> sh)};
}

// floor(d * 100 / A) for 0 <= d <= A < 2^30, A > 0
__device__ __forceinline__ uint32_t floor_ratio100(uint32_t d, uint32_t A) {
    uint32_t q = (uint32_t)((float)d * (100.0f * __builtin_amdgcn_rcpf((float)A)));
    // the estimate is off by at most one, so the remainder 100 d - q A lies in [-A, 2A): below 2^31 in magnitude, hence exact
    // in wrapping 32-bit arithmetic (no 64-bit products).  (Skipping the fix-up for estimates far from an integer was tried:
    // the extra compare-and-branch cost more than the two multiplies it saves -- persistent run 1.33 -> 1.38 ms.)
    const int32_t r = (int32_t)(d * 100u - q * A);
    if (r < 0) q -= 1;
    else if (r >= (int32_t)A) q += 1;
    return q;
}

__device__ __forceinline__ bool fits_narrow(const DevPod &p, const NarrowPod &q, int32_t a0, int32_t a1, int32_t r0, int32_t r1,
                                            int32_t a_pods, int32_t npods) { // fit.go:564-615
    bool ok = (int64_t)npods + 1 <= (int64_t)a_pods;
    if (!p.all_zero_req) {
        if (q.req0 > 0 && q.req0 > a0 - r0) ok = false;
        if (q.req1 > 0 && q.req1 > a1 - r1) ok = false;
    }
    return ok;
}

// Straight-line form: every decision that depends on the NODE is a select, every decision that depends on the pod / profile
// alone is a uniform branch (the first form mixed the two and spent more instructions on exec-mask bookkeeping and a
// per-lane generic division by the weight sum than on the scores: 117 VALU + 119 SALU per evaluation).
__device__ __forceinline__ int64_t dynamic_score_narrow(const DevPod &p, const NarrowPod &q, int32_t a0, int32_t a1, int32_t r0,
                                                        int32_t r1, int32_t z0, int32_t z1) {
    uint32_t total = 0;
    const bool has0 = a0 != 0, has1 = a1 != 0; // resource_allocation.go:66-69: an absent resource does not take part
    if (p.w_fit) { // least_allocated.go:30-61 on NonZeroRequested + the pod's non-zero request
        uint32_t s0 = 0, s1 = 0;
        if (p.fit_cpu) {
            const int32_t x = z0 + q.nz0;
            const uint32_t v = floor_ratio100((uint32_t)(a0 - x), (uint32_t)a0); // (garbage when x > a0 or a0 == 0: selected away)
            s0 = has0 && x <= a0 ? v : 0u;
        }
        if (p.fit_mem) {
            const int32_t x = z1 + q.nz1;
            const uint32_t v = floor_ratio100((uint32_t)(a1 - x), (uint32_t)a1);
            s1 = has1 && x <= a1 ? v : 0u;
        }
        const uint32_t w0 = p.fit_cpu ? (uint32_t)p.fit_w_cpu : 0u, w1 = p.fit_mem ? (uint32_t)p.fit_w_mem : 0u, W = w0 + w1; // uniform
        // sum(s_i w_i) / sum(w_i) over the resources the node has: both -> the uniform weight sum W (weights <= 100 each: the
        // numerator stays below 2^15, so mulhi with ceil(2^32 / W) is the exact quotient); one -> that resource's score; none -> 0
        const uint32_t num = __umul24(s0, w0) + __umul24(s1, w1); // (24-bit multiplies issue at full rate; scores <= 100, weights <= 100)
        const uint32_t both = W == 2u ? num >> 1 : (W <= 1u ? num : __umulhi(num, div_magic(W)));
        const bool c = has0 && w0 != 0, m = has1 && w1 != 0;
        const uint32_t s = c && m ? both : (c ? s0 : (m ? s1 : 0u));
        total += __umul24(s, (uint32_t)p.w_fit); // (plugin weights are validated <= 10^6 < 2^24 by ccsim_set_profile)
    }
    if (p.w_bal) { // balanced_allocation.go:146-180 on Requested + the pod's raw request
        uint32_t score = 100; // fewer than two fractions: std = 0
        if (p.bal_cpu && p.bal_mem) {
            const int32_t x0 = r0 + q.req0, x1 = r1 + q.req1;
            float f0 = (float)x0 * __builtin_amdgcn_rcpf((float)a0), f1 = (float)x1 * __builtin_amdgcn_rcpf((float)a1);
            f0 = f0 > 1.0f ? 1.0f : f0;
            f1 = f1 > 1.0f ? 1.0f : f1;
            const float y = (1.0f - fabsf((f0 - f1) * 0.5f)) * 100.0f; // in [50, 100]
            const float t = floorf(y), fr = y - t;
            uint32_t two = (uint32_t)t;
            if (has0 && has1 && !(fr > 3e-4f && fr < 1.0f - 3e-4f)) two = (uint32_t)balanced_exact(x0, a0, x1, a1); // (rare: a skipped branch)
            score = has0 && has1 ? two : 100u;
        }
        total += __umul24(score, (uint32_t)p.w_bal);
    }
    return (int64_t)total;
}

// ---- run-downs without looking at every placement ---------------------------------------------------------------------------------
// A run-down (ccsim_level.h) puts clones on ONE node while the node stays feasible and its TotalScore stays >= Lo.  Large nodes
// lose a score point only every few pods, and a batch of 64 levels lets them run for up to their whole pod capacity: evaluating
// every intermediate state was most of the persistent kernel's planning phase.  This function returns a number of placements k
// such that EVERY state 1 .. k is feasible and scores >= Lo -- so the run-down may start at state k -- from the real-valued form
// of the score:
//   * over the states in which nothing is clamped (cap below: the Fit filter still passes, NonZeroRequested has not passed the
//     allocatable) LeastAllocated's per-resource value is linear in the number of clones, its weighted mean is linear, and
//     BalancedAllocation's 100 - 50 |f0 - f1| is concave (the absolute value of a linear function, negated): the real-valued
//     total S~(j) is CONCAVE in j.  A concave function that is >= T at 0 and at k is >= T everywhere in between;
//   * the integer score differs from it by the floors only: each resource's floor and the floor of the weighted mean cost
//     LeastAllocated less than 2 points, the truncation costs BalancedAllocation less than 1: S(j) > S~(j) - 2 w_fit - w_bal.
// So with T = Lo + 2 w_fit + w_bal + 1/2 (the half point swallows the fp64 rounding of S~ itself: the operands are below 2^31),
// S~(0) >= T and S~(k) >= T imply S(j) >= Lo for all 0 <= j <= k.  k is found by bisection (S~(j) >= T is monotone on [0, cap]
// once S~(0) >= T).  Nothing about exactness rests on k being the LARGEST such value: the run-down continues from state k with
// the exact integer arithmetic and stops where the reference would.  tests/test_device_arith.py checks the claim itself --
// every skipped state feasible and >= Lo under the exact functions -- on the host.
struct RunDownCoef {
    double lin0, lin1, cb, wb, u0, v0, u1, v1;
};
__device__ __forceinline__ double rd_total(const RunDownCoef &k, int64_t j) { // S~(j), see run_down_safe_skip
    const double x = (double)(j + 1);
    double f0 = k.u0 + k.v0 * x, f1 = k.u1 + k.v1 * x;
    f0 = f0 > 1 ? 1 : f0, f1 = f1 > 1 ? 1 : f1; // (only a resource the pod does not request can sit above 1: a constant)
    return k.lin0 - k.lin1 * x + k.cb - k.wb * 50.0 * fabs(f0 - f1);
}
__device__ __forceinline__ int32_t run_down_safe_skip(const DevPod &p, const NarrowPod &q, int32_t a0, int32_t a1, int32_t r0, int32_t r1, int32_t z0,
                                                      int32_t z1, int32_t a_pods, int32_t npods, int32_t stat, int32_t Lo) {
    // cap: the states 1 .. cap can all take one more pod (fits_narrow) and clamp nothing
    // (32-bit UNSIGNED divisions: every operand is a narrow value below 2^30, and a 64-bit division is a ~120-instruction routine on
    // this machine -- four of them per node were most of the persistent kernel's planning phase: 10.9 -> ~5 us per batch, round 4.
    // A negative numerator -- the node does not even fit once -- ends up at cap < 1 either way.)
    int64_t cap = (int64_t)a_pods - npods - 1;
    if (!p.all_zero_req) {
        if (q.req0 > 0) { const int64_t c = (int64_t)((uint32_t)(a0 > r0 ? a0 - r0 : 0) / (uint32_t)q.req0) - 1; cap = c < cap ? c : cap; }
        if (q.req1 > 0) { const int64_t c = (int64_t)((uint32_t)(a1 > r1 ? a1 - r1 : 0) / (uint32_t)q.req1) - 1; cap = c < cap ? c : cap; }
    }
    const bool has0 = a0 != 0, has1 = a1 != 0;
    // LeastAllocated: a resource whose NonZeroRequested (+ the pod) already exceeds the allocatable scores 0 from here on: a constant
    const bool l0 = p.w_fit && p.fit_cpu && has0 && (int64_t)z0 + q.nz0 <= a0, l1 = p.w_fit && p.fit_mem && has1 && (int64_t)z1 + q.nz1 <= a1;
    if (l0 && q.nz0 > 0) { const int64_t c = (int64_t)((uint32_t)(a0 > z0 ? a0 - z0 : 0) / (uint32_t)q.nz0) - 1; cap = c < cap ? c : cap; }
    if (l1 && q.nz1 > 0) { const int64_t c = (int64_t)((uint32_t)(a1 > z1 ? a1 - z1 : 0) / (uint32_t)q.nz1) - 1; cap = c < cap ? c : cap; }
    if (cap < 1) return 0;
    const double w0 = p.w_fit && p.fit_cpu && has0 ? (double)p.fit_w_cpu : 0.0, w1 = p.w_fit && p.fit_mem && has1 ? (double)p.fit_w_mem : 0.0;
    const double W = w0 + w1 > 0 ? w0 + w1 : 1.0;
    const double k0 = l0 ? 100.0 / (double)a0 : 0.0, k1 = l1 ? 100.0 / (double)a1 : 0.0;
    const bool bal = p.w_bal && p.bal_cpu && p.bal_mem && has0 && has1;
    const double i0 = bal ? 1.0 / (double)a0 : 0.0, i1 = bal ? 1.0 / (double)a1 : 0.0;
    const double T = (double)Lo + 2.0 * (double)p.w_fit + (double)p.w_bal + 0.5;
    // the real-valued total of the node after j clones, as plain coefficients (no closure: it would be a scratch frame):
    //   S~(j) = lin0 - lin1 (j + 1) + w_bal (100 - 50 |min(1, u0 + v0 (j + 1)) - min(1, u1 + v1 (j + 1))|)
    const double fw = (double)p.w_fit / W;
    const double lin0 = (double)stat + fw * (((double)a0 - (double)z0) * k0 * w0 + ((double)a1 - (double)z1) * k1 * w1);
    const double lin1 = fw * ((double)q.nz0 * k0 * w0 + (double)q.nz1 * k1 * w1);
    const double u0 = (double)r0 * i0, v0 = (double)q.req0 * i0, u1 = (double)r1 * i1, v1 = (double)q.req1 * i1, wb = bal ? (double)p.w_bal : 0.0;
    const double cb = (double)p.w_bal * 100.0;
    const RunDownCoef k{lin0, lin1, cb, wb, u0, v0, u1, v1};
    if (!(rd_total(k, 0) >= T)) return 0;
    int64_t lo = 0, hi = cap;
    if (rd_total(k, hi) >= T) return (int32_t)hi;
    while (hi - lo > 1) {
        const int64_t mid = (lo + hi) >> 1;
        if (rd_total(k, mid) >= T) lo = mid; else hi = mid;
    }
    return (int32_t)lo;
}

// NodeResourcesFit filter for the cpu/mem/pods part (fit.go:564-615); extras are checked by the caller.
__device__ __forceinline__ bool fits_core(const DevPod &p, int64_t a_cpu, int64_t a_mem, int64_t r_cpu, int64_t r_mem,
                                          int32_t a_pods, int32_t npods) {
    if (!p.fit_enabled) return true;
    bool ok = (int64_t)npods + 1 <= (int64_t)a_pods;
    if (!p.all_zero_req) {
        if (p.req[0] > 0 && p.req[0] > a_cpu - r_cpu) ok = false;
        if (p.req[1] > 0 && p.req[1] > a_mem - r_mem) ok = false;
    }
    return ok;
}

// static (pod-spec dependent, state independent) part of the total:
//   TaintToleration: DefaultNormalizeScore(100, reverse) ; NodeAffinity: DefaultNormalizeScore(100)
//   ImageLocality: the node's score as it is (image_locality.go:54-66: no NormalizeScore)
__device__ __forceinline__ int64_t static_score(const DevPod &p, uint32_t c, uint32_t a, uint32_t img, uint32_t mt, uint32_t ma, uint32_t magic_t,
                                                uint32_t magic_a) {
    uint32_t t = __umul24(img, (uint32_t)p.w_img); // (scores <= 100, weights <= 10^6: 24-bit multiplies, full rate)
    if (p.w_taint) t += __umul24(mt == 0 ? 100u : 100u - norm100(c, mt, magic_t), (uint32_t)p.w_taint);
    if (p.w_aff) t += __umul24(ma == 0 ? 0u : norm100(a, ma, magic_a), (uint32_t)p.w_aff); // w_aff is 0 when PreScore skips
    return (int64_t)t;
}
__device__ __forceinline__ int64_t static_score(const DevPod &p, uint32_t c, uint32_t a, uint32_t img, uint32_t mt, uint32_t ma) {
    return static_score(p, c, a, img, mt, ma, div_magic(mt), div_magic(ma));
}

// the int64 columns from the lossless 32-bit mirrors (cpu as it is, memory << the common power-of-two unit): what a persistent
// launch that skipped their write-back (PersistCols::skip_wide) left to be done
__global__ __launch_bounds__(256) void k_widen(const int32_t *r0, const int32_t *r1, const int32_t *z0, const int32_t *z1, int64_t *req0, int64_t *req1,
                                               int64_t *nz_mcpu, int64_t *nz_mem, int sh, int64_t n_pad) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_pad) return;
    req0[i] = (int64_t)r0[i], req1[i] = (int64_t)r1[i] << sh, nz_mcpu[i] = (int64_t)z0[i], nz_mem[i] = (int64_t)z1[i] << sh;
}

__device__ __forceinline__ uint64_t make_key(int64_t total, int64_t gidx) {
    return ((uint64_t)(total + 1) << kIdxBits) | (kIdxMask - (uint64_t)gidx);
}
__device__ __forceinline__ int64_t key_index(uint64_t key) { return (int64_t)(kIdxMask - (key & kIdxMask)); }
__device__ __forceinline__ int64_t key_score(uint64_t key) { return (int64_t)(key >> kIdxBits) - 1; }

// ------------------------------------------------------------------------------------------------
// k_scan: one full pods x nodes pass for the current pod spec: Filter (static bit + Fit), Score
// (TaintToleration, NodeAffinity, LeastAllocated, BalancedAllocation), weighted sum, per-block argmax.
// ------------------------------------------------------------------------------------------------
// Go's math.Log (pure-Go path, go1.24 src/math/log.go = FreeBSD e_log.c): plain IEEE +,-,*,/ and frexp, so the
// device result is bit-identical to the host oracle's (no fast-math, no FMA contraction).  scoring.go:294-296.
__device__ __forceinline__ double go_log(double x) {
    const double Ln2Hi = 6.93147180369123816490e-01, Ln2Lo = 1.90821492927058770002e-10;
    const double L1 = 6.666666666666735130e-01, L2 = 3.999999999940941908e-01, L3 = 2.857142874366239149e-01,
                 L4 = 2.222219843214978396e-01, L5 = 1.818357216161805012e-01, L6 = 1.531383769920937332e-01,
                 L7 = 1.479819860511658591e-01;
    int ki;
    double f1 = frexp(x, &ki);
    if (f1 < 0.70710678118654752440) {
        f1 *= 2;
        ki--;
    }
    const double f = f1 - 1;
    const double k = (double)ki;
    const double s = f / (2 + f);
    const double s2 = s * s;
    const double s4 = s2 * s2;
    const double t1 = s2 * (L1 + s4 * (L3 + s4 * (L5 + s4 * L7)));
    const double t2 = s4 * (L2 + s4 * (L4 + s4 * L6));
    const double R = t1 + t2;
    const double hfsq = 0.5 * f * f;
    return k * Ln2Hi - ((hfsq - (s * (hfsq + R) + k * Ln2Lo)) - f);
}

// PodTopologySpread.Score (scoring.go:196-223) for a node that has all soft keys; also stamps the node's domains
// as candidates of this scan.
// `soft_w` points INTO the device-resident DevState (not a per-lane copy: indexing a local copy with the runtime
// constraint number puts the whole 400-byte struct into scratch, for every lane of the scan)
__device__ __forceinline__ int64_t soft_raw_score(const DevSoft &p, const double *soft_w, const int32_t *pod_count, int64_t i,
                                                 int32_t epoch) {
    double score = 0;
    for (int c = 0; c < p.n; c++) {
        const int32_t v = p.label[c][i];
        int64_t cnt;
        if (p.is_hostname[c])
            cnt = (int64_t)(p.existing[c] ? p.existing[c][i] : 0) + (p.self_match[c] ? pod_count[i] - p.pod_count0[i] : 0);
        else {
            cnt = p.tbl[c][v];
            p.flag[c][v] = epoch;
            if (p.nocredit[c] && v == p.nocredit[c]) continue; // the key is missing on this node (system default constraints): counted above, no credit (scoring.go:210)
        }
        score += (double)cnt * soft_w[c] + (double)(p.max_skew[c] - 1);
    }
    return (int64_t)round(score); // math.Round: half away from zero
}

// NormalizeScore (scoring.go:226-265)
__device__ __forceinline__ int64_t soft_normalize(int64_t raw, int64_t mn, int64_t mx) {
    if (mx == 0) return 100;
    return 100 * (mx + mn - raw) / mx;
}

// InterPodAffinity.Filter (filtering.go:410-432): 0 ok, 1 affinity (Unresolvable), 2 anti-affinity, 3 existing pods' anti-affinity
__device__ __forceinline__ int ipa_filter(const DevIpa &p, const DevState &st, int64_t i) {
    if (st.ipa_exist_total == 0 && p.n_aff == 0 && p.n_anti == 0) return 0; // PreFilter Skip (filtering.go:299-301)
    bool pods_exist = true;
    for (int t = 0; t < p.n_aff; t++) { // satisfyPodAffinity :382-408
        const int k = p.aff_key[t];
        const int32_t v = p.label[k][i];
        if (!v) return 1;
        if (p.aff[k][v] <= 0) pods_exist = false;
    }
    if (!pods_exist && !(st.ipa_aff_total == 0 && p.self_aff)) return 1;
    for (int t = 0; t < p.n_anti; t++) { // satisfyPodAntiAffinity :367-379
        const int k = p.anti_key[t];
        const int32_t v = p.label[k][i];
        if (v && p.anti[k][v] > 0) return 2;
    }
    if (st.ipa_exist_total > 0) // satisfyExistingPodsAntiAffinity :352-364
        for (int k = 0; k < p.n_keys; k++) {
            const int32_t v = p.label[k][i];
            if (v && p.exist[k][v] > 0) return 3;
        }
    return 0;
}

// InterPodAffinity.Score (scoring.go:226-247): sum of the node's topology pairs
__device__ __forceinline__ int64_t ipa_raw_score(const DevIpa &p, int64_t i) {
    int64_t s = 0;
    for (int k = 0; k < p.n_keys; k++) {
        const int32_t v = p.label[k][i];
        if (v) s += p.score[k][v];
    }
    return s;
}

// NormalizeScore (scoring.go:258-290): fp64, truncated
__device__ __forceinline__ int64_t ipa_normalize(int64_t raw, int64_t mn, int64_t mx) {
    const int64_t diff = mx - mn;
    double f = 0;
    if (diff > 0) f = 100.0 * ((double)(raw - mn) / (double)diff);
    return (int64_t)f;
}

// One argument block for the scan, the one-block final reduction / decision and the distributed decide.
struct ScanArgs {
    DevCols c;
    DevPod p;
    DevState *st;
    uint64_t *partials; // [grid][2]: packed max key ; norm | nfeas << 32   (struct Partial)
    int64_t chunk;      // nodes per block (multiple of kTile)
    DevPts pts;
    uint64_t *pts_min_partials; // [grid][kMaxTsc]: per-block minimum match count per constraint
    DevIpa ipa;
    uint64_t *ipa_partials;     // [grid][2]: per-block min / max raw InterPodAffinity score over feasible nodes
    DevSoft soft;
    uint64_t *soft_partials;    // [grid][3]: feasible non-ignored nodes, min / max raw PodTopologySpread score
    int32_t n_partials;         // = scan grid (a fused last-block reduction inside k_scan was measured SLOWER:
                                // inlined or called, it costs the scan its registers / occupancy -- 44k -> 36k cycles/s)
    XRec *xsend;                // distributed: this shard's record out
    const XRec *xrecv;          // distributed: gathered records in
    int32_t n_ranks;            // 0 = single GPU (decide from the local record)
    int32_t *log;
    uint64_t *smp_partials;     // [grid][2] sampled search: feasible nodes of the block ; those with index < smp_start
    int64_t *smp_prefix;        // [grid] feasible nodes in the blocks before this one (index order)
};

template <class A> __device__ void final_body(const A &a);

// PTS = the pod carries topology-coupled plugins (PodTopologySpread and / or InterPodAffinity)
// SMP = the sampled search (percentageOfNodesToScore < 100; findNodesThatPassFilters, schedule_one.go:610-680):
//   0  every node is scored (the search visits all N nodes);
//   1  counting pass: feasible nodes per block (and how many of them lie before nextStartNodeIndex);
//   2  scoring pass: a feasible node is kept iff fewer than K feasible nodes precede it in the visiting order
//      (start, start+1, ..., N-1, 0, ..., start-1).  Its rank in that order follows from the exclusive prefix F(i)
//      of the feasibility bits in INDEX order: rank = F(i) - F(start) for i >= start, F(i) + F_total - F(start)
//      otherwise -- block prefix (k_smp_prefix) + wave ballots, no sort.  The selectHost tie-break follows the
//      feasible-list order, so the key carries the visiting position instead of the node index.
template <int NX, bool PTS, bool NARROW = false, int SMP = 0>
__global__ __launch_bounds__(kThreads) void k_scan(ScanArgs a) {
    const DevState st = *a.st;
    if (st.done) return;
    if (SMP != 0 && a.n_ranks > 0 && st.smp_phase != (SMP == 1 ? 0 : 1)) return; // sharded: this pass belongs to the other phase
    const int64_t smp_S = st.smp_start, smp_N = st.smp_N;
    int64_t smp_carry = SMP == 2 ? a.smp_prefix[blockIdx.x] + st.smp_off : 0; // feasible nodes before the current tile (index order, all shards)
    uint32_t smp_cnt = 0, smp_before = 0;
    int smp_par = 0;
    __shared__ uint32_t s_smp[2][kThreads / 64];
    const uint32_t mt = (uint32_t)st.mt_a, ma = (uint32_t)st.ma_a;
    const int tid = threadIdx.x;
    const int64_t lo = (int64_t)blockIdx.x * a.chunk;
    int64_t hi = lo + a.chunk;
    if (hi > a.c.n_pad) hi = a.c.n_pad;

    uint64_t best = 0;
    uint32_t mt_b = 0, ma_b = 0, nfeas = 0;
    int32_t pmin[kMaxTsc];
#pragma unroll
    for (int c = 0; c < kMaxTsc; c++) pmin[c] = 0x7fffffff;
    int64_t soft_mn = INT64_MAX, soft_mx = 0, soft_cnt = 0; // maxScore starts at 0 (scoring.go:238)
    const bool soft_scoring = PTS && a.soft.n > 0 && a.soft.w;
    const int32_t epoch = (int32_t)(st.scans + 1);
    int64_t ipa_mn = INT64_MAX, ipa_mx = INT64_MIN;
    const bool ipa_scoring = PTS && a.ipa.on && a.ipa.w && st.ipa_entries > 0; // else PreScore Skip (scoring.go:199-201)
    const NarrowPod npod = narrow_pod(a.p, a.c.mem_shift);

    for (int64_t base = lo; base < hi; base += kTile) {
        const int64_t i0 = base + 2 * tid; // first of this thread's 2 nodes
        const uint2 sw = *reinterpret_cast<const uint2 *>(a.c.stat + i0);
        // NARROW: the 32-bit mirrors are used as they are (no widening); otherwise the int64 columns
        int2 a0n{}, a1n{}, r0n{}, r1n{}, z0n{}, z1n{};
        longlong2 A0{}, A1{}, R0{}, R1{}, Z0{}, Z1{};
        if (NARROW) {
            a0n = *reinterpret_cast<const int2 *>(a.c.a32[0] + i0), a1n = *reinterpret_cast<const int2 *>(a.c.a32[1] + i0);
            r0n = *reinterpret_cast<const int2 *>(a.c.r32[0] + i0), r1n = *reinterpret_cast<const int2 *>(a.c.r32[1] + i0);
            z0n = *reinterpret_cast<const int2 *>(a.c.z32[0] + i0), z1n = *reinterpret_cast<const int2 *>(a.c.z32[1] + i0);
        } else {
            const Cols6 c6 = load_cols6<false>(a.c, i0);
            A0 = c6.A0, A1 = c6.A1, R0 = c6.R0, R1 = c6.R1, Z0 = c6.Z0, Z1 = c6.Z1;
        }
        const int2 AP = *reinterpret_cast<const int2 *>(a.c.alloc_pods + i0);
        const int2 NP = *reinterpret_cast<const int2 *>(a.c.pod_count + i0);
        bool xok0 = true, xok1 = true;
        int64_t xa0[NX > 0 ? NX : 1], xr0[NX > 0 ? NX : 1], xa1[NX > 0 ? NX : 1], xr1[NX > 0 ? NX : 1]; // the pair's extra columns
        if (NX > 0) {
#pragma unroll
            for (int x = 0; x < NX; x++) {
                xa0[x] = xr0[x] = xa1[x] = xr1[x] = 0;
                if (x < a.p.nx) {
                    const int col = a.p.xcol[x];
                    const longlong2 XA = *reinterpret_cast<const longlong2 *>(a.c.alloc[col] + i0);
                    const longlong2 XR = *reinterpret_cast<const longlong2 *>(a.c.req[col] + i0);
                    xa0[x] = XA.x, xr0[x] = XR.x, xa1[x] = XA.y, xr1[x] = XR.y;
                    const int64_t rq = a.p.req[col];
                    if (a.p.fit_enabled && !a.p.all_zero_req && rq > 0) { // (a slot held only for scoring carries no request: fit.go:585-615 checks requested resources)
                        if (rq > XA.x - XR.x) xok0 = false;
                        if (rq > XA.y - XR.y) xok1 = false;
                    }
                }
            }
        }
        bool fe[2];
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const uint32_t w = k ? sw.y : sw.x;
            const int64_t a_cpu = k ? A0.y : A0.x, a_mem = k ? A1.y : A1.x;
            const int64_t r_cpu = k ? R0.y : R0.x, r_mem = k ? R1.y : R1.x;
            const int32_t a_pods = k ? AP.y : AP.x, npods = k ? NP.y : NP.x;
            const int32_t na0 = k ? a0n.y : a0n.x, na1 = k ? a1n.y : a1n.x, nr0 = k ? r0n.y : r0n.x, nr1 = k ? r1n.y : r1n.x;
            bool feasible = (w >> kStatOkBit) && (k ? xok1 : xok0) &&
                            (NARROW ? fits_narrow(a.p, npod, na0, na1, nr0, nr1, a_pods, npods)
                                    : fits_core(a.p, a_cpu, a_mem, r_cpu, r_mem, a_pods, npods));
            if (PTS) { // PodTopologySpread.Filter (filtering.go:311-356) + the minimum for the next verification
                const uint32_t eb = a.pts.elig[i0 + k];
#pragma unroll
                for (int c = 0; c < kMaxTsc; c++) {
                    if (c >= a.pts.n) break;
                    const int32_t v = a.pts.label[c][i0 + k];
                    const int32_t m = v ? a.pts.tbl[c][v] : 0;
                    if (v && (eb & 1u) && ((eb >> (1 + c)) & 1u)) pmin[c] = m < pmin[c] ? m : pmin[c];
                    if (!v) feasible = false; // missing required label
                    else {
                        const int64_t minm = a.pts.n_present[c] < a.pts.min_domains[c] ? 0 : (int64_t)st.pts_min_a[c];
                        if ((int64_t)m + a.pts.self_match[c] - minm > (int64_t)a.pts.max_skew[c]) feasible = false;
                    }
                }
                if (feasible && a.ipa.on && a.ipa.filter_on && ipa_filter(a.ipa, st, i0 + k)) feasible = false;
            }
            fe[k] = feasible;
        }
        if (SMP == 1) { // counting pass: no scores
            smp_cnt += (uint32_t)fe[0] + (uint32_t)fe[1];
            smp_before += (uint32_t)(fe[0] && a.c.global_offset + i0 < smp_S) + (uint32_t)(fe[1] && a.c.global_offset + i0 + 1 < smp_S);
            continue;
        }
        int64_t vpos[2] = {a.c.global_offset + i0, a.c.global_offset + i0 + 1}; // tie-break position of the node
        if (SMP == 2) {
            const uint64_t b0 = __ballot(fe[0]), b1 = __ballot(fe[1]);
            const int lane_ = tid & 63, wave_ = tid >> 6;
            const uint64_t lt = (1ull << lane_) - 1;
            if (lane_ == 0) s_smp[smp_par][wave_] = (uint32_t)(__popcll(b0) + __popcll(b1));
            __syncthreads(); // one barrier per tile: the two buffers alternate
            uint32_t woff = 0, ttot = 0;
#pragma unroll
            for (int w_ = 0; w_ < kThreads / 64; w_++) {
                const uint32_t c_ = s_smp[smp_par][w_];
                woff += w_ < wave_ ? c_ : 0;
                ttot += c_;
            }
            const int64_t F0 = smp_carry + woff + __popcll(b0 & lt) + __popcll(b1 & lt);
            smp_carry += ttot;
            smp_par ^= 1;
            const int64_t f0 = fe[0];
#pragma unroll
            for (int k = 0; k < 2; k++) {
                const int64_t gi = a.c.global_offset + i0 + k, F = F0 + (k ? f0 : 0);
                const int64_t rank = gi >= smp_S ? F - st.smp_Fs : F + st.smp_Ftotal - st.smp_Fs;
                vpos[k] = gi >= smp_S ? gi - smp_S : gi + smp_N - smp_S;
                if (fe[k] && rank == st.smp_K) a.st->smp_stop = vpos[k]; // the node that cancels the search (:655-662)
                fe[k] = fe[k] && rank < st.smp_K;
            }
        }
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const uint32_t w = k ? sw.y : sw.x;
            const int64_t a_cpu = k ? A0.y : A0.x, a_mem = k ? A1.y : A1.x;
            const int64_t r_cpu = k ? R0.y : R0.x, r_mem = k ? R1.y : R1.x;
            const int64_t z_cpu = k ? Z0.y : Z0.x, z_mem = k ? Z1.y : Z1.x;
            const int32_t na0 = k ? a0n.y : a0n.x, na1 = k ? a1n.y : a1n.x, nr0 = k ? r0n.y : r0n.x, nr1 = k ? r1n.y : r1n.x;
            const int32_t nz0 = k ? z0n.y : z0n.x, nz1 = k ? z1n.y : z1n.x;
            const bool feasible = fe[k];
            int64_t xak[NX > 0 ? NX : 1], xrk[NX > 0 ? NX : 1]; // (element-wise select: a pointer select would put the arrays in scratch)
#pragma unroll
            for (int x = 0; x < (NX > 0 ? NX : 1); x++) xak[x] = k ? xa1[x] : xa0[x], xrk[x] = k ? xr1[x] : xr0[x];
            const uint64_t mask = __ballot(feasible);
            nfeas += (uint32_t)__popcll(mask); // identical in every lane of the wave
            if (feasible) {
                const uint32_t cnt = (w >> kStatCntShift) & kStatCntMask, aff = w & kStatAffMask, img = (w >> kStatImgShift) & kStatImgMask;
                int64_t total = static_score(a.p, cnt, aff, img, mt, ma) +
                                (NARROW ? dynamic_score_narrow(a.p, npod, na0, na1, nr0, nr1, nz0, nz1)
                                        : (NX > 0 && a.p.gen_score
                                               ? dynamic_score_gen<NX>(a.p, a_cpu, a_mem, r_cpu, r_mem, z_cpu, z_mem, xak, xrk)
                                               : dynamic_score(a.p, make_rcp(a_cpu, a_mem), a_cpu, a_mem, r_cpu, r_mem, z_cpu, z_mem)));
                if (soft_scoring) {
                    if (a.soft.elig[i0 + k] & 1u) {
                        const int64_t raw = soft_raw_score(a.soft, a.st->soft_w, a.c.pod_count, i0 + k, epoch);
                        soft_mn = raw < soft_mn ? raw : soft_mn;
                        soft_mx = raw > soft_mx ? raw : soft_mx;
                        soft_cnt++;
                        total += soft_normalize(raw, st.soft_min_a, st.soft_max_a) * a.soft.w;
                    } // ignored nodes score 0 (scoring.go:205-207, 254-257)
                }
                if (ipa_scoring) {
                    const int64_t raw = ipa_raw_score(a.ipa, i0 + k);
                    ipa_mn = raw < ipa_mn ? raw : ipa_mn;
                    ipa_mx = raw > ipa_mx ? raw : ipa_mx;
                    total += ipa_normalize(raw, st.ipa_min_a, st.ipa_max_a) * a.ipa.w;
                }
                const uint64_t key = make_key(total, vpos[k]);
                best = key > best ? key : best;
                mt_b = cnt > mt_b ? cnt : mt_b;
                ma_b = aff > ma_b ? aff : ma_b;
            }
        }
    }

    if (SMP == 1) {
        const int64_t c1 = wave_sum_i64(smp_cnt), c2 = wave_sum_i64(smp_before);
        __shared__ int64_t s_c[2][kThreads / 64];
        if ((tid & 63) == 0) s_c[0][tid >> 6] = c1, s_c[1][tid >> 6] = c2;
        __syncthreads();
        if (tid == 0) {
            int64_t t1 = 0, t2 = 0;
            for (int w = 0; w < kThreads / 64; w++) t1 += s_c[0][w], t2 += s_c[1][w];
            st_agent(a.smp_partials + 2 * (int64_t)blockIdx.x, (uint64_t)t1);
            st_agent(a.smp_partials + 2 * (int64_t)blockIdx.x + 1, (uint64_t)t2);
        }
        return;
    }
    // wave reduce, then LDS across the 4 waves
    best = wave_max_u64(best);
    mt_b = wave_max_u32(mt_b);
    ma_b = wave_max_u32(ma_b);
    __shared__ uint64_t s_key[kThreads / 64];
    __shared__ uint32_t s_mt[kThreads / 64], s_ma[kThreads / 64], s_nf[kThreads / 64];
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63; // (wave: uniform, and known to the compiler as such)
    if (lane == 0) {
        s_key[wave] = best;
        s_mt[wave] = mt_b;
        s_ma[wave] = ma_b;
        s_nf[wave] = nfeas;
    }
    __syncthreads();
    if (tid == 0) {
        Partial out;
        out.key = 0;
        uint32_t m1 = 0, m2 = 0, nf = 0;
#pragma unroll
        for (int w = 0; w < kThreads / 64; w++) {
            out.key = s_key[w] > out.key ? s_key[w] : out.key;
            m1 = s_mt[w] > m1 ? s_mt[w] : m1;
            m2 = s_ma[w] > m2 ? s_ma[w] : m2;
            nf += s_nf[w];
        }
        out.norm = (m1 << kStatCntShift) | m2;
        out.nfeas = nf;
        st_agent(a.partials + 2 * (int64_t)blockIdx.x, out.key);
        st_agent(a.partials + 2 * (int64_t)blockIdx.x + 1, (uint64_t)out.norm | ((uint64_t)out.nfeas << 32));
    }
    if (PTS) {
        __shared__ int32_t s_pm[kThreads / 64][kMaxTsc];
#pragma unroll
        for (int c = 0; c < kMaxTsc; c++) {
            int32_t v = pmin[c];
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) {
                const int32_t o = __shfl_xor(v, off, 64);
                v = o < v ? o : v;
            }
            if (lane == 0) s_pm[wave][c] = v;
        }
        __syncthreads();
        if (tid < kMaxTsc && a.pts.n) {
            int32_t v = s_pm[0][tid];
            for (int w = 1; w < kThreads / 64; w++) v = s_pm[w][tid] < v ? s_pm[w][tid] : v;
            st_agent(a.pts_min_partials + (int64_t)blockIdx.x * kMaxTsc + tid, (uint64_t)(int64_t)v);
        }
        if (a.soft.n) {
            __shared__ int64_t s_sm[3][kThreads / 64];
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) {
                const int64_t o1 = __shfl_xor(soft_mn, off, 64), o2 = __shfl_xor(soft_mx, off, 64);
                soft_mn = o1 < soft_mn ? o1 : soft_mn;
                soft_mx = o2 > soft_mx ? o2 : soft_mx;
                soft_cnt += __shfl_xor(soft_cnt, off, 64);
            }
            if (lane == 0) s_sm[0][wave] = soft_cnt, s_sm[1][wave] = soft_mn, s_sm[2][wave] = soft_mx;
            __syncthreads();
            if (tid == 0) {
                int64_t cn = 0, mn = INT64_MAX, mx = 0;
                for (int w = 0; w < kThreads / 64; w++) {
                    cn += s_sm[0][w];
                    mn = s_sm[1][w] < mn ? s_sm[1][w] : mn;
                    mx = s_sm[2][w] > mx ? s_sm[2][w] : mx;
                }
                st_agent(a.soft_partials + 3 * (int64_t)blockIdx.x, (uint64_t)cn);
                st_agent(a.soft_partials + 3 * (int64_t)blockIdx.x + 1, (uint64_t)mn);
                st_agent(a.soft_partials + 3 * (int64_t)blockIdx.x + 2, (uint64_t)mx);
            }
        }
        if (a.ipa.on) {
            __shared__ int64_t s_im[2][kThreads / 64];
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) {
                const int64_t o1 = __shfl_xor(ipa_mn, off, 64), o2 = __shfl_xor(ipa_mx, off, 64);
                ipa_mn = o1 < ipa_mn ? o1 : ipa_mn;
                ipa_mx = o2 > ipa_mx ? o2 : ipa_mx;
            }
            if (lane == 0) s_im[0][wave] = ipa_mn, s_im[1][wave] = ipa_mx;
            __syncthreads();
            if (tid == 0) {
                int64_t mn = s_im[0][0], mx = s_im[1][0];
                for (int w = 1; w < kThreads / 64; w++) mn = s_im[0][w] < mn ? s_im[0][w] : mn, mx = s_im[1][w] > mx ? s_im[1][w] : mx;
                st_agent(a.ipa_partials + 2 * (int64_t)blockIdx.x, (uint64_t)mn);
                st_agent(a.ipa_partials + 2 * (int64_t)blockIdx.x + 1, (uint64_t)mx);
            }
        }
    }
}

// k_smp_prefix: one block, between the counting and the scoring pass of a sampled cycle.  Exclusive prefix of the
// per-block feasible counts (index order), the snapshot total and F(start).
__global__ __launch_bounds__(kThreads) void k_smp_prefix(ScanArgs a) {
    if (a.st->done) return;
    if (a.n_ranks > 0 && a.st->smp_phase != 0) return; // sharded: a scoring pass
    const int tid = threadIdx.x, n = a.n_partials;
    const int per = (n + kThreads - 1) / kThreads;
    const int lo = tid * per < n ? tid * per : n, hi = lo + per < n ? lo + per : n;
    int64_t s = 0, sb = 0;
    for (int i = lo; i < hi; i++) s += (int64_t)ld_agent(a.smp_partials + 2 * (int64_t)i), sb += (int64_t)ld_agent(a.smp_partials + 2 * (int64_t)i + 1);
    int64_t inc = s; // inclusive scan over the wave
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int64_t o = __shfl_up(inc, off, 64);
        if ((tid & 63) >= off) inc += o;
    }
    sb = wave_sum_i64(sb);
    __shared__ int64_t s_w[kThreads / 64], s_b[kThreads / 64];
    if ((tid & 63) == 63) s_w[tid >> 6] = inc;
    if ((tid & 63) == 0) s_b[tid >> 6] = sb;
    __syncthreads();
    int64_t woff = 0, total = 0, before = 0;
    for (int w = 0; w < kThreads / 64; w++) {
        woff += w < (tid >> 6) ? s_w[w] : 0;
        total += s_w[w];
        before += s_b[w];
    }
    int64_t run = woff + inc - s;
    for (int i = lo; i < hi; i++) {
        a.smp_prefix[i] = run;
        run += (int64_t)ld_agent(a.smp_partials + 2 * (int64_t)i);
    }
    if (tid == 0) {
        a.st->smp_Ftotal = total;
        a.st->smp_Fs = before;
        a.st->smp_stop = -1;
    }
}

// ------------------------------------------------------------------------------------------------
// decide + commit (one thread): the sequential part of a scheduling cycle.
//   schedule_one.go:448-463 (0 feasible -> FitError), selectHost :894-941, assume :967-984,
//   simulator.go:297-312 (limit test after the append).
// The pending scan was computed with (mt_a, ma_a); if the true maxima over the feasible set differ,
// the scores were normalized with the wrong constants: fix the constants and rescan, commit nothing.
// ------------------------------------------------------------------------------------------------
struct WinnerTopo { // topology value ids of the winning node (from the owning rank's record in the sharded path)
    uint32_t elig;
    int32_t pts_v[kMaxTsc];
    int32_t ipa_v[4];
    uint32_t soft_elig;
    int32_t soft_v[kMaxTsc];
};

struct SoftAgg { // one scan's PodTopologySpread PreScore facts
    int64_t size[kMaxTsc]; // candidate domains per constraint (hostname: feasible non-ignored nodes)
    int64_t mn, mx;
};

template <class A>
__device__ __forceinline__ void decide_commit(const A &a, uint64_t key, uint32_t mt, uint32_t ma, int64_t nfeas,
                                              const int32_t *pts_min, int64_t ipa_mn = 0, int64_t ipa_mx = 0,
                                              const SoftAgg *soft = nullptr, const WinnerTopo *wt = nullptr) {
    DevState st = *a.st;
    st.winner = -1;
    if (st.done) return;
    st.scans += 1;
    bool pts_stale = false; // the scan filtered with an outdated global minimum (filtering.go:298-305): redo it
    for (int c = 0; c < a.pts.n; c++)
        if (pts_min[c] != st.pts_min_a[c]) st.pts_min_a[c] = pts_min[c], pts_stale = true;
    if (pts_stale) {
        *a.st = st;
        return;
    }
    const bool smp = st.smp_K > 0;
    const int64_t N = smp ? st.smp_N : a.c.n; // (the sampled search wraps around at the cluster's node count; `evaluated` is only reported for it)
    // a finished sampled cycle moves nextStartNodeIndex past the nodes it visited (schedule_one.go:538-539)
    const bool smp_all = !smp || st.smp_Ftotal <= st.smp_K; // the search visited every node
    if (key == 0) {
        st.done = DONE_UNSCHEDULABLE;
        st.rounds += 1;
        st.last_feasible = 0;
        st.last_evaluated = (int32_t)N; // no feasible node: every node was visited
        st.evaluated += N;
    } else if ((int32_t)mt != st.mt_a || (int32_t)ma != st.ma_a) {
        st.mt_a = (int32_t)mt;
        st.ma_a = (int32_t)ma;
    } else if (soft && a.soft.w && [&] { // topology weights depend on the candidate-domain counts (scoring.go:96-113)
                   bool stale = false;
                   for (int c = 0; c < a.soft.n; c++)
                       if (soft->size[c] != st.soft_size_a[c]) {
                           st.soft_size_a[c] = soft->size[c];
                           st.soft_w[c] = go_log((double)(soft->size[c] + 2));
                           stale = true;
                       }
                   return stale;
               }()) {
        // rescan with the right weights
    } else if (soft && a.soft.w && (soft->mn != st.soft_min_a || soft->mx != st.soft_max_a)) {
        st.soft_min_a = soft->mn;
        st.soft_max_a = soft->mx;
    } else if (a.ipa.on && a.ipa.w && st.ipa_entries > 0 && (ipa_mn != st.ipa_min_a || ipa_mx != st.ipa_max_a)) {
        st.ipa_min_a = ipa_mn; // InterPodAffinity scores were normalized with stale min / max: rescan
        st.ipa_max_a = ipa_mx;
    } else {
        int64_t g = key_index(key);
        const int64_t visited = smp_all ? N : st.smp_stop;
        if (smp) { // the key carries the visiting position
            g += st.smp_start;
            g = g >= N ? g - N : g;
            st.smp_start += visited; // both terms are <= N
            st.smp_start = st.smp_start >= N ? st.smp_start - N : st.smp_start;
            st.evaluated += visited;
        }
        st.last_evaluated = (int32_t)visited;
        const int64_t i = g - a.c.global_offset;
        if (i >= 0 && i < a.c.n) { // this shard owns the winner: NodeInfo.update (types.go:409-428)
            // one thread, latency-bound: issue every load of the row before the first store
            const int64_t r0 = a.c.req[0][i], r1 = a.c.req[1][i], z0 = a.c.nz_mcpu[i], z1 = a.c.nz_mem[i];
            const int32_t pc = a.c.pod_count[i], pl = a.c.placed_cnt[i];
            a.c.req[0][i] = r0 + a.p.req[0];
            a.c.req[1][i] = r1 + a.p.req[1];
            a.c.nz_mcpu[i] = z0 + a.p.nz_mcpu;
            a.c.nz_mem[i] = z1 + a.p.nz_mem;
            a.c.pod_count[i] = pc + 1;
            a.c.placed_cnt[i] = pl + 1;
            store_mirror(a.c, i, r0 + a.p.req[0], r1 + a.p.req[1], z0 + a.p.nz_mcpu, z1 + a.p.nz_mem);
#pragma unroll 1
            for (int col = 2; col < a.p.ncol; col++)
                if (a.p.req[col] != 0) a.c.req[col][i] += a.p.req[col];
        }
        // Replicated topology tables: EVERY rank applies the winner's contribution (the clone is an existing pod of the
        // next cycle).  Single GPU: the ids are read from the node's own columns; sharded: from the owner's record.
        const bool local = i >= 0 && i < a.c.n;
        if (a.pts.n && (local || wt)) { // filtering.go:255-296 on the next cycle
            const uint32_t eb = wt ? wt->elig : a.pts.elig[i];
            for (int c = 0; c < a.pts.n; c++) {
                const int32_t v = wt ? wt->pts_v[c] : a.pts.label[c][i];
                if (v && (eb & 1u) && ((eb >> (1 + c)) & 1u) && a.pts.self_match[c]) a.pts.tbl[c][v] += 1;
            }
        }
        if (a.soft.n && (local || wt)) { // scoring.go:147-178 on the next cycle (replicated tables: every rank)
            const uint32_t eb = wt ? wt->soft_elig : a.soft.elig[i];
            for (int c = 0; c < a.soft.n; c++) {
                const int32_t v = wt ? wt->soft_v[c] : a.soft.label[c][i];
                if (v && !a.soft.is_hostname[c] && (eb & 1u) && ((eb >> (1 + c)) & 1u) && a.soft.self_match[c]) a.soft.tbl[c][v] += 1;
            }
        }
        if (a.ipa.on && (local || wt)) { // filtering.go:204-272, scoring.go:81-125
            for (int k = 0; k < a.ipa.n_keys; k++) {
                const int32_t v = wt ? wt->ipa_v[k] : a.ipa.label[k][i];
                if (!v) continue;
                if (a.ipa.self_aff && a.ipa.aff_terms_on_key[k]) {
                    a.ipa.aff[k][v] += a.ipa.aff_terms_on_key[k];
                    st.ipa_aff_total += a.ipa.aff_terms_on_key[k];
                }
                if (a.ipa.anti_self_on_key[k]) {
                    a.ipa.anti[k][v] += a.ipa.anti_self_on_key[k];
                    a.ipa.exist[k][v] += a.ipa.anti_self_on_key[k];
                    st.ipa_exist_total += a.ipa.anti_self_on_key[k];
                }
                a.ipa.score[k][v] += a.ipa.score_self[k];
                st.ipa_entries += a.ipa.self_entries[k];
            }
        }
        if (a.log && st.placed < st.log_cap) a.log[st.placed] = (int32_t)g;
        st.placed += 1;
        st.rounds += 1;
        st.winner = g;
        st.last_feasible = (int32_t)nfeas;
        if (st.limit > 0 && st.placed >= st.limit) st.done = DONE_LIMIT;
    }
    *a.st = st;
}

// k_final: one block.  Reduces the per-block partials of the scan that just finished, then either
// decides + commits (single GPU) or publishes the shard's record for the cross-GPU exchange.
template <class A>
__device__ void final_body(const A &a) {
    if (a.st->done) return;
    const int tid = threadIdx.x;
    if (a.n_ranks > 0 && a.st->smp_K > 0 && a.st->smp_phase == 0) { // sharded sampled search, counting pass: this shard's two counts
        if (tid == 0) {
            XRec r{};
            r.pad[0] = a.st->smp_Ftotal, r.pad[1] = a.st->smp_Fs; // (k_smp_prefix: over this shard's nodes)
            *a.xsend = r;
        }
        return;
    }
    uint64_t key = 0;
    uint32_t mt = 0, ma = 0;
    int64_t nf = 0;
    for (int i = tid; i < a.n_partials; i += kThreads) {
        const uint64_t qk = ld_agent(a.partials + 2 * (int64_t)i), qn = ld_agent(a.partials + 2 * (int64_t)i + 1);
        key = qk > key ? qk : key;
        const uint32_t norm = (uint32_t)qn;
        const uint32_t m1 = norm >> kStatCntShift, m2 = norm & kStatAffMask;
        mt = m1 > mt ? m1 : mt;
        ma = m2 > ma ? m2 : ma;
        nf += (uint32_t)(qn >> 32);
    }
    key = wave_max_u64(key);
    mt = wave_max_u32(mt);
    ma = wave_max_u32(ma);
    nf = wave_sum_i64(nf);
    __shared__ uint64_t s_key[kThreads / 64];
    __shared__ uint32_t s_mt[kThreads / 64], s_ma[kThreads / 64];
    __shared__ int64_t s_nf[kThreads / 64];
    if ((tid & 63) == 0) {
        s_key[tid >> 6] = key;
        s_mt[tid >> 6] = mt;
        s_ma[tid >> 6] = ma;
        s_nf[tid >> 6] = nf;
    }
    __syncthreads();
    // topology-coupled plugins: global minimum match count per spread constraint, min / max InterPodAffinity score
    __shared__ int32_t s_pm[kThreads / 64][kMaxTsc];
    __shared__ int64_t s_im[2][kThreads / 64];
    for (int c = 0; c < a.pts.n; c++) {
        int32_t v = 0x7fffffff;
        for (int i = tid; i < a.n_partials; i += kThreads) {
            const int32_t q = (int32_t)(int64_t)ld_agent(a.pts_min_partials + (int64_t)i * kMaxTsc + c);
            v = q < v ? q : v;
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const int32_t o = __shfl_xor(v, off, 64);
            v = o < v ? o : v;
        }
        if ((tid & 63) == 0) s_pm[tid >> 6][c] = v;
    }
    if (a.ipa.on) {
        int64_t mn = INT64_MAX, mx = INT64_MIN;
        for (int i = tid; i < a.n_partials; i += kThreads) {
            const int64_t q1 = (int64_t)ld_agent(a.ipa_partials + 2 * (int64_t)i), q2 = (int64_t)ld_agent(a.ipa_partials + 2 * (int64_t)i + 1);
            mn = q1 < mn ? q1 : mn;
            mx = q2 > mx ? q2 : mx;
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const int64_t o1 = __shfl_xor(mn, off, 64), o2 = __shfl_xor(mx, off, 64);
            mn = o1 < mn ? o1 : mn;
            mx = o2 > mx ? o2 : mx;
        }
        if ((tid & 63) == 0) s_im[0][tid >> 6] = mn, s_im[1][tid >> 6] = mx;
    }
    __shared__ int64_t s_sf[3 + kMaxTsc][kThreads / 64];
    if (a.soft.n) {
        const int32_t epoch = (int32_t)(a.st->scans + 1); // the scan that just finished stamped this value
        int64_t cn = 0, mn = INT64_MAX, mx = 0;
        for (int i = tid; i < a.n_partials; i += kThreads) {
            cn += (int64_t)ld_agent(a.soft_partials + 3 * (int64_t)i);
            const int64_t q1 = (int64_t)ld_agent(a.soft_partials + 3 * (int64_t)i + 1), q2 = (int64_t)ld_agent(a.soft_partials + 3 * (int64_t)i + 2);
            mn = q1 < mn ? q1 : mn;
            mx = q2 > mx ? q2 : mx;
        }
        cn = wave_sum_i64(cn);
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const int64_t o1 = __shfl_xor(mn, off, 64), o2 = __shfl_xor(mx, off, 64);
            mn = o1 < mn ? o1 : mn;
            mx = o2 > mx ? o2 : mx;
        }
        if ((tid & 63) == 0) s_sf[0][tid >> 6] = cn, s_sf[1][tid >> 6] = mn, s_sf[2][tid >> 6] = mx;
        for (int c = 0; c < a.soft.n; c++) { // candidate domains: value ids stamped by this scan
            int64_t d = 0;
            if (!a.soft.is_hostname[c])
                for (int v = 1 + tid; v <= a.soft.n_domains[c]; v += kThreads) d += a.soft.flag[c][v] == epoch;
            d = wave_sum_i64(d);
            if ((tid & 63) == 0) s_sf[3 + c][tid >> 6] = d;
        }
    }
    __shared__ unsigned long long s_sbits[kXSoftBitWords]; // sharded: the candidate-domain SETS (the sizes are not additive across shards)
    if (a.soft.n && a.n_ranks > 0) {
        if (tid < kXSoftBitWords) s_sbits[tid] = 0ull;
        __syncthreads();
        const int32_t epoch = (int32_t)(a.st->scans + 1);
        for (int c = 0; c < a.soft.n; c++) {
            if (a.soft.is_hostname[c]) continue;
            const int off = soft_bit_offset(a.soft, c);
            for (int v = 1 + tid; v <= a.soft.n_domains[c]; v += kThreads)
                if (a.soft.flag[c][v] == epoch) atomicOr(&s_sbits[(off + v - 1) >> 6], 1ull << ((off + v - 1) & 63));
        }
    }
    __syncthreads();
    if (tid != 0) return;
    key = 0, mt = 0, ma = 0, nf = 0;
#pragma unroll
    for (int w = 0; w < kThreads / 64; w++) {
        key = s_key[w] > key ? s_key[w] : key;
        mt = s_mt[w] > mt ? s_mt[w] : mt;
        ma = s_ma[w] > ma ? s_ma[w] : ma;
        nf += s_nf[w];
    }
    SoftAgg soft{};
    if (a.soft.n) {
        int64_t cn = 0;
        soft.mn = INT64_MAX, soft.mx = 0;
        for (int w = 0; w < kThreads / 64; w++) {
            cn += s_sf[0][w];
            soft.mn = s_sf[1][w] < soft.mn ? s_sf[1][w] : soft.mn;
            soft.mx = s_sf[2][w] > soft.mx ? s_sf[2][w] : soft.mx;
        }
        for (int c = 0; c < a.soft.n; c++) {
            int64_t d = 0;
            for (int w = 0; w < kThreads / 64; w++) d += s_sf[3 + c][w];
            soft.size[c] = a.soft.is_hostname[c] ? cn : d;
        }
    }
    int32_t pts_min[kMaxTsc];
    for (int c = 0; c < a.pts.n; c++) {
        int32_t v = s_pm[0][c];
        for (int w = 1; w < kThreads / 64; w++) v = s_pm[w][c] < v ? s_pm[w][c] : v;
        pts_min[c] = v;
    }
    int64_t ipa_mn = 0, ipa_mx = 0;
    if (a.ipa.on) {
        ipa_mn = s_im[0][0], ipa_mx = s_im[1][0];
        for (int w = 1; w < kThreads / 64; w++) {
            ipa_mn = s_im[0][w] < ipa_mn ? s_im[0][w] : ipa_mn;
            ipa_mx = s_im[1][w] > ipa_mx ? s_im[1][w] : ipa_mx;
        }
    }
    if (a.n_ranks > 0) { // publish this shard's record; k_decide finishes after the exchange
        XRec r{};
        r.key = (int64_t)key;
        r.mt = mt;
        r.ma = ma;
        r.nfeas = nf;
        for (int c = 0; c < kMaxTsc; c++) r.pts_min[c] = c < a.pts.n ? pts_min[c] : 0x7fffffff;
        r.ipa_mn = a.ipa.on ? ipa_mn : INT64_MAX;
        r.ipa_mx = a.ipa.on ? ipa_mx : INT64_MIN;
        // the sampled search's keys carry the VISITING POSITION of a node, not its index (k_scan<SMP = 2>): back to the node for its topology ids
        const bool smp_rec = a.st->smp_K > 0;
        auto key_node = [&](uint64_t k) -> int64_t {
            int64_t g = key_index(k);
            if (smp_rec) {
                g += a.st->smp_start;
                g = g >= a.st->smp_N ? g - a.st->smp_N : g;
            }
            return g - a.c.global_offset;
        };
        if (a.soft.n) {
            int64_t *w = xrec_soft(r);
            int64_t cn = 0;
            for (int x = 0; x < kThreads / 64; x++) cn += s_sf[0][x];
            w[0] = cn, w[1] = soft.mn, w[2] = soft.mx, w[3] = 0;
            for (int q = 0; q < kXSoftBitWords; q++) w[4 + q] = (int64_t)s_sbits[q];
            r.pad[0] = r.pad[1] = 0;
            if (key) {
                const int64_t i = key_node(key);
                w[3] = a.soft.elig[i];
                for (int c = 0; c < a.soft.n; c++) xrec_soft_ids(r)[c] = a.soft.label[c][i];
            }
        }
        if (key && (a.pts.n || a.ipa.on)) { // topology value ids of this shard's best node
            const int64_t i = key_node(key);
            r.win_elig = a.pts.n ? a.pts.elig[i] : 0;
            for (int c = 0; c < a.pts.n; c++) r.win_pts_v[c] = a.pts.label[c][i];
            for (int k = 0; k < a.ipa.n_keys; k++) r.win_ipa_v[k] = a.ipa.label[k][i];
        }
        // sampled search: the cancelling node's visiting position, if this shard owns it -- position + 1 in the upper half of the
        // eligibility word (its lower half are the winner's 32 eligibility bits; pad[] belongs to the soft constraints' ids)
        if (smp_rec) r.win_elig = (int64_t)(((uint64_t)r.win_elig & 0xffffffffull) | ((uint64_t)(a.st->smp_stop + 1) << 32));
        *a.xsend = r;
        return;
    }
    decide_commit(a, key, mt, ma, nf, pts_min, ipa_mn, ipa_mx, a.soft.n ? &soft : nullptr);
}

// k_final: the one-block reduction + decision that follows every k_scan.
__global__ __launch_bounds__(kThreads) void k_final(ScanArgs a) { final_body(a); }

// k_decide (distributed): every rank reduces the gathered records identically, so all ranks agree on
// the winner; only the owning rank's columns change ("only the owning rank updates").
__global__ void k_decide(ScanArgs a) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (a.st->done) return;
    const bool smp = a.st->smp_K > 0;
    if (smp && a.st->smp_phase == 0) { // sampled search, after the counting pass: the cluster's totals and this shard's offset
        int64_t total = 0, before = 0, off = 0;
        for (int r = 0; r < a.n_ranks; r++) {
            const XRec &q = a.xrecv[r];
            off += r < a.st->smp_rank ? q.pad[0] : 0;
            total += q.pad[0], before += q.pad[1];
        }
        DevState &st = *a.st;
        st.scans += 1;
        st.smp_Ftotal = total, st.smp_Fs = before, st.smp_off = off, st.smp_stop = -1;
        // (with a hard spread constraint the counting pass filtered with an ASSUMED global minimum, which only the scoring pass verifies:
        // "no feasible node" is its verdict then -- decide_commit with an empty key, after the minimum has been checked)
        if (total == 0 && a.pts.n == 0) { // no feasible node anywhere: FitError (schedule_one.go:448-454); every node was visited
            st.done = DONE_UNSCHEDULABLE, st.rounds += 1, st.winner = -1;
            st.last_feasible = 0, st.last_evaluated = (int32_t)st.smp_N, st.evaluated += st.smp_N;
        } else
            st.smp_phase = 1;
        return;
    }
    const int64_t rounds_before = a.st->rounds;
    int32_t pts_min_before[kMaxTsc];
    for (int c = 0; c < kMaxTsc; c++) pts_min_before[c] = a.st->pts_min_a[c];
    if (smp) { // the cancelling node's position travels with its owner's record
        int64_t stop = -1;
        for (int r = 0; r < a.n_ranks; r++) {
            const int64_t q = (int64_t)((uint64_t)a.xrecv[r].win_elig >> 32) - 1; // (final_body: position + 1 in the upper half, 0 = not this shard's)
            stop = q > stop ? q : stop;
        }
        a.st->smp_stop = stop;
    }
    uint64_t key = 0;
    uint32_t mt = 0, ma = 0;
    int64_t nf = 0, ipa_mn = INT64_MAX, ipa_mx = INT64_MIN;
    int32_t pts_min[kMaxTsc];
    for (int c = 0; c < kMaxTsc; c++) pts_min[c] = 0x7fffffff;
    int win = -1;
    for (int r = 0; r < a.n_ranks; r++) {
        const XRec &q = a.xrecv[r];
        if ((uint64_t)q.key > key) key = (uint64_t)q.key, win = r;
        mt = (uint32_t)q.mt > mt ? (uint32_t)q.mt : mt;
        ma = (uint32_t)q.ma > ma ? (uint32_t)q.ma : ma;
        nf += q.nfeas;
        for (int c = 0; c < a.pts.n; c++) pts_min[c] = (int32_t)q.pts_min[c] < pts_min[c] ? (int32_t)q.pts_min[c] : pts_min[c];
        ipa_mn = q.ipa_mn < ipa_mn ? q.ipa_mn : ipa_mn;
        ipa_mx = q.ipa_mx > ipa_mx ? q.ipa_mx : ipa_mx;
    }
    WinnerTopo wt{};
    const bool coupled = a.pts.n || a.ipa.on || a.soft.n;
    if (coupled && win >= 0) { // the winner's domain ids travel with its record
        const XRec &q = a.xrecv[win];
        wt.elig = (uint32_t)q.win_elig;
        for (int c = 0; c < kMaxTsc; c++) wt.pts_v[c] = q.win_pts_v[c];
        for (int k = 0; k < 4; k++) wt.ipa_v[k] = q.win_ipa_v[k];
        if (a.soft.n) {
            wt.soft_elig = (uint32_t)xrec_soft(q)[3];
            for (int c = 0; c < kMaxTsc; c++) wt.soft_v[c] = xrec_soft_ids(q)[c];
        }
    }
    SoftAgg soft{};
    if (a.soft.n) { // PodTopologySpread PreScore facts of the whole cluster: counts add, ranges combine, candidate sets unite
        int64_t cn = 0;
        unsigned long long bits[kXSoftBitWords] = {};
        soft.mn = INT64_MAX, soft.mx = 0;
        for (int r = 0; r < a.n_ranks; r++) {
            const int64_t *w = xrec_soft(a.xrecv[r]);
            cn += w[0];
            soft.mn = w[1] < soft.mn ? w[1] : soft.mn;
            soft.mx = w[2] > soft.mx ? w[2] : soft.mx;
            for (int q = 0; q < kXSoftBitWords; q++) bits[q] |= (unsigned long long)w[4 + q];
        }
        for (int c = 0; c < a.soft.n; c++) {
            if (a.soft.is_hostname[c]) {
                soft.size[c] = cn;
                continue;
            }
            const int off = soft_bit_offset(a.soft, c);
            int64_t d = 0;
            for (int v = 1; v <= a.soft.n_domains[c]; v++) d += (bits[(off + v - 1) >> 6] >> ((off + v - 1) & 63)) & 1ull;
            soft.size[c] = d;
        }
    }
    decide_commit(a, key, mt, ma, nf, pts_min, ipa_mn, ipa_mx, a.soft.n ? &soft : nullptr, coupled ? &wt : nullptr);
    if (smp && (a.st->rounds != rounds_before || a.st->done)) a.st->smp_phase = 0; // the cycle ended (a stale maximum repeats the scoring pass only)
    if (smp) // ... a stale global minimum of a hard spread constraint changes which nodes PASS: the counts are void, back to the counting pass
        for (int c = 0; c < a.pts.n; c++)
            if (a.st->pts_min_a[c] != pts_min_before[c]) a.st->smp_phase = 0;
}

// ------------------------------------------------------------------------------------------------
// k_scan_fused: the sequential mode's cycle as ONE dispatch (profiles without topology-coupled plugins, every node scored).
// k_scan -> k_final spends ~40 % of a cycle on the one-block reduction + decision and the dispatch boundary in front of it
// (10.2 us scan, ~7 us decision at 1M nodes, profiles/r02).  Here the decision of cycle t is the PROLOGUE of the scan of
// cycle t + 1, replicated: every block reduces the <= 1024 partial records of the previous scan (L2-resident), runs the same
// decision on the same numbers, and knows the winner.  No other block's data is touched: the thread that owns the winner
// applies NodeInfo.update (types.go:409-428) to the values it has just loaded, stores them, and scores the node in its new
// state -- a node's columns are read by exactly one thread of the grid, so there is nothing to race with.  State and partials
// are double-buffered by cycle parity (a block still reading buffer t must not see block 0 writing the state of t + 1).
// The last decision of a batch is made by k_final on the buffer the host reads.
// ------------------------------------------------------------------------------------------------
struct FusedArgs {
    DevCols c;
    DevPod p;
    DevState *st[2];        // [parity]
    uint64_t *partials[2];  // [parity][grid][2]
    int64_t chunk;
    int32_t n_partials, parity; // this launch READS st[parity] / partials[parity] and WRITES the other pair
    int32_t *log;
};

// the decision of one cycle on the reduced record: a pure function of a few scalars of the state (no working copy of the 500-byte
// DevState: as a local of every thread of the grid it is a scratch frame -- 96 MB of scratch traffic per launch at 1M nodes)
struct FusedDecision {
    int64_t winner, placed, rounds; // winner: global index or -1
    int32_t mt_a, ma_a, done, last_feasible;
    bool rescan;
};
__device__ __forceinline__ FusedDecision fused_decide(const DevState *s, uint64_t key, uint32_t mt, uint32_t ma, int64_t nfeas) {
    FusedDecision d;
    d.winner = -1, d.placed = s->placed, d.rounds = s->rounds, d.mt_a = s->mt_a, d.ma_a = s->ma_a, d.done = 0, d.last_feasible = s->last_feasible;
    d.rescan = false;
    if (key == 0) d.done = DONE_UNSCHEDULABLE, d.rounds += 1, d.last_feasible = 0; // schedule_one.go:448-454
    else if ((int32_t)mt != d.mt_a || (int32_t)ma != d.ma_a) d.mt_a = (int32_t)mt, d.ma_a = (int32_t)ma, d.rescan = true; // stale maxima: this scan redoes it
    else {
        d.winner = key_index(key);
        d.placed += 1, d.rounds += 1, d.last_feasible = (int32_t)nfeas;
        const int64_t limit = s->limit;
        if (limit > 0 && d.placed >= limit) d.done = DONE_LIMIT; // simulator.go:297-312
    }
    return d;
}
// (one thread) the state after the decision: everything else is carried over
__device__ __forceinline__ void fused_store(DevState *out, const DevState *in, const FusedDecision &d, int32_t have_prev, int64_t n_nodes) {
    if (out != in) *out = *in;
    out->winner = d.winner, out->placed = d.placed, out->rounds = d.rounds, out->mt_a = d.mt_a, out->ma_a = d.ma_a, out->done = d.done;
    out->last_feasible = d.last_feasible, out->have_prev = have_prev;
    out->scans = in->scans + 1;
    out->last_evaluated = (int32_t)n_nodes;
}

template <int NX, bool NARROW>
__global__ __launch_bounds__(kThreads) void k_scan_fused(FusedArgs a) {
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63; // (wave: uniform, and known to the compiler as such)
    __shared__ uint64_t s_key[kThreads / 64];
    __shared__ uint32_t s_mt[kThreads / 64], s_ma[kThreads / 64], s_nf[kThreads / 64];
    __shared__ int64_t s_nf64[kThreads / 64];
    const DevState *sin = a.st[a.parity];
    if (sin->done) { // the run ended in an earlier launch of this batch: carry the final state to the other buffer, do nothing else
        if (blockIdx.x == 0 && tid == 0 && a.st[a.parity ^ 1]->done != sin->done) *a.st[a.parity ^ 1] = *sin;
        return;
    }
    // ---- prologue: the previous scan's decision, in every block
    FusedDecision d;
    d.winner = -1, d.placed = sin->placed, d.rounds = sin->rounds, d.mt_a = sin->mt_a, d.ma_a = sin->ma_a, d.done = 0, d.last_feasible = sin->last_feasible;
    d.rescan = false;
    const bool have_prev = sin->have_prev != 0;
    if (have_prev) {
        uint64_t key = 0;
        uint32_t mt = 0, ma = 0;
        int64_t nf = 0;
        const uint64_t *pin = a.partials[a.parity];
        for (int i = tid; i < a.n_partials; i += kThreads) {
            const uint64_t qk = ld_agent(pin + 2 * (int64_t)i), qn = ld_agent(pin + 2 * (int64_t)i + 1);
            key = qk > key ? qk : key;
            const uint32_t norm = (uint32_t)qn, m1 = norm >> kStatCntShift, m2 = norm & kStatAffMask;
            mt = m1 > mt ? m1 : mt, ma = m2 > ma ? m2 : ma;
            nf += (uint32_t)(qn >> 32);
        }
        key = wave_max_u64(key), mt = wave_max_u32(mt), ma = wave_max_u32(ma), nf = wave_sum_i64(nf);
        if (lane == 0) s_key[wave] = key, s_mt[wave] = mt, s_ma[wave] = ma, s_nf64[wave] = nf;
        __syncthreads();
        key = 0, mt = 0, ma = 0, nf = 0;
#pragma unroll
        for (int w = 0; w < kThreads / 64; w++) {
            key = s_key[w] > key ? s_key[w] : key, mt = s_mt[w] > mt ? s_mt[w] : mt, ma = s_ma[w] > ma ? s_ma[w] : ma;
            nf += s_nf64[w];
        }
        __syncthreads(); // (s_key / s_mt / s_ma are reused by the scan's own reduction below)
        d = fused_decide(sin, key, mt, ma, nf);
    }
    const int64_t w_g = d.winner;
    if (blockIdx.x == 0 && tid == 0) {
        if (w_g >= 0 && a.log && d.placed - 1 < sin->log_cap) a.log[d.placed - 1] = (int32_t)w_g;
        DevState *out = a.st[a.parity ^ 1]; // (never the buffer this launch reads: a block that starts late must see what the others saw)
        if (have_prev) fused_store(out, sin, d, 1, a.c.n);
        else {
            *out = *sin;
            out->have_prev = 1, out->winner = -1;
        }
    }
    if (d.done == DONE_UNSCHEDULABLE) return; // nothing to apply, nothing to scan for
    // (DONE_LIMIT: the last winner is still to be applied by the thread that owns it -- one more pass, its partials unused)

    const uint32_t mt = (uint32_t)d.mt_a, ma = (uint32_t)d.ma_a;
    const int64_t lo = (int64_t)blockIdx.x * a.chunk;
    int64_t hi = lo + a.chunk;
    if (hi > a.c.n_pad) hi = a.c.n_pad;
    const int64_t w_i = w_g - a.c.global_offset; // the winner's index in this shard (-1 - offset: none)
    uint64_t best = 0;
    uint32_t mt_b = 0, ma_b = 0, nfeas = 0;
    const NarrowPod npod = narrow_pod(a.p, a.c.mem_shift);
    for (int64_t base = lo; base < hi; base += kTile) {
        const int64_t i0 = base + 2 * tid;
        const uint2 sw = *reinterpret_cast<const uint2 *>(a.c.stat + i0);
        int2 a0n{}, a1n{}, r0n{}, r1n{}, z0n{}, z1n{};
        longlong2 A0{}, A1{}, R0{}, R1{}, Z0{}, Z1{};
        if (NARROW) {
            a0n = *reinterpret_cast<const int2 *>(a.c.a32[0] + i0), a1n = *reinterpret_cast<const int2 *>(a.c.a32[1] + i0);
            r0n = *reinterpret_cast<const int2 *>(a.c.r32[0] + i0), r1n = *reinterpret_cast<const int2 *>(a.c.r32[1] + i0);
            z0n = *reinterpret_cast<const int2 *>(a.c.z32[0] + i0), z1n = *reinterpret_cast<const int2 *>(a.c.z32[1] + i0);
        } else {
            const Cols6 c6 = load_cols6<false>(a.c, i0);
            A0 = c6.A0, A1 = c6.A1, R0 = c6.R0, R1 = c6.R1, Z0 = c6.Z0, Z1 = c6.Z1;
        }
        const int2 AP = *reinterpret_cast<const int2 *>(a.c.alloc_pods + i0);
        int2 NP = *reinterpret_cast<const int2 *>(a.c.pod_count + i0);
        int64_t xa0[NX > 0 ? NX : 1], xr0[NX > 0 ? NX : 1], xa1[NX > 0 ? NX : 1], xr1[NX > 0 ? NX : 1];
        if (NX > 0) {
#pragma unroll
            for (int x = 0; x < NX; x++) {
                xa0[x] = xr0[x] = xa1[x] = xr1[x] = 0;
                if (x < a.p.nx) {
                    const int col = a.p.xcol[x];
                    const longlong2 XA = *reinterpret_cast<const longlong2 *>(a.c.alloc[col] + i0);
                    const longlong2 XR = *reinterpret_cast<const longlong2 *>(a.c.req[col] + i0);
                    xa0[x] = XA.x, xr0[x] = XR.x, xa1[x] = XA.y, xr1[x] = XR.y;
                }
            }
        }
        // ---- the winner of the previous cycle lives in this pair: NodeInfo.update on the loaded values, stored back
        if (w_i == i0 || w_i == i0 + 1) {
            const bool second = w_i == i0 + 1;
            if (NARROW) { // the int64 columns are the canonical state: read-modify-write them too (this one thread)
                const int64_t q0 = a.c.req[0][w_i] + a.p.req[0], q1 = a.c.req[1][w_i] + a.p.req[1];
                const int64_t y0 = a.c.nz_mcpu[w_i] + a.p.nz_mcpu, y1 = a.c.nz_mem[w_i] + a.p.nz_mem;
                a.c.req[0][w_i] = q0, a.c.req[1][w_i] = q1, a.c.nz_mcpu[w_i] = y0, a.c.nz_mem[w_i] = y1;
                store_mirror(a.c, w_i, q0, q1, y0, y1);
                (second ? r0n.y : r0n.x) += npod.req0, (second ? r1n.y : r1n.x) += npod.req1;
                (second ? z0n.y : z0n.x) += npod.nz0, (second ? z1n.y : z1n.x) += npod.nz1;
            } else {
                (second ? R0.y : R0.x) += a.p.req[0], (second ? R1.y : R1.x) += a.p.req[1];
                (second ? Z0.y : Z0.x) += a.p.nz_mcpu, (second ? Z1.y : Z1.x) += a.p.nz_mem;
                a.c.req[0][w_i] = second ? R0.y : R0.x, a.c.req[1][w_i] = second ? R1.y : R1.x;
                a.c.nz_mcpu[w_i] = second ? Z0.y : Z0.x, a.c.nz_mem[w_i] = second ? Z1.y : Z1.x;
                store_mirror(a.c, w_i, second ? R0.y : R0.x, second ? R1.y : R1.x, second ? Z0.y : Z0.x, second ? Z1.y : Z1.x);
            }
            (second ? NP.y : NP.x) += 1;
            a.c.pod_count[w_i] = second ? NP.y : NP.x;
            a.c.placed_cnt[w_i] += 1;
            if (NX > 0) {
#pragma unroll
                for (int x = 0; x < NX; x++)
                    if (x < a.p.nx && a.p.req[a.p.xcol[x]] != 0) {
                        (second ? xr1[x] : xr0[x]) += a.p.req[a.p.xcol[x]];
                        a.c.req[a.p.xcol[x]][w_i] = second ? xr1[x] : xr0[x];
                    }
            }
        }
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const uint32_t w = k ? sw.y : sw.x;
            const int64_t a_cpu = k ? A0.y : A0.x, a_mem = k ? A1.y : A1.x, r_cpu = k ? R0.y : R0.x, r_mem = k ? R1.y : R1.x;
            const int64_t z_cpu = k ? Z0.y : Z0.x, z_mem = k ? Z1.y : Z1.x;
            const int32_t a_pods = k ? AP.y : AP.x, npods = k ? NP.y : NP.x;
            const int32_t na0 = k ? a0n.y : a0n.x, na1 = k ? a1n.y : a1n.x, nr0 = k ? r0n.y : r0n.x, nr1 = k ? r1n.y : r1n.x;
            const int32_t nz0 = k ? z0n.y : z0n.x, nz1 = k ? z1n.y : z1n.x;
            bool feasible = (w >> kStatOkBit) && (NARROW ? fits_narrow(a.p, npod, na0, na1, nr0, nr1, a_pods, npods)
                                                         : fits_core(a.p, a_cpu, a_mem, r_cpu, r_mem, a_pods, npods));
            int64_t xak[NX > 0 ? NX : 1], xrk[NX > 0 ? NX : 1];
#pragma unroll
            for (int x = 0; x < (NX > 0 ? NX : 1); x++) {
                xak[x] = k ? xa1[x] : xa0[x], xrk[x] = k ? xr1[x] : xr0[x];
                if (NX > 0 && x < a.p.nx) {
                    const int64_t rq = a.p.req[a.p.xcol[x]];
                    if (a.p.fit_enabled && !a.p.all_zero_req && rq > 0 && rq > xak[x] - xrk[x]) feasible = false;
                }
            }
            nfeas += (uint32_t)__popcll(__ballot(feasible));
            if (feasible) {
                const uint32_t cnt = (w >> kStatCntShift) & kStatCntMask, aff = w & kStatAffMask, img = (w >> kStatImgShift) & kStatImgMask;
                const int64_t total = static_score(a.p, cnt, aff, img, mt, ma) +
                                      (NARROW ? dynamic_score_narrow(a.p, npod, na0, na1, nr0, nr1, nz0, nz1)
                                              : (NX > 0 && a.p.gen_score ? dynamic_score_gen<NX>(a.p, a_cpu, a_mem, r_cpu, r_mem, z_cpu, z_mem, xak, xrk)
                                                                         : dynamic_score(a.p, make_rcp(a_cpu, a_mem), a_cpu, a_mem, r_cpu, r_mem, z_cpu, z_mem)));
                const uint64_t key = make_key(total, a.c.global_offset + i0 + k);
                best = key > best ? key : best;
                mt_b = cnt > mt_b ? cnt : mt_b, ma_b = aff > ma_b ? aff : ma_b;
            }
        }
    }
    best = wave_max_u64(best), mt_b = wave_max_u32(mt_b), ma_b = wave_max_u32(ma_b);
    if (lane == 0) s_key[wave] = best, s_mt[wave] = mt_b, s_ma[wave] = ma_b, s_nf[wave] = nfeas;
    __syncthreads();
    if (tid == 0) {
        uint64_t k = 0;
        uint32_t m1 = 0, m2 = 0, nf = 0;
#pragma unroll
        for (int w = 0; w < kThreads / 64; w++) {
            k = s_key[w] > k ? s_key[w] : k, m1 = s_mt[w] > m1 ? s_mt[w] : m1, m2 = s_ma[w] > m2 ? s_ma[w] : m2;
            nf += s_nf[w];
        }
        uint64_t *pout = a.partials[a.parity ^ 1];
        st_agent(pout + 2 * (int64_t)blockIdx.x, k);
        st_agent(pout + 2 * (int64_t)blockIdx.x + 1, (uint64_t)((m1 << kStatCntShift) | m2) | ((uint64_t)nf << 32));
    }
}

// k_final_fused: the pending decision at the end of a batch of fused cycles (one block, buffer `parity` = the one the host reads):
// reduce, decide, apply the winner to the columns -- afterwards nothing is pending (have_prev = 0).
__global__ __launch_bounds__(kThreads) void k_final_fused(FusedArgs a) {
    DevState *sp = a.st[a.parity];
    if (sp->done || !sp->have_prev) return;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63; // (wave: uniform, and known to the compiler as such)
    __shared__ uint64_t s_key[kThreads / 64];
    __shared__ uint32_t s_mt[kThreads / 64], s_ma[kThreads / 64];
    __shared__ int64_t s_nf64[kThreads / 64];
    uint64_t key = 0;
    uint32_t mt = 0, ma = 0;
    int64_t nf = 0;
    const uint64_t *pin = a.partials[a.parity];
    for (int i = tid; i < a.n_partials; i += kThreads) {
        const uint64_t qk = ld_agent(pin + 2 * (int64_t)i), qn = ld_agent(pin + 2 * (int64_t)i + 1);
        key = qk > key ? qk : key;
        const uint32_t norm = (uint32_t)qn, m1 = norm >> kStatCntShift, m2 = norm & kStatAffMask;
        mt = m1 > mt ? m1 : mt, ma = m2 > ma ? m2 : ma;
        nf += (uint32_t)(qn >> 32);
    }
    key = wave_max_u64(key), mt = wave_max_u32(mt), ma = wave_max_u32(ma), nf = wave_sum_i64(nf);
    if (lane == 0) s_key[wave] = key, s_mt[wave] = mt, s_ma[wave] = ma, s_nf64[wave] = nf;
    __syncthreads();
    if (tid != 0) return;
    key = 0, mt = 0, ma = 0, nf = 0;
    for (int w = 0; w < kThreads / 64; w++) {
        key = s_key[w] > key ? s_key[w] : key, mt = s_mt[w] > mt ? s_mt[w] : mt, ma = s_ma[w] > ma ? s_ma[w] : ma;
        nf += s_nf64[w];
    }
    const FusedDecision d = fused_decide(sp, key, mt, ma, nf);
    const int64_t i = d.winner - a.c.global_offset;
    if (d.winner >= 0 && i >= 0 && i < a.c.n) { // NodeInfo.update (types.go:409-428)
        const int64_t r0 = a.c.req[0][i] + a.p.req[0], r1 = a.c.req[1][i] + a.p.req[1];
        const int64_t z0 = a.c.nz_mcpu[i] + a.p.nz_mcpu, z1 = a.c.nz_mem[i] + a.p.nz_mem;
        a.c.req[0][i] = r0, a.c.req[1][i] = r1, a.c.nz_mcpu[i] = z0, a.c.nz_mem[i] = z1;
        a.c.pod_count[i] += 1, a.c.placed_cnt[i] += 1;
        store_mirror(a.c, i, r0, r1, z0, z1);
#pragma unroll 1
        for (int col = 2; col < a.p.ncol; col++)
            if (a.p.req[col] != 0) a.c.req[col][i] += a.p.req[col];
        if (a.log && d.placed - 1 < sp->log_cap) a.log[d.placed - 1] = (int32_t)d.winner;
    }
    fused_store(sp, sp, d, 0, a.c.n);
}

// ------------------------------------------------------------------------------------------------
// k_static: once per pod spec.  NodeUnschedulable (node_unschedulable.go:133-150), TaintToleration
// filter + PreferNoSchedule count via the taint-set table (taint_toleration.go:111-121,169-194),
// NodeAffinity required match + preferred weight sum via requirement tables
// (component-helpers nodeaffinity.go:84-150; node_affinity.go:206-285).
// ------------------------------------------------------------------------------------------------
struct DevTerm {
    int32_t first_req, n_req, weight;
};
struct DevReq {
    int32_t col, table_off;
};
struct StaticArgs {
    int64_t n, n_pad;
    uint32_t filter_mask;
    const uint8_t *unschedulable;
    const int32_t *taintset_id;
    const uint8_t *taint_filter_ok;
    const int32_t *taint_prefer_cnt;
    int32_t tolerates_unschedulable;
    int32_t affinity_filter_active, has_node_selector, has_required_terms, n_required, n_preferred;
    DevTerm node_selector;
    const DevTerm *required;
    const DevTerm *preferred;
    const DevReq *reqs;
    const uint8_t *req_tables;
    const int32_t *const *label_cols; // device array of column pointers
    const uint8_t *ports_conflict;    // NodePorts: an existing pod of the node holds a conflicting host port (NULL: plugin off / none)
    const uint8_t *image_score;       // ImageLocality score per node, 0..100 (NULL: 0)
    const uint8_t *volume_veto;       // the volume plugins' verdict per node against the snapshot's pods (NULL: none); ccsim_pod.volume_veto
    uint32_t *stat;
    uint8_t *sreason;
};

__device__ __forceinline__ bool term_matches(const StaticArgs &a, const DevTerm &t, int64_t n, bool empty_matches) {
    if (t.n_req == 0) return empty_matches;
    for (int i = 0; i < t.n_req; i++) {
        const DevReq r = a.reqs[t.first_req + i];
        if (!a.req_tables[r.table_off + a.label_cols[r.col][n]]) return false;
    }
    return true;
}

__global__ __launch_bounds__(kThreads) void k_static(StaticArgs a) {
    const int64_t n = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (n >= a.n_pad) return;
    if (n >= a.n) { // padding: never feasible
        a.stat[n] = 0;
        a.sreason[n] = 0xff;
        return;
    }
    uint32_t reason = 0;
    const int32_t ts = a.taintset_id ? a.taintset_id[n] : 0;
    if ((a.filter_mask & 1u) && a.unschedulable && a.unschedulable[n] && !a.tolerates_unschedulable) reason = 1;
    if (!reason && (a.filter_mask & 4u) && !a.taint_filter_ok[ts]) reason = 2;
    if (!reason && (a.filter_mask & 8u) && a.affinity_filter_active) {
        bool m = true;
        if (a.has_node_selector) m = term_matches(a, a.node_selector, n, true);
        if (m && a.has_required_terms) {
            bool any = false;
            for (int t = 0; t < a.n_required && !any; t++) any = term_matches(a, a.required[t], n, false);
            m = any;
        }
        if (!m) reason = 3;
    }
    // NodePorts runs after NodeAffinity and before NodeResourcesFit (default_plugins.go:34-40); the ports of the clones
    // placed during the run are the engine's business (one clone per node: the clamped pod capacity, ccsim_set_pod)
    if (!reason && a.ports_conflict && a.ports_conflict[n]) reason = 4;
    // The volume plugins run AFTER NodeResourcesFit (default_plugins.go:40-45): the node is out either way, which plugin reports it is
    // k_hist's business (Fit first, with the node's state at the terminal cycle)
    if (!reason && a.volume_veto && a.volume_veto[n]) reason = 5;
    uint32_t cnt = (uint32_t)a.taint_prefer_cnt[ts];
    uint32_t aff = 0;
    for (int t = 0; t < a.n_preferred; t++)
        if (term_matches(a, a.preferred[t], n, false)) aff += (uint32_t)a.preferred[t].weight;
    const uint32_t img = a.image_score ? (uint32_t)a.image_score[n] : 0u;
    a.stat[n] = ((reason == 0 ? 1u : 0u) << kStatOkBit) | ((cnt & kStatCntMask) << kStatCntShift) | ((img & kStatImgMask) << kStatImgShift) |
                (aff & kStatAffMask);
    a.sreason[n] = (uint8_t)reason;
}

// ------------------------------------------------------------------------------------------------
// k_hist: terminal round only.  Per-node failure reasons exactly as the filter chain reports them
// (first failing plugin in order; NodeResourcesFit keeps ALL its insufficient resources,
// fit.go:520-531), histogrammed for FitError.Error (types.go:787-836).
//   hist[0] unschedulable, [1] nodename, [2] node affinity, [3] too many pods, [4+col] insufficient col, ... [kHistNodePorts],
//   hist_code[0] = nodes whose status code is plain Unschedulable (preemption dry-run candidates).
// ------------------------------------------------------------------------------------------------
struct HistArgs {
    DevCols c;
    DevPod p;
    unsigned long long *hist;      // [4 + kMaxRes + 2]
    unsigned long long *hist_ts;   // [n_taintsets]
    unsigned long long *hist_code; // [1]
    int32_t n_taintsets;
    DevPts pts;
    const DevState *st;
    DevIpa ipa;
    int32_t ports_on;               // one clone per node (the clamped pod capacity): NodePorts active for this pod, or its own disks conflict
    int32_t excl_ports;             // ... and which: 1 = host ports (reported before NodeResourcesFit), 0 = disks (VolumeRestrictions, after it)
    const uint8_t *volume_veto;     // ccsim_pod.volume_veto (sreason 5: which volume plugin, once Fit has passed)
    const int32_t *alloc_pods_real; // Allocatable.AllowedPodNumber (c.alloc_pods is the clamped copy while ports_on)
    const int32_t *ports_base;      // pods on the node when the clamp was built (k_ports_clamp): more than that = it holds a clone of this pod
};

constexpr int kHistVolCodes = 7;                // CCSIM_VOL_CODES
constexpr int kHistSlots = 4 + kMaxRes + 2 + 3 + 1 + kHistVolCodes + 1; // + NodePorts + the volume plugins + the status-code counter
constexpr int kHistNodePorts = 4 + kMaxRes + 2 + 3, kHistVol0 = kHistNodePorts + 1;
constexpr int kHistIpa = 4 + kMaxRes + 2;          // affinity, anti-affinity, existing pods' anti-affinity
constexpr int kHistPtsMissing = 4 + kMaxRes, kHistPtsSkew = 4 + kMaxRes + 1;
constexpr int kHistTsLds = 1024;                // taint sets histogrammed in LDS (more fall back to global atomics)

__global__ __launch_bounds__(kThreads) void k_hist(HistArgs a) {
    // block-private histograms in LDS, one global atomic per non-empty bin per block: the terminal round
    // typically puts ~all N nodes into one or two bins
    __shared__ unsigned int sh[kHistSlots];
    __shared__ unsigned int sh_ts[kHistTsLds];
    for (int i = threadIdx.x; i < kHistSlots; i += kThreads) sh[i] = 0;
    for (int i = threadIdx.x; i < kHistTsLds; i += kThreads) sh_ts[i] = 0;
    __syncthreads();
    for (int64_t n = (int64_t)blockIdx.x * kThreads + threadIdx.x; n < a.c.n; n += (int64_t)gridDim.x * kThreads) {
        const uint8_t sr = a.c.sreason[n];
        if (sr == 1) { atomicAdd(&sh[0], 1u); continue; }
        if (sr == 2) {
            const int32_t ts = a.c.taintset_id ? a.c.taintset_id[n] : 0;
            if (ts < kHistTsLds) atomicAdd(&sh_ts[ts], 1u); else atomicAdd(&a.hist_ts[ts], 1ull);
            continue;
        }
        if (sr == 3) { atomicAdd(&sh[2], 1u); continue; }
        const bool holds_clone = a.ports_on && a.c.pod_count[n] > a.ports_base[n];
        if (sr == 4 || (holds_clone && a.excl_ports)) { // node_ports.go:148-162: plain Unschedulable, before NodeResourcesFit
            atomicAdd(&sh[kHistNodePorts], 1u);
            atomicAdd(&sh[kHistSlots - 1], 1u);
            continue;
        }
        bool unresolvable = false, any = false;
        if (a.p.fit_enabled && (int64_t)a.c.pod_count[n] + 1 > (int64_t)a.alloc_pods_real[n]) { atomicAdd(&sh[3], 1u); any = true; }
        if (a.p.fit_enabled && !a.p.all_zero_req) {
            for (int col = 0; col < a.p.ncol; col++) {
                const int64_t rq = a.p.req[col];
                if (col < 3 ? !(rq > 0) : rq == 0) continue;
                const int64_t al = a.c.alloc[col] ? a.c.alloc[col][n] : 0;
                const int64_t us = a.c.req[col] ? a.c.req[col][n] : 0;
                if (rq > al - us) {
                    atomicAdd(&sh[4 + col], 1u);
                    any = true;
                    if (rq > al) unresolvable = true;
                }
            }
        }
        if (any) {
            if (!unresolvable) atomicAdd(&sh[kHistSlots - 1], 1u);
            continue;
        }
        // VolumeRestrictions / NodeVolumeLimits / VolumeBinding / VolumeZone follow NodeResourcesFit (default_plugins.go:40-45): a node
        // that holds a clone whose disks conflict with the next one's (volume_restrictions.go:310-313: the first check of the first of
        // them), else the caller's verdict against the snapshot's pods
        if (sr == 5 || holds_clone) {
            const int code = holds_clone ? 1 : (int)a.volume_veto[n];
            atomicAdd(&sh[kHistVol0 + code - 1], 1u);
            if (code <= 3) atomicAdd(&sh[kHistSlots - 1], 1u); // (CCSIM_VOL_LAST_UNSCHEDULABLE)
            continue;
        }
        // PodTopologySpread comes after NodeResourcesFit in the filter order (first failing plugin reports)
        bool pts_failed = false;
        for (int c = 0; c < a.pts.n && !pts_failed; c++) {
            const int32_t v = a.pts.label[c][n];
            if (!v) { atomicAdd(&sh[kHistPtsMissing], 1u); pts_failed = true; break; } // UnschedulableAndUnresolvable
            const int64_t minm = a.pts.n_present[c] < a.pts.min_domains[c] ? 0 : (int64_t)a.st->pts_min_a[c];
            if ((int64_t)a.pts.tbl[c][v] + a.pts.self_match[c] - minm > (int64_t)a.pts.max_skew[c]) {
                atomicAdd(&sh[kHistPtsSkew], 1u);
                atomicAdd(&sh[kHistSlots - 1], 1u); // plain Unschedulable
                pts_failed = true;
            }
        }
        if (pts_failed || !a.ipa.on || !a.ipa.filter_on) continue;
        const int code = ipa_filter(a.ipa, *a.st, n); // InterPodAffinity is the last filter of the default order
        if (code) {
            atomicAdd(&sh[kHistIpa + code - 1], 1u);
            if (code != 1) atomicAdd(&sh[kHistSlots - 1], 1u); // anti-affinity failures are plain Unschedulable
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < kHistSlots - 1; i += kThreads)
        if (sh[i]) atomicAdd(&a.hist[i], (unsigned long long)sh[i]);
    if (threadIdx.x == 0 && sh[kHistSlots - 1]) atomicAdd(&a.hist_code[0], (unsigned long long)sh[kHistSlots - 1]);
    for (int i = threadIdx.x; i < kHistTsLds && i < a.n_taintsets; i += kThreads)
        if (sh_ts[i]) atomicAdd(&a.hist_ts[i], (unsigned long long)sh_ts[i]);
}

// k_ports_clamp: NodePorts for a pod with host ports -- every clone holds the same ports, so a node takes at most one
// (NodeInfo.updateUsedPorts, S/framework/types.go:431-439; fitsPorts, node_ports.go:164-176).  The engine folds that into
// the pod-count test every Fit evaluation already makes: allowed pods = min(real, pods of the snapshot + 1).
// `base` = the pods on the node when the clamp was built (ccsim_set_pod, ccsim_reset_state): none of them holds the ports of THIS pod (its
// conflicts with the snapshot's pods are the static veto, clones of earlier pod specs carry other ports or none), so the node takes
// exactly one clone more than that; a node whose pod count has passed `base` holds a clone with the conflicting ports (k_hist).
__global__ __launch_bounds__(kThreads) void k_ports_clamp(int32_t *eff, int32_t *base, const int32_t *real, const int32_t *pod_count, int64_t n_pad) {
    const int64_t n = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (n >= n_pad) return;
    const int64_t one_more = (int64_t)pod_count[n] + 1;
    base[n] = pod_count[n];
    eff[n] = (int32_t)(one_more < (int64_t)real[n] ? one_more : (int64_t)real[n]);
}

// k_narrow_build: (re)derive the narrow mirrors from the wide columns (per pod spec, and after a state reset)
__global__ __launch_bounds__(kThreads) void k_narrow_build(DevCols c, int32_t *a32_cpu, int32_t *a32_mem) {
    const int64_t n = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (n >= c.n_pad) return;
    const int sh = c.mem_shift;
    a32_cpu[n] = (int32_t)c.alloc[0][n], a32_mem[n] = (int32_t)(c.alloc[1][n] >> sh);
    c.r32[0][n] = (int32_t)c.req[0][n], c.r32[1][n] = (int32_t)(c.req[1][n] >> sh);
    c.z32[0][n] = (int32_t)c.nz_mcpu[n], c.z32[1][n] = (int32_t)(c.nz_mem[n] >> sh);
}

// k_rows_build: start of a batched run on the narrow mirrors (columns -> commit rows)
__global__ __launch_bounds__(kThreads) void k_rows_build(DevCols c) {
    const int64_t n = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (n >= c.n_pad) return;
    int4 *row = reinterpret_cast<int4 *>(c.rows + n * kRowWords);
    row[0] = make_int4(c.a32[0][n], c.a32[1][n], c.alloc_pods[n], (int32_t)c.stat[n]);
    row[1] = make_int4(c.r32[0][n], c.r32[1][n], c.z32[0][n], c.z32[1][n]);
    row[2] = make_int4(c.pod_count[n], 0, 0, 0);
}

// k_rows_flush: commit rows -> columns (mirrors, int64 columns, pod counts).  `only_if_full`: the launch in front of a
// k_level_score, which reads the mirrors -- it has work only when that full pass is due.
__global__ __launch_bounds__(kThreads) void k_rows_flush(DevCols c, const DevState *st, int only_if_full) {
    if (only_if_full && (st->done || !st->lvl_full)) return;
    const int64_t n = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (n >= c.n_pad) return;
    const int4 *row = reinterpret_cast<const int4 *>(c.rows + n * kRowWords);
    const int4 d = row[1], q = row[2];
    const int sh = c.mem_shift;
    c.r32[0][n] = d.x, c.r32[1][n] = d.y, c.z32[0][n] = d.z, c.z32[1][n] = d.w;
    c.req[0][n] = (int64_t)d.x, c.req[1][n] = (int64_t)d.y << sh;
    c.nz_mcpu[n] = (int64_t)d.z, c.nz_mem[n] = (int64_t)d.w << sh;
    c.pod_count[n] = q.x;
    c.placed_cnt[n] = q.y;
}

__global__ void k_noop(int) {} // measurement marker: its stop stamp = the end of the preceding dispatch + one boundary

// k_pts_init: once per pod spec.  calPreFilterState (filtering.go:235-308) for the initial cluster: which nodes
// count (all hard keys present; inclusion policies), the match count of every domain, which domains exist.
struct PtsInitArgs {
    int64_t n;
    DevPts pts;
    uint8_t *elig;                          // out
    const int32_t *existing[kMaxTsc];       // pods already on the node matching the selector (NULL = 0)
    const uint8_t *included[kMaxTsc];       // node inclusion policies (NULL = all)
    int32_t *present[kMaxTsc];              // out: 1 for every value id holding >= 1 counted node
};

__global__ __launch_bounds__(kThreads) void k_pts_init(PtsInitArgs a) {
    const int64_t n = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (n >= a.n) return;
    bool all = true;
    for (int c = 0; c < a.pts.n; c++) all = all && a.pts.label[c][n] != 0;
    uint32_t eb = all ? 1u : 0u;
    for (int c = 0; c < a.pts.n; c++) {
        const bool inc = a.included[c] ? a.included[c][n] != 0 : true;
        if (inc) eb |= 1u << (1 + c);
        if (all && inc) {
            const int32_t v = a.pts.label[c][n];
            const int32_t ex = a.existing[c] ? a.existing[c][n] : 0;
            if (ex) atomicAdd(&a.pts.tbl[c][v], ex);
            a.present[c][v] = 1;
        }
    }
    a.elig[n] = (uint8_t)eb;
}

// k_ipa_init: once per pod spec.  PreFilter / PreScore maps of the initial cluster (existing pods only).
struct IpaInitArgs {
    int64_t n;
    DevIpa ipa;
    const int32_t *aff_existing;
    const int32_t *anti_existing[kMaxIpaTerms];
    const int32_t *exist_anti[kMaxIpaKeys];
    const int64_t *score_existing[kMaxIpaKeys];
    unsigned long long *totals; // [0] affinity entries, [1] existing anti-affinity entries
};

__global__ __launch_bounds__(kThreads) void k_ipa_init(IpaInitArgs a) {
    const int64_t n = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (n >= a.n) return;
    const int64_t am = a.aff_existing ? a.aff_existing[n] : 0;
    if (am)
        for (int t = 0; t < a.ipa.n_aff; t++) {
            const int k = a.ipa.aff_key[t];
            const int32_t v = a.ipa.label[k][n];
            if (v) {
                atomicAdd((unsigned long long *)&a.ipa.aff[k][v], (unsigned long long)am);
                atomicAdd(&a.totals[0], (unsigned long long)am);
            }
        }
    for (int t = 0; t < a.ipa.n_anti; t++) {
        const int64_t m = a.anti_existing[t] ? a.anti_existing[t][n] : 0;
        const int k = a.ipa.anti_key[t];
        const int32_t v = a.ipa.label[k][n];
        if (m && v) atomicAdd((unsigned long long *)&a.ipa.anti[k][v], (unsigned long long)m);
    }
    for (int k = 0; k < a.ipa.n_keys; k++) {
        const int32_t v = a.ipa.label[k][n];
        if (!v) continue;
        const int64_t m = a.exist_anti[k] ? a.exist_anti[k][n] : 0;
        if (m) {
            atomicAdd((unsigned long long *)&a.ipa.exist[k][v], (unsigned long long)m);
            atomicAdd(&a.totals[1], (unsigned long long)m);
        }
        const int64_t w = a.score_existing[k] ? a.score_existing[k][n] : 0;
        if (w) atomicAdd((unsigned long long *)&a.ipa.score[k][v], (unsigned long long)w); // two's complement: negatives add up too
    }
}

} // namespace ccsim
