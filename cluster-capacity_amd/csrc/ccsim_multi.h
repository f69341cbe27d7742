// ccsim_multi.h -- several pod specs cycled round-robin against one snapshot (BASELINE.json configs[4]: 100k nodes x 1024
// genpod-shaped specs with a DoNotSchedule zone spread and required hostname anti-affinity to their own label).
//
// Semantics (include/ccsim.h, ccsim_set_pods): placement i is a clone of spec i mod P; every cycle is the reference's
// schedulePod (S/schedule_one.go:430-478) for that pod against everything placed so far.  One cycle per pass -- a full
// pods x nodes scan, a one-block decision, a commit -- costs ~20 us, and 1024 different pods share nothing between passes
// except the node columns.  So a WINDOW of W consecutive pods (all different specs, W <= P) is resolved per pass:
//
//   k_multi_scan    P pods x nodes: a workgroup owns 2048 nodes (8 per thread, their narrow columns read ONCE into
//                   registers and shared by the pods of the workgroup's pod chunk) x 8 pods; per (pod, node): static word of
//                   the pod's class, NodeResourcesFit, the pod's spread tables (staged in LDS), its hostname anti-affinity
//                   bit, TaintToleration / NodeAffinity normalization with the pod's assumed maxima, LeastAllocated,
//                   BalancedAllocation; per (pod, workgroup): the best TWO packed (score, index) keys, feasible count,
//                   the true maxima and their holder counts.  All against the state S0 at the start of the window.
//   k_multi_select  one wave per pod: top-K of the per-workgroup keys (the pod's candidate list, best first).
//   multi_commit_inorder (wave 0 of k_multi_commit_par's launch when windows keep ending early)
//                   ONE wave, pods in order.  Pod j's true argmax over the state S_j (= S0 + the placements of pods
//                   0..j-1 of the window) is max(best UNTOUCHED node, best TOUCHED node): a node no earlier pod of the window
//                   was placed on has the state, feasibility and score the scan saw (the pod's own spread / anti-affinity
//                   state only changes through its OWN clones, and it appears once per window; the normalization maxima
//                   cannot move while an untouched holder remains), so the first untouched entry of the candidate list is
//                   the best untouched node; the <= 64 touched nodes live in the wave's lanes and are re-evaluated for
//                   pod j exactly.  Whenever that argument does not cover a pod (candidate list exhausted, both recorded
//                   keys of one workgroup touched, too few holders of a maximum left, an assumed maximum was wrong) the
//                   window ENDS before that pod and the next window starts with it: never a guess.
//   k_multi_refresh the SCORE MEMO (round 4).  What the scan computes per (pod spec, node) -- static verdict, NodeResourcesFit,
//                   the spec's anti-affinity bit, the weighted score sum under the spec's assumed maxima -- depends on the node's
//                   columns and on the spec alone, and a window changes at most 64 nodes.  With 288 GB of HBM the whole
//                   specs x nodes matrix stays resident (one 32-bit word per pair: 0 = does not fit, else TotalScore + 1;
//                   410 MB for config 5): a scan that finds a spec's row stamped with the maxima it assumes reads the word
//                   instead of recomputing it (~250 -> ~40 VALU instructions per pair), applies the spec's spread filter
//                   (the only part that moves with the spec's OWN placements) and ranks; a scan that does not, computes and
//                   fills the row, and k_multi_select stamps it.  After the commit this kernel recomputes the words of the
//                   window's touched nodes for every stamped spec (<= 64 x P pairs).  Rows are invalidated when a run begins.
// Results are identical to the oracle's round-robin loop (oracle/ccref.c ccref_run_multi; tests/test_multi.py).
#pragma once
#include "ccsim_level.h"

namespace ccsim {

constexpr int kMWindowMax = 128;  // pods per window (round 6: 64 -> 128; the assignment's per-pod rows live in the lanes of the commit kernel's first two waves)
constexpr int kMSeqMax = 64;      // ... of a window the IN-ORDER commit takes (one wave: lane = pod / touched node); MState::seq_windows caps the window
#ifndef CCSIM_MPOD_CHUNK
#define CCSIM_MPOD_CHUNK 2 // (build-time knob for A/B runs; round 5: 8 -> 2, see kMLeanChunk)
#endif
constexpr int kMPodChunk = CCSIM_MPOD_CHUNK; // pods per scan workgroup ...
constexpr int kMPodSub = 2;                  // ... whose per-node words are held in registers at a time
// Round 5: the LEAN form of the scan (every pod of the chunk has a valid memo row: the steady state) is not bound by bytes or by total
// arithmetic but by how many instructions ONE wave executes back to back -- at 8 pods per workgroup ~1500 in the evaluation, 6.5 us at
// the three waves a SIMD holds.  Fewer pods per workgroup, more workgroups: measured (profiles/r05/bench_c5_pod_chunk.txt) 8 pods per
// workgroup 48.0 us per window, 4: 45.3, 2: 44.8.  (A chunk's pods can also be spread over kMLeanPer lean workgroups while the general
// form keeps 8 per workgroup -- one kernel then carries the general form's registers into the lean form's occupancy: 49.9 us.  So the
// chunk itself is 2 and the general form re-reads the node columns per 2 pods: it runs for a spec's first scan and after normalization
// events only, and when the memo is off: 7.3e5 -> see DESIGN 4.3.)
constexpr int kMLeanChunk = kMPodChunk;
constexpr int kMLeanPer = kMPodChunk / kMLeanChunk;
static_assert(kMPodChunk % kMLeanChunk == 0, "lean workgroups per pod chunk");
static_assert(kMPodChunk % kMPodSub == 0, "pods per scan workgroup: a multiple of kMPodSub");
constexpr int kMNodesPerThread = 4;
constexpr int kMBlockNodes = kThreads * kMNodesPerThread; // 1024
constexpr int kMTopK = 8;
constexpr int kMTouched = kMWindowMax; // touched nodes a window can hold (the in-order commit: kMSeqMax, one per lane)
constexpr int kMTsc = 2;          // hard spread constraints per spec
constexpr int kMDomMax = 63;      // value ids 0..62 per spread table

struct MPod { // one pod spec (device array of P)
    int32_t req0, req1, nz0, nz1; // narrow units
    int32_t cls;                  // static class: row of stat_cls / sreason_cls
    int32_t all_zero_req, w_bal, w_aff; // per-pod plugin switches (BalancedAllocation skips best-effort pods, NodeAffinity scores only preferred terms)
    int32_t n_tsc;
    int32_t tsc_slot[kMTsc];      // which of the engine's two spread label columns
    int32_t tsc_max_skew[kMTsc], tsc_min_dom[kMTsc], tsc_self[kMTsc], tsc_ndom[kMTsc], tsc_npresent[kMTsc];
    int32_t tsc_tbl[kMTsc];       // offset of the count table (ndom + 1 int32) in tbl_pool
    int32_t tsc_inc[kMTsc];       // inclusion array id (row of inc_pool), -1 = every node
    int32_t anti;                 // required anti-affinity to its own clones on the one-node-per-domain key
    int32_t mt_a, ma_a;           // normalization maxima assumed by the next scan
    int64_t req_wide[2], nz_wide[2]; // the int64 columns follow the narrow ones at commit
    // PodTopologySpread.Filter per DOMAIN, as the spec's tables stand: bit v = a node with value id v passes constraint c's skew
    // test (bit 0 -- the node lacks the key -- is 0; an unused slot passes everything).  Derived state: k_multi_masks rebuilds
    // it from the tables when a run begins, k_multi_refresh after a placement of the spec (the only thing that moves its tables).
    uint64_t tsc_allow[kMTsc];
};

struct MState {
    int64_t placed, limit, rounds, windows, stops;
    int64_t memo_scans, full_scans; // pods of the windows whose scan read its score memo row / computed (and filled) it
    int64_t scan_prof[4];           // k_multi_scan, workgroup (0, 0): 10 ns ticks in [0] loads issued + staging, [1] the pods' evaluation, [2] merge; [3] scans
    int64_t stop_count[8]; // windows ended by reason (multi_commit_inorder / k_multi_commit_par): diagnostics
    int64_t prof[8];       // multi_commit_inorder: 10 ns ticks in [0] prologue, [1] touched-node evaluation, [2] candidate walk, [3] new touched node, [4] commit, [5] epilogue; [6] new touched nodes, [7] third-key loads
    int32_t done, stop_spec;
    int32_t next_pod;   // spec of the next cycle
    int32_t win_n;      // pods the pending window covers
    int32_t single_pod; // >= 0: ccsim_schedule_pod -- one cycle of that spec
    int32_t last_feasible, last_evaluated;
    int64_t winner;
    int64_t log_cap;
    int32_t seq_windows; // > 0: the in-order commit (multi_commit_inorder) handles the next windows, else the assign + verify one
    int32_t epoch, committed_epoch; // k_multi_select bumps epoch once per window; the commit kernel that takes the window records it
    int32_t n_touched;  // nodes the last committed window placed pods on (MultiArgs::touched): k_multi_refresh's work list
    int32_t placed_first, n_placed; // ... and the specs it placed: placed_first, placed_first + 1, ... (mod P): their spread masks are due
};

struct MPartial { // per (pod of the window, scan workgroup)
    uint64_t key1, key2; // best two nodes of the workgroup's nodes for the pod: ((score+1) << 40) | ~index ; 0 = none
    uint64_t key3;       // the third best: never a candidate, only the BOUND on what the workgroup hides once its two are touched
    uint32_t nfeas, mt, ma; // feasible nodes; the true normalization maxima over them
    uint32_t c_mt, c_ma;    // nodes holding the pod's ASSUMED maxima (== the true holders whenever the assumption stands)
    uint32_t pad;
};

struct MCand { // per pod of the window, after k_multi_select
    uint64_t key[kMTopK];
    uint64_t bound; // the best key NOT in the list (0 = the list holds every feasible node's workgroup-best-two)
    int32_t n, nfeas;
    uint32_t mt, ma, c_mt, c_ma;
};

struct MultiArgs {
    PersistCols c;         // narrow mirrors + the int64 columns the commit keeps in step
    DevPod prof;           // profile constants (weights, resource lists); per-pod fields come from MPod
    MState *st;
    MPod *pods;
    int32_t n_pods;
    const uint32_t *stat_cls;   // [n_cls][n_pad]
    int64_t n_pad;
    const int32_t *tsc_label[kMTsc]; // the engine's (at most) two spread label columns
    int32_t *tbl_pool;
    const uint8_t *present_pool;     // same offsets as tbl_pool: 1 = the domain holds >= 1 counted node
    const uint8_t *inc_pool;         // [n_inc][n_pad]
    uint32_t *anti_bits;             // [n_pods][n_pad / 32]
    MPartial *partials;              // [kMWindowMax][n_blocks]
    int32_t n_blocks;
    MCand *cands;                    // [kMWindowMax]
    int32_t *log;
    int32_t *per_spec;               // [n_pods]
    int32_t window;                  // pods per window (<= kMWindowMax, <= n_pods)
    // the score memo (header comment; nullptr = off: every scan computes)
    // [n_pods][n_pad] 16-bit words (round 5: half the bytes of the scan's dominant stream): bits 0..10 = 0 when the static verdict /
    // NodeResourcesFit / anti-affinity reject the pair, else TotalScore + 1; bits 11..14 = how the node's taint count / affinity sum
    // stand to the maxima the row was computed under (kMemoEqT ...): the lean scan needs no static word
    uint16_t *memo;
    int32_t *memo_stamp;             // [n_pods][2]: the (taint, affinity) maxima the row was computed under; -1 = no row
    int32_t *touched;                // [kMTouched] shard-local indices
    int32_t *vsync;                  // [kMVsyncWords] what k_multi_commit_par's verifying workgroups hand to workgroup 0 (zero between launches)
};

__device__ __forceinline__ uint32_t op_or_u32(uint32_t a, uint32_t b) { return a | b; }
__device__ __forceinline__ uint32_t wave_or_u32(uint32_t v) {
    CCSIM_DPP_STEP32(v, 0u, op_or_u32, 0x111, 0xf) CCSIM_DPP_STEP32(v, 0u, op_or_u32, 0x112, 0xf)
    CCSIM_DPP_STEP32(v, 0u, op_or_u32, 0x114, 0xf) CCSIM_DPP_STEP32(v, 0u, op_or_u32, 0x118, 0xf)
    CCSIM_DPP_STEP32(v, 0u, op_or_u32, 0x142, 0xa) CCSIM_DPP_STEP32(v, 0u, op_or_u32, 0x143, 0xc)
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
constexpr uint32_t kMemoScoreMask = 0x7ffu; // TotalScore + 1 <= 2047 (ccsim_set_pods checks the weights; else the memo stays off)
constexpr uint32_t kMemoEqT = 1u << 11, kMemoGtT = 1u << 12, kMemoEqA = 1u << 13, kMemoGtA = 1u << 14;
__device__ __forceinline__ uint16_t m_memo_word(bool ok, uint32_t total, uint32_t cnt, uint32_t aff, uint32_t mt, uint32_t ma) {
    if (!ok) return 0;
    return (uint16_t)((total + 1u) | (cnt == mt ? kMemoEqT : 0u) | (cnt > mt ? kMemoGtT : 0u) | (aff == ma ? kMemoEqA : 0u) | (aff > ma ? kMemoGtA : 0u));
}
// What a scan that only sees those bits can say about a true maximum over its feasible nodes: the assumed one when a holder is
// among them, else a value that DIFFERS from it -- above or below as the truth is.  The commit ends the window before such a pod
// and takes the value as the next assumption; the scan that follows (the row's stamp no longer matches: the general form) reports
// the exact maximum, and one more zero-progress window later the assumption is exact.  Never a wrong score: every score the
// commit uses was computed under maxima it has checked against the true ones.
__device__ __forceinline__ uint32_t m_max_from_level(uint32_t level, uint32_t assumed, uint32_t nfeas) {
    return level >= 2u ? assumed + 1u : level == 1u ? assumed : (assumed == 0u || nfeas == 0u ? 0u : assumed - 1u);
}

__device__ __forceinline__ uint64_t uni64(uint64_t v) { // a wave-uniform 64-bit value, into scalar registers
    return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
}

// is spec pi's memo row the one a scan under the spec's current assumed maxima would compute?
__device__ __forceinline__ bool m_memo_valid(const MultiArgs &a, int pi) {
    return a.memo && a.memo_stamp[2 * pi] == a.pods[pi].mt_a && a.memo_stamp[2 * pi + 1] == a.pods[pi].ma_a;
}

__device__ __forceinline__ DevPod m_devpod(const DevPod &prof, const MPod &q) {
    DevPod p = prof;
    p.all_zero_req = q.all_zero_req, p.w_bal = q.w_bal, p.w_aff = q.w_aff;
    return p;
}

// PodTopologySpread.Filter for one node (filtering.go:311-356) from the pod's count table; 0 ok, 1 missing label, 2 skew
__device__ __forceinline__ int m_pts_check(const MPod &q, int c, int32_t v, int32_t match, int32_t minm) {
    if (!v) return 1;
    const int64_t mm = q.tsc_npresent[c] < q.tsc_min_dom[c] ? 0 : (int64_t)minm;
    return (int64_t)match + q.tsc_self[c] - mm > (int64_t)q.tsc_max_skew[c] ? 2 : 0;
}

// A staged (LDS) table entry carries its domain's presence: absent domains (no counted node: excluded from the minimum,
// match count 0) hold kMAbsent.  (m_tbl_min over the GLOBAL presence array is a chain of dependent loads: it ran once per
// scan workgroup and once per commit lane and dominated both kernels.)
constexpr int32_t kMAbsent = 0x40000000;
__device__ __forceinline__ int32_t m_stage(int32_t count, uint8_t present) { return present ? count : kMAbsent; }
__device__ __forceinline__ int32_t m_count(int32_t staged) { return staged & (kMAbsent - 1); }
__device__ __forceinline__ int32_t m_staged_min(const int32_t *staged, int ndom) {
    int32_t m = 0x7fffffff;
    for (int v = 1; v <= ndom; v++)
        if (staged[v] < kMAbsent) m = staged[v] < m ? staged[v] : m;
    return m;
}

// MPod::tsc_allow[c] from the constraint's staged table (staged(v) = count, or kMAbsent for a domain without a counted node:
// excluded from the minimum, match count 0); `bump` = the domain whose count is one higher than the table says (the placement
// being applied), 0 = none.  (match + self - min > maxSkew) <=> match > lim; the minimum counts as 0 while fewer than
// minDomains domains exist (filtering.go:56-69, 311-356).
template <class Staged>
__device__ __forceinline__ uint64_t m_allow_mask(const MPod &q, int c, Staged staged, int bump) {
    if (c >= q.n_tsc) return ~0ull;
    const int ndom = q.tsc_ndom[c];
    int32_t mn = 0x7fffffff;
    for (int v = 1; v <= ndom; v++) {
        const int32_t x = staged(v);
        if (x < kMAbsent) { const int32_t y = x + (v == bump ? 1 : 0); mn = y < mn ? y : mn; }
    }
    const int64_t lim = (int64_t)q.tsc_max_skew[c] + (q.tsc_npresent[c] < q.tsc_min_dom[c] ? 0 : (int64_t)mn) - q.tsc_self[c];
    uint64_t mask = ~1ull;
    for (int v = 1; v <= ndom; v++) {
        const int32_t x = staged(v);
        const int64_t y = x < kMAbsent ? (int64_t)x + (v == bump ? 1 : 0) : 0;
        if (y > lim) mask &= ~(1ull << v);
    }
    return mask;
}

// minimum match count over the domains holding a counted node (CriticalPaths[c][0], filtering.go:298-305)
__device__ __forceinline__ int32_t m_tbl_min(const int32_t *tbl, const uint8_t *present, int ndom) {
    int32_t m = 0x7fffffff;
    for (int v = 1; v <= ndom; v++)
        if (present[v]) m = tbl[v] < m ? tbl[v] : m;
    return m;
}

// ------------------------------------------------------------------------------------------------------------------
// multi_scan_lean: k_multi_scan for a workgroup whose pods ALL have a valid memo row (the steady state of a run).  Measured
// on the general form (rocprofv3 + s_memrealtime, profiles/r04/c5_pmc_summary.txt): with the memo the evaluation fell to ~40
// VALU instructions per pair, and the kernel stayed at 48 us -- the pod loop issues a dependent scalar-load -> vector-load
// chain per pod (its words are fetched two pods ahead, after the descriptors' scalar loads) and the waves sit in those round
// trips, not in arithmetic.  Here nothing is fetched pod by pod: the descriptors go to LDS first, then the memo words and
// static words of ALL the chunk's pods (2 x 8 x 4 loads per thread) are issued at once, the spread filter is a bit test against
// the spec's per-domain masks (MPod::tsc_allow: no table, no minimum, no LDS read per pair), and the evaluation runs from registers.
// ------------------------------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------------------------------
// multi_select_pod: ONE WAVE -- the top-K of the scan workgroups' best-two keys for pod j of the window, sorted, + the aggregates.
// ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void multi_select_pod(const MultiArgs &a, const int j, const int next_pod) {
    const int lane = threadIdx.x & 63;
    const MPartial *pp = a.partials + (int64_t)j * a.n_blocks;
    uint64_t best[kMTopK];
#pragma unroll
    for (int k = 0; k < kMTopK; k++) best[k] = 0;
    uint32_t nf = 0, mt = 0, ma = 0, cmt = 0, cma = 0;
    uint64_t dropped = 0; // the best key this lane saw and does not keep (it may still be the wave's (K+1)-th)
    // every lane keeps the sorted top-K of its share, then K rounds of wave max + pop
    for (int b = lane; b < a.n_blocks; b += 64) {
        const MPartial q = pp[b];
        for (int h = 0; h < 2; h++) {
            uint64_t key = h ? q.key2 : q.key1;
#pragma unroll
            for (int k = 0; k < kMTopK; k++)
                if (key > best[k]) { const uint64_t t = best[k]; best[k] = key; key = t; }
            dropped = key > dropped ? key : dropped;
        }
        nf += q.nfeas;
        mt = q.mt > mt ? q.mt : mt, ma = q.ma > ma ? q.ma : ma;
        cmt += q.c_mt, cma += q.c_ma; // holders of the pod's ASSUMED maxima (meaningful iff they are the true ones: checked by the commit)
    }
    const uint32_t wmt = wave_max_u32(mt), wma = wave_max_u32(ma);
    const uint32_t wcmt = wave_sum_u32(cmt), wcma = wave_sum_u32(cma), wnf = wave_sum_u32(nf);
    MCand &out = a.cands[j]; // (the keys go straight to memory: `out.key[n++]` on a local copy was a dynamically indexed array in scratch)
    int n = 0;
#pragma unroll 1
    for (int k = 0; k < kMTopK; k++) {
        const uint64_t m = wave_max_u64(best[0]);
        if (!m) break;
        if (lane == 0) out.key[n] = m;
        n++;
        if (best[0] == m) { // unique keys: one lane pops
#pragma unroll
            for (int x = 0; x + 1 < kMTopK; x++) best[x] = best[x + 1];
            best[kMTopK - 1] = 0;
        }
    }
    if (lane == 0)
        for (int k = n; k < kMTopK; k++) out.key[k] = 0;
    // what the list does not hold: the lanes' remaining (and dropped) candidates; the workgroups' hidden nodes are bounded
    // per workgroup (MPartial::key3) when the commit needs them
    const uint64_t rest = wave_max_u64(best[0] > dropped ? best[0] : dropped);
    if (lane == 0) {
        out.n = n, out.nfeas = (int32_t)wnf, out.mt = wmt, out.ma = wma, out.c_mt = wcmt, out.c_ma = wcma, out.bound = rest;
        if (j == 0) a.st->epoch = a.st->epoch + 1; // this window's candidates exist: exactly one commit kernel may consume them
        const int pi = (next_pod + j) % a.n_pods;
        atomicAdd(reinterpret_cast<unsigned long long *>(m_memo_valid(a, pi) ? &a.st->memo_scans : &a.st->full_scans), 1ull);
        if (a.memo) { // the scan is over: the pod's memo row now holds what a scan under these maxima computes (read or just filled)
            a.memo_stamp[2 * pi] = a.pods[pi].mt_a, a.memo_stamp[2 * pi + 1] = a.pods[pi].ma_a;
        }
    }
}

// The loads of the lean form -- issued by k_multi_scan TOGETHER with the loads that decide which form runs (one round trip instead of two).
struct MLeanLoads {
    uint32_t lv[kMNodesPerThread]; // the thread's four consecutive nodes: value ids of the two spread label columns, one byte each
    uint2 cv[kMLeanChunk];         // per pod: the four nodes' memo words
    int32_t desc;                  // this thread's word of the pods' descriptors (tid < kMLeanChunk * sizeof(MPod) / 4)
};
__device__ __forceinline__ void multi_scan_lean_loads(const MultiArgs &a, const int next_pod, const int j0, const int jn, const int64_t base, MLeanLoads &L) {
    const int tid = threadIdx.x;
    // a thread owns FOUR CONSECUTIVE nodes (local indices 4 tid .. 4 tid + 3): the labels arrive as one 16-byte load per column, a
    // pod's four 16-bit memo words as one 8-byte load (n_pad is a multiple of four: ccsim_set_pods checks)
    const int64_t i4 = base + 4 * tid;
    const bool in4 = i4 < a.c.n_pad;
    // (unconditional 16- / 8-byte loads from an address that is always valid, the result masked: `cond ? *p : zero` on a vector made
    // the compiler select between p and a zero in SCRATCH and load dword by dword)
    const int64_t i4c = in4 ? i4 : 0;
    const bool has0 = a.tsc_label[0] != nullptr, has1 = a.tsc_label[1] != nullptr;
    const int4 l0 = *reinterpret_cast<const int4 *>((has0 ? a.tsc_label[0] : a.c.pod_count) + i4c);
    const int4 l1 = *reinterpret_cast<const int4 *>((has1 ? a.tsc_label[1] : a.c.pod_count) + i4c);
    const uint32_t m0 = in4 && has0 ? (uint32_t)kMDomMax : 0u, m1 = in4 && has1 ? (uint32_t)kMDomMax : 0u;
    L.lv[0] = ((uint32_t)l0.x & m0) | (((uint32_t)l1.x & m1) << 8), L.lv[1] = ((uint32_t)l0.y & m0) | (((uint32_t)l1.y & m1) << 8);
    L.lv[2] = ((uint32_t)l0.z & m0) | (((uint32_t)l1.z & m1) << 8), L.lv[3] = ((uint32_t)l0.w & m0) | (((uint32_t)l1.w & m1) << 8);
    static_assert(kMNodesPerThread == 4, "four consecutive nodes per thread");
#pragma unroll
    for (int jj = 0; jj < kMLeanChunk; jj++) {
        const bool on = jj < jn;
        const int pi = (next_pod + j0 + (on ? jj : 0)) % a.n_pods;
        const uint2 w = *reinterpret_cast<const uint2 *>(a.memo + (int64_t)pi * a.n_pad + i4c);
        const uint32_t mk = on && in4 ? 0xffffffffu : 0u;
        L.cv[jj] = make_uint2(w.x & mk, w.y & mk);
    }
    constexpr int kWords = (int)(sizeof(MPod) / 4);
    static_assert(kMLeanChunk * kWords <= kThreads, "one descriptor word per thread");
    L.desc = 0;
    if (tid < kMLeanChunk * kWords) {
        const int jj = tid / kWords, w = tid % kWords;
        const int pi = (next_pod + j0 + (jj < jn ? jj : 0)) % a.n_pods;
        L.desc = reinterpret_cast<const int32_t *>(&a.pods[pi])[w];
    }
}

__device__ __forceinline__ void multi_scan_lean(const MultiArgs &a, const int next_pod, const int j0, const int jn, const int64_t base,
                                                const unsigned long long sp_t0, const MLeanLoads &L) {
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6); // (wave: uniform, and known to the compiler as such)
    __shared__ MPod l_pod[kMLeanChunk];
    __shared__ uint32_t l_k[kMLeanChunk][3][kThreads / 64];
    __shared__ uint32_t l_u[kMLeanChunk][5][kThreads / 64];
    const auto &lv = L.lv; // (references, not pointers: the struct stays in registers)
    const auto &cv = L.cv;
    {
        constexpr int kWords = (int)(sizeof(MPod) / 4);
        if (tid < kMLeanChunk * kWords) reinterpret_cast<int32_t *>(&l_pod[0])[tid] = L.desc;
    }
    __syncthreads();
    const unsigned long long sp_t1 = __builtin_amdgcn_s_memrealtime();
    // (no branch per pod: a pod beyond the window's end has all-zero words and ranks nothing -- the eight pods' reduction chains are
    // independent and the scheduler interleaves them; with a uniform branch around each pod they ran one after the other)
#pragma unroll
    for (int jj = 0; jj < kMLeanChunk; jj++) {
        {
            const MPod &q = l_pod[jj];
            // the spread filter per domain is the spec's own state (MPod::tsc_allow): two 64-bit masks in scalar registers
            const uint64_t allow0 = uni64(q.tsc_allow[0]), allow1 = uni64(q.tsc_allow[1]);
            const bool sl0 = uni32(q.tsc_slot[0]) != 0, sl1 = uni32(q.tsc_slot[1]) != 0;
            uint32_t k1 = 0, k2 = 0, k3 = 0, lv_bits = 0, acc = 0;
            const uint32_t words[4] = {cv[jj].x & 0xffffu, cv[jj].x >> 16, cv[jj].y & 0xffffu, cv[jj].y >> 16};
#pragma unroll
            for (int k = 0; k < kMNodesPerThread; k++) {
                // Branch-free (round 5): as nested `if`s this loop compiled to three exec-mask branches per pair -- 96 per workgroup pass --
                // and the three waves a SIMD holds here cannot hide their issue bubbles (8.4 us of "evaluation" for ~1500 instructions)
                const uint32_t word = words[k];
                const uint32_t v0 = (sl0 ? lv[k] >> 8 : lv[k]) & (uint32_t)kMDomMax, v1 = (sl1 ? lv[k] >> 8 : lv[k]) & (uint32_t)kMDomMax;
                const uint32_t sc = word & kMemoScoreMask; // TotalScore + 1, 0 = the pair is out
                const uint32_t pass = (uint32_t)((allow0 >> v0) & (allow1 >> v1) & 1ull);
                const uint32_t okm = (sc != 0u ? pass : 0u) ? 0xffffffffu : 0u; // all ones: the node is feasible for the pod today
                uint32_t key = (((sc << 10) | (1023u - (uint32_t)(4 * tid + k))) & okm); // | the lower index wins a tie
                // insert into the sorted triple (keys of feasible nodes are unique and non-zero)
                uint32_t t = key > k1 ? key : k1; key = key > k1 ? k1 : key; k1 = t;
                t = key > k2 ? key : k2; key = key > k2 ? k2 : key; k2 = t;
                k3 = key > k3 ? key : k3;
                // (the word's four flag bits as they are: OR over the feasible nodes, one reduction for both maxima)
                const uint32_t wm = word & okm;
                lv_bits |= wm;
                acc += (okm & 1u) + ((uint32_t)((wm & (kMemoEqT | kMemoGtT)) == kMemoEqT) << 10) + ((uint32_t)((wm & (kMemoEqA | kMemoGtA)) == kMemoEqA) << 20);
            }
            const uint32_t w1 = wave_max_u32(k1);
            const bool h1 = k1 == w1 && w1 != 0;
            const uint32_t x2 = h1 ? k2 : k1, y2 = h1 ? k3 : k2;
            const uint32_t w2 = wave_max_u32(x2);
            const bool h2 = x2 == w2 && w2 != 0;
            const uint32_t w3 = wave_max_u32(h2 ? y2 : x2);
            const uint32_t wbits = wave_or_u32(lv_bits);
            const uint32_t wlt = (wbits & kMemoGtT) ? 2u : (wbits & kMemoEqT) ? 1u : 0u, wla = (wbits & kMemoGtA) ? 2u : (wbits & kMemoEqA) ? 1u : 0u;
            const uint32_t packed = wave_sum_u32(acc);
            if (lane == 0 && jj < jn)
                l_k[jj][0][wave] = w1, l_k[jj][1][wave] = w2, l_k[jj][2][wave] = w3, l_u[jj][0][wave] = packed & 1023u, l_u[jj][1][wave] = wlt,
                l_u[jj][2][wave] = wla, l_u[jj][3][wave] = (packed >> 10) & 1023u, l_u[jj][4][wave] = packed >> 20;
        }
    }
    const unsigned long long sp_t2 = __builtin_amdgcn_s_memrealtime();
    __syncthreads();
    if (tid < jn) {
        const int jj = tid;
        uint32_t b1 = 0, b2 = 0, b3 = 0, lt = 0, la = 0;
        MPartial o{};
        for (int x = 0; x < kThreads / 64; x++) {
            for (int h = 0; h < 3; h++) {
                const uint32_t key = l_k[jj][h][x];
                if (key > b1) b3 = b2, b2 = b1, b1 = key; else if (key > b2) b3 = b2, b2 = key; else if (key > b3) b3 = key;
            }
            o.nfeas += l_u[jj][0][x];
            lt = l_u[jj][1][x] > lt ? l_u[jj][1][x] : lt, la = l_u[jj][2][x] > la ? l_u[jj][2][x] : la;
            o.c_mt += l_u[jj][3][x], o.c_ma += l_u[jj][4][x];
        }
        o.mt = m_max_from_level(lt, (uint32_t)l_pod[jj].mt_a, o.nfeas), o.ma = m_max_from_level(la, (uint32_t)l_pod[jj].ma_a, o.nfeas);
        auto widen = [&](uint32_t key) -> uint64_t {
            return key ? make_key((int64_t)(key >> 10) - 1, a.c.global_offset + base + (int64_t)(1023u - (key & 1023u))) : 0ull;
        };
        o.key1 = widen(b1), o.key2 = widen(b2), o.key3 = widen(b3);
        a.partials[(int64_t)(j0 + jj) * a.n_blocks + blockIdx.x] = o;
    }
    if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) {
        const unsigned long long sp_t3 = __builtin_amdgcn_s_memrealtime();
        a.st->scan_prof[0] += (int64_t)(sp_t1 - sp_t0), a.st->scan_prof[1] += (int64_t)(sp_t2 - sp_t1), a.st->scan_prof[2] += (int64_t)(sp_t3 - sp_t2), a.st->scan_prof[3] += 1;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// k_multi_scan: grid (node workgroups, pod chunks).
// Measured (rocprofv3, C5 100k x 1024, profiles/r02/c5_kernel_stats.csv): the first form of this kernel -- 4 pods per
// workgroup; tables staged, THEN node columns loaded, THEN the pods' per-node words, then per pod a descriptor load and a
// two-barrier reduction -- took 65 us per window: a chain of dependent global round trips per workgroup at 4 waves per
// SIMD, not arithmetic (6.4 M evaluations are ~10 us of VALU work).  This form keeps every load in flight early: node
// columns and the first pods' words are issued before the staging barriers, the pod descriptors come from LDS, the words of
// pod j + kMPodSub are fetched into the registers pod j just released, and the per-pod reductions meet at ONE barrier.
// ------------------------------------------------------------------------------------------------------------------
#ifndef CCSIM_MSCAN_OCC
#define CCSIM_MSCAN_OCC 4 // (build-time knob for A/B runs: waves per SIMD the register allocation must leave room for)
#endif
__global__ __launch_bounds__(kThreads, CCSIM_MSCAN_OCC) void k_multi_scan(MultiArgs a) {
    const int32_t done = a.st->done, win_n = a.st->win_n, next_pod = a.st->next_pod; // (not the whole MState: it would sit in ~60 SGPRs)
    if (done) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6); // (wave: uniform, and known to the compiler as such)
    const int sub = blockIdx.y % kMLeanPer; // this workgroup's share of the chunk in the lean form (kMLeanChunk)
    const int j0 = (blockIdx.y / kMLeanPer) * kMPodChunk;
    if (j0 >= win_n) return;
    const unsigned long long sp_t0 = __builtin_amdgcn_s_memrealtime();
    const int jn = win_n - j0 < kMPodChunk ? win_n - j0 : kMPodChunk;
    const int64_t base = (int64_t)blockIdx.x * kMBlockNodes;
    if (a.memo) { // every pod of the chunk with a valid memo row (lane jj looks at pod jj): the lean form
        const bool on = lane < jn;
        const int pi = (next_pod + j0 + (on ? lane : 0)) % a.n_pods;
        // ONE round trip: the words that decide the form, and with them everything the lean form loads (wasted when the general form
        // runs: each spec's first scan and the scans after a normalization event)
        const int32_t st0 = a.memo_stamp[2 * pi], st1 = a.memo_stamp[2 * pi + 1], as0 = a.pods[pi].mt_a, as1 = a.pods[pi].ma_a;
        const int lj0 = j0 + sub * kMLeanChunk, ljn = jn - sub * kMLeanChunk < kMLeanChunk ? jn - sub * kMLeanChunk : kMLeanChunk;
        MLeanLoads L;
        multi_scan_lean_loads(a, next_pod, lj0, ljn > 0 ? ljn : 0, base, L);
        const bool valid = st0 == as0 && st1 == as1;
        if (__ballot(on && !valid) == 0) {
            if (ljn > 0) multi_scan_lean(a, next_pod, lj0, ljn, base, sp_t0, L);
            return;
        }
    }
    if (sub != 0) return; // the general form: the chunk's first workgroup does all of its pods

    __shared__ MPod s_pod[kMPodChunk];
    __shared__ int32_t s_tbl[kMPodChunk][kMTsc][kMDomMax + 1];
    __shared__ int32_t s_min[kMPodChunk][kMTsc];
    __shared__ uint32_t s_k[kMPodChunk][3][kThreads / 64];
    __shared__ uint32_t s_u[kMPodChunk][5][kThreads / 64];
    __shared__ int32_t s_memo[kMPodChunk]; // 1 = pod jj's memo row is valid

    // (1) this thread's nodes: narrow columns -> registers, once for all pods of the chunk (independent of everything below)
    int32_t a0[kMNodesPerThread], a1[kMNodesPerThread], r0[kMNodesPerThread], r1[kMNodesPerThread], z0[kMNodesPerThread], z1[kMNodesPerThread];
    int32_t room[kMNodesPerThread]; // 1 = the node still has room for one more pod (fit.go:567-576: the same test for every pod)
    uint32_t lv[kMNodesPerThread]; // the node's value ids of the two spread label columns, one byte each (ids <= kMDomMax - 1)
#pragma unroll
    for (int k = 0; k < kMNodesPerThread; k++) {
        const int64_t i = base + k * kThreads + tid;
        const bool in = i < a.c.n_pad;
        a0[k] = in ? a.c.a32[0][i] : 0, a1[k] = in ? a.c.a32[1][i] : 0;
        r0[k] = in ? a.c.r32[0][i] : 0, r1[k] = in ? a.c.r32[1][i] : 0;
        z0[k] = in ? a.c.z32[0][i] : 0, z1[k] = in ? a.c.z32[1][i] : 0;
        room[k] = in && (int64_t)a.c.pod_count[i] + 1 <= (int64_t)a.c.alloc_pods[i] ? 1 : 0;
        const uint32_t l0 = in && a.tsc_label[0] ? (uint32_t)a.tsc_label[0][i] : 0u, l1 = in && a.tsc_label[1] ? (uint32_t)a.tsc_label[1][i] : 0u;
        lv[k] = (l0 & (uint32_t)kMDomMax) | ((l1 & (uint32_t)kMDomMax) << 8);
    }

    // (2) the per-(pod, node) words -- static word of the pod's class, the pod's anti-affinity bits -- of the first kMPodSub
    // pods; slot jj % kMPodSub is refilled with pod jj + kMPodSub's words as soon as pod jj has been evaluated
    uint32_t wv[kMPodSub][kMNodesPerThread], bv[kMPodSub][kMNodesPerThread];
    auto fetch_words = [&](int slot, int jj) {
        const bool on = jj < jn;
        const int pi = (next_pod + j0 + (on ? jj : 0)) % a.n_pods;
        const int32_t cls = a.pods[pi].cls, anti = a.pods[pi].anti; // (uniform address: scalar loads)
        const uint32_t *stat = a.stat_cls + (int64_t)cls * a.n_pad;
        const uint32_t *bits = a.anti_bits + (int64_t)pi * (a.n_pad / 32);
        const bool memo = m_memo_valid(a, pi); // (nothing writes stamps or maxima while a scan runs: the same answer in the evaluation below)
        const uint16_t *row = a.memo + (int64_t)pi * a.n_pad;
#pragma unroll
        for (int k = 0; k < kMNodesPerThread; k++) {
            const int64_t i = base + k * kThreads + tid;
            const bool in = on && i < a.c.n_pad;
            wv[slot][k] = in ? stat[i] : 0u;
            bv[slot][k] = memo ? (in ? (uint32_t)row[i] & kMemoScoreMask : 0u) : (in && anti ? bits[i >> 5] : 0u); // the memo word takes the anti-affinity word's register
        }
    };
#pragma unroll
    for (int jj = 0; jj < kMPodSub; jj++) fetch_words(jj, jj);

    // (3) the chunk's pod descriptors -> LDS, then their spread tables; one lane per (pod, constraint) derives the minimum
    // over present domains
    {
        constexpr int kWords = (int)(sizeof(MPod) / 4);
        int32_t *dst = reinterpret_cast<int32_t *>(&s_pod[0]);
        for (int i = tid; i < kMPodChunk * kWords; i += kThreads) {
            const int jj = i / kWords, w = i % kWords;
            const int pi = (next_pod + j0 + (jj < jn ? jj : 0)) % a.n_pods;
            dst[i] = reinterpret_cast<const int32_t *>(&a.pods[pi])[w];
        }
        if (tid < kMPodChunk) s_memo[tid] = tid < jn && m_memo_valid(a, (next_pod + j0 + tid) % a.n_pods) ? 1 : 0;
    }
    __syncthreads();
    for (int i = tid; i < kMPodChunk * kMTsc * (kMDomMax + 1); i += kThreads) {
        const int jj = i / (kMTsc * (kMDomMax + 1)), c = (i / (kMDomMax + 1)) % kMTsc, v = i % (kMDomMax + 1);
        int32_t x = 0;
        if (jj < jn) {
            const MPod &q = s_pod[jj];
            if (c < q.n_tsc && v <= q.tsc_ndom[c]) x = m_stage(a.tbl_pool[q.tsc_tbl[c] + v], a.present_pool[q.tsc_tbl[c] + v]);
            if (c < q.n_tsc && v == 0) x = kMAbsent - 1; // value id 0 = the node lacks the topology key (filtering.go:328-332): above every limit
        }
        s_tbl[jj][c][v] = x;
    }
    __syncthreads();
    if (tid < kMPodChunk * kMTsc) {
        const int jj = tid / kMTsc, c = tid % kMTsc;
        int32_t m = 0x7fffffff;
        if (jj < jn) {
            const MPod &q = s_pod[jj];
            if (c < q.n_tsc) m = m_staged_min(&s_tbl[jj][c][0], q.tsc_ndom[c]);
        }
        s_min[jj][c] = m;
    }
    __syncthreads();

    const unsigned long long sp_t1 = __builtin_amdgcn_s_memrealtime();
    // Inside a workgroup a node is its 10-bit local index and a TotalScore fits 21 bits (checked by ccsim_set_pods), so the
    // running top three are 32-bit keys ((score + 1) << 10 | 1023 - local index: same order as the global 64-bit keys);
    // thread jj widens pod jj's three survivors at the end.
    static_assert(kMBlockNodes == 1024, "local node index: 10 bits");
#pragma unroll 1
    for (int sub = 0; sub < kMPodChunk; sub += kMPodSub) { // (a runtime loop: the fully unrolled chunk did not fit the instruction cache)
        if (sub >= jn) break;
#pragma unroll
        for (int slot = 0; slot < kMPodSub; slot++) {
            const int jj = sub + slot;
            if (jj >= jn) break;
            const MPod &q = s_pod[jj];
            DevPod p = a.prof;
            p.all_zero_req = uni32(q.all_zero_req), p.w_bal = uni32(q.w_bal), p.w_aff = uni32(q.w_aff);
            const NarrowPod nq{uni32(q.req0), uni32(q.req1), uni32(q.nz0), uni32(q.nz1)}; // (uniform values: kept in SGPRs)
            // everything that depends on the pod alone, once per pod: the normalization divisors' magics, and per spread
            // constraint the largest domain count the skew test lets through -- (match + self - min > maxSkew) <=> match > lim
            // (filtering.go:311-356; min counts as 0 while fewer than minDomains domains exist, :56-69); an unused slot passes all
            const uint32_t mt = (uint32_t)uni32(q.mt_a), ma = (uint32_t)uni32(q.ma_a);
            const uint32_t Mt = (uint32_t)uni32((int)div_magic(mt)), Ma = (uint32_t)uni32((int)div_magic(ma));
            const int32_t n_tsc = uni32(q.n_tsc);
            int32_t lim[kMTsc];
            bool sl[kMTsc];
#pragma unroll
            for (int c = 0; c < kMTsc; c++) {
                const int32_t mm = q.tsc_npresent[c] < q.tsc_min_dom[c] ? 0 : s_min[jj][c];
                lim[c] = uni32(c < n_tsc ? q.tsc_max_skew[c] + mm - q.tsc_self[c] : 0x7fffffff);
                sl[c] = uni32(q.tsc_slot[c]) != 0;
            }
            const uint32_t my_bit = (uint32_t)(tid & 31); // (the workgroup's first node and k * kThreads are multiples of 32)
            const bool memo = uni32(s_memo[jj]) != 0;
            uint16_t *row = a.memo ? a.memo + (int64_t)((next_pod + j0 + jj) % a.n_pods) * a.n_pad : nullptr;
            uint32_t k1 = 0, k2 = 0, k3 = 0;
            uint32_t mtb = 0, mab = 0;
            uint32_t acc = 0; // three 10-bit counters: feasible nodes | holders of the assumed taint maximum << 10 | of the affinity one << 20
#pragma unroll
            for (int k = 0; k < kMNodesPerThread; k++) {
                const uint32_t w = wv[slot][k];
                const uint32_t cnt = (w >> kStatCntShift) & kStatCntMask, aff = w & kStatAffMask, img = (w >> kStatImgShift) & kStatImgMask;
                bool ok;
                uint32_t total;
                if (memo) { // the pair's word: everything but the spread filter
                    ok = bv[slot][k] != 0u;
                    total = bv[slot][k] - 1u;
                } else {
                    ok = (w >> kStatOkBit) && fits_narrow(p, nq, a0[k], a1[k], r0[k], r1[k], room[k], 0);
                    ok = ok && !((bv[slot][k] >> my_bit) & 1u); // satisfyPodAntiAffinity / existing pods' anti-affinity (filtering.go:352-379)
                    total = 0;
                    if (ok || row) // (a memo row is filled for every pair that fits, whatever the spread filter says today)
                        total = ok ? (uint32_t)(static_score(p, cnt, aff, img, mt, ma, Mt, Ma) + dynamic_score_narrow(p, nq, a0[k], a1[k], r0[k], r1[k], z0[k], z1[k])) : 0u;
                    const int64_t i = base + k * kThreads + tid;
                    if (row && i < a.c.n_pad) row[i] = m_memo_word(ok, total, cnt, aff, mt, ma);
                }
#pragma unroll
                for (int c = 0; c < kMTsc; c++) { // PodTopologySpread.Filter: one LDS read and one compare per constraint
                    const uint32_t v = (sl[c] ? lv[k] >> 8 : lv[k]) & (uint32_t)kMDomMax; // (ids are validated <= kMDomMax - 1)
                    ok = ok && m_count(s_tbl[jj][c][v]) <= lim[c];
                }
                if (!ok) continue;
                const uint32_t key = ((total + 1u) << 10) | (1023u - (uint32_t)(k * kThreads + tid));
                if (key > k1) k3 = k2, k2 = k1, k1 = key; else if (key > k2) k3 = k2, k2 = key; else if (key > k3) k3 = key;
                // the true maxima over the feasible set, and how many nodes hold the ASSUMED ones (only read when the two agree)
                mtb = cnt > mtb ? cnt : mtb, mab = aff > mab ? aff : mab;
                acc += 1u | ((uint32_t)(cnt == mt) << 10) | ((uint32_t)(aff == ma) << 20);
            }
            if (sub + kMPodSub < kMPodChunk) fetch_words(slot, jj + kMPodSub); // the slot is free: pod jj + kMPodSub's words, consumed kMPodSub pods later
            // wave top-3 (keys are unique: exactly one lane holds a wave maximum): best, second, third over the wave
            const uint32_t w1 = wave_max_u32(k1);
            const bool h1 = k1 == w1 && w1 != 0;
            const uint32_t x2 = h1 ? k2 : k1, y2 = h1 ? k3 : k2; // this lane's best two once the wave's best is removed
            const uint32_t w2 = wave_max_u32(x2);
            const bool h2 = x2 == w2 && w2 != 0;
            const uint32_t w3 = wave_max_u32(h2 ? y2 : x2);
            const uint32_t wmt = wave_max_u32(mtb), wma = wave_max_u32(mab);
            // (counts per wave <= 256: three 10-bit fields in one reduction)
            const uint32_t packed = wave_sum_u32(acc);
            const uint32_t wnf = packed & 1023u, wcmt = (packed >> 10) & 1023u, wcma = packed >> 20;
            if (lane == 0)
                s_k[jj][0][wave] = w1, s_k[jj][1][wave] = w2, s_k[jj][2][wave] = w3, s_u[jj][0][wave] = wnf, s_u[jj][1][wave] = wmt,
                s_u[jj][2][wave] = wma, s_u[jj][3][wave] = wcmt, s_u[jj][4][wave] = wcma;
        }
    }
    const unsigned long long sp_t2 = __builtin_amdgcn_s_memrealtime();
    __syncthreads();
    if (tid < jn) { // thread jj merges pod jj's four wave results and widens the keys
        const int jj = tid;
        uint32_t b1 = 0, b2 = 0, b3 = 0;
        MPartial o{};
        for (int x = 0; x < kThreads / 64; x++) {
            for (int h = 0; h < 3; h++) {
                const uint32_t key = s_k[jj][h][x];
                if (key > b1) b3 = b2, b2 = b1, b1 = key; else if (key > b2) b3 = b2, b2 = key; else if (key > b3) b3 = key;
            }
            o.nfeas += s_u[jj][0][x];
            o.mt = s_u[jj][1][x] > o.mt ? s_u[jj][1][x] : o.mt, o.ma = s_u[jj][2][x] > o.ma ? s_u[jj][2][x] : o.ma;
            o.c_mt += s_u[jj][3][x], o.c_ma += s_u[jj][4][x]; // holders of the ASSUMED maxima
        }
        auto widen = [&](uint32_t key) -> uint64_t {
            return key ? make_key((int64_t)(key >> 10) - 1, a.c.global_offset + base + (int64_t)(1023u - (key & 1023u))) : 0ull;
        };
        o.key1 = widen(b1), o.key2 = widen(b2), o.key3 = widen(b3);
        a.partials[(int64_t)(j0 + jj) * a.n_blocks + blockIdx.x] = o;
    }
    if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) { // (one workgroup's view: measurement aid, ccsim_debug_multi_memo)
        const unsigned long long sp_t3 = __builtin_amdgcn_s_memrealtime();
        a.st->scan_prof[0] += (int64_t)(sp_t1 - sp_t0), a.st->scan_prof[1] += (int64_t)(sp_t2 - sp_t1), a.st->scan_prof[2] += (int64_t)(sp_t3 - sp_t2), a.st->scan_prof[3] += 1;
    }
}

constexpr int kMParThreads = 1024;
// Round 6: the verification is VALU time on ONE CU (a wave64 instruction takes four cycles on a 16-lane SIMD: ~2.2 us per 1024 pairs,
// 19.7 us for the 8128 pairs of a 128-pod window) -- so kMParGroups workgroups are launched.  Every one of them runs the assignment
// (the same inputs, the same result: nothing it reads is written before the others are done), verifies its share of the pairs and hands
// its findings to workgroup 0 through MultiArgs::vsync; workgroup 0 waits for them, then applies.
constexpr int kMParGroups = 8;
constexpr int kMVsMt = kMParGroups, kMVsMa = kMVsMt + kMWindowMax, kMVsHdr = kMVsMa + kMWindowMax, kMVsyncWords = kMVsHdr + 4; // (words 0 .. kMParGroups - 1: one per workgroup)
// kMVsHdr: (the window is the parallel commit's, its pods, its first pod) as k_multi_select saw them.  Workgroup 0 is the only reader and
// writer of MState in the commit's launch; the others must not look at it -- workgroup 0 rewrites it at the end (a struct store: not
// atomic), and a workgroup dispatched late would act on a torn or a new state, find itself a share and leave a stale arrival word behind.
constexpr int kMVsSpinLimit = 1 << 24; // (polls of the arrival word: seconds; the workgroups are co-resident -- 8 of them on 256 CUs)

// ------------------------------------------------------------------------------------------------------------------
// k_multi_select: one wave per pod of the window (multi_select_pod).  Round 5 tried the selection in the scan's TAIL (the last workgroup
// of a pod chunk to deliver its partials selects for the chunk: a ticket per chunk): with a release fence per workgroup the scan became a
// 279 us kernel (3 136 L2 write-backs), with agent-scope word stores and a relaxed ticket 46 us -- 98 same-address device-scope atomics
// per chunk from eight XCDs serialize at ~0.3 us each.  A launch of its own costs 5.8 us: it stays (profiles/r05/bench_c5_pod_chunk.txt).
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_multi_select(MultiArgs a) {
    const int32_t done = a.st->done, win_n = a.st->win_n, next_pod = a.st->next_pod;
    if (blockIdx.x == 0 && threadIdx.x == 0) { // what k_multi_commit_par's workgroups 1 .. 7 go by (kMVsHdr): nothing in that launch writes these words
        a.vsync[kMVsHdr + 0] = !done && a.st->seq_windows <= 0 ? 1 : 0; // the window is the parallel commit's
        a.vsync[kMVsHdr + 1] = win_n, a.vsync[kMVsHdr + 2] = next_pod;
    }
    if (done || (int)blockIdx.x >= win_n) return;
    multi_select_pod(a, (int)blockIdx.x, next_pod);
}

// ------------------------------------------------------------------------------------------------------------------
// multi_commit_inorder: one wave; pods of the window in order (see the header comment).  Runs as wave 0 of
// k_multi_commit_par's launch when MState::seq_windows says so (a launch of its own cost ~5 us per window for a kernel
// that returns at once in all but a few windows): the other waves have left, the barriers below are wave 0's alone.
// ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void multi_commit_inorder(const MultiArgs &a) {
    MState st = *a.st;
    if (st.done || st.seq_windows <= 0 || st.committed_epoch == st.epoch) return; // (k_multi_commit_par had this window)
    st.seq_windows -= 1, st.committed_epoch = st.epoch;
    const int lane = threadIdx.x;
    const int W = st.win_n < kMSeqMax ? st.win_n : kMSeqMax; // (the commit that set seq_windows capped the window; a run that starts in this mode -- CCSIM_MULTI_SEQ -- is capped here)
    __shared__ MPod s_pod[kMSeqMax];
    __shared__ MCand s_cd[kMSeqMax];
    __shared__ int32_t s_tbl[kMSeqMax][kMTsc][kMDomMax + 1]; // the pods' spread tables (their own clones only: fixed until their turn)
    __shared__ int32_t s_min[kMSeqMax][kMTsc];
    __shared__ uint32_t s_w[kMSeqMax][kMSeqMax];  // static word of touched node t for pod j
    __shared__ uint8_t s_f[kMSeqMax][kMSeqMax];   // bit0 anti-affinity hit, bit 1+c counted for spread constraint c

    unsigned long long pf0 = 0, pf1 = 0, pf2 = 0, pf3 = 0, pf4 = 0, pf5 = 0, pf6 = 0, pf7 = 0, t_prev = __builtin_amdgcn_s_memrealtime();
#define MTICK(v) do { const unsigned long long t_now = __builtin_amdgcn_s_memrealtime(); v += t_now - t_prev; t_prev = t_now; } while (0)
    // everything the in-order loop needs about the window's pods -> LDS, all loads in flight together
    if (lane < W) s_pod[lane] = a.pods[(st.next_pod + lane) % a.n_pods], s_cd[lane] = a.cands[lane];
    __syncthreads();
    for (int i = lane; i < W * kMTsc * (kMDomMax + 1); i += 64) {
        const int j = i / (kMTsc * (kMDomMax + 1)), c = (i / (kMDomMax + 1)) % kMTsc, v = i % (kMDomMax + 1);
        const MPod &q = s_pod[j];
        s_tbl[j][c][v] = (c < q.n_tsc && v <= q.tsc_ndom[c]) ? m_stage(a.tbl_pool[q.tsc_tbl[c] + v], a.present_pool[q.tsc_tbl[c] + v]) : 0;
    }
    __syncthreads();
    if (lane < W) {
        const MPod &q = s_pod[lane];
        for (int c = 0; c < kMTsc; c++)
            s_min[lane][c] = c < q.n_tsc ? m_staged_min(&s_tbl[lane][c][0], q.tsc_ndom[c]) : 0;
    }
    __syncthreads();

    MTICK(pf0);
    // lane t: touched node t
    int64_t t_idx = -1; // shard-local index
    int32_t ta0 = 0, ta1 = 0, tr0 = 0, tr1 = 0, tz0 = 0, tz1 = 0, tap = 0, tnp = 0, tl0 = 0, tl1 = 0, tplaced = 0;
    int nt = 0;
    int committed = 0;
    int stop_reason = 0; // 0 window complete
    int64_t my_node = -1; // lane j: where pod j of the window went
    uint32_t my_f = 0;
    int32_t my_l0 = 0, my_l1 = 0;
    DevPod p = a.prof;   // profile constants once; the three per-pod switches are set per pod (a per-pod COPY lived in scratch)

#pragma unroll 1
    for (int j = 0; j < W; j++) {
        const int pi = (st.next_pod + j) % a.n_pods;
        const MPod &q = s_pod[j];
        const MCand &cd = s_cd[j];
        // an assumed normalization maximum was wrong: the pod's scores are invalid -- fix it, end the window here
        if ((int32_t)cd.mt != q.mt_a || (int32_t)cd.ma != q.ma_a) {
            // (the later pods of the window get their maxima fixed too, so that one window repairs a whole cycle of specs)
            for (int jj = j + lane; jj < W; jj += 64) {
                const int pj = (st.next_pod + jj) % a.n_pods;
                a.pods[pj].mt_a = (int32_t)s_cd[jj].mt, a.pods[pj].ma_a = (int32_t)s_cd[jj].ma;
            }
            stop_reason = 1;
            break;
        }
        if (cd.nfeas == 0) { // schedule_one.go:448-454: no node fits -- touched nodes only lost room, the pod's own state did not move
            st.done = DONE_UNSCHEDULABLE, st.stop_spec = pi, st.rounds += 1, st.last_feasible = 0, st.winner = -1;
            stop_reason = 2;
            break;
        }
        // a normalization maximum can only move if EVERY feasible holder lost its room, and only touched nodes changed: as
        // long as the scan counted more holders than there are touched nodes holding the maximum, an untouched one remains
        {
            const uint32_t w = lane < nt ? s_w[lane][j] : 0u;
            const int th_mt = __popcll(__ballot(lane < nt && ((w >> kStatCntShift) & kStatCntMask) == cd.mt));
            const int th_ma = __popcll(__ballot(lane < nt && (w & kStatAffMask) == cd.ma));
            if ((cd.mt > 0 && (int)cd.c_mt <= th_mt) || (cd.ma > 0 && q.w_aff && (int)cd.c_ma <= th_ma)) {
                stop_reason = 3;
                break;
            }
        }
        p.all_zero_req = q.all_zero_req, p.w_bal = q.w_bal, p.w_aff = q.w_aff;
        const NarrowPod nq{q.req0, q.req1, q.nz0, q.nz1};
        // touched nodes, re-evaluated for this pod in their current state
        uint64_t tkey = 0;
        if (lane < nt) {
            const uint32_t w = s_w[lane][j];
            const uint32_t f = s_f[lane][j];
            bool ok = (w >> kStatOkBit) && fits_narrow(p, nq, ta0, ta1, tr0, tr1, tap, tnp) && !(f & 1u);
#pragma unroll
            for (int c = 0; c < kMTsc; c++)
                if (c < q.n_tsc && ok) {
                    const int32_t v = q.tsc_slot[c] ? tl1 : tl0;
                    ok = m_pts_check(q, c, v, m_count(s_tbl[j][c][v < 0 || v > kMDomMax ? 0 : v]), s_min[j][c]) == 0;
                }
            if (ok) {
                const uint32_t cnt = (w >> kStatCntShift) & kStatCntMask, aff = w & kStatAffMask, img = (w >> kStatImgShift) & kStatImgMask;
                const int64_t total = static_score(p, cnt, aff, img, cd.mt, cd.ma) + dynamic_score_narrow(p, nq, ta0, ta1, tr0, tr1, tz0, tz1);
                tkey = make_key(total, a.c.global_offset + t_idx);
            }
        }
        tkey = wave_max_u64(tkey);
        MTICK(pf1);
        // The best UNTOUCHED node: the first untouched entry of the candidate list.  What the list cannot show is BOUNDED:
        // a scan workgroup whose best two are both touched hides nodes no better than its third key; a list that ran out
        // hides nothing better than the best key left out of it.  The winner must beat every such bound, else the window ends.
        uint64_t ukey = 0, bound = 0;
        {
            int64_t sb0 = -1, sb1 = -1, sb2 = -1; // workgroups of the skipped (touched) entries, and how many each
            int sc0 = 0, sc1 = 0, sc2 = 0;
            bool overflow = false;
            int k = 0;
            for (; k < cd.n; k++) {
                const int64_t li = key_index(cd.key[k]) - a.c.global_offset;
                const bool touched = __ballot(lane < nt && t_idx == li) != 0;
                if (!touched) {
                    ukey = cd.key[k];
                    break;
                }
                const int64_t blk = li / kMBlockNodes;
                int cnt2;
                if (blk == sb0) cnt2 = ++sc0;
                else if (blk == sb1) cnt2 = ++sc1;
                else if (blk == sb2) cnt2 = ++sc2;
                else if (sb0 < 0) sb0 = blk, cnt2 = sc0 = 1;
                else if (sb1 < 0) sb1 = blk, cnt2 = sc1 = 1;
                else if (sb2 < 0) sb2 = blk, cnt2 = sc2 = 1;
                else { overflow = true; break; }
                if (cnt2 == 2) { // both recorded nodes of that workgroup are touched: its hidden nodes are bounded by its third key
                    const uint64_t k3 = a.partials[(int64_t)j * a.n_blocks + blk].key3;
                    bound = k3 > bound ? k3 : bound;
                    pf7++;
                }
            }
            if (overflow) bound = ~0ull;
            else if (!ukey && cd.bound > bound) bound = cd.bound; // the list ran out
        }
        MTICK(pf2);
        const uint64_t win = tkey > ukey ? tkey : ukey;
        if (bound && win < bound) { // (keys are unique: win == bound cannot happen)
            stop_reason = 4;
            break;
        }
        if (!win) { // every feasible node of the scan is touched and none of them fits any more: let the next scan say so
            stop_reason = 5;
            break;
        }
        const int64_t g = key_index(win), li = g - a.c.global_offset;
        int slot = __ffsll((unsigned long long)__ballot(lane < nt && t_idx == li)) - 1;
        if (slot < 0) { // first placement on this node in the window: it becomes a touched node
            if (nt >= kMSeqMax) {
                stop_reason = 6;
                break;
            }
            slot = nt++;
            // lane l < W gathers the node's static word / anti-affinity bit / inclusion bits for pod l of the window
            if (lane < W) {
                const int pl = (st.next_pod + lane) % a.n_pods;
                const MPod &ql = s_pod[lane];
                s_w[slot][lane] = a.stat_cls[(int64_t)ql.cls * a.n_pad + li];
                uint32_t f = 0;
                if (ql.anti) f |= (a.anti_bits[(int64_t)pl * (a.n_pad / 32) + (li >> 5)] >> (li & 31)) & 1u;
                bool all = true; // counted iff the node has ALL the pod's hard keys and passes the inclusion policies (filtering.go:267-277)
                for (int c2 = 0; c2 < ql.n_tsc; c2++) all = all && (ql.tsc_slot[c2] ? a.tsc_label[1][li] : a.tsc_label[0][li]) != 0;
                for (int c = 0; c < ql.n_tsc; c++) {
                    const bool inc = ql.tsc_inc[c] < 0 || a.inc_pool[(int64_t)ql.tsc_inc[c] * a.n_pad + li] != 0;
                    if (all && inc) f |= 2u << c;
                }
                s_f[slot][lane] = (uint8_t)f;
            }
            if (lane == slot) {
                t_idx = li;
                ta0 = a.c.a32[0][li], ta1 = a.c.a32[1][li], tr0 = a.c.r32[0][li], tr1 = a.c.r32[1][li];
                tz0 = a.c.z32[0][li], tz1 = a.c.z32[1][li], tap = a.c.alloc_pods[li], tnp = a.c.pod_count[li];
                tl0 = a.tsc_label[0] ? a.tsc_label[0][li] : 0, tl1 = a.tsc_label[1] ? a.tsc_label[1][li] : 0;
                tplaced = 0;
            }
            __syncthreads(); // (one wave: orders the LDS rows before the next pod reads them)
            pf6++;
            MTICK(pf3);
        }
        // NodeInfo.update (types.go:409-428) on the winner's lane.  The pod's own plugin state (spread tables, anti-affinity
        // bit, per-spec count) is only read again in a LATER window: lane j applies it after the loop, all pods in parallel
        // (a read-modify-write per pod inside the loop cost a memory round trip per pod).
        if (lane == slot) {
            tr0 += q.req0, tr1 += q.req1, tz0 += q.nz0, tz1 += q.nz1, tnp += 1, tplaced += 1;
            const int64_t at = st.placed + committed;
            if (a.log && at < st.log_cap) a.log[at] = (int32_t)g;
        }
        {
            const int32_t wl0 = __builtin_amdgcn_readlane(tl0, slot), wl1 = __builtin_amdgcn_readlane(tl1, slot);
            if (lane == j) my_node = li, my_f = s_f[slot][j], my_l0 = wl0, my_l1 = wl1;
        }
        committed++;
        MTICK(pf4);
        st.winner = g;
        st.last_feasible = cd.nfeas;
        if (st.limit > 0 && st.placed + committed >= st.limit) {
            st.done = DONE_LIMIT;
            stop_reason = 7;
            break;
        }
    }
    // the committed pods' own plugin state: lane j = pod j
    if (lane < committed && my_node >= 0) {
        const int pi = (st.next_pod + lane) % a.n_pods;
        const MPod &q = s_pod[lane];
#pragma unroll
        for (int c = 0; c < kMTsc; c++) { // filtering.go:255-296 on the next cycle of this spec
            const bool counted = c < q.n_tsc && ((my_f >> (1 + c)) & 1u) && q.tsc_self[c];
            const int dv = q.tsc_slot[c] ? my_l1 : my_l0;
            if (counted) a.tbl_pool[q.tsc_tbl[c] + dv] += 1; // (k_multi_refresh derives the spec's new masks)
        }
        if (q.anti) atomicOr(&a.anti_bits[(int64_t)pi * (a.n_pad / 32) + (my_node >> 5)], 1u << (my_node & 31));
        a.per_spec[pi] += 1;
    }
    // touched nodes -> columns (mirrors, int64 columns, pod counts, per-node result)
    if (lane < nt && tplaced > 0) {
        const int sh = a.c.mem_shift;
        a.c.r32[0][t_idx] = tr0, a.c.r32[1][t_idx] = tr1, a.c.z32[0][t_idx] = tz0, a.c.z32[1][t_idx] = tz1;
        a.c.req[0][t_idx] = (int64_t)tr0, a.c.req[1][t_idx] = (int64_t)tr1 << sh;
        a.c.nz_mcpu[t_idx] = (int64_t)tz0, a.c.nz_mem[t_idx] = (int64_t)tz1 << sh;
        a.c.pod_count[t_idx] = tnp;
        a.c.placed_cnt[t_idx] += tplaced;
    }
    if (lane < nt && a.touched) a.touched[lane] = (int32_t)t_idx;
    if (lane == 0) {
        st.n_touched = nt;
        st.placed_first = st.next_pod, st.n_placed = committed;
        st.placed += committed, st.rounds += committed;
        st.windows += 1, st.stops += stop_reason != 0 && stop_reason != 7 && stop_reason != 2;
        st.next_pod = st.single_pod >= 0 ? st.next_pod : (int32_t)((st.next_pod + committed) % a.n_pods);
        // the next window: as many pods as fit (all different specs; not beyond the limit)
        int64_t wn = a.window < a.n_pods ? a.window : a.n_pods;
        if (st.seq_windows > 0 && wn > kMSeqMax) wn = kMSeqMax; // (the next window is this kernel's again)
        if (st.limit > 0 && st.limit - st.placed < wn) wn = st.limit - st.placed;
        if (st.single_pod >= 0) wn = st.done || committed ? 0 : 1;
        st.win_n = (int32_t)(wn < 0 ? 0 : wn);
        if (st.single_pod >= 0 && committed) st.done = st.done ? st.done : DONE_LIMIT; // one cycle asked for, one done
        *a.st = st;
        a.st->stop_count[stop_reason & 7] += 1;
        MTICK(pf5);
        a.st->prof[0] += pf0, a.st->prof[1] += pf1, a.st->prof[2] += pf2, a.st->prof[3] += pf3, a.st->prof[4] += pf4, a.st->prof[5] += pf5, a.st->prof[6] += pf6, a.st->prof[7] += pf7; // (indexed in memory: a runtime index into the register copy would put it in scratch)
    }
}

// ------------------------------------------------------------------------------------------------------------------
// k_multi_commit_par: the window's commit when pods rarely share nodes (config 5: 1024 different pods, every one of
// them lands on a node no other pod of the window chose -- measured: 99 879 new nodes per 100 000 pods).  ASSIGN, then
// VERIFY, then APPLY:
//   A  (one wave, in order, ~20 instructions per pod) pod j takes the first entry of its candidate list that no earlier
//      pod of the window took, under the same bounds as multi_commit_inorder (hidden third keys, exhausted list);
//   B  (all threads, every pair t < j in parallel) the nodes taken by earlier pods are the only nodes whose state differs
//      from what the scan saw -- each carries exactly one more pod -- so pod j's choice is right iff none of them, in
//      that state, beats its candidate for pod j (exact filter + score of pod j on node w_t + pod_t), and its
//      normalization maxima still have an untouched holder.  The first pod that fails ends the window: it is re-scanned
//      against the committed state, as pod 0 of the next window (which never fails).  When windows keep ending early the
//      in-order commit takes over for a while (st.seq_windows);
//   C  (lane j = pod j) NodeInfo.update on the distinct winners, the pods' own spread tables / anti-affinity bits.
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kMParThreads) void k_multi_commit_par(MultiArgs a) {
    __shared__ MState s_st;
    __shared__ MPod s_pod[kMWindowMax];
    __shared__ MCand s_cd[kMWindowMax];
    __shared__ int64_t s_win[kMWindowMax];   // shard-local node index pod j was assigned
    __shared__ uint64_t s_wkey[kMWindowMax]; // ... and its key
    __shared__ int32_t s_node[kMWindowMax][10]; // the winners' columns as the scan saw them: a0 a1 r0 r1 z0 z1 alloc_pods pods l0 l1
    __shared__ int s_wa, s_fail, s_reason, s_unsched;
    __shared__ int64_t s_pick[kMWindowMax];
    __shared__ int s_taken[kMWindowMax];
    __shared__ int s_th_mt[kMWindowMax], s_th_ma[kMWindowMax];
    __shared__ int s_stop[kMWindowMax];
    __shared__ uint32_t s_mag[kMWindowMax][2]; // div_magic of pod j's two normalization maxima: once per pod, not once per pair (two fp64 divisions)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6); // (wave: uniform, and known to the compiler as such)
    unsigned long long tp[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_prev = __builtin_amdgcn_s_memrealtime();
#define PT(i) do { const unsigned long long t_now = __builtin_amdgcn_s_memrealtime(); tp[i] += t_now - t_prev; t_prev = t_now; } while (0)
    const int grp = (int)blockIdx.x;
    // Who takes part is decided by the window's size alone, and workgroups 1 .. 7 learn it from the words k_multi_select left (kMVsHdr),
    // never from MState: a workgroup without a share of the W (W - 1) / 2 pairs leaves before it reads anything workgroup 0 will write;
    // the others are all awaited.
    const int32_t h_par = a.vsync[kMVsHdr + 0], h_w = a.vsync[kMVsHdr + 1], h_next = a.vsync[kMVsHdr + 2];
    if (grp != 0 && (!h_par || grp * kMParThreads >= h_w * (h_w - 1) / 2)) return;
    if (tid == 0 && grp == 0) s_st = *a.st;
    __syncthreads();
    if (grp == 0) {
        if (!s_st.done && s_st.seq_windows > 0 && s_st.committed_epoch != s_st.epoch) { // this window is the in-order commit's
            if (wave == 0) multi_commit_inorder(a);
            return;
        }
        if (s_st.done || s_st.seq_windows > 0 || s_st.committed_epoch == s_st.epoch) return;
    }
    const int W = grp == 0 ? s_st.win_n : h_w; // (the same: k_multi_select wrote the header from the state workgroup 0 reads)
    const int32_t next_pod = grp == 0 ? s_st.next_pod : h_next;
    // (a word per thread and step instead of a struct per thread -- coalesced, all loads in flight -- was tried for the two copies below:
    // 3.0 -> 4.1 us, the index arithmetic costs more than the sixty-four cache lines per load instruction)
    if (tid < W) s_pod[tid] = a.pods[(next_pod + tid) % a.n_pods], s_cd[tid] = a.cands[tid], s_th_mt[tid] = 0, s_th_ma[tid] = 0;
    if (tid == 0) s_wa = W, s_fail = W, s_reason = 0, s_unsched = -1;
    __syncthreads();
    if (tid < W) s_mag[tid][0] = div_magic(s_cd[tid].mt), s_mag[tid][1] = div_magic(s_cd[tid].ma); // (read behind the assignment's barriers)
    PT(0);
    // (no spread tables here: assignment and verification test the spread filter through the specs' per-domain masks,
    // MPod::tsc_allow, and k_multi_refresh derives the placed specs' new masks from the tables the apply step increments)
    PT(1);
    // ---- A: assignment.  Pod j takes the first entry of its list that no EARLIER pod of the window took.  Thread j of the first
    // kMWindowMax threads (two waves since round 6: windows of up to 128 pods) = pod j (its list position, the bounds of what it
    // skipped); whether its current candidate is held by a lower pod is asked of the WHOLE workgroup: 8 threads per pod, 16 earlier
    // pods each, against the picks in LDS (the first form walked the lower lanes with v_readlane inside one wave: ~400 dependent
    // instructions per round, 8.6 us per window).  A pod that finds its candidate taken moves on; repeated until nobody moves.  This
    // reaches the in-order result: a pod only abandons a node that a LOWER pod holds, and the lowest holder of a node never moves, so a
    // node once taken stays taken for every higher pod -- candidates only move forward, at most K steps each; a few rounds when
    // conflicts are rare.
    {
        static_assert(kMParThreads == 8 * kMWindowMax, "eight checking threads per pod");
        const bool aw = tid < kMWindowMax; // an assignment thread (waves 0 and 1)
        const int j = tid;
        const bool live = aw && j < W;
        int stop = 0; // why this pod cannot be assigned (0 = it can)
        int k = 0;
        uint64_t bound = 0;
        int64_t sb0 = -1, sb1 = -1, sb2 = -1;
        int sc0 = 0, sc1 = 0, sc2 = 0;
        const MCand &cd = s_cd[live ? j : 0]; // (indexed in LDS: a register copy with a runtime index would live in scratch)
        if (live) {
            const MPod &q = s_pod[j];
            if ((int32_t)cd.mt != q.mt_a || (int32_t)cd.ma != q.ma_a) stop = 1; // an assumed normalization maximum was wrong
            else if (cd.nfeas == 0) stop = 2;                                     // Unschedulable -- if the pods before it stand
        }
        int64_t pick = live && !stop && cd.n > 0 ? key_index(cd.key[0]) - a.c.global_offset : -1;
#pragma unroll 1
        for (int round = 0; round < kMTopK * kMWindowMax + 2; round++) {
            if (aw) s_pick[tid] = pick;
            __syncthreads();
            bool hit = false;
            { // thread (pod jp, eighth tq): is pod jp's candidate one of the picks of pods tq, tq + 8, ... below it?
                const int jp = tid >> 3, tq = tid & 7;
                const int64_t mine = s_pick[jp];
                int h = 0; // (bitwise, not `||`: sixteen short-circuit tests were sixteen exec-mask branches)
#pragma unroll
                for (int i = 0; i < kMWindowMax / 8; i++) {
                    const int t = tq + 8 * i;
                    h |= (int)(t < jp) & (int)(s_pick[t] == mine);
                }
                hit = h != 0 && mine >= 0;
                const unsigned long long b = __ballot(hit);
                if (tq == 0) s_taken[jp] = (int)((b >> (lane & 56)) & 0xffull); // (the pod's 8 threads are 8 consecutive lanes)
            }
            // (the barrier behind the verdicts also says whether anybody moves: two barriers a round -- the next round's picks are
            // written behind this one, its verdicts behind the next)
            const bool any_moves = __syncthreads_or(hit ? 1 : 0) != 0;
            const bool taken = aw && s_taken[tid] != 0;
            if (taken) { // move on: remember the scan workgroup of the entry skipped
                const int64_t blk = pick / kMBlockNodes;
                int cnt2 = 0;
                if (blk == sb0) cnt2 = ++sc0;
                else if (blk == sb1) cnt2 = ++sc1;
                else if (blk == sb2) cnt2 = ++sc2;
                else if (sb0 < 0) sb0 = blk, cnt2 = sc0 = 1;
                else if (sb1 < 0) sb1 = blk, cnt2 = sc1 = 1;
                else if (sb2 < 0) sb2 = blk, cnt2 = sc2 = 1;
                else bound = ~0ull;
                if (cnt2 == 2) { // both recorded nodes of that scan workgroup are taken: what it hides is bounded by its third key
                    const uint64_t k3 = a.partials[(int64_t)j * a.n_blocks + blk].key3;
                    bound = k3 > bound ? k3 : bound;
                }
                k++;
                pick = k < cd.n ? key_index(cd.key[k < kMTopK ? k : 0]) - a.c.global_offset : -1;
            }
            if (!any_moves) break;
        }
        // the first pod that cannot be assigned ends the window: its index and reason, over both assignment waves
        if (aw) {
            uint64_t ukey = 0;
            if (live && !stop) {
                if (k >= cd.n) { // the list ran out
                    if (cd.bound > bound) bound = cd.bound;
                    stop = 5;
                } else {
                    ukey = cd.key[k < kMTopK ? k : 0];
                    if (bound && ukey < bound) stop = 4; // what the list hides may beat the candidate: the next scan will know
                }
            }
            s_stop[tid] = live ? stop : 0;
            s_win[tid] = pick, s_wkey[tid] = ukey; // (entries at and behind the stop are not read)
            // (a wrong assumed maximum, stop == 1, is repaired by workgroup 0 once the other workgroups have read the pods: below)
            const uint64_t stopped = __ballot(live && stop != 0);
            if (stopped && lane == 0) atomicMin(&s_wa, (wave << 6) + __ffsll((unsigned long long)stopped) - 1); // (s_wa starts at W)
        }
        __syncthreads();
        if (tid == 0) {
            const int wa0 = s_wa;
            const int reason = wa0 < W ? s_stop[wa0] : 0;
            s_reason = reason, s_unsched = reason == 2 ? wa0 : -1;
        }
    }
    __syncthreads();
    PT(2);
    const int wa = s_wa;
    // the winners' columns (as the scan saw them: nobody wrote since)
    for (int i = tid; i < wa * 10; i += kMParThreads) {
        const int j = i / 10, f = i % 10;
        const int64_t n = s_win[j];
        int32_t v;
        switch (f) {
        case 0: v = a.c.a32[0][n]; break;
        case 1: v = a.c.a32[1][n]; break;
        case 2: v = a.c.r32[0][n]; break;
        case 3: v = a.c.r32[1][n]; break;
        case 4: v = a.c.z32[0][n]; break;
        case 5: v = a.c.z32[1][n]; break;
        case 6: v = a.c.alloc_pods[n]; break;
        case 7: v = a.c.pod_count[n]; break;
        case 8: v = a.tsc_label[0] ? a.tsc_label[0][n] : 0; break;
        default: v = a.tsc_label[1] ? a.tsc_label[1][n] : 0; break;
        }
        s_node[j][f] = v;
    }
    __syncthreads();

    PT(3);
    // ---- B: verification, every pair (t < j) in parallel.  The wa (wa - 1) / 2 pairs are dealt evenly -- pair p and pair
    // p + 1024 to thread p (the first form gave pod j's pairs to 16 fixed threads: the threads of the last pods carried four
    // evaluations, those of the first pods none, and the evaluations are the phase's time: ~300 VALU instructions each on ONE
    // CU); the per-pair words are loaded first, all in flight together
    {
        const int npairs = wa * (wa - 1) / 2;
        DevPod p = a.prof;
        constexpr int kAll = kMParThreads * kMParGroups; // verifying threads of the launch
        constexpr int kPer = (kMWindowMax * (kMWindowMax - 1) / 2 + kAll - 1) / kAll; // pairs per thread: 1 at 128 pods over 8 workgroups
        int pj[kPer], pt[kPer];
        uint32_t wv[kPer], bv[kPer];
#pragma unroll
        for (int i = 0; i < kPer; i++) {
            const int pp = grp * kMParThreads + tid + i * kAll;
            const bool on = pp < npairs;
            // pair index -> (j, t), t < j: j (j - 1) / 2 <= pp < (j + 1) j / 2
            int j = (int)((1.0f + __builtin_sqrtf(1.0f + 8.0f * (float)pp)) * 0.5f);
            while (j * (j - 1) / 2 > pp) j--;
            while ((j + 1) * j / 2 <= pp) j++;
            pj[i] = on ? j : -1, pt[i] = on ? pp - j * (j - 1) / 2 : 0;
            const MPod &q = s_pod[on ? j : 0];
            const int64_t n = on ? s_win[pt[i]] : 0;
            wv[i] = on ? a.stat_cls[(int64_t)q.cls * a.n_pad + n] : 0u;
            bv[i] = on && q.anti ? a.anti_bits[(int64_t)((next_pod + j) % a.n_pods) * (a.n_pad / 32) + (n >> 5)] : 0u;
        }
#pragma unroll
        for (int i = 0; i < kPer; i++) {
            const int j = pj[i], t = pt[i];
            if (j < 0) continue;
            const MPod &q = s_pod[j];
            const MCand &cd = s_cd[j];
            const int64_t n = s_win[t];
            const uint32_t w = wv[i];
            const uint32_t cnt = (w >> kStatCntShift) & kStatCntMask, aff = w & kStatAffMask, img = (w >> kStatImgShift) & kStatImgMask;
            if (cnt == cd.mt) atomicAdd(&s_th_mt[j], 1);
            if (aff == cd.ma) atomicAdd(&s_th_ma[j], 1);
            bool ok = (w >> kStatOkBit) && !((bv[i] >> (n & 31)) & 1u);
            if (!ok) continue;
            p.all_zero_req = q.all_zero_req, p.w_bal = q.w_bal, p.w_aff = q.w_aff;
            const NarrowPod nq{q.req0, q.req1, q.nz0, q.nz1};
            const MPod &qt = s_pod[t];
            // node w_t after pod t's placement (NodeInfo.update, types.go:409-428)
            const int32_t a0 = s_node[t][0], a1 = s_node[t][1], r0 = s_node[t][2] + qt.req0, r1 = s_node[t][3] + qt.req1;
            const int32_t z0 = s_node[t][4] + qt.nz0, z1 = s_node[t][5] + qt.nz1, ap = s_node[t][6], np = s_node[t][7] + 1;
            ok = fits_narrow(p, nq, a0, a1, r0, r1, ap, np);
#pragma unroll
            for (int c = 0; c < kMTsc; c++)
                if (c < q.n_tsc && ok) { // PodTopologySpread.Filter on node w_t: pod j's tables have not moved (its own clones only)
                    const int32_t v = q.tsc_slot[c] ? s_node[t][9] : s_node[t][8];
                    ok = ((q.tsc_allow[c] >> (v < 0 || v > kMDomMax ? 0 : v)) & 1ull) != 0;
                }
            if (!ok) continue;
            const int64_t total = static_score(p, cnt, aff, img, cd.mt, cd.ma, s_mag[j][0], s_mag[j][1]) + dynamic_score_narrow(p, nq, a0, a1, r0, r1, z0, z1);
            if (make_key(total, a.c.global_offset + n) > s_wkey[j]) atomicMin(&s_fail, j); // pod j prefers a node an earlier pod took
        }
    }
    __syncthreads();
    { // the other workgroups' findings -> workgroup 0.  ONE word per workgroup (arrived | it sent counts << 1 | (kMWindowMax - its first
      // failing pod) << 8), stored with release semantics and polled by lane g of workgroup 0's first wave: the hand-over is one trip to
      // L2 on either side (the first form -- an arrival counter, a word for the failure, the counts read behind a barrier -- was four
      // dependent trips: 6 us of the 8.5 us phase).  The counts of taken holders of a normalization maximum travel only for a pod
      // whose maximum is not zero (a maximum of zero cannot move: the test below does not read them).
        const int npairs = W * (W - 1) / 2; // (of the window as it was scanned: a window the assignment cut short is awaited all the same)
        if (grp != 0) {
            bool sent = false;
            if (tid < wa) {
                const MCand &cd = s_cd[tid];
                if (cd.mt > 0 && s_th_mt[tid]) atomicAdd(&a.vsync[kMVsMt + tid], s_th_mt[tid]), sent = true;
                if (cd.ma > 0 && s_th_ma[tid]) atomicAdd(&a.vsync[kMVsMa + tid], s_th_ma[tid]), sent = true;
            }
            if (sent) __threadfence();
            const int any = __syncthreads_or(sent ? 1 : 0);
            if (tid == 0) {
                const int f = s_fail;
                __hip_atomic_store(&a.vsync[grp], 1 | (any ? 2 : 0) | ((f < wa ? kMWindowMax - f : 0) << 8), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            }
            return;
        }
        int expect = 0;
        for (int g = 1; g < kMParGroups; g++) expect += g * kMParThreads < npairs ? 1 : 0;
        if (expect) { // (workgroups 1 .. expect take part)
            __shared__ int s_sent;
            if (tid == 0) s_sent = 0;
            if (wave == 0) {
                const bool mine = lane >= 1 && lane <= expect;
                int v = 0, spins = 0;
                while (true) {
                    if (mine && !v) v = __hip_atomic_load(&a.vsync[lane], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
                    if (__ballot(mine && !v) == 0) break;
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > kMVsSpinLimit) __builtin_trap(); // (fail loudly: the launch aborts, the engine reports -EIO)
                }
                if (mine) {
                    a.vsync[lane] = 0; // (zero again for the next launch)
                    if (v >> 8) atomicMin(&s_fail, kMWindowMax - (v >> 8));
                    if (v & 2) s_sent = 1;
                }
            }
            __syncthreads();
            if (s_sent && tid < wa) {
                const int m = __hip_atomic_load(&a.vsync[kMVsMt + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const int f = __hip_atomic_load(&a.vsync[kMVsMa + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (m) s_th_mt[tid] += m, a.vsync[kMVsMt + tid] = 0;
                if (f) s_th_ma[tid] += f, a.vsync[kMVsMa + tid] = 0;
            }
        }
        // every workgroup has read the window's pods: a wrong assumed maximum is repaired now (one window fixes a whole cycle of specs)
        if (tid < W && s_stop[tid] == 1) {
            const int pj = (next_pod + tid) % a.n_pods;
            a.pods[pj].mt_a = (int32_t)s_cd[tid].mt, a.pods[pj].ma_a = (int32_t)s_cd[tid].ma;
        }
    }
    if (tid < wa) { // a normalization maximum may have moved if every holder the scan counted is among the taken nodes
        const MPod &q = s_pod[tid];
        const MCand &cd = s_cd[tid];
        if ((cd.mt > 0 && (int)cd.c_mt <= s_th_mt[tid]) || (cd.ma > 0 && q.w_aff && (int)cd.c_ma <= s_th_ma[tid])) atomicMin(&s_fail, tid);
    }
    __syncthreads();
    PT(4);
    const int fail = s_fail;
    const int ok_n = fail < wa ? fail : wa; // pods 0 .. ok_n-1 stand

    // ---- C: apply (thread j = pod j; the winners are distinct nodes) ------------------------------------------------
    if (tid < ok_n) {
        const int j = tid, pi = (next_pod + j) % a.n_pods;
        const MPod &q = s_pod[j];
        const int64_t n = s_win[j];
        const int sh = a.c.mem_shift;
        const int32_t r0 = s_node[j][2] + q.req0, r1 = s_node[j][3] + q.req1, z0 = s_node[j][4] + q.nz0, z1 = s_node[j][5] + q.nz1;
        a.c.r32[0][n] = r0, a.c.r32[1][n] = r1, a.c.z32[0][n] = z0, a.c.z32[1][n] = z1;
        a.c.req[0][n] = (int64_t)r0, a.c.req[1][n] = (int64_t)r1 << sh;
        a.c.nz_mcpu[n] = (int64_t)z0, a.c.nz_mem[n] = (int64_t)z1 << sh;
        a.c.pod_count[n] = s_node[j][7] + 1;
        a.c.placed_cnt[n] += 1;
        bool all = true; // counted iff the node has ALL the pod's hard keys and passes the inclusion policies (filtering.go:267-277)
#pragma unroll
        for (int c = 0; c < kMTsc; c++)
            if (c < q.n_tsc) all = all && (q.tsc_slot[c] ? s_node[j][9] : s_node[j][8]) != 0;
#pragma unroll
        for (int c = 0; c < kMTsc; c++) {
            const bool counted = c < q.n_tsc && all && q.tsc_self[c] && (q.tsc_inc[c] < 0 || a.inc_pool[(int64_t)q.tsc_inc[c] * a.n_pad + n] != 0);
            const int dv = q.tsc_slot[c] ? s_node[j][9] : s_node[j][8];
            if (counted) a.tbl_pool[q.tsc_tbl[c] + dv] += 1; // (k_multi_refresh derives the spec's new masks)
        }
        if (q.anti) atomicOr(&a.anti_bits[(int64_t)pi * (a.n_pad / 32) + (n >> 5)], 1u << (n & 31));
        a.per_spec[pi] += 1;
        const int64_t at = s_st.placed + j;
        if (a.log && at < s_st.log_cap) a.log[at] = (int32_t)(a.c.global_offset + n);
        if (a.touched) a.touched[j] = (int32_t)n;
    }
    if (tid == 0) {
        MState st = s_st;
        st.committed_epoch = st.epoch;
        st.n_touched = ok_n;
        st.placed_first = next_pod, st.n_placed = ok_n;
        int reason = fail < wa ? 3 : s_reason; // 3: a pod preferred a taken node / a maximum may have moved
        if (ok_n > 0) st.winner = a.c.global_offset + s_win[ok_n - 1], st.last_feasible = s_cd[ok_n - 1].nfeas;
        st.placed += ok_n, st.rounds += ok_n;
        if (fail >= wa && s_unsched == wa) { // the pod after the last assigned one is Unschedulable, and everything before it stands
            st.done = DONE_UNSCHEDULABLE, st.stop_spec = (next_pod + wa) % a.n_pods, st.rounds += 1, st.last_feasible = 0, st.winner = -1;
            reason = 2;
        }
        if (st.limit > 0 && st.placed >= st.limit) st.done = st.done ? st.done : DONE_LIMIT, reason = reason ? reason : 7;
        st.windows += 1, st.stops += reason != 0 && reason != 7 && reason != 2;
        st.next_pod = st.single_pod >= 0 ? st.next_pod : (int32_t)((st.next_pod + ok_n) % a.n_pods);
        // windows that keep ending early because pods prefer taken nodes: the in-order commit handles that regime
        if (reason == 3 && ok_n < 4 && W >= 16) st.seq_windows = 4;
        int64_t wn = a.window < a.n_pods ? a.window : a.n_pods;
        if (st.seq_windows > 0 && wn > kMSeqMax) wn = kMSeqMax; // (the in-order commit holds a pod / a touched node per lane)
        if (st.limit > 0 && st.limit - st.placed < wn) wn = st.limit - st.placed;
        if (st.single_pod >= 0) wn = st.done || ok_n ? 0 : 1;
        st.win_n = (int32_t)(wn < 0 ? 0 : wn);
        if (st.single_pod >= 0 && ok_n) st.done = st.done ? st.done : DONE_LIMIT;
        *a.st = st;
        a.st->stop_count[reason & 7] += 1;
        PT(5);
        for (int i = 0; i < 6; i++) a.st->prof[i] += tp[i];
    }
}

// ------------------------------------------------------------------------------------------------------------------
// k_multi_masks: thread = pod spec.  MPod::tsc_allow from the spec's tables as they stand (when a run begins: the tables may
// have been restored by ccsim_reset_state).
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_multi_masks(MultiArgs a) {
    const int pi = blockIdx.x * 256 + threadIdx.x;
    if (pi >= a.n_pods) return;
    const MPod q = a.pods[pi];
    for (int c = 0; c < kMTsc; c++) {
        const int32_t *tbl = a.tbl_pool + q.tsc_tbl[c];
        const uint8_t *present = a.present_pool + q.tsc_tbl[c];
        a.pods[pi].tsc_allow[c] = m_allow_mask(q, c, [&](int v) { return m_stage(tbl[v], present[v]); }, 0);
    }
}

// ------------------------------------------------------------------------------------------------------------------
// k_multi_refresh: grid (spec blocks, kMTouched), after the commit.  (1) The specs the window placed: their per-domain spread
// masks (block (0, t), a wave per constraint).  (2) Block row t = touched node t of the window just committed; thread = pod
// spec.  Recomputes the memo word of (spec, node) from the node's columns as the commit left them -- for every spec whose
// row is stamped with the maxima it assumes today (any other row is recomputed as a whole by the spec's next scan).
// Idempotent: a launch that follows a window no commit kernel took repeats the last one.
// ------------------------------------------------------------------------------------------------------------------
constexpr int kMRefreshThreads = 256;
__global__ __launch_bounds__(kMRefreshThreads) void k_multi_refresh(MultiArgs a) {
    const int t = blockIdx.y;
    if (blockIdx.x == 0 && t < a.st->n_placed && threadIdx.x < 64 * kMTsc) {
        // the t-th spec the window placed: its spread masks from its tables as the commit left them -- wave c = constraint c,
        // lane v = domain v (m_allow_mask, a domain per lane: the minimum over the present domains, then the skew test)
        const int ps = (a.st->placed_first + t) % a.n_pods, c = threadIdx.x >> 6, v = threadIdx.x & 63;
        const MPod &q = a.pods[ps];
        if (c < q.n_tsc) { // (uniform per wave)
            const bool dom = v >= 1 && v <= q.tsc_ndom[c];
            const int32_t x = dom ? m_stage(a.tbl_pool[q.tsc_tbl[c] + v], a.present_pool[q.tsc_tbl[c] + v]) : kMAbsent;
            const uint32_t mn = ~wave_max_u32(~(x < kMAbsent ? (uint32_t)x : 0x7fffffffu));
            const int64_t lim = (int64_t)q.tsc_max_skew[c] + (q.tsc_npresent[c] < q.tsc_min_dom[c] ? 0 : (int64_t)mn) - q.tsc_self[c];
            const bool pass = !dom || (int64_t)m_count(x) <= lim;
            const uint64_t mask = (uint64_t)__ballot(pass) & ~1ull;
            if (v == 0) a.pods[ps].tsc_allow[c] = mask;
        }
    }
    if (!a.memo || t >= a.st->n_touched) return;
    const int pi = blockIdx.x * kMRefreshThreads + threadIdx.x;
    if (pi >= a.n_pods || !m_memo_valid(a, pi)) return;
    const int64_t n = a.touched[t];
    const MPod &q = a.pods[pi];
    DevPod p = a.prof;
    p.all_zero_req = q.all_zero_req, p.w_bal = q.w_bal, p.w_aff = q.w_aff;
    const NarrowPod nq{q.req0, q.req1, q.nz0, q.nz1};
    const uint32_t w = a.stat_cls[(int64_t)q.cls * a.n_pad + n];
    const uint32_t anti = q.anti ? (a.anti_bits[(int64_t)pi * (a.n_pad / 32) + (n >> 5)] >> (n & 31)) & 1u : 0u;
    const int32_t a0 = a.c.a32[0][n], a1 = a.c.a32[1][n], r0 = a.c.r32[0][n], r1 = a.c.r32[1][n], z0 = a.c.z32[0][n], z1 = a.c.z32[1][n];
    const int32_t room = (int64_t)a.c.pod_count[n] + 1 <= (int64_t)a.c.alloc_pods[n] ? 1 : 0;
    const bool ok = (w >> kStatOkBit) && fits_narrow(p, nq, a0, a1, r0, r1, room, 0) && !anti;
    uint16_t word = 0;
    if (ok) {
        const uint32_t cnt = (w >> kStatCntShift) & kStatCntMask, aff = w & kStatAffMask, img = (w >> kStatImgShift) & kStatImgMask;
        const uint32_t total = (uint32_t)(static_score(p, cnt, aff, img, (uint32_t)q.mt_a, (uint32_t)q.ma_a) + dynamic_score_narrow(p, nq, a0, a1, r0, r1, z0, z1));
        word = m_memo_word(true, total, cnt, aff, (uint32_t)q.mt_a, (uint32_t)q.ma_a);
    }
    a.memo[(int64_t)pi * a.n_pad + n] = word;
}

// ------------------------------------------------------------------------------------------------------------------
// k_multi_hist: FitError diagnosis of the pod that ended the run (types.go:787-836): first failing plugin in filter order.
// ------------------------------------------------------------------------------------------------------------------
struct MultiHistArgs {
    MultiArgs m;
    int32_t pod;
    const uint8_t *sreason_cls; // [n_cls][n_pad]
    const int32_t *taintset_id;
    const int64_t *alloc[2];
    int64_t n;
    unsigned long long *hist, *hist_ts, *hist_code;
};

__global__ __launch_bounds__(kThreads) void k_multi_hist(MultiHistArgs h) {
    const MultiArgs &a = h.m;
    const MPod q = a.pods[h.pod];
    __shared__ int32_t s_min[kMTsc];
    if (threadIdx.x < kMTsc) s_min[threadIdx.x] = threadIdx.x < q.n_tsc ? m_tbl_min(a.tbl_pool + q.tsc_tbl[threadIdx.x], a.present_pool + q.tsc_tbl[threadIdx.x], q.tsc_ndom[threadIdx.x]) : 0;
    __syncthreads();
    for (int64_t n = (int64_t)blockIdx.x * kThreads + threadIdx.x; n < h.n; n += (int64_t)gridDim.x * kThreads) {
        const uint8_t sr = h.sreason_cls[(int64_t)q.cls * a.n_pad + n];
        if (sr == 1) { atomicAdd(&h.hist[0], 1ull); continue; }
        if (sr == 2) { atomicAdd(&h.hist_ts[h.taintset_id ? h.taintset_id[n] : 0], 1ull); continue; }
        if (sr == 3) { atomicAdd(&h.hist[2], 1ull); continue; }
        bool any = false, unresolvable = false;
        if ((int64_t)a.c.pod_count[n] + 1 > (int64_t)a.c.alloc_pods[n]) atomicAdd(&h.hist[3], 1ull), any = true;
        if (!q.all_zero_req)
            for (int col = 0; col < 2; col++) {
                const int64_t rq = q.req_wide[col];
                if (!(rq > 0)) continue;
                const int64_t al = h.alloc[col][n], us = a.c.req[col][n];
                if (rq > al - us) {
                    atomicAdd(&h.hist[4 + col], 1ull), any = true;
                    if (rq > al) unresolvable = true;
                }
            }
        if (any) {
            if (!unresolvable) atomicAdd(&h.hist_code[0], 1ull);
            continue;
        }
        bool failed = false;
        for (int c = 0; c < q.n_tsc && !failed; c++) {
            const int32_t v = q.tsc_slot[c] ? a.tsc_label[1][n] : a.tsc_label[0][n];
            const int r = m_pts_check(q, c, v, v ? a.tbl_pool[q.tsc_tbl[c] + v] : 0, s_min[c]);
            if (r == 1) atomicAdd(&h.hist[kHistPtsMissing], 1ull), failed = true;
            else if (r == 2) atomicAdd(&h.hist[kHistPtsSkew], 1ull), atomicAdd(&h.hist_code[0], 1ull), failed = true;
        }
        if (failed || !q.anti) continue;
        if ((a.anti_bits[(int64_t)h.pod * (a.n_pad / 32) + (n >> 5)] >> (n & 31)) & 1u) // satisfyPodAntiAffinity fails first (filtering.go:410-432)
            atomicAdd(&h.hist[kHistIpa + 1], 1ull), atomicAdd(&h.hist_code[0], 1ull);
    }
}

} // namespace ccsim
