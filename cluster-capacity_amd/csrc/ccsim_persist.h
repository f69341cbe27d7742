// ccsim_persist.h -- CCSIM_MODE_BATCHED as ONE persistent launch: the whole level-batched run (ccsim_level.h explains
// why levels are exact) inside a single kernel whose workgroups keep their nodes in LDS for the whole run.
//
// Why: the multi-kernel form pays, per score level, one commit dispatch (a latency chain of ~18 us at 1M nodes:
// launch, dense read of the 4-byte score cache, compaction barriers, row gather from HBM, run-down, store) plus a
// one-block decision kernel (~7 us) plus two dispatch boundaries.  Nothing in a level needs HBM: a level touches a
// few per cent of the nodes and the decision needs five numbers.  So:
//   * one 512-thread workgroup per CU, K x 512 nodes per workgroup (K <= 8), the node state (36 B per node in the narrow
//     units of ccsim_kernels.h + a 2-byte work-list slot) resident in the CU's 160 KiB of LDS: 256 CUs x 4096 nodes =
//     1 048 576 nodes per GPU -- the BASELINE 1M-node snapshot exactly fits one MI355X;
//   * per level every workgroup scans the 16-bit scores of its own nodes (LDS), compacts the level's nodes into an LDS
//     work list, its first waves run them down (wave_run_down, ccsim_level.h) and re-score them, and the block
//     contributes <= 5 packed 64-bit words to a grid-wide reduction;
//   * the grid-wide reduction doubles as the grid barrier (grid_reduce below): relaxed agent-scope atomics into
//     8 group-sharded slots, hierarchical arrival counters, one generation word to poll -- no fences, because the only
//     data that crosses workgroups are those atomically updated words ("8-B agent atomics both sides",
//     MI355X_MICROARCH.md), and no zeroing, because max-words carry a generation tag and add-words are cumulative;
//   * every workgroup runs the same decision state machine on the same reduced numbers (replicated, like the ranks of
//     the sharded protocol), so nothing is broadcast.
// HBM is read once when the run starts (36 B per node) and written once when it ends; in between the kernel is bound
// by the grid-barrier latency (tools/barrier_bench.hip) plus the run-down arithmetic of the level.
//
// Exactness.  The FAST path commits a level blindly and validates afterwards: if the level exhausted every feasible
// holder of a normalization maximum (P/helper/normalize_score.go:28-56) or crossed --max-limit, the whole level is
// rolled back (integer adds undone from the per-node `took`) and redone on the ORDERED path: plan (run-down lengths,
// exhausted holders, highest exhausted index = the cut) -> grid reduce + per-block prefix -> commit the nodes up to the
// cut in canonical order with positions (limit clamp, placement log) -> grid reduce.  With a placement log every level
// takes the ordered path.  Results are identical to the multi-kernel batched mode and to the sequential mode.
#pragma once
#include "ccsim_level.h"

namespace ccsim {

constexpr int kPThreads = 512; // 8 waves = 2 per SIMD: a 256-VGPR budget (the run-down's working set spills at 128)
constexpr int kPWaves = kPThreads / 64;
constexpr int kPGroups = 8;        // arrival / slot sharding (one group per XCD when dispatch is round-robin)
constexpr int kPMaxGrid = 1024;
constexpr int kPMaxRanks = 8;      // GPUs of one box (mailbox form)
constexpr uint32_t kScInf = 0xffffu; // 16-bit score: infeasible
constexpr int kTagShift = 40;        // max-words: generation tag above a 40-bit payload
constexpr int DONE_ERROR = 3;
constexpr unsigned kGenErr = 0xffffffffu; // PersistSync::gen: the run is abandoned
constexpr int kPSpinLimit = 1 << 22;
constexpr int kPMboxSpinLimit = 1 << 21; // polls of a rank's mailbox (two system-scope loads + s_sleep each: a few seconds) before the rank gives up

struct PersistSync { // global memory, one per rank; zeroed by the host before every launch
    unsigned long long slot[2][kPGroups][8]; // [parity][group][word]; words 0,3,4,5,6: tagged max; 1,2,7: cumulative add
    unsigned int garrive[kPGroups][16];      // cumulative arrivals per group (one 64-byte line each)
    unsigned int top[16];                    // cumulative group completions
    unsigned int gen[16];                    // generations released so far; kGenErr: abandoned.  Advanced ONLY by compare-and-swap, so a
                                             // generation is either released or abandoned for every workgroup alike ("all write back or none")
    unsigned int err[16];                    // [0] a workgroup of this rank gave up (what the host reads)
    unsigned int blockT[kPMaxGrid];          // ordered path: planned placements per workgroup
    unsigned long long prof[16];             // workgroup 0: s_memrealtime ticks per phase (see PTICK) + [6] level passes
};

// Mailbox form (several ranks -- the GPUs of one box, or VIRTUAL ranks inside one grid for validation): one box per rank, in
// fine-grained memory mapped into every peer.  The workgroup that completes a rank's local reduction writes the rank's eight words
// into row [parity][its rank] of EVERY rank's box (its own included) as sixteen 8-byte granules {tag : 32 | half a word : 32}: a
// granule is written by ONE store, so tag and data arrive together (MI355X_MICROARCH.md "handoff-1to1": data-tagged granules need
// no fence, no flag behind the data) -- over xGMI that is G - 1 remote 128-byte writes per generation.  Every workgroup of a rank
// polls its own rank's box (local memory) until all G rows carry the generation's tag, then combines the rows exactly like the
// eight group slots of the local form.  Tags are (launch sequence << 20 | generation + 1): nothing is zeroed between launches.
struct PersistMailbox {
    unsigned long long g[2][kPMaxRanks][16]; // [parity][source rank][2 * word + half]
    unsigned int err[16];                    // [0] = launch tag of a run some rank abandoned
    unsigned int pad[16];
};

// the columns the persistent kernel touches (a slim argument block: the full DevCols would sit in ~90 SGPRs)
struct PersistCols {
    const int32_t *a32[2];
    int32_t *r32[2], *z32[2];
    const int32_t *alloc_pods;
    int32_t *pod_count, *placed_cnt;
    const uint32_t *stat;
    int64_t *req[2], *nz_mcpu, *nz_mem;
    int64_t n_pad, global_offset;
    int32_t mem_shift;
    // round 4 -- the step frame folded into the launch:
    int32_t from_pristine;          // load the PRISTINE wide columns (a pending ccsim_reset_state: no separate restore pass)
    const int64_t *p_req[2], *p_nz[2];
    const int32_t *p_pod_count;
    int64_t n;                      // real nodes (the FitError histogram leaves the padding out)
    const uint8_t *sreason;         // first failing static filter per node (k_static)
    const int32_t *taintset_id;
    unsigned long long *hist, *hist_ts, *hist_code; // the terminal cycle's diagnosis, from the state in LDS (types.go:787-836); null = not here
    int32_t n_taintsets, cnt_assign; // cnt_assign: placed_cnt is written, not added to (first launch of a run)
    void *cnt_narrow;               // the per-run counts once more in 1- or 2-byte elements (ccsim_report.per_node_count_narrow), nullptr = not asked for
    int32_t cnt_narrow_width;
    int32_t *rows;                  // mailbox form: the final state goes to the commit rows (k_rows_flush publishes it once every rank agrees)
    int32_t skip_wide;              // the int64 columns are NOT written back: they are a function of the mirrors (value << unit), and the
                                    // engine re-derives them when an entry point that reads them comes along (ensure_cols: k_widen) --
                                    // 32 of the write-back's 56 B per node
};

struct PersistArgs {
    PersistCols c;
    DevPod p;
    DevState *st;
    PersistSync *sync;   // [number of (virtual) ranks on this device]
    int32_t *log;
    int32_t want_log;
    int32_t max_syncs; // generations per launch (the host relaunches an unfinished run: state lives in the columns)
    int32_t seq_steps; // run-down placements a lane evaluates itself before the wave-cooperative tail (ccsim_level.h)
    int32_t level_batch; // fast path: score levels resolved per grid-wide sync (>= 1)
    int32_t prof;        // measurement runs: per-phase s_memrealtime stamps (each stamp costs a few hundred ns)
    int32_t fault;       // test knob (CCSIM_PERSIST_FAULT=1): workgroup 0 never arrives at the first barrier -- the lost-workgroup path
    int32_t spec_cut;    // every global node index fits 24 bits: the event prediction carries the node, and a blind batch may END at the event
    int32_t end_at_empty; // a batch that ends at an event and leaves no feasible node ends the launch (no re-score pass that would find none; CCSIM_PERSIST_END=0: A/B knob)
    int32_t hint_valid, hint_mt, hint_ma; // the normalization maxima the last launch STARTED with (assumed, verified in the scores' reduce)
    // mailbox form
    int32_t n_ranks, rank;   // ranks of the job / this device's rank (virtual ranks: rank of workgroup b = b / bpr)
    int32_t vranks, bpr;     // virtual ranks inside this grid (0 = a real rank per device), workgroups per virtual rank
    uint32_t tag_base;       // launch sequence << 20
    PersistMailbox *mbox[kPMaxRanks]; // every rank's box as THIS device addresses it
    int32_t *ok_flag;        // mailbox form: preset to 1 by the host, zeroed by any workgroup that gives up or by an unfinished run -- the
                             // ranks agree on it (ncclAllReduce min, enqueued right behind the launch) before anything is published
};

template <int K>
struct PersistLds {
    int32_t a0[K * kPThreads], a1[K * kPThreads];
    int32_t r0[K * kPThreads], r1[K * kPThreads], z0[K * kPThreads], z1[K * kPThreads];
    uint32_t pods[K * kPThreads]; // allocatable pods << 16 | pods on the node
    uint32_t ws[K * kPThreads];   // bit31 static filters passed | bit30 holds max prefer-count | bit29 holds max affinity sum | static score
    uint32_t sct[K * kPThreads];  // low 16: TotalScore (kScInf infeasible) | high 16: placements of the pending level (took)
    uint16_t list[K * kPThreads]; // work list of the level (local node ids)
};

__device__ __forceinline__ unsigned p_ld_u32(const unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned long long p_ld_u64(const unsigned long long *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long p_ld_sys_u64(const unsigned long long *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Grid-wide reduction + barrier.  Thread 0 has put this workgroup's contribution into s_v[0..7] (LDS; zero = nothing to
// add); on return s_red[0..7] (LDS) holds the result over every workgroup of every rank: words 0, 3, 4, 5, 6 combine with MAX
// (payload < 2^40), words 1, 2, 7 with ADD.  One generation = one call by every workgroup.
struct GridCtx {
    PersistSync *s;
    unsigned gen_no;             // generations completed
    unsigned gsize, ngroups, g;  // this workgroup's group (within its rank)
    unsigned long long prev0, prev1; // wave 0: cumulative value of this lane's (group | rank, word) at the last read, per parity (two
                                 // scalars, not an array: a runtime index would put the whole context into a scratch frame)
    unsigned long long *s_v;     // LDS [8] in
    unsigned long long *s_red;   // LDS [8] out
    int *s_err;                  // LDS
    int fault;                   // (test knob, see PersistArgs)
    // mailbox form
    int n_ranks, rank;
    uint32_t tag_base;
    PersistMailbox *const *mbox;
    unsigned long long *s_rk;    // LDS [kPMaxRanks][8]: the ranks' words of the last generation (the ordered path's prefix over lower ranks)
    int *s_flag;                 // LDS: this workgroup completed its rank's local reduction
};

__device__ __forceinline__ bool is_max_word(int w) { return w == 0 || (w >= 3 && w <= 6); }

// a value every lane holds identically (read from LDS): tell the compiler, so that it lives in SGPRs
__device__ __forceinline__ unsigned long long uni64(unsigned long long v) {
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}

// thread 0: this workgroup's words into its group's slots, then the arrival.  Returns true in the workgroup whose arrival completed
// the rank (every group complete).
__device__ __forceinline__ bool grid_contribute(GridCtx &gc, unsigned par) {
    PersistSync *s = gc.s;
    const unsigned long long tag = (unsigned long long)(gc.gen_no + 1) << kTagShift;
#pragma unroll
    for (int w = 0; w < 8; w++) {
        const unsigned long long x = gc.s_v[w];
        if (x == 0) continue;
        if (is_max_word(w)) __hip_atomic_fetch_max(&s->slot[par][gc.g][w], tag | x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else __hip_atomic_fetch_add(&s->slot[par][gc.g][w], x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        gc.s_v[w] = 0;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // contributions performed before the arrival is counted
    const bool lost = gc.fault && blockIdx.x == 0 && gc.gen_no == 0; // (injected: behaves like a workgroup that is not resident)
    const unsigned a = lost ? 0xfffffff0u : __hip_atomic_fetch_add(&s->garrive[gc.g][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (a + 1 == gc.gsize * (gc.gen_no + 1)) {
        const unsigned t = __hip_atomic_fetch_add(&s->top[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return t + 1 == gc.ngroups * (gc.gen_no + 1);
    }
    return false;
}

// wave 0: the rank's eight words out of its group slots.  lane = word * 8 + group; the 8 groups of a word are reduced on the DPP
// network; lane w * 8 + 7 ends up with word w.  CUMULATIVE: add-words are returned as the running sums the slots hold (the
// mailbox form ships those, its readers take differences); otherwise as this generation's value.
template <bool CUMULATIVE>
__device__ __forceinline__ unsigned long long grid_read_slots(GridCtx &gc, unsigned par) {
    const int w = threadIdx.x >> 3, g = threadIdx.x & 7;
    const unsigned long long cur = (unsigned)g < gc.ngroups ? p_ld_u64(&gc.s->slot[par][g][w]) : 0ull;
    unsigned long long val;
    if (is_max_word(w)) val = (cur >> kTagShift) == (unsigned long long)(gc.gen_no + 1) ? (cur & ((1ull << kTagShift) - 1)) : 0ull;
    else if (CUMULATIVE) val = cur;
    else {
        val = cur - (par ? gc.prev1 : gc.prev0);
        if (par) gc.prev1 = cur; else gc.prev0 = cur;
    }
#pragma unroll
    for (int st = 0; st < 3; st++) { // row_shr 1, 2, 4
        const unsigned long long o = dpp_move_u64(0ull, val, st);
        val = is_max_word(w) ? (o > val ? o : val) : val + o;
    }
    return val;
}

template <bool MB>
__device__ __forceinline__ void grid_reduce(GridCtx &gc) {
    PersistSync *s = gc.s;
    const unsigned par = gc.gen_no & 1u;
    if (!MB) {
        if (threadIdx.x == 0) {
            if (grid_contribute(gc, par)) { // the last arrival releases the generation -- unless somebody has abandoned the run
                unsigned expect = gc.gen_no;
                (void)__hip_atomic_compare_exchange_strong(&s->gen[0], &expect, gc.gen_no + 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            int spins = 0;
            for (;;) {
                unsigned g = p_ld_u32(&s->gen[0]);
                if (g != kGenErr && g >= gc.gen_no + 1) break;
                if (g != kGenErr && ++spins > kPSpinLimit) { // bounded: a lost workgroup must not hang the GPU.  Abandon by compare-and-swap:
                    unsigned expect = gc.gen_no;              // if the release won the race after all, this workgroup goes on like every other
                    (void)__hip_atomic_compare_exchange_strong(&s->gen[0], &expect, kGenErr, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    g = p_ld_u32(&s->gen[0]);
                    if (g != kGenErr) break;
                }
                if (g == kGenErr) {
                    __hip_atomic_store(&s->err[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    *gc.s_err = 1;
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        __syncthreads();
        if (threadIdx.x < 64) {
            const unsigned long long val = grid_read_slots<false>(gc, par);
            if ((threadIdx.x & 7) == 7) gc.s_red[threadIdx.x >> 3] = val;
        }
        __syncthreads();
        gc.gen_no += 1;
        return;
    }
    // ---- mailbox form ----------------------------------------------------------------------------------------------------------
    if (threadIdx.x == 0) *gc.s_flag = grid_contribute(gc, par) ? 1 : 0;
    __syncthreads();
    const uint32_t tag = gc.tag_base | (gc.gen_no + 1);
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        if (uni32(*gc.s_flag)) { // this workgroup completed the rank: publish the rank's words to every box
            const unsigned long long rv = grid_read_slots<true>(gc, par);
            const int q = lane & 15;                                                 // granule: word q >> 1, half q & 1
            const unsigned long long wv = (unsigned long long)__shfl((long long)rv, (q >> 1) * 8 + 7, 64);
            const unsigned long long gran = ((unsigned long long)tag << 32) | (uint32_t)(q & 1 ? wv >> 32 : wv);
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int peer = (lane >> 4) + 4 * h;
                if (peer < gc.n_ranks) __hip_atomic_store(&gc.mbox[peer]->g[par][gc.rank][q], gran, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
        // every workgroup: wait for the G rows of this generation in its own rank's box.  lane = rank * 8 + word
        const int r = lane >> 3, w = lane & 7;
        const PersistMailbox *box = gc.mbox[gc.rank];
        unsigned long long lo = 0, hi = 0;
        int spins = 0;
        bool bad = false;
        for (;;) {
            bool ok = true;
            if (r < gc.n_ranks) {
                lo = p_ld_sys_u64(&box->g[par][r][2 * w]), hi = p_ld_sys_u64(&box->g[par][r][2 * w + 1]);
                ok = (uint32_t)(lo >> 32) == tag && (uint32_t)(hi >> 32) == tag;
            }
            if (__ballot(ok) == ~0ull) break;
            ++spins;
            if (spins > kPMboxSpinLimit || ((spins & 255) == 0 && (__hip_atomic_load(&box->err[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == (gc.tag_base | 1u) ||
                                                               p_ld_u32(&s->err[0])))) {
                bad = true;
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
        if (bad) { // bounded: tell every rank (they stop waiting for this one), remember it for the host
            if (lane < gc.n_ranks) __hip_atomic_store(&gc.mbox[lane]->err[0], gc.tag_base | 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if (lane == 0) {
                __hip_atomic_store(&s->err[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                *gc.s_err = 1;
            }
        }
        unsigned long long val = r < gc.n_ranks ? ((hi & 0xffffffffull) << 32) | (lo & 0xffffffffull) : 0ull;
        if (!is_max_word(w)) { // the boxes carry running sums: this generation's value is the difference
            const unsigned long long d = val - (par ? gc.prev1 : gc.prev0);
            if (par) gc.prev1 = val; else gc.prev0 = val;
            val = d;
        }
        gc.s_rk[r * 8 + w] = val;
#pragma unroll
        for (int off = 8; off < 64; off <<= 1) { // over the ranks (lanes 8 apart)
            const unsigned long long o = (unsigned long long)__shfl_xor((long long)val, off, 64);
            val = is_max_word(w) ? (o > val ? o : val) : val + o;
        }
        if (r == 0) gc.s_red[w] = val;
    }
    __syncthreads();
    gc.gen_no += 1;
}

// wave_run_down for long run-downs (several levels per sync): the lane-sequential prefix as in ccsim_level.h, then the
// nodes still running are finished FOUR AT A TIME -- each 16-lane row of the wave takes one node, lane r of the row
// evaluates the node after k0 + r + 1 further placements (closed form), one ballot per round finds every row's first stop.
// The 64-candidates-per-node form of ccsim_level.h spends a whole wave on a node that needs 10-30 more placements.
__device__ __forceinline__ NodeNarrow nd_shfl(const NodeNarrow &n, int src) { // per-lane source: LDS permutes, independent of one another
    NodeNarrow o;
    o.a0 = __shfl(n.a0, src, 64), o.a1 = __shfl(n.a1, src, 64), o.r0 = __shfl(n.r0, src, 64), o.r1 = __shfl(n.r1, src, 64);
    o.z0 = __shfl(n.z0, src, 64), o.z1 = __shfl(n.z1, src, 64), o.a_pods = __shfl(n.a_pods, src, 64), o.npods = __shfl(n.npods, src, 64);
    o.w = (uint32_t)__shfl((int32_t)n.w, src, 64);
    o.placed = 0;
    return o;
}

// M: the level the node runs down to -- PER LANE (a batch that ends inside a level takes that level only up to the cut: ev_cut)
__device__ __forceinline__ int32_t wave_run_down_rows(const RunCtx &cx, const NodeNarrow &n, int32_t stat, int32_t M, bool mine, bool &feas_after,
                                                      int seq_steps) {
    const int lane = threadIdx.x & 63, row = lane >> 4, rl = lane & 15;
    int32_t my_j = 0;
    feas_after = true;
    if (!__ballot(mine)) return 0;
    NodeNarrow cur = n;
    bool running = mine;
    if (running) { // the states that cannot end the run-down are not looked at (ccsim_kernels.h run_down_safe_skip)
        const int32_t k = run_down_safe_skip(cx.p, cx.q, cur.a0, cur.a1, cur.r0, cur.r1, cur.z0, cur.z1, cur.a_pods, cur.npods, stat, M);
        if (k > 0) nd_apply(cx, cur, k), my_j = k;
    }
#pragma unroll 1
    for (int it = 0; it < seq_steps && __ballot(running); it++) {
        if (running) {
            nd_apply(cx, cur, 1);
            my_j++;
            feas_after = nd_feasible(cx, cur);
            running = feas_after && nd_score(cx, cur, (int64_t)stat, NoRcp{}) >= (int64_t)M;
        }
    }
    uint64_t todo = __ballot(running);
#pragma unroll 1
    while (todo) {
        int src = -1; // the node of this lane's row: the row-th lowest lane still running
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const int b = todo ? __ffsll((unsigned long long)todo) - 1 : -1;
            if (row == g) src = b;
            if (todo) todo &= todo - 1;
        }
        const bool active = src >= 0;
        const NodeNarrow base = nd_shfl(cur, active ? src : 0);
        const int32_t bstat = __shfl(stat, active ? src : 0, 64), bM = __shfl(M, active ? src : 0, 64);
        const int32_t room = (int32_t)nd_room(base); // after `room` more placements the node is full
        int32_t j = 0;
        bool f_end = true, row_done = !active;
#pragma unroll 1
        for (int32_t k0 = 0;; k0 += 16) {
            NodeNarrow t = base;
            const int32_t k = k0 + rl + 1;
            nd_apply(cx, t, k < room ? k : room);
            const bool f = nd_feasible(cx, t); // (k >= room: the pod count alone makes it infeasible)
            const bool stop = !(f && nd_score(cx, t, (int64_t)bstat, NoRcp{}) >= (int64_t)bM);
            const uint64_t sm = __ballot(stop), fm = __ballot(f);
            const uint32_t seg = (uint32_t)(sm >> (row * 16)) & 0xffffu;
            if (!row_done && seg) {
                const int first = __ffs((int)seg) - 1;
                j = k0 + first + 1;
                f_end = (fm >> (row * 16 + first)) & 1ull;
                row_done = true;
            }
            if (!__ballot(!row_done) || k0 > (1 << 20)) break; // (a run-down is bounded by the node's pod capacity)
        }
#pragma unroll
        for (int g = 0; g < 4; g++) { // hand each row's result to the lane that owns the node
            const int sg = __builtin_amdgcn_readlane(src, g * 16);
            const int32_t jg = __builtin_amdgcn_readlane(j, g * 16);
            const int fg = __builtin_amdgcn_readlane((int)f_end, g * 16);
            if (sg >= 0 && lane == sg) my_j += jg, feas_after = fg != 0;
        }
    }
    return my_j;
}

template <int K>
__device__ __forceinline__ NodeNarrow p_load_node(const PersistLds<K> &L, int li) {
    NodeNarrow n;
    n.a0 = L.a0[li], n.a1 = L.a1[li], n.r0 = L.r0[li], n.r1 = L.r1[li], n.z0 = L.z0[li], n.z1 = L.z1[li];
    const uint32_t pd = L.pods[li];
    n.a_pods = (int32_t)(pd >> 16), n.npods = (int32_t)(pd & 0xffffu);
    n.w = L.ws[li];
    n.placed = 0;
    return n;
}
template <int K>
__device__ __forceinline__ void p_store_dyn(PersistLds<K> &L, int li, const NodeNarrow &n) {
    L.r0[li] = n.r0, L.r1[li] = n.r1, L.z0[li] = n.z0, L.z1[li] = n.z1;
    L.pods[li] = ((uint32_t)n.a_pods << 16) | (uint32_t)n.npods;
}

// wave 0 combines the per-wave partials of a block reduction (one LDS read per lane + log2(waves) shuffles; a loop in
// thread 0 was a chain of 16 dependent LDS reads per word: 1.6 us per level)
__device__ __forceinline__ unsigned long long comb_max(const unsigned long long *arr) {
    return wave_max_u64((threadIdx.x & 63) < kPWaves ? arr[threadIdx.x & 63] : 0ull);
}
__device__ __forceinline__ unsigned long long comb_add(const unsigned long long *arr) {
    return (unsigned long long)wave_sum_i64((int64_t)((threadIdx.x & 63) < kPWaves ? arr[threadIdx.x & 63] : 0ull));
}

// Event keys: (0x10000 - level) << 24 | node (0 = none).  Their maximum is the LOWEST level and, of its nodes, the LAST: where a
// normalization maximum loses its last feasible holder.  Of the two maxima the event that comes first in canonical order counts:
// the higher level, then the lower node.
__device__ __forceinline__ unsigned long long event_key(int32_t level, int64_t node) { return ((unsigned long long)(0x10000 - level) << 24) | (unsigned long long)node; }
__device__ __forceinline__ void pick_event(unsigned long long kmt, unsigned long long kma, bool with_node, int32_t &ev_level, int64_t &ev_cut) {
    ev_level = -1, ev_cut = -1;
    if (kmt) ev_level = 0x10000 - (int32_t)(kmt >> 24), ev_cut = (int64_t)(kmt & 0xffffffull);
    if (kma) {
        const int32_t lv = 0x10000 - (int32_t)(kma >> 24);
        const int64_t ct = (int64_t)(kma & 0xffffffull);
        if (lv > ev_level || (lv == ev_level && ct < ev_cut)) ev_level = lv, ev_cut = ct;
    }
    if (!with_node) ev_cut = -1;
}

struct PBlockRed { // per-wave partials of a block reduction
    unsigned long long mx[5][kPWaves]; // max-words 0, 3, 4, 5, 6
    unsigned long long ad[2][kPWaves]; // add-words 1, 2
};

// FitError diagnosis slots (ccsim_kernels.h k_hist): the persistent kernel's epilogue fills the same histogram from the state in LDS
constexpr int kPHistSlots = 4 + kMaxRes + 2 + 3 + 1 + 1;

// MB = false: one device, the local grid reduce.  MB = true: several ranks (devices, or virtual ranks inside this grid) joined by
// mailboxes; the final state goes to the commit rows instead of the columns (published by k_rows_flush once every rank agrees).
template <int K, bool MB>
__global__ __launch_bounds__(kPThreads) void k_level_persist(PersistArgs a) {
    __shared__ PersistLds<K> L;
    __shared__ PBlockRed R;
    __shared__ unsigned long long s_v[8], s_red[8];
    __shared__ unsigned long long s_rk[MB ? kPMaxRanks * 8 : 1], s_mbox[MB ? kPMaxRanks : 1];
    __shared__ int s_err, s_n, s_flag;
    __shared__ int s_cnt[K][kPWaves], s_off[K][kPWaves];
    __shared__ long long s_scan[kPWaves];
    __shared__ unsigned int s_hist[kPHistSlots];

    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63; // (wave: uniform, and known to the compiler as such)
    // ranks: workgroup b of a grid with virtual ranks belongs to rank b / bpr; a real rank owns its whole grid
    int my_rank = 0, lb = (int)blockIdx.x, lgrid = (int)gridDim.x;
    if (MB) {
        my_rank = a.rank;
        if (a.vranks > 0) {
            my_rank = (int)blockIdx.x / a.bpr;
            lb = (int)blockIdx.x - my_rank * a.bpr;
            lgrid = (int)gridDim.x - my_rank * a.bpr < a.bpr ? (int)gridDim.x - my_rank * a.bpr : a.bpr;
        }
    }
    PersistSync *const sync = a.sync + (MB && a.vranks > 0 ? my_rank : 0);
    const int64_t base = (int64_t)blockIdx.x * K * kPThreads; // first node of this workgroup (index into this device's columns)
    const RunCtx cx{a.p, narrow_pod(a.p, a.c.mem_shift)};
    const uint64_t lt_mask = lane ? (~0ull >> (64 - lane)) : 0ull;

    // (no local copy of the 450-byte DevState: it would live in scratch)
    if (a.st->done) return;
    const int64_t log_cap = a.st->log_cap;
    GridCtx gc;
    gc.s = sync, gc.gen_no = 0, gc.ngroups = (unsigned)lgrid < (unsigned)kPGroups ? (unsigned)lgrid : (unsigned)kPGroups;
    gc.g = (unsigned)lb % gc.ngroups;
    gc.gsize = (unsigned)lgrid / gc.ngroups + (gc.g < (unsigned)lgrid % gc.ngroups ? 1u : 0u);
    gc.prev0 = gc.prev1 = 0, gc.s_v = s_v, gc.s_red = s_red, gc.s_err = &s_err, gc.fault = a.fault;
    gc.n_ranks = MB ? a.n_ranks : 1, gc.rank = my_rank, gc.tag_base = a.tag_base, gc.mbox = (PersistMailbox *const *)s_mbox, gc.s_rk = s_rk, gc.s_flag = &s_flag;
    if (tid == 0) s_err = 0, s_n = 0, s_flag = 0;
    if (tid < 8) s_v[tid] = 0;
    if (MB && tid < kPMaxRanks) s_mbox[tid] = (unsigned long long)(tid < a.n_ranks ? a.mbox[tid] : nullptr);

    unsigned long long pf[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, t_prev = __builtin_amdgcn_s_memrealtime();
#define PTICK(i) do { if (a.prof) { const unsigned long long t_now = __builtin_amdgcn_s_memrealtime(); pf[i] += t_now - t_prev; t_prev = t_now; } } while (0)

    // ---- load: narrow state -> LDS (the only bulk HBM read of the run).  After a ccsim_reset_state the pristine wide columns are
    // read directly (and narrowed here): the restore pass of the step frame is this load --------------------------------------------
    {
        const int sh = a.c.mem_shift;
#pragma unroll
        for (int k = 0; k < K; k++) {
            const int li = k * kPThreads + tid;
            const int64_t i = base + li;
            const bool in = i < a.c.n_pad;
            L.a0[li] = in ? a.c.a32[0][i] : 0, L.a1[li] = in ? a.c.a32[1][i] : 0;
            int32_t r0 = 0, r1 = 0, z0 = 0, z1 = 0, pc = 0;
            if (in) {
                if (a.c.from_pristine) {
                    r0 = (int32_t)a.c.p_req[0][i], r1 = (int32_t)(a.c.p_req[1][i] >> sh);
                    z0 = (int32_t)a.c.p_nz[0][i], z1 = (int32_t)(a.c.p_nz[1][i] >> sh);
                    pc = a.c.p_pod_count[i];
                } else {
                    r0 = a.c.r32[0][i], r1 = a.c.r32[1][i], z0 = a.c.z32[0][i], z1 = a.c.z32[1][i];
                    pc = a.c.pod_count[i];
                }
            }
            L.r0[li] = r0, L.r1[li] = r1, L.z0[li] = z0, L.z1[li] = z1;
            L.pods[li] = in ? (((uint32_t)a.c.alloc_pods[i] << 16) | ((uint32_t)pc & 0xffffu)) : 0u;
            L.ws[li] = 0, L.sct[li] = kScInf;
        }
    }
    __syncthreads();
    PTICK(7);

    // replicated run state (identical in every thread of every workgroup)
    int64_t placed = a.st->placed, rounds = a.st->rounds, scans = a.st->scans;
    const int64_t limit = a.st->limit;
    const bool want_log = a.want_log != 0;
    uint32_t mt = 0, ma = 0;
    int64_t c_mt = 0, c_ma = 0, nfeas = 0;
    int32_t M = 0, last_feasible = a.st->last_feasible;
    int done = 0;
    bool rescore = true, ordered = want_log;
    bool first_score = true; // the first re-score computes every node's score; the later ones (new normalization maxima) only swap the static part
    bool have_max = false;   // the new maxima came with the reduce that found the event (words 3, 4): no reduce of their own
    int kb = a.level_batch; // levels the fast path resolves per sync: halved when a batch had to be rolled back, doubled after a clean one
    int64_t last_k = 1, last_xmt = 0, last_xma = 0; // the last clean pass: levels resolved, holders of the normalization maxima it exhausted
    // A blind batch that exhausted every holder of a normalization maximum is rolled back.  Where was the event?  Every holder that
    // filled up reports the score it had before its last clone; the lowest of them is (the scores falling along a run-down) the level at
    // which the LAST holder went, `ev_level`: the levels above it are redone as one blind batch, that level in canonical order.
    // Only a guess for speed -- the validation decides again, and halving remains the fallback (round 2 halved from the start:
    // 14 of 33 iterations of a C4 run were rolled-back attempts).
    int32_t ev_level = -1;
    // ... and WHICH node: the prediction (and a rolled-back batch's report) carries the index of the holder that goes last, so a blind
    // batch may END at the event -- the levels above ev_level, and of level ev_level the nodes up to ev_cut, exactly what the reference
    // places before its normalization constants change -- instead of stopping above it and taking the level in canonical order (two
    // more grid-wide syncs per event).  Validated like every batch: the holders that filled up report (level, index) of the last one.
    int64_t ev_cut = -1;
    const int64_t gbase = a.c.global_offset + base; // canonical index of this workgroup's first node
    bool hint = a.hint_valid != 0; // the first re-score assumes the maxima the last launch started with
    bool first_consts = true;

    while (!done && (int)gc.gen_no < a.max_syncs) {
        if (rescore) {
            ev_level = -1, ev_cut = -1; // (a level of the old score scale)
            // ---- normalization maxima over the feasible set (P/helper/normalize_score.go:28-56).  First time: every node's raw static
            // word comes from HBM (all K loads of a thread in flight together -- one after the other they were K dependent L2 round
            // trips) and its Fit verdict from the state.  Later: the pass that found the event left the raw words in `ws`, the dynamic
            // part of every feasible node's score in `sct`, and brought the new maxima with its reduce.
            if (hint) mt = (uint32_t)a.hint_mt, ma = (uint32_t)a.hint_ma; // (verified below, with the scores' reduce)
            else if (!have_max) {
                uint32_t wst[K];
#pragma unroll
                for (int k = 0; k < K; k++) {
                    const int64_t i = base + k * kPThreads + tid;
                    wst[k] = i < a.c.n_pad ? a.c.stat[i] : 0u;
                }
                uint32_t lmt = 0, lma = 0;
#pragma unroll
                for (int k = 0; k < K; k++) {
                    const int li = k * kPThreads + tid;
                    const uint32_t w = wst[k];
                    NodeNarrow n = p_load_node<K>(L, li);
                    n.w = w;
                    if (nd_feasible(cx, n)) {
                        const uint32_t cnt = (w >> kStatCntShift) & kStatCntMask, aff = w & kStatAffMask;
                        lmt = cnt > lmt ? cnt : lmt, lma = aff > lma ? aff : lma;
                    }
                }
                lmt = wave_max_u32(lmt), lma = wave_max_u32(lma);
                if (lane == 0) R.mx[0][wave] = lmt, R.mx[1][wave] = lma;
                __syncthreads();
                if (wave == 0) {
                    const unsigned long long m0 = comb_max(R.mx[0]), m1 = comb_max(R.mx[1]);
                    if (lane == 0) s_v[0] = m0, s_v[3] = m1; // 0 contributes nothing, and 0 is the neutral result
                }
                grid_reduce<MB>(gc);
                if (uni32(s_err)) break;
                mt = (uint32_t)uni64(s_red[0]), ma = (uint32_t)uni64(s_red[3]);
            }
            have_max = false;
            PTICK(8);
            // ---- every node's TotalScore under (mt, ma).  The first time the resource scores are evaluated (resource_allocation.go);
            // later only the static part is swapped: TotalScore - old static part (the low half of `ws`) + new static part -- the
            // resource scores do not move when the normalization constants do.
            uint32_t lmax = 0, lnf = 0, lcmt = 0, lcma = 0, lbad = 0; // lmax: score + 1; lbad: feasible nodes above an ASSUMED maximum
            uint32_t wst[K];
#pragma unroll
            for (int k = 0; k < K; k++) { // the raw static words: all K loads of a thread in flight together
                const int64_t i = base + k * kPThreads + tid;
                wst[k] = i < a.c.n_pad ? a.c.stat[i] : 0u;
            }
#pragma unroll
            for (int k = 0; k < K; k++) {
                const int li = k * kPThreads + tid;
                const uint32_t w = wst[k];
                uint32_t sc = kScInf, wsv = w & (1u << kStatOkBit);
                uint32_t dyn = 0;
                bool feas;
                if (first_score) {
                    NodeNarrow n = p_load_node<K>(L, li);
                    n.w = w;
                    feas = nd_feasible(cx, n);
                    if (feas) dyn = (uint32_t)dynamic_score_narrow(a.p, cx.q, n.a0, n.a1, n.r0, n.r1, n.z0, n.z1);
                } else {
                    const uint32_t old = L.sct[li] & 0xffffu;
                    feas = old != kScInf;
                    dyn = old - (L.ws[li] & 0xffffu);
                }
                bool holder = false;
                if (feas) {
                    const uint32_t cnt = (w >> kStatCntShift) & kStatCntMask, aff = w & kStatAffMask, img = (w >> kStatImgShift) & kStatImgMask;
                    const uint32_t nstat = (uint32_t)static_score(a.p, cnt, aff, img, mt, ma);
                    sc = nstat + dyn;
                    lmax = sc + 1 > lmax ? sc + 1 : lmax;
                    lnf++, lcmt += cnt == mt, lcma += aff == ma;
                    lbad += (cnt > mt || aff > ma) ? 1u : 0u;
                    wsv |= (cnt == mt ? 1u << 30 : 0u) | (aff == ma ? 1u << 29 : 0u) | nstat;
                    holder = !want_log && ((mt > 0 && cnt == mt) || (ma > 0 && aff == ma));
                } // (a node the Fit filter rejects never becomes feasible again: placements only add pods)
                L.ws[li] = wsv;
                L.sct[li] = sc;
                // the holders of a maximum go to the work list: their event prediction below runs densely, not under a divergent branch
                const uint64_t b = __ballot(holder);
                if (b) {
                    int wbase = 0;
                    if (lane == 0) wbase = atomicAdd(&s_n, __popcll(b));
                    wbase = __builtin_amdgcn_readfirstlane(wbase);
                    if (holder) L.list[wbase + __popcll(b & lt_mask)] = (uint16_t)li;
                }
            }
            __syncthreads();
            // Where will these constants end?  When the last feasible holder of a maximum fills up -- and a node's run-down depends on
            // nothing but the node: every holder evaluates, once, the score it will have before the clone that fills it (the Fit filter's
            // capacity in closed form, fit.go:564-615); the lowest of them per maximum is the level of that event, the higher of the two
            // the first one.  The batches then stop above it and take that level in canonical order without a failed attempt first (a
            // guess for speed like `ev_level` after a roll-back: every batch is validated).
            unsigned long long pl_mt = 0, pl_ma = 0; // max of (0x10000 - predicted level) << 24 | node: the LOWEST level, of its holders the LAST
            {
                const int total_h = uni32(s_n);
#pragma unroll 1
                for (int r0 = tid; r0 < total_h; r0 += kPThreads) {
                    const int li = L.list[r0];
                    const NodeNarrow n = p_load_node<K>(L, li);
                    int32_t room = n.a_pods - n.npods; // clones until the node is full (>= 1: it is feasible)
                    if (!a.p.all_zero_req) {
                        if (cx.q.req0 > 0) room = (n.a0 - n.r0) / cx.q.req0 < room ? (n.a0 - n.r0) / cx.q.req0 : room;
                        if (cx.q.req1 > 0) room = (n.a1 - n.r1) / cx.q.req1 < room ? (n.a1 - n.r1) / cx.q.req1 : room;
                    }
                    NodeNarrow q = n;
                    nd_apply(cx, q, (int64_t)(room - 1));
                    const uint32_t sp = (n.w & 0xffffu) + (uint32_t)dynamic_score_narrow(a.p, cx.q, q.a0, q.a1, q.r0, q.r1, q.z0, q.z1);
                    const unsigned long long key = ((unsigned long long)(0x10000u - sp) << 24) | (unsigned long long)(a.spec_cut ? gbase + li : 0);
                    if (mt > 0 && (n.w >> 30 & 1u)) pl_mt = key > pl_mt ? key : pl_mt;
                    if (ma > 0 && (n.w >> 29 & 1u)) pl_ma = key > pl_ma ? key : pl_ma;
                }
            }
            lmax = wave_max_u32(lmax);
            lnf = wave_sum_u32(lnf), lcmt = wave_sum_u32(lcmt), lcma = wave_sum_u32(lcma), lbad = wave_sum_u32(lbad);
            pl_mt = wave_max_u64(pl_mt), pl_ma = wave_max_u64(pl_ma);
            if (lane == 0) {
                R.mx[0][wave] = lmax, R.ad[0][wave] = (unsigned long long)lnf | ((unsigned long long)lcmt << 32);
                R.ad[1][wave] = (unsigned long long)lcma | ((unsigned long long)lbad << 32);
                R.mx[1][wave] = pl_mt, R.mx[2][wave] = pl_ma;
            }
            __syncthreads();
            if (wave == 0) {
                const unsigned long long v0 = comb_max(R.mx[0]), v1 = comb_add(R.ad[0]), v2 = comb_add(R.ad[1]);
                const unsigned long long v3 = comb_max(R.mx[1]), v4 = comb_max(R.mx[2]);
                if (lane == 0) {
                    s_v[0] = v0, s_v[1] = v1, s_v[2] = v2, s_v[3] = v3, s_v[4] = v4;
                    s_n = 0; // (the work list is the level's again)
                }
            }
            grid_reduce<MB>(gc);
            if (uni32(s_err)) break;
            nfeas = (int64_t)(uni64(s_red[1]) & 0xffffffffull), c_mt = (int64_t)(uni64(s_red[1]) >> 32), c_ma = (int64_t)(uni64(s_red[2]) & 0xffffffffull);
            if (hint) { // the assumed maxima hold iff no feasible node lies above them and each has a feasible holder
                hint = false;
                if ((uni64(s_red[2]) >> 32) != 0 || (mt > 0 && c_mt == 0) || (ma > 0 && c_ma == 0)) continue; // stale: the maxima first, then the scores again
            }
            first_score = false;
            scans += 1;
            if (first_consts && lb == 0 && tid == 0) a.st->p_mt0 = (int32_t)mt, a.st->p_ma0 = (int32_t)ma; // (the launch's first constants: the next launch's hint)
            first_consts = false;
            pick_event(uni64(s_red[3]), uni64(s_red[4]), a.spec_cut != 0, ev_level, ev_cut);
            if (uni64(s_red[0]) == 0) { // schedule_one.go:448-454: no feasible node
                done = DONE_UNSCHEDULABLE, rounds += 1, last_feasible = 0;
                break;
            }
            M = (int32_t)uni64(s_red[0]) - 1;
            last_feasible = (int32_t)nfeas;
            rescore = false;
            ordered = want_log;
            PTICK(5);
            continue;
        }

        // ---- one level: nodes with TotalScore == M -------------------------------------------------------------
        // (a) every thread looks at its own K scores; level nodes go to the work list
        // The fast path resolves the levels M .. Lo in one go: every node scoring >= Lo runs down until it scores < Lo --
        // for a single node that is exactly the sequence of its run-downs at the levels in between (a run-down continues
        // while the score stays >= the level, a node below the level waits for its own), and without a log or a limit the
        // interleaving across nodes is unobservable; a normalization event or a crossed limit inside the batch is caught
        // by the validation below and the batch is redone level by level.
        // A batch that exhausts the last feasible holder of a normalization maximum is rolled back: do not try one when, at
        // the rate of the last pass, the holders would run out within twice its span (a heuristic for speed only -- the
        // validation below decides)
        bool spec = false; // this batch ends AT the event: level ev_level up to node ev_cut (per-node threshold)
        if (!ordered && ev_level >= 0 && M <= ev_level) {
            if (ev_cut >= 0 && M == ev_level) spec = true;
            else ordered = true, ev_level = -1, ev_cut = -1; // the level of the located event: in canonical order
        }
        int kcap = kb;
        if (!ordered && !spec && ev_level >= 0) {
            if (ev_cut >= 0) spec = M - ev_level + 1 <= kb;                   // the levels above the event and the event itself in one batch
            else if (M - ev_level < kcap) kcap = M - ev_level;                // ... or only the levels above it
        }
        if (mt > 0 && last_xmt > 0) {
            const int64_t lv = c_mt * last_k / last_xmt / 2;
            kcap = lv < kcap ? (lv < 1 ? 1 : (int)lv) : kcap;
        }
        if (ma > 0 && last_xma > 0) {
            const int64_t lv = c_ma * last_k / last_xma / 2;
            kcap = lv < kcap ? (lv < 1 ? 1 : (int)lv) : kcap;
        }
        const int32_t Lo = ordered ? M : spec ? ev_level : (M - (kcap - 1) > 0 ? M - (kcap - 1) : 0);
        const int64_t bcut = spec ? ev_cut : kNoCut; // nodes beyond it stop one level higher
        uint32_t mymax = 0; // score + 1 over the nodes this pass leaves alone
        int wtot = 0;
#pragma unroll
        for (int k = 0; k < K; k++) {
            const uint32_t sc = L.sct[k * kPThreads + tid] & 0xffffu;
            const bool lv = sc >= (uint32_t)Lo + (gbase + k * kPThreads + tid > bcut ? 1u : 0u) && sc != kScInf;
            if (!lv && sc != kScInf) mymax = sc + 1 > mymax ? sc + 1 : mymax;
            const int c = __popcll(__ballot(lv));
            wtot += c;
            if (ordered && lane == 0) s_cnt[k][wave] = c;
        }
        if (!ordered) { // any order will do: one LDS atomic per wave
            int wbase = 0;
            if (lane == 0 && wtot) wbase = atomicAdd(&s_n, wtot);
            wbase = __builtin_amdgcn_readfirstlane(wbase);
#pragma unroll
            for (int k = 0; k < K; k++) {
                const uint32_t sc = L.sct[k * kPThreads + tid] & 0xffffu;
                const bool lv = sc >= (uint32_t)Lo + (gbase + k * kPThreads + tid > bcut ? 1u : 0u) && sc != kScInf;
                const uint64_t b = __ballot(lv);
                if (lv) L.list[wbase + __popcll(b & lt_mask)] = (uint16_t)(k * kPThreads + tid);
                wbase += __popcll(b);
            }
            __syncthreads();
        } else { // canonical order (k-major, then thread): exclusive scan of the K x 16 wave counts by wave 0
            __syncthreads();
            if (wave == 0) {
                int run = 0;
#pragma unroll
                for (int k = 0; k < K; k++) {
                    const int c = lane < kPWaves ? s_cnt[k][lane] : 0;
                    int inc = c;
#pragma unroll
                    for (int off = 1; off < kPWaves; off <<= 1) {
                        const int o = __shfl_up(inc, off, 64);
                        if (lane >= off) inc += o;
                    }
                    if (lane < kPWaves) s_off[k][lane] = run + inc - c;
                    run += __shfl(inc, kPWaves - 1, 64);
                }
                if (lane == 0) s_n = run;
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < K; k++) {
                const uint32_t sc = L.sct[k * kPThreads + tid] & 0xffffu;
                const bool lv = sc >= (uint32_t)Lo && sc != kScInf; // (ordered: Lo == M, the maximum)
                const uint64_t b = __ballot(lv);
                if (lv) L.list[s_off[k][wave] + __popcll(b & lt_mask)] = (uint16_t)(k * kPThreads + tid);
            }
            __syncthreads();
        }
        const int total = uni32(s_n);
        PTICK(0);

        // (b) PLAN: every level node's run-down length -> the high half of its score word.  A blind batch applies it on the spot
        // (the node's new state and score: what (c) does for the ordered path after the cut is known)
        uint32_t committed = 0, x_nf = 0, x_mt = 0, x_ma = 0;
        unsigned long long xl_mt = 0, xl_ma = 0; // blind batches: event key (level = score before the last clone, node) of the holders that filled up, maximum
        {
            uint32_t T = 0, e_mt = 0, e_ma = 0;
            int64_t cmt = 0, cma = 0; // global index + 1 of the highest exhausted holder
#pragma unroll 1
            for (int r0 = 0; r0 < total; r0 += kPThreads) {
                const int nwork = total - r0 < kPThreads ? total - r0 : kPThreads;
                const int e = lane * kPWaves + wave; // entries dealt round-robin to the waves: every SIMD gets its share of run-downs
                const bool mine = e < nwork;
                NodeNarrow n;
                nd_zero(n);
                int li = 0;
                if (mine) li = L.list[r0 + e], n = p_load_node<K>(L, li);
                bool fend = true;
                int32_t j = 0;
                if (wave < nwork) j = wave_run_down_rows(cx, n, (int32_t)(n.w & 0xffffu), Lo + (gbase + li > bcut ? 1 : 0), mine, fend, a.seq_steps);
                if (mine) {
                    if (ordered) {
                        L.sct[li] = (L.sct[li] & 0xffffu) | ((uint32_t)j << 16);
                        T += (uint32_t)j;
                        if (!fend) {
                            const int64_t gi = a.c.global_offset + base + li + 1;
                            if (mt > 0 && (n.w >> 30 & 1u)) e_mt++, cmt = gi > cmt ? gi : cmt;
                            if (ma > 0 && (n.w >> 29 & 1u)) e_ma++, cma = gi > cma ? gi : cma;
                        }
                    } else {
                        if (j > 0) nd_apply(cx, n, j), p_store_dyn<K>(L, li, n);
                        const bool f = nd_feasible(cx, n);
                        const uint32_t s = f ? (uint32_t)nd_score(cx, n, (int64_t)(n.w & 0xffffu), NoRcp{}) : kScInf;
                        L.sct[li] = s | ((uint32_t)j << 16);
                        committed += (uint32_t)j;
                        if (f) mymax = s + 1 > mymax ? s + 1 : mymax;
                        else {
                            x_nf++, x_mt += n.w >> 30 & 1u, x_ma += n.w >> 29 & 1u;
                            if (j > 0 && (n.w >> 29 & 3u)) { // a holder filled up in a blind batch: its score before the last clone
                                NodeNarrow q = n;
                                nd_apply(cx, q, -1);
                                const uint32_t sp = (uint32_t)nd_score(cx, q, (int64_t)(q.w & 0xffffu), NoRcp{});
                                const unsigned long long key = event_key((int32_t)sp, a.spec_cut ? gbase + li : 0);
                                if (n.w >> 30 & 1u) xl_mt = key > xl_mt ? key : xl_mt; // (the lowest level, of its nodes the last)
                                if (n.w >> 29 & 1u) xl_ma = key > xl_ma ? key : xl_ma;
                            }
                        }
                    }
                }
            }
            if (ordered) {
                T = wave_sum_u32(T), e_mt = wave_sum_u32(e_mt), e_ma = wave_sum_u32(e_ma);
                cmt = wave_max_i64(cmt), cma = wave_max_i64(cma);
                if (lane == 0) {
                    R.ad[0][wave] = (unsigned long long)T | ((unsigned long long)e_mt << 32), R.ad[1][wave] = e_ma;
                    R.mx[1][wave] = (unsigned long long)cmt, R.mx[2][wave] = (unsigned long long)cma;
                }
            }
        }
        PTICK(1);
        int64_t cut = kNoCut, remaining = kNoCut, prefix_b = 0;
        if (ordered) {
            __syncthreads();
            if (wave == 0) {
                const unsigned long long v1 = comb_add(R.ad[0]), v2 = comb_add(R.ad[1]), v3 = comb_max(R.mx[1]), v4 = comb_max(R.mx[2]);
                if (lane == 0) {
                    s_v[1] = v1, s_v[2] = v2, s_v[3] = v3, s_v[4] = v4;
                    __hip_atomic_store(&sync->blockT[lb], (unsigned)(v1 & 0xffffffffull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            grid_reduce<MB>(gc); // (its s_waitcnt covers the blockT store)
            if (uni32(s_err)) break;
            const int64_t ge_mt = (int64_t)(uni64(s_red[1]) >> 32), ge_ma = (int64_t)uni64(s_red[2]);
            if (mt > 0 && ge_mt == c_mt && (int64_t)uni64(s_red[3]) - 1 < cut) cut = (int64_t)uni64(s_red[3]) - 1; // every feasible holder exhausted
            if (ma > 0 && ge_ma == c_ma && (int64_t)uni64(s_red[4]) - 1 < cut) cut = (int64_t)uni64(s_red[4]) - 1;
            remaining = limit > 0 ? limit - placed : kNoCut;
            // placements of this level that belong to lower workgroups (canonical order == workgroup order; over ranks: rank order)
            long long pb = 0;
            for (int b = tid; b < lb; b += kPThreads) pb += (long long)p_ld_u32(&sync->blockT[b]);
            if (MB && tid < my_rank) pb += (long long)(s_rk[tid * 8 + 1] & 0xffffffffull); // the lower ranks' planned placements of this level
            pb = wave_sum_i64(pb);
            if (lane == 0) s_scan[wave] = pb;
            __syncthreads();
            for (int w = 0; w < kPWaves; w++) prefix_b += (int64_t)uni64((unsigned long long)s_scan[w]);
        } else
            __syncthreads(); // the new states and scores are in LDS

        // (c) APPLY (ordered path): rewrite the level's nodes up to the cut / the limit in canonical order, re-score them
        if (ordered) {
            int64_t carry = prefix_b;
#pragma unroll 1
            for (int r0 = 0; r0 < total; r0 += kPThreads) {
                const int nwork = total - r0 < kPThreads ? total - r0 : kPThreads;
                const bool mine = tid < nwork;
                int li = 0;
                int32_t took = 0;
                if (mine) li = L.list[r0 + tid], took = (int32_t)(L.sct[li] >> 16);
                {
                    const int32_t j = took;
                    const int64_t incl = wave_incl_scan_i64(j);
                    __syncthreads(); // s_scan reuse across rounds
                    if (lane == 63) s_scan[wave] = incl;
                    __syncthreads();
                    int64_t before = 0, tot = 0;
#pragma unroll
                    for (int w = 0; w < kPWaves; w++) {
                        if (w < wave) before += s_scan[w];
                        tot += s_scan[w];
                    }
                    const int64_t pos = carry + before + incl - j;
                    carry += tot;
                    const int64_t gi = a.c.global_offset + base + li;
                    int64_t allowed = remaining - pos;
                    allowed = allowed < 0 ? 0 : allowed;
                    took = (mine && gi <= cut) ? (int32_t)(j < allowed ? j : allowed) : 0;
                    if (a.log && took > 0)
                        for (int32_t q = 0; q < took; q++) {
                            const int64_t at = placed + pos + q;
                            if (at < log_cap) a.log[at] = (int32_t)gi;
                        }
                }
                if (mine) {
                    NodeNarrow n = p_load_node<K>(L, li);
                    if (took > 0) nd_apply(cx, n, took), p_store_dyn<K>(L, li, n);
                    const bool f = nd_feasible(cx, n);
                    const uint32_t s = f ? (uint32_t)nd_score(cx, n, (int64_t)(n.w & 0xffffu), NoRcp{}) : kScInf;
                    L.sct[li] = s | ((uint32_t)took << 16);
                    committed += (uint32_t)took;
                    if (f) mymax = s + 1 > mymax ? s + 1 : mymax;
                    else x_nf++, x_mt += n.w >> 30 & 1u, x_ma += n.w >> 29 & 1u;
                }
            }
        }
        // A pass that (probably) ends at a normalization event is followed by new constants: their maxima -- over the nodes still
        // feasible AFTER this pass -- ride on its reduce (words 5, 6) instead of a reduce of their own.
        uint32_t nx_mt = 0, nx_ma = 0;
        const bool with_max = spec || (ordered && cut != kNoCut);
        if (with_max) {
            __syncthreads(); // (the scores this pass rewrote)
            uint32_t wst[K];
#pragma unroll
            for (int k = 0; k < K; k++) {
                const int64_t i = base + k * kPThreads + tid;
                wst[k] = i < a.c.n_pad ? a.c.stat[i] : 0u;
            }
#pragma unroll
            for (int k = 0; k < K; k++)
                if ((L.sct[k * kPThreads + tid] & 0xffffu) != kScInf) {
                    const uint32_t cnt = (wst[k] >> kStatCntShift) & kStatCntMask, aff = wst[k] & kStatAffMask;
                    nx_mt = cnt > nx_mt ? cnt : nx_mt, nx_ma = aff > nx_ma ? aff : nx_ma;
                }
        }
        PTICK(2);
        // (d) block reduction -> grid reduction
        mymax = wave_max_u32(mymax);
        committed = wave_sum_u32(committed), x_nf = wave_sum_u32(x_nf), x_mt = wave_sum_u32(x_mt), x_ma = wave_sum_u32(x_ma);
        xl_mt = wave_max_u64(xl_mt), xl_ma = wave_max_u64(xl_ma);
        if (with_max) nx_mt = wave_max_u32(nx_mt), nx_ma = wave_max_u32(nx_ma);
        if (lane == 0) {
            R.mx[0][wave] = mymax;
            R.mx[1][wave] = xl_mt, R.mx[2][wave] = xl_ma, R.mx[3][wave] = nx_mt, R.mx[4][wave] = nx_ma;
            R.ad[0][wave] = (unsigned long long)committed | ((unsigned long long)x_nf << 32);
            R.ad[1][wave] = (unsigned long long)x_mt | ((unsigned long long)x_ma << 32);
        }
        __syncthreads();
        if (wave == 0) {
            const unsigned long long v0 = comb_max(R.mx[0]), v1 = comb_add(R.ad[0]), v2 = comb_add(R.ad[1]);
            const unsigned long long v3 = comb_max(R.mx[1]), v4 = comb_max(R.mx[2]), v5 = comb_max(R.mx[3]), v6 = comb_max(R.mx[4]);
            if (lane == 0) {
                s_v[0] = v0, s_v[1] = v1, s_v[2] = v2, s_v[3] = v3, s_v[4] = v4, s_v[5] = v5, s_v[6] = v6;
                s_n = 0; // next level's list (every thread read `total` before the barrier above)
            }
        }
        PTICK(3);
        grid_reduce<MB>(gc);
        PTICK(4);
        pf[6] += 1;
        if (uni32(s_err)) break;
        const int64_t g_committed = (int64_t)(uni64(s_red[1]) & 0xffffffffull), g_xnf = (int64_t)(uni64(s_red[1]) >> 32);
        const int64_t g_xmt = (int64_t)(uni64(s_red[2]) & 0xffffffffull), g_xma = (int64_t)(uni64(s_red[2]) >> 32);
        const uint32_t g_next = (uint32_t)uni64(s_red[0]);

        bool event_done = ordered && cut != kNoCut; // this pass ended exactly where a normalization maximum lost its last feasible holder
        if (!ordered) {
            // validate the blind commit: did it exhaust every holder of a normalization maximum, or cross the limit?
            const bool ex_mt = mt > 0 && g_xmt == c_mt, ex_ma = ma > 0 && g_xma == c_ma;
            const bool cut_event = ex_mt || ex_ma;
            const bool over = limit > 0 && placed + g_committed > limit;
            if (spec && cut_event && !over) {
                // a batch cut at the predicted event stands iff every maximum that ran out of holders did so exactly there: the
                // holders that filled up report the (level, node) of the last one.  Anything else is rolled back like any batch.
                const unsigned long long want = event_key(ev_level, ev_cut);
                event_done = (!ex_mt || uni64(s_red[3]) == want) && (!ex_ma || uni64(s_red[4]) == want);
            }
            if ((cut_event && !event_done) || over) { // undo the whole level, redo it in canonical order
#pragma unroll 1
                for (int r0 = 0; r0 < total; r0 += kPThreads)
                    if (r0 + tid < total) {
                        const int li = L.list[r0 + tid];
                        const int32_t tk = (int32_t)(L.sct[li] >> 16);
                        NodeNarrow n = p_load_node<K>(L, li);
                        nd_apply(cx, n, -(int64_t)tk);
                        p_store_dyn<K>(L, li, n);
                        L.sct[li] = (uint32_t)nd_score(cx, n, (int64_t)(n.w & 0xffffu), NoRcp{}); // (it was feasible: it took pods)
                    }
                __syncthreads();
                if (Lo < M || spec) {
                    kb = (M - Lo + 1) >> 1; // a batch: retry with half the levels, still blind (the event is somewhere inside) ...
                    kb = kb < 1 ? 1 : kb;
                    if (cut_event && !over) { // ... unless the holders that filled up say where
                        const int32_t old_level = ev_level;
                        const int64_t old_cut = spec ? ev_cut : -1;
                        ev_level = -1, ev_cut = -1; // (whatever was predicted did not hold)
                        int32_t ev = -1;
                        int64_t ec = -1;
                        pick_event(ex_mt ? uni64(s_red[3]) : 0ull, ex_ma ? uni64(s_red[4]) : 0ull, a.spec_cut != 0, ev, ec);
                        if (ev >= Lo && ev <= M) {
                            ev_level = ev, ev_cut = ec, kb = a.level_batch; // (the batch is cut at the event where Lo is chosen)
                            if (spec && ev == old_level && ec == old_cut) ev_cut = -1; // the same place again: that level in canonical order
                        }
                    }
                    if (spec && Lo == M && (ev_level < 0 || over)) ordered = true; // (one level, nothing learnt / the limit inside it: canonical order)
                } else
                    ordered = true; // one level: redo it in canonical order
                continue;
            }
            if (spec) ev_level = -1, ev_cut = -1; // the prediction is spent (the event happened, or the batch ended short of it)
        }
        placed += g_committed, rounds += g_committed;
        if (!ordered) kb = 2 * kb < a.level_batch ? 2 * kb : a.level_batch;
        last_k = M - Lo + 1, last_xmt = g_xmt, last_xma = g_xma;
        nfeas -= g_xnf, c_mt -= g_xmt, c_ma -= g_xma;
        scans += 1;
        if (event_done) { // a normalization maximum lost its last feasible holder: new constants (their maxima came with this reduce)
            mt = (uint32_t)uni64(s_red[5]), ma = (uint32_t)uni64(s_red[6]);
            have_max = true;
        }
        if (limit > 0 && placed >= limit) { // simulator.go:297-312: tested after the append
            done = DONE_LIMIT;
            break;
        }
        if (event_done) {
            // The event took the last feasible node with it (the run's last holder of a maximum is the run's last node: C4 ends this way):
            // nothing is left to re-score -- `nfeas` is exact (the scores' reduce counted the feasible nodes, every validated batch took off
            // the ones it filled), so the launch ends here instead of after one more re-score pass and its grid-wide reduce (round 6: 7 -> 6
            // syncs per C4 run).  schedule_one.go:448-454: no feasible node -> FitError.
            if (nfeas == 0 && a.end_at_empty) {
                done = DONE_UNSCHEDULABLE, rounds += 1, last_feasible = 0;
                break;
            }
            rescore = true;
            ordered = want_log;
            continue;
        }
        ordered = want_log;
        if (g_next == 0) {
            done = DONE_UNSCHEDULABLE, rounds += 1, last_feasible = 0;
            break;
        }
        last_feasible = (int32_t)nfeas;
        M = (int32_t)g_next - 1;
    }

    // ---- the end of the launch ------------------------------------------------------------------------------------------------------
    __syncthreads();
    // A grid barrier that timed out (a workgroup was not resident: CU mask, a shared GPU) leaves this launch's levels half
    // committed across the grid.  Nothing is written then: the columns still hold the state the launch started from, and the
    // host continues on the multi-kernel path.  A generation is released or abandoned by compare-and-swap on ONE word,
    // so every workgroup leaves its last barrier the same way: either all write or none does.  The mailbox form writes the
    // commit rows, which nobody reads until every rank has reported success to the host.
    if (s_err || p_ld_u32(&sync->err[0])) {
        if (lb == 0 && tid == 0) a.st->done = DONE_ERROR;
        if (MB && tid == 0 && a.ok_flag) __hip_atomic_store(a.ok_flag, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    if (MB && tid == 0 && a.ok_flag && !done) __hip_atomic_store(a.ok_flag, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // (out of syncs: not a finished run)
    PTICK(9);
    const int sh = a.c.mem_shift;
    const bool diag = !MB && done == DONE_UNSCHEDULABLE && a.c.hist != nullptr;
    unsigned int *const sh_ts = reinterpret_cast<unsigned int *>(L.list); // (the work list is free now) taint-set bins: K * 256 of them
    constexpr int kTsBins = K * kPThreads / 2;
    if (diag) {
        __syncthreads();
        for (int i = tid; i < kPHistSlots; i += kPThreads) s_hist[i] = 0;
        for (int i = tid; i < kTsBins; i += kPThreads) sh_ts[i] = 0;
        __syncthreads();
    }
#pragma unroll 1
    for (int k = 0; k < K; k++) {
        const int li = k * kPThreads + tid;
        const int64_t i = base + li;
        if (i >= a.c.n_pad) continue;
        const int32_t r0 = L.r0[li], r1 = L.r1[li], z0 = L.z0[li], z1 = L.z1[li];
        const int32_t np = (int32_t)(L.pods[li] & 0xffffu), np0 = a.c.from_pristine ? a.c.p_pod_count[i] : a.c.pod_count[i];
        if (MB) { // commit rows (ccsim_kernels.h DevCols::rows): r0 r1 z0 z1 | pods placed - -
            int4 *row = reinterpret_cast<int4 *>(a.c.rows + i * kRowWords);
            row[1] = make_int4(r0, r1, z0, z1);
            row[2] = make_int4(np, np - np0 + (a.c.cnt_assign ? 0 : a.c.placed_cnt[i]), 0, 0);
            continue;
        }
        a.c.r32[0][i] = r0, a.c.r32[1][i] = r1, a.c.z32[0][i] = z0, a.c.z32[1][i] = z1;
        if (!a.c.skip_wide) {
            a.c.req[0][i] = (int64_t)r0, a.c.req[1][i] = (int64_t)r1 << sh;
            a.c.nz_mcpu[i] = (int64_t)z0, a.c.nz_mem[i] = (int64_t)z1 << sh;
        }
        a.c.pod_count[i] = np;
        const int32_t pc = a.c.cnt_assign ? np - np0 : a.c.placed_cnt[i] + np - np0;
        a.c.placed_cnt[i] = pc;
        if (a.c.cnt_narrow) { // (the host checked that every count fits the width)
            if (a.c.cnt_narrow_width == 1) reinterpret_cast<uint8_t *>(a.c.cnt_narrow)[i] = (uint8_t)pc;
            else reinterpret_cast<uint16_t *>(a.c.cnt_narrow)[i] = (uint16_t)pc;
        }
        if (diag && i < a.c.n) {
            // FitError diagnosis of the terminal cycle (types.go:787-836; the same bins as k_hist): first failing plugin in filter
            // order, NodeResourcesFit keeps all its reasons (fit.go:520-531).  The persistent form runs without extra resource columns,
            // host ports and topology-coupled plugins, so the node's state in LDS decides everything after the static filters.
            const uint8_t sr = a.c.sreason[i];
            if (sr == 1) atomicAdd(&s_hist[0], 1u);
            else if (sr == 2) {
                const int32_t ts = a.c.taintset_id ? a.c.taintset_id[i] : 0;
                if (ts < kTsBins) atomicAdd(&sh_ts[ts], 1u);
                else atomicAdd(&a.c.hist_ts[ts], 1ull);
            } else if (sr == 3) atomicAdd(&s_hist[2], 1u);
            else if (sr == 4) atomicAdd(&s_hist[4 + kMaxRes + 2 + 3], 1u), atomicAdd(&s_hist[kPHistSlots - 1], 1u);
            else {
                bool any = false, unresolvable = false;
                const int32_t a_pods = (int32_t)(L.pods[li] >> 16);
                if (a.p.fit_enabled && np + 1 > a_pods) atomicAdd(&s_hist[3], 1u), any = true;
                if (a.p.fit_enabled && !a.p.all_zero_req) {
                    const int32_t al0 = L.a0[li], al1 = L.a1[li];
                    if (cx.q.req0 > 0 && cx.q.req0 > al0 - r0) atomicAdd(&s_hist[4], 1u), any = true, unresolvable = unresolvable || cx.q.req0 > al0;
                    if (cx.q.req1 > 0 && cx.q.req1 > al1 - r1) atomicAdd(&s_hist[5], 1u), any = true, unresolvable = unresolvable || cx.q.req1 > al1;
                }
                if (any && !unresolvable) atomicAdd(&s_hist[kPHistSlots - 1], 1u);
            }
        }
    }
    if (diag) {
        __syncthreads();
        for (int i = tid; i < kPHistSlots - 1; i += kPThreads)
            if (s_hist[i]) atomicAdd(&a.c.hist[i], (unsigned long long)s_hist[i]);
        if (tid == 0 && s_hist[kPHistSlots - 1]) atomicAdd(&a.c.hist_code[0], (unsigned long long)s_hist[kPHistSlots - 1]);
        for (int i = tid; i < kTsBins && i < a.c.n_taintsets; i += kPThreads)
            if (sh_ts[i]) atomicAdd(&a.c.hist_ts[i], (unsigned long long)sh_ts[i]);
    }
    PTICK(10);
    if (lb == 0 && tid == 0 && (!MB || a.vranks == 0 || my_rank == 0)) {
#pragma unroll
        for (int i = 0; i < 12; i++) sync->prof[i] = pf[i];
        DevState *st = a.st;
        st->placed = placed, st->rounds = rounds, st->scans = scans;
        st->done = done;
        st->mt_a = (int32_t)mt, st->ma_a = (int32_t)ma;
        st->last_feasible = last_feasible;
        st->lvl_full = 1, st->lvl_valid = 0, st->lvl_plan_only = 0; // the multi-kernel path's score cache knows nothing of this run
        st->winner = -1;
    }
#undef PTICK
}

} // namespace ccsim
