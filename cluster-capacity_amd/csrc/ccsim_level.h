// ccsim_level.h -- CCSIM_MODE_BATCHED: exact level-batched resolution of the placement loop.
//
// The reference places ONE pod per scheduling cycle (S/schedule_one.go:66-137 driven by
// pkg/framework/simulator.go:297-381).  With one repeated pod spec and the plugin set of this engine a
// cycle changes only the winner's NodeInfo (S/framework/types.go:409-428), and the canonical
// selectHost (S/schedule_one.go:894-941, ties -> lowest position) always takes the lowest-index node
// among those holding the current maximum TotalScore M.  Therefore, as long as the normalization
// constants (max PreferNoSchedule count / max preferred-affinity sum over the feasible set,
// P/helper/normalize_score.go:28-56) do not move:
//
//   * the winner keeps winning while its own score stays >= M and it stays feasible (every other node
//     is <= M and, at == M, has a higher index)                                     -> a "run-down";
//   * then the next node holding M (in index order) is run down, and so on          -> a "level";
//   * when no node holds M any more the next level is the new maximum (< M).
//
// One pass = k_level_commit + k_level_final (one block):
//   (a) k_level_commit COMMITS the current level: every node holding M runs down independently and rewrites its own
//       state (no atomics, no cross-node traffic).  Run-downs are evaluated per lane for the first few
//       placements and WAVE-COOPERATIVELY for long ones (wave_run_down below): <= kSeqSteps + 2 dependent
//       steps for a 110-pod node instead of 110;
//   (b) the same kernel RE-SCORES exactly the nodes it changed and stores the result in the per-node score cache
//       (cscore, 4 B/node) -- a placement changes one node, every other cached score is still exact -- and reduces the
//       NEXT level from the cache: packed max key, how many nodes hold it, how many nodes the commit left infeasible
//       and how many of those held a normalization maximum;
//   (c) k_level_final reduces the per-block partials, keeps the feasible / holder counts current and decides.
// The full pods x nodes pass (k_level_score: Filter + Score of every node from its columns, same arithmetic and bytes
// as k_scan) runs only while the cache is invalid: the first pass of a run, and when a normalization maximum loses its
// last feasible holder.  A level is committed blindly unless something could end it early; then one extra PLAN pass
// (k_level_commit, no commit) measures it:
//   * the level could exhaust the last feasible holder of a normalization maximum (level size >= holder
//     count): nodes after that holder ("cut") must be re-scored with new constants -> commit up to the cut;
//   * --max-limit, or the caller wants the placement log: the commit becomes ORDERED (exclusive scan of
//     the run-down lengths: block prefixes from k_level_final + in-block scan) so that a prefix of the
//     level can be committed and every placement knows its position in the sequence.
// The placement sequence, per-node counts and terminal state are IDENTICAL to CCSIM_MODE_SEQUENTIAL
// (tests/test_level_model.py proves the argument on the CPU against the oracle; tests/test_gpu_parity.py
// checks this kernel).
//
// Roofline: the full pass is an HBM stream (60 B/node algorithmic); the commit pass is a latency chain inside one
// dispatch that moves 4 B/node (the cache) + one 64-byte row per level node.
#pragma once
#include "ccsim_kernels.h"

namespace ccsim {

constexpr int64_t kNoCut = (int64_t)1 << 62;

// per-block result of the score pass: 32 bytes
struct __attribute__((aligned(16))) LevelPartial {
    uint64_t key;        // block max: ((score+1) << 40) | (2^40-1 - global idx); 0 = nothing feasible
    uint32_t mt, ma;     // max prefer-count / affinity-sum over the block's feasible nodes
    uint32_t c_mt, c_ma; // how many feasible nodes hold those block maxima
    uint32_t nfeas;
    uint32_t n_top;      // how many nodes hold the block's maximum score
};

// one node in registers
template <int NX>
struct NodeRegs {
    int64_t a_cpu, a_mem, r_cpu, r_mem, z_cpu, z_mem;
    int64_t xa[NX > 0 ? NX : 1], xr[NX > 0 ? NX : 1];
    int32_t a_pods, npods;
    uint32_t w; // static word
};

template <int NX>
__device__ __forceinline__ bool node_feasible(const DevPod &p, const NodeRegs<NX> &n) {
    bool ok = (n.w >> kStatOkBit) && fits_core(p, n.a_cpu, n.a_mem, n.r_cpu, n.r_mem, n.a_pods, n.npods);
    if (NX > 0 && p.fit_enabled && !p.all_zero_req) {
#pragma unroll
        for (int x = 0; x < NX; x++)
            if (x < p.nx && p.req[p.xcol[x]] > 0 && p.req[p.xcol[x]] > n.xa[x] - n.xr[x]) ok = false; // (a slot held only for scoring carries no request)
    }
    return ok;
}

// NodeInfo.update (types.go:409-428) applied k times, closed form
template <int NX>
__device__ __forceinline__ void node_apply(const DevPod &p, NodeRegs<NX> &n, int64_t k) {
    n.r_cpu += k * p.req[0];
    n.r_mem += k * p.req[1];
    n.z_cpu += k * p.nz_mcpu;
    n.z_mem += k * p.nz_mem;
    n.npods += (int32_t)k;
    if (NX > 0) {
#pragma unroll
        for (int x = 0; x < NX; x++)
            if (x < p.nx) n.xr[x] += k * p.req[p.xcol[x]];
    }
}

template <int NX>
__device__ __forceinline__ int64_t node_score(const DevPod &p, const NodeRegs<NX> &n, int64_t stat, const NodeRcp &rc) {
    if (NX > 0 && p.gen_score) return stat + dynamic_score_gen<NX>(p, n.a_cpu, n.a_mem, n.r_cpu, n.r_mem, n.z_cpu, n.z_mem, n.xa, n.xr);
    return stat + dynamic_score(p, rc, n.a_cpu, n.a_mem, n.r_cpu, n.r_mem, n.z_cpu, n.z_mem);
}
template <int NX>
__device__ __forceinline__ int64_t node_score(const DevPod &p, const NodeRegs<NX> &n, int64_t stat) {
    return node_score<NX>(p, n, stat, make_rcp(n.a_cpu, n.a_mem));
}

// (src is wave-uniform at every call site: the lowest set bit of a ballot)
__device__ __forceinline__ int64_t bcast_i64(int64_t v, int src) { return lane_bcast_i64(v, src); }
__device__ __forceinline__ int32_t bcast_i32(int32_t v, int src) { return lane_bcast_i32(v, src); }

template <int NX>
__device__ __forceinline__ NodeRegs<NX> bcast_node(const NodeRegs<NX> &n, int src) {
    NodeRegs<NX> o;
    o.a_cpu = bcast_i64(n.a_cpu, src), o.a_mem = bcast_i64(n.a_mem, src);
    o.r_cpu = bcast_i64(n.r_cpu, src), o.r_mem = bcast_i64(n.r_mem, src);
    o.z_cpu = bcast_i64(n.z_cpu, src), o.z_mem = bcast_i64(n.z_mem, src);
    o.a_pods = bcast_i32(n.a_pods, src), o.npods = bcast_i32(n.npods, src);
    o.w = (uint32_t)bcast_i32((int32_t)n.w, src);
#pragma unroll
    for (int x = 0; x < (NX > 0 ? NX : 1); x++) {
        o.xa[x] = NX > 0 ? bcast_i64(n.xa[x], src) : 0;
        o.xr[x] = NX > 0 ? bcast_i64(n.xr[x], src) : 0;
    }
    return o;
}

// ------------------------------------------------------------------------------------------------
// One node of the commit pass behind a small common interface (nd_*), in two representations:
//   NodeRegs<NX>  the int64 columns (any snapshot, NX extended resources);
//   NodeNarrow    the 32-bit mirrors (DevCols::narrow: every value stays below 2^31 for the whole run, memory in the
//                 common power-of-two unit; see "NARROW arithmetic" in ccsim_kernels.h).  A run-down step costs ~5x fewer
//                 VALU instructions (no 64-bit multiply / divide emulation, no fp64), and the run-downs are what the
//                 commit pass spends its time on.
// ------------------------------------------------------------------------------------------------
struct RunCtx {
    const DevPod &p;
    NarrowPod q; // the pod in narrow units (unused by NodeRegs)
};

struct NodeNarrow {
    int32_t a0, a1, r0, r1, z0, z1, a_pods, npods;
    uint32_t w;
    int32_t placed; // pods this run has put on the node (travels in the commit row)
};
struct NoRcp {};

template <int NX> __device__ __forceinline__ bool nd_feasible(const RunCtx &cx, const NodeRegs<NX> &n) { return node_feasible<NX>(cx.p, n); }
template <int NX> __device__ __forceinline__ void nd_apply(const RunCtx &cx, NodeRegs<NX> &n, int64_t k) { node_apply<NX>(cx.p, n, k); }
template <int NX> __device__ __forceinline__ NodeRcp nd_rcp(const NodeRegs<NX> &n) { return make_rcp(n.a_cpu, n.a_mem); }
template <int NX> __device__ __forceinline__ int64_t nd_score(const RunCtx &cx, const NodeRegs<NX> &n, int64_t stat, const NodeRcp &rc) {
    return node_score<NX>(cx.p, n, stat, rc);
}
template <int NX> __device__ __forceinline__ NodeRegs<NX> nd_bcast(const NodeRegs<NX> &n, int src) { return bcast_node<NX>(n, src); }
template <int NX> __device__ __forceinline__ int64_t nd_room(const NodeRegs<NX> &) { return (int64_t)1 << 40; } // int64: no clamp needed
template <int NX> __device__ __forceinline__ uint32_t nd_word(const NodeRegs<NX> &n) { return n.w; }

__device__ __forceinline__ bool nd_feasible(const RunCtx &cx, const NodeNarrow &n) {
    return (n.w >> kStatOkBit) && fits_narrow(cx.p, cx.q, n.a0, n.a1, n.r0, n.r1, n.a_pods, n.npods);
}
__device__ __forceinline__ void nd_apply(const RunCtx &cx, NodeNarrow &n, int64_t k) {
    const int32_t kk = (int32_t)k;
    n.r0 += kk * cx.q.req0, n.r1 += kk * cx.q.req1;
    n.z0 += kk * cx.q.nz0, n.z1 += kk * cx.q.nz1;
    n.npods += kk;
}
__device__ __forceinline__ NoRcp nd_rcp(const NodeNarrow &) { return NoRcp{}; }
__device__ __forceinline__ int64_t nd_score(const RunCtx &cx, const NodeNarrow &n, int64_t stat, const NoRcp &) {
    return stat + dynamic_score_narrow(cx.p, cx.q, n.a0, n.a1, n.r0, n.r1, n.z0, n.z1);
}
__device__ __forceinline__ NodeNarrow nd_bcast(const NodeNarrow &n, int src) {
    NodeNarrow o;
    o.a0 = bcast_i32(n.a0, src), o.a1 = bcast_i32(n.a1, src), o.r0 = bcast_i32(n.r0, src), o.r1 = bcast_i32(n.r1, src);
    o.z0 = bcast_i32(n.z0, src), o.z1 = bcast_i32(n.z1, src), o.a_pods = bcast_i32(n.a_pods, src), o.npods = bcast_i32(n.npods, src);
    o.w = (uint32_t)bcast_i32((int32_t)n.w, src);
    o.placed = 0;
    return o;
}
// Placements after which the node is certainly full (NodeResourcesFit pod count): the closed-form candidates of the
// cooperative tail are clamped to it, so 32-bit state never leaves the range the narrow mode was validated for.
__device__ __forceinline__ int64_t nd_room(const NodeNarrow &n) { return n.a_pods > n.npods ? (int64_t)(n.a_pods - n.npods) : 1; }
__device__ __forceinline__ uint32_t nd_word(const NodeNarrow &n) { return n.w; }
// placements a run-down may take without evaluating the states in between (ccsim_kernels.h run_down_safe_skip; narrow form only)
template <int NX> __device__ __forceinline__ int32_t nd_skip(const RunCtx &, const NodeRegs<NX> &, int64_t, int64_t) { return 0; }
__device__ __forceinline__ int32_t nd_skip(const RunCtx &cx, const NodeNarrow &n, int64_t stat, int64_t M) {
    return run_down_safe_skip(cx.p, cx.q, n.a0, n.a1, n.r0, n.r1, n.z0, n.z1, n.a_pods, n.npods, (int32_t)stat, (int32_t)M);
}

// Run-downs, two regimes.  Every lane of the wave must call this (wave-uniform control flow).
// `mine` = this lane's node holds the level (feasible, score == M).
//   1. kSeqSteps placements evaluated by the lane itself (all level lanes of the wave in parallel): most
//      nodes leave the level after a few pods.
//   2. lanes still running (large nodes whose score moves once per tens of pods) are finished
//      WAVE-COOPERATIVELY, one node at a time: its registers are broadcast, lane l evaluates the node after
//      l+1 further placements (closed form), and one __ballot finds the first placement after which the node
//      is infeasible or scores < M: <= 2 steps for a 110-pod node instead of 110 dependent iterations.
// Returns this lane's run-down length (0 if !mine) and whether its node is still feasible afterwards.
#ifndef CCSIM_SEQ_STEPS
#define CCSIM_SEQ_STEPS 6
#endif
constexpr int kSeqSteps = CCSIM_SEQ_STEPS;

template <class Node>
__device__ __forceinline__ int32_t wave_run_down(const RunCtx &cx, const Node &n, int64_t stat, int64_t M, bool mine, bool &feas_after,
                                                 int seq_steps = kSeqSteps) {
    const int lane = threadIdx.x & 63;
    int32_t my_j = 0;
    feas_after = true;
    if (!__ballot(mine)) return 0;
    Node cur = n;
    bool running = mine;
    const auto rc = nd_rcp(n); // allocatable never changes: one reciprocal pair per node
    if (running) {
        const int32_t k = nd_skip(cx, cur, stat, M);
        if (k > 0) nd_apply(cx, cur, k), my_j = k;
    }
#pragma unroll 1
    for (int it = 0; it < seq_steps && __ballot(running); it++) {
        if (running) {
            nd_apply(cx, cur, 1);
            my_j++;
            feas_after = nd_feasible(cx, cur);
            running = feas_after && nd_score(cx, cur, stat, rc) >= M;
        }
    }
    uint64_t todo = __ballot(running);
    while (todo) {
        const int src = __ffsll((unsigned long long)todo) - 1;
        todo &= todo - 1;
        const Node base = nd_bcast(cur, src);
        const int64_t bstat = bcast_i64(stat, src);
        const auto brc = nd_rcp(base);
        const int64_t room = nd_room(base); // after `room` more placements the node is full
        int32_t j = 0;
        bool f_end = true;
        for (int32_t k0 = 0;; k0 += 64) {
            Node t = base;
            const int64_t k = (int64_t)k0 + lane + 1;
            nd_apply(cx, t, k < room ? k : room);
            const bool f = nd_feasible(cx, t); // (k >= room: the pod count alone makes it infeasible)
            const bool stop = !(f && nd_score(cx, t, bstat, brc) >= M);
            const uint64_t sm = __ballot(stop);
            if (sm) {
                const int first = __ffsll((unsigned long long)sm) - 1;
                j = k0 + first + 1;
                f_end = (__ballot(f) >> first) & 1ull;
                break;
            }
            if (k0 > (1 << 30)) { // unreachable with the Fit filter on (pod capacity bounds a run-down)
                j = k0 + 64;
                break;
            }
        }
        if (lane == src) my_j += j, feas_after = f_end;
    }
    return my_j;
}

template <int NX>
__device__ __forceinline__ void load_pair(const DevCols &c, const DevPod &p, int64_t i0, NodeRegs<NX> (&nd)[2]) {
    const uint2 sw = *reinterpret_cast<const uint2 *>(c.stat + i0);
    const Cols6 c6 = load_cols6<false>(c, i0);
    const longlong2 A0 = c6.A0, A1 = c6.A1, R0 = c6.R0, R1 = c6.R1, Z0 = c6.Z0, Z1 = c6.Z1;
    const int2 AP = *reinterpret_cast<const int2 *>(c.alloc_pods + i0);
    const int2 NP = *reinterpret_cast<const int2 *>(c.pod_count + i0);
    nd[0].w = sw.x, nd[1].w = sw.y;
    nd[0].a_cpu = A0.x, nd[1].a_cpu = A0.y;
    nd[0].a_mem = A1.x, nd[1].a_mem = A1.y;
    nd[0].r_cpu = R0.x, nd[1].r_cpu = R0.y;
    nd[0].r_mem = R1.x, nd[1].r_mem = R1.y;
    nd[0].z_cpu = Z0.x, nd[1].z_cpu = Z0.y;
    nd[0].z_mem = Z1.x, nd[1].z_mem = Z1.y;
    nd[0].a_pods = AP.x, nd[1].a_pods = AP.y;
    nd[0].npods = NP.x, nd[1].npods = NP.y;
#pragma unroll
    for (int x = 0; x < (NX > 0 ? NX : 1); x++) {
        nd[0].xa[x] = nd[1].xa[x] = nd[0].xr[x] = nd[1].xr[x] = 0;
        if (NX > 0 && x < p.nx) {
            const int col = p.xcol[x];
            const longlong2 XA = *reinterpret_cast<const longlong2 *>(c.alloc[col] + i0);
            const longlong2 XR = *reinterpret_cast<const longlong2 *>(c.req[col] + i0);
            nd[0].xa[x] = XA.x, nd[1].xa[x] = XA.y;
            nd[0].xr[x] = XR.x, nd[1].xr[x] = XR.y;
        }
    }
}

// write the dynamic columns of one node back (only nodes that took pods: sparse 8-byte stores)
template <int NX>
__device__ __forceinline__ void store_dyn(const DevCols &c, const DevPod &p, int64_t i, const NodeRegs<NX> &n, int32_t took) {
    c.req[0][i] = n.r_cpu;
    c.req[1][i] = n.r_mem;
    c.nz_mcpu[i] = n.z_cpu;
    c.nz_mem[i] = n.z_mem;
    c.pod_count[i] = n.npods;
    c.placed_cnt[i] += took;
    store_mirror(c, i, n.r_cpu, n.r_mem, n.z_cpu, n.z_mem);
    if (NX > 0) {
#pragma unroll
        for (int x = 0; x < NX; x++)
            if (x < p.nx) c.req[p.xcol[x]][i] = n.xr[x];
    }
}

__device__ __forceinline__ int64_t wave_max_i64(int64_t v) { // (values >= -1 at every call site: biased to unsigned)
    return (int64_t)(wave_max_u64((uint64_t)(v + 1))) - 1;
}
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) { return wave_sum_u32_dpp(v); }
// inclusive prefix sum across the 64 lanes of a wave
__device__ __forceinline__ int64_t wave_incl_scan_i64(int64_t v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        int64_t o = __shfl_up(v, off, 64);
        if (lane >= off) v += o;
    }
    return v;
}

constexpr int64_t kEvBase = 1ll << 40; // located events travel as kEvBase - score (maximum-combined, like the plan pass's cut indices)

struct __attribute__((aligned(16))) CommitPartial { // per k_level_commit block: 64 bytes
    int64_t committed;      // placements committed by this block in this pass
    int64_t T;              // plan pass: placements of the block's nodes at level st.lvl_M
    int64_t cut_mt, cut_ma; // plan pass: highest global index among exhausted holders of st.mt_a / st.ma_a (-1 none);
                            // blind batch: kEvBase - (lowest score a holder that filled up had before its last clone) (-1 none)
    uint32_t e_mt, e_ma;    // plan pass: how many holders their run-down exhausts
    // commit pass: the block's part of the NEXT level, from the score cache (unchanged nodes) and the re-scored level nodes
    uint64_t key;           // block max ((score+1) << 40 | (2^40-1 - global idx)); 0 = nothing feasible
    uint32_t n_top;         // nodes holding the block's maximum score
    uint32_t x_nf;          // level nodes this commit left infeasible ...
    uint32_t x_mt, x_ma;    // ... and how many of them held the normalization maxima st.mt_a / st.ma_a
};
static_assert(sizeof(CommitPartial) == 64, "CommitPartial layout");

struct LevelArgs {
    DevCols c;
    DevPod p;
    const DevState *st;
    LevelPartial *partials;     // k_level_score: one per block
    CommitPartial *cpartials;   // k_level_commit: one per block
    const int64_t *blockprefix; // [commit grid] exclusive prefix of the planned level's per-block placements
    int32_t *cscore;            // TotalScore of every node under (mt_a, ma_a), -1 = infeasible: written by the score
                                // pass, read by the commit pass to find the level without re-evaluating anything
    int32_t *log;
    int64_t chunk;  // nodes per k_level_score block (multiple of kTile)
    int64_t cchunk; // nodes per k_level_commit block
};

// single-node load (commit pass workers: sparse)
template <int NX>
__device__ __forceinline__ void load_one(const DevCols &c, const DevPod &p, int64_t i, NodeRegs<NX> &n) {
    n.w = c.stat[i];
    n.a_cpu = c.alloc[0][i], n.a_mem = c.alloc[1][i];
    n.r_cpu = c.req[0][i], n.r_mem = c.req[1][i];
    n.z_cpu = c.nz_mcpu[i], n.z_mem = c.nz_mem[i];
    n.a_pods = c.alloc_pods[i], n.npods = c.pod_count[i];
#pragma unroll
    for (int x = 0; x < (NX > 0 ? NX : 1); x++) {
        const bool on = NX > 0 && x < p.nx;
        n.xa[x] = on ? c.alloc[p.xcol[x]][i] : 0;
        n.xr[x] = on ? c.req[p.xcol[x]][i] : 0;
    }
}

template <int NX> __device__ __forceinline__ void nd_load(const DevCols &c, const DevPod &p, int64_t i, NodeRegs<NX> &n) { load_one<NX>(c, p, i, n); }
template <int NX> __device__ __forceinline__ void nd_store(const DevCols &c, const DevPod &p, int64_t i, const NodeRegs<NX> &n, int32_t took, int32_t = 0) {
    store_dyn<NX>(c, p, i, n, took);
}
template <int NX> __device__ __forceinline__ void nd_zero(NodeRegs<NX> &n) {
    n.a_cpu = n.a_mem = n.r_cpu = n.r_mem = n.z_cpu = n.z_mem = 0, n.a_pods = n.npods = 0, n.w = 0;
#pragma unroll
    for (int x = 0; x < (NX > 0 ? NX : 1); x++) n.xa[x] = n.xr[x] = 0;
}
__device__ __forceinline__ void nd_load(const DevCols &c, const DevPod &, int64_t i, NodeNarrow &n) { // one commit row: 48 bytes
    const int4 *row = reinterpret_cast<const int4 *>(c.rows + i * kRowWords);
    const int4 s = row[0], d = row[1], q = row[2];
    n.a0 = s.x, n.a1 = s.y, n.a_pods = s.z, n.w = (uint32_t)s.w;
    n.r0 = d.x, n.r1 = d.y, n.z0 = d.z, n.z1 = d.w;
    n.npods = q.x, n.placed = q.y;
}
__device__ __forceinline__ void nd_store(const DevCols &c, const DevPod &, int64_t i, const NodeNarrow &n, int32_t took, int32_t pass = 0) {
    int4 *row = reinterpret_cast<int4 *>(c.rows + i * kRowWords); // the columns follow at the next k_rows_flush
    row[1] = make_int4(n.r0, n.r1, n.z0, n.z1);
    row[2] = make_int4(n.npods, n.placed + took, took, pass); // (clones of THIS pass + its stamp: what a roll-back of a blind batch undoes)
}
__device__ __forceinline__ void nd_zero(NodeNarrow &n) {
    n.a0 = n.a1 = n.r0 = n.r1 = n.z0 = n.z1 = 0, n.a_pods = n.npods = 0, n.w = 0, n.placed = 0;
}

template <int NX, bool NARROW> struct CommitNode { using type = NodeRegs<NX>; };
template <int NX> struct CommitNode<NX, true> { using type = NodeNarrow; };

// running reduction state of one thread over the nodes it scored
struct LevelAcc {
    uint64_t best = 0;
    int64_t top = -1; // maximum score seen by this thread and how many of its nodes hold it
    uint32_t ntop = 0, mt = 0, ma = 0, cmt = 0, cma = 0, nfeas = 0;
    __device__ __forceinline__ void add(int64_t score, int64_t gidx, uint32_t cnt, uint32_t aff) {
        const uint64_t key = make_key(score, gidx);
        best = key > best ? key : best;
        nfeas++;
        if (score > top) top = score, ntop = 1; else if (score == top) ntop++;
        if (cnt > mt) mt = cnt, cmt = 1; else if (cnt == mt) cmt++;
        if (aff > ma) ma = aff, cma = 1; else if (aff == ma) cma++;
    }
};

// the commit pass's running reduction: maximum score, lowest index holding it, how many nodes hold it
struct LevelTop {
    uint64_t best = 0;
    int64_t top = -1;
    uint32_t ntop = 0;
    __device__ __forceinline__ void add(int64_t score, int64_t gidx) {
        const uint64_t key = make_key(score, gidx);
        best = key > best ? key : best;
        if (score > top) top = score, ntop = 1; else if (score == top) ntop++;
    }
};

// ------------------------------------------------------------------------------------------------
// k_level_score: the full pods x nodes pass of the batched mode.  Filter + Score of every node (same arithmetic and
// bytes as k_scan), the packed max key, the size of the top level, the normalization maxima with their holder
// counts, the feasible count -- and one int32 per node: its TotalScore (the commit pass's index of the level).
// No barriers, no LDS list: it runs at k_scan's occupancy.  HBM roofline: 60 B read (+ 4 B written) per node.
// ------------------------------------------------------------------------------------------------
template <int NX, bool NARROW = false>
__global__ __launch_bounds__(kThreads) void k_level_score(LevelArgs a) {
    const DevState st = *a.st;
    // Runs only while the score cache is invalid (first pass of a run; the normalization constants moved).  Otherwise
    // the commit pass keeps the cache current -- it re-scores exactly the nodes it changed -- and derives the next
    // level from it: a placement changes one node, so re-reading the other 10^6 - |level| rows would be wasted traffic.
    if (st.done || !st.lvl_full) return;
    constexpr int kWaves = kThreads / 64;
    __shared__ uint64_t s_key[kWaves];
    __shared__ uint32_t s_u[6][kWaves];
    const uint32_t mt = (uint32_t)st.mt_a, ma = (uint32_t)st.ma_a;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63; // (wave: uniform, and known to the compiler as such)
    const int64_t lo = (int64_t)blockIdx.x * a.chunk;
    int64_t hi = lo + a.chunk;
    if (hi > a.c.n_pad) hi = a.c.n_pad;

    LevelAcc acc;
    const NarrowPod npod = narrow_pod(a.p, a.c.mem_shift);
    for (int64_t base = lo; base < hi; base += kTile) {
        const int64_t i0 = base + 2 * tid;
        if (NARROW) { // 36 B per node, 32-bit arithmetic (ccsim_kernels.h "NARROW arithmetic")
            const uint2 sw = *reinterpret_cast<const uint2 *>(a.c.stat + i0);
            const int2 a0 = *reinterpret_cast<const int2 *>(a.c.a32[0] + i0), a1 = *reinterpret_cast<const int2 *>(a.c.a32[1] + i0);
            const int2 r0 = *reinterpret_cast<const int2 *>(a.c.r32[0] + i0), r1 = *reinterpret_cast<const int2 *>(a.c.r32[1] + i0);
            const int2 z0 = *reinterpret_cast<const int2 *>(a.c.z32[0] + i0), z1 = *reinterpret_cast<const int2 *>(a.c.z32[1] + i0);
            const int2 AP = *reinterpret_cast<const int2 *>(a.c.alloc_pods + i0), NP = *reinterpret_cast<const int2 *>(a.c.pod_count + i0);
            int2 cs;
#pragma unroll
            for (int k = 0; k < 2; k++) {
                const uint32_t w = k ? sw.y : sw.x;
                const int32_t na0 = k ? a0.y : a0.x, na1 = k ? a1.y : a1.x, nr0 = k ? r0.y : r0.x, nr1 = k ? r1.y : r1.x;
                int32_t sc = -1;
                if ((w >> kStatOkBit) && fits_narrow(a.p, npod, na0, na1, nr0, nr1, k ? AP.y : AP.x, k ? NP.y : NP.x)) {
                    const uint32_t cnt = (w >> kStatCntShift) & kStatCntMask, aff = w & kStatAffMask, img = (w >> kStatImgShift) & kStatImgMask;
                    const int64_t s64 = static_score(a.p, cnt, aff, img, mt, ma) +
                                        dynamic_score_narrow(a.p, npod, na0, na1, nr0, nr1, k ? z0.y : z0.x, k ? z1.y : z1.x);
                    acc.add(s64, a.c.global_offset + i0 + k, cnt, aff);
                    sc = (int32_t)s64;
                }
                (k ? cs.y : cs.x) = sc;
            }
            *reinterpret_cast<int2 *>(a.cscore + i0) = cs;
            continue;
        }
        NodeRegs<NX> nd[2];
        load_pair<NX>(a.c, a.p, i0, nd);
        int2 cs;
#pragma unroll
        for (int k = 0; k < 2; k++) {
            int32_t sc = -1;
            if (node_feasible<NX>(a.p, nd[k])) {
                const uint32_t cnt = (nd[k].w >> kStatCntShift) & kStatCntMask, aff = nd[k].w & kStatAffMask, img = (nd[k].w >> kStatImgShift) & kStatImgMask;
                const int64_t s64 = node_score<NX>(a.p, nd[k], static_score(a.p, cnt, aff, img, mt, ma));
                acc.add(s64, a.c.global_offset + i0 + k, cnt, aff);
                sc = (int32_t)s64;
            }
            (k ? cs.y : cs.x) = sc;
        }
        *reinterpret_cast<int2 *>(a.cscore + i0) = cs;
    }
    {
        const uint64_t wbest = wave_max_u64(acc.best);
        const int64_t wtop = wave_max_i64(acc.top);
        const uint32_t wmt = wave_max_u32(acc.mt), wma = wave_max_u32(acc.ma);
        const uint32_t wntop = wave_sum_u32(acc.top == wtop ? acc.ntop : 0u);
        const uint32_t wcmt = wave_sum_u32(acc.mt == wmt ? acc.cmt : 0u), wcma = wave_sum_u32(acc.ma == wma ? acc.cma : 0u);
        const uint32_t wnf = wave_sum_u32(acc.nfeas);
        if (lane == 0) {
            s_key[wave] = wbest;
            s_u[0][wave] = wmt, s_u[1][wave] = wma, s_u[2][wave] = wcmt, s_u[3][wave] = wcma, s_u[4][wave] = wnf, s_u[5][wave] = wntop;
        }
    }
    __syncthreads();
    if (tid == 0) {
        LevelPartial out{};
#pragma unroll
        for (int w = 0; w < kWaves; w++) {
            const uint64_t kw = s_key[w];
            if (kw && (!out.key || key_score(kw) > key_score(out.key))) out.n_top = s_u[5][w];
            else if (kw && key_score(kw) == key_score(out.key)) out.n_top += s_u[5][w];
            out.key = kw > out.key ? kw : out.key;
            if (s_u[0][w] > out.mt) out.mt = s_u[0][w], out.c_mt = s_u[2][w]; else if (s_u[0][w] == out.mt) out.c_mt += s_u[2][w];
            if (s_u[1][w] > out.ma) out.ma = s_u[1][w], out.c_ma = s_u[3][w]; else if (s_u[1][w] == out.ma) out.c_ma += s_u[3][w];
            out.nfeas += s_u[4][w];
        }
        a.partials[blockIdx.x] = out;
    }
}

// ------------------------------------------------------------------------------------------------
// k_level_commit: commits (or, in a plan pass, only measures) the level st.lvl_M.  Reads 4 B per node (the cached
// TotalScore) to find the level; level nodes are sparse (a few % of a tile) and a run-down costs hundreds of VALU
// instructions per step, so they are COMPACTED: owners append their level nodes' indices to a block-wide LDS work
// list in canonical order (ballot/popcount prefix), and the first lanes of the block take one entry each, gather
// the node's columns, run it down (wave_run_down) and rewrite its columns -- a tile's run-downs execute in one or
// two densely packed waves.  Ordered commits (limit / log) add an exclusive scan of the run-down lengths.
// ------------------------------------------------------------------------------------------------
constexpr int kGroupTiles = 4; // tiles compacted together: 2048 nodes, 8 KiB of LDS work list

template <int NX, bool NARROW = false>
__global__ __launch_bounds__(kThreads) void k_level_commit(LevelArgs a) {
    using Node = typename CommitNode<NX, NARROW>::type;
    const RunCtx cx{a.p, narrow_pod(a.p, a.c.mem_shift)};
    const DevState st = *a.st;
    if (st.done) return;
    if (NARROW && st.lvl_rollback) { // undo the blind batch of pass lvl_pass: its clones off the rows, the cached scores back
        const uint32_t mt = (uint32_t)st.mt_a, ma = (uint32_t)st.ma_a;
        const int64_t lo = (int64_t)blockIdx.x * a.cchunk;
        int64_t hi = lo + a.cchunk;
        if (hi > a.c.n_pad) hi = a.c.n_pad;
        for (int64_t i = lo + threadIdx.x; i < hi; i += kThreads) {
            const int4 q = reinterpret_cast<const int4 *>(a.c.rows + i * kRowWords)[2];
            if (q.w != st.lvl_pass || q.z <= 0) continue;
            typename CommitNode<NX, NARROW>::type n;
            nd_load(a.c, a.p, i, n);
            nd_apply(cx, n, -(int64_t)q.z);
            nd_store(a.c, a.p, i, n, -q.z, 0);
            const uint32_t nw = nd_word(n);
            const uint32_t cnt = (nw >> kStatCntShift) & kStatCntMask, aff = nw & kStatAffMask, img = (nw >> kStatImgShift) & kStatImgMask;
            a.cscore[i] = (int32_t)nd_score(cx, n, static_score(a.p, cnt, aff, img, mt, ma), nd_rcp(n)); // (it took clones: it was feasible)
        }
        return;
    }
    const bool plan_only = st.lvl_plan_only != 0;
    const bool commit_on = st.lvl_valid != 0 && !plan_only;
    if (!plan_only && !commit_on) return; // nothing planned (first pass, or the constants just changed)
    constexpr int kWaves = kThreads / 64;
    __shared__ int32_t s_idx[kGroupTiles * kTile]; // work list: node index relative to the block's chunk, canonical order
    __shared__ uint32_t s_u[2][kWaves];
    __shared__ int64_t s_l[4][kWaves];
    __shared__ int32_t s_cnt[kGroupTiles][kWaves];

    const uint32_t mt = (uint32_t)st.mt_a, ma = (uint32_t)st.ma_a;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63; // (wave: uniform, and known to the compiler as such)
    const int64_t lo = (int64_t)blockIdx.x * a.cchunk;
    int64_t hi = lo + a.cchunk;
    if (hi > a.c.n_pad) hi = a.c.n_pad;
    const bool ordered = commit_on && st.lvl_prefix != 0;
    // a blind batch takes the levels lvl_M .. lvl_Lo in one pass: every node scoring >= Lo runs down until it scores < Lo -- for a
    // single node exactly its run-downs at the levels in between, and without a log or a limit the interleaving across nodes is
    // unobservable (ccsim_persist.h has the argument; level_decide validates the batch afterwards).  One level: Lo == M.
    const int32_t M = (int32_t)(commit_on && st.lvl_blind ? st.lvl_Lo : st.lvl_M);
    const int32_t pass_stamp = st.lvl_pass + 1;
    const uint64_t lt_mask = lane ? (~0ull >> (64 - lane)) : 0ull;
    const bool blind_batch = commit_on && st.lvl_blind != 0 && !plan_only;

    int64_t committed = 0;
    int64_t carry = ordered ? st.lvl_rank_prefix + a.blockprefix[blockIdx.x] : 0;
    int64_t T = 0, cut_mt = -1, cut_ma = -1; // plan pass
    uint32_t e_mt = 0, e_ma = 0;
    LevelTop acc;                            // commit pass: the next level
    uint32_t x_nf = 0, x_mt = 0, x_ma = 0;

    // Tiles are handled in groups of kGroupTiles: ONE compaction and ONE worker phase per group (the phases of a
    // block are latency chains -- gather loads, dependent run-down steps -- so fewer, fuller phases win).
    for (int64_t gbase = lo; gbase < hi; gbase += (int64_t)kGroupTiles * kTile) {
        uint32_t flags = 0; // bit 2t / 2t+1: this thread's first / second node of tile t holds the level
        int wcnt[kGroupTiles];
#pragma unroll
        for (int t = 0; t < kGroupTiles; t++) {
            const int64_t base = gbase + (int64_t)t * kTile;
            wcnt[t] = 0;
            if (base < hi) { // block-uniform
                const int64_t i0 = base + 2 * tid;
                const int2 cs = *reinterpret_cast<const int2 *>(a.cscore + i0);
                const bool l0 = cs.x >= M && (a.c.global_offset + i0) <= st.lvl_cut; // (M is the maximum unless this is a batch: >= is == then)
                const bool l1 = cs.y >= M && (a.c.global_offset + i0 + 1) <= st.lvl_cut;
                const uint64_t b0 = __ballot(l0), b1 = __ballot(l1);
                flags |= (l0 ? 1u : 0u) << (2 * t) | (l1 ? 1u : 0u) << (2 * t + 1);
                if (!plan_only) { // nodes this pass does not touch keep their cached score
                    if (!l0 && cs.x >= 0) acc.add(cs.x, a.c.global_offset + i0);
                    if (!l1 && cs.y >= 0) acc.add(cs.y, a.c.global_offset + i0 + 1);
                }
                wcnt[t] = __popcll(b0 & lt_mask) + __popcll(b1 & lt_mask); // level nodes of lower lanes in this wave
                if (lane == 0) s_cnt[t][wave] = __popcll(b0) + __popcll(b1);
            } else if (lane == 0)
                s_cnt[t][wave] = 0;
        }
        __syncthreads();
        // canonical order: tile-major, then thread, then the thread's two nodes
        int total = 0;
#pragma unroll
        for (int t = 0; t < kGroupTiles; t++) {
            int off = total + wcnt[t];
#pragma unroll
            for (int w = 0; w < kWaves; w++) {
                if (w < wave) off += s_cnt[t][w];
                total += s_cnt[t][w];
            }
            const int64_t i0 = gbase + (int64_t)t * kTile + 2 * tid;
            if (flags & (1u << (2 * t))) s_idx[off] = (int32_t)(i0 - lo), off++;
            if (flags & (1u << (2 * t + 1))) s_idx[off] = (int32_t)(i0 + 1 - lo);
        }
        __syncthreads();
        for (int r0 = 0; r0 < total; r0 += kThreads) { // one entry per worker lane per round
            const int nwork = total - r0 < kThreads ? total - r0 : kThreads;
            const bool mine = tid < nwork;
            Node n;
            nd_zero(n);
            int64_t nidx = 0;
            if (mine) {
                nidx = lo + s_idx[r0 + tid];
                nd_load(a.c, a.p, nidx, n);
            }
            const uint32_t nw = nd_word(n);
            const uint32_t cnt = (nw >> kStatCntShift) & kStatCntMask, aff = nw & kStatAffMask, img = (nw >> kStatImgShift) & kStatImgMask;
            const int64_t nstat = static_score(a.p, cnt, aff, img, mt, ma);
            bool fend = true;
            int64_t j = 0;
            if ((wave * 64) < nwork) j = wave_run_down<Node>(cx, n, nstat, M, mine, fend); // wave-uniform
            const int64_t g = a.c.global_offset + nidx;
            if (plan_only) {
                if (mine) {
                    T += j;
                    if (!fend) {
                        if (mt > 0 && cnt == mt) e_mt++, cut_mt = g > cut_mt ? g : cut_mt;
                        if (ma > 0 && aff == ma) e_ma++, cut_ma = g > cut_ma ? g : cut_ma;
                    }
                }
            } else {
                int64_t took = j, pos = 0;
                if (ordered) { // position of this node's first placement inside the level
                    const int64_t incl = wave_incl_scan_i64(j);
                    __syncthreads(); // s_l reuse across rounds
                    if (lane == 63) s_l[0][wave] = incl;
                    __syncthreads();
                    int64_t before = 0, tot = 0;
#pragma unroll
                    for (int w = 0; w < kWaves; w++) {
                        if (w < wave) before += s_l[0][w];
                        tot += s_l[0][w];
                    }
                    pos = carry + before + incl - j;
                    carry += tot;
                    int64_t allowed = st.lvl_remaining - pos;
                    if (allowed < 0) allowed = 0;
                    if (took > allowed) took = allowed;
                }
                if (mine && took > 0) {
                    nd_apply(cx, n, took);
                    nd_store(a.c, a.p, nidx, n, (int32_t)took, pass_stamp);
                    committed += took;
                    if (ordered && a.log) {
                        for (int64_t q = 0; q < took; q++) {
                            const int64_t at = st.placed + pos + q;
                            if (at < st.log_cap) a.log[at] = (int32_t)g;
                        }
                    }
                }
                if (mine) { // re-score the node in the state it was left in: the cache stays exact
                    const bool f = nd_feasible(cx, n);
                    const int64_t s = f ? nd_score(cx, n, nstat, nd_rcp(n)) : -1;
                    a.cscore[nidx] = (int32_t)s;
                    if (f) acc.add(s, g);
                    else {
                        x_nf++;
                        x_mt += cnt == mt ? 1u : 0u;
                        x_ma += aff == ma ? 1u : 0u;
                        // a holder of a normalization maximum filled up in a blind batch: its score before the last clone -- should the
                        // batch fail validation, the lowest of them is where the last holder went (level_decide, DevState::lvl_ev)
                        if (blind_batch && took > 0 && ((mt > 0 && cnt == mt) || (ma > 0 && aff == ma))) {
                            auto q = n;
                            nd_apply(cx, q, -1);
                            const int64_t sp = nd_score(cx, q, nstat, nd_rcp(q));
                            if (mt > 0 && cnt == mt) cut_mt = kEvBase - sp > cut_mt ? kEvBase - sp : cut_mt; // (max of the complement: the lowest score)
                            if (ma > 0 && aff == ma) cut_ma = kEvBase - sp > cut_ma ? kEvBase - sp : cut_ma;
                        }
                    }
                }
            }
        }
        __syncthreads(); // the list and the counters are rewritten by the next group
    }

    committed = wave_sum_i64(committed);
    if (blind_batch) cut_mt = wave_max_i64(cut_mt), cut_ma = wave_max_i64(cut_ma); // (block-uniform; the plan pass's fields carry the located levels)
    if (plan_only) { // block-uniform
        T = wave_sum_i64(T);
        cut_mt = wave_max_i64(cut_mt);
        cut_ma = wave_max_i64(cut_ma);
        e_mt = wave_sum_u32(e_mt);
        e_ma = wave_sum_u32(e_ma);
    }
    __shared__ uint64_t s_k[kWaves];
    __shared__ uint32_t s_x[4][kWaves];
    if (!plan_only) {
        const uint64_t wbest = wave_max_u64(acc.best);
        const int64_t wtop = wave_max_i64(acc.top);
        const uint32_t wntop = wave_sum_u32(acc.top == wtop ? acc.ntop : 0u);
        x_nf = wave_sum_u32(x_nf), x_mt = wave_sum_u32(x_mt), x_ma = wave_sum_u32(x_ma);
        if (lane == 0) s_k[wave] = wbest, s_x[0][wave] = wntop, s_x[1][wave] = x_nf, s_x[2][wave] = x_mt, s_x[3][wave] = x_ma;
    } else if (lane == 0)
        s_k[wave] = 0, s_x[0][wave] = s_x[1][wave] = s_x[2][wave] = s_x[3][wave] = 0;
    if (lane == 0) {
        s_u[0][wave] = e_mt, s_u[1][wave] = e_ma;
        s_l[0][wave] = T, s_l[1][wave] = committed, s_l[2][wave] = cut_mt, s_l[3][wave] = cut_ma;
    }
    __syncthreads();
    if (tid == 0) {
        CommitPartial out{};
        out.cut_mt = out.cut_ma = -1;
#pragma unroll
        for (int w = 0; w < kWaves; w++) {
            const uint64_t kw = s_k[w];
            if (kw && (!out.key || key_score(kw) > key_score(out.key))) out.n_top = s_x[0][w];
            else if (kw && key_score(kw) == key_score(out.key)) out.n_top += s_x[0][w];
            out.key = kw > out.key ? kw : out.key;
            out.x_nf += s_x[1][w], out.x_mt += s_x[2][w], out.x_ma += s_x[3][w];
            out.e_mt += s_u[0][w], out.e_ma += s_u[1][w];
            out.T += s_l[0][w];
            out.committed += s_l[1][w];
            out.cut_mt = s_l[2][w] > out.cut_mt ? s_l[2][w] : out.cut_mt;
            out.cut_ma = s_l[3][w] > out.cut_ma ? s_l[3][w] : out.cut_ma;
        }
        a.cpartials[blockIdx.x] = out;
    }
}

// ------------------------------------------------------------------------------------------------
// Aggregate of one pass over a shard (or, after the exchange, over the whole cluster).
// ------------------------------------------------------------------------------------------------
struct LevelAgg {
    uint64_t key;
    uint32_t mt, c_mt, ma, c_ma;
    int64_t nfeas, committed, n_top;
    int64_t T, e_mt, e_ma, cut_mt, cut_ma; // plan pass
};

// The sequential part of a level: simulator.go:297-312 limit test, schedule_one.go:448-454 FitError,
// normalization-constant tracking, and how the next level is to be committed.
__device__ __forceinline__ void level_issue(DevState &st, int64_t n_top, uint32_t mt, uint32_t c_mt, uint32_t ma, uint32_t c_ma, bool want_log, bool must_plan) {
    // how the level st.lvl_M is to be committed
    st.lvl_cut = kNoCut;
    st.lvl_remaining = kNoCut;
    st.lvl_prefix = 0;
    st.lvl_rank_prefix = 0;
    st.lvl_plan_only = 0, st.lvl_valid = 0, st.lvl_blind = 0;
    st.lvl_Lo = st.lvl_M;
    // a rolled-back batch located its event at level lvl_ev: the levels above it in one batch, that level itself in canonical order
    if (st.lvl_ev >= 0 && st.lvl_M <= st.lvl_ev) st.lvl_ev = -1, must_plan = true;
    else if (st.lvl_kb > 1 && !want_log) { // several levels, blind; validated by the next level_decide (roll-back + fewer levels if it fails)
        int64_t lo = st.lvl_M - (st.lvl_kb - 1);
        if (st.lvl_ev >= 0 && lo <= st.lvl_ev) lo = st.lvl_ev + 1;
        st.lvl_Lo = lo > 0 ? lo : 0;
        st.lvl_blind = 1, st.lvl_valid = 1;
        st.prev_nfeas = st.cur_nfeas, st.prev_c_mt = st.cur_c_mt, st.prev_c_ma = st.cur_c_ma;
        return;
    }
    // Could anything end this level early?  The limit / the log need positions; a normalization maximum
    // can only move if ALL its feasible holders are exhausted, i.e. the level has at least that many nodes.
    const bool plan = must_plan || want_log || st.limit > 0 || (mt > 0 && n_top >= (int64_t)c_mt) || (ma > 0 && n_top >= (int64_t)c_ma);
    if (plan) st.lvl_plan_only = 1;
    else st.lvl_valid = 1;
}

__device__ __forceinline__ void level_decide(DevState &st, const LevelAgg &g, bool want_log) {
    st.scans += 1;
    st.winner = -1;
    if (st.lvl_rollback) { // the pass undid the batch: the same level again, fewer levels at once (one level: measured first)
        st.lvl_rollback = 0;
        level_issue(st, 0, (uint32_t)st.mt_a, (uint32_t)st.lvl_c_mt, (uint32_t)st.ma_a, (uint32_t)st.lvl_c_ma, want_log, st.lvl_kb <= 1);
        return;
    }
    if (st.lvl_plan_only) { // the pass measured level lvl_M (nothing moved): now commit it, carefully
        st.lvl_plan_only = 0;
        st.lvl_valid = 1;
        int64_t cut = kNoCut;
        if (st.mt_a > 0 && g.e_mt == st.lvl_c_mt && g.cut_mt < cut) cut = g.cut_mt; // every feasible holder exhausted
        if (st.ma_a > 0 && g.e_ma == st.lvl_c_ma && g.cut_ma < cut) cut = g.cut_ma;
        st.lvl_cut = cut;
        st.lvl_remaining = st.limit > 0 ? st.limit - st.placed : kNoCut;
        st.lvl_prefix = (want_log || (st.limit > 0 && st.placed + g.T > st.limit)) ? 1 : 0;
        return;
    }
    const bool incremental = !st.lvl_full; // g came from the score cache (commit pass), not from a full pass
    if (incremental) st.lvl_pass += 1;     // (a commit pass ran: its rows carry this stamp)
    if (incremental && st.lvl_blind) {
        // validate the blind batch: did it exhaust every feasible holder of a normalization maximum (the nodes after that holder
        // should have been re-scored first), or cross --max-limit (the prefix that fits is a matter of order)?
        const bool cut_event = (st.mt_a > 0 && g.c_mt == 0) || (st.ma_a > 0 && g.c_ma == 0);
        const bool over = st.limit > 0 && st.placed + g.committed > st.limit;
        if (cut_event || over) {
            st.lvl_rollback = 1, st.lvl_valid = 0, st.lvl_blind = 0;
            const int64_t span = st.lvl_M - st.lvl_Lo + 1;
            st.lvl_kb = span > 1 ? (int32_t)(span >> 1) : 1; // retry with half the levels ...
            if (cut_event && !over) { // ... unless the holders that filled up say where the event was (ccsim_persist.h ev_level)
                int64_t ev = -1;
                if (st.mt_a > 0 && g.c_mt == 0 && g.cut_mt > 0) ev = kEvBase - g.cut_mt;
                if (st.ma_a > 0 && g.c_ma == 0 && g.cut_ma > 0 && kEvBase - g.cut_ma > ev) ev = kEvBase - g.cut_ma; // (the first event: the higher level)
                if (ev >= st.lvl_Lo && ev <= st.lvl_M) st.lvl_ev = ev, st.lvl_kb = st.lvl_kb_max > 1 ? st.lvl_kb_max : 2;
            }
            st.cur_nfeas = st.prev_nfeas, st.cur_c_mt = st.prev_c_mt, st.cur_c_ma = st.prev_c_ma;
            return;
        }
        st.lvl_kb = 2 * st.lvl_kb < st.lvl_kb_max ? 2 * st.lvl_kb : st.lvl_kb_max;
    } else if (incremental && st.lvl_kb < st.lvl_kb_max)
        st.lvl_kb = 2 * st.lvl_kb < st.lvl_kb_max ? (st.lvl_kb < 1 ? 1 : 2 * st.lvl_kb) : st.lvl_kb_max; // (an ordered level went through: batches again)
    st.placed += g.committed;
    st.rounds += g.committed;
    st.lvl_valid = 0, st.lvl_blind = 0;
    if (st.limit > 0 && st.placed >= st.limit) {
        st.done = DONE_LIMIT;
        return;
    }
    if (g.key == 0) {
        st.done = DONE_UNSCHEDULABLE;
        st.rounds += 1;
        st.last_feasible = 0;
        return;
    }
    st.last_feasible = (int32_t)g.nfeas;
    if (incremental && ((st.mt_a > 0 && g.c_mt == 0) || (st.ma_a > 0 && g.c_ma == 0))) {
        st.lvl_full = 1; // the last feasible holder of a normalization maximum is gone: every cached score is stale
        st.lvl_ev = -1;  // (a level of the old score scale)
        return;
    }
    if ((int32_t)g.mt != st.mt_a || (int32_t)g.ma != st.ma_a) { // scores above used stale constants: rescan
        st.mt_a = (int32_t)g.mt;
        st.ma_a = (int32_t)g.ma;
        st.lvl_full = 1;
        st.lvl_ev = -1;
        return;
    }
    st.lvl_full = 0;
    st.lvl_M = key_score(g.key);
    st.lvl_c_mt = g.c_mt; // holder counts of the maxima, for the plan pass's "all holders exhausted?" test
    st.lvl_c_ma = g.c_ma;
    level_issue(st, g.n_top, g.mt, g.c_mt, g.ma, g.c_ma, want_log, false);
}

struct LevelFinalArgs {
    DevState *st;
    const LevelPartial *partials; // score pass
    int32_t n_partials;
    const CommitPartial *cpartials; // commit pass
    int32_t n_cpartials;
    int64_t *blockprefix;
    XRec *xsend;       // distributed: this shard's record out
    const XRec *xrecv; // distributed: gathered records in
    int32_t n_ranks;   // 0 = single GPU
    int32_t rank;
    int32_t want_log;
    // which kernels ran before this one in the pass.  The graph of a single-GPU run is 2 x [k_level_score, k_level_final]
    // followed by R x [k_level_commit, k_level_final]: a pass whose kernels cannot do what the state asks for (a full
    // pass is due inside the commit-only stretch, or nothing is due in a score-only pass) is a no-op.
    int32_t commit_launched, score_launched;
};

// k_level_final: one block.  Reduces the per-block partials of the pass that just ran (commit partials: what was
// committed or planned; score partials: the next level) in one sweep each; on one GPU also decides.  After a plan
// pass it leaves the exclusive per-block prefix of the level's placements in blockprefix[].
constexpr int kFinalThreads = 256;

__global__ __launch_bounds__(kFinalThreads) void k_level_final(LevelFinalArgs a) {
    if (a.st->done) return;
    if (a.st->lvl_full ? !a.score_launched : !a.commit_launched) return; // block-uniform: see LevelFinalArgs
    if (a.st->lvl_rollback) { // the commit kernel undid a blind batch: nothing to reduce
        if (threadIdx.x == 0) {
            if (a.n_ranks > 0) {
                XRec r{};
                *a.xsend = r; // (every rank decides the same way from its replicated state)
            } else {
                DevState st = *a.st;
                LevelAgg g{};
                level_decide(st, g, a.want_log != 0);
                *a.st = st;
            }
        }
        return;
    }
    constexpr int kWaves = kFinalThreads / 64;
    __shared__ uint64_t s_key[kWaves];
    __shared__ uint32_t s_u[4][kWaves];
    __shared__ int64_t s_l[8][kWaves];
    __shared__ int64_t s_x[3][kWaves];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63; // (wave: uniform, and known to the compiler as such)
    const bool plan_pass = a.st->lvl_plan_only != 0;
    const bool commit_ran = plan_pass || a.st->lvl_valid != 0; // else k_level_commit exited without writing partials
    const bool full_pass = a.st->lvl_full != 0;                // k_level_score ran: the next level comes from its partials

    uint64_t key = 0;
    uint32_t mt = 0, cmt = 0, ma = 0, cma = 0;
    int64_t nf = 0, committed = 0, T = 0, e_mt = 0, e_ma = 0, cut_mt = -1, cut_ma = -1, ntop = 0;
    int64_t x_nf = 0, x_mt = 0, x_ma = 0;
    if (full_pass)
        for (int i = tid; i < a.n_partials; i += kFinalThreads) {
            const LevelPartial q = a.partials[i];
            if (q.key) { // size of the top level: nodes holding the maximum score, over the blocks whose maximum it is
                if (!key || key_score(q.key) > key_score(key)) ntop = q.n_top;
                else if (key_score(q.key) == key_score(key)) ntop += q.n_top;
            }
            key = q.key > key ? q.key : key;
            if (q.mt > mt) mt = q.mt, cmt = q.c_mt; else if (q.mt == mt) cmt += q.c_mt;
            if (q.ma > ma) ma = q.ma, cma = q.c_ma; else if (q.ma == ma) cma += q.c_ma;
            nf += q.nfeas;
        }
    if (commit_ran)
        for (int i = tid; i < a.n_cpartials; i += kFinalThreads) {
            const CommitPartial q = a.cpartials[i];
            committed += q.committed;
            T += q.T, e_mt += q.e_mt, e_ma += q.e_ma;
            cut_mt = q.cut_mt > cut_mt ? q.cut_mt : cut_mt;
            cut_ma = q.cut_ma > cut_ma ? q.cut_ma : cut_ma;
            if (!plan_pass && !full_pass) { // the next level, from the score cache
                if (q.key) {
                    if (!key || key_score(q.key) > key_score(key)) ntop = q.n_top;
                    else if (key_score(q.key) == key_score(key)) ntop += q.n_top;
                }
                key = q.key > key ? q.key : key;
                x_nf += q.x_nf, x_mt += q.x_mt, x_ma += q.x_ma;
            }
        }
    {
        const uint64_t wkey = wave_max_u64(key);
        const uint32_t wmt = wave_max_u32(mt), wma = wave_max_u32(ma);
        const uint32_t wcmt = wave_sum_u32(mt == wmt ? cmt : 0u), wcma = wave_sum_u32(ma == wma ? cma : 0u);
        ntop = wave_sum_i64((key && wkey && key_score(key) == key_score(wkey)) ? ntop : 0);
        nf = wave_sum_i64(nf), committed = wave_sum_i64(committed);
        T = wave_sum_i64(T), e_mt = wave_sum_i64(e_mt), e_ma = wave_sum_i64(e_ma);
        cut_mt = wave_max_i64(cut_mt), cut_ma = wave_max_i64(cut_ma);
        x_nf = wave_sum_i64(x_nf), x_mt = wave_sum_i64(x_mt), x_ma = wave_sum_i64(x_ma);
        if (lane == 0) s_x[0][wave] = x_nf, s_x[1][wave] = x_mt, s_x[2][wave] = x_ma;
        if (lane == 0) {
            s_key[wave] = wkey;
            s_u[0][wave] = wmt, s_u[1][wave] = wma, s_u[2][wave] = wcmt, s_u[3][wave] = wcma;
            s_l[0][wave] = nf, s_l[1][wave] = committed, s_l[2][wave] = T, s_l[3][wave] = e_mt, s_l[4][wave] = e_ma;
            s_l[5][wave] = cut_mt, s_l[6][wave] = cut_ma, s_l[7][wave] = ntop;
        }
    }
    __syncthreads();
    if (tid == 0) {
        LevelAgg g{};
        g.cut_mt = g.cut_ma = -1;
        for (int w = 0; w < kWaves; w++) {
            const uint64_t kw = s_key[w];
            if (kw) {
                if (!g.key || key_score(kw) > key_score(g.key)) g.n_top = s_l[7][w];
                else if (key_score(kw) == key_score(g.key)) g.n_top += s_l[7][w];
            }
            g.key = kw > g.key ? kw : g.key;
            if (s_u[0][w] > g.mt) g.mt = s_u[0][w], g.c_mt = s_u[2][w]; else if (s_u[0][w] == g.mt) g.c_mt += s_u[2][w];
            if (s_u[1][w] > g.ma) g.ma = s_u[1][w], g.c_ma = s_u[3][w]; else if (s_u[1][w] == g.ma) g.c_ma += s_u[3][w];
            g.nfeas += s_l[0][w], g.committed += s_l[1][w], g.T += s_l[2][w], g.e_mt += s_l[3][w], g.e_ma += s_l[4][w];
            g.cut_mt = s_l[5][w] > g.cut_mt ? s_l[5][w] : g.cut_mt;
            g.cut_ma = s_l[6][w] > g.cut_ma ? s_l[6][w] : g.cut_ma;
        }
        if (!plan_pass && !full_pass) { // incremental pass: the maxima stand, their holders and the feasible count shrink
            int64_t dn = 0, dt = 0, da = 0;
            for (int w = 0; w < kWaves; w++) dn += s_x[0][w], dt += s_x[1][w], da += s_x[2][w];
            g.mt = (uint32_t)a.st->mt_a, g.ma = (uint32_t)a.st->ma_a;
            g.nfeas = a.st->cur_nfeas - dn;
            g.c_mt = (uint32_t)(a.st->cur_c_mt - dt), g.c_ma = (uint32_t)(a.st->cur_c_ma - da);
        }
        if (!plan_pass) { // this shard's feasible count and holder counts after the pass
            a.st->cur_nfeas = g.nfeas;
            a.st->cur_c_mt = g.c_mt, a.st->cur_c_ma = g.c_ma;
        }
        if (a.n_ranks > 0) { // publish this shard's record; k_level_decide finishes after the exchange
            XRec r{};
            r.key = (int64_t)g.key, r.mt = g.mt, r.ma = g.ma, r.nfeas = g.nfeas;
            r.c_mt = g.c_mt, r.c_ma = g.c_ma, r.committed = g.committed, r.n_top = g.n_top;
            r.T = g.T, r.e_mt = g.e_mt, r.e_ma = g.e_ma, r.cut_mt = g.cut_mt, r.cut_ma = g.cut_ma;
            *a.xsend = r;
        } else {
            DevState st = *a.st;
            level_decide(st, g, a.want_log != 0);
            *a.st = st;
        }
    }
    if (!plan_pass) return; // block-uniform
    // exclusive prefix over commit blocks (canonical node order == block order) of the level's placements
    __syncthreads();
    const int per = (a.n_cpartials + kFinalThreads - 1) / kFinalThreads;
    const int b0 = tid * per < a.n_cpartials ? tid * per : a.n_cpartials;
    const int b1 = (b0 + per < a.n_cpartials) ? b0 + per : a.n_cpartials;
    int64_t mine = 0;
    for (int i = b0; i < b1; i++) mine += a.cpartials[i].T;
    const int64_t incl = wave_incl_scan_i64(mine);
    if (lane == 63) s_l[0][wave] = incl;
    __syncthreads();
    int64_t run = incl - mine;
    for (int w = 0; w < wave; w++) run += s_l[0][w];
    for (int i = b0; i < b1; i++) {
        a.blockprefix[i] = run;
        run += a.cpartials[i].T;
    }
}

// k_level_decide (distributed): every rank reduces the gathered records identically; shards are
// contiguous ranges of the canonical order, so a rank's placements in a level come after those of all
// lower ranks.
__global__ void k_level_decide(LevelFinalArgs a) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    DevState st = *a.st;
    if (st.done) return;
    if (st.lvl_full ? !a.score_launched : !a.commit_launched) return; // the pass was a no-op on every rank (LevelFinalArgs)
    if (st.lvl_rollback) {
        LevelAgg g0{};
        level_decide(st, g0, a.want_log != 0);
        *a.st = st;
        return;
    }
    LevelAgg g{};
    g.cut_mt = g.cut_ma = -1;
    int64_t before = 0;
    for (int r = 0; r < a.n_ranks; r++) {
        const XRec q = a.xrecv[r];
        g.key = (uint64_t)q.key > g.key ? (uint64_t)q.key : g.key;
        if ((uint32_t)q.mt > g.mt) g.mt = (uint32_t)q.mt, g.c_mt = (uint32_t)q.c_mt; else if ((uint32_t)q.mt == g.mt) g.c_mt += (uint32_t)q.c_mt;
        if ((uint32_t)q.ma > g.ma) g.ma = (uint32_t)q.ma, g.c_ma = (uint32_t)q.c_ma; else if ((uint32_t)q.ma == g.ma) g.c_ma += (uint32_t)q.c_ma;
        g.nfeas += q.nfeas;
        g.committed += q.committed;
        if (r < a.rank) before += q.T;
        g.T += q.T, g.e_mt += q.e_mt, g.e_ma += q.e_ma;
        g.cut_mt = q.cut_mt > g.cut_mt ? q.cut_mt : g.cut_mt;
        g.cut_ma = q.cut_ma > g.cut_ma ? q.cut_ma : g.cut_ma;
    }
    const int64_t top = g.key ? key_score(g.key) : -1;
    for (int r = 0; r < a.n_ranks; r++) {
        const XRec q = a.xrecv[r];
        if (q.key && key_score((uint64_t)q.key) == top) g.n_top += q.n_top;
    }
    const bool plan_pass = st.lvl_plan_only != 0;
    if (!plan_pass && st.lvl_full) { // after a full pass: this shard's holder counts refer to the GLOBAL maxima
        const XRec own = a.xrecv[a.rank];
        if ((uint32_t)own.mt != g.mt) st.cur_c_mt = 0;
        if ((uint32_t)own.ma != g.ma) st.cur_c_ma = 0;
    }
    level_decide(st, g, a.want_log != 0);
    if (plan_pass) st.lvl_rank_prefix = before;
    *a.st = st;
}

} // namespace ccsim
