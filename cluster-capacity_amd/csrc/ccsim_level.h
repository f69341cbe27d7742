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
// One k_level pass reads every node column once and
//   (a) COMMITS the current level: every node holding M runs down independently and rewrites its own
//       columns (no atomics, no cross-node traffic).  Run-downs are evaluated per lane for the first few
//       placements and WAVE-COOPERATIVELY for long ones (wave_run_down below): <= kSeqSteps + 2 dependent
//       steps for a 110-pod node instead of 110;
//   (b) evaluates Filter + Score of every node in its post-commit state and reduces the next level
//       (packed max key), the normalization maxima with their holder counts, and the feasible count.
// k_level_final (one block) reduces the per-block partials and decides.  A level is committed blindly
// unless something could end it early; then one extra PLAN pass (same kernel, no commit) measures it:
//   * the level could exhaust the last feasible holder of a normalization maximum (level size >= holder
//     count): nodes after that holder ("cut") must be re-scored with new constants -> commit up to the cut;
//   * --max-limit, or the caller wants the placement log: the commit becomes ORDERED (exclusive scan of
//     the run-down lengths: block prefixes from k_level_final + in-block scan) so that a prefix of the
//     level can be committed and every placement knows its position in the sequence.
// The placement sequence, per-node counts and terminal state are IDENTICAL to CCSIM_MODE_SEQUENTIAL
// (tests/test_level_model.py proves the argument on the CPU against the oracle; tests/test_gpu_parity.py
// checks this kernel).
//
// Roofline: HBM.  A pass reads every enabled column once (same algorithmic bytes as k_scan) and rewrites
// only the columns of the nodes that took pods.
#pragma once
#include "ccsim_kernels.h"

namespace ccsim {

constexpr int64_t kNoCut = (int64_t)1 << 62;

// per-block result of one pass: 96 bytes
struct __attribute__((aligned(16))) LevelPartial {
    uint64_t key;        // post-commit block max: ((score+1) << 40) | (2^40-1 - global idx); 0 = nothing feasible
    uint32_t mt, ma;     // max prefer-count / affinity-sum over the block's post-commit feasible nodes
    uint32_t c_mt, c_ma; // how many feasible nodes hold those block maxima
    uint32_t nfeas;
    uint32_t n_top;      // how many nodes hold the block's maximum score
    uint32_t e_mt, e_ma; // plan pass: holders of st.mt_a / st.ma_a that their run-down exhausts
    int64_t T;              // plan pass: placements of the block's nodes at level st.lvl_M
    int64_t cut_mt, cut_ma; // plan pass: highest global index among those exhausted holders (-1 none)
    int64_t committed;      // placements committed by this block in this pass
    int64_t pad;
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
            if (x < p.nx && p.req[p.xcol[x]] > n.xa[x] - n.xr[x]) ok = false;
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
    return stat + dynamic_score(p, rc, n.a_cpu, n.a_mem, n.r_cpu, n.r_mem, n.z_cpu, n.z_mem);
}
template <int NX>
__device__ __forceinline__ int64_t node_score(const DevPod &p, const NodeRegs<NX> &n, int64_t stat) {
    return node_score<NX>(p, n, stat, make_rcp(n.a_cpu, n.a_mem));
}

__device__ __forceinline__ int64_t bcast_i64(int64_t v, int src) { return __shfl(v, src, 64); }
__device__ __forceinline__ int32_t bcast_i32(int32_t v, int src) { return __shfl(v, src, 64); }

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

// Run-downs, two regimes.  Every lane of the wave must call this (wave-uniform control flow).
// `mine` = this lane's node holds the level (feasible, score == M).
//   1. kSeqSteps placements evaluated by the lane itself (all level lanes of the wave in parallel): most
//      nodes leave the level after a few pods.
//   2. lanes still running (large nodes whose score moves once per tens of pods) are finished
//      WAVE-COOPERATIVELY, one node at a time: its registers are broadcast, lane l evaluates the node after
//      l+1 further placements (closed form), and one __ballot finds the first placement after which the node
//      is infeasible or scores < M: <= 2 steps for a 110-pod node instead of 110 dependent iterations.
// Returns this lane's run-down length (0 if !mine) and whether its node is still feasible afterwards.
constexpr int kSeqSteps = 6;

template <int NX>
__device__ __forceinline__ int32_t wave_run_down(const DevPod &p, const NodeRegs<NX> &n, int64_t stat, int64_t M, bool mine,
                                                 bool &feas_after) {
    const int lane = threadIdx.x & 63;
    int32_t my_j = 0;
    feas_after = true;
    if (!__ballot(mine)) return 0;
    NodeRegs<NX> cur = n;
    bool running = mine;
    const NodeRcp rc = make_rcp(n.a_cpu, n.a_mem); // allocatable never changes: one reciprocal pair per node
#pragma unroll 1
    for (int it = 0; it < kSeqSteps && __ballot(running); it++) {
        if (running) {
            node_apply<NX>(p, cur, 1);
            my_j++;
            feas_after = node_feasible<NX>(p, cur);
            running = feas_after && node_score<NX>(p, cur, stat, rc) >= M;
        }
    }
    uint64_t todo = __ballot(running);
    while (todo) {
        const int src = __ffsll((unsigned long long)todo) - 1;
        todo &= todo - 1;
        const NodeRegs<NX> base = bcast_node<NX>(cur, src);
        const int64_t bstat = bcast_i64(stat, src);
        const NodeRcp brc = make_rcp(base.a_cpu, base.a_mem);
        int32_t j = 0;
        bool f_end = true;
        for (int32_t k0 = 0;; k0 += 64) {
            NodeRegs<NX> t = base;
            node_apply<NX>(p, t, (int64_t)k0 + lane + 1);
            const bool f = node_feasible<NX>(p, t);
            const bool stop = !(f && node_score<NX>(p, t, bstat, brc) >= M);
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
    const longlong2 A0 = *reinterpret_cast<const longlong2 *>(c.alloc[0] + i0);
    const longlong2 A1 = *reinterpret_cast<const longlong2 *>(c.alloc[1] + i0);
    const longlong2 R0 = *reinterpret_cast<const longlong2 *>(c.req[0] + i0);
    const longlong2 R1 = *reinterpret_cast<const longlong2 *>(c.req[1] + i0);
    const longlong2 Z0 = *reinterpret_cast<const longlong2 *>(c.nz_mcpu + i0);
    const longlong2 Z1 = *reinterpret_cast<const longlong2 *>(c.nz_mem + i0);
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
    if (NX > 0) {
#pragma unroll
        for (int x = 0; x < NX; x++)
            if (x < p.nx) c.req[p.xcol[x]][i] = n.xr[x];
    }
}

__device__ __forceinline__ int64_t wave_max_i64(int64_t v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        int64_t o = __shfl_xor(v, off, 64);
        v = o > v ? o : v;
    }
    return v;
}
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
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

struct LevelArgs {
    DevCols c;
    DevPod p;
    const DevState *st;
    LevelPartial *partials;
    const int64_t *blockprefix; // [grid] exclusive prefix of the planned level's per-block placements
    int32_t *log;
    int64_t chunk; // nodes per block (multiple of kTile)
};

// Level nodes are sparse (a few % of a tile) and a run-down costs hundreds of VALU instructions per step,
// so they are COMPACTED: owners append their level nodes (registers + static score + index) to a block-wide
// LDS work list in canonical order (ballot/popcount prefix), and the first lanes of the block take one
// entry each -- the run-downs of a whole tile execute in a handful of densely packed waves instead of one
// or two active lanes in every wave.  The worker lane also commits the node (closed-form update, stores)
// and scores it in its new state; the owner skips it.
constexpr int kListCap = kThreads; // entries per round (one per worker lane)

template <int NX>
struct LevelList { // structure-of-arrays in LDS: conflict-free per-lane access
    int64_t f64[7 + 2 * NX][kListCap]; // a_cpu a_mem r_cpu r_mem z_cpu z_mem stat, xa[NX], xr[NX]
    int64_t idx[kListCap];             // node index inside the shard
    int32_t f32[3][kListCap];          // a_pods npods w
};

template <int NX>
__device__ __forceinline__ void list_put(LevelList<NX> &L, int pos, const NodeRegs<NX> &n, int64_t stat, int64_t idx) {
    L.f64[0][pos] = n.a_cpu, L.f64[1][pos] = n.a_mem, L.f64[2][pos] = n.r_cpu, L.f64[3][pos] = n.r_mem;
    L.f64[4][pos] = n.z_cpu, L.f64[5][pos] = n.z_mem, L.f64[6][pos] = stat;
#pragma unroll
    for (int x = 0; x < NX; x++) L.f64[7 + x][pos] = n.xa[x], L.f64[7 + NX + x][pos] = n.xr[x];
    L.idx[pos] = idx;
    L.f32[0][pos] = n.a_pods, L.f32[1][pos] = n.npods, L.f32[2][pos] = (int32_t)n.w;
}

template <int NX>
__device__ __forceinline__ void list_get(const LevelList<NX> &L, int pos, NodeRegs<NX> &n, int64_t &stat, int64_t &idx) {
    n.a_cpu = L.f64[0][pos], n.a_mem = L.f64[1][pos], n.r_cpu = L.f64[2][pos], n.r_mem = L.f64[3][pos];
    n.z_cpu = L.f64[4][pos], n.z_mem = L.f64[5][pos], stat = L.f64[6][pos];
    n.xa[0] = n.xr[0] = 0;
#pragma unroll
    for (int x = 0; x < NX; x++) n.xa[x] = L.f64[7 + x][pos], n.xr[x] = L.f64[7 + NX + x][pos];
    idx = L.idx[pos];
    n.a_pods = L.f32[0][pos], n.npods = L.f32[1][pos], n.w = (uint32_t)L.f32[2][pos];
}

// running reduction state of one thread over the nodes it scored
struct LevelAcc {
    uint64_t best = 0;
    int64_t top = -1; // maximum post-commit score seen by this thread and how many of its nodes hold it
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

#ifndef CCSIM_LEVEL_WAVES
#define CCSIM_LEVEL_WAVES 3
#endif
template <int NX>
__global__ __launch_bounds__(kThreads, CCSIM_LEVEL_WAVES) void k_level(LevelArgs a) {
    const DevState st = *a.st;
    if (st.done) return;
    constexpr int kWaves = kThreads / 64;
    __shared__ LevelList<NX> s_list;
    __shared__ uint64_t s_key[kWaves];
    __shared__ uint32_t s_u[8][kWaves];
    __shared__ int64_t s_l[4][kWaves];
    __shared__ int32_t s_cnt[2][kWaves]; // double-buffered by tile parity (no barrier between tiles when a tile has no level node)

    const uint32_t mt = (uint32_t)st.mt_a, ma = (uint32_t)st.ma_a;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int64_t lo = (int64_t)blockIdx.x * a.chunk;
    int64_t hi = lo + a.chunk;
    if (hi > a.c.n_pad) hi = a.c.n_pad;
    const bool plan_only = st.lvl_plan_only != 0;
    const bool commit_on = st.lvl_valid != 0 && !plan_only;
    const bool active = commit_on || plan_only;
    const bool ordered = commit_on && st.lvl_prefix != 0;
    const int64_t M = st.lvl_M;
    const uint64_t lt_mask = lane ? (~0ull >> (64 - lane)) : 0ull;

    LevelAcc acc;
    int64_t committed = 0;
    int64_t carry = ordered ? st.lvl_rank_prefix + a.blockprefix[blockIdx.x] : 0;
    int64_t T = 0, cut_mt = -1, cut_ma = -1; // plan pass
    uint32_t e_mt = 0, e_ma = 0;

    int par = 0;
    for (int64_t base = lo; base < hi; base += kTile, par ^= 1) {
        const int64_t i0 = base + 2 * tid;
        NodeRegs<NX> nd[2];
        load_pair<NX>(a.c, a.p, i0, nd);
        bool feas[2], lvl[2];
        int64_t sc[2], stat[2];
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const uint32_t cnt = (nd[k].w >> kStatCntShift) & kStatCntMask, aff = nd[k].w & kStatAffMask;
            stat[k] = static_score(a.p, cnt, aff, mt, ma);
            feas[k] = node_feasible<NX>(a.p, nd[k]);
            sc[k] = feas[k] ? node_score<NX>(a.p, nd[k], stat[k]) : -1;
            lvl[k] = active && feas[k] && sc[k] == M && (a.c.global_offset + i0 + k) <= st.lvl_cut;
        }
        // nodes that are not in the level keep their state: scored by their owner (before the worker phase, so
        // that the pair's registers are dead while the work list is processed)
#pragma unroll
        for (int k = 0; k < 2; k++) {
            if (feas[k] && !lvl[k]) {
                const uint32_t cnt = (nd[k].w >> kStatCntShift) & kStatCntMask, aff = nd[k].w & kStatAffMask;
                acc.add(sc[k], a.c.global_offset + i0 + k, cnt, aff);
            }
        }
        if (active) { // block-uniform
            // canonical-order position of this thread's level nodes in the block's work list
            const uint64_t b0 = __ballot(lvl[0]), b1 = __ballot(lvl[1]);
            if (lane == 0) s_cnt[par][wave] = __popcll(b0) + __popcll(b1);
            __syncthreads();
            int off = __popcll(b0 & lt_mask) + __popcll(b1 & lt_mask), total = 0;
#pragma unroll
            for (int w = 0; w < kWaves; w++) {
                if (w < wave) off += s_cnt[par][w];
                total += s_cnt[par][w];
            }
            const int pos0 = off, pos1 = off + (lvl[0] ? 1 : 0);
            for (int r0 = 0; r0 < total; r0 += kListCap) { // block-uniform; > 1 round only for dense levels
                if (lvl[0] && pos0 >= r0 && pos0 < r0 + kListCap) list_put<NX>(s_list, pos0 - r0, nd[0], stat[0], i0);
                if (lvl[1] && pos1 >= r0 && pos1 < r0 + kListCap) list_put<NX>(s_list, pos1 - r0, nd[1], stat[1], i0 + 1);
                __syncthreads();
                const int nwork = total - r0 < kListCap ? total - r0 : kListCap;
                const bool mine = tid < nwork;
                NodeRegs<NX> n;
                int64_t nstat = 0, nidx = 0;
                n.a_cpu = n.a_mem = n.r_cpu = n.r_mem = n.z_cpu = n.z_mem = 0, n.a_pods = n.npods = 0, n.w = 0;
#pragma unroll
                for (int x = 0; x < (NX > 0 ? NX : 1); x++) n.xa[x] = n.xr[x] = 0;
                if (mine) list_get<NX>(s_list, tid, n, nstat, nidx);
                bool fend = true;
                int64_t j = 0;
                if ((wave * 64) < nwork) j = wave_run_down<NX>(a.p, n, nstat, M, mine, fend); // wave-uniform
                const int64_t g = a.c.global_offset + nidx;
                const uint32_t cnt = (n.w >> kStatCntShift) & kStatCntMask, aff = n.w & kStatAffMask;
                if (plan_only) {
                    if (mine) {
                        T += j;
                        if (!fend) {
                            if (mt > 0 && cnt == mt) e_mt++, cut_mt = g > cut_mt ? g : cut_mt;
                            if (ma > 0 && aff == ma) e_ma++, cut_ma = g > cut_ma ? g : cut_ma;
                        }
                        acc.add(M, g, cnt, aff); // unchanged: still feasible at level M
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
                    if (mine) {
                        if (took > 0) {
                            node_apply<NX>(a.p, n, took);
                            store_dyn<NX>(a.c, a.p, nidx, n, (int32_t)took);
                            committed += took;
                            if (ordered && a.log) {
                                for (int64_t q = 0; q < took; q++) {
                                    const int64_t at = st.placed + pos + q;
                                    if (at < st.log_cap) a.log[at] = (int32_t)g;
                                }
                            }
                        }
                        if (node_feasible<NX>(a.p, n)) acc.add(node_score<NX>(a.p, n, nstat), g, cnt, aff);
                    }
                }
                __syncthreads(); // the list is rewritten next round / next tile
            }
        }
    }

    // ---- block reduce ----
    {
        const uint64_t wbest = wave_max_u64(acc.best);
        const int64_t wtop = wave_max_i64(acc.top);
        const uint32_t wmt = wave_max_u32(acc.mt), wma = wave_max_u32(acc.ma);
        const uint32_t wntop = wave_sum_u32(acc.top == wtop ? acc.ntop : 0u);
        const uint32_t wcmt = wave_sum_u32(acc.mt == wmt ? acc.cmt : 0u), wcma = wave_sum_u32(acc.ma == wma ? acc.cma : 0u);
        const uint32_t wnf = wave_sum_u32(acc.nfeas);
        committed = wave_sum_i64(committed);
        if (plan_only) { // block-uniform
            T = wave_sum_i64(T);
            cut_mt = wave_max_i64(cut_mt);
            cut_ma = wave_max_i64(cut_ma);
            e_mt = wave_sum_u32(e_mt);
            e_ma = wave_sum_u32(e_ma);
        }
        if (lane == 0) {
            s_key[wave] = wbest;
            s_u[0][wave] = wmt, s_u[1][wave] = wma, s_u[2][wave] = wcmt, s_u[3][wave] = wcma, s_u[4][wave] = wnf;
            s_u[5][wave] = wntop, s_u[6][wave] = e_mt, s_u[7][wave] = e_ma;
            s_l[0][wave] = T, s_l[1][wave] = committed, s_l[2][wave] = cut_mt, s_l[3][wave] = cut_ma;
        }
    }
    __syncthreads();
    if (tid == 0) {
        LevelPartial out{};
        out.cut_mt = out.cut_ma = -1;
#pragma unroll
        for (int w = 0; w < kWaves; w++) {
            const uint64_t kw = s_key[w];
            if (kw && (!out.key || key_score(kw) > key_score(out.key))) out.n_top = s_u[5][w];
            else if (kw && key_score(kw) == key_score(out.key)) out.n_top += s_u[5][w];
            out.key = kw > out.key ? kw : out.key;
            if (s_u[0][w] > out.mt) out.mt = s_u[0][w], out.c_mt = s_u[2][w]; else if (s_u[0][w] == out.mt) out.c_mt += s_u[2][w];
            if (s_u[1][w] > out.ma) out.ma = s_u[1][w], out.c_ma = s_u[3][w]; else if (s_u[1][w] == out.ma) out.c_ma += s_u[3][w];
            out.nfeas += s_u[4][w];
            out.e_mt += s_u[6][w], out.e_ma += s_u[7][w];
            out.T += s_l[0][w];
            out.committed += s_l[1][w];
            out.cut_mt = s_l[2][w] > out.cut_mt ? s_l[2][w] : out.cut_mt;
            out.cut_ma = s_l[3][w] > out.cut_ma ? s_l[3][w] : out.cut_ma;
        }
        a.partials[blockIdx.x] = out;
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
__device__ __forceinline__ void level_decide(DevState &st, const LevelAgg &g, bool want_log) {
    st.scans += 1;
    st.winner = -1;
    if (st.lvl_plan_only) { // the pass measured level lvl_M (nothing moved): now commit it, carefully
        st.lvl_plan_only = 0;
        st.lvl_valid = 1;
        int64_t cut = kNoCut;
        if (st.mt_a > 0 && g.e_mt == (int64_t)g.c_mt && g.cut_mt < cut) cut = g.cut_mt;
        if (st.ma_a > 0 && g.e_ma == (int64_t)g.c_ma && g.cut_ma < cut) cut = g.cut_ma;
        st.lvl_cut = cut;
        st.lvl_remaining = st.limit > 0 ? st.limit - st.placed : kNoCut;
        st.lvl_prefix = (want_log || (st.limit > 0 && st.placed + g.T > st.limit)) ? 1 : 0;
        return;
    }
    st.placed += g.committed;
    st.rounds += g.committed;
    st.lvl_valid = 0;
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
    if ((int32_t)g.mt != st.mt_a || (int32_t)g.ma != st.ma_a) { // scores above used stale constants: rescan
        st.mt_a = (int32_t)g.mt;
        st.ma_a = (int32_t)g.ma;
        return;
    }
    st.lvl_M = key_score(g.key);
    st.lvl_cut = kNoCut;
    st.lvl_remaining = kNoCut;
    st.lvl_prefix = 0;
    st.lvl_rank_prefix = 0;
    // Could anything end this level early?  The limit / the log need positions; a normalization maximum
    // can only move if ALL its feasible holders are exhausted, i.e. the level has at least that many nodes.
    const bool plan = want_log || st.limit > 0 || (g.mt > 0 && g.n_top >= (int64_t)g.c_mt) ||
                      (g.ma > 0 && g.n_top >= (int64_t)g.c_ma);
    if (plan) st.lvl_plan_only = 1;
    else st.lvl_valid = 1;
}

struct LevelFinalArgs {
    DevState *st;
    const LevelPartial *partials;
    int32_t n_partials;
    int64_t *blockprefix;
    XRec *xsend;       // distributed: this shard's record out
    const XRec *xrecv; // distributed: gathered records in
    int32_t n_ranks;   // 0 = single GPU
    int32_t rank;
    int32_t want_log;
};

// k_level_final: one block of kFinalThreads.  Reduces the per-block partials in ONE sweep (the block is
// latency-bound: every thread reads at most a couple of 96-byte partials); on one GPU also decides.  After a
// plan pass it leaves the exclusive per-block prefix of the level's placements in blockprefix[].
constexpr int kFinalThreads = 256;

__global__ __launch_bounds__(kFinalThreads) void k_level_final(LevelFinalArgs a) {
    if (a.st->done) return;
    constexpr int kWaves = kFinalThreads / 64;
    __shared__ uint64_t s_key[kWaves];
    __shared__ uint32_t s_u[4][kWaves];
    __shared__ int64_t s_l[8][kWaves];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const bool plan_pass = a.st->lvl_plan_only != 0;

    uint64_t key = 0;
    uint32_t mt = 0, cmt = 0, ma = 0, cma = 0;
    int64_t nf = 0, committed = 0, T = 0, e_mt = 0, e_ma = 0, cut_mt = -1, cut_ma = -1, ntop = 0;
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
        committed += q.committed;
        T += q.T, e_mt += q.e_mt, e_ma += q.e_ma;
        cut_mt = q.cut_mt > cut_mt ? q.cut_mt : cut_mt;
        cut_ma = q.cut_ma > cut_ma ? q.cut_ma : cut_ma;
    }
    {
        const uint64_t wkey = wave_max_u64(key);
        const uint32_t wmt = wave_max_u32(mt), wma = wave_max_u32(ma);
        const uint32_t wcmt = wave_sum_u32(mt == wmt ? cmt : 0u), wcma = wave_sum_u32(ma == wma ? cma : 0u);
        ntop = wave_sum_i64((key && wkey && key_score(key) == key_score(wkey)) ? ntop : 0);
        nf = wave_sum_i64(nf), committed = wave_sum_i64(committed);
        T = wave_sum_i64(T), e_mt = wave_sum_i64(e_mt), e_ma = wave_sum_i64(e_ma);
        cut_mt = wave_max_i64(cut_mt), cut_ma = wave_max_i64(cut_ma);
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
    // exclusive prefix over blocks (canonical node order == block order) of the level's placements
    __syncthreads();
    const int per = (a.n_partials + kFinalThreads - 1) / kFinalThreads;
    const int b0 = tid * per < a.n_partials ? tid * per : a.n_partials;
    const int b1 = (b0 + per < a.n_partials) ? b0 + per : a.n_partials;
    int64_t mine = 0;
    for (int i = b0; i < b1; i++) mine += a.partials[i].T;
    const int64_t incl = wave_incl_scan_i64(mine);
    if (lane == 63) s_l[0][wave] = incl;
    __syncthreads();
    int64_t run = incl - mine;
    for (int w = 0; w < wave; w++) run += s_l[0][w];
    for (int i = b0; i < b1; i++) {
        a.blockprefix[i] = run;
        run += a.partials[i].T;
    }
}

// k_level_decide (distributed): every rank reduces the gathered records identically; shards are
// contiguous ranges of the canonical order, so a rank's placements in a level come after those of all
// lower ranks.
__global__ void k_level_decide(LevelFinalArgs a) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    DevState st = *a.st;
    if (st.done) return;
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
    level_decide(st, g, a.want_log != 0);
    if (plan_pass) st.lvl_rank_prefix = before;
    *a.st = st;
}

} // namespace ccsim
