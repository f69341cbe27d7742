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
//   k_multi_commit  ONE wave, pods in order.  Pod j's true argmax over the state S_j (= S0 + the placements of pods
//                   0..j-1 of the window) is max(best UNTOUCHED node, best TOUCHED node): a node no earlier pod of the window
//                   was placed on has the state, feasibility and score the scan saw (the pod's own spread / anti-affinity
//                   state only changes through its OWN clones, and it appears once per window; the normalization maxima
//                   cannot move while an untouched holder remains), so the first untouched entry of the candidate list is
//                   the best untouched node; the <= 64 touched nodes live in the wave's lanes and are re-evaluated for
//                   pod j exactly.  Whenever that argument does not cover a pod (candidate list exhausted, both recorded
//                   keys of one workgroup touched, too few holders of a maximum left, an assumed maximum was wrong) the
//                   window ENDS before that pod and the next window starts with it: never a guess.
// Results are identical to the oracle's round-robin loop (oracle/ccref.c ccref_run_multi; tests/test_multi.py).
#pragma once
#include "ccsim_level.h"

namespace ccsim {

constexpr int kMWindowMax = 64;   // pods per window (= lanes of the commit wave holding per-pod rows)
constexpr int kMPodChunk = 8;     // pods per scan workgroup
constexpr int kMNodesPerThread = 8;
constexpr int kMBlockNodes = kThreads * kMNodesPerThread; // 2048
constexpr int kMTopK = 8;
constexpr int kMTouched = 64;     // touched nodes a window can hold (one per lane)
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
};

struct MState {
    int64_t placed, limit, rounds, windows, stops;
    int32_t done, stop_spec;
    int32_t next_pod;   // spec of the next cycle
    int32_t win_n;      // pods the pending window covers
    int32_t single_pod; // >= 0: ccsim_schedule_pod -- one cycle of that spec
    int32_t last_feasible, last_evaluated;
    int64_t winner;
    int64_t log_cap;
};

struct MPartial { // per (pod of the window, scan workgroup)
    uint64_t key1, key2; // best two nodes of the workgroup's 2048 for the pod: ((score+1) << 40) | ~index ; 0 = none
    uint32_t nfeas, mt, ma, c_mt, c_ma, pad;
};

struct MCand { // per pod of the window, after k_multi_select
    uint64_t key[kMTopK];
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
};

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

// minimum match count over the domains holding a counted node (CriticalPaths[c][0], filtering.go:298-305)
__device__ __forceinline__ int32_t m_tbl_min(const int32_t *tbl, const uint8_t *present, int ndom) {
    int32_t m = 0x7fffffff;
    for (int v = 1; v <= ndom; v++)
        if (present[v]) m = tbl[v] < m ? tbl[v] : m;
    return m;
}

// ------------------------------------------------------------------------------------------------------------------
// k_multi_scan: grid (node workgroups, pod chunks).
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void k_multi_scan(MultiArgs a) {
    const MState st = *a.st;
    if (st.done) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j0 = blockIdx.y * kMPodChunk;
    if (j0 >= st.win_n) return;
    const int jn = st.win_n - j0 < kMPodChunk ? st.win_n - j0 : kMPodChunk;
    const int64_t base = (int64_t)blockIdx.x * kMBlockNodes;

    __shared__ int32_t s_tbl[kMPodChunk][kMTsc][kMDomMax + 1];
    __shared__ int32_t s_min[kMPodChunk][kMTsc];
    __shared__ uint64_t s_k[2][kThreads / 64];
    __shared__ uint32_t s_u[5][kThreads / 64];
    // stage the chunk's spread tables; one lane per (pod, constraint) derives the minimum over present domains
    for (int i = tid; i < kMPodChunk * kMTsc * (kMDomMax + 1); i += kThreads) {
        const int jj = i / (kMTsc * (kMDomMax + 1)), c = (i / (kMDomMax + 1)) % kMTsc, v = i % (kMDomMax + 1);
        int32_t x = 0;
        if (jj < jn) {
            const MPod &q = a.pods[(st.next_pod + j0 + jj) % a.n_pods];
            if (c < q.n_tsc && v <= q.tsc_ndom[c]) x = a.tbl_pool[q.tsc_tbl[c] + v];
        }
        s_tbl[jj][c][v] = x;
    }
    __syncthreads();
    if (tid < kMPodChunk * kMTsc) {
        const int jj = tid / kMTsc, c = tid % kMTsc;
        int32_t m = 0x7fffffff;
        if (jj < jn) {
            const MPod &q = a.pods[(st.next_pod + j0 + jj) % a.n_pods];
            if (c < q.n_tsc) m = m_tbl_min(&s_tbl[jj][c][0], a.present_pool + q.tsc_tbl[c], q.tsc_ndom[c]);
        }
        s_min[jj][c] = m;
    }
    __syncthreads();

    // this thread's 8 nodes: narrow columns -> registers, once for all pods of the chunk
    int32_t a0[kMNodesPerThread], a1[kMNodesPerThread], r0[kMNodesPerThread], r1[kMNodesPerThread], z0[kMNodesPerThread], z1[kMNodesPerThread];
    int32_t ap[kMNodesPerThread], np[kMNodesPerThread], lv0[kMNodesPerThread], lv1[kMNodesPerThread];
#pragma unroll
    for (int k = 0; k < kMNodesPerThread; k++) {
        const int64_t i = base + k * kThreads + tid;
        const bool in = i < a.c.n_pad;
        a0[k] = in ? a.c.a32[0][i] : 0, a1[k] = in ? a.c.a32[1][i] : 0;
        r0[k] = in ? a.c.r32[0][i] : 0, r1[k] = in ? a.c.r32[1][i] : 0;
        z0[k] = in ? a.c.z32[0][i] : 0, z1[k] = in ? a.c.z32[1][i] : 0;
        ap[k] = in ? a.c.alloc_pods[i] : 0, np[k] = in ? a.c.pod_count[i] : 0;
        lv0[k] = in && a.tsc_label[0] ? a.tsc_label[0][i] : 0;
        lv1[k] = in && a.tsc_label[1] ? a.tsc_label[1][i] : 0;
    }

#pragma unroll 1
    for (int jj = 0; jj < jn; jj++) {
        const int pi = (st.next_pod + j0 + jj) % a.n_pods;
        const MPod q = a.pods[pi];
        const DevPod p = m_devpod(a.prof, q);
        const NarrowPod nq{q.req0, q.req1, q.nz0, q.nz1};
        const uint32_t mt = (uint32_t)q.mt_a, ma = (uint32_t)q.ma_a;
        const uint32_t *stat = a.stat_cls + (int64_t)q.cls * a.n_pad;
        const uint32_t *bits = q.anti ? a.anti_bits + (int64_t)pi * (a.n_pad / 32) : nullptr;
        uint64_t k1 = 0, k2 = 0;
        uint32_t nf = 0, mtb = 0, mab = 0, cmt = 0, cma = 0;
#pragma unroll
        for (int k = 0; k < kMNodesPerThread; k++) {
            const int64_t i = base + k * kThreads + tid;
            if (i >= a.c.n_pad) continue;
            const uint32_t w = stat[i];
            bool ok = (w >> kStatOkBit) && fits_narrow(p, nq, a0[k], a1[k], r0[k], r1[k], ap[k], np[k]);
            if (ok && bits) ok = !((bits[i >> 5] >> (i & 31)) & 1u); // satisfyPodAntiAffinity / existing pods' anti-affinity (filtering.go:352-379)
            for (int c = 0; c < q.n_tsc && ok; c++) {
                const int32_t v = q.tsc_slot[c] ? lv1[k] : lv0[k];
                ok = m_pts_check(q, c, v, s_tbl[jj][c][v < 0 || v > kMDomMax ? 0 : v], s_min[jj][c]) == 0;
            }
            if (!ok) continue;
            const uint32_t cnt = (w >> kStatCntShift) & kStatCntMask, aff = w & kStatAffMask;
            const int64_t total = static_score(p, cnt, aff, mt, ma) + dynamic_score_narrow(p, nq, a0[k], a1[k], r0[k], r1[k], z0[k], z1[k]);
            const uint64_t key = make_key(total, a.c.global_offset + i);
            if (key > k1) k2 = k1, k1 = key; else if (key > k2) k2 = key;
            nf++;
            if (cnt > mtb) mtb = cnt, cmt = 1; else if (cnt == mtb) cmt++;
            if (aff > mab) mab = aff, cma = 1; else if (aff == mab) cma++;
        }
        // workgroup top-2: wave maxima of k1, then of max(k2, the lanes' k1 that lost)
        const uint64_t w1 = wave_max_u64(k1);
        const uint64_t w2 = wave_max_u64(k1 == w1 ? k2 : k1); // (keys are unique: exactly one lane holds w1)
        const uint32_t wmt = wave_max_u32(mtb), wma = wave_max_u32(mab);
        const uint32_t wcmt = wave_sum_u32(mtb == wmt ? cmt : 0u), wcma = wave_sum_u32(mab == wma ? cma : 0u), wnf = wave_sum_u32(nf);
        __syncthreads(); // (s_k / s_u reuse across pods)
        if (lane == 0) s_k[0][wave] = w1, s_k[1][wave] = w2, s_u[0][wave] = wnf, s_u[1][wave] = wmt, s_u[2][wave] = wma, s_u[3][wave] = wcmt, s_u[4][wave] = wcma;
        __syncthreads();
        if (tid == 0) {
            MPartial o{};
            for (int x = 0; x < kThreads / 64; x++) {
                for (int h = 0; h < 2; h++) {
                    const uint64_t key = s_k[h][x];
                    if (key > o.key1) o.key2 = o.key1, o.key1 = key; else if (key > o.key2) o.key2 = key;
                }
                o.nfeas += s_u[0][x];
                if (s_u[1][x] > o.mt) o.mt = s_u[1][x], o.c_mt = s_u[3][x]; else if (s_u[1][x] == o.mt) o.c_mt += s_u[3][x];
                if (s_u[2][x] > o.ma) o.ma = s_u[2][x], o.c_ma = s_u[4][x]; else if (s_u[2][x] == o.ma) o.c_ma += s_u[4][x];
            }
            a.partials[(int64_t)(j0 + jj) * a.n_blocks + blockIdx.x] = o;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// k_multi_select: one wave per pod of the window -- the top-K of the workgroups' best-two keys, sorted, + the aggregates.
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_multi_select(MultiArgs a) {
    const MState st = *a.st;
    const int j = blockIdx.x, lane = threadIdx.x;
    if (st.done || j >= st.win_n) return;
    const MPartial *pp = a.partials + (int64_t)j * a.n_blocks;
    uint64_t best[kMTopK];
#pragma unroll
    for (int k = 0; k < kMTopK; k++) best[k] = 0;
    uint32_t nf = 0, mt = 0, ma = 0, cmt = 0, cma = 0;
    // every lane keeps the sorted top-K of its share, then K rounds of wave max + pop
    for (int b = lane; b < a.n_blocks; b += 64) {
        const MPartial q = pp[b];
        for (int h = 0; h < 2; h++) {
            uint64_t key = h ? q.key2 : q.key1;
#pragma unroll
            for (int k = 0; k < kMTopK; k++)
                if (key > best[k]) { const uint64_t t = best[k]; best[k] = key; key = t; }
        }
        nf += q.nfeas;
        if (q.mt > mt) mt = q.mt, cmt = q.c_mt; else if (q.mt == mt) cmt += q.c_mt;
        if (q.ma > ma) ma = q.ma, cma = q.c_ma; else if (q.ma == ma) cma += q.c_ma;
    }
    const uint32_t wmt = wave_max_u32(mt), wma = wave_max_u32(ma);
    const uint32_t wcmt = wave_sum_u32(mt == wmt ? cmt : 0u), wcma = wave_sum_u32(ma == wma ? cma : 0u), wnf = wave_sum_u32(nf);
    MCand out{};
    int n = 0;
#pragma unroll 1
    for (int k = 0; k < kMTopK; k++) {
        const uint64_t m = wave_max_u64(best[0]);
        if (!m) break;
        out.key[n++] = m;
        if (best[0] == m) { // unique keys: one lane pops
#pragma unroll
            for (int x = 0; x + 1 < kMTopK; x++) best[x] = best[x + 1];
            best[kMTopK - 1] = 0;
        }
    }
    if (lane == 0) {
        out.n = n, out.nfeas = (int32_t)wnf, out.mt = wmt, out.ma = wma, out.c_mt = wcmt, out.c_ma = wcma;
        a.cands[j] = out;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// k_multi_commit: one wave; pods of the window in order (see the header comment).
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_multi_commit(MultiArgs a) {
    MState st = *a.st;
    if (st.done) return;
    const int lane = threadIdx.x;
    const int W = st.win_n;
    __shared__ uint32_t s_w[kMTouched][kMWindowMax];  // static word of touched node t for pod j
    __shared__ uint8_t s_f[kMTouched][kMWindowMax];   // bit0 anti-affinity hit, bit 1+c counted for spread constraint c
    __shared__ int32_t s_min[kMWindowMax][kMTsc];

    // lane j: pod j of the window -- the minimum of its spread tables (its own clones only: fixed during the window)
    if (lane < W) {
        const MPod &q = a.pods[(st.next_pod + lane) % a.n_pods];
        for (int c = 0; c < kMTsc; c++)
            s_min[lane][c] = c < q.n_tsc ? m_tbl_min(a.tbl_pool + q.tsc_tbl[c], a.present_pool + q.tsc_tbl[c], q.tsc_ndom[c]) : 0;
    }
    __syncthreads();

    // lane t: touched node t
    int64_t t_idx = -1; // shard-local index
    int32_t ta0 = 0, ta1 = 0, tr0 = 0, tr1 = 0, tz0 = 0, tz1 = 0, tap = 0, tnp = 0, tl0 = 0, tl1 = 0, tplaced = 0;
    int nt = 0;
    int committed = 0;
    int stop_reason = 0; // 0 window complete

#pragma unroll 1
    for (int j = 0; j < W; j++) {
        const int pi = (st.next_pod + j) % a.n_pods;
        const MPod q = a.pods[pi];
        const MCand cd = a.cands[j];
        // an assumed normalization maximum was wrong: the pod's scores are invalid -- fix it, end the window here
        if ((int32_t)cd.mt != q.mt_a || (int32_t)cd.ma != q.ma_a) {
            if (lane == 0) a.pods[pi].mt_a = (int32_t)cd.mt, a.pods[pi].ma_a = (int32_t)cd.ma;
            // (the later pods of the window get their maxima fixed too, so that one window repairs a whole cycle of specs)
            for (int jj = j + 1 + lane; jj < W; jj += 64) {
                const int pj = (st.next_pod + jj) % a.n_pods;
                a.pods[pj].mt_a = (int32_t)a.cands[jj].mt, a.pods[pj].ma_a = (int32_t)a.cands[jj].ma;
            }
            stop_reason = 1;
            break;
        }
        if (cd.nfeas == 0) { // schedule_one.go:448-454: no node fits -- touched nodes only lost room, the pod's own state did not move
            st.done = DONE_UNSCHEDULABLE, st.stop_spec = pi, st.rounds += 1, st.last_feasible = 0, st.winner = -1;
            stop_reason = 2;
            break;
        }
        // a maximum whose feasible holders could all be among the touched nodes may have moved
        if ((cd.mt > 0 && (int)cd.c_mt <= nt) || (cd.ma > 0 && q.w_aff && (int)cd.c_ma <= nt)) {
            stop_reason = 3;
            break;
        }
        const DevPod p = m_devpod(a.prof, q);
        const NarrowPod nq{q.req0, q.req1, q.nz0, q.nz1};
        // touched nodes, re-evaluated for this pod in their current state
        uint64_t tkey = 0;
        if (lane < nt) {
            const uint32_t w = s_w[lane][j];
            const uint32_t f = s_f[lane][j];
            bool ok = (w >> kStatOkBit) && fits_narrow(p, nq, ta0, ta1, tr0, tr1, tap, tnp) && !(f & 1u);
            for (int c = 0; c < q.n_tsc && ok; c++) {
                const int32_t v = q.tsc_slot[c] ? tl1 : tl0;
                ok = m_pts_check(q, c, v, v ? a.tbl_pool[q.tsc_tbl[c] + v] : 0, s_min[j][c]) == 0;
            }
            if (ok) {
                const uint32_t cnt = (w >> kStatCntShift) & kStatCntMask, aff = w & kStatAffMask;
                const int64_t total = static_score(p, cnt, aff, cd.mt, cd.ma) + dynamic_score_narrow(p, nq, ta0, ta1, tr0, tr1, tz0, tz1);
                tkey = make_key(total, a.c.global_offset + t_idx);
            }
        }
        tkey = wave_max_u64(tkey);
        // first untouched entry of the candidate list; two skipped entries of one scan workgroup hide that workgroup's third
        uint64_t ukey = 0;
        bool unknown = false;
        {
            int64_t skipped_blk[2] = {-1, -1};
            int k = 0;
            for (; k < cd.n; k++) {
                const int64_t li = key_index(cd.key[k]) - a.c.global_offset;
                const bool touched = __ballot(lane < nt && t_idx == li) != 0;
                if (!touched) { ukey = cd.key[k]; break; }
                const int64_t blk = li / kMBlockNodes;
                if (blk == skipped_blk[0] || blk == skipped_blk[1]) { unknown = true; break; }
                if (skipped_blk[0] < 0) skipped_blk[0] = blk; else if (skipped_blk[1] < 0) skipped_blk[1] = blk; else { unknown = true; break; }
            }
            // the list ran out while feasible untouched nodes may exist beyond it
            if (!unknown && !ukey && k >= cd.n && cd.nfeas > cd.n) unknown = true;
        }
        if (unknown) {
            stop_reason = 4;
            break;
        }
        const uint64_t win = tkey > ukey ? tkey : ukey;
        if (!win) { // every feasible node of the scan is touched and none of them fits any more: let the next scan say so
            stop_reason = 5;
            break;
        }
        const int64_t g = key_index(win), li = g - a.c.global_offset;
        int slot = __ffsll((unsigned long long)__ballot(lane < nt && t_idx == li)) - 1;
        if (slot < 0) { // first placement on this node in the window: it becomes a touched node
            if (nt >= kMTouched) {
                stop_reason = 6;
                break;
            }
            slot = nt++;
            // lane l < W gathers the node's static word / anti-affinity bit / inclusion bits for pod l of the window
            if (lane < W) {
                const int pl = (st.next_pod + lane) % a.n_pods;
                const MPod &ql = a.pods[pl];
                s_w[slot][lane] = a.stat_cls[(int64_t)ql.cls * a.n_pad + li];
                uint32_t f = 0;
                if (ql.anti) f |= (a.anti_bits[(int64_t)pl * (a.n_pad / 32) + (li >> 5)] >> (li & 31)) & 1u;
                for (int c = 0; c < ql.n_tsc; c++) {
                    const int32_t v = ql.tsc_slot[c] ? a.tsc_label[1][li] : a.tsc_label[0][li];
                    bool all = true; // counted iff the node has ALL the pod's hard keys and passes the inclusion policies (filtering.go:267-277)
                    for (int c2 = 0; c2 < ql.n_tsc; c2++) all = all && (ql.tsc_slot[c2] ? a.tsc_label[1][li] : a.tsc_label[0][li]) != 0;
                    const bool inc = ql.tsc_inc[c] < 0 || a.inc_pool[(int64_t)ql.tsc_inc[c] * a.n_pad + li] != 0;
                    if (v && all && inc) f |= 2u << c;
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
        }
        // NodeInfo.update (types.go:409-428) on the winner's lane; the pod's own plugin state
        if (lane == slot) {
            tr0 += q.req0, tr1 += q.req1, tz0 += q.nz0, tz1 += q.nz1, tnp += 1, tplaced += 1;
            const uint32_t f = s_f[slot][j];
            for (int c = 0; c < q.n_tsc; c++)
                if (((f >> (1 + c)) & 1u) && q.tsc_self[c]) a.tbl_pool[q.tsc_tbl[c] + (q.tsc_slot[c] ? tl1 : tl0)] += 1;
            if (q.anti) {
                atomicOr(&a.anti_bits[(int64_t)pi * (a.n_pad / 32) + (li >> 5)], 1u << (li & 31));
                s_f[slot][j] |= 1u;
            }
            const int64_t at = st.placed + committed;
            if (a.log && at < st.log_cap) a.log[at] = (int32_t)g;
            a.per_spec[pi] += 1;
        }
        committed++;
        st.winner = g;
        st.last_feasible = cd.nfeas;
        if (st.limit > 0 && st.placed + committed >= st.limit) {
            st.done = DONE_LIMIT;
            stop_reason = 7;
            break;
        }
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
    if (lane == 0) {
        st.placed += committed, st.rounds += committed;
        st.windows += 1, st.stops += stop_reason != 0 && stop_reason != 7 && stop_reason != 2;
        st.next_pod = st.single_pod >= 0 ? st.next_pod : (int32_t)((st.next_pod + committed) % a.n_pods);
        // the next window: as many pods as fit (all different specs; not beyond the limit)
        int64_t wn = a.window < a.n_pods ? a.window : a.n_pods;
        if (st.limit > 0 && st.limit - st.placed < wn) wn = st.limit - st.placed;
        if (st.single_pod >= 0) wn = st.done || committed ? 0 : 1;
        st.win_n = (int32_t)(wn < 0 ? 0 : wn);
        if (st.single_pod >= 0 && committed) st.done = st.done ? st.done : DONE_LIMIT; // one cycle asked for, one done
        *a.st = st;
    }
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
