// ccsim_coupled.h -- ONE template with topology-coupled plugins (PodTopologySpread, InterPodAffinity), resolved in exact
// WINDOWS of placements per node pass instead of one pass per placement.
//
// The reference recomputes, every scheduling cycle, calPreFilterState over all nodes (P/podtopologyspread/filtering.go:235-308),
// the inter-pod affinity maps (P/interpodaffinity/filtering.go:204-309) and both plugins' scores
// (P/podtopologyspread/scoring.go:118-265, P/interpodaffinity/scoring.go:128-290).  The sequential mode of this engine does
// the same with one k_scan + one k_final per placement: 2 launches and ~20 us per pod whatever the node count.
//
// The argument (tests/coupled_model.py is its executable form, checked placement by placement against the oracle on the CPU;
// `device_plan=True` there is exactly the variant implemented here).  For one template a node's verdict and TotalScore split
// into a NODE-LOCAL part A(n) -- static filters, NodePorts clamp, NodeResourcesFit, and the TaintToleration / NodeAffinity /
// ImageLocality / LeastAllocated / BalancedAllocation scores under assumed normalization maxima -- that changes only when a
// clone lands on n, and a COUPLED part that reads per-domain tables at the node's own topology values plus cycle-wide scalars.
// Nodes that agree on everything the coupled part reads form a CLASS: same topology value for every key shared by several
// nodes; same table ENTRIES for every key whose values are unique per node (kubernetes.io/hostname: the "domain" is the node
// and its entries are node state).  Inside a class every node untouched since the pass has the same coupled verdict and raw
// scores in every later cycle, so the class's best node is the head of its member list sorted by (A desc, index asc).
// One pass therefore serves a window of W cycles:
//   k_cw_scan    every node: local feasibility, A(n), the class tuple -> exact class table (hash + full-tuple check);
//                per class: members, TaintToleration / NodeAffinity maxima
//   k_cw_top     per block of 1024 nodes: the L best members of every class (L rounds of LDS atomic-max), holders of the
//                class maxima
//   k_cw_merge   per class: the L best members over all blocks
//   k_cw_decide  ONE workgroup, the cycle loop in ONE wave (no barriers inside a cycle): shared-key tables, class records and
//                the touched nodes live in LDS; per cycle the hard constraints' minima, every candidate's coupled verdict, the
//                feasible / candidate-domain counts behind the PodTopologySpread weights, raw-score min / max, totals, argmax,
//                commit (NodeInfo.update S/framework/types.go:409-428 + table updates).
// Whatever the window cannot know ends it BEFORE the cycle in question (the next pass starts from the exact state): a class
// whose next head is not among the L members kept; normalization maxima that differ from the ones A was computed with; the
// last node at the minimum of a hard constraint over a unique key taken.  What the mode cannot represent at all (more than
// kCwMaxClasses classes, a tuple beyond kCwTuple components, two tuples with one hash, an entry beyond int32) sets
// DevState::cw_fallback and the run continues in the one-pass-per-placement loop -- an optimisation with an exact fallback,
// never a different answer.
#pragma once
#include "ccsim_kernels.h"
#include <type_traits>

namespace ccsim {

constexpr int kCwTuple = 20;        // int32 components of a class tuple, at FIXED positions (registers in the scan, not a scratch frame):
                                    // hard constraint c -> c (c < 4); soft c -> 4 + c; inter-pod key 0 -> 8..12, 1 -> 13..17, 2 -> 18, 3 -> 19
constexpr int kCwMaxCons = 4, kCwKeyPos0 = 8, kCwKeyPos1 = 13, kCwKeyPos2 = 18, kCwKeyPos3 = 19;
constexpr int kCwMaxClasses = 256;  // classes per window
constexpr int kCwSlots = 1024;      // global open-addressing class table (power of two)
constexpr int kCwBlockSlots = 256;  // per-block LDS class table of the scan (power of two)
constexpr int kCwMaxList = 64;      // L
constexpr int kCwMaxWindow = 256;   // W of the general decide kernel (a touched node keeps its whole tuple in LDS)
constexpr int kCwFastWindow = 4096; // W of the lane-per-candidate kernel (a touched node is (index, clones)): 64 classes x the 64 list members (round 5; 2048 = 64 x 32 in round 4)
                                    // its staging area holds for each carry 2048 cycles per pass (round 4; 1024 before: the pass's fixed work --
                                    // scan, top, merges: ~280 us at 1M nodes / 64 zones -- halves per placement)
constexpr int kCwThreads = 256, kCwPerThread = 4, kCwTile = kCwThreads * kCwPerThread;
constexpr int kCwMaxKeys = 6400;    // k_cw_merge stages blocks x L keys of one class in LDS (32 KiB)
constexpr int kCwLdsI32 = 4096;     // k_cw_decide: int32 words of shared-key tables (hard / soft counts, presence flags, candidate bitmaps)
constexpr int kCwLdsI64 = 2048;     // ... int64 words (four InterPodAffinity tables per shared key)
constexpr int kCwCtlClasses = 0, kCwCtlGiveUp = 1, kCwCtlFastDone = 2;

// Where each plugin input sits in the class tuple, and where its table lives in the decide kernel's LDS (host-built per pod spec).
struct CwPlan {
    int32_t n_comp;
    int32_t h_comp[kMaxTsc], h_unique[kMaxTsc], h_off[kMaxTsc], h_len[kMaxTsc], h_pres[kMaxTsc]; // hard constraint c
    int32_t s_comp[kMaxTsc], s_off[kMaxTsc], s_len[kMaxTsc], s_bm[kMaxTsc];                      // soft constraint c (hostname: per node)
    int32_t k_comp[kMaxIpaKeys], k_unique[kMaxIpaKeys], k_off[kMaxIpaKeys], k_len[kMaxIpaKeys];   // InterPodAffinity key k
    int32_t i32_words, i64_words;
    int32_t window, list_len;
    int32_t sweep; // k_cw_decide_fast resolves whole ROUNDS of placements at once where it can (CCSIM_CW_SWEEP=0: one placement per step, the A/B and test knob)
    const int32_t *h_present[kMaxTsc]; // domain-presence flags of hard constraint c (k_pts_init)
};

struct __attribute__((aligned(16))) CwClass {
    int32_t tuple[kCwTuple];
    uint32_t nf;     // node-feasible members
    uint32_t mt, ma; // max PreferNoSchedule count / preferred-affinity sum over them
    uint32_t ht, ha; // how many members hold those maxima
    uint32_t id;     // dense class number (claim order)
    uint32_t pad[2];
};

struct __attribute__((aligned(16))) CwPart {
    uint32_t nf, mt, ht, ma, ha, pad[3];
};
// (max, holders) pairs combine as: the larger maximum with its holders; equal maxima add their holders
__device__ __forceinline__ void cw_part_add(CwPart &x, const CwPart &y) {
    x.nf += y.nf;
    if (y.mt > x.mt) x.mt = y.mt, x.ht = y.ht; else if (y.mt == x.mt) x.ht += y.ht;
    if (y.ma > x.ma) x.ma = y.ma, x.ha = y.ha; else if (y.ma == x.ma) x.ha += y.ha;
}

struct CwWork {
    unsigned long long *keys; // [kCwSlots] tuple hash of the class in the slot, 0 = empty
    uint32_t *ready;          // [kCwSlots] the claimant has written the tuple
    CwClass *cls;             // [kCwSlots]
    uint32_t *ctl;            // [16]
    int32_t *slot_of_id;      // [kCwMaxClasses]
    int32_t *node_slot;       // [n_pad] class slot of a node-feasible node, -1 otherwise
    int32_t *node_A;          // [n_pad] its node-local score
    int32_t *node_A1;         // [n_pad] ... after ONE more clone of the template (-1: the node could not take a second one)
    unsigned long long *prof; // [16] k_cw_decide: 10 ns ticks per phase (measurement runs)
    unsigned long long *top;  // [blocks][kCwMaxClasses][L]
    unsigned long long *top2; // [groups][kCwMaxClasses][L]: the lists of `merge_group` blocks merged (snapshots with more blocks than one merge stages)
    int32_t merge_group;      // blocks one k_cw_merge workgroup merges (merge_group * L keys fit its LDS)
    unsigned long long *lists; // [kCwMaxClasses][L]
    unsigned long long *umin; // [blocks][kMaxTsc] unique-key hard constraints: (minimum << 32) | counted nodes at the minimum
    int32_t n_blocks;
    // per (block, class) and per (merge group, class): members, maxima, holders of the maxima -- written by k_cw_top, reduced by k_cw_merge
    // into CwClass.  (Round 4 accumulated them with atomics on the class record: ~4 000 same-address device-scope atomics per class and
    // pass at 1M nodes, from every XCD -- that serialization, not the 36 MB of columns, was the scan's 130-170 us.)
    CwPart *part;  // [blocks][kCwMaxClasses]
    CwPart *part2; // [groups][kCwMaxClasses]
    // node-range shards (round 5; "windows on shards" below): this rank's window record, the gathered records of all ranks, and what
    // k_cw_xunify makes of them -- the cluster's classes and their merged lists, identical on every rank
    unsigned char *xsend, *xrecv;
    uint32_t *xhdr;  // [4] classes of the cluster, give-up flag
    CwClass *xcls;   // [kCwXClasses]
    uint4 *xent;     // [kCwXClasses][kCwMaxList] staged list entries {key lo, key hi, A after one more clone, meta}
    int32_t x_ranks, x_rank;
};


__device__ __forceinline__ uint64_t cw_hash(const int32_t *t, int n) {
    uint64_t h = 0x9e3779b97f4a7c15ull;
#pragma unroll
    for (int i = 0; i < kCwTuple; i++)
        if (i < n) {
            h ^= (uint64_t)(uint32_t)t[i] + 0x9e3779b97f4a7c15ull + (h << 6) + (h >> 2);
            h *= 0xff51afd7ed558ccdull;
            h ^= h >> 33;
        }
    return h ? h : 1ull;
}

__device__ __forceinline__ uint32_t ld_u32_agent(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT); }

struct CwScanArgs {
    DevCols c;
    DevPod p;
    const DevState *st;
    DevPts pts;
    DevSoft soft;
    DevIpa ipa;
    CwPlan plan;
    CwWork w;
};

// int64 table entry -> tuple component; beyond int32 the windowed mode gives up (exactly, not approximately)
__device__ __forceinline__ int32_t cw_narrow_entry(int64_t v, uint32_t *giveup) {
    if (v > 0x3fffffffll || v < -0x3fffffffll) {
        __hip_atomic_store(giveup, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return 0;
    }
    return (int32_t)v;
}

// ------------------------------------------------------------------------------------------------------------------------
// k_cw_scan: one block = kCwTile consecutive nodes.
// ------------------------------------------------------------------------------------------------------------------------
template <int NX, bool NARROW>
__global__ __launch_bounds__(kCwThreads) void k_cw_scan(CwScanArgs a) {
    const DevState &st = *a.st;
    if (st.done || st.cw_fallback) return;
    __shared__ unsigned long long b_key[kCwBlockSlots];
    __shared__ int32_t b_tuple[kCwBlockSlots][kCwTuple];
    __shared__ int32_t b_gslot[kCwBlockSlots];
    __shared__ uint32_t b_um[kMaxTsc], b_uc[kMaxTsc];
    const int tid = threadIdx.x;
    for (int s = tid; s < kCwBlockSlots; s += kCwThreads) b_key[s] = 0, b_gslot[s] = -1;
    if (tid < kMaxTsc) b_um[tid] = 0x7fffffffu, b_uc[tid] = 0;
    __syncthreads();
    const uint32_t mt = (uint32_t)st.mt_a, ma = (uint32_t)st.ma_a;
    const NarrowPod npod = narrow_pod(a.p, a.c.mem_shift);
    uint32_t *giveup = a.w.ctl + kCwCtlGiveUp;
    int lslot[kCwPerThread];
    int32_t lA[kCwPerThread], lA1[kCwPerThread];

    int32_t tup4[kCwPerThread][kCwTuple]; // (the four nodes of a thread are worked on together: their loads overlap)
    bool mine4[kCwPerThread];
#pragma unroll
    for (int j = 0; j < kCwPerThread; j++) {
        const int64_t i = (int64_t)blockIdx.x * kCwTile + (int64_t)j * kCwThreads + tid;
        lslot[j] = -1, lA[j] = 0, lA1[j] = -1;
        bool feas = false;
        int32_t(&tup)[kCwTuple] = tup4[j];
#pragma unroll
        for (int q = 0; q < kCwTuple; q++) tup[q] = 0;
        uint32_t cnt = 0, aff = 0;
        if (i < a.c.n) {
            const uint32_t w = a.c.stat[i];
            const uint32_t eb = a.pts.n ? a.pts.elig[i] : 1u;
            // unique-key hard constraints: the minimum over the counted nodes (filtering.go:298-305), whatever their feasibility
            for (int c = 0; c < a.pts.n; c++)
                if (a.plan.h_unique[c]) {
                    const int32_t v = a.pts.label[c][i];
                    if (v && (eb & 1u) && ((eb >> (1 + c)) & 1u)) {
                        const uint32_t m = (uint32_t)a.pts.tbl[c][v];
                        const uint32_t old = atomicMin(&b_um[c], m);
                        (void)old;
                    }
                }
            feas = (w >> kStatOkBit) != 0;
            int64_t a_cpu = 0, a_mem = 0, r_cpu = 0, r_mem = 0, z_cpu = 0, z_mem = 0;
            int32_t na0 = 0, na1 = 0, nr0 = 0, nr1 = 0, nz0 = 0, nz1 = 0;
            const int32_t a_pods = a.c.alloc_pods[i], npods = a.c.pod_count[i];
            int64_t xa[NX > 0 ? NX : 1], xr[NX > 0 ? NX : 1];
            if (NARROW) {
                na0 = a.c.a32[0][i], na1 = a.c.a32[1][i], nr0 = a.c.r32[0][i], nr1 = a.c.r32[1][i], nz0 = a.c.z32[0][i], nz1 = a.c.z32[1][i];
                feas = feas && fits_narrow(a.p, npod, na0, na1, nr0, nr1, a_pods, npods);
            } else {
                a_cpu = a.c.alloc[0][i], a_mem = a.c.alloc[1][i], r_cpu = a.c.req[0][i], r_mem = a.c.req[1][i];
                z_cpu = a.c.nz_mcpu[i], z_mem = a.c.nz_mem[i];
                feas = feas && fits_core(a.p, a_cpu, a_mem, r_cpu, r_mem, a_pods, npods);
#pragma unroll
                for (int x = 0; x < (NX > 0 ? NX : 1); x++) {
                    xa[x] = xr[x] = 0;
                    if (NX > 0 && x < a.p.nx) {
                        const int col = a.p.xcol[x];
                        xa[x] = a.c.alloc[col][i], xr[x] = a.c.req[col][i];
                        const int64_t rq = a.p.req[col];
                        if (a.p.fit_enabled && !a.p.all_zero_req && rq > 0 && rq > xa[x] - xr[x]) feas = false;
                    }
                }
            }
            // a node without one of the hard constraints' keys never passes PodTopologySpread (filtering.go:325-329): no class
            if (a.pts.n && !(eb & 1u)) feas = false;
            if (feas) {
                cnt = (w >> kStatCntShift) & kStatCntMask, aff = w & kStatAffMask;
                const uint32_t img = (w >> kStatImgShift) & kStatImgMask;
                const int64_t A = static_score(a.p, cnt, aff, img, mt, ma) +
                                  (NARROW ? dynamic_score_narrow(a.p, npod, na0, na1, nr0, nr1, nz0, nz1)
                                          : (NX > 0 && a.p.gen_score ? dynamic_score_gen<NX>(a.p, a_cpu, a_mem, r_cpu, r_mem, z_cpu, z_mem, xa, xr)
                                                                     : dynamic_score(a.p, make_rcp(a_cpu, a_mem), a_cpu, a_mem, r_cpu, r_mem, z_cpu, z_mem)));
                lA[j] = (int32_t)A;
                // the node after one more clone (NodeInfo.update, types.go:409-428): what the window needs of a node it has placed on
                {
                    bool f1;
                    int64_t A1;
                    if (NARROW) {
                        f1 = fits_narrow(a.p, npod, na0, na1, nr0 + npod.req0, nr1 + npod.req1, a_pods, npods + 1);
                        A1 = dynamic_score_narrow(a.p, npod, na0, na1, nr0 + npod.req0, nr1 + npod.req1, nz0 + npod.nz0, nz1 + npod.nz1);
                    } else {
                        const int64_t q0 = r_cpu + a.p.req[0], q1 = r_mem + a.p.req[1], y0 = z_cpu + a.p.nz_mcpu, y1 = z_mem + a.p.nz_mem;
                        f1 = fits_core(a.p, a_cpu, a_mem, q0, q1, a_pods, npods + 1);
                        int64_t xr1[NX > 0 ? NX : 1];
#pragma unroll
                        for (int x = 0; x < (NX > 0 ? NX : 1); x++) {
                            xr1[x] = 0;
                            if (NX > 0 && x < a.p.nx) {
                                const int64_t rq = a.p.req[a.p.xcol[x]];
                                xr1[x] = xr[x] + rq;
                                if (a.p.fit_enabled && !a.p.all_zero_req && rq > 0 && rq > xa[x] - xr1[x]) f1 = false;
                            }
                        }
                        A1 = NX > 0 && a.p.gen_score ? dynamic_score_gen<NX>(a.p, a_cpu, a_mem, q0, q1, y0, y1, xa, xr1)
                                                     : dynamic_score(a.p, make_rcp(a_cpu, a_mem), a_cpu, a_mem, q0, q1, y0, y1);
                    }
                    lA1[j] = f1 ? (int32_t)(A1 + static_score(a.p, cnt, aff, img, mt, ma)) : -1;
                }
                // ---- the class tuple: everything the coupled plugins read of this node (compile-time positions)
#pragma unroll
                for (int c = 0; c < kCwMaxCons; c++)
                    if (c < a.pts.n) {
                        const int32_t v = a.pts.label[c][i];
                        tup[c] = a.plan.h_unique[c] ? ((v ? 1 : 0) | (v ? a.pts.tbl[c][v] << 1 : 0)) : v;
                    }
#pragma unroll
                for (int c = 0; c < kCwMaxCons; c++)
                    if (c < a.soft.n) {
                        const int32_t v = a.soft.label[c][i];
                        int32_t t = v;
                        if (a.soft.is_hostname[c]) {
                            const int32_t ct = (a.soft.existing[c] ? a.soft.existing[c][i] : 0) + (a.soft.self_match[c] ? npods - a.soft.pod_count0[i] : 0);
                            t = (v ? 1 : 0) | (ct << 1);
                        }
                        tup[kCwMaxCons + c] = t;
                    }
                if (a.ipa.on) {
#pragma unroll
                    for (int k = 0; k < kMaxIpaKeys; k++)
                        if (k < a.ipa.n_keys) {
                            const int32_t v = a.ipa.label[k][i];
                            constexpr int kPos[4] = {kCwKeyPos0, kCwKeyPos1, kCwKeyPos2, kCwKeyPos3};
                            const int q = kPos[k];
                            if (k >= 2 || !a.plan.k_unique[k]) tup[q] = v; // (a unique key beyond the second: no plan, see cw_make_plan)
                            else {
                                tup[q] = v ? 1 : 0;
                                tup[q + 1] = v ? cw_narrow_entry(a.ipa.aff[k][v], giveup) : 0;
                                tup[q + 2] = v ? cw_narrow_entry(a.ipa.anti[k][v], giveup) : 0;
                                tup[q + 3] = v ? cw_narrow_entry(a.ipa.exist[k][v], giveup) : 0;
                                tup[q + 4] = v ? cw_narrow_entry(a.ipa.score[k][v], giveup) : 0;
                                // The domain of a unique key is the node itself.  If it already holds a pod that one of the incoming pod's
                                // required anti-affinity terms on this key matches, or a pod whose own anti-affinity terms match the incoming
                                // pod, InterPodAffinity rejects the node in this cycle and in every later one (filtering.go:352-379; the
                                // counts only grow): it is no candidate any more and forms no class.  (Round 4: the nodes that took a clone
                                // of a pod with hostname anti-affinity doubled the classes -- one more per zone -- and pushed the synthetic
                                // cluster's 64 zones past what the lane-per-candidate kernel holds.)
                                if (a.ipa.filter_on && v) {
                                    int n_anti_k = 0;
                                    for (int t = 0; t < a.ipa.n_anti; t++) n_anti_k += a.ipa.anti_key[t] == k;
                                    if ((n_anti_k && tup[q + 2] > 0) || (a.st->ipa_exist_total > 0 && tup[q + 3] > 0)) feas = false;
                                }
                            }
                        }
                }
            }
        }
        // ---- block-local class table (LDS): one global atomic per (block, class) instead of one per node
        bool mine = false;
        int s = -1;
        if (feas) {
            const uint64_t h = cw_hash(tup, kCwTuple);
            s = (int)(h & (kCwBlockSlots - 1));
            int probes = 0;
            for (;;) {
                const unsigned long long old = atomicCAS(&b_key[s], 0ull, (unsigned long long)h);
                if (old == 0ull) {
                    mine = true;
#pragma unroll
                    for (int q = 0; q < kCwTuple; q++) b_tuple[s][q] = tup[q];
                    break;
                }
                if (old == h) break;
                s = (s + 1) & (kCwBlockSlots - 1);
                if (++probes >= kCwBlockSlots) {
                    __hip_atomic_store(giveup, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // more classes in one block than the table holds
                    s = -1;
                    break;
                }
            }
        }
        lslot[j] = s, mine4[j] = mine;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < kCwPerThread; j++)
        if (lslot[j] >= 0 && !mine4[j]) { // same hash: the tuples must be the same tuple
            bool same = true;
#pragma unroll
            for (int q = 0; q < kCwTuple; q++) same = same && b_tuple[lslot[j]][q] == tup4[j][q];
            if (!same) __hip_atomic_store(giveup, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    // how many counted nodes of this block sit at the block's minimum (second sweep over the same entries: L2 hits)
    bool any_unique = false;
    for (int c = 0; c < a.pts.n; c++) any_unique = any_unique || a.plan.h_unique[c];
    if (any_unique) {
        __syncthreads();
#pragma unroll 1
        for (int j = 0; j < kCwPerThread; j++) {
            const int64_t i = (int64_t)blockIdx.x * kCwTile + (int64_t)j * kCwThreads + tid;
            if (i >= a.c.n) continue;
            const uint32_t eb = a.pts.elig[i];
            for (int c = 0; c < a.pts.n; c++)
                if (a.plan.h_unique[c]) {
                    const int32_t v = a.pts.label[c][i];
                    if (v && (eb & 1u) && ((eb >> (1 + c)) & 1u) && (uint32_t)a.pts.tbl[c][v] == b_um[c]) atomicAdd(&b_uc[c], 1u);
                }
        }
    }
    __syncthreads();
    if (tid < kMaxTsc) a.w.umin[(int64_t)blockIdx.x * kMaxTsc + tid] = ((unsigned long long)b_um[tid] << 32) | b_uc[tid];

    // ---- block table -> global table.  Phase a: claim / find (no waiting); phase b: wait for foreign claimants, compare tuples
    int gs[(kCwBlockSlots + kCwThreads - 1) / kCwThreads];
    bool foreign[(kCwBlockSlots + kCwThreads - 1) / kCwThreads];
#pragma unroll
    for (int r = 0; r < (kCwBlockSlots + kCwThreads - 1) / kCwThreads; r++) {
        const int s = r * kCwThreads + tid;
        gs[r] = -1, foreign[r] = false;
        if (s >= kCwBlockSlots || b_key[s] == 0ull) continue;
        const unsigned long long h = b_key[s];
        int g = (int)(h & (kCwSlots - 1)), probes = 0;
        for (;;) {
            // look before claiming: once a class is in the table (after the first few blocks) every later block only READS its slot --
            // a device-scope load, served in parallel -- instead of queueing a compare-and-swap on the same address behind ~1000 others
            unsigned long long old = __hip_atomic_load(&a.w.keys[g], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (old == 0ull) old = atomicCAS(&a.w.keys[g], 0ull, h);
            if (old == 0ull) {
                const uint32_t id = atomicAdd(&a.w.ctl[kCwCtlClasses], 1u);
                if (id >= (uint32_t)kCwMaxClasses) __hip_atomic_store(giveup, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else {
#pragma unroll
                    for (int q = 0; q < kCwTuple; q++) a.w.cls[g].tuple[q] = b_tuple[s][q];
                    a.w.cls[g].id = id;
                    a.w.slot_of_id[id] = g;
                }
                __hip_atomic_store(&a.w.ready[g], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
            if (old == h) {
                foreign[r] = true;
                break;
            }
            g = (g + 1) & (kCwSlots - 1);
            if (++probes >= kCwSlots) {
                __hip_atomic_store(giveup, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                g = -1;
                break;
            }
        }
        gs[r] = g;
        b_gslot[s] = g; // (members, maxima and holders of the class: k_cw_top per block, k_cw_merge over the blocks -- no atomics on the class record)
    }
#pragma unroll
    for (int r = 0; r < (kCwBlockSlots + kCwThreads - 1) / kCwThreads; r++) {
        const int s = r * kCwThreads + tid;
        if (!foreign[r] || gs[r] < 0) continue;
        int spins = 0;
        while (!ld_u32_agent(&a.w.ready[gs[r]]) && ++spins < (1 << 20)) __builtin_amdgcn_s_sleep(1);
        bool same = spins < (1 << 20);
        if (same && !__hip_atomic_load(giveup, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
            for (int q = 0; q < kCwTuple; q++) same = same && a.w.cls[gs[r]].tuple[q] == b_tuple[s][q];
        if (!same) __hip_atomic_store(giveup, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < kCwPerThread; j++) {
        const int64_t i = (int64_t)blockIdx.x * kCwTile + (int64_t)j * kCwThreads + tid;
        if (i < a.c.n_pad) {
            a.w.node_slot[i] = lslot[j] >= 0 ? b_gslot[lslot[j]] : -1;
            a.w.node_A[i] = lA[j];
            a.w.node_A1[i] = lA1[j];
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// k_cw_top: per block of kCwTile nodes, the L best members of every class: round r publishes, per class, the best key not
// yet taken (one LDS atomic-max per live node), its owner retires.  Also counts the holders of the class maxima.
// ------------------------------------------------------------------------------------------------------------------------
struct CwTopArgs {
    DevCols c;
    const DevState *st;
    CwWork w;
    int32_t list_len;
};

__global__ __launch_bounds__(kCwThreads) void k_cw_top(CwTopArgs a) {
    if (a.st->done || a.st->cw_fallback) return;
    if (__hip_atomic_load(a.w.ctl + kCwCtlGiveUp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
    __shared__ unsigned long long cand[2][kCwMaxClasses];
    __shared__ uint32_t l_nf[kCwMaxClasses], l_mt[kCwMaxClasses], l_ma[kCwMaxClasses], l_ht[kCwMaxClasses], l_ha[kCwMaxClasses];
    const int tid = threadIdx.x;
    const int C = (int)a.w.ctl[kCwCtlClasses];
    for (int id = tid; id < C; id += kCwThreads) l_nf[id] = 0, l_mt[id] = 0, l_ma[id] = 0, l_ht[id] = 0, l_ha[id] = 0;
    __syncthreads();
    int id_[kCwPerThread];
    uint64_t key_[kCwPerThread];
    uint32_t cnt_[kCwPerThread], aff_[kCwPerThread];
    // this block's members of every class, their TaintToleration / NodeAffinity maxima ...
#pragma unroll
    for (int j = 0; j < kCwPerThread; j++) {
        const int64_t i = (int64_t)blockIdx.x * kCwTile + (int64_t)j * kCwThreads + tid;
        id_[j] = -1, key_[j] = 0, cnt_[j] = 0, aff_[j] = 0;
        if (i < a.c.n) {
            const int32_t slot = a.w.node_slot[i];
            if (slot >= 0) {
                id_[j] = (int)a.w.cls[slot].id;
                key_[j] = make_key((int64_t)a.w.node_A[i], a.c.global_offset + i);
                const uint32_t w = a.c.stat[i];
                cnt_[j] = (w >> kStatCntShift) & kStatCntMask, aff_[j] = w & kStatAffMask;
                atomicAdd(&l_nf[id_[j]], 1u);
                if (cnt_[j]) atomicMax(&l_mt[id_[j]], cnt_[j]);
                if (aff_[j]) atomicMax(&l_ma[id_[j]], aff_[j]);
            }
        }
    }
    __syncthreads();
    // ... and how many of them hold the block's maxima (k_cw_merge combines the blocks: the larger maximum with its holders)
#pragma unroll
    for (int j = 0; j < kCwPerThread; j++)
        if (id_[j] >= 0) {
            if (cnt_[j] == l_mt[id_[j]]) atomicAdd(&l_ht[id_[j]], 1u);
            if (aff_[j] == l_ma[id_[j]]) atomicAdd(&l_ha[id_[j]], 1u);
        }
    unsigned long long *out = a.w.top + (size_t)blockIdx.x * kCwMaxClasses * a.list_len;
    for (int id = tid; id < C; id += kCwThreads) cand[0][id] = 0ull, cand[1][id] = 0ull;
    __syncthreads();
#pragma unroll 1
    for (int r = 0; r < a.list_len; r++) { // two barriers per round: the other buffer is cleared while this one is read
        const int pb = r & 1;
#pragma unroll
        for (int j = 0; j < kCwPerThread; j++)
            if (id_[j] >= 0) atomicMax(&cand[pb][id_[j]], (unsigned long long)key_[j]);
        __syncthreads();
#pragma unroll
        for (int j = 0; j < kCwPerThread; j++)
            if (id_[j] >= 0 && cand[pb][id_[j]] == key_[j]) id_[j] = -1; // (keys are unique: they carry the node index)
        for (int id = tid; id < C; id += kCwThreads) out[(size_t)id * a.list_len + r] = cand[pb][id], cand[pb ^ 1][id] = 0ull;
        __syncthreads();
    }
    for (int id = tid; id < C; id += kCwThreads) {
        CwPart q;
        q.nf = l_nf[id], q.mt = l_mt[id], q.ht = l_ht[id], q.ma = l_ma[id], q.ha = l_ha[id], q.pad[0] = q.pad[1] = q.pad[2] = 0;
        a.w.part[(size_t)blockIdx.x * kCwMaxClasses + id] = q;
    }
}

// k_cw_merge: workgroup (class id, group g) merges the sorted lists of blocks [g * G, (g + 1) * G) of `src` into the class's L best
// among them: L rounds over the lists' current heads (keys staged in LDS; the rounds are ONE wave, which needs no block barrier).
// One launch when the snapshot's blocks fit one group (dst = the class lists); else two levels: groups -> CwWork::top2 -> lists
// (G^2 blocks: 10 M nodes at L = 64).
constexpr int kCwMergeThreads = 64;
__global__ __launch_bounds__(kCwThreads) void k_cw_merge(CwTopArgs a, const unsigned long long *__restrict__ src, int nb_all, int G, unsigned long long *__restrict__ dst,
                                                          const CwPart *__restrict__ psrc, CwPart *__restrict__ pdst) {
    if (a.st->done || a.st->cw_fallback) return;
    if (__hip_atomic_load(a.w.ctl + kCwCtlGiveUp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
    const int C = (int)a.w.ctl[kCwCtlClasses], id = blockIdx.x, g = blockIdx.y, L = a.list_len;
    if (id >= C) return;
    const int b_first = g * G, nb = (nb_all - b_first < G ? nb_all - b_first : G);
    __shared__ unsigned long long s_k[kCwMaxKeys];
    __shared__ uint8_t s_head[kCwMaxKeys];
    __shared__ CwPart s_part[kCwThreads / 64];
    for (int q = threadIdx.x; q < nb * L; q += kCwThreads) { // staging: all four waves
        const int b = q / L, r = q % L;
        s_k[q] = src[((size_t)(b_first + b) * kCwMaxClasses + id) * L + r];
    }
    for (int b = threadIdx.x; b < nb; b += kCwThreads) s_head[b] = 0;
    {   // the class's members / maxima / holders over these blocks (or groups): pdst = nullptr -> the class record itself
        CwPart acc;
        acc.nf = acc.mt = acc.ht = acc.ma = acc.ha = 0, acc.pad[0] = acc.pad[1] = acc.pad[2] = 0;
        for (int b = threadIdx.x; b < nb; b += kCwThreads) cw_part_add(acc, psrc[(size_t)(b_first + b) * kCwMaxClasses + id]);
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            CwPart o;
            o.nf = (uint32_t)__shfl_xor((int)acc.nf, off), o.mt = (uint32_t)__shfl_xor((int)acc.mt, off), o.ht = (uint32_t)__shfl_xor((int)acc.ht, off);
            o.ma = (uint32_t)__shfl_xor((int)acc.ma, off), o.ha = (uint32_t)__shfl_xor((int)acc.ha, off);
            cw_part_add(acc, o);
        }
        if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = acc;
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int w = 1; w < kCwThreads / 64; w++) cw_part_add(acc, s_part[w]);
            if (pdst) pdst[(size_t)g * kCwMaxClasses + id] = acc;
            else {
                CwClass &k = a.w.cls[a.w.slot_of_id[id]];
                k.nf = acc.nf, k.mt = acc.mt, k.ma = acc.ma, k.ht = acc.ht, k.ha = acc.ha;
            }
        }
    }
    __syncthreads();
    if (threadIdx.x >= kCwMergeThreads) return; // the rounds: one wave, no block barrier
    const int lane = threadIdx.x;
    unsigned long long mine = 0; // rank `lane` of the merged list ends up in lane `lane` (L <= 64)
    if (nb <= 2 * kCwMergeThreads) { // the usual size: a lane keeps the current heads of its (at most two) blocks in registers
        const int b0 = lane, b1 = lane + kCwMergeThreads;
        int h0 = 0, h1 = 0;
        unsigned long long k0 = b0 < nb ? s_k[b0 * L] : 0ull, k1 = b1 < nb ? s_k[b1 * L] : 0ull;
        for (int r = 0; r < L; r++) {
            const unsigned long long K = wave_max_u64(k0 > k1 ? k0 : k1);
            if (lane == r) mine = K;
            if (K == 0ull) break; // (wave-uniform)
            if (k0 == K) h0 += 1, k0 = h0 < L ? s_k[b0 * L + h0] : 0ull;
            else if (k1 == K) h1 += 1, k1 = h1 < L ? s_k[b1 * L + h1] : 0ull;
        }
    } else
        for (int r = 0; r < L; r++) {
            unsigned long long best = 0;
            for (int b = lane; b < nb; b += kCwMergeThreads) {
                const int hd = s_head[b];
                const unsigned long long k = hd < L ? s_k[b * L + hd] : 0ull;
                best = k > best ? k : best;
            }
            const unsigned long long K = wave_max_u64(best);
            if (K)
                for (int b = lane; b < nb; b += kCwMergeThreads) {
                    const int hd = s_head[b];
                    if (hd < L && s_k[b * L + hd] == K) s_head[b] = (uint8_t)(hd + 1);
                }
            if (lane == r) mine = K;
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
        }
    if (lane < L) dst[((size_t)g * kCwMaxClasses + id) * L + lane] = mine;
}

// ------------------------------------------------------------------------------------------------------------------------
// Windows on node-range shards (round 5) for the shape the 64-class kernel takes with every winner
// leaving for good -- ONE hard constraint over a shared key + ONE unique-per-node inter-pod key (zone spread + hostname
// anti-affinity: BASELINE config 5's pod shape).  Per window and rank: the pass over the shard as on one GPU (scan / top / merge: the
// keys carry global indices), k_cw_xpack (the rank's classes -- tuple, statistics, the staged entries of their lists -- into a
// fixed-size record), ONE all-gather of the records (73 KB per rank; the caller's collective or ncclAllGather in ccsim_dist_run),
// k_cw_xunify (classes with equal tuples are one class: first occurrence in rank order names it; their lists merge by key; their
// statistics combine like the blocks' -- the same bytes in, the same result out on every rank), then the deciding wave REPLICATED on
// every rank on the merged lists (k_cw_decide_fast<.., SH>): identical placements everywhere.  Owners apply a placement to their
// columns and to the unique key's table entries (a node's own entries are only ever read by its owner's scan); every rank adds it to
// the shared key's domain counts (the domain id rides in the record) and to the replicated run state.  Whatever a rank cannot
// represent travels as a flag in its record: every rank falls back to the one-pass-per-placement protocol alike.
// ------------------------------------------------------------------------------------------------------------------------
constexpr int kCwXClasses = 64;
struct __attribute__((aligned(16))) CwXHdr {
    uint32_t C, giveup, L, pad;
};
struct __attribute__((aligned(16))) CwXClass {
    int32_t tuple[kCwTuple];
    uint32_t nf, mt, ma, ht, ha, pad;
    unsigned long long hash;
    uint4 ent[kCwMaxList];
};
constexpr size_t kCwXBytes = sizeof(CwXHdr) + (size_t)kCwXClasses * sizeof(CwXClass);
constexpr int kCwRecZoneShift = 53; // a (node, clones) record also carries the shared key's domain id of a clone that counts: idx : 40 | clones : 13 | domain : 7
constexpr uint64_t kCwRecClonesMask = (1ull << (kCwRecZoneShift - kIdxBits)) - 1;

struct CwXArgs {
    DevCols c;
    const DevState *st;
    DevPts pts;
    CwWork w;
    int32_t list_len;
};

// one workgroup per class of this rank (grid kCwXClasses x 64 threads): thread m stages list member m
__global__ __launch_bounds__(64) void k_cw_xpack(CwXArgs a) {
    CwXHdr *hdr = reinterpret_cast<CwXHdr *>(a.w.xsend);
    CwXClass *out = reinterpret_cast<CwXClass *>(a.w.xsend + sizeof(CwXHdr));
    const int c = blockIdx.x, m = threadIdx.x, L = a.list_len;
    const bool idle = a.st->done || a.st->cw_fallback;
    const uint32_t giveup = idle ? 1u : __hip_atomic_load(a.w.ctl + kCwCtlGiveUp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int C = idle ? 0 : (int)a.w.ctl[kCwCtlClasses];
    if (c == 0 && m == 0) hdr->C = (uint32_t)(C <= kCwXClasses ? C : 0), hdr->giveup = giveup || C > kCwXClasses ? 1u : 0u, hdr->L = (uint32_t)L, hdr->pad = 0;
    if (giveup || C > kCwXClasses || c >= C) return;
    const CwClass &k = a.w.cls[a.w.slot_of_id[c]];
    CwXClass &o = out[c];
    if (m < kCwTuple) o.tuple[m] = k.tuple[m];
    if (m == 0) o.nf = k.nf, o.mt = k.mt, o.ma = k.ma, o.ht = k.ht, o.ha = k.ha, o.pad = 0, o.hash = cw_hash(k.tuple, kCwTuple);
    uint4 r = make_uint4(0u, 0u, 0u, 0u);
    if (m < L) {
        const unsigned long long key = a.w.lists[c * L + m];
        if (key) {
            const int64_t i = key_index(key) - a.c.global_offset;
            r.x = (uint32_t)key, r.y = (uint32_t)(key >> 32), r.z = (uint32_t)a.w.node_A1[i];
            r.w = (a.c.stat[i] & (kStatAffMask | (kStatCntMask << kStatCntShift))) | (((a.pts.n ? (uint32_t)a.pts.elig[i] : 0u) & kStatImgMask) << kStatImgShift);
        }
    }
    o.ent[m] = r; // (kCwMaxList == 64 == the workgroup)
}

// one workgroup per class of the CLUSTER (grid kCwXClasses x 256 threads); every workgroup names the classes itself (the same few
// kilobytes, the same answer), then merges its own
__global__ __launch_bounds__(kCwThreads) void k_cw_xunify(CwXArgs a) {
    if (a.st->done || a.st->cw_fallback) return;
    const int R = a.w.x_ranks, tid = threadIdx.x, g = blockIdx.x;
    __shared__ unsigned long long s_hash[8 * kCwXClasses];
    __shared__ int16_t s_lead[8 * kCwXClasses], s_gid[8 * kCwXClasses];
    __shared__ int s_G, s_bad;
    auto hdr_of = [&](int r) { return reinterpret_cast<const CwXHdr *>(a.w.xrecv + (size_t)r * kCwXBytes); };
    auto cls_of = [&](int r, int c) { return reinterpret_cast<const CwXClass *>(a.w.xrecv + (size_t)r * kCwXBytes + sizeof(CwXHdr)) + c; };
    if (tid == 0) {
        int bad = R > 8 ? 1 : 0;
        for (int r = 0; r < R && r < 8; r++) bad |= hdr_of(r)->giveup != 0;
        s_bad = bad, s_G = 0;
    }
    for (int p = tid; p < 8 * kCwXClasses; p += kCwThreads) {
        const int r = p / kCwXClasses, c = p % kCwXClasses;
        s_hash[p] = r < R && c < (int)hdr_of(r)->C ? cls_of(r, c)->hash : 0ull; // (cw_hash never returns 0)
    }
    __syncthreads();
    if (s_bad) {
        if (g == 0 && tid == 0) a.w.xhdr[0] = 0, a.w.xhdr[1] = 1;
        return;
    }
    // the first pair (rank, class) in rank order with the same tuple names the class
    for (int p = tid; p < 8 * kCwXClasses; p += kCwThreads) {
        int lead = -1;
        const unsigned long long h = s_hash[p];
        if (h) {
            lead = p;
            for (int q = 0; q < p; q++)
                if (s_hash[q] == h) {
                    const CwXClass *x = cls_of(q / kCwXClasses, q % kCwXClasses), *y = cls_of(p / kCwXClasses, p % kCwXClasses);
                    bool same = true;
                    for (int t = 0; t < kCwTuple; t++) same = same && x->tuple[t] == y->tuple[t];
                    if (same) {
                        lead = q;
                        break;
                    }
                    atomicOr(&s_bad, 1); // two tuples, one hash: the windowed mode's give-up condition everywhere else too
                }
        }
        s_lead[p] = (int16_t)lead;
    }
    __syncthreads();
    for (int p = tid; p < 8 * kCwXClasses; p += kCwThreads) {
        int gid = -1;
        if (s_lead[p] == p) {
            gid = 0;
            for (int q = 0; q < p; q++) gid += s_lead[q] == q ? 1 : 0;
            atomicMax(&s_G, gid + 1);
        }
        s_gid[p] = (int16_t)gid;
    }
    __syncthreads();
    const int G = s_G;
    if (s_bad || G > kCwXClasses) {
        if (g == 0 && tid == 0) a.w.xhdr[0] = 0, a.w.xhdr[1] = 1;
        return;
    }
    if (g == 0 && tid == 0) a.w.xhdr[0] = (uint32_t)G, a.w.xhdr[1] = 0, a.w.xhdr[2] = 0, a.w.xhdr[3] = 0;
    if (g >= G || tid >= 64) return;
    // this class: at most one list per rank (a rank's classes have distinct tuples); lane r merges rank r's
    const int lane = tid;
    const CwXClass *mine = nullptr;
    if (lane < R)
        for (int c = 0; c < (int)hdr_of(lane)->C; c++) {
            const int p = lane * kCwXClasses + c;
            if (s_gid[s_lead[p]] == g) mine = cls_of(lane, c);
        }
    CwPart acc;
    acc.nf = mine ? mine->nf : 0, acc.mt = mine ? mine->mt : 0, acc.ht = mine ? mine->ht : 0, acc.ma = mine ? mine->ma : 0, acc.ha = mine ? mine->ha : 0;
    acc.pad[0] = acc.pad[1] = acc.pad[2] = 0;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        CwPart o;
        o.nf = (uint32_t)__shfl_xor((int)acc.nf, off), o.mt = (uint32_t)__shfl_xor((int)acc.mt, off), o.ht = (uint32_t)__shfl_xor((int)acc.ht, off);
        o.ma = (uint32_t)__shfl_xor((int)acc.ma, off), o.ha = (uint32_t)__shfl_xor((int)acc.ha, off);
        cw_part_add(acc, o);
    }
    int head = 0;
    uint4 cur = mine ? mine->ent[0] : make_uint4(0u, 0u, 0u, 0u);
    for (int m = 0; m < kCwMaxList; m++) {
        const uint64_t mykey = ((uint64_t)cur.y << 32) | cur.x;
        const uint64_t K = wave_max_u64(mykey);
        if (K != 0ull && mykey == K) { // (keys are unique: they carry the node index)
            a.w.xent[(size_t)g * kCwMaxList + m] = cur;
            head += 1;
            cur = head < kCwMaxList ? mine->ent[head] : make_uint4(0u, 0u, 0u, 0u);
        }
        if (K == 0ull && lane == 0) a.w.xent[(size_t)g * kCwMaxList + m] = make_uint4(0u, 0u, 0u, 0u);
    }
    const unsigned long long has = __ballot(mine != nullptr);
    if (lane == __ffsll(has) - 1) { // the class's first rank: its tuple is the tuple
        CwClass &k = a.w.xcls[g];
        for (int t = 0; t < kCwTuple; t++) k.tuple[t] = mine->tuple[t];
        k.nf = acc.nf, k.mt = acc.mt, k.ma = acc.ma, k.ht = acc.ht, k.ha = acc.ha, k.id = (uint32_t)g, k.pad[0] = k.pad[1] = 0;
    }
}

// behind the deciding wave of a sharded window: nobody took it (a shape or a size the 64-class form declines) -> every rank, alike,
// continues with one pass per placement; the rank's class table is left empty either way
__global__ __launch_bounds__(kCwThreads) void k_cw_xfallback(CwXArgs a, DevState *st) {
    if (st->done || st->cw_fallback) return;
    const bool took = __hip_atomic_load(a.w.ctl + kCwCtlFastDone, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
    if (took) {
        if (threadIdx.x == 0) __hip_atomic_store(a.w.ctl + kCwCtlFastDone, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    const int C = (int)a.w.ctl[kCwCtlClasses];
    for (int id = threadIdx.x; id < C && id < kCwMaxClasses; id += kCwThreads) {
        const int gs = a.w.slot_of_id[id];
        a.w.keys[gs] = 0ull, a.w.ready[gs] = 0u;
        CwClass &k = a.w.cls[gs];
        k.nf = k.mt = k.ma = k.ht = k.ha = 0u;
    }
    __syncthreads();
    if (threadIdx.x == 0) a.w.ctl[kCwCtlClasses] = 0u, a.w.ctl[kCwCtlGiveUp] = 0u, st->cw_fallback = 1;
}

// ------------------------------------------------------------------------------------------------------------------------
// k_cw_decide: the window's cycles.
// ------------------------------------------------------------------------------------------------------------------------
struct CwDecideArgs {
    DevCols c;
    DevPod p;
    DevState *st;
    DevPts pts;
    DevSoft soft;
    DevIpa ipa;
    CwPlan plan;
    CwWork w;
    int32_t *log;
};

constexpr int kCwCand = kCwMaxClasses + kCwMaxWindow;
constexpr int kCwListLds = 2048; // class-list entries (classes x L) whose node facts the decide kernel stages in LDS

struct CwLds {
    int32_t i32[kCwLdsI32];
    long long i64[kCwLdsI64];
    int32_t c_tuple[kCwMaxClasses][kCwTuple];
    uint32_t c_nf[kCwMaxClasses], c_mt[kCwMaxClasses], c_ma[kCwMaxClasses], c_ht[kCwMaxClasses], c_ha[kCwMaxClasses];
    int32_t c_head[kCwMaxClasses];
    unsigned long long c_key[kCwMaxClasses]; // the head's (A, index) key, 0 = list exhausted
    // node facts of every list entry (staged by all threads before the cycles start: a winner costs no trip to HBM)
    unsigned long long li_key[kCwListLds];
    uint32_t li_stat[kCwListLds], li_elig[kCwListLds];
    int32_t li_A1[kCwListLds];
    // nodes that received a clone in this window.  `alive` lists the ones that may still win (a node that is full, or whose own
    // clone blocks it through a required anti-affinity term, never comes back: the counts only grow)
    int32_t t_tuple[kCwMaxWindow][kCwTuple];
    long long t_gidx[kCwMaxWindow];
    int32_t t_A[kCwMaxWindow];
    uint32_t t_cnt[kCwMaxWindow], t_aff[kCwMaxWindow], t_took[kCwMaxWindow], t_elig[kCwMaxWindow];
    int32_t alive[kCwMaxWindow];
    long long e_rp[kCwCand], e_ri[kCwCand]; // per candidate: raw PodTopologySpread / InterPodAffinity score
    uint32_t e_fl[kCwCand];                 // bit0 feasible, bit1 has all soft keys
    // per-constraint scalars of the cycle loop (runtime-indexed: registers would become a scratch frame)
    int32_t mn[kMaxTsc], u_min[kMaxTsc];
    uint32_t u_cnt[kMaxTsc];
    double soft_w[kMaxTsc];
    long long soft_size[kMaxTsc];
    int32_t s_nt;
    uint32_t fast_done;       // k_cw_decide: the workgroup's ONE read of the fast kernel's "window done" flag
    unsigned long long pf[8]; // k_cw_decide_fast: phase ticks of a measurement run
};

// LDS traffic inside one wave needs no barrier, only the data back: wait for the LDS / scalar counters, not for HBM stores
__device__ __forceinline__ void cw_lds_sync() {
    __builtin_amdgcn_s_waitcnt(0xc07f); // lgkmcnt(0), vmcnt / expcnt untouched
    __builtin_amdgcn_wave_barrier();
}

// The plugin parameters the cycle loop reads, as a register image (every value is wave-uniform: SGPRs).  Read through the
// argument pointer they were ~25 dependent scalar loads per candidate pass: 2 us of the first version's 4.6 us per cycle
// (profiles/r03/bench_coupled.txt).  The bounds are compile-time so that every index is: <2,2,2> is the common pod (config 5's
// shape has one hard constraint and one key), <4,4,4> everything else the plan admits.
template <int MH, int MS, int MK>
struct CwP {
    int nh, ns, nk;
    int h_comp[MH], h_unique[MH], h_off[MH], h_len[MH], h_pres[MH], h_skew[MH], h_self[MH], h_usemin[MH];
    int s_comp[MS], s_off[MS], s_len[MS], s_bm[MS], s_host[MS], s_self[MS], s_skew[MS], s_nocredit[MS];
    int k_comp[MK], k_unique[MK], k_off[MK], k_len[MK], k_aff[MK], k_anti[MK], k_daff[MK], k_danti[MK], k_dent[MK];
    long long k_dscore[MK];
    int soft_w, ipa_w, ipa_filter, ipa_any_term, self_aff;
    __device__ __forceinline__ void load(const CwDecideArgs &a) {
        nh = a.pts.n, ns = a.soft.n, nk = a.ipa.on ? a.ipa.n_keys : 0;
#pragma unroll
        for (int c = 0; c < MH; c++) {
            const bool on = c < nh;
            h_comp[c] = on ? a.plan.h_comp[c] : 0, h_unique[c] = on ? a.plan.h_unique[c] : 0, h_off[c] = on ? a.plan.h_off[c] : 0;
            h_len[c] = on ? a.plan.h_len[c] : 0, h_pres[c] = on ? a.plan.h_pres[c] : 0;
            h_skew[c] = on ? a.pts.max_skew[c] : 0, h_self[c] = on ? a.pts.self_match[c] : 0;
            h_usemin[c] = on && !(a.pts.n_present[c] < a.pts.min_domains[c]) ? 1 : 0; // else the global minimum counts as 0 (filtering.go:56-69)
        }
#pragma unroll
        for (int c = 0; c < MS; c++) {
            const bool on = c < ns;
            s_comp[c] = on ? a.plan.s_comp[c] : 0, s_off[c] = on ? a.plan.s_off[c] : 0, s_len[c] = on ? a.plan.s_len[c] : 0, s_bm[c] = on ? a.plan.s_bm[c] : 0;
            s_host[c] = on ? a.soft.is_hostname[c] : 0, s_self[c] = on ? a.soft.self_match[c] : 0, s_skew[c] = on ? a.soft.max_skew[c] : 0;
            s_nocredit[c] = on ? a.soft.nocredit[c] : 0;
        }
#pragma unroll
        for (int k = 0; k < MK; k++) {
            const bool on = k < nk;
            k_comp[k] = on ? a.plan.k_comp[k] : 0, k_unique[k] = on ? a.plan.k_unique[k] : 0, k_off[k] = on ? a.plan.k_off[k] : 0, k_len[k] = on ? a.plan.k_len[k] : 0;
            k_aff[k] = on ? a.ipa.aff_terms_on_key[k] : 0; // required affinity terms over this key (they all read the same pair)
            int anti = 0;
            for (int q = 0; on && q < a.ipa.n_anti; q++) anti += a.ipa.anti_key[q] == k;
            k_anti[k] = anti;
            k_daff[k] = on && a.ipa.self_aff ? a.ipa.aff_terms_on_key[k] : 0, k_danti[k] = on ? a.ipa.anti_self_on_key[k] : 0;
            k_dent[k] = on ? a.ipa.self_entries[k] : 0, k_dscore[k] = on ? a.ipa.score_self[k] : 0;
        }
        soft_w = a.soft.n > 0 ? a.soft.w : 0, ipa_w = a.ipa.on ? a.ipa.w : 0, ipa_filter = a.ipa.on && a.ipa.filter_on;
        ipa_any_term = a.ipa.on && (a.ipa.n_aff || a.ipa.n_anti), self_aff = a.ipa.on ? a.ipa.self_aff : 0;
    }
};

// one candidate's coupled verdict against the tables (minima of the hard constraints in L.mn).
// `*dead`: the failure is permanent (required anti-affinity against pods that are there to stay: the counts only grow).
template <int MH, int MS, int MK>
__device__ __forceinline__ bool cw_coupled_ok(const CwP<MH, MS, MK> &P, const CwLds &L, const int32_t *t, int64_t aff_total, int64_t exist_total,
                                              bool *dead) {
    bool ok = true;
#pragma unroll
    for (int c = 0; c < MH; c++)
        if (c < P.nh) { // PodTopologySpread.Filter (filtering.go:311-356)
            const int32_t tv = t[P.h_comp[c]];
            int32_t v, m;
            if (P.h_unique[c]) v = tv & 1, m = tv >> 1;
            else v = tv, m = v ? L.i32[P.h_off[c] + v] : 0;
            if (!v) ok = false;
            const int64_t minm = P.h_usemin[c] ? (int64_t)L.mn[c] : 0;
            if ((int64_t)m + P.h_self[c] - minm > (int64_t)P.h_skew[c]) ok = false;
        }
    if (P.ipa_filter && !(exist_total == 0 && !P.ipa_any_term)) { // filtering.go:410-432
        bool pods_exist = true, aff_ok = true, any_aff = false;
#pragma unroll
        for (int k = 0; k < MK; k++)
            if (k < P.nk) {
                const int c0 = P.k_comp[k];
                const int32_t v = t[c0];
                if (P.k_aff[k]) { // satisfyPodAffinity :382-408
                    any_aff = true;
                    if (!v) aff_ok = false;
                    const int64_t cntv = !v ? 0 : (P.k_unique[k] ? (int64_t)t[c0 + 1] : L.i64[P.k_off[k] + v]);
                    if (cntv <= 0) pods_exist = false;
                }
                if (P.k_anti[k]) { // satisfyPodAntiAffinity :367-379
                    const int64_t cntv = !v ? 0 : (P.k_unique[k] ? (int64_t)t[c0 + 2] : L.i64[P.k_off[k] + P.k_len[k] + v]);
                    if (cntv > 0) ok = false, *dead = true;
                }
                if (exist_total > 0) { // satisfyExistingPodsAntiAffinity :352-364
                    const int64_t cntv = !v ? 0 : (P.k_unique[k] ? (int64_t)t[c0 + 3] : L.i64[P.k_off[k] + 2 * P.k_len[k] + v]);
                    if (cntv > 0) ok = false, *dead = true;
                }
            }
        if (any_aff && (!aff_ok || (!pods_exist && !(aff_total == 0 && P.self_aff)))) ok = false;
    }
    return ok;
}

template <int MH, int MS, int MK>
__device__ __forceinline__ bool cw_soft_keys(const CwP<MH, MS, MK> &P, const int32_t *t) {
    bool all = true;
#pragma unroll
    for (int c = 0; c < MS; c++)
        if (c < P.ns) {
            const int32_t tv = t[P.s_comp[c]];
            all = all && (P.s_host[c] ? (tv & 1) : tv) != 0;
        }
    return all;
}

// the local verdict and score of node i after `k` clones were added to what its columns hold (the window applies its
// placements to the columns when it ends)
__device__ __forceinline__ int32_t cw_local_after(const CwDecideArgs &a, int64_t i, int64_t k, uint32_t mt_a, uint32_t ma_a) {
    const int64_t a_cpu = a.c.alloc[0][i], a_mem = a.c.alloc[1][i];
    const int64_t r0 = a.c.req[0][i] + k * a.p.req[0], r1 = a.c.req[1][i] + k * a.p.req[1];
    const int64_t z0 = a.c.nz_mcpu[i] + k * a.p.nz_mcpu, z1 = a.c.nz_mem[i] + k * a.p.nz_mem;
    const int32_t pc = a.c.pod_count[i] + (int32_t)k, a_pods = a.c.alloc_pods[i];
    const uint32_t w = a.c.stat[i];
    int64_t xa[kMaxExtra], xr[kMaxExtra];
    bool ok = fits_core(a.p, a_cpu, a_mem, r0, r1, a_pods, pc);
#pragma unroll
    for (int x = 0; x < kMaxExtra; x++) {
        xa[x] = xr[x] = 0;
        if (x < a.p.nx) {
            const int col = a.p.xcol[x];
            xa[x] = a.c.alloc[col][i], xr[x] = a.c.req[col][i] + k * a.p.req[col];
            const int64_t rq = a.p.req[col];
            if (a.p.fit_enabled && !a.p.all_zero_req && rq > 0 && rq > xa[x] - xr[x]) ok = false;
        }
    }
    if (!ok) return -1;
    const uint32_t cnt = (w >> kStatCntShift) & kStatCntMask, aff = w & kStatAffMask, img = (w >> kStatImgShift) & kStatImgMask;
    return (int32_t)(static_score(a.p, cnt, aff, img, mt_a, ma_a) +
                     (a.p.gen_score ? dynamic_score_gen<kMaxExtra>(a.p, a_cpu, a_mem, r0, r1, z0, z1, xa, xr)
                                    : dynamic_score(a.p, make_rcp(a_cpu, a_mem), a_cpu, a_mem, r0, r1, z0, z1)));
}

// ------------------------------------------------------------------------------------------------------------------------
// k_cw_decide_fast: the same cycles with LANE = CANDIDATE.  A single wave issues one instruction every ~5 clocks, so a cycle
// costs what its instruction count says (the general kernel below executes ~2000 per cycle: 4.3 us; profiles/r03).  Here every
// class -- and every touched node that may still win -- lives in the registers of one lane: its tuple components, its own
// copy of the domain counts it reads, members, maxima, list head.  A placement is a handful of uniform broadcasts
// (v_readlane from the winner's lane) and compare-and-add on all lanes; the minimum of a hard constraint is a DPP reduction
// over the lanes that stand for the domains; nothing in a cycle touches HBM unless a touched node wins again.
// Takes the window when: <= 2 hard constraints (shared keys with <= 63 domains, or unique keys), no ScheduleAnyway scoring,
// <= 2 inter-pod keys with PreScore skipping (no weights to sum), <= kCwFastClasses classes, entries within int32.  Otherwise
// it leaves everything untouched and the general kernel, launched right behind it, does the window.
// ------------------------------------------------------------------------------------------------------------------------
constexpr int kCwFastClasses = 48; // (the other lanes hold touched nodes that may still win)

// a value every lane holds identically but the compiler cannot know it (it came through a vector load): say so, or every
// quantity derived from it -- loop counters, branch conditions -- is handled as divergent, with exec-mask bookkeeping
__device__ __forceinline__ int64_t cw_uni64(int64_t v) {
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)(uint64_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)((uint64_t)v >> 32));
    return (int64_t)(((uint64_t)hi << 32) | lo);
}
__device__ __forceinline__ int32_t rl32(int32_t v, int src) { return __builtin_amdgcn_readlane(v, src); }
__device__ __forceinline__ uint64_t rl64(uint64_t v, int src) {
    return ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), src) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, src);
}
// v_writelane_b32: lane `lane` of `old` takes the uniform `value` (this clang has no builtin for it; the label names the LLVM intrinsic)
extern "C" __device__ int cw_writelane(int value, int lane, int old) __asm("llvm.amdgcn.writelane");
__device__ __forceinline__ int32_t wave_min_i32_nonneg(int32_t v) { return (int32_t)(0x7fffffffu - wave_max_u32(0x7fffffffu - (uint32_t)v)); }

// The shape of the pod is a template argument: NH hard constraints, bit c of HU = constraint c is over a unique-per-node key;
// NK inter-pod keys, bit k of KU = key k is unique per node.  A runtime flag costs an SGPR and a branch in every cycle, and
// the first (runtime-flag) form of this kernel spent a third of its instructions moving spilled SGPRs through VGPR lanes.
// PROF: phase ticks (s_memrealtime + an LDS add per phase) compiled in only for measurement runs (CCSIM_CW_PROF=1).
//
// What a cycle does NOT do (the second form of this loop; the first one is in profiles/r03/cw_kernel_stats.csv at 1.45 us):
//  * no LDS round trip for the winner's facts: every class lane holds its list head's record (key, A after one more clone,
//    count / sum / eligibility bits) in registers, the commit reads it with v_readlane, and the lane fetches its next head
//    with a load that nothing waits for before the next cycle's argmax;
//  * no per-cycle store: the placement log and the (node, clones) records of the epilogue are kept one per lane (lane = cycle
//    mod 64) and written 64 at a time;
//  * no 64-bit run counters: a window is <= 2048 cycles, so the loop counts in 32 bits against bounds computed once (limit,
//    log capacity), and the inter-pod totals are (was zero, is positive) flags plus 32-bit deltas added to the state at the end;
//  * the argmax is a 32-bit DPP maximum of the keys' high words (the score); the low words (lowest index first) are only
//    reduced when several lanes share the score.
constexpr int kCwFastListLds = 4096; // class-list entries the lane-per-candidate kernel stages (64 classes x 64 members: 64 KiB)
struct CwFastLds {
    uint4 ent[kCwFastListLds + 64];             // class c, member m at c * (L + 1) + m: {key lo, key hi, A after one more clone, meta}; a zero key ends the list
    unsigned long long rec[kCwFastWindow + 64]; // nodes that received clones in this window: index | clones << kIdxBits
    int32_t s_nt;
    unsigned long long pf[8];
    // sweeps (a round of placements at once, see `sweep` in the kernel): per value id of the shared key -- candidates of the domain /
    // "the domain took a clone in this sweep" (both all-zero outside a sweep); the participants' facts by position; the log's current group
    int32_t dflag[72];
    unsigned long long dbest[72], skey[64];
    uint32_t byrank[64];
    int32_t lg[64];
};
static_assert(sizeof(CwFastLds) <= sizeof(CwLds), "both decide kernels are launched with sizeof(CwLds) of LDS");
constexpr uint32_t kCwMetaStat = kStatAffMask | (kStatCntMask << kStatCntShift);
// stat's count (TaintToleration) and sum (NodeAffinity) fields where they are; the node's eligibility bits in the image field
__device__ __forceinline__ uint32_t cw_meta(uint32_t stat, uint32_t elig) { return (stat & kCwMetaStat) | ((elig & kStatImgMask) << kStatImgShift); }

// FULL (round 4): the form for MORE classes / domains than the standard one takes -- up to 64 classes and 64 domains of a shared
// key (the synthetic 1M-node cluster has 64 zones: one class per zone).  Every lane is a class lane, so nothing is left for touched
// nodes and for the scratch lane the standard form writes a departing winner's record to: a winner that is gone for good (full, or
// blocked by its own clone through a required anti-affinity term -- every winner of BASELINE config 5's pod shape) only leaves its
// (node, clones) record; one that could still win ends the window, and the next pass sees it in its new state.  Domains sit in
// lane value - 1.  A separate instantiation, launched behind the standard one: it takes the windows that one declined.
// SH (round 5): the window of a node-range SHARD -- classes and staged list entries come merged over the ranks from k_cw_xunify, the wave
// runs replicated on every rank, and the epilogue applies a placement to the columns and the unique key's entries only where the node
// is this rank's (the shared key's domain count and the run state on every rank).
template <int NH, int HU, int NK, int KU, bool PROF, bool FULL = false, bool SH = false>
__global__ __launch_bounds__(kCwThreads) void k_cw_decide_fast(const CwDecideArgs *__restrict__ ap) {
    const CwDecideArgs &a = *ap;
    extern __shared__ __attribute__((aligned(16))) unsigned char cw_lds_raw[];
    CwFastLds &L = *reinterpret_cast<CwFastLds *>(cw_lds_raw);
    DevState &S = *a.st;
    if (S.done || S.cw_fallback) return;
    if (!SH && __hip_atomic_load(a.w.ctl + kCwCtlGiveUp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return; // (the general kernel records the fallback)
    if (SH && a.w.xhdr[1]) return; // (some rank gave up: k_cw_xfallback records it on every rank)
    if (FULL && !SH && __hip_atomic_load(a.w.ctl + kCwCtlFastDone, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return; // (the standard form took this window)
    const int tid = threadIdx.x, lane = tid & 63;
    const int dlane = FULL ? lane + 1 : lane; // the value id of the shared-key domain this lane stands for
    const int LL = uni32(SH ? kCwMaxList : a.plan.list_len), W = uni32(a.plan.window < kCwFastWindow ? a.plan.window : kCwFastWindow);
    const int C = uni32(SH ? (int)a.w.xhdr[0] : (int)a.w.ctl[kCwCtlClasses]);
    const int64_t own_lo = a.c.global_offset, own_hi = a.c.global_offset + a.c.n; // (SH: the nodes whose columns are this rank's)
    // members of a class list this kernel uses: all L, or as many as the staging area holds for C classes (many classes with long
    // lists: the window then ends where a class has used up its shorter list -- earlier, never differently)
    const int LU = uni32(C > 0 && C * LL > kCwFastListLds ? kCwFastListLds / C : LL), LS = LU + 1;
    constexpr bool HU0 = (HU & 1) != 0, HU1 = (HU & 2) != 0, KU0 = (KU & 1) != 0, KU1 = (KU & 2) != 0;
    // ---- does this window qualify?  (uniform; nothing has been modified yet)
    bool fits = C <= (FULL ? 64 : kCwFastClasses) && LU >= 1;
    if (NH > 0) fits = fits && a.pts.max_skew[0] <= (1 << 29);
    if (NH > 1) fits = fits && a.pts.max_skew[1] <= (1 << 29);
    if (NH > 0 && !HU0) fits = fits && a.plan.h_len[0] <= (FULL ? 65 : 64);
    if (NH > 1 && !HU1) fits = fits && a.plan.h_len[1] <= (FULL ? 65 : 64);
    if (NK > 0 && a.ipa.w) { // PreScore must skip for the whole window: no entries now, none added by a clone (scoring.go:199-201)
        fits = fits && S.ipa_entries == 0 && a.ipa.self_entries[0] == 0;
        if (NK > 1) fits = fits && a.ipa.self_entries[1] == 0;
    }
    for (int k = 0; k < NK; k++) // (what a clone adds per key, summed in 32 bits over <= 2048 cycles)
        fits = fits && a.ipa.aff_terms_on_key[k] < (1 << 16) && a.ipa.anti_self_on_key[k] < (1 << 16) && a.ipa.self_entries[k] < (1 << 16);
    if (!uni32(fits)) return;
    unsigned long long t_prev = 0ull;
    if (PROF) t_prev = __builtin_amdgcn_s_memrealtime();
#define CW_TICK(i) do { if (PROF) { const unsigned long long t_now = __builtin_amdgcn_s_memrealtime(); if (lane == 0) L.pf[i] += t_now - t_prev; t_prev = t_now; } } while (0)

    // ---- prologue (all threads): the node facts of every list entry -> LDS
    for (int q = tid; q < C * LS; q += kCwThreads) {
        const int c = q / LS, m = q - c * LS;
        uint4 r = make_uint4(0u, 0u, 0u, 0u);
        if (SH) { // (staged by the owners' k_cw_xpack, merged by k_cw_xunify)
            if (m < LU) r = a.w.xent[(size_t)c * kCwMaxList + m];
        } else if (m < LU) {
            const unsigned long long key = a.w.lists[c * LL + m];
            if (key) {
                const int64_t i = key_index(key) - a.c.global_offset;
                r.x = (uint32_t)key, r.y = (uint32_t)(key >> 32), r.z = (uint32_t)a.w.node_A1[i];
                r.w = cw_meta(a.c.stat[i], a.pts.n ? (uint32_t)a.pts.elig[i] : 0u);
            }
        }
        L.ent[q] = r;
    }
    if (tid == 0) L.s_nt = -1; // -1: the window was not taken
    if (tid < 8) L.pf[tid] = 0;
    if (tid < 72) L.dbest[tid] = 0ull, L.dflag[tid] = 0;
    __syncthreads();

    if (tid < 64) {
        // ---- uniform parameters (SGPRs; what the shape does not use is never loaded)
        const int h_self0 = uni32(NH > 0 ? a.pts.self_match[0] : 0), h_self1 = uni32(NH > 1 ? a.pts.self_match[1] : 0);
        const int h_skew0 = uni32(NH > 0 ? a.pts.max_skew[0] : 0), h_skew1 = uni32(NH > 1 ? a.pts.max_skew[1] : 0);
        const bool h_usemin0 = uni32(NH > 0 && !(a.pts.n_present[0] < a.pts.min_domains[0])) != 0, h_usemin1 = uni32(NH > 1 && !(a.pts.n_present[1] < a.pts.min_domains[1])) != 0;
        const int k_aff0 = uni32(NK > 0 ? a.ipa.aff_terms_on_key[0] : 0), k_aff1 = uni32(NK > 1 ? a.ipa.aff_terms_on_key[1] : 0);
        int k_anti0 = 0, k_anti1 = 0;
        for (int q = 0; NK > 0 && q < uni32(a.ipa.n_anti); q++) k_anti0 += a.ipa.anti_key[q] == 0, k_anti1 += a.ipa.anti_key[q] == 1;
        k_anti0 = uni32(k_anti0), k_anti1 = uni32(k_anti1);
        const int k_daff0 = uni32(NK > 0 && a.ipa.self_aff ? a.ipa.aff_terms_on_key[0] : 0), k_daff1 = uni32(NK > 1 && a.ipa.self_aff ? a.ipa.aff_terms_on_key[1] : 0);
        const int k_danti0 = uni32(NK > 0 ? a.ipa.anti_self_on_key[0] : 0), k_danti1 = uni32(NK > 1 ? a.ipa.anti_self_on_key[1] : 0);
        const int k_dent0 = uni32(NK > 0 ? a.ipa.self_entries[0] : 0), k_dent1 = uni32(NK > 1 ? a.ipa.self_entries[1] : 0);
        const bool ipa_filter = uni32(NK > 0 && a.ipa.filter_on) != 0, ipa_any_term = uni32(NK > 0 && (a.ipa.n_aff || a.ipa.n_anti)) != 0, self_aff = uni32(NK > 0 && a.ipa.self_aff) != 0;
        const bool any_aff = k_aff0 || k_aff1;
        const uint32_t mt_a = (uint32_t)uni32(S.mt_a), ma_a = (uint32_t)uni32(S.ma_a);
        const bool track = uni32(a.p.w_taint != 0 || a.p.w_aff != 0) != 0; // else every count / sum is 0 and the maxima cannot move

        // ---- lane = candidate: the class records into registers
        int32_t hv0 = 0, hv1 = 0, hc0 = 0, hc1 = 0, kv0 = 0, kv1 = 0, kf0 = 0, kf1 = 0, kn0 = 0, kn1 = 0, ke0 = 0, ke1 = 0;
        uint32_t nfm = 0, cmt = 0, cma = 0, cht = 0, cha = 0;
        int32_t head = 0, hA1 = 0; // class lanes: position in the list; the head's A after one more clone
        uint32_t hmeta = 0, tk = 0; // the head's (class lanes) / the node's own (touched lanes) count, sum and eligibility bits; clones this window
        uint64_t key = 0;
        bool over = false;
        const bool cls = lane < C;
        if (cls) {
            const CwClass &k = SH ? a.w.xcls[lane] : a.w.cls[a.w.slot_of_id[lane]];
            nfm = k.nf, cmt = k.mt, cma = k.ma, cht = k.ht, cha = k.ha;
            const uint4 r = L.ent[lane * LS];
            key = ((uint64_t)r.y << 32) | r.x, hA1 = (int32_t)r.z, hmeta = r.w;
            if (NH > 0) hv0 = k.tuple[0], hc0 = HU0 ? hv0 >> 1 : (hv0 ? a.pts.tbl[0][hv0] : 0);
            if (NH > 1) hv1 = k.tuple[1], hc1 = HU1 ? hv1 >> 1 : (hv1 ? a.pts.tbl[1][hv1] : 0);
            if (NK > 0) {
                kv0 = k.tuple[kCwKeyPos0];
                if (KU0) kf0 = k.tuple[kCwKeyPos0 + 1], kn0 = k.tuple[kCwKeyPos0 + 2], ke0 = k.tuple[kCwKeyPos0 + 3];
                else if (kv0) {
                    const int64_t x = a.ipa.aff[0][kv0], y = a.ipa.anti[0][kv0], z = a.ipa.exist[0][kv0];
                    over = over || x > 0x3fffffff || y > 0x3fffffff || z > 0x3fffffff;
                    kf0 = (int32_t)x, kn0 = (int32_t)y, ke0 = (int32_t)z;
                }
            }
            if (NK > 1) {
                kv1 = k.tuple[kCwKeyPos1];
                if (KU1) kf1 = k.tuple[kCwKeyPos1 + 1], kn1 = k.tuple[kCwKeyPos1 + 2], ke1 = k.tuple[kCwKeyPos1 + 3];
                else if (kv1) {
                    const int64_t x = a.ipa.aff[1][kv1], y = a.ipa.anti[1][kv1], z = a.ipa.exist[1][kv1];
                    over = over || x > 0x3fffffff || y > 0x3fffffff || z > 0x3fffffff;
                    kf1 = (int32_t)x, kn1 = (int32_t)y, ke1 = (int32_t)z;
                }
            }
        }
        // What the verdicts need not test again in every cycle:
        //  * a lane that is no candidate has nfm == 0 (lanes behind the candidates, lane 63 after a write that was not meant);
        //  * a node without the topology key of a hard constraint carries a count no skew admits (filtering.go:326-330);
        //  * a node without an inter-pod key has zero counts for it, and keeps them (a clone only counts where the key is);
        //  * `nleft`: members of the class list not yet placed, the head included (1 on every other lane) -- the class is
        //    exhausted when it reaches 0, known without waiting for the next head's record to arrive from LDS;
        //  * without TaintToleration / NodeAffinity scores the holder counts stay 1.
        constexpr int32_t kNoKey = 0x7fffffff, kCountCap = 1 << 30; // (counts and skews beyond 2^30 / 2^29: the general kernel's business)
        if (NH > 0) over = over || hc0 >= kCountCap;
        if (NH > 1) over = over || hc1 >= kCountCap;
        if (NH > 0 && cls && (HU0 ? !(hv0 & 1) : hv0 == 0)) hc0 = kNoKey;
        if (NH > 1 && cls && (HU1 ? !(hv1 & 1) : hv1 == 0)) hc1 = kNoKey;
        // count + self - minimum <= maxSkew as count - min(minimum, 2^30) <= maxSkew - self: no overflow with kNoKey on the left, and
        // with no domain present at all (minimum = MaxInt32: every node WITH the key passes, filtering.go:298-305) it still fails
        const int32_t h_rhs0 = h_skew0 - h_self0, h_rhs1 = h_skew1 - h_self1;
        if (NK > 0 && kv0 == 0) kf0 = kn0 = ke0 = 0;
        if (NK > 1 && kv1 == 0) kf1 = kn1 = ke1 = 0;
        uint32_t nleft = 1u;
        if (cls) {
            nleft = 0u;
            for (int m = 0; m < LU; m++) nleft += L.ent[lane * LS + m].y != 0u ? 1u : 0u;
        }
        if (!track) cht = cha = 1u;
        // lane = domain (shared-key hard constraints): count and presence, for the minimum
        int32_t dc0 = 0, dc1 = 0;
        bool dp0 = false, dp1 = false;
        if (NH > 0 && !HU0 && dlane >= 1 && dlane < a.plan.h_len[0]) dc0 = a.pts.tbl[0][dlane], dp0 = a.plan.h_present[0][dlane] != 0;
        if (NH > 1 && !HU1 && dlane >= 1 && dlane < a.plan.h_len[1]) dc1 = a.pts.tbl[1][dlane], dp1 = a.plan.h_present[1][dlane] != 0;
        // the minimum of a hard constraint and how many domains sit at it (filtering.go:298-305): counts only grow, so the minimum
        // moves only when the last domain at it is taken -- shared keys: recomputed then (one DPP reduction); unique keys: the
        // pass's (minimum, nodes at it), and the window ends when they run out
        int32_t min0 = 0x7fffffff, min1 = 0x7fffffff;
        uint32_t nmin0 = 0, nmin1 = 0;
        bool remin0 = NH > 0 && !HU0, remin1 = NH > 1 && !HU1;
        if ((NH > 0 && HU0) || (NH > 1 && HU1))
            for (int c = 0; c < NH; c++)
                if (c == 0 ? HU0 : HU1) {
                    uint32_t m = 0x7fffffffu;
                    for (int b = lane; b < a.w.n_blocks; b += 64) {
                        const uint32_t q = (uint32_t)(a.w.umin[(int64_t)b * kMaxTsc + c] >> 32);
                        m = q < m ? q : m;
                    }
                    m = 0x7fffffffu - wave_max_u32(0x7fffffffu - m);
                    uint32_t n_at = 0;
                    for (int b = lane; b < a.w.n_blocks; b += 64) {
                        const unsigned long long q = a.w.umin[(int64_t)b * kMaxTsc + c];
                        if ((uint32_t)(q >> 32) == m) n_at += (uint32_t)q;
                    }
                    n_at = wave_sum_u32_dpp(n_at);
                    if (c == 0) min0 = (int32_t)m, nmin0 = n_at; else min1 = (int32_t)m, nmin1 = n_at;
                }
        if (__ballot(over) == 0ull) { // (else: an entry beyond int32 -- the general kernel's business)
            int ncand = C, nrec = 0, cycles = 0;
            const int64_t placed0 = cw_uni64(S.placed), limit = cw_uni64(S.limit), log_cap = cw_uni64(S.log_cap);
            int Wl = W; // cycles this window may run: the window, and what --max-limit leaves (simulator.go:297-312: tested after the append)
            if (limit > 0 && limit - placed0 < (int64_t)Wl) Wl = (int)(limit - placed0);
            int log_room = 0; // how many of them the placement log still takes
            if (a.log != nullptr && log_cap > placed0) log_room = log_cap - placed0 < (int64_t)W ? (int)(log_cap - placed0) : W;
            Wl = uni32(Wl), log_room = uni32(log_room);
            // the inter-pod totals enter the verdicts only as (no matching pod anywhere) / (some existing pod has a term): flags,
            // and what the clones add as 32-bit sums for the state
            bool aff_zero = uni32(S.ipa_aff_total == 0) != 0, exist_pos = uni32(S.ipa_exist_total > 0) != 0;
            int32_t d_aff = 0, d_exist = 0, d_ent = 0;
            bool stale_maxima = false, end_window = false, unsched = false;
            uint32_t new_mt = mt_a, new_ma = ma_a, lf = 0; // lf: this lane's share of the feasible count of the last cycle
            int32_t mylog = 0;  // lane l: the winner of cycle (64 j + l), until the 64 are stored together
            uint64_t myrec = 0; // lane l: record (64 j + l) of a node that left the candidates, likewise
            CW_TICK(0);
            // One cycle's verdicts (filtering.go:311-356, interpodaffinity/filtering.go:410-432) on every lane; `&` / `|`: straight-line
            // mask arithmetic.  The loop below is rotated -- verdicts of the NEXT cycle at the bottom, one exit -- so that every way
            // out leaves the lane state where it is (with exits in mid-cycle the compiler copied all of it once per cycle).
            auto verdicts = [&]() -> bool {
                if (NH > 0 && !HU0 && remin0) min0 = wave_min_i32_nonneg(dp0 ? dc0 : 0x7fffffff), nmin0 = (uint32_t)__popcll(__ballot(dp0 && dc0 == min0)), remin0 = false;
                if (NH > 1 && !HU1 && remin1) min1 = wave_min_i32_nonneg(dp1 ? dc1 : 0x7fffffff), nmin1 = (uint32_t)__popcll(__ballot(dp1 && dc1 == min1)), remin1 = false;
                bool ok = nfm > 0;
                if (NH > 0) ok &= hc0 - (h_usemin0 ? (min0 < kCountCap ? min0 : kCountCap) : 0) <= h_rhs0;
                if (NH > 1) ok &= hc1 - (h_usemin1 ? (min1 < kCountCap ? min1 : kCountCap) : 0) <= h_rhs1;
                if (NK > 0 && ipa_filter && (exist_pos || ipa_any_term)) {
                    if (any_aff) {
                        bool pods_exist = true, aff_ok = true;
                        if (k_aff0) aff_ok &= kv0 != 0, pods_exist &= kf0 > 0;
                        if (NK > 1 && k_aff1) aff_ok &= kv1 != 0, pods_exist &= kf1 > 0;
                        ok &= aff_ok & (pods_exist | (aff_zero && self_aff));
                    }
                    if (k_anti0) ok &= kn0 <= 0;
                    if (NK > 1 && k_anti1) ok &= kn1 <= 0;
                    if (exist_pos) {
                        ok &= ke0 <= 0;
                        if (NK > 1) ok &= ke1 <= 0;
                    }
                }
                return ok;
            };
            bool ok = verdicts(), go = true;
            if (__ballot(ok) == 0ull) unsched = true, go = false; // the pass saw every node: schedule_one.go:448-454
            else if (track) {
                const uint32_t mt_now = wave_max_u32(ok ? cmt : 0u), ma_now = wave_max_u32(ok ? cma : 0u);
                if (mt_now != mt_a || ma_now != ma_a) stale_maxima = true, new_mt = mt_now, new_ma = ma_now, go = false; // A was computed under other maxima: redo the pass
            }
            int wl = 0;    // the winner's lane
            int64_t g = 0; // ... and node
            // ---- argmax (selectHost, schedule_one.go:894-941): the key IS (A, lowest index first); its high word carries A
            auto argmax = [&]() {
                lf = ok ? nfm : 0u;
                CW_TICK(2);
                const uint32_t khi = ok ? (uint32_t)(key >> 32) : 0u; // (> 0 on every feasible lane: make_key stores A + 1)
                const uint32_t bhi = wave_max_u32(khi);
                const bool top = ok & (khi == bhi);
                unsigned long long tm = __ballot(top);
                if (tm & (tm - 1ull)) { // several lanes share the score: the low words decide
                    const uint32_t klo = top ? (uint32_t)key : 0u;
                    const uint32_t blo = wave_max_u32(klo);
                    tm = __ballot(top & (klo == blo));
                }
                wl = __builtin_amdgcn_readfirstlane(__ffsll(tm) - 1);
                g = key_index(rl64(key, wl));
                CW_TICK(4);
            };
            // (v_writelane: one lane of a register takes a uniform value -- no compare, no select)
#define CW_PUT(var, val, at) var = (decltype(var))cw_writelane((int)(val), (at), (int)(var))
#define CW_PUT64(var, val, at) var = ((uint64_t)(uint32_t)cw_writelane((int)(uint32_t)((val) >> 32), (at), (int)(uint32_t)((var) >> 32)) << 32) | (uint32_t)cw_writelane((int)(uint32_t)(val), (at), (int)(uint32_t)(var))
            // ---- commit, for a winner that is a class's head (the tag says so) or a node that already received clones in this window.
            // The class form is straight-line: what only happens on one side of a condition is done on both, to a place where it
            // does no harm (lane 63 is no candidate while the loop runs; the next record slot is rewritten before it is stored).
            // With the lane state rewritten on one side of a branch only, the compiler kept it under two names and copied one
            // into the other every cycle.
            auto commit = [&](auto w_cls_tag) {
                constexpr bool w_cls = decltype(w_cls_tag)::value;
                const int32_t w_hv0 = NH > 0 ? rl32(hv0, wl) : 0, w_hv1 = NH > 1 ? rl32(hv1, wl) : 0, w_kv0 = NK > 0 ? rl32(kv0, wl) : 0, w_kv1 = NK > 1 ? rl32(kv1, wl) : 0;
                const int32_t w_hc0 = NH > 0 ? rl32(hc0, wl) : 0, w_hc1 = NH > 1 ? rl32(hc1, wl) : 0; // (before this clone)
                const uint32_t w_meta = (uint32_t)rl32((int32_t)hmeta, wl);
                const uint32_t w_el = (w_meta >> kStatImgShift) & kStatImgMask, w_cnt = (w_meta >> kStatCntShift) & kStatCntMask, w_aff = w_meta & kStatAffMask;
                int32_t A_next;
                uint32_t w_tk; // clones on the winner in this window, this one included
                if (w_cls) {
                    A_next = rl32(hA1, wl), w_tk = 1u;
                    if (lane == wl) { // the class loses its head; the next one's record comes from LDS (nobody waits for it in this cycle)
                        nfm -= 1, head += 1, nleft -= 1;
                        if (track) cht -= w_cnt == cmt ? 1u : 0u, cha -= w_aff == cma ? 1u : 0u;
                        const uint4 r = L.ent[lane * LS + head]; // (member L of every class is the zero record)
                        key = ((uint64_t)r.y << 32) | r.x, hA1 = (int32_t)r.z, hmeta = r.w;
                    }
                } else {
                    w_tk = (uint32_t)rl32((int32_t)tk, wl) + 1u;
                    A_next = uni32(cw_local_after(a, g - a.c.global_offset, (int64_t)w_tk, mt_a, ma_a)); // (a node winning again: one trip to its columns)
                }
                // the clone is an existing pod of the next cycle (filtering.go:255-296, interpodaffinity/filtering.go:204-272)
                int32_t n_hv0 = w_hv0, n_hc0 = w_hc0, n_hv1 = w_hv1, n_hc1 = w_hc1; // the winner's own components after the clone
                uint64_t zbits = 0; // (a clone that counts for a shared-key constraint names its domain in its record: kCwRecZoneShift)
                if (NH > 0) {
                    const bool counts = (w_el & 1u) && ((w_el >> 1) & 1u) && h_self0 && (HU0 ? (w_hv0 & 1) : w_hv0) != 0;
                    if (!HU0 && counts) zbits = (uint64_t)(uint32_t)w_hv0 << kCwRecZoneShift;
                    const bool at_min = counts && w_hc0 == min0;
                    nmin0 -= at_min ? 1u : 0u;
                    if (HU0) {
                        end_window = end_window || (at_min && nmin0 == 0); // the last node at the minimum: the new one takes a pass
                        n_hv0 += counts ? 2 : 0;
                    } else {
                        hc0 += (counts & (hv0 == w_hv0)) ? 1 : 0; // every candidate of the domain, the winner's class included
                        dc0 += (counts & (dlane == w_hv0)) ? 1 : 0;
                        remin0 = remin0 || (at_min && nmin0 == 0);
                    }
                    n_hc0 += counts ? 1 : 0;
                }
                if (NH > 1) {
                    const bool counts = (w_el & 1u) && ((w_el >> 2) & 1u) && h_self1 && (HU1 ? (w_hv1 & 1) : w_hv1) != 0;
                    const bool at_min = counts && w_hc1 == min1;
                    nmin1 -= at_min ? 1u : 0u;
                    if (HU1) {
                        end_window = end_window || (at_min && nmin1 == 0);
                        n_hv1 += counts ? 2 : 0;
                    } else {
                        hc1 += (counts & (hv1 == w_hv1)) ? 1 : 0;
                        dc1 += (counts & (dlane == w_hv1)) ? 1 : 0;
                        remin1 = remin1 || (at_min && nmin1 == 0);
                    }
                    n_hc1 += counts ? 1 : 0;
                }
                int32_t n_kf0 = 0, n_kn0 = 0, n_ke0 = 0, n_kf1 = 0, n_kn1 = 0, n_ke1 = 0;
                if (NK > 0) {
                    const int on = w_kv0 != 0 ? -1 : 0; // (the node has the key: its clone counts, there and on the totals)
                    const int da = k_daff0 & on, dn = k_danti0 & on;
                    d_aff += da, d_exist += dn, d_ent += k_dent0 & on;
                    aff_zero = aff_zero && da == 0, exist_pos = exist_pos || dn != 0;
                    if (!KU0) {
                        const bool same = (kv0 == w_kv0) & (on != 0);
                        kf0 += same ? da : 0, kn0 += same ? dn : 0, ke0 += same ? dn : 0;
                    }
                    n_kf0 = rl32(kf0, wl) + (KU0 ? da : 0), n_kn0 = rl32(kn0, wl) + (KU0 ? dn : 0), n_ke0 = rl32(ke0, wl) + (KU0 ? dn : 0);
                }
                if (NK > 1) {
                    const int on = w_kv1 != 0 ? -1 : 0;
                    const int da = k_daff1 & on, dn = k_danti1 & on;
                    d_aff += da, d_exist += dn, d_ent += k_dent1 & on;
                    aff_zero = aff_zero && da == 0, exist_pos = exist_pos || dn != 0;
                    if (!KU1) {
                        const bool same = (kv1 == w_kv1) & (on != 0);
                        kf1 += same ? da : 0, kn1 += same ? dn : 0, ke1 += same ? dn : 0;
                    }
                    n_kf1 = rl32(kf1, wl) + (KU1 ? da : 0), n_kn1 = rl32(kn1, wl) + (KU1 ? dn : 0), n_ke1 = rl32(ke1, wl) + (KU1 ? dn : 0);
                }
                // does the node stay a candidate?  Full, or blocked for good by a required anti-affinity term against what is there to stay: no
                bool dead = A_next < 0;
                if (NK > 0 && ipa_filter) {
                    if ((k_anti0 && w_kv0 && n_kn0 > 0) || (NK > 1 && k_anti1 && w_kv1 && n_kn1 > 0)) dead = true;
                    if (exist_pos && ((w_kv0 && n_ke0 > 0) || (NK > 1 && w_kv1 && n_ke1 > 0))) dead = true;
                }
                const uint64_t R = zbits | ((uint64_t)w_tk << kIdxBits) | (uint64_t)g; // its (node, clones) record for the epilogue, if it leaves
                if (FULL) { // no lane to keep it in: its record, and the window ends here unless the node is gone for good
                    CW_PUT64(myrec, R, nrec & 63);
                    nrec += 1;
                    if ((nrec & 63) == 0) L.rec[nrec - 64 + lane] = myrec;
                    end_window = end_window || !dead;
                } else if (w_cls) {
                    // it stays: a new lane behind the candidates (it does not: lane 63)
                    const int tl = dead ? 63 : ncand;
                    if (NH > 0) { CW_PUT(hv0, n_hv0, tl); CW_PUT(hc0, n_hc0, tl); }
                    if (NH > 1) { CW_PUT(hv1, n_hv1, tl); CW_PUT(hc1, n_hc1, tl); }
                    if (NK > 0) { CW_PUT(kv0, w_kv0, tl); CW_PUT(kf0, n_kf0, tl); CW_PUT(kn0, n_kn0, tl); CW_PUT(ke0, n_ke0, tl); }
                    if (NK > 1) { CW_PUT(kv1, w_kv1, tl); CW_PUT(kf1, n_kf1, tl); CW_PUT(kn1, n_kn1, tl); CW_PUT(ke1, n_ke1, tl); }
                    CW_PUT(nfm, dead ? 0 : 1, tl); CW_PUT(cmt, w_cnt, tl); CW_PUT(cma, w_aff, tl); CW_PUT(cht, 1, tl); CW_PUT(cha, 1, tl);
                    CW_PUT(hmeta, w_meta, tl); CW_PUT(tk, 1, tl);
                    const uint64_t nkey = make_key((int64_t)(dead ? 0 : A_next), g);
                    CW_PUT64(key, nkey, tl);
                    ncand += dead ? 0 : 1;
                    CW_PUT64(myrec, R, nrec & 63); // (kept if it left: the slot is the next record's otherwise)
                    nrec += dead ? 1 : 0;
                    if (dead && (nrec & 63) == 0) L.rec[nrec - 64 + lane] = myrec;
                } else if (!dead) { // it keeps the lane it has
                    if (NH > 0) { CW_PUT(hv0, n_hv0, wl); CW_PUT(hc0, n_hc0, wl); }
                    if (NH > 1) { CW_PUT(hv1, n_hv1, wl); CW_PUT(hc1, n_hc1, wl); }
                    if (NK > 0) { CW_PUT(kf0, n_kf0, wl); CW_PUT(kn0, n_kn0, wl); CW_PUT(ke0, n_ke0, wl); }
                    if (NK > 1) { CW_PUT(kf1, n_kf1, wl); CW_PUT(kn1, n_kn1, wl); CW_PUT(ke1, n_ke1, wl); }
                    CW_PUT(tk, w_tk, wl);
                    const uint64_t nkey = make_key((int64_t)A_next, g);
                    CW_PUT64(key, nkey, wl);
                } else { // a touched node leaves: its record, and the last candidate lane takes its place
                    CW_PUT64(myrec, R, nrec & 63);
                    nrec += 1;
                    if ((nrec & 63) == 0) L.rec[nrec - 64 + lane] = myrec;
                    const int last = ncand - 1;
                    if (NH > 0) { CW_PUT(hv0, rl32(hv0, last), wl); CW_PUT(hc0, rl32(hc0, last), wl); }
                    if (NH > 1) { CW_PUT(hv1, rl32(hv1, last), wl); CW_PUT(hc1, rl32(hc1, last), wl); }
                    if (NK > 0) { CW_PUT(kv0, rl32(kv0, last), wl); CW_PUT(kf0, rl32(kf0, last), wl); CW_PUT(kn0, rl32(kn0, last), wl); CW_PUT(ke0, rl32(ke0, last), wl); }
                    if (NK > 1) { CW_PUT(kv1, rl32(kv1, last), wl); CW_PUT(kf1, rl32(kf1, last), wl); CW_PUT(kn1, rl32(kn1, last), wl); CW_PUT(ke1, rl32(ke1, last), wl); }
                    CW_PUT(cmt, rl32((int32_t)cmt, last), wl); CW_PUT(cma, rl32((int32_t)cma, last), wl); // (nfm = 1, holders = 1: as before)
                    CW_PUT(hmeta, rl32((int32_t)hmeta, last), wl); CW_PUT(tk, rl32((int32_t)tk, last), wl);
                    const uint64_t mk = rl64(key, last);
                    CW_PUT64(key, mk, wl);
                    CW_PUT(nfm, 0, last); // (the vacated lane is no candidate)
                    ncand -= 1;
                }
                CW_PUT(mylog, (int32_t)g, cycles & 63);
                cycles += 1;
                if ((cycles & 63) == 0 && cycles - 64 + lane < log_room && (!SH || ((int64_t)mylog >= own_lo && (int64_t)mylog < own_hi)))
                    a.log[placed0 + (cycles - 64 + lane)] = mylog; // (SH: a rank logs its own placements; the others' positions stay -1)
                CW_TICK(5);
            };
            // ---- the next cycle: does it run in this window?
            auto next_cycle = [&]() {
                go = !end_window && cycles < Wl && (FULL || ncand < 64);
                if (go) {
                    ok = verdicts();
                    // nothing feasible (the next pass finds out why); a class's next head is not among the members kept; the
                    // class's own maximum lost its last holder; the maxima moved
                    const uint64_t okm = __ballot(ok);
                    bool stop = okm == 0ull;
                    stop |= (__ballot(min(min(cht, cha), nleft) == 0u) & okm) != 0ull;
                    if (track) stop |= (int)((__ballot((cmt > mt_a) | (cma > ma_a)) & okm) != 0ull) | (int)((__ballot(cmt == mt_a) & okm) == 0ull) | (int)((__ballot(cma == ma_a) & okm) == 0ull);
                    go = !stop;
                }
            };
            // ---- SWEEP (round 5): a whole ROUND of placements at once.  One hard constraint over a shared key (zones) and at most a
            // unique-per-node inter-pod key (hostname): when every candidate lane of the cycle at hand is a class whose domain sits AT THE
            // CAP of the skew test (count + self - minimum == maxSkew: with maxSkew 1 every feasible domain does), whose head counts for
            // the constraint, and no two of them share a domain, then the next cycles are known without running them: a clone makes its
            // domain infeasible until the minimum moves (filtering.go:311-356), the minimum moves only when the last domain at it is
            // taken, nothing else a verdict reads changes (a unique key's entries are the winner's own), so cycle p is won by the p-th
            // best head of this cycle's candidates -- the candidates sorted by key -- for as long as the cycle loop's own stop tests
            // pass.  Those are tests on the candidates that are LEFT, so they become positions in the sorted order: the cycle budget;
            // the first head that would stay a candidate after its clone (the loop's business); the last holder of an assumed
            // normalization maximum (DefaultNormalizeScore over the cycle's feasible nodes: cycle p runs iff a holder sits at a position
            // >= p); the candidate that takes the last domain at the minimum (the verdicts after it read another minimum).  Ranks by
            // all-pairs comparison (m readlane steps), facts by position through LDS, then every lane applies its own share: ~1-2 us
            // per round of up to 64 placements instead of ~0.4-0.7 us per placement.  Exactly the cycles the loop would run, in its order.
            constexpr bool SWEEP = NH == 1 && !HU0 && (NK == 0 || (NK == 1 && KU0));
            const bool sweep_on = SWEEP && uni32(a.plan.sweep) != 0 && h_self0 != 0;
            int n_sweeps = 0, n_swept = 0;
            auto sweep = [&]() -> bool {
                const uint64_t okm = __ballot(ok);
                if (__popcll(okm) < 4) return false;
                if (!FULL && (okm >> C) != 0ull) return false; // a node that already took a clone is a candidate: the loop's business
                // the first clone of a run moves the inter-pod totals from "no matching pod anywhere" to "some": the loop's business too
                if (NK > 0 && ((aff_zero && k_daff0 != 0) || (!exist_pos && k_danti0 != 0))) return false;
                const uint32_t el = (hmeta >> kStatImgShift) & kStatImgMask;
                const bool counts = (el & 1u) && ((el >> 1) & 1u) && hv0 != 0;
                const int32_t min_eff = h_usemin0 ? (min0 < kCountCap ? min0 : kCountCap) : 0;
                if (__ballot(ok && !(counts && hc0 - min_eff == h_rhs0)) != 0ull) return false;
                // the best head of every domain takes part (several classes may share a domain: nodes whose own entries differ); the
                // others of the domain stay candidates until their domain's clone lands, i.e. up to the same position
                if (ok) atomicMax(&L.dbest[hv0], (unsigned long long)key);
                L.byrank[lane] = 0u;
                cw_lds_sync();
                const uint64_t dk = ok ? (uint64_t)L.dbest[hv0] : 0ull;
                const bool part = ok && key == dk;
                L.skey[lane] = part ? (unsigned long long)key : 0ull;
                cw_lds_sync();
                if (part) L.dbest[hv0] = 0ull;
                const int m = __popcll(__ballot(part));
                if (m < 4) return false;
                // would the head stay a candidate after its clone?  (commit's `dead`, per lane)
                bool stays = hA1 >= 0;
                if (NK > 0 && ipa_filter && kv0 != 0) {
                    if (k_anti0 && kn0 + k_danti0 > 0) stays = false;
                    if ((exist_pos || k_danti0 != 0) && ke0 + k_danti0 > 0) stays = false;
                }
                // position of every candidate's DOMAIN in the order the cycles would take them (keys are unique: they carry the node
                // index): how many participants lie ahead of the domain's best head -- 64 broadcast reads from LDS
                uint32_t rank = 0;
#pragma unroll 8
                for (int j = 0; j < 64; j++) rank += (uint64_t)L.skey[j] > dk ? 1u : 0u;
                const uint32_t w_cnt = (hmeta >> kStatCntShift) & kStatCntMask, w_aff = hmeta & kStatAffMask;
                if (ok) atomicOr(&L.byrank[rank], (part ? ((hc0 == min0 ? 1u : 0u) | (stays ? 2u : 0u)) : 0u) | (cmt == mt_a ? 4u : 0u) | (cma == ma_a ? 8u : 0u));
                cw_lds_sync();
                const uint32_t pfl = lane < m ? L.byrank[lane] : 0u; // lane = position from here on
                const uint64_t M_min = __ballot((pfl & 1u) != 0u), M_stay = __ballot((pfl & 2u) != 0u), M_t = __ballot((pfl & 4u) != 0u), M_a = __ballot((pfl & 8u) != 0u);
                int T = m < Wl - cycles ? m : Wl - cycles;
                if (M_stay != 0ull) { const int q = __ffsll((unsigned long long)M_stay) - 1; T = T < q ? T : q; }
                if (track) {
                    const int jt = M_t ? 64 - __clzll(M_t) : 0, ja = M_a ? 64 - __clzll(M_a) : 0;
                    T = T < jt ? T : jt, T = T < ja ? T : ja;
                }
                if (nmin0 > 0u && (uint32_t)__popcll(M_min) >= nmin0) { // the candidate that takes the last domain at the minimum ends the round
                    uint64_t mm = M_min;
                    for (uint32_t q = 1; q < nmin0; q++) mm &= mm - 1ull;
                    const int cut = __ffsll((unsigned long long)mm); // (1-based: positions 0 .. cut - 1 run)
                    T = T < cut ? T : cut;
                }
                T = uni32(T);
                if (T < 2) return false;
                // ---- the T cycles, every lane its own share
                const bool take = part && rank < (uint32_t)T;
                lf = (ok && rank >= (uint32_t)(T - 1)) ? nfm : 0u; // the feasible nodes of the last of them
                const int64_t gi = key_index(key);
                const int cycles0 = cycles, nrec0 = nrec;
                // (node, clones) records and the log: positions cycles0 + rank; the lanes keep the current group of 64 as the loop does
                if (lane < (nrec0 & 63)) L.rec[(nrec0 & ~63) + lane] = myrec;
                if (lane < (cycles0 & 63)) {
                    if ((cycles0 & ~63) + lane < log_room && (!SH || ((int64_t)mylog >= own_lo && (int64_t)mylog < own_hi))) a.log[placed0 + ((cycles0 & ~63) + lane)] = mylog;
                    L.lg[lane] = mylog;
                }
                cw_lds_sync();
                if (take) {
                    L.dflag[hv0] = 1;
                    L.rec[nrec0 + (int)rank] = ((uint64_t)(uint32_t)hv0 << kCwRecZoneShift) | (1ull << kIdxBits) | (uint64_t)gi; // (a sweep's clones all count)
                    const int pos = cycles0 + (int)rank;
                    if (pos < log_room && (!SH || (gi >= own_lo && gi < own_hi))) a.log[placed0 + pos] = (int32_t)gi;
                    if (pos >= ((cycles0 + T) & ~63)) L.lg[pos & 63] = (int32_t)gi;
                }
                cw_lds_sync();
                hc0 += L.dflag[hv0 <= 64 ? hv0 : 0];
                dc0 += dlane <= 64 ? L.dflag[dlane] : 0;
                cycles += T, nrec += T;
                if (lane < (nrec & 63)) myrec = L.rec[(nrec & ~63) + lane];
                if (lane < (cycles & 63)) mylog = L.lg[lane];
                cw_lds_sync();
                if (take) L.dflag[hv0] = 0;
                const int dmin = __popcll(M_min & (T >= 64 ? ~0ull : ((1ull << T) - 1ull)));
                nmin0 -= (uint32_t)dmin;
                remin0 = remin0 || (dmin > 0 && nmin0 == 0u);
                if (NK > 0) {
                    const int n_on = __popcll(__ballot(take && kv0 != 0));
                    d_aff += k_daff0 * n_on, d_exist += k_danti0 * n_on, d_ent += k_dent0 * n_on;
                    aff_zero = aff_zero && !(n_on != 0 && k_daff0 != 0), exist_pos = exist_pos || (n_on != 0 && k_danti0 != 0);
                }
                if (take) { // the class loses its head (commit's class form)
                    nfm -= 1, head += 1, nleft -= 1;
                    if (track) cht -= w_cnt == cmt ? 1u : 0u, cha -= w_aff == cma ? 1u : 0u;
                    const uint4 r = L.ent[lane * LS + head];
                    key = ((uint64_t)r.y << 32) | r.x, hA1 = (int32_t)r.z, hmeta = r.w;
                }
                n_sweeps += 1, n_swept += T;
                return true;
            };
            // The hot loop runs while class heads win (every cycle of a pod whose clones exclude each other); a node winning AGAIN
            // takes a trip to its columns (cw_local_after, with loops over the extra resources): kept out of the hot loop's body,
            // whose register allocation it spoiled (SGPR spills reloaded in every cycle).
            for (;;) {
                bool again = false;
                asm volatile("" ::: "memory"); // (keeps this header apart from the hot loop's: merged, they are one loop with the slow path inside)
#pragma unroll 1
                while (uni32(go)) {
                    if (SWEEP && sweep_on && uni32(sweep())) {
                        next_cycle();
                        continue;
                    }
                    argmax();
                    if (wl >= C) {
                        again = true;
                        break;
                    }
                    commit(std::true_type{});
                    next_cycle();
                }
                if (!uni32(again)) break;
                commit(std::false_type{});
                next_cycle();
            }
#undef CW_PUT
#undef CW_PUT64
            cw_lds_sync();
            ncand = uni32(ncand), nrec = uni32(nrec), cycles = uni32(cycles);
            // what the lanes still hold: the log's last partial group, the records' last partial group, the touched nodes that stayed
            if (lane < (cycles & 63) && (cycles & ~63) + lane < log_room && (!SH || ((int64_t)mylog >= own_lo && (int64_t)mylog < own_hi))) a.log[placed0 + ((cycles & ~63) + lane)] = mylog;
            if (lane < (nrec & 63)) L.rec[(nrec & ~63) + lane] = myrec;
            if (lane >= C && lane < ncand) L.rec[nrec + (lane - C)] = ((uint64_t)tk << kIdxBits) | (uint64_t)key_index(key);
            const uint32_t nf_last = wave_sum_u32_dpp(lf);
            if (lane == 0) {
                const int64_t placed = placed0 + cycles;
                S.placed = placed, S.rounds += cycles + (unsched ? 1 : 0), S.scans += 1;
                if (NK > 0) S.ipa_aff_total += d_aff, S.ipa_exist_total += d_exist, S.ipa_entries += d_ent;
                if (unsched) S.last_feasible = 0;
                else if (cycles > 0) S.last_feasible = (int32_t)nf_last;
                S.winner = -1;
                if (stale_maxima) S.mt_a = (int32_t)new_mt, S.ma_a = (int32_t)new_ma;
                if (NH > 0) S.pts_min_a[0] = min0; // (the terminal histogram reads them: k_hist)
                if (NH > 1) S.pts_min_a[1] = min1;
                S.cw_windows += 1;
                S.cw_sweeps += n_sweeps, S.cw_swept += n_swept;
                if (FULL) S.cw_full_windows += 1; else S.cw_fast_windows += 1;
                S.done = unsched ? DONE_UNSCHEDULABLE : (limit > 0 && placed >= limit ? DONE_LIMIT : 0);
                L.s_nt = nrec + (ncand - C);
                if (PROF) L.pf[7] += (unsigned long long)cycles;
            }
        }
    }
    __syncthreads();
    const int nt = L.s_nt;
    if (nt < 0) return; // not taken: the general kernel runs the window
    // ---- epilogue (all threads): everything a placement changes follows from (node, clones): columns (NodeInfo.update,
    // types.go:409-428) and table entries at the node's own topology values -- integer adds, in any order
    for (int ti = tid; ti < nt; ti += kCwThreads) {
        const unsigned long long R = L.rec[ti];
        const int64_t i = (int64_t)(R & kIdxMask) - a.c.global_offset;
        const int64_t k = (int64_t)((R >> kIdxBits) & kCwRecClonesMask);
        if (SH) { // the shared key's domain count: every rank (the table is replicated); everything else: the node's owner
            const int32_t v = (int32_t)(R >> kCwRecZoneShift);
            if (NH > 0 && v) atomicAdd(&a.pts.tbl[0][v], (int32_t)k);
            if (i < 0 || i >= a.c.n) continue;
        }
        const int64_t r0 = a.c.req[0][i] + k * a.p.req[0], r1 = a.c.req[1][i] + k * a.p.req[1];
        const int64_t z0 = a.c.nz_mcpu[i] + k * a.p.nz_mcpu, z1 = a.c.nz_mem[i] + k * a.p.nz_mem;
        a.c.req[0][i] = r0, a.c.req[1][i] = r1, a.c.nz_mcpu[i] = z0, a.c.nz_mem[i] = z1;
        a.c.pod_count[i] += (int32_t)k, a.c.placed_cnt[i] += (int32_t)k;
        store_mirror(a.c, i, r0, r1, z0, z1);
#pragma unroll 1
        for (int col = 2; col < a.p.ncol; col++)
            if (a.p.req[col] != 0) a.c.req[col][i] += k * a.p.req[col];
        if (NH > 0 && !SH) {
            const uint32_t eb = a.pts.elig[i];
            for (int c = 0; c < NH; c++) {
                const int32_t v = a.pts.label[c][i];
                if (v && (eb & 1u) && ((eb >> (1 + c)) & 1u) && a.pts.self_match[c]) atomicAdd(&a.pts.tbl[c][v], (int32_t)k);
            }
        }
        for (int kk = 0; kk < NK; kk++) {
            const int32_t v = a.ipa.label[kk][i];
            if (!v) continue;
            if (a.ipa.self_aff && a.ipa.aff_terms_on_key[kk]) atomicAdd((unsigned long long *)&a.ipa.aff[kk][v], (unsigned long long)(k * a.ipa.aff_terms_on_key[kk]));
            if (a.ipa.anti_self_on_key[kk]) {
                atomicAdd((unsigned long long *)&a.ipa.anti[kk][v], (unsigned long long)(k * a.ipa.anti_self_on_key[kk]));
                atomicAdd((unsigned long long *)&a.ipa.exist[kk][v], (unsigned long long)(k * a.ipa.anti_self_on_key[kk]));
            }
            if (a.ipa.score_self[kk]) atomicAdd((unsigned long long *)&a.ipa.score[kk][v], (unsigned long long)(k * a.ipa.score_self[kk]));
        }
    }
    // ---- leave the class table empty for the next pass, and tell the general kernel that the window is done
    const int C_loc = SH ? (int)a.w.ctl[kCwCtlClasses] : C; // (SH: C counts the cluster's classes, the table holds this rank's)
    for (int id = tid; id < C_loc; id += kCwThreads) {
        const int g = a.w.slot_of_id[id];
        a.w.keys[g] = 0ull, a.w.ready[g] = 0u;
        CwClass &k = a.w.cls[g];
        k.nf = k.mt = k.ma = k.ht = k.ha = 0u;
    }
    __syncthreads();
    if (tid == 0) {
        a.w.ctl[kCwCtlClasses] = 0u;
        __hip_atomic_store(a.w.ctl + kCwCtlFastDone, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (PROF) {
            L.pf[6] += __builtin_amdgcn_s_memrealtime() - t_prev;
            for (int i = 0; i < 8; i++) a.w.prof[i] += L.pf[i];
        }
    }
#undef CW_TICK
}

// (the argument block is read through a pointer: its arrays are indexed with runtime constraint numbers, which on a by-value
// kernel argument means a 2.5 KB scratch copy per lane; from memory they are scalar loads)
template <int MH, int MS, int MK>
__global__ __launch_bounds__(kCwThreads) void k_cw_decide(const CwDecideArgs *__restrict__ ap) {
    const CwDecideArgs &a = *ap;
    extern __shared__ __attribute__((aligned(16))) unsigned char cw_lds_raw[];
    CwLds &L = *reinterpret_cast<CwLds *>(cw_lds_raw);
    DevState &S = *a.st;
    // k_cw_decide_fast did this window?  ONE read for the whole workgroup (thread 0 -> LDS -> barrier): the flag is cleared
    // right here, and a wave that loaded it after the store would run the general body on a window already committed
    if (threadIdx.x == 0) {
        L.fast_done = __hip_atomic_load(a.w.ctl + kCwCtlFastDone, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (L.fast_done) __hip_atomic_store(a.w.ctl + kCwCtlFastDone, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (L.fast_done) return;
    if (S.done || S.cw_fallback) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int LL = uni32(a.plan.list_len), W = uni32(a.plan.window < kCwMaxWindow ? a.plan.window : kCwMaxWindow);
    const bool giveup = __hip_atomic_load(a.w.ctl + kCwCtlGiveUp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
    const int C = uni32(giveup ? 0 : (int)a.w.ctl[kCwCtlClasses]);
    const bool staged = C * LL <= kCwListLds; // else list entries are read from HBM when they are needed
    unsigned long long t_prev = __builtin_amdgcn_s_memrealtime(), pf[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const bool prof = a.w.prof != nullptr;
#define CW_TICK(i) do { if (prof) { const unsigned long long t_now = __builtin_amdgcn_s_memrealtime(); pf[i] += t_now - t_prev; t_prev = t_now; } } while (0)

    // ---- prologue (all threads): class records, list entries with their nodes' facts, shared-key tables -> LDS
    if (!giveup) {
        for (int q = tid; q < C * kCwTuple; q += kCwThreads) {
            const int id = q / kCwTuple;
            L.c_tuple[id][q % kCwTuple] = a.w.cls[a.w.slot_of_id[id]].tuple[q % kCwTuple];
        }
        for (int id = tid; id < C; id += kCwThreads) {
            const CwClass &k = a.w.cls[a.w.slot_of_id[id]];
            L.c_nf[id] = k.nf, L.c_mt[id] = k.mt, L.c_ma[id] = k.ma, L.c_ht[id] = k.ht, L.c_ha[id] = k.ha;
            L.c_head[id] = 0;
            L.c_key[id] = a.w.lists[(size_t)id * LL];
        }
        if (staged)
            for (int q = tid; q < C * LL; q += kCwThreads) {
                const unsigned long long key = a.w.lists[q];
                L.li_key[q] = key;
                if (key) {
                    const int64_t i = key_index(key) - a.c.global_offset;
                    L.li_stat[q] = a.c.stat[i];
                    L.li_elig[q] = (a.pts.n ? (uint32_t)a.pts.elig[i] : 0u) | ((a.soft.n ? (uint32_t)a.soft.elig[i] : 0u) << 16);
                    L.li_A1[q] = a.w.node_A1[i];
                }
            }
        for (int c = 0; c < a.pts.n; c++)
            if (!a.plan.h_unique[c])
                for (int v = tid; v < a.plan.h_len[c]; v += kCwThreads) {
                    L.i32[a.plan.h_off[c] + v] = a.pts.tbl[c][v];
                    L.i32[a.plan.h_pres[c] + v] = v ? a.plan.h_present[c][v] : 0;
                }
        for (int c = 0; c < a.soft.n; c++)
            if (!a.soft.is_hostname[c])
                for (int v = tid; v < a.plan.s_len[c]; v += kCwThreads) L.i32[a.plan.s_off[c] + v] = a.soft.tbl[c][v];
        if (a.ipa.on)
            for (int k = 0; k < a.ipa.n_keys; k++)
                if (!a.plan.k_unique[k]) {
                    const int len = a.plan.k_len[k];
                    for (int v = tid; v < len; v += kCwThreads) {
                        L.i64[a.plan.k_off[k] + v] = a.ipa.aff[k][v];
                        L.i64[a.plan.k_off[k] + len + v] = a.ipa.anti[k][v];
                        L.i64[a.plan.k_off[k] + 2 * len + v] = a.ipa.exist[k][v];
                        L.i64[a.plan.k_off[k] + 3 * len + v] = a.ipa.score[k][v];
                    }
                }
    }
    if (tid == 0) L.s_nt = 0;
    __syncthreads();

    if (tid < 64) {
        // ================= the cycle loop: wave 0 only, no block barriers =================
        CwP<MH, MS, MK> P;
        P.load(a);
        CW_TICK(0);
        int nt = 0, na = 0; // touched nodes ; how many of them may still win
        int64_t placed = cw_uni64(S.placed), rounds = cw_uni64(S.rounds);
        const int64_t limit = cw_uni64(S.limit), log_cap = cw_uni64(S.log_cap);
        int64_t aff_total = cw_uni64(S.ipa_aff_total), exist_total = cw_uni64(S.ipa_exist_total), entries = cw_uni64(S.ipa_entries);
        const uint32_t mt_a = (uint32_t)uni32(S.mt_a), ma_a = (uint32_t)uni32(S.ma_a);
        int done = 0, last_feasible = uni32(S.last_feasible);
        bool stale_maxima = false;
        uint32_t new_mt = mt_a, new_ma = ma_a;
        // unique-key hard constraints: (minimum, counted nodes at it) from the pass
        if (lane < kMaxTsc) L.u_min[lane] = 0x7fffffff, L.u_cnt[lane] = 0, L.mn[lane] = 0x7fffffff, L.soft_w[lane] = 0, L.soft_size[lane] = -1;
#pragma unroll
        for (int c = 0; c < MH; c++)
            if (c < P.nh && P.h_unique[c]) {
                uint32_t m = 0x7fffffffu;
                for (int b = lane; b < a.w.n_blocks; b += 64) {
                    const uint32_t q = (uint32_t)(a.w.umin[(int64_t)b * kMaxTsc + c] >> 32);
                    m = q < m ? q : m;
                }
                m = 0x7fffffffu - wave_max_u32(0x7fffffffu - m);
                uint32_t n_at = 0;
                for (int b = lane; b < a.w.n_blocks; b += 64) {
                    const unsigned long long q = a.w.umin[(int64_t)b * kMaxTsc + c];
                    if ((uint32_t)(q >> 32) == m) n_at += (uint32_t)q;
                }
                n_at = wave_sum_u32_dpp(n_at);
                if (lane == 0) L.u_min[c] = (int32_t)m, L.u_cnt[c] = n_at;
            }
        bool end_window = giveup;
        int cycles = 0;
        cw_lds_sync();
        CW_TICK(1);

#pragma unroll 1
        while (!end_window && !done && cycles < W) {
            // ---- minima of the hard constraints (filtering.go:298-305)
#pragma unroll
            for (int c = 0; c < MH; c++)
                if (c < P.nh) {
                    int32_t mc;
                    if (P.h_unique[c]) mc = L.u_min[c];
                    else {
                        uint32_t m = 0x7fffffffu;
                        for (int v = 1 + lane; v < P.h_len[c]; v += 64)
                            if (L.i32[P.h_pres[c] + v]) {
                                const uint32_t q = (uint32_t)L.i32[P.h_off[c] + v];
                                m = q < m ? q : m;
                            }
                        mc = (int32_t)(0x7fffffffu - wave_max_u32(0x7fffffffu - m));
                    }
                    if (lane == 0) L.mn[c] = mc;
                }
#pragma unroll
            for (int c = 0; c < MS; c++)
                if (c < P.ns && !P.s_host[c] && P.soft_w)
                    for (int q = lane; q < (P.s_len[c] + 31) / 32; q += 64) L.i32[P.s_bm[c] + q] = 0;
            cw_lds_sync();
            const int ncand = C + na;
            // ---- stage 1: every candidate's verdict; feasible count, ignored count, maxima, candidate domains
            uint32_t nf = 0, nign = 0, mt_now = 0, ma_now = 0;
            bool unknown = false, need_head = false;
            for (int base = 0; base < ncand; base += 64) {
                const int q = base + lane;
                uint32_t fl = 0;
                if (q < ncand) {
                    const bool is_cls = q < C;
                    const int ti = is_cls ? 0 : L.alive[q - C];
                    const int32_t *t = is_cls ? L.c_tuple[q] : L.t_tuple[ti];
                    const bool node_ok = is_cls ? L.c_nf[q] > 0 : true;
                    bool dead = false;
                    if (node_ok && cw_coupled_ok(P, L, t, aff_total, exist_total, &dead)) {
                        const bool sk = cw_soft_keys(P, t);
                        fl = 1u | (sk ? 2u : 0u);
                        const uint32_t members = is_cls ? L.c_nf[q] : 1u;
                        nf += members;
                        if (!sk) nign += members;
                        const uint32_t cm = is_cls ? L.c_mt[q] : L.t_cnt[ti], ca = is_cls ? L.c_ma[q] : L.t_aff[ti];
                        mt_now = cm > mt_now ? cm : mt_now, ma_now = ca > ma_now ? ca : ma_now;
                        if (is_cls && (L.c_ht[q] == 0 || L.c_ha[q] == 0)) unknown = true; // the class's own maximum lost its last holder
                        if (is_cls && L.c_key[q] == 0ull) need_head = true;                 // its next head is not among the members kept
                        if (sk && P.soft_w) {
#pragma unroll
                            for (int c = 0; c < MS; c++)
                                if (c < P.ns && !P.s_host[c]) {
                                    const int32_t v = t[P.s_comp[c]];
                                    atomicOr((unsigned int *)&L.i32[P.s_bm[c] + (v >> 5)], 1u << (v & 31));
                                }
                        }
                    }
                    L.e_fl[q] = fl;
                }
            }
            nf = wave_sum_u32_dpp(nf), nign = wave_sum_u32_dpp(nign);
            mt_now = wave_max_u32(mt_now), ma_now = wave_max_u32(ma_now);
            unknown = __ballot(unknown) != 0ull, need_head = __ballot(need_head) != 0ull;
            CW_TICK(2);
            if (nf == 0) {
                if (cycles == 0) { // the pass saw every node: schedule_one.go:448-454
                    done = DONE_UNSCHEDULABLE, rounds += 1, last_feasible = 0;
                }
                break; // else: nothing feasible among what the window knows -- the next pass decides
            }
            if (cycles == 0 && (mt_now != mt_a || ma_now != ma_a)) { // A was computed under other maxima: redo the pass
                stale_maxima = true, new_mt = mt_now, new_ma = ma_now;
                break;
            }
            if (cycles > 0 && (unknown || need_head || mt_now != mt_a || ma_now != ma_a)) break;
            // ---- stage 2: PodTopologySpread weights (scoring.go:96-113,294-296), raw scores, their min / max
            const bool soft_on = P.ns > 0 && P.soft_w;
            const bool ipa_on = P.nk > 0 && P.ipa_w && entries > 0; // else PreScore Skip (scoring.go:199-201)
            int64_t p_mn = INT64_MAX, p_mx = 0, i_mn = INT64_MAX, i_mx = INT64_MIN;
            if (soft_on || ipa_on) {
                cw_lds_sync(); // (the candidate bitmaps and flags above)
                if (soft_on) {
#pragma unroll
                    for (int c = 0; c < MS; c++)
                        if (c < P.ns) {
                            int64_t sz;
                            if (P.s_host[c]) sz = (int64_t)nf - (int64_t)nign;
                            else {
                                uint32_t bits = 0;
                                for (int q = lane; q < (P.s_len[c] + 31) / 32; q += 64) bits += (uint32_t)__popc((unsigned)L.i32[P.s_bm[c] + q]);
                                sz = wave_sum_u32_dpp(bits);
                            }
                            if (sz != L.soft_size[c] && lane == 0) L.soft_size[c] = sz, L.soft_w[c] = go_log((double)(sz + 2));
                        }
                }
                cw_lds_sync();
                for (int base = 0; base < ncand; base += 64) {
                    const int q = base + lane;
                    if (q < ncand && (L.e_fl[q] & 1u)) {
                        const int32_t *t = q < C ? L.c_tuple[q] : L.t_tuple[L.alive[q - C]];
                        if (soft_on && (L.e_fl[q] & 2u)) { // scoring.go:196-223
                            double sc = 0;
#pragma unroll
                            for (int c = 0; c < MS; c++)
                                if (c < P.ns) {
                                    const int32_t tv = t[P.s_comp[c]];
                                    if (!P.s_host[c] && tv == P.s_nocredit[c] && tv != 0) continue; // key missing under the system defaults: no credit (scoring.go:210)
                                    const int64_t ct = P.s_host[c] ? (int64_t)(tv >> 1) : (int64_t)L.i32[P.s_off[c] + tv];
                                    sc += (double)ct * L.soft_w[c] + (double)(P.s_skew[c] - 1);
                                }
                            const int64_t raw = (int64_t)round(sc);
                            L.e_rp[q] = raw;
                            p_mn = raw < p_mn ? raw : p_mn, p_mx = raw > p_mx ? raw : p_mx;
                        }
                        if (ipa_on) { // scoring.go:226-247
                            int64_t raw = 0;
#pragma unroll
                            for (int k = 0; k < MK; k++)
                                if (k < P.nk) {
                                    const int c0 = P.k_comp[k];
                                    const int32_t v = t[c0];
                                    if (v) raw += P.k_unique[k] ? (int64_t)t[c0 + 4] : L.i64[P.k_off[k] + 3 * P.k_len[k] + v];
                                }
                            L.e_ri[q] = raw;
                            i_mn = raw < i_mn ? raw : i_mn, i_mx = raw > i_mx ? raw : i_mx;
                        }
                    }
                }
                if (soft_on) {
                    p_mn = INT64_MAX - (int64_t)wave_max_u64((uint64_t)(INT64_MAX - p_mn)); // (values >= 0)
                    p_mx = (int64_t)wave_max_u64((uint64_t)p_mx);
                }
                if (ipa_on) { // signed: bias to unsigned order
                    i_mn = (int64_t)(~wave_max_u64(~((uint64_t)i_mn ^ 0x8000000000000000ull)) ^ 0x8000000000000000ull);
                    i_mx = (int64_t)(wave_max_u64((uint64_t)i_mx ^ 0x8000000000000000ull) ^ 0x8000000000000000ull);
                }
            }
            CW_TICK(3);
            // ---- stage 3: totals, argmax (selectHost, schedule_one.go:894-941: lowest index among the maxima)
            uint64_t best = 0;
            int best_q = -1;
            for (int base = 0; base < ncand; base += 64) {
                const int q = base + lane;
                if (q < ncand && (L.e_fl[q] & 1u)) {
                    const uint64_t hk = q < C ? L.c_key[q] : 0ull;
                    const int ti = q < C ? 0 : L.alive[q - C];
                    int64_t total = q < C ? key_score(hk) : (int64_t)L.t_A[ti];
                    const int64_t gi = q < C ? key_index(hk) : (int64_t)L.t_gidx[ti];
                    if (soft_on && (L.e_fl[q] & 2u)) total += soft_normalize(L.e_rp[q], p_mn, p_mx) * P.soft_w; // ignored nodes score 0
                    if (ipa_on) total += ipa_normalize(L.e_ri[q], i_mn, i_mx) * P.ipa_w;
                    const uint64_t key = make_key(total, gi);
                    if (key > best) best = key, best_q = q;
                }
            }
            const uint64_t wbest = wave_max_u64(best);
            const uint64_t owner = __ballot(best == wbest && best_q >= 0);
            const int wl = __ffsll((unsigned long long)owner) - 1;
            const int wq = __builtin_amdgcn_readlane(best_q, uni32(wl));
            const int64_t g = key_index(wbest);
            CW_TICK(4);
            // ---- commit (every lane holds the same wq / g; lane 0 writes)
            int ti; // the winner's touched record
            int32_t A_next; // its local score after this clone, -1 = it takes no further clone
            if (wq < C) {
                ti = nt;
                const int hd0 = L.c_head[wq], e0 = wq * LL + hd0;
                uint32_t w, el;
                if (staged) w = (uint32_t)uni32((int)L.li_stat[e0]), el = (uint32_t)uni32((int)L.li_elig[e0]), A_next = uni32(L.li_A1[e0]);
                else {
                    const int64_t i = g - a.c.global_offset;
                    w = a.c.stat[i], A_next = a.w.node_A1[i];
                    el = (a.pts.n ? (uint32_t)a.pts.elig[i] : 0u) | ((a.soft.n ? (uint32_t)a.soft.elig[i] : 0u) << 16);
                }
                if (lane < kCwTuple) L.t_tuple[ti][lane] = L.c_tuple[wq][lane];
                if (lane == 0) {
                    const uint32_t cnt = (w >> kStatCntShift) & kStatCntMask, aff = w & kStatAffMask;
                    L.t_gidx[ti] = g, L.t_cnt[ti] = cnt, L.t_aff[ti] = aff, L.t_took[ti] = 0, L.t_elig[ti] = el;
                    L.c_nf[wq] -= 1;
                    L.c_ht[wq] -= cnt == L.c_mt[wq] ? 1u : 0u;
                    L.c_ha[wq] -= aff == L.c_ma[wq] ? 1u : 0u;
                    const int hd = hd0 + 1;
                    L.c_head[wq] = hd;
                    L.c_key[wq] = hd < LL ? (staged ? L.li_key[e0 + 1] : a.w.lists[(size_t)wq * LL + hd]) : 0ull;
                }
                nt += 1;
            } else {
                ti = uni32(L.alive[wq - C]);
                A_next = uni32(cw_local_after(a, g - a.c.global_offset, (int64_t)uni32((int)L.t_took[ti]) + 1, mt_a, ma_a)); // (a node winning again: one trip to its columns)
            }
            cw_lds_sync();
            const uint32_t el = (uint32_t)uni32((int)L.t_elig[ti]);
            // the clone is an existing pod of the next cycle: tables, the node's own entries, totals
#pragma unroll
            for (int c = 0; c < MH; c++)
                if (c < P.nh) { // filtering.go:255-296
                    const int q = P.h_comp[c];
                    const int32_t tv = L.t_tuple[ti][q];
                    const bool counted = (el & 1u) && ((el >> (1 + c)) & 1u) && P.h_self[c];
                    if (P.h_unique[c]) {
                        if (counted && (tv & 1)) {
                            if ((tv >> 1) == L.u_min[c]) { // the winner leaves the minimum
                                const uint32_t left = L.u_cnt[c] - 1;
                                if (lane == 0) L.u_cnt[c] = left;
                                if (left == 0) end_window = true; // the new minimum takes a pass over the nodes
                            }
                            if (lane == 0) L.t_tuple[ti][q] = tv + 2;
                        }
                    } else if (counted && tv && lane == 0)
                        L.i32[P.h_off[c] + tv] += 1;
                }
#pragma unroll
            for (int c = 0; c < MS; c++)
                if (c < P.ns) { // scoring.go:147-178
                    const int q = P.s_comp[c];
                    const int32_t tv = L.t_tuple[ti][q];
                    const uint32_t se = el >> 16;
                    if (P.s_host[c]) {
                        if (P.s_self[c] && lane == 0) L.t_tuple[ti][q] = tv + 2;
                    } else if (tv && (se & 1u) && ((se >> (1 + c)) & 1u) && P.s_self[c] && lane == 0)
                        L.i32[P.s_off[c] + tv] += 1;
                }
#pragma unroll
            for (int k = 0; k < MK; k++)
                if (k < P.nk) { // filtering.go:204-272, scoring.go:81-125
                    const int q = P.k_comp[k];
                    const int32_t v = L.t_tuple[ti][q];
                    if (v) {
                        const int64_t d_aff = P.k_daff[k], d_anti = P.k_danti[k];
                        aff_total += d_aff, exist_total += d_anti, entries += P.k_dent[k];
                        if (lane == 0) {
                            if (P.k_unique[k]) {
                                L.t_tuple[ti][q + 1] += (int32_t)d_aff, L.t_tuple[ti][q + 2] += (int32_t)d_anti, L.t_tuple[ti][q + 3] += (int32_t)d_anti;
                                L.t_tuple[ti][q + 4] += (int32_t)P.k_dscore[k];
                            } else {
                                const int len = P.k_len[k], o = P.k_off[k];
                                L.i64[o + v] += d_aff, L.i64[o + len + v] += d_anti, L.i64[o + 2 * len + v] += d_anti, L.i64[o + 3 * len + v] += P.k_dscore[k];
                            }
                        }
                    }
                }
            if (lane == 0) {
                L.t_A[ti] = A_next;
                L.t_took[ti] += 1;
                if (a.log && placed < log_cap) a.log[placed] = (int32_t)g;
            }
            cw_lds_sync();
            // does the node stay a candidate?  Full, or blocked for good by its own clone (required anti-affinity): no.
            {
                bool dead = A_next < 0;
                if (!dead) (void)cw_coupled_ok(P, L, L.t_tuple[ti], aff_total, exist_total, &dead);
                const bool was_alive = wq >= C;
                if (!dead && !was_alive) {
                    if (lane == 0) L.alive[na] = ti;
                    na += 1;
                } else if (dead && was_alive) { // remove: the last alive entry takes its place
                    if (lane == 0) L.alive[wq - C] = L.alive[na - 1];
                    na -= 1;
                }
            }
            placed += 1, rounds += 1, cycles += 1;
            last_feasible = (int32_t)nf;
            if (limit > 0 && placed >= limit) done = DONE_LIMIT; // simulator.go:297-312: tested after the append
            cw_lds_sync();
            CW_TICK(5);
        }
        // ---- the window is over: run state back to the device struct (lane 0)
        if (lane == 0) {
            S.placed = placed, S.rounds = rounds, S.scans += 1;
            S.ipa_aff_total = aff_total, S.ipa_exist_total = exist_total, S.ipa_entries = entries;
            S.last_feasible = last_feasible;
            S.winner = -1;
            if (stale_maxima) S.mt_a = (int32_t)new_mt, S.ma_a = (int32_t)new_ma;
            if (giveup) S.cw_fallback = 1;
            for (int c = 0; c < P.nh; c++) S.pts_min_a[c] = L.mn[c]; // (the terminal histogram reads it: k_hist)
            S.cw_windows += 1;
            S.done = done;
            L.s_nt = nt;
            if (prof) pf[7] += (unsigned long long)cycles;
        }
    }
    __syncthreads();
    const int nt = L.s_nt;
    // ---- epilogue (all threads): the window's placements onto the columns (NodeInfo.update, types.go:409-428) and the tables
    // back to HBM -- they are the canonical state every other path reads
    if (!giveup) {
        for (int ti = tid; ti < nt; ti += kCwThreads) {
            const int64_t i = L.t_gidx[ti] - a.c.global_offset;
            const int64_t k = (int64_t)L.t_took[ti];
            const int64_t r0 = a.c.req[0][i] + k * a.p.req[0], r1 = a.c.req[1][i] + k * a.p.req[1];
            const int64_t z0 = a.c.nz_mcpu[i] + k * a.p.nz_mcpu, z1 = a.c.nz_mem[i] + k * a.p.nz_mem;
            a.c.req[0][i] = r0, a.c.req[1][i] = r1, a.c.nz_mcpu[i] = z0, a.c.nz_mem[i] = z1;
            a.c.pod_count[i] += (int32_t)k, a.c.placed_cnt[i] += (int32_t)k;
            store_mirror(a.c, i, r0, r1, z0, z1);
#pragma unroll 1
            for (int col = 2; col < a.p.ncol; col++)
                if (a.p.req[col] != 0) a.c.req[col][i] += k * a.p.req[col];
            // unique keys: the node's own table entries
            for (int c = 0; c < a.pts.n; c++)
                if (a.plan.h_unique[c]) {
                    const int32_t v = a.pts.label[c][i], tv = L.t_tuple[ti][a.plan.h_comp[c]];
                    if (v) a.pts.tbl[c][v] = tv >> 1;
                }
            if (a.ipa.on)
                for (int kk = 0; kk < a.ipa.n_keys; kk++)
                    if (a.plan.k_unique[kk]) {
                        const int32_t v = a.ipa.label[kk][i];
                        const int q = a.plan.k_comp[kk];
                        if (v) {
                            a.ipa.aff[kk][v] = L.t_tuple[ti][q + 1], a.ipa.anti[kk][v] = L.t_tuple[ti][q + 2];
                            a.ipa.exist[kk][v] = L.t_tuple[ti][q + 3], a.ipa.score[kk][v] = L.t_tuple[ti][q + 4];
                        }
                    }
        }
        for (int c = 0; c < a.pts.n; c++)
            if (!a.plan.h_unique[c])
                for (int v = 1 + tid; v < a.plan.h_len[c]; v += kCwThreads) a.pts.tbl[c][v] = L.i32[a.plan.h_off[c] + v];
        for (int c = 0; c < a.soft.n; c++)
            if (!a.soft.is_hostname[c])
                for (int v = 1 + tid; v < a.plan.s_len[c]; v += kCwThreads) a.soft.tbl[c][v] = L.i32[a.plan.s_off[c] + v];
        if (a.ipa.on)
            for (int k = 0; k < a.ipa.n_keys; k++)
                if (!a.plan.k_unique[k]) {
                    const int len = a.plan.k_len[k];
                    for (int v = 1 + tid; v < len; v += kCwThreads) {
                        a.ipa.aff[k][v] = L.i64[a.plan.k_off[k] + v];
                        a.ipa.anti[k][v] = L.i64[a.plan.k_off[k] + len + v];
                        a.ipa.exist[k][v] = L.i64[a.plan.k_off[k] + 2 * len + v];
                        a.ipa.score[k][v] = L.i64[a.plan.k_off[k] + 3 * len + v];
                    }
                }
    }
    // ---- leave the class table empty for the next pass
    const int Cused = (int)a.w.ctl[kCwCtlClasses] < kCwMaxClasses ? (int)a.w.ctl[kCwCtlClasses] : kCwMaxClasses;
    __syncthreads();
    if (!giveup)
        for (int id = tid; id < Cused; id += kCwThreads) {
            const int g = a.w.slot_of_id[id];
            a.w.keys[g] = 0ull, a.w.ready[g] = 0u;
            CwClass &k = a.w.cls[g];
            k.nf = k.mt = k.ma = k.ht = k.ha = 0u;
        }
    __syncthreads();
    if (tid == 0 && !giveup) a.w.ctl[kCwCtlClasses] = 0u;
    if (tid == 0 && prof) {
        pf[6] += __builtin_amdgcn_s_memrealtime() - t_prev;
        for (int i = 0; i < 8; i++) a.w.prof[i] += pf[i];
    }
#undef CW_TICK
}

} // namespace ccsim
