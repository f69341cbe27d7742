// ccsim_sampled.h -- the SAMPLED SEARCH (percentageOfNodesToScore < 100: the reference's DEFAULT, adaptive 50 % ... 5 %) as a resident
// structure instead of node passes (round 5; SURVEY 8(a) row a4, 8(d) "mode B").
//
// Reference: findNodesThatFitPod / findNodesThatPassFilters / numFeasibleNodesToFind (S/schedule_one.go:482-564, 610-693, 697-723).  A
// cycle visits the nodes in ring order from nextStartNodeIndex, keeps the first K feasible ones, stops at the (K+1)-th, scores the K kept,
// picks the maximum (first in visiting order on ties), and moves the start index past what it visited (:538-539).  The
// sequential mode does that literally: a counting pass, a prefix, a scoring pass -- three dispatches over every node, 58 us per cycle at
// 1M nodes although a cycle changes ONE node.
//
// For a template without topology-coupled plugins a node's verdict and TotalScore depend on the node alone plus the two normalization
// maxima over the K kept nodes (TaintToleration / NodeAffinity, P/helper/normalize_score.go:28-56).  So what a cycle needs is resident:
//   memo[n]        TotalScore of every node under the ASSUMED maxima (mt_a, ma_a), -1 = infeasible (k_sb_build, once; one word per cycle after)
//   per BLOCK of 2^shift nodes (256 at 1M): feasible nodes, the best (score, lowest index) key, the maxima of the two raw scores over
//                  the feasible nodes -- 16 bytes per block, all blocks in the LDS of ONE workgroup for the whole run
// and a cycle is (k_sb_cycles, one persistent workgroup, no grid-wide anything):
//   1. ring prefix of the blocks' feasible counts from the start block -> the block that holds the (K+1)-th feasible node;
//   2. that block and the start block (both are cut by the stretch) are read node by node (memo + static word: 2 x 2 KB), every block
//      in between contributes its summary: argmax (ties: lowest ring position) and the maxima over exactly the K kept nodes;
//   3. maxima differ from the assumed ones -> rebuild under the true ones (as decide_commit does: "stale maxima: rescan"); else
//   4. NodeInfo.update on the winner (S/framework/types.go:409-428), its memo word, its block's summary, the start index.
// Three dependent trips to L2 per cycle instead of three passes over HBM.  Exactly the oracle's cycle: same nodes visited, same K
// kept, same winner, same start index (tests/test_sampling.py compares evaluated_total / last_feasible / the log cycle by cycle).
#pragma once
#include "ccsim_level.h"

#ifndef CCSIM_LAP_THREADS
#define CCSIM_LAP_THREADS 512 // the workgroup of k_sb_laps (a build-time A/B knob: 256, 512, 1024)
#endif

namespace ccsim {

constexpr int kSbThreads = 256;
constexpr int kSbWaves = kSbThreads / 64;
constexpr int kSbMaxBlocks = 8192; // block summaries resident in LDS (128 KiB)
constexpr int kSbMaxShift = 10;    // a block is at most 1024 nodes = four consecutive nodes per thread of the cycle kernel

struct SbArgs {
    DevCols c;
    DevPod p;
    DevState *st;
    int32_t *memo;              // [n_pad]
    uint8_t *flag8;             // [n_pad] k_sb_laps: a node's raw scores against the ASSUMED maxima (kLapOver | kLapHitT | kLapHitA), whether feasible or not
    uint32_t *sb_fc;            // [n_blocks] feasible nodes of the block
    unsigned long long *sb_key; // [n_blocks] make_key(best TotalScore, lowest index holding it), 0 = no feasible node
    uint32_t *sb_mx;            // [n_blocks] (max PreferNoSchedule count << 16) | max preferred-affinity sum, over the feasible nodes
    int32_t *log;
    int32_t shift, n_blocks, max_cycles;
    unsigned long long *prof; // k_sb_laps, measurement runs (CCSIM_SB_PROF=1): 10 ns ticks per phase, summed over the run; else nullptr
    int64_t slow_floor; // k_sb_laps: stretches whose maxima differ from the assumed ones are re-evaluated node by node while they cover <= max(this, N / 4) nodes
    int32_t handover;   // k_sb_laps: end the launch when fewer feasible nodes are left than the search keeps (every node is visited from then on,
                        // one cycle per lap): DevState::smp_phase = 2 tells the host to go on with k_sf_cycles (ccsim_search_full.h)
};

// one node under the assumed maxima: TotalScore, or -1 (the wide path: any snapshot; the narrow mirrors give the same number by construction)
__device__ __forceinline__ int32_t sb_node_score(const DevPod &p, const NodeRegs<kMaxExtra> &nd, uint32_t mt, uint32_t ma) {
    if (!node_feasible<kMaxExtra>(p, nd)) return -1;
    const uint32_t cnt = (nd.w >> kStatCntShift) & kStatCntMask, aff = nd.w & kStatAffMask, img = (nd.w >> kStatImgShift) & kStatImgMask;
    return (int32_t)node_score<kMaxExtra>(p, nd, static_score(p, cnt, aff, img, mt, ma));
}

// k_sb_build: memo + block summaries of the whole snapshot under (mt_a, ma_a).  One workgroup per block.  Runs while DevState::sb_dirty.
template <bool NARROW>
__global__ __launch_bounds__(256) void k_sb_build(SbArgs a) {
    const DevState &st = *a.st;
    if (st.done || !st.sb_dirty) return;
    const uint32_t mt = (uint32_t)st.mt_a, ma = (uint32_t)st.ma_a;
    const int tid = threadIdx.x, B = 1 << a.shift;
    const int64_t base = (int64_t)blockIdx.x << a.shift;
    const NarrowPod npod = narrow_pod(a.p, a.c.mem_shift);
    uint32_t fc = 0, bmt = 0, bma = 0;
    uint64_t best = 0;
    for (int j = tid; j < B; j += 256) {
        const int64_t i = base + j;
        if (i >= a.c.n_pad) break;
        int32_t sc = -1;
        const uint32_t w = a.c.stat[i];
        if (i < a.c.n) {
            if (NARROW) {
                const int32_t na0 = a.c.a32[0][i], na1 = a.c.a32[1][i], nr0 = a.c.r32[0][i], nr1 = a.c.r32[1][i];
                if ((w >> kStatOkBit) && fits_narrow(a.p, npod, na0, na1, nr0, nr1, a.c.alloc_pods[i], a.c.pod_count[i])) {
                    const uint32_t cnt = (w >> kStatCntShift) & kStatCntMask, aff = w & kStatAffMask, img = (w >> kStatImgShift) & kStatImgMask;
                    sc = (int32_t)(static_score(a.p, cnt, aff, img, mt, ma) + dynamic_score_narrow(a.p, npod, na0, na1, nr0, nr1, a.c.z32[0][i], a.c.z32[1][i]));
                }
            } else {
                NodeRegs<kMaxExtra> nd;
                load_one<kMaxExtra>(a.c, a.p, i, nd);
                sc = sb_node_score(a.p, nd, mt, ma);
            }
        }
        a.memo[i] = sc;
        if (a.flag8) {
            const uint32_t cnt = (w >> kStatCntShift) & kStatCntMask, aff = w & kStatAffMask;
            a.flag8[i] = (uint8_t)(((cnt > mt || aff > ma) ? 1u : 0u) | (cnt == mt ? 2u : 0u) | (aff == ma ? 4u : 0u));
        }
        if (sc >= 0) {
            const uint32_t cnt = (w >> kStatCntShift) & kStatCntMask, aff = w & kStatAffMask;
            fc += 1, bmt = cnt > bmt ? cnt : bmt, bma = aff > bma ? aff : bma;
            const uint64_t k = make_key((int64_t)sc, a.c.global_offset + i);
            best = k > best ? k : best;
        }
    }
    __shared__ uint32_t s_fc[4], s_mt[4], s_ma[4];
    __shared__ uint64_t s_k[4];
    fc = wave_sum_u32_dpp(fc), bmt = wave_max_u32(bmt), bma = wave_max_u32(bma), best = wave_max_u64(best);
    if ((tid & 63) == 0) s_fc[tid >> 6] = fc, s_mt[tid >> 6] = bmt, s_ma[tid >> 6] = bma, s_k[tid >> 6] = best;
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 4; w++) fc += s_fc[w], bmt = s_mt[w] > bmt ? s_mt[w] : bmt, bma = s_ma[w] > bma ? s_ma[w] : bma, best = s_k[w] > best ? s_k[w] : best;
        a.sb_fc[blockIdx.x] = fc, a.sb_key[blockIdx.x] = best, a.sb_mx[blockIdx.x] = (bmt << 16) | bma;
    }
}

// NodeInfo.update for one more clone on node i (S/framework/types.go:409-428; schedule_one.go:967-984 assume), and the node's memo word
// under the assumed maxima afterwards.  The caller has invalidated its L1 if the row may have been written before in this launch.
template <bool NARROW, class A> // (A: SbArgs, or ccsim_sampled_zone.h's SzArgs -- c, p, memo)
__device__ __forceinline__ int32_t sb_place(const A &a, const NarrowPod &npod, int64_t i, uint32_t mt_a, uint32_t ma_a) {
    constexpr int NX = NARROW ? 0 : kMaxExtra; // (the narrow mirrors exist for pods without extra resource columns only)
    NodeRegs<NX> nd;
    int32_t na0 = 0, na1 = 0;
    if (NARROW) na0 = a.c.a32[0][i], na1 = a.c.a32[1][i];
    load_one<NX>(a.c, a.p, i, nd);
    node_apply<NX>(a.p, nd, 1);
    store_dyn<NX>(a.c, a.p, i, nd, 1);
    int32_t nm = -1;
    if (NARROW) { // (the lossless mirrors: the same number as the wide path, ccsim_kernels.h "NARROW arithmetic")
        const int32_t nr0 = (int32_t)nd.r_cpu, nr1 = (int32_t)(nd.r_mem >> a.c.mem_shift), nz0 = (int32_t)nd.z_cpu, nz1 = (int32_t)(nd.z_mem >> a.c.mem_shift);
        nm = -1;
        if ((nd.w >> kStatOkBit) && fits_narrow(a.p, npod, na0, na1, nr0, nr1, nd.a_pods, nd.npods)) {
            const uint32_t cnt = (nd.w >> kStatCntShift) & kStatCntMask, aff = nd.w & kStatAffMask, img = (nd.w >> kStatImgShift) & kStatImgMask;
            nm = (int32_t)(static_score(a.p, cnt, aff, img, mt_a, ma_a) + dynamic_score_narrow(a.p, npod, na0, na1, nr0, nr1, nz0, nz1));
        }
    } else if constexpr (!NARROW)
        nm = sb_node_score(a.p, nd, mt_a, ma_a);
    __hip_atomic_store((uint32_t *)(a.memo + i), (uint32_t)nm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return nm;
}

struct SbLds {
    uint32_t fc[kSbMaxBlocks];
    unsigned long long key[kSbMaxBlocks];
    uint32_t mx[kSbMaxBlocks];
    // per-wave partials (one barrier between writing and reading each group; the groups alternate by use)
    uint32_t w_ct[kSbWaves], w_ch[kSbWaves], w_ls[kSbWaves], w_pf[kSbWaves], w_pmt[kSbWaves], w_pma[kSbWaves], w_ce[kSbWaves];
    unsigned long long w_pk[kSbWaves], w_key[kSbWaves];
    uint32_t w_mt[kSbWaves], w_ma[kSbWaves];
    long long w_stop[kSbWaves];
    // what one thread found and every thread needs
    int32_t cross_r, cross_need;
    int32_t nm, nm2;           // the new memo words of the last winner and of the one before (-1: it left the feasible nodes)
    long long nm_idx, nm2_idx; // ... and their indices.  (A memo store is certain to be in L2 only two barriers-5 later: the thread that
                               // made it fences at the start of its NEXT commit, off the path every other wave waits on.)
};

__device__ __forceinline__ int32_t ld_memo(const int32_t *p) { return (int32_t)__hip_atomic_load((const uint32_t *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// inclusive scan across the 64 lanes of a wave
__device__ __forceinline__ uint32_t sb_wave_incl(uint32_t v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = (uint32_t)__shfl_up((int)v, off);
        v += lane >= off ? o : 0u;
    }
    return v;
}

// k_sb_cycles: up to max_cycles scheduling cycles in ONE workgroup of 256 threads.  A block's nodes are dealt to the threads in
// runs of NP = block / 256 consecutive nodes (index order = thread order, then position in the run).  Five barriers and two
// dependent trips to L2 per cycle: the start block of the NEXT cycle and the winner's block are fetched while one thread applies
// the placement; the block the stretch ends in is the trip that cannot be known earlier.
template <bool NARROW>
__global__ __launch_bounds__(kSbThreads) void k_sb_cycles(SbArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sb_lds_raw[];
    SbLds &L = *reinterpret_cast<SbLds *>(sb_lds_raw);
    DevState &S = *a.st;
    if (S.done) return;
    // (DevState::sb_dirty: k_sb_build is enqueued in front of every launch of this kernel and has rebuilt memo and summaries if the flag
    // was set -- under the maxima this launch reads below; the flag is cleared at the end of this launch unless the maxima moved again)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6); // (wave: uniform, and known to the compiler as such)
    const int nb = a.n_blocks, sh = a.shift, B = 1 << sh, NP = B / kSbThreads; // NP in {1, 2, 4}
    const int64_t N = a.c.n;
    for (int b = tid; b < nb; b += kSbThreads) L.fc[b] = a.sb_fc[b], L.key[b] = a.sb_key[b], L.mx[b] = a.sb_mx[b];
    uint32_t ft = 0;
    for (int b = tid; b < nb; b += kSbThreads) ft += a.sb_fc[b];
    ft = wave_sum_u32_dpp(ft);
    if (lane == 0) L.w_ls[wave] = ft;
    if (tid == 0) L.nm_idx = -1, L.nm = -1, L.nm2_idx = -1, L.nm2 = -1;
    __syncthreads();
    int64_t Ftotal = 0;
    for (int w = 0; w < kSbWaves; w++) Ftotal += L.w_ls[w];
    __syncthreads();
    // the run state every thread carries (updated identically from broadcast values; thread 0 writes it back)
    const int64_t K = S.smp_K, limit = S.limit, log_cap = S.log_cap;
    const uint32_t mt_a = (uint32_t)S.mt_a, ma_a = (uint32_t)S.ma_a;
    const NarrowPod npod = narrow_pod(a.p, a.c.mem_shift);
    int64_t start = S.smp_start, placed = S.placed, rounds = S.rounds, scans = S.scans, evaluated = S.evaluated, winner = -1;
    int32_t last_feasible = S.last_feasible, last_evaluated = S.last_evaluated, done = 0, dirty = 0;
    uint32_t new_mt = mt_a, new_ma = ma_a;
    int64_t pend_blk = -1; // the block of the last winner: its summary is recomputed from the words fetched while the placement was applied
    const int E = (nb - 1 + kSbThreads - 1) / kSbThreads; // full blocks per thread in the ring scan
    constexpr int kNP = (1 << kSbMaxShift) / kSbThreads;
    int32_t ms[kNP], mp[kNP];
    uint32_t ws[kNP], wp[kNP];
    // (memo words are read past the CU's vector L1: a plain load may hit a line fetched BEFORE this workgroup's own thread 0 rewrote a word
    // of it some cycles ago -- measured: the first form of this kernel with plain loads placed differently near the end of whole runs,
    // where the same few blocks are read again and again)
    auto fetch = [&](int64_t blk, int32_t *m, uint32_t *w) {
#pragma unroll
        for (int j = 0; j < kNP; j++) {
            m[j] = -1, w[j] = 0;
            const int64_t i = (blk << sh) + (int64_t)tid * NP + j;
            if (j < NP && i < a.c.n_pad) m[j] = ld_memo(a.memo + i), w[j] = a.c.stat[i];
        }
    };
    fetch(start >> sh, ms, ws);
#pragma unroll
    for (int j = 0; j < kNP; j++) mp[j] = -1, wp[j] = 0;

    for (int cyc = 0; cyc < a.max_cycles && !done && !dirty; cyc++) {
        if (Ftotal == 0) { // schedule_one.go:448-454: every node was visited, none passed
            done = DONE_UNSCHEDULABLE, rounds += 1, scans += 1, last_feasible = 0, last_evaluated = (int32_t)N, evaluated += N, winner = -1;
            break;
        }
        const int sb = (int)(start >> sh);
        const int64_t i_s0 = ((int64_t)sb << sh) + (int64_t)tid * NP; // this thread's first node of the start block
        if (tid == 0) L.cross_r = -1;
        // ---- ring scan over the full blocks: entry r = 1 .. nb - 1 is block (sb + r) mod nb
        const int r_lo = tid * E + 1, r_hi = (tid + 1) * E < nb - 1 ? (tid + 1) * E : nb - 1;
        uint32_t ls = 0;
        for (int r = r_lo; r <= r_hi; r++) {
            int b = sb + r;
            b = b >= nb ? b - nb : b;
            ls += L.fc[b];
        }
        // feasible nodes of the start block behind / before the start index among this thread's run; the pending block's partial summary
        uint32_t ct = 0, ch = 0, pf = 0, pmt = 0, pma = 0;
        uint64_t pk = 0;
#pragma unroll
        for (int j = 0; j < kNP; j++) {
            if (ms[j] >= 0) (i_s0 + j >= start ? ct : ch) += 1u;
            if (mp[j] >= 0) {
                const uint32_t cnt = (wp[j] >> kStatCntShift) & kStatCntMask, aff = wp[j] & kStatAffMask;
                pf += 1, pmt = cnt > pmt ? cnt : pmt, pma = aff > pma ? aff : pma;
                const uint64_t k = make_key((int64_t)mp[j], (pend_blk << sh) + (int64_t)tid * NP + j);
                pk = k > pk ? k : pk;
            }
        }
        const uint32_t ict = sb_wave_incl(ct), ich = sb_wave_incl(ch), ils = sb_wave_incl(ls);
        if (pend_blk >= 0) pf = wave_sum_u32_dpp(pf), pmt = wave_max_u32(pmt), pma = wave_max_u32(pma), pk = wave_max_u64(pk);
        if (lane == 63) L.w_ct[wave] = ict, L.w_ch[wave] = ich, L.w_ls[wave] = ils;
        if (lane == 0 && pend_blk >= 0) L.w_pf[wave] = pf, L.w_pmt[wave] = pmt, L.w_pma[wave] = pma, L.w_pk[wave] = pk;
        __syncthreads(); // ---- barrier 1
        uint32_t tailF = 0, headF = 0, fullF = 0, bt = 0, bh = 0, bl = 0;
#pragma unroll
        for (int w = 0; w < kSbWaves; w++) {
            const uint32_t x = L.w_ct[w], y = L.w_ch[w], z = L.w_ls[w];
            tailF += x, headF += y, fullF += z;
            bt += w < wave ? x : 0u, bh += w < wave ? y : 0u, bl += w < wave ? z : 0u;
        }
        const uint32_t rk_tail0 = bt + ict - ct, rk_head0 = bh + ich - ch, ex = bl + ils - ls; // exclusive prefixes over the threads
        // the pending block's summary: every thread holds it (the walk below meets the block in some thread), thread 0 files it
        uint32_t P_fc = 0, P_mx = 0;
        uint64_t P_key = 0;
        if (pend_blk >= 0) {
            uint32_t qmt = 0, qma = 0;
#pragma unroll
            for (int w = 0; w < kSbWaves; w++) {
                P_fc += L.w_pf[w], qmt = L.w_pmt[w] > qmt ? L.w_pmt[w] : qmt, qma = L.w_pma[w] > qma ? L.w_pma[w] : qma;
                P_key = L.w_pk[w] > P_key ? L.w_pk[w] : P_key;
            }
            P_mx = (qmt << 16) | qma;
            if (tid == 0) {
                L.fc[pend_blk] = P_fc, L.key[pend_blk] = P_key, L.mx[pend_blk] = P_mx;
                a.sb_fc[pend_blk] = P_fc, a.sb_key[pend_blk] = P_key, a.sb_mx[pend_blk] = P_mx; // (the global copy stays current: the next launch reloads it)
            }
        }
        // ---- where does the stretch end?  (uniform: every thread holds the same totals)
        const bool all = Ftotal <= K; // fewer feasible nodes than wanted: the search visits every node (:538: processed = N)
        int mode = 0; // 0 all, 1 the (K+1)-th feasible node is behind the start index in the start block, 2 in a full block, 3 before the start index in the start block
        if (!all) mode = (int64_t)tailF >= K + 1 ? 1 : ((int64_t)tailF + fullF >= K + 1 ? 2 : 3);
        uint64_t best = 0;
        uint32_t cmt = 0, cma = 0;
        int64_t stop = -1;
        auto ringpos = [&](int64_t i) -> int64_t { return i >= start ? i - start : i + N - start; };
        auto take_node = [&](int32_t m, uint32_t w, int64_t i) {
            const uint64_t k = ((uint64_t)((int64_t)m + 1) << kIdxBits) | (kIdxMask - (uint64_t)ringpos(i));
            best = k > best ? k : best;
            const uint32_t cnt = (w >> kStatCntShift) & kStatCntMask, aff = w & kStatAffMask;
            cmt = cnt > cmt ? cnt : cmt, cma = aff > cma ? aff : cma;
        };
        // the start block's share
        {
            const int64_t need_head = K - ((int64_t)tailF + fullF); // (mode 3: how many of the nodes before the start index are among the first K)
            uint32_t rt = rk_tail0, rh = rk_head0;
#pragma unroll
            for (int j = 0; j < kNP; j++)
                if (ms[j] >= 0) {
                    const int64_t i = i_s0 + j;
                    if (i >= start) {
                        if (mode != 1 || (int64_t)rt < K) take_node(ms[j], ws[j], i);
                        else if ((int64_t)rt == K) stop = i;
                        rt += 1;
                    } else {
                        if (mode == 0 || (mode == 3 && (int64_t)rh < need_head)) take_node(ms[j], ws[j], i);
                        else if (mode == 3 && (int64_t)rh == need_head) stop = i;
                        rh += 1;
                    }
                }
        }
        // the full blocks' share: whole blocks inside the stretch by their summaries; the one the stretch ends in is found here
        if (mode != 1) {
            int64_t run = (int64_t)tailF + ex;
            uint64_t bk = 0; // the best summary key among this thread's blocks: its entries follow the ring, so on equal scores the first one stands
            for (int r = r_lo; r <= r_hi; r++) {
                int b = sb + r;
                b = b >= nb ? b - nb : b;
                const bool pend = b == pend_blk;
                const int64_t f = pend ? P_fc : L.fc[b];
                if (mode == 0 || mode == 3 || run + f <= K) {
                    const uint64_t k = pend ? P_key : (uint64_t)L.key[b];
                    if (k) {
                        bk = (k >> kIdxBits) > (bk >> kIdxBits) ? k : bk;
                        const uint32_t x = pend ? P_mx : L.mx[b];
                        cmt = (x >> 16) > cmt ? (x >> 16) : cmt, cma = (x & 0xffffu) > cma ? (x & 0xffffu) : cma;
                    }
                } else if (run <= K) { // run <= K < run + f: the (K+1)-th feasible node of the visiting order is in this block
                    L.cross_r = r, L.cross_need = (int32_t)(K - run);
                }
                run += f;
            }
            if (bk) {
                const uint64_t rk = ((uint64_t)(key_score(bk) + 1) << kIdxBits) | (kIdxMask - (uint64_t)ringpos(key_index(bk)));
                best = rk > best ? rk : best;
            }
        }
        pend_blk = -1;
        __syncthreads(); // ---- barrier 2
        if (mode == 2) { // the block the stretch ends in, node by node: the one trip to L2 nothing could have started earlier
            int b = sb + L.cross_r;
            b = b >= nb ? b - nb : b;
            int32_t me[kNP];
            uint32_t we[kNP];
            fetch(b, me, we);
            const int64_t i_e0 = ((int64_t)b << sh) + (int64_t)tid * NP;
            uint32_t ce = 0;
#pragma unroll
            for (int j = 0; j < kNP; j++) {
                if (j < NP && i_e0 + j == L.nm_idx) me[j] = L.nm; // (the store before that one was fenced before the last barrier)
                ce += me[j] >= 0 ? 1u : 0u;
            }
            const uint32_t ice = sb_wave_incl(ce);
            if (lane == 63) L.w_ce[wave] = ice;
            __syncthreads(); // ---- barrier 3
            uint32_t re = ice - ce;
#pragma unroll
            for (int w = 0; w < kSbWaves; w++) re += w < wave ? L.w_ce[w] : 0u;
            const int32_t need = L.cross_need;
#pragma unroll
            for (int j = 0; j < kNP; j++)
                if (me[j] >= 0) {
                    if ((int32_t)re < need) take_node(me[j], we[j], i_e0 + j);
                    else if ((int32_t)re == need) stop = i_e0 + j;
                    re += 1;
                }
        }
        // ---- the cycle's argmax, the maxima over the kept nodes, the node the search stopped at
        best = wave_max_u64(best), cmt = wave_max_u32(cmt), cma = wave_max_u32(cma);
        {
            const uint64_t sp = wave_max_u64((uint64_t)(stop + 1));
            if (lane == 0) L.w_key[wave] = best, L.w_mt[wave] = cmt, L.w_ma[wave] = cma, L.w_stop[wave] = (long long)sp;
        }
        __syncthreads(); // ---- barrier 4
        uint64_t sp1 = 0;
#pragma unroll
        for (int w = 0; w < kSbWaves; w++) {
            best = L.w_key[w] > best ? L.w_key[w] : best, cmt = L.w_mt[w] > cmt ? L.w_mt[w] : cmt, cma = L.w_ma[w] > cma ? L.w_ma[w] : cma;
            sp1 = (uint64_t)L.w_stop[w] > sp1 ? (uint64_t)L.w_stop[w] : sp1;
        }
        stop = (int64_t)sp1 - 1;
        scans += 1;
        if (cmt != mt_a || cma != ma_a) { // the scores were normalized with other maxima than the kept nodes': rebuild under the true ones
            new_mt = cmt, new_ma = cma, dirty = 1;
            break;
        }
        // ---- commit (schedule_one.go:967-984 assume -> NodeInfo.update) by one thread, while every thread fetches what the next cycle
        // starts with: its start block, and the winner's block for the summary
        const int64_t g_ring = (int64_t)(kIdxMask - (best & kIdxMask));
        int64_t g = start + g_ring;
        g = g >= N ? g - N : g;
        const int64_t visited = all ? N : ringpos(stop);
        if (!all) start = stop;
        pend_blk = g >> sh;
        fetch(start >> sh, ms, ws);
        fetch(pend_blk, mp, wp);
        if (tid == 0) {
            const int64_t i = g;
            // The node may have won before in THIS launch: its columns were rewritten by this very thread, and the CU's vector L1 does not
            // take a store's data -- a plain load would hit the line as it was fetched for the earlier placement and the update below
            // would be lost (measured: with the 6 KB a cycle of this kernel reads, such lines survive; placements went astray after a
            // node's second clone in one launch).  Acquire at agent scope = invalidate the L1 before the row is read.  The release in
            // front of it waits for the LAST cycle's stores (long done by now): from the coming barrier on, that cycle's memo word is in L2.
            __threadfence();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            const int32_t nm = sb_place<NARROW>(a, npod, i, mt_a, ma_a);
            L.nm2 = L.nm, L.nm2_idx = L.nm_idx;
            L.nm = nm, L.nm_idx = i;
            if (nm < 0) L.fc[pend_blk] -= 1; // (the ring scan of the next cycle reads it; the summary proper follows from the words fetched above)
            if (a.log && placed < log_cap) a.log[placed] = (int32_t)g;
        }
        __syncthreads(); // ---- barrier 5
        const int32_t nm = L.nm;
        {   // the words fetched above predate the placement: the winner's own is the one thread 0 has just computed
            const int64_t i_n0 = ((start >> sh) << sh) + (int64_t)tid * NP, i_p0 = (pend_blk << sh) + (int64_t)tid * NP;
            const int64_t g2 = L.nm2_idx; // (the winner before: its word may not have reached L2 when the fetches above were issued)
            const int32_t nm2 = L.nm2;
#pragma unroll
            for (int j = 0; j < kNP; j++) {
                if (j < NP && i_n0 + j == g2) ms[j] = nm2;
                if (j < NP && i_p0 + j == g2) mp[j] = nm2;
                if (j < NP && i_n0 + j == g) ms[j] = nm;
                if (j < NP && i_p0 + j == g) mp[j] = nm;
            }
        }
        if (nm < 0) Ftotal -= 1;
        placed += 1, rounds += 1, winner = g, evaluated += visited, last_evaluated = (int32_t)visited;
        last_feasible = (int32_t)(all ? Ftotal + (nm < 0 ? 1 : 0) : K);
        if (limit > 0 && placed >= limit) done = DONE_LIMIT; // simulator.go:297-312
    }
    // the last winner's block summary, if it is still pending
    if (pend_blk >= 0) {
        uint32_t pf = 0, pmt = 0, pma = 0;
        uint64_t pk = 0;
#pragma unroll
        for (int j = 0; j < kNP; j++)
            if (mp[j] >= 0) {
                const uint32_t cnt = (wp[j] >> kStatCntShift) & kStatCntMask, aff = wp[j] & kStatAffMask;
                pf += 1, pmt = cnt > pmt ? cnt : pmt, pma = aff > pma ? aff : pma;
                const uint64_t k = make_key((int64_t)mp[j], (pend_blk << sh) + (int64_t)tid * NP + j);
                pk = k > pk ? k : pk;
            }
        pf = wave_sum_u32_dpp(pf), pmt = wave_max_u32(pmt), pma = wave_max_u32(pma), pk = wave_max_u64(pk);
        __syncthreads();
        if (lane == 0) L.w_pf[wave] = pf, L.w_pmt[wave] = pmt, L.w_pma[wave] = pma, L.w_pk[wave] = pk;
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < kSbWaves; w++) pf += L.w_pf[w], pmt = L.w_pmt[w] > pmt ? L.w_pmt[w] : pmt, pma = L.w_pma[w] > pma ? L.w_pma[w] : pma, pk = L.w_pk[w] > pk ? L.w_pk[w] : pk;
            a.sb_fc[pend_blk] = pf, a.sb_key[pend_blk] = pk, a.sb_mx[pend_blk] = (pmt << 16) | pma;
        }
    }
    if (tid == 0) {
        S.smp_start = start, S.placed = placed, S.rounds = rounds, S.scans = scans, S.evaluated = evaluated, S.winner = winner;
        S.last_feasible = last_feasible, S.last_evaluated = last_evaluated, S.done = done;
        S.sb_dirty = dirty;
        if (dirty) S.mt_a = (int32_t)new_mt, S.ma_a = (int32_t)new_ma;
        S.sb_cycles += 1;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// k_sb_laps (round 6): the same search a LAP of the ring at a time.
//
// The next cycle starts exactly at the node this cycle stopped at -- its (K+1)-th feasible node (:538-539 with :655-662) -- the winner
// lies among the K kept nodes, and a placement changes that one node.  So with F feasible nodes at ring ranks 0 .. F-1 from the start
// index, cycle j keeps ranks [jK, (j+1)K) and stops at rank (j+1)K for every j < J = (F-1) / K: the J cycles of one lap of the ring
// read disjoint stretches that no earlier cycle of the lap has written, and their normalization maxima are over their own K nodes.
// They are evaluated side by side from the state at the start of the lap (tests/sampled_lap_model.py is this kernel in Python, checked
// against the oracle's visiting loop on the CPU).  The block summaries are the leaves of a binary tree in LDS -- per node the best
// (score, lowest index) key, the two maxima, the feasible count of its subtree -- so a lap costs O(J log blocks), not O(blocks):
//   1. the block holding ring rank jK, by descent on the counts (J lanes of one wave): it is CUT by a stretch boundary; every block
//      between two cuts lies wholly inside one stretch (K >= block size: one boundary per block at most);
//   2. the cut blocks, one wave each, node by node (the one dependent trip to L2 of the evaluation): the feasible nodes before the
//      boundary go to the stretch that ends there, the boundary node and those behind it to the next; the whole blocks of a stretch
//      by a range query on the tree (J lanes of another wave, meanwhile);
//   3. a stretch whose kept nodes' maxima differ from the assumed ones is re-evaluated node by node under its own (TotalScore is
//      static part + state part: the score under other maxima follows from the memo word and the static word) -- unless such
//      stretches cover more than a quarter of the ring: then the first of them ends the lap and everything is rebuilt under its maxima;
//   4. the J placements by J lanes of one wave (disjoint nodes), the winners' blocks re-read by the other waves meanwhile; then their
//      leaves, while wave 0 already finds the NEXT lap's cuts (the counts change only when a winner leaves the feasible nodes); the tree
//      above the leaves level by level in wave 1 while the others wait for the next lap's cut blocks -- three barriers per lap.
// F <= K is the degenerate lap of one stretch without a boundary (every node is visited, the start index stays).  One workgroup: a lap
// is ~40 KB of loads and two dependent trips to L2; what it needs from the rest of the chip is nothing, and a grid-wide barrier per
// lap would cost more than the lap.
constexpr int kLapThreads = CCSIM_LAP_THREADS, kLapWaves = kLapThreads / 64;
constexpr int kLapCuts = 32, kLapMaxJ = kLapCuts - 1;  // stretches per lap: cuts 0 .. J
constexpr int kLapSlots = kLapWaves - 1;               // waves that share a lap's cut blocks (all but the tree's), and its winners' blocks (all but the finder)
constexpr int kLapRounds = (kLapCuts + kLapSlots - 1) / kLapSlots; // cuts (and winners) per wave
constexpr int kLapMaxBlocks = 4096;                    // leaves of the tree (2 x 4096 nodes x 16 B = 128 KiB of LDS)
constexpr uint32_t kLapOver = 1u, kLapHitT = 2u, kLapHitA = 4u; // a stretch's maxima equal the assumed ones iff its flags are kLapHitT | kLapHitA
static_assert(kLapWaves >= 4, "k_sb_laps: wave 0 finds the next lap's cuts, wave 1 keeps the tree and answers the range queries, the last wave places the pods");

struct LapLds {
    // heap order: node p has children 2p, 2p + 1; leaf of block b = T[nbp + b].  x, y = key; z = (max PreferNoSchedule count << 16) |
    // max preferred-affinity sum over the subtree's feasible nodes; w = its feasible nodes
    uint4 T[2 * kLapMaxBlocks];
    // per stretch j: the best kept node, separately for the nodes at or behind the lap's start index [0] and before it [1] (the ring wraps
    // once: inside either part visiting order = index order, so the summaries' absolute keys compare as they are; part [0] comes first);
    // flags: a kept node above an assumed maximum / holding the assumed taint maximum / holding the assumed affinity maximum
    unsigned long long s_key[2][kLapCuts];
    uint32_t s_flag[kLapCuts];
    // per cut c: cut 0 is the start index, cut j the boundary between stretch j - 1 and stretch j (its node = where cycle j - 1 stopped)
    int32_t cut_blk[kLapCuts], cut_need[kLapCuts], cut_kind[kLapCuts]; // kind 0: a whole block, 1: the start block before the start index
    int32_t cut_node[kLapCuts];
    uint32_t cut_tail[kLapCuts]; // feasible nodes of the cut's block at or behind (index >=) its node, as of the lap's start
    int32_t nm[kLapCuts], g[kLapCuts]; // the winners and their memo words after the placements
    uint32_t w_mt[kLapWaves], w_ma[kLapWaves];
    unsigned long long w_key[kLapWaves];
    uint32_t bc[4];
};

// inclusive prefix sum across the 64 lanes: the DPP steps of wave_sum_u32_dpp, without the final broadcast
__device__ __forceinline__ uint32_t lap_wave_incl(uint32_t v) {
    CCSIM_DPP_STEP32(v, 0u, op_add_u32, 0x111, 0xf) CCSIM_DPP_STEP32(v, 0u, op_add_u32, 0x112, 0xf)
    CCSIM_DPP_STEP32(v, 0u, op_add_u32, 0x114, 0xf) CCSIM_DPP_STEP32(v, 0u, op_add_u32, 0x118, 0xf)
    CCSIM_DPP_STEP32(v, 0u, op_add_u32, 0x142, 0xa) CCSIM_DPP_STEP32(v, 0u, op_add_u32, 0x143, 0xc)
    return v;
}
// the greatest key among the lanes with `in` set, lanes ordered by index within one part of the ring: highest score, then the lowest lane
__device__ __forceinline__ unsigned long long lap_wave_best(bool in, unsigned long long k) {
    const uint32_t sc = in ? (uint32_t)(k >> kIdxBits) : 0u, top = wave_max_u32(sc);
    if (top == 0) return 0ull;
    const int l = __ffsll((long long)__ballot(in && sc == top)) - 1;
    return (unsigned long long)lane_bcast_i64((int64_t)k, l);
}
__device__ __forceinline__ uint32_t lap_flags(uint32_t cnt, uint32_t aff, uint32_t mt_a, uint32_t ma_a) {
    return ((cnt > mt_a || aff > ma_a) ? kLapOver : 0u) | (cnt == mt_a ? kLapHitT : 0u) | (aff == ma_a ? kLapHitA : 0u);
}
__device__ __forceinline__ uint32_t lap_wave_or3(bool in, uint32_t f) { // OR of 3-bit flags over the lanes with `in` set: three ballots
    return (__ballot(in && (f & 1u)) ? 1u : 0u) | (__ballot(in && (f & 2u)) ? 2u : 0u) | (__ballot(in && (f & 4u)) ? 4u : 0u);
}
__device__ __forceinline__ uint32_t lap_pkmax(uint32_t a, uint32_t b) { // the two 16-bit maxima at once
    const uint32_t hi = (a & 0xffff0000u) > (b & 0xffff0000u) ? (a & 0xffff0000u) : (b & 0xffff0000u), lo = (a & 0xffffu) > (b & 0xffffu) ? (a & 0xffffu) : (b & 0xffffu);
    return hi | lo;
}
__device__ __forceinline__ unsigned long long lap_key(const uint4 &n) { return ((unsigned long long)n.y << 32) | n.x; }
// lanes of ONE wave go on to read what other lanes of it have just written to LDS: a wave's LDS instructions execute in order, so all
// this has to stop is the compiler moving one across
__device__ __forceinline__ void lap_wave_sync() {
    __asm__ volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    __asm__ volatile("" ::: "memory");
}

template <bool NARROW, int NP> // NP nodes per lane of a cut block: blocks of 64 x NP nodes
__global__ __launch_bounds__(kLapThreads) void k_sb_laps(SbArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sb_lds_raw[];
    LapLds &L = *reinterpret_cast<LapLds *>(sb_lds_raw);
    DevState &S = *a.st;
    if (S.done || S.smp_phase == 2) return; // (2: handed over to the full search, see SbArgs::handover)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6); // (wave: known to be uniform, so what follows from it stays in scalar registers)
    const int nb = a.n_blocks, sh = a.shift; // (1 << sh == 64 * NP)
    int nbp = 2, levels = 1;
    while (nbp < nb) nbp <<= 1, levels++;
    const int32_t N = (int32_t)a.c.n;
    const uint32_t K = (uint32_t)S.smp_K;
    const int64_t limit = S.limit, log_cap = S.log_cap;
    const uint32_t mt_a = (uint32_t)S.mt_a, ma_a = (uint32_t)S.ma_a;
    const NarrowPod npod = narrow_pod(a.p, a.c.mem_shift);
    int32_t start = (int32_t)S.smp_start;
    int64_t placed = S.placed, rounds = S.rounds, scans = S.scans, evaluated = S.evaluated, winner = S.winner;
    int32_t last_feasible = S.last_feasible, last_evaluated = S.last_evaluated, done = 0, dirty = 0, laps = 0, slow = 0;
    uint32_t new_mt = mt_a, new_ma = ma_a;
    auto ringpos = [&](int32_t i) -> int32_t { return i >= start ? i - start : i + N - start; };
    // one block, NP consecutive nodes per lane.  Plain (vector) loads: every wave invalidates its L1 at the start of a lap, and what
    // the lap before wrote was released to L2 by the committing wave before the barrier in front of that
    auto fetch = [&](int blk, int32_t *m, uint32_t *w) {
        const int64_t i0 = ((int64_t)blk << sh) + (int64_t)lane * NP;
        if (NP == 4) {
            const int4 x = *reinterpret_cast<const int4 *>(a.memo + i0);
            const uint4 y = *reinterpret_cast<const uint4 *>(a.c.stat + i0);
            m[0] = x.x, m[1] = x.y, m[2] = x.z, m[3] = x.w, w[0] = y.x, w[1] = y.y, w[2] = y.z, w[3] = y.w;
        } else
            m[0] = a.memo[i0], w[0] = a.c.stat[i0];
    };
    // ---- the tree: leaves from the summaries k_sb_build (or the launch before) left, then level by level
    for (int b = tid; b < nbp; b += kLapThreads) {
        uint4 n = make_uint4(0u, 0u, 0u, 0u);
        if (b < nb) {
            const unsigned long long k = a.sb_key[b];
            n = make_uint4((uint32_t)k, (uint32_t)(k >> 32), a.sb_mx[b], a.sb_fc[b]);
        }
        L.T[nbp + b] = n;
    }
    for (int w = nbp >> 1; w >= 1; w >>= 1) {
        __syncthreads();
        for (int p = w + tid; p < 2 * w; p += kLapThreads) {
            const uint4 c0 = L.T[2 * p], c1 = L.T[2 * p + 1];
            const unsigned long long k0 = lap_key(c0), k1 = lap_key(c1), k = k0 > k1 ? k0 : k1;
            L.T[p] = make_uint4((uint32_t)k, (uint32_t)(k >> 32), lap_pkmax(c0.z, c1.z), c0.w + c1.w);
        }
    }
    if (wave == 0) { // the start block's feasible nodes at or behind the start index
        int32_t m[NP];
        uint32_t w[NP];
        fetch(start >> sh, m, w);
        uint32_t t = 0;
#pragma unroll
        for (int k = 0; k < NP; k++) t += (m[k] >= 0 && ((start >> sh) << sh) + lane * NP + k >= start) ? 1u : 0u;
        t = wave_sum_u32_dpp(t);
        if (lane == 0) L.bc[0] = t;
    }
    __syncthreads();
    if (tid == 0) { // feasible nodes before the start block: the counts of the left siblings on the way up
        uint32_t s = 0;
        for (int p = nbp + (start >> sh); p > 1; p >>= 1)
            if (p & 1) s += L.T[p - 1].w;
        L.bc[1] = s;
    }
    __syncthreads();
    uint32_t Ftotal = L.T[1].w, tailF = L.bc[0];
    uint32_t Pst = L.bc[1] + (L.T[nbp + (start >> sh)].w - tailF); // feasible nodes with an index below the start index
    int prev_Jc = 0;
    int64_t budget = a.max_cycles;
    const int64_t slow_cap = a.slow_floor > N / 4 ? a.slow_floor : N / 4;
    unsigned long long t_prev = a.prof ? __builtin_amdgcn_s_memrealtime() : 0ull, pf[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define LAP_TICK(i) do { if (a.prof) { const unsigned long long t_now = __builtin_amdgcn_s_memrealtime(); pf[i] += t_now - t_prev; t_prev = t_now; } } while (0)
    bool all = false;
    int J = 0, hand = 0;
    auto plan = [&]() { // the coming lap: how many stretches
        all = Ftotal <= K; // fewer feasible nodes than wanted: the search visits every node (:538: processed = N)
        if (all && a.handover && Ftotal > 0 && !done && !dirty) { // ... from here to the end of the run: the full search's kernel takes over
            hand = 1, J = 0;
            return;
        }
        J = all ? 1 : (int)((Ftotal - 1) / K < (uint32_t)kLapMaxJ ? (Ftotal - 1) / K : (uint32_t)kLapMaxJ);
        if (limit > 0 && limit - placed < J) J = (int)(limit - placed); // (the stretches behind the limit are never looked at)
        if (budget < J) J = (int)budget;
        if (Ftotal == 0 || done || dirty) J = 0;
    };
    // wave 0: the block of every ring rank jK of the coming lap, by descent on the counts
    auto find_cuts = [&]() {
        if (lane < kLapCuts) L.s_key[0][lane] = 0, L.s_key[1][lane] = 0, L.s_flag[lane] = 0, L.cut_blk[lane] = -1;
        lap_wave_sync();
        const int sb = start >> sh;
        if (!all && lane >= 1 && lane <= J) {
            uint32_t rem = Pst + (uint32_t)lane * K; // the rank among the feasible nodes in INDEX order
            rem = rem >= Ftotal ? rem - Ftotal : rem;
            int p = 1;
            while (p < nbp) {
                const uint32_t l = L.T[2 * p].w;
                p = 2 * p + (rem >= l ? 1 : 0), rem -= rem >= l ? l : 0u;
            }
            const int b = p - nbp;
            L.cut_blk[lane] = b, L.cut_need[lane] = (int32_t)rem, L.cut_kind[lane] = b == sb ? 1 : 0; // (in the start block: before the start index, K >= block size)
        }
        if (all && J == 1 && lane == 1 && L.T[nbp + sb].w > tailF) L.cut_blk[1] = sb, L.cut_need[1] = 0x7fffffff, L.cut_kind[1] = 1; // (all of it belongs to the one stretch)
    };
    plan();
    if (wave == 0) find_cuts();
    __syncthreads();
    // which of the waves' slots a cut block (every wave but the tree's) / a winner's block (every wave but the finder) falls to
    const int cslot = wave == 0 ? 0 : wave - 1, wslot = wave - 1;

    while (J > 0) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); // (see fetch)
        const int sb = start >> sh;
        // ---- 2. the cut blocks node by node, by every wave but wave 1 (all loads first, then the ranks -- stage by stage over a wave's
        // cuts, so that their dependent chains overlap).  Wave 1 meanwhile: keys and maxima above the last lap's winners, level by level,
        // then the stretches' range queries.
        if (wave == 1) {
            int p = lane < prev_Jc ? (nbp + (L.g[lane] >> sh)) >> 1 : 0;
            for (int lv = 0; lv < levels && prev_Jc > 0; lv++) {
                if (p >= 1) {
                    const uint4 c0 = L.T[2 * p], c1 = L.T[2 * p + 1];
                    const unsigned long long k0 = lap_key(c0), k1 = lap_key(c1), k = k0 > k1 ? k0 : k1;
                    L.T[p].x = (uint32_t)k, L.T[p].y = (uint32_t)(k >> 32), L.T[p].z = lap_pkmax(c0.z, c1.z); // (.w is wave 0's)
                }
                lap_wave_sync();
                p >>= 1;
            }
            LAP_TICK(4);
            if ((lane & 31) < J) {
                // the whole blocks of stretch (lane mod 32): ring positions strictly between its two cuts (position of block b: b behind
                // the start block, b + nb before it; the start block itself is position sb as cut 0 and sb + nb as a cut before the start
                // index).  Lanes 0 .. 31 ask for the part at or behind the start index, lanes 32 .. 63 for the part before it.
                const int j = lane & 31, part = lane >> 5;
                auto pos = [&](int c) -> int {
                    if (c == 0) return sb;
                    const int b = L.cut_blk[c];
                    return L.cut_kind[c] == 1 ? sb + nb : (b > sb ? b : b + nb);
                };
                const int pl = pos(j) + 1, pr = all ? sb + nb : pos(j + 1);
                int l = part == 0 ? pl : (pl > nb ? pl : nb) - nb, r = part == 0 ? (pr < nb ? pr : nb) : pr - nb;
                if (l < r) {
                    unsigned long long k = 0;
                    uint32_t mx = 0;
                    for (l += nbp, r += nbp; l < r; l >>= 1, r >>= 1) {
                        if (l & 1) {
                            const uint4 n = L.T[l++];
                            const unsigned long long kn = lap_key(n);
                            k = kn > k ? kn : k, mx = lap_pkmax(mx, n.z);
                        }
                        if (r & 1) {
                            const uint4 n = L.T[--r];
                            const unsigned long long kn = lap_key(n);
                            k = kn > k ? kn : k, mx = lap_pkmax(mx, n.z);
                        }
                    }
                    if (k) atomicMax(&L.s_key[part][j], k), atomicOr(&L.s_flag[j], lap_flags(mx >> 16, mx & 0xffffu, mt_a, ma_a));
                }
            }
            LAP_TICK(5);
        } else {
            int32_t cm[kLapRounds][NP];
            uint32_t cf[kLapRounds]; // the nodes' flag bytes
            int cb[kLapRounds];
#pragma unroll
            for (int rd = 0; rd < kLapRounds; rd++) {
                const int c = cslot + rd * kLapSlots;
                cb[rd] = c == 0 ? sb : (c <= J ? L.cut_blk[c] : -1);
                if (cb[rd] >= 0) {
                    const int64_t i0 = ((int64_t)cb[rd] << sh) + (int64_t)lane * NP;
                    if (NP == 4) {
                        const int4 x = *reinterpret_cast<const int4 *>(a.memo + i0);
                        cm[rd][0] = x.x, cm[rd][1] = x.y, cm[rd][2] = x.z, cm[rd][3] = x.w;
                        cf[rd] = *reinterpret_cast<const uint32_t *>(a.flag8 + i0);
                    } else
                        cm[rd][0] = a.memo[i0], cf[rd] = a.flag8[i0];
                }
            }
#pragma unroll
            for (int rd = 0; rd < kLapRounds; rd++) {
                const int c = cslot + rd * kLapSlots;
                if (cb[rd] < 0) continue; // (wave-uniform)
                const int kind = c == 0 ? 2 : L.cut_kind[c];
                const int32_t need = c == 0 ? 0 : L.cut_need[c];
                const int32_t blo = cb[rd] << sh, i0 = blo + lane * NP;
                const bool part1 = kind == 1 || (kind == 0 && cb[rd] < sb); // the nodes of this segment lie before the lap's start index
                // the segment: the whole block, its nodes before the start index (kind 1), or at and behind it (cut 0)
                const int32_t seg_lo = kind == 2 ? start : blo, seg_n = (kind == 1 ? start : blo + (1 << sh)) - seg_lo;
                uint32_t fm = 0, ca = 0; // this lane's feasible nodes inside the segment (bit k), feasible nodes at all
#pragma unroll
                for (int k = 0; k < NP; k++) {
                    const bool fe = cm[rd][k] >= 0;
                    ca += fe ? 1u : 0u;
                    fm |= (fe && (uint32_t)(i0 + k - seg_lo) < (uint32_t)seg_n) ? 1u << k : 0u;
                }
                const uint32_t cnt = (uint32_t)__popc(fm), incl = lap_wave_incl(cnt);
                // the first np of them come before the boundary (the ranks below `need`), the others at or behind it
                const int32_t d = need - (int32_t)(incl - cnt), np = d < 0 ? 0 : (d > (int32_t)cnt ? (int32_t)cnt : d);
                int32_t pm = -1, pi = 0, xm = -1, xi = 0, seen = 0, stop = -1;
                uint32_t pmask = 0, xmask = 0;
#pragma unroll
                for (int k = 0; k < NP; k++)
                    if (fm >> k & 1u) {
                        if (seen < np) {
                            if (cm[rd][k] > pm) pm = cm[rd][k], pi = i0 + k; // (strictly greater: the lowest index stands on equal scores)
                            pmask |= 0xffu << (8 * k);
                        } else {
                            if (seen == np) stop = i0 + k;
                            if (cm[rd][k] > xm) xm = cm[rd][k], xi = i0 + k;
                            xmask |= 0xffu << (8 * k);
                        }
                        seen += 1;
                    }
                uint32_t pfl = cf[rd] & pmask, xfl = cf[rd] & xmask; // OR of the side's flag bytes
                pfl |= pfl >> 16, pfl |= pfl >> 8, xfl |= xfl >> 16, xfl |= xfl >> 8;
                int32_t st = start;
                if (c > 0) {
                    const unsigned long long sm = __ballot(d >= 0 && d < (int32_t)cnt); // the one lane that holds the boundary node
                    st = sm ? lane_bcast_i32(stop, __ffsll((long long)sm) - 1) : -1;
                    const uint32_t top = wave_max_u32((uint32_t)(pm + 1));
                    if (top) {
                        const int l = __ffsll((long long)__ballot((uint32_t)(pm + 1) == top)) - 1;
                        const unsigned long long k = make_key((int64_t)top - 1, (int64_t)lane_bcast_i32(pi, l));
                        const uint32_t f = lap_wave_or3(pm >= 0, pfl);
                        if (lane == 0) atomicMax(&L.s_key[part1 ? 1 : 0][c - 1], k), atomicOr(&L.s_flag[c - 1], f);
                    }
                }
                if (c < J) {
                    const uint32_t top = wave_max_u32((uint32_t)(xm + 1));
                    if (top) {
                        const int l = __ffsll((long long)__ballot((uint32_t)(xm + 1) == top)) - 1;
                        const unsigned long long k = make_key((int64_t)top - 1, (int64_t)lane_bcast_i32(xi, l));
                        const uint32_t f = lap_wave_or3(xm >= 0, xfl);
                        if (lane == 0) atomicMax(&L.s_key[part1 ? 1 : 0][c], k), atomicOr(&L.s_flag[c], f);
                    }
                }
                // feasible nodes of the block at or behind the cut's node: the segment's part behind the boundary, plus (a cut before
                // the start index) the start block's nodes at or behind the start index
                const uint32_t seg_f = (uint32_t)lane_bcast_i32((int32_t)incl, 63);
                uint32_t t = c == 0 ? seg_f : (st >= 0 ? seg_f - (uint32_t)need : 0u);
                if (kind == 1) t += wave_sum_u32_dpp(ca) - seg_f;
                if (lane == 0) L.cut_node[c] = st, L.cut_tail[c] = t;
            }
        }
        __syncthreads(); // ---- barrier B: the stretches' keys and flags are complete
        LAP_TICK(0);
        // ---- 3. which stretches stand (every wave holds the lap's J stretches in its lanes 0 .. J - 1)
        const bool have = lane < J;
        unsigned long long key = 0;
        if (have) {
            const unsigned long long k0 = L.s_key[0][lane], k1 = L.s_key[1][lane];
            key = (k1 >> kIdxBits) > (k0 >> kIdxBits) ? k1 : k0; // (on equal scores the part at or behind the start index: it is visited first)
        }
        const int32_t c0 = have ? L.cut_node[lane] : 0, c1 = have && !all ? L.cut_node[lane + 1] : 0;
        const int32_t span = all ? N : ringpos(c1) - ringpos(c0);
        const bool mism = have && L.s_flag[lane] != (kLapHitT | kLapHitA);
        unsigned long long mm = __ballot(mism);
        int Jc = J;
        if (mm) {
            // the true maxima of a stretch, node by node (all waves on one stretch at a time)
            auto stretch_maxima = [&](int32_t first, int32_t sp, uint32_t &mt, uint32_t &ma) {
                uint32_t x = 0, y = 0;
                for (int32_t d = tid; d < sp; d += kLapThreads) {
                    int32_t i = first + d;
                    i = i >= N ? i - N : i;
                    if (a.memo[i] >= 0) {
                        const uint32_t w = a.c.stat[i], cnt = (w >> kStatCntShift) & kStatCntMask, aff = w & kStatAffMask;
                        x = cnt > x ? cnt : x, y = aff > y ? aff : y;
                    }
                }
                x = wave_max_u32(x), y = wave_max_u32(y);
                if (lane == 0) L.w_mt[wave] = x, L.w_ma[wave] = y;
                __syncthreads();
#pragma unroll
                for (int w = 0; w < kLapWaves; w++) x = L.w_mt[w] > x ? L.w_mt[w] : x, y = L.w_ma[w] > y ? L.w_ma[w] : y;
                __syncthreads();
                mt = x, ma = y;
            };
            const int32_t cover = (int32_t)wave_sum_u32_dpp(mism ? (uint32_t)span : 0u);
            if (cover > slow_cap) { // rebuild under the first such stretch's maxima; the stretches before it stand
                const int js = __ffsll((long long)mm) - 1;
                Jc = js, dirty = 1, scans += 1;
                stretch_maxima(lane_bcast_i32(c0, js), lane_bcast_i32(span, js), new_mt, new_ma);
            } else {
                while (mm) { // re-evaluated under its own maxima: TotalScore = memo word - static part under the assumed maxima + static part under its own
                    const int js = __ffsll((long long)mm) - 1;
                    mm &= mm - 1;
                    const int32_t first = lane_bcast_i32(c0, js), sp = lane_bcast_i32(span, js), rp0 = ringpos(first);
                    uint32_t mtj, maj;
                    stretch_maxima(first, sp, mtj, maj);
                    const uint32_t g_ta = div_magic(mt_a), g_aa = div_magic(ma_a), g_tj = div_magic(mtj), g_aj = div_magic(maj);
                    unsigned long long bk = 0; // a RING key here: the stretch may wrap
                    for (int32_t d = tid; d < sp; d += kLapThreads) {
                        int32_t i = first + d;
                        i = i >= N ? i - N : i;
                        const int32_t m = a.memo[i];
                        if (m >= 0) {
                            const uint32_t w = a.c.stat[i], cnt = (w >> kStatCntShift) & kStatCntMask, aff = w & kStatAffMask;
                            const int64_t sc = (int64_t)m - static_score(a.p, cnt, aff, 0u, mt_a, ma_a, g_ta, g_aa) + static_score(a.p, cnt, aff, 0u, mtj, maj, g_tj, g_aj);
                            const unsigned long long rk = ((unsigned long long)(sc + 1) << kIdxBits) | (kIdxMask - (unsigned long long)(rp0 + d));
                            bk = rk > bk ? rk : bk;
                        }
                    }
                    bk = wave_max_u64(bk);
                    if (lane == 0) L.w_key[wave] = bk;
                    __syncthreads();
#pragma unroll
                    for (int w = 0; w < kLapWaves; w++) bk = L.w_key[w] > bk ? L.w_key[w] : bk;
                    __syncthreads();
                    if (lane == js) { // back to an absolute key
                        int32_t gi = start + (int32_t)(kIdxMask - (bk & kIdxMask));
                        gi = gi >= N ? gi - N : gi;
                        key = make_key((int64_t)(bk >> kIdxBits) - 1, (int64_t)gi);
                    }
                    slow += 1;
                }
            }
        }
        // ---- 4. the placements (schedule_one.go:967-984 assume -> NodeInfo.update) by the lanes of the last wave, one node each; the
        // other waves fetch the winners' blocks meanwhile (winner j is wave (j mod waves)'s)
        const int32_t g = lane < Jc ? (int32_t)key_index(key) : -1;
        LAP_TICK(1);
        if (wave == kLapWaves - 1) {
            if (lane < Jc) {
                const int32_t nm = sb_place<NARROW>(a, npod, (int64_t)g, mt_a, ma_a);
                L.nm[lane] = nm, L.g[lane] = g;
                if (a.log && placed + lane < log_cap) a.log[placed + lane] = g;
            }
            __threadfence(); // the rows and memo words are in L2 before anyone passes the barrier below
            LAP_TICK(6);
        }
        int32_t pm[kLapRounds][NP];
        uint32_t pw[kLapRounds][NP];
        int pb[kLapRounds];
#pragma unroll
        for (int rd = 0; rd < kLapRounds; rd++) { // (winner j is wave 1 + (j mod (waves - 1))'s: wave 0 will be finding the next lap's cuts)
            const int jw = wslot + rd * kLapSlots;
            pb[rd] = -1;
            if (wave > 0 && jw < Jc) pb[rd] = lane_bcast_i32(g, jw) >> sh, fetch(pb[rd], pm[rd], pw[rd]);
            else {
#pragma unroll
                for (int k = 0; k < NP; k++) pm[rd][k] = -1, pw[rd][k] = 0;
            }
        }
        __syncthreads(); // ---- barrier C: the winners' new memo words are in LDS (and in L2)
        LAP_TICK(2);
        // ---- the run state, identically in every thread
        {
            const int32_t ns = all ? start : L.cut_node[Jc]; // where the last committed cycle stopped
            const uint32_t nt = L.cut_tail[all ? 0 : Jc];
            const bool left = lane < Jc && L.nm[lane] < 0; // winners that are not feasible any more
            const bool in_tail = left && (L.g[lane] >> sh) == (ns >> sh) && L.g[lane] >= ns;
            const int n_left = __popcll(__ballot(left)), n_tail = __popcll(__ballot(in_tail)), n_below = __popcll(__ballot(left && L.g[lane] < ns));
            if (Jc > 0) {
                evaluated += all ? N : ringpos(ns);
                last_evaluated = all ? N : ringpos(ns) - ringpos(L.cut_node[Jc - 1]);
                last_feasible = (int32_t)(all ? Ftotal : K);
                winner = lane_bcast_i32(g, Jc - 1);
            }
            if (!all) { // the new start index is ring rank Jc x K of this lap: its rank in index order
                Pst += (uint32_t)Jc * K;
                Pst = Pst >= Ftotal ? Pst - Ftotal : Pst;
            }
            Pst -= (uint32_t)n_below;
            placed += Jc, rounds += Jc, scans += Jc, budget -= Jc > 0 ? Jc : 1;
            Ftotal -= (uint32_t)n_left;
            tailF = nt - (uint32_t)n_tail;
            start = ns;
            prev_Jc = Jc;
            laps += 1;
            if (limit > 0 && placed >= limit) done = DONE_LIMIT; // simulator.go:297-312
            if (budget <= 0 && !done) J = 0;
            else plan();
        }
        if (wave == 0) {
            // ---- the counts above (and in) the leaves of the winners that left the feasible nodes, then the next lap's cuts
            if (lane < prev_Jc && L.nm[lane] < 0)
                for (int p = nbp + (L.g[lane] >> sh); p >= 1; p >>= 1) atomicSub(&L.T[p].w, 1u);
            if (J > 0) find_cuts();
            LAP_TICK(7);
        } else {
            // ---- the winners' leaves (keys and maxima; the counts are wave 0's).  The words fetched above may predate the placements:
            // the winners' own come from LDS.  Two winners at most share a block, and they are neighbours (a stretch holds K >= block
            // size feasible nodes).  The maxima change only when a node left the feasible ones.
            const int32_t nmv = lane < prev_Jc ? L.nm[lane] : 0;
#pragma unroll
            for (int rd = 0; rd < kLapRounds; rd++) {
                const int jw = wslot + rd * kLapSlots;
                if (pb[rd] < 0) continue; // (wave-uniform)
                const int32_t i0 = (pb[rd] << sh) + lane * NP;
                bool gone = false;
                for (int x = jw - 1; x <= jw + 1; x++)
                    if (x >= 0 && x < prev_Jc) {
                        const int32_t gx = lane_bcast_i32(g, x);
                        if ((gx >> sh) != pb[rd]) continue;
                        const int32_t nx = lane_bcast_i32(nmv, x);
                        gone = gone || nx < 0;
#pragma unroll
                        for (int k = 0; k < NP; k++)
                            if (i0 + k == gx) pm[rd][k] = nx;
                    }
                int32_t bm = -1, bi = 0;
#pragma unroll
                for (int k = 0; k < NP; k++)
                    if (pm[rd][k] > bm) bm = pm[rd][k], bi = i0 + k;
                const uint32_t top = wave_max_u32((uint32_t)(bm + 1));
                unsigned long long k = 0;
                if (top) k = make_key((int64_t)top - 1, (int64_t)lane_bcast_i32(bi, __ffsll((long long)__ballot((uint32_t)(bm + 1) == top)) - 1));
                if (lane == 0) L.T[nbp + pb[rd]].x = (uint32_t)k, L.T[nbp + pb[rd]].y = (uint32_t)(k >> 32); // (the tree above it: wave 1, in the next lap)
                if (gone) {
                    uint32_t x = 0, y = 0;
#pragma unroll
                    for (int k2 = 0; k2 < NP; k2++)
                        if (pm[rd][k2] >= 0) {
                            const uint32_t cnt = (pw[rd][k2] >> kStatCntShift) & kStatCntMask, aff = pw[rd][k2] & kStatAffMask;
                            x = cnt > x ? cnt : x, y = aff > y ? aff : y;
                        }
                    x = wave_max_u32(x), y = wave_max_u32(y);
                    if (lane == 0) L.T[nbp + pb[rd]].z = (x << 16) | y;
                }
            }
        }
        __syncthreads(); // ---- barrier D: the winners' leaves are written, the next lap's cuts are known
        LAP_TICK(3);
    }
    if (!done && !dirty && Ftotal == 0 && budget > 0) // schedule_one.go:448-454: every node was visited, none passed
        done = DONE_UNSCHEDULABLE, rounds += 1, scans += 1, last_feasible = 0, last_evaluated = N, evaluated += N, winner = -1;
#undef LAP_TICK
    if (!dirty) // the summaries the next launch reloads (after a rebuild request k_sb_build writes them all)
        for (int b = tid; b < nb; b += kLapThreads) {
            const uint4 n = L.T[nbp + b];
            a.sb_fc[b] = n.w, a.sb_key[b] = lap_key(n), a.sb_mx[b] = n.z;
        }
    if (a.prof) {
        if (tid == 0)
            for (int i = 0; i < 4; i++) a.prof[i] += pf[i];
        if (tid == 64) a.prof[4] += pf[4], a.prof[5] += pf[5]; // (wave 1: the tree above the last winners, the range queries: both inside [0])
        if (tid == kLapThreads - 64) a.prof[6] += pf[6];       // (the committing wave's own time: rows, memo words, fence: inside [2])
        if (tid == 0) a.prof[7] += pf[7];                       // (wave 0: the counts, the next lap's cuts; the rest of [3] is waiting for the leaves)
    }
    if (tid == 0) {
        S.smp_start = start, S.placed = placed, S.rounds = rounds, S.scans = scans, S.evaluated = evaluated, S.winner = winner;
        S.last_feasible = last_feasible, S.last_evaluated = last_evaluated, S.done = done;
        S.sb_dirty = dirty;
        if (dirty) S.mt_a = (int32_t)new_mt, S.ma_a = (int32_t)new_ma;
        S.sb_cycles += 1, S.sb_laps += laps, S.sb_slow += slow;
        if (hand) S.smp_phase = 2;
    }
}

} // namespace ccsim
