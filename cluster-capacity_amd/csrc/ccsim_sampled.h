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
    uint32_t *sb_fc;            // [n_blocks] feasible nodes of the block
    unsigned long long *sb_key; // [n_blocks] make_key(best TotalScore, lowest index holding it), 0 = no feasible node
    uint32_t *sb_mx;            // [n_blocks] (max PreferNoSchedule count << 16) | max preferred-affinity sum, over the feasible nodes
    int32_t *log;
    int32_t shift, n_blocks, max_cycles;
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
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
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
            NodeRegs<kMaxExtra> nd;
            int32_t na0 = 0, na1 = 0;
            // The node may have won before in THIS launch: its columns were rewritten by this very thread, and the CU's vector L1 does not
            // take a store's data -- a plain load would hit the line as it was fetched for the earlier placement and the update below
            // would be lost (measured: with the 6 KB a cycle of this kernel reads, such lines survive; placements went astray after a
            // node's second clone in one launch).  Acquire at agent scope = invalidate the L1 before the row is read.  The release in
            // front of it waits for the LAST cycle's stores (long done by now): from the coming barrier on, that cycle's memo word is in L2.
            __threadfence();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            if (NARROW) na0 = a.c.a32[0][i], na1 = a.c.a32[1][i];
            load_one<kMaxExtra>(a.c, a.p, i, nd);
            node_apply<kMaxExtra>(a.p, nd, 1);
            store_dyn<kMaxExtra>(a.c, a.p, i, nd, 1);
            int32_t nm;
            if (NARROW) { // (the lossless mirrors: the same number as the wide path, ccsim_kernels.h "NARROW arithmetic")
                const int32_t nr0 = (int32_t)nd.r_cpu, nr1 = (int32_t)(nd.r_mem >> a.c.mem_shift), nz0 = (int32_t)nd.z_cpu, nz1 = (int32_t)(nd.z_mem >> a.c.mem_shift);
                nm = -1;
                if ((nd.w >> kStatOkBit) && fits_narrow(a.p, npod, na0, na1, nr0, nr1, nd.a_pods, nd.npods)) {
                    const uint32_t cnt = (nd.w >> kStatCntShift) & kStatCntMask, aff = nd.w & kStatAffMask, img = (nd.w >> kStatImgShift) & kStatImgMask;
                    nm = (int32_t)(static_score(a.p, cnt, aff, img, mt_a, ma_a) + dynamic_score_narrow(a.p, npod, na0, na1, nr0, nr1, nz0, nz1));
                }
            } else
                nm = sb_node_score(a.p, nd, mt_a, ma_a);
            __hip_atomic_store((uint32_t *)(a.memo + i), (uint32_t)nm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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

} // namespace ccsim
