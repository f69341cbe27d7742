// ccsim_search_full.h -- the FULL search (percentageOfNodesToScore = 100: every node filtered and scored each cycle) of one template
// without topology-coupled plugins on the resident block summaries of ccsim_sampled.h (round 6; SURVEY 8(a) rows a2 / a4, the B2 seam
// S/scheduler.go:88-91 a Go host calls once per pod).
//
// Reference: schedulePod (S/schedule_one.go:430-478): findNodesThatFitPod over all nodes (numFeasibleNodesToFind = N, :697-723), the
// scores of every feasible node, selectHost (first maximum in visiting order = index order: all N nodes are processed, so
// nextStartNodeIndex returns to where it was, :538-539), assume (:967-984).  The sequential mode does that as one pass over every node
// per cycle (k_scan_fused: 36 MB at 1M nodes, 13 us) although a cycle changes ONE node.
//
// What a cycle needs is resident after k_sb_build: memo[n] = TotalScore of node n under the assumed normalization maxima (-1 =
// infeasible), and per block of 2^shift nodes the best (score, lowest index) key, the feasible count and the two raw-score maxima over
// the feasible nodes.  With every node kept, the maxima over the kept nodes are the maxima over ALL feasible nodes = the maximum of the
// blocks' maxima; the winner is the maximum of the blocks' keys.  One WAVE runs the cycles (no barrier anywhere in the loop):
//   the 4096 block keys live in LDS, the maxima of 64 GROUPS of 64 blocks in the wave's registers (lane = group);
//   cycle: winner g = the greatest group key -> its node row and its block's 2^shift memo words are fetched (ONE trip to L2: the only
//   dependent one) -- meanwhile the best key among the OTHER blocks of its group and among the OTHER groups are reduced -- the row takes
//   the clone (NodeInfo.update), the node's new memo word replaces the fetched one, the block's new key is one wave reduction, and the
//   next winner is max(that, the two maxima computed in the shadow of the trip).
// A node that stops being feasible may have held a maximum: the block's, the group's and the global maxima are recomputed (rare); if the
// global ones differ from the assumed ones the launch ends and k_sb_build runs again under the true ones ("stale maxima: rescan").
#pragma once
#include "ccsim_sampled.h"

namespace ccsim {

constexpr int kSfThreads = 256, kSfWaves = kSfThreads / 64;
constexpr int kSfMaxBlocks = 4096, kSfGroups = kSfMaxBlocks / 64;

struct SfLds {
    unsigned long long key[kSfMaxBlocks];
    uint32_t mx[kSfMaxBlocks];
    unsigned long long g_key[kSfGroups];
    uint32_t g_mx[kSfGroups];
    uint32_t w_fc[kSfWaves];
};

__device__ __forceinline__ uint32_t sf_wave_pkmax(uint32_t v) { // the two 16-bit maxima of a packed word, across the wave
    return (wave_max_u32(v >> 16) << 16) | wave_max_u32(v & 0xffffu);
}

template <bool NARROW, int NP> // NP memo words per lane of a block: blocks of 64 x NP nodes, lane l holds nodes l, l + 64, ...
__global__ __launch_bounds__(kSfThreads) void k_sf_cycles(SbArgs a) {
    __shared__ SfLds L;
    DevState &S = *a.st;
    if (S.done) return; // (sb_dirty: k_sb_build in front of this launch has just rebuilt memo and summaries under S.mt_a / S.ma_a; the flag is cleared below)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nb = a.n_blocks, sh = a.shift; // (1 << sh == 64 * NP)
    const int32_t N = (int32_t)a.c.n;
    // The sampled search hands over to this kernel once fewer feasible nodes are left than it wants to keep (every node is visited from then
    // on, schedule_one.go:538, and nextStartNodeIndex stays): the visiting order -- the tie-break -- starts at that index S0, so the keys
    // in here carry RING positions behind the score; the summaries in memory carry indices (k_sb_build, k_sb_laps).
    const int32_t S0 = S.smp_K > 0 ? (int32_t)S.smp_start : 0;
    auto rpos = [&](int32_t i) -> int32_t { return i >= S0 ? i - S0 : i + N - S0; };
    auto node_at = [&](int32_t rp) -> int32_t { return rp + S0 >= N ? rp + S0 - N : rp + S0; };
    auto to_ring = [&](unsigned long long k) -> unsigned long long { return k ? (k & ~kIdxMask) | (kIdxMask - (unsigned long long)rpos((int32_t)key_index(k))) : 0ull; };
    auto to_index = [&](unsigned long long k) -> unsigned long long { return k ? (k & ~kIdxMask) | (kIdxMask - (unsigned long long)node_at((int32_t)key_index(k))) : 0ull; };
    // ---- the block summaries the build (or the launch before) left; the groups' maxima
    uint32_t fc = 0;
    for (int b = tid; b < kSfMaxBlocks; b += kSfThreads) {
        const bool in = b < nb;
        L.key[b] = in ? to_ring(a.sb_key[b]) : 0ull, L.mx[b] = in ? a.sb_mx[b] : 0u;
        fc += in ? a.sb_fc[b] : 0u;
    }
    fc = wave_sum_u32_dpp(fc);
    if (lane == 0) L.w_fc[wave] = fc;
    __syncthreads();
    if (S0 > 0 && wave == 0) { // the block the ring starts in: its best node in RING order, from its words (the summary's is the lowest index)
        const int sb0 = S0 >> sh;
        unsigned long long best = 0;
#pragma unroll
        for (int k = 0; k < NP; k++) {
            const int64_t i = ((int64_t)sb0 << sh) + k * 64 + lane;
            const int32_t m = (NP == 4 || NP == 1 || i < a.c.n_pad) ? ld_memo(a.memo + i) : -1;
            const unsigned long long kk = m >= 0 ? make_key((int64_t)m, (int64_t)rpos((int32_t)i)) : 0ull;
            best = kk > best ? kk : best;
        }
        best = wave_max_u64(best);
        if (lane == 0) L.key[sb0] = best;
    }
    __syncthreads();
    const int ngroups = (nb + 63) >> 6;
    for (int gq = wave; gq < kSfGroups; gq += kSfWaves) {
        unsigned long long k = 0;
        uint32_t m = 0;
        if (gq < ngroups) k = wave_max_u64(L.key[gq * 64 + lane]), m = sf_wave_pkmax(L.mx[gq * 64 + lane]);
        if (lane == 0) L.g_key[gq] = k, L.g_mx[gq] = m;
    }
    __syncthreads();
    if (wave != 0) return;

    const int64_t n_pad = a.c.n_pad, limit = S.limit, log_cap = S.log_cap;
    const uint32_t mt_a = (uint32_t)S.mt_a, ma_a = (uint32_t)S.ma_a;
    const NarrowPod npod = narrow_pod(a.p, a.c.mem_shift);
    int64_t placed = S.placed, rounds = S.rounds, scans = S.scans, evaluated = S.evaluated, winner = S.winner;
    int32_t last_feasible = S.last_feasible, last_evaluated = S.last_evaluated, done = 0, dirty = 0, cycles = 0;
    uint32_t Ftotal = L.w_fc[0] + L.w_fc[1] + L.w_fc[2] + L.w_fc[3];
    unsigned long long gk = L.g_key[lane];
    uint32_t gm = L.g_mx[lane];
    uint32_t root_mx = sf_wave_pkmax(gm);
    unsigned long long top = wave_max_u64(gk);
    int64_t budget = a.max_cycles;
    constexpr int NX = NARROW ? 0 : kMaxExtra;
    // What the wave holds between cycles, NOT yet in memory: the node that won last (pg), its row after its clones (nd, pc) and its memo
    // word (cur_m) -- the emptiest node of a cluster wins again and again until its score has come down to the next one's, and costs no
    // trip while it does; its block's memo words (bm: pg's own slot is stale until pg is evicted); the best key among the block's other
    // nodes (ob), the group's other blocks (og), the other groups (orr) -- they stand while the winner stays where it is.
    int32_t bm[NP], pg = -1, pb = -1, na0 = 0, na1 = 0, pc = 0, cur_m = -1;
    NodeRegs<NX> nd;
    nd_zero(nd);
    unsigned long long ob = 0, og = 0, orr = 0, cur_key = 0, rest = top;
#pragma unroll
    for (int k = 0; k < NP; k++) bm[k] = -1;
    auto store_row = [&](int32_t gx, const NodeRegs<NX> &n, int32_t cnt, int32_t m) { // a node that was held: its row and memo word to memory
        if (gx < 0 || lane != 0) return;
        a.c.req[0][gx] = n.r_cpu, a.c.req[1][gx] = n.r_mem, a.c.nz_mcpu[gx] = n.z_cpu, a.c.nz_mem[gx] = n.z_mem, a.c.pod_count[gx] = n.npods;
        a.c.placed_cnt[gx] = cnt;
        store_mirror(a.c, (int64_t)gx, n.r_cpu, n.r_mem, n.z_cpu, n.z_mem);
        if (NX > 0) {
#pragma unroll
            for (int x = 0; x < NX; x++)
                if (x < a.p.nx) a.c.req[a.p.xcol[x]][gx] = n.xr[x];
        }
        __hip_atomic_store((uint32_t *)(a.memo + gx), (uint32_t)m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    int32_t ck_b = -1;
    unsigned long long ck_leaf = 0;
    auto close_block = [&]() { // the held block's key to LDS (to memory: with the row's stores), its group's key
        if (pb < 0) return;
        const unsigned long long leaf = cur_key > ob ? cur_key : ob, gnew = leaf > og ? leaf : og;
        if (lane == 0) L.key[pb] = leaf;
        ck_b = pb, ck_leaf = leaf;
        gk = lane == (pb >> 6) ? gnew : gk;
        lap_wave_sync();
    };
    unsigned long long t_prev = a.prof ? __builtin_amdgcn_s_memrealtime() : 0ull, pf[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define SF_TICK(i) do { if (a.prof) { const unsigned long long t_now = __builtin_amdgcn_s_memrealtime(); pf[i] += t_now - t_prev; t_prev = t_now; } } while (0)

    for (;;) {
        if (limit > 0 && placed >= limit) { // simulator.go:297-312
            done = DONE_LIMIT;
            break;
        }
        if (budget <= 0) break;
        if (top == 0) { // schedule_one.go:448-454: every node was visited, none passed
            done = DONE_UNSCHEDULABLE, rounds += 1, scans += 1, last_feasible = 0, last_evaluated = N, evaluated += N, winner = -1;
            break;
        }
        if (root_mx != ((mt_a << 16) | ma_a)) { // the memo words stand under other maxima than the feasible nodes hold now
            dirty = 1, scans += 1;
            break;
        }
        const int32_t g = node_at((int32_t)key_index(top));
        if (g != pg) {
            // ---- another node: the held one goes to memory; ONE trip for the winner's row (the same address in every lane, so one request:
            // every lane then holds the row and computes the same scores) and, if it lies in another block, that block's memo words
            // (L1-bypassing: the launch has written some of them)
            const int32_t b = g >> sh, grp = b >> 6;
            const bool other_block = b != pb;
            if (other_block) close_block();
#pragma unroll
            for (int k = 0; k < NP; k++) bm[k] = (pb << sh) + k * 64 + lane == pg ? cur_m : bm[k]; // the held node's slot of its block's words
            const NodeRegs<NX> od = nd; // ... and its row aside: stored BEHIND the trip below (see there)
            const int32_t opg = pg, opc = pc, cur_m_prev = cur_m;
            if (other_block) {
#pragma unroll
                for (int k = 0; k < NP; k++) {
                    const int64_t i = ((int64_t)b << sh) + k * 64 + lane;
                    bm[k] = (NP <= 4 || i < n_pad) ? ld_memo(a.memo + i) : -1; // (n_pad is a multiple of 512: blocks of 64 or 256 nodes never reach beyond it)
                }
            }
            if (NARROW) na0 = a.c.a32[0][g], na1 = a.c.a32[1][g];
            load_one<NX>(a.c, a.p, (int64_t)g, nd);
            pc = a.c.placed_cnt[g];
            if (other_block) { // in the trip's shadow: the best key of the group's other blocks, of the other groups
                const unsigned long long lv = L.key[grp * 64 + lane];
                og = wave_max_u64(lane == (b & 63) ? 0ull : lv), orr = wave_max_u64(lane == grp ? 0ull : gk);
            }
            SF_TICK(0);
            unsigned long long best = 0;
#pragma unroll
            for (int k = 0; k < NP; k++) {
                const int32_t i = (b << sh) + k * 64 + lane;
                const unsigned long long kk = (bm[k] >= 0 && i != g) ? make_key((int64_t)bm[k], (int64_t)rpos(i)) : 0ull;
                best = kk > best ? kk : best;
            }
            ob = wave_max_u64(best);
            rest = ob > og ? ob : og;
            rest = orr > rest ? orr : rest;
            pf[4] += 1, pf[5] += other_block ? 1 : 0;
            pg = g, pb = b;
            SF_TICK(1);
            // ---- the node that was held and its block's key go to memory now: memory operations retire in the order they were issued, so
            // stores in front of the loads above would have put their acknowledgements into the one trip the cycle waits for; behind them
            // nobody waits for these (the next loads follow a streak's evaluation later)
            __builtin_amdgcn_sched_barrier(0);
            store_row(opg, od, opc, cur_m_prev);
            if (ck_b >= 0 && lane == 0) a.sb_key[ck_b] = to_index(ck_leaf);
            ck_b = -1;
            SF_TICK(2);
        }
        // ---- NodeInfo.update (S/framework/types.go:409-428) and the node's score afterwards -- for the next 64 clones at once: lane j
        // holds the node after j + 1 more clones.  The node wins the next cycle too while its key stays above everything else's (`rest`
        // does not move meanwhile: nothing but this node changes), so the first lane whose key falls below it ends the streak: its state
        // is the one the node is left in.  (Every state is evaluated with the same exact functions a cycle at a time would use.)
        int64_t cap = budget;
        if (limit > 0 && limit - placed < cap) cap = limit - placed;
        NodeRegs<NX> nj = nd;
        node_apply<NX>(a.p, nj, (int64_t)lane + 1);
        int32_t mj = -1;
        if (NARROW) {
            const int32_t nr0 = (int32_t)nj.r_cpu, nr1 = (int32_t)(nj.r_mem >> a.c.mem_shift), nz0 = (int32_t)nj.z_cpu, nz1 = (int32_t)(nj.z_mem >> a.c.mem_shift);
            if ((nj.w >> kStatOkBit) && fits_narrow(a.p, npod, na0, na1, nr0, nr1, nj.a_pods, nj.npods)) {
                const uint32_t cnt = (nj.w >> kStatCntShift) & kStatCntMask, aff = nj.w & kStatAffMask, img = (nj.w >> kStatImgShift) & kStatImgMask;
                mj = (int32_t)(static_score(a.p, cnt, aff, img, mt_a, ma_a) + dynamic_score_narrow(a.p, npod, na0, na1, nr0, nr1, nz0, nz1));
            }
        } else if constexpr (!NARROW)
            mj = sb_node_score(a.p, nj, mt_a, ma_a);
        const unsigned long long kj = mj >= 0 ? make_key((int64_t)mj, (int64_t)rpos(g)) : 0ull;
        const unsigned long long fail = __ballot(!(kj > rest));
        int32_t r = fail ? __ffsll((long long)fail) : 64; // clones placed: up to and including the first state that loses
        r = (int64_t)r > cap ? (int32_t)cap : r;
        const int32_t nm = lane_bcast_i32(mj, r - 1);
        node_apply<NX>(a.p, nd, (int64_t)r);
        pc += r;
        cur_m = nm, cur_key = nm >= 0 ? make_key((int64_t)nm, (int64_t)rpos(g)) : 0ull;
        if (lane < r && a.log && placed + lane < log_cap) a.log[placed + lane] = g;
        SF_TICK(3);
        pf[7] += 1;
        last_feasible = (int32_t)Ftotal;
        if (nm < 0) { // the node left the feasible ones: counts, and the maxima it may have held
            Ftotal -= 1;
            uint32_t x = 0, y = 0;
#pragma unroll
            for (int k = 0; k < NP; k++) {
                const int64_t i = ((int64_t)pb << sh) + k * 64 + lane;
                if (bm[k] >= 0 && i != g) {
                    const uint32_t w = a.c.stat[i], cnt = (w >> kStatCntShift) & kStatCntMask, aff = w & kStatAffMask;
                    x = cnt > x ? cnt : x, y = aff > y ? aff : y;
                }
            }
            const uint32_t lm = (wave_max_u32(x) << 16) | wave_max_u32(y);
            if (lane == 0) L.mx[pb] = lm, a.sb_mx[pb] = lm, atomicSub(&a.sb_fc[pb], 1u);
            lap_wave_sync();
            const uint32_t gmn = sf_wave_pkmax(L.mx[(pb >> 6) * 64 + lane]);
            gm = lane == (pb >> 6) ? gmn : gm;
            root_mx = sf_wave_pkmax(gm);
        }
        top = cur_key > rest ? cur_key : rest;
        placed += r, rounds += r, scans += r, evaluated += (int64_t)r * N, last_evaluated = N, winner = g, budget -= r, cycles += r;
    }
    close_block();
    store_row(pg, nd, pc, cur_m);
    if (ck_b >= 0 && lane == 0) a.sb_key[ck_b] = to_index(ck_leaf);
#undef SF_TICK
    if (a.prof && lane == 0)
        for (int i = 0; i < 8; i++) a.prof[i] += pf[i];
    if (dirty) { // the maxima the rebuild runs under: those of the feasible nodes
        if (lane == 0) S.mt_a = (int32_t)(root_mx >> 16), S.ma_a = (int32_t)(root_mx & 0xffffu);
    }
    if (lane == 0) {
        S.placed = placed, S.rounds = rounds, S.scans = scans, S.evaluated = evaluated, S.winner = winner;
        S.last_feasible = last_feasible, S.last_evaluated = last_evaluated, S.done = done;
        S.sb_dirty = dirty;
        S.sb_cycles += 1, S.sb_laps += cycles;
    }
}

} // namespace ccsim
