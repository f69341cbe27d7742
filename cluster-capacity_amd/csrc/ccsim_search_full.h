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
    // ---- the block summaries the build (or the launch before) left; the groups' maxima
    uint32_t fc = 0;
    for (int b = tid; b < kSfMaxBlocks; b += kSfThreads) {
        const bool in = b < nb;
        L.key[b] = in ? a.sb_key[b] : 0ull, L.mx[b] = in ? a.sb_mx[b] : 0u;
        fc += in ? a.sb_fc[b] : 0u;
    }
    fc = wave_sum_u32_dpp(fc);
    if (lane == 0) L.w_fc[wave] = fc;
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

    const int32_t N = (int32_t)a.c.n;
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
        const int32_t g = (int32_t)key_index(top), b = g >> sh, grp = b >> 6;
        // ---- the one trip: the winner's block of memo words (L1-bypassing: the launch has written some of them) and its row -- the same
        // address in every lane, so one request; every lane then holds the row and computes the same new score
        int32_t bm[NP];
#pragma unroll
        for (int k = 0; k < NP; k++) {
            const int64_t i = ((int64_t)b << sh) + k * 64 + lane;
            bm[k] = (NP == 4 || i < n_pad) ? ld_memo(a.memo + i) : -1; // (n_pad is a multiple of 512: blocks of 256 nodes never reach beyond it)
        }
        NodeRegs<NX> nd;
        int32_t na0 = 0, na1 = 0;
        if (NARROW) na0 = a.c.a32[0][g], na1 = a.c.a32[1][g];
        load_one<NX>(a.c, a.p, (int64_t)g, nd);
        const int32_t pc = a.c.placed_cnt[g];
        // ---- in its shadow: the best key of the group's other blocks, of the other groups
        const unsigned long long lv = L.key[grp * 64 + lane];
        const unsigned long long og = wave_max_u64(lane == (b & 63) ? 0ull : lv), orr = wave_max_u64(lane == grp ? 0ull : gk);
        // ---- NodeInfo.update (S/framework/types.go:409-428), the node's score afterwards
        node_apply<NX>(a.p, nd, 1);
        int32_t nm = -1;
        if (NARROW) {
            const int32_t nr0 = (int32_t)nd.r_cpu, nr1 = (int32_t)(nd.r_mem >> a.c.mem_shift), nz0 = (int32_t)nd.z_cpu, nz1 = (int32_t)(nd.z_mem >> a.c.mem_shift);
            if ((nd.w >> kStatOkBit) && fits_narrow(a.p, npod, na0, na1, nr0, nr1, nd.a_pods, nd.npods)) {
                const uint32_t cnt = (nd.w >> kStatCntShift) & kStatCntMask, aff = nd.w & kStatAffMask, img = (nd.w >> kStatImgShift) & kStatImgMask;
                nm = (int32_t)(static_score(a.p, cnt, aff, img, mt_a, ma_a) + dynamic_score_narrow(a.p, npod, na0, na1, nr0, nr1, nz0, nz1));
            }
        } else if constexpr (!NARROW)
            nm = sb_node_score(a.p, nd, mt_a, ma_a);
        if (lane == 0) {
            a.c.req[0][g] = nd.r_cpu, a.c.req[1][g] = nd.r_mem, a.c.nz_mcpu[g] = nd.z_cpu, a.c.nz_mem[g] = nd.z_mem, a.c.pod_count[g] = nd.npods;
            a.c.placed_cnt[g] = pc + 1;
            store_mirror(a.c, (int64_t)g, nd.r_cpu, nd.r_mem, nd.z_cpu, nd.z_mem);
            if (NX > 0) {
#pragma unroll
                for (int x = 0; x < NX; x++)
                    if (x < a.p.nx) a.c.req[a.p.xcol[x]][g] = nd.xr[x];
            }
            __hip_atomic_store((uint32_t *)(a.memo + g), (uint32_t)nm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (a.log && placed < log_cap) a.log[placed] = g;
        }
        // ---- the block's new key, the group's, the next winner
        unsigned long long best = 0;
#pragma unroll
        for (int k = 0; k < NP; k++) {
            const int32_t i = (b << sh) + k * 64 + lane;
            bm[k] = i == g ? nm : bm[k];
            const unsigned long long kk = bm[k] >= 0 ? make_key((int64_t)bm[k], (int64_t)i) : 0ull;
            best = kk > best ? kk : best;
        }
        const unsigned long long leaf = wave_max_u64(best), gnew = leaf > og ? leaf : og;
        if (lane == 0) L.key[b] = leaf, a.sb_key[b] = leaf;
        gk = lane == grp ? gnew : gk;
        top = gnew > orr ? gnew : orr;
        last_feasible = (int32_t)Ftotal;
        if (nm < 0) { // the node left the feasible ones: counts, and the maxima it may have held
            Ftotal -= 1;
            uint32_t x = 0, y = 0;
#pragma unroll
            for (int k = 0; k < NP; k++) {
                const int64_t i = ((int64_t)b << sh) + k * 64 + lane;
                if (bm[k] >= 0) {
                    const uint32_t w = a.c.stat[i], cnt = (w >> kStatCntShift) & kStatCntMask, aff = w & kStatAffMask;
                    x = cnt > x ? cnt : x, y = aff > y ? aff : y;
                }
            }
            const uint32_t lm = (wave_max_u32(x) << 16) | wave_max_u32(y);
            if (lane == 0) L.mx[b] = lm, a.sb_mx[b] = lm, atomicSub(&a.sb_fc[b], 1u);
            lap_wave_sync();
            const uint32_t gmn = sf_wave_pkmax(L.mx[grp * 64 + lane]);
            gm = lane == grp ? gmn : gm;
            root_mx = sf_wave_pkmax(gm);
        }
        lap_wave_sync(); // (the next cycle reads L.key)
        placed += 1, rounds += 1, scans += 1, evaluated += N, last_evaluated = N, winner = g, budget -= 1, cycles += 1;
    }
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
