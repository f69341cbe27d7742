// ccsim_sampled_zone.h -- the SAMPLED SEARCH (percentageOfNodesToScore < 100: the reference's default) of ONE template with a hard
// PodTopologySpread constraint over a shared key (zones) and, optionally, required inter-pod anti-affinity over a key whose values are
// unique per node (kubernetes.io/hostname): BASELINE config 5's pod shape as one template -- what both hosts run when the flag is left
// unset (round 6).  The sequential mode does such a cycle literally: a counting pass, a prefix, a scoring pass over
// every node (35 us per cycle at 1M nodes).
//
// Reference: findNodesThatPassFilters (S/schedule_one.go:610-693) with PodTopologySpread's PreFilter / Filter
// (P/podtopologyspread/filtering.go:235-356) and InterPodAffinity's Filter (P/interpodaffinity/filtering.go:352-432).
//
// For this shape a node's verdict splits into a node-local part -- static filters, NodeResourcesFit, the inter-pod terms at the node's OWN
// (unique) topology value: it changes only when a clone lands on that node -- and ONE bit per zone: is the zone's match count within maxSkew
// of the global minimum (filtering.go:311-356).  The eligible zones are a 64-bit mask E, recomputed from 64 counts every cycle.  So what a
// cycle needs is resident, as in ccsim_sampled.h, but per (block of nodes, zone):
//   memo[n]            TotalScore of node n under the assumed maxima if its node-local part passes, else -1; zone8[n] = the node's zone
//   ent_key[b][z]      the best (score, lowest index) key among the zone-z nodes of block b that pass node-locally, with their flags
//   cntz[z][b]         how many there are (zone-major: when a zone enters or leaves E, one row is added to / taken from the masked counts)
// A cycle (k_sz_cycles, one persistent workgroup):
//   1. E from the zone counts; the masked per-block counts fE[b] (LDS) follow E by the rows of the zones that changed;
//   2. ring prefix of fE from the start block -> the block that holds the (K+1)-th feasible node;
//   3. that block and the start block node by node under E; the blocks in between by their (block, eligible zone) entries;
//   4. maxima differ from the assumed ones -> rebuild under the true ones; else
//   5. NodeInfo.update on the winner, the zone's count, the inter-pod tables, its memo word, its (block, zone) entry.
// Exactly the oracle's cycle (tests/test_sampling.py::test_sampled_zone_*).
#pragma once
#include "ccsim_sampled.h"

namespace ccsim {

constexpr int kSzThreads = 512, kSzWaves = kSzThreads / 64;
constexpr int kSzMaxBlocks = 4096; // blocks of 256 (64 on small snapshots) nodes
constexpr int kSzZones = 64;       // topology values of the hard constraint (value ids 1 .. 64)
constexpr int kSzE = kSzMaxBlocks / kSzThreads;
// This kernel's keys: (TotalScore + 1) << 40 | (2^28 - 1 - node index) << 12 | info.  Highest score first, then the lowest index; the twelve
// info bits below the (unique) index never decide.  info = the node's flags against the assumed maxima (3 bits) | counted by the constraint << 3
// | carries the inter-pod key << 4 | zone (value id 1 .. 64) << 5: what the placement needs to know about the winner travels with the key.  An
// (block, zone) ENTRY is the key of the group's best node with the flag bits replaced by the OR over the group.
constexpr unsigned long long kSzIdxMask = (1ull << 28) - 1;
__device__ __forceinline__ unsigned long long sz_key(int32_t score, int32_t idx, uint32_t info5, uint32_t zone) {
    return ((unsigned long long)((int64_t)score + 1) << kIdxBits) | ((kSzIdxMask - (unsigned long long)idx) << 12) | (unsigned long long)((info5 & 31u) | (zone << 5));
}
__device__ __forceinline__ int32_t sz_key_index(unsigned long long k) { return (int32_t)(kSzIdxMask - ((k >> 12) & kSzIdxMask)); }

struct SzArgs {
    DevCols c;
    DevPod p;
    DevState *st;
    DevPts pts;
    DevIpa ipa;
    int32_t *memo;               // [n_pad]
    uint8_t *zone8, *flag8;      // [n_pad] the node's zone (value id of the constraint's key, 0 = key absent); its info bits 0 .. 4 (sz_key)
    unsigned long long *ent_key; // [64][kSzMaxBlocks] zone-major: an eligible zone's blocks of a stretch are one run of memory
    uint8_t *cntz;               // [64][kSzMaxBlocks] zone-major
    uint32_t *over;              // [1] set by k_sz_build when some (block, zone) count does not fit a byte: the host takes the three-pass cycle
    int32_t *log;
    int32_t shift, n_blocks, max_cycles;
    int32_t n_values;            // topology values of the constraint (value ids 1 .. n_values <= 64)
    unsigned long long *present; // [1] zones that hold a counted node (filtering.go:274-277): the global minimum is over them (k_sz_build)
    unsigned long long *prof;
};

// the node-local verdict and score of node i: -1, or TotalScore under (mt, ma).  Inter-pod terms at the node's own topology value (a key
// unique per node: the tables' entries at that value are node state).
template <bool NARROW>
__device__ __forceinline__ int32_t sz_node_word(const SzArgs &a, const NarrowPod &npod, int64_t i, uint32_t mt, uint32_t ma) {
    const uint32_t w = a.c.stat[i];
    int32_t sc = -1;
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
    if (sc >= 0 && a.pts.label[0][i] == 0) sc = -1; // the node lacks the constraint's key (filtering.go:322-326)
    if (sc >= 0 && a.ipa.on && a.ipa.filter_on && ipa_filter(a.ipa, *a.st, i)) sc = -1;
    return sc;
}

// k_sz_build: memo, zone and flag bytes, the (block, zone) entries of the whole snapshot under (mt_a, ma_a).  One workgroup per block.
template <bool NARROW>
__global__ __launch_bounds__(256) void k_sz_build(SzArgs a) {
    const DevState &st = *a.st;
    if (st.done || !st.sb_dirty) return;
    const uint32_t mt = (uint32_t)st.mt_a, ma = (uint32_t)st.ma_a;
    const int tid = threadIdx.x, B = 1 << a.shift;
    const int64_t base = (int64_t)blockIdx.x << a.shift;
    const NarrowPod npod = narrow_pod(a.p, a.c.mem_shift);
    __shared__ unsigned long long s_key[kSzZones];
    __shared__ uint32_t s_cnt[kSzZones], s_flg[kSzZones];
    if (tid < kSzZones) s_key[tid] = 0, s_cnt[tid] = 0, s_flg[tid] = 0;
    __syncthreads();
    for (int j = tid; j < B; j += 256) {
        const int64_t i = base + j;
        if (i >= a.c.n_pad) break;
        int32_t sc = -1;
        uint32_t z = 0, fl = 0;
        if (i < a.c.n) {
            sc = sz_node_word<NARROW>(a, npod, i, mt, ma);
            z = (uint32_t)a.pts.label[0][i];
            const uint32_t w = a.c.stat[i], cnt = (w >> kStatCntShift) & kStatCntMask, aff = w & kStatAffMask;
            fl = ((cnt > mt || aff > ma) ? 1u : 0u) | (cnt == mt ? 2u : 0u) | (aff == ma ? 4u : 0u);
            const uint32_t eb = a.pts.elig[i];
            const bool counted = (eb & 1u) && ((eb >> 1) & 1u);
            if (z >= 1 && z <= kSzZones && counted) atomicOr(a.present, 1ull << (z - 1)); // (idempotent: the same bits on every rebuild)
            fl |= (counted && a.pts.self_match[0] ? 8u : 0u) | ((a.ipa.on && a.ipa.label[0][i] != 0) ? 16u : 0u);
        }
        a.memo[i] = sc, a.zone8[i] = (uint8_t)z, a.flag8[i] = (uint8_t)fl;
        if (sc >= 0 && z >= 1 && z <= kSzZones) {
            atomicMax(&s_key[z - 1], sz_key(sc, (int32_t)i, fl, z));
            atomicAdd(&s_cnt[z - 1], 1u), atomicOr(&s_flg[z - 1], fl & 7u);
        }
    }
    __syncthreads();
    if (tid < kSzZones) {
        a.ent_key[(int64_t)tid * kSzMaxBlocks + blockIdx.x] = s_key[tid] ? (s_key[tid] & ~7ull) | s_flg[tid] : 0ull;
        a.cntz[(int64_t)tid * kSzMaxBlocks + blockIdx.x] = (uint8_t)(s_cnt[tid] > 255u ? 255u : s_cnt[tid]);
        if (s_cnt[tid] > 255u) atomicOr(a.over, 1u);
    }
}

struct SzLds {
    uint16_t fE[kSzMaxBlocks];   // feasible nodes of the block in the zones of E_prev
    int32_t zF[kSzZones];        // feasible (node-local) nodes per zone
    unsigned long long s_key[2]; // the cycle's best kept node: at or behind the start index [0], before it [1]
    uint32_t s_flag;
    uint32_t w_scan[kSzWaves], w_mt[kSzWaves], w_ma[kSzWaves];
    unsigned long long w_key[kSzWaves];
    int32_t stop_blk, stop_need, stop_kind, stop_node; // kind 0 a whole block, 1 the start block before the start index, -1 none (every node is visited)
    uint32_t tailF, headF;
    int32_t new_word;
    uint8_t elist[kSzZones]; // the eligible zones in ascending order
    uint32_t ent_cnt_new, ent_cnt_old;
};

template <bool NARROW, int NP> // NP nodes per lane of a cut block: blocks of 64 x NP nodes
__global__ __launch_bounds__(kSzThreads) void k_sz_cycles(SzArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sb_lds_raw[];
    SzLds &L = *reinterpret_cast<SzLds *>(sb_lds_raw);
    DevState &S = *a.st;
    if (S.done) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6); // (wave: known to be uniform, so what follows from it stays in scalar registers)
    const int nb = a.n_blocks, sh = a.shift;
    const int32_t N = (int32_t)a.c.n;
    const uint32_t K = (uint32_t)S.smp_K;
    const int64_t limit = S.limit, log_cap = S.log_cap;
    const uint32_t mt_a = (uint32_t)S.mt_a, ma_a = (uint32_t)S.ma_a;
    const NarrowPod npod = narrow_pod(a.p, a.c.mem_shift);
    int32_t start = (int32_t)S.smp_start;
    int64_t placed = S.placed, rounds = S.rounds, scans = S.scans, evaluated = S.evaluated, winner = S.winner;
    int32_t last_feasible = S.last_feasible, last_evaluated = S.last_evaluated, done = 0, dirty = 0, cycles = 0;
    uint32_t new_mt = mt_a, new_ma = ma_a;
    auto ringpos = [&](int32_t i) -> int32_t { return i >= start ? i - start : i + N - start; };
    // one block, NP consecutive nodes per lane: memo words, zone bytes, flag bytes (plain loads: every wave invalidates its L1 at the start
    // of a cycle, and what the cycle before wrote was released to L2 before the barrier in front of that)
    auto fetch = [&](int blk, int32_t *m, uint32_t &zw, uint32_t &fw) {
        const int64_t i0 = ((int64_t)blk << sh) + (int64_t)lane * NP;
        if (NP == 4) {
            const int4 x = *reinterpret_cast<const int4 *>(a.memo + i0);
            m[0] = x.x, m[1] = x.y, m[2] = x.z, m[3] = x.w;
            zw = *reinterpret_cast<const uint32_t *>(a.zone8 + i0), fw = *reinterpret_cast<const uint32_t *>(a.flag8 + i0);
        } else
            m[0] = a.memo[i0], zw = a.zone8[i0], fw = a.flag8[i0];
    };
    // ---- the per-zone state: match counts, feasible nodes; the masked counts start empty (E_prev = no zone)
    if (tid < kSzZones) L.zF[tid] = 0;
    int32_t zc_lane = lane < a.n_values ? a.pts.tbl[0][lane + 1] : 0; // every wave: lane z holds zone z's match count (TpValueToMatchNum of the one hard constraint)
    for (int b = tid; b < kSzMaxBlocks; b += kSzThreads) L.fE[b] = 0;
    __syncthreads();
    for (int z = wave; z < kSzZones; z += kSzWaves) { // feasible nodes per zone
        uint32_t s = 0;
        for (int b = lane; b < nb; b += 64) s += a.cntz[(int64_t)z * kSzMaxBlocks + b];
        s = wave_sum_u32_dpp(s);
        if (lane == 0) L.zF[z] = (int32_t)s;
    }
    __syncthreads();
    unsigned long long E_prev = 0;
    const unsigned long long present = *a.present;
    int64_t budget = a.max_cycles;
    unsigned long long t_prev = a.prof ? __builtin_amdgcn_s_memrealtime() : 0ull, pf[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define SZ_TICK(i) do { if (a.prof) { const unsigned long long t_now = __builtin_amdgcn_s_memrealtime(); pf[i] += t_now - t_prev; t_prev = t_now; } } while (0)
    const int Eb = (nb - 1 + kSzThreads - 1) / kSzThreads; // ring entries per thread
    auto eligible = [&](int32_t zc) -> unsigned long long { // (wave-level: lane = zone, zc = its match count)
        const bool pres = (present >> lane) & 1ull;
        const uint32_t mn_all = ~wave_max_u32(pres ? ~(uint32_t)zc : 0u); // min over the present zones (0xffffffff if there is none)
        const int64_t minm = a.pts.n_present[0] < a.pts.min_domains[0] ? 0 : (int64_t)(int32_t)mn_all;
        return __ballot(lane < a.n_values && (int64_t)zc + a.pts.self_match[0] - minm <= (int64_t)a.pts.max_skew[0]);
    };
    bool carried = false; // the start block's counts under E were left by the cycle before (its stop block IS this start block)

    while (!done && !dirty && budget > 0) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        const int sb = start >> sh;
        // ---- 1. the eligible zones (filtering.go:311-356; every wave, lane = zone): count + selfMatch - min <= maxSkew, the minimum over the
        // zones that hold a counted node (0 when there are fewer of them than minDomains, :105-117)
        unsigned long long E = eligible(zc_lane);
        const uint32_t FE = wave_sum_u32_dpp(((E >> lane) & 1ull) ? (uint32_t)L.zF[lane] : 0u);
        if (FE == 0) { // schedule_one.go:448-454: every node was visited, none passed
            done = DONE_UNSCHEDULABLE, rounds += 1, scans += 1, last_feasible = 0, last_evaluated = N, evaluated += N, winner = -1;
            break;
        }
        if (wave == 1 && ((E >> lane) & 1ull)) L.elist[__popcll(E & ((1ull << lane) - 1ull))] = (uint8_t)lane;
        const bool all = FE <= K; // fewer feasible nodes than wanted: the search visits every node (:538: processed = N)
        // the masked counts follow E: the rows of the zones that entered or left
        {
            unsigned long long diff = E ^ E_prev;
            while (diff) {
                const int z = __ffsll((long long)diff) - 1;
                diff &= diff - 1;
                const bool add = (E >> z) & 1ull;
                for (int b = tid; b < nb; b += kSzThreads) {
                    const uint16_t c = a.cntz[(int64_t)z * kSzMaxBlocks + b];
                    L.fE[b] = add ? (uint16_t)(L.fE[b] + c) : (uint16_t)(L.fE[b] - c);
                }
            }
            E_prev = E;
        }
        // the start block under E: its words for step 3 (wave 0); its feasible nodes at or behind / before the start index -- counted here in
        // the first cycle of a launch, left in LDS by the cycle before otherwise (its stop block IS this start block)
        int32_t sm[NP];
        uint32_t szw = 0, sfw = 0;
        if (wave == 0) {
            fetch(sb, sm, szw, sfw);
            if (!carried) {
                uint32_t t = 0, h = 0;
#pragma unroll
                for (int k = 0; k < NP; k++) {
                    const uint32_t z = (szw >> (8 * k)) & 0xffu;
                    const bool fe = sm[k] >= 0 && z && ((E >> (z - 1)) & 1ull);
                    const int32_t i = (sb << sh) + lane * NP + k;
                    t += (fe && i >= start) ? 1u : 0u, h += (fe && i < start) ? 1u : 0u;
                }
                t = wave_sum_u32_dpp(t), h = wave_sum_u32_dpp(h);
                if (lane == 0) L.tailF = t, L.headF = h;
            }
            if (lane == 0) L.s_key[0] = 0, L.s_key[1] = 0, L.s_flag = 0, L.stop_blk = -1, L.stop_kind = -1, L.stop_node = -1;
        }
        __syncthreads(); // ---- barrier 1: fE, tailF
        SZ_TICK(0);
        // ---- 2. ring prefix over the whole blocks: entry r = 1 .. nb - 1 is block (sb + r) mod nb; which one holds rank K
        const uint32_t tailF = L.tailF, headF = L.headF;
        const int r_lo = tid * Eb + 1, r_hi = (tid + 1) * Eb < nb - 1 ? (tid + 1) * Eb : nb - 1;
        uint32_t ef[kSzE], ls = 0;
#pragma unroll
        for (int e = 0; e < kSzE; e++) {
            ef[e] = 0;
            if (e < Eb && r_lo + e <= r_hi) {
                int b = sb + r_lo + e;
                b = b >= nb ? b - nb : b;
                ef[e] = L.fE[b];
            }
            ls += ef[e];
        }
        const uint32_t ils = lap_wave_incl(ls);
        if (lane == 63) L.w_scan[wave] = ils;
        __syncthreads(); // ---- barrier 2
        uint32_t fullF = 0, before = 0;
#pragma unroll
        for (int w = 0; w < kSzWaves; w++) {
            const uint32_t z = L.w_scan[w];
            fullF += z, before += w < wave ? z : 0u;
        }
        if (!all) {
            uint32_t run = tailF + before + ils - ls; // feasible nodes in front of this thread's first entry
#pragma unroll
            for (int e = 0; e < kSzE; e++) {
                if (ef[e] && run <= K && K < run + ef[e]) { // rank K (the (K+1)-th feasible node) lies in this block
                    int b = sb + r_lo + e;
                    b = b >= nb ? b - nb : b;
                    L.stop_blk = b, L.stop_need = (int32_t)(K - run), L.stop_kind = 0;
                }
                run += ef[e];
            }
            if (tid == 0 && tailF + fullF <= K) L.stop_blk = sb, L.stop_need = (int32_t)(K - tailF - fullF), L.stop_kind = 1; // in the start block, before the start index
        }
        __syncthreads(); // ---- barrier 3: where the stretch ends
        SZ_TICK(1);
        // ---- 3. the kept nodes: the start block at or behind the start index (wave 0), the stop block before the stop node (wave 1; all of the
        // start block's head when every node is visited), the blocks in between by their (block, eligible zone) entries (waves 2 ..)
        const int stop_blk = L.stop_blk, stop_kind = L.stop_kind;
        const int32_t stop_need = L.stop_need;
        int32_t nxm[NP]; // (wave 1: the words of the block the stretch ends in)
        uint32_t nxzw = 0, nxfw = 0;
        auto offer = [&](unsigned long long k, uint32_t fl, bool part1) { // (wave-level: lane 0 files the wave's best)
            const unsigned long long kb = lap_wave_best(k != 0, k);
            const uint32_t f = lap_wave_or3(k != 0, fl);
            if (lane == 0 && kb) atomicMax(&L.s_key[part1 ? 1 : 0], kb), atomicOr(&L.s_flag, f);
        };
        if (wave == 0) {
            unsigned long long bk = 0;
            uint32_t bf = 0;
#pragma unroll
            for (int k = 0; k < NP; k++) {
                const uint32_t z = (szw >> (8 * k)) & 0xffu;
                const int32_t i = (sb << sh) + lane * NP + k;
                if (sm[k] >= 0 && z && ((E >> (z - 1)) & 1ull) && i >= start) {
                    const unsigned long long key = sz_key(sm[k], i, (sfw >> (8 * k)) & 31u, z);
                    bk = key > bk ? key : bk, bf |= (sfw >> (8 * k)) & 7u;
                }
            }
            offer(bk, bf, false);
        } else if (wave == 1) {
            const int blk = all ? sb : stop_blk;
            fetch(blk, nxm, nxzw, nxfw); // (also the NEXT cycle's start block: kept for its counts, step 5)
            if (blk >= 0 && (!all || headF > 0)) {
                int32_t *m = nxm;
                const uint32_t zw = nxzw, fw = nxfw;
                const int32_t i0 = (blk << sh) + lane * NP;
                const bool head = all || stop_kind == 1; // the segment: the start block's nodes before the start index, or a whole block
                uint32_t fm = 0;
#pragma unroll
                for (int k = 0; k < NP; k++) {
                    const uint32_t z = (zw >> (8 * k)) & 0xffu;
                    fm |= (m[k] >= 0 && z && ((E >> (z - 1)) & 1ull) && (!head || i0 + k < start)) ? 1u << k : 0u;
                }
                const uint32_t cnt = (uint32_t)__popc(fm), incl = lap_wave_incl(cnt);
                const int32_t need = all ? 0x7fffffff : stop_need, d = need - (int32_t)(incl - cnt), np = d < 0 ? 0 : (d > (int32_t)cnt ? (int32_t)cnt : d);
                unsigned long long bk = 0;
                uint32_t bf = 0;
                int32_t seen = 0, stop = -1;
#pragma unroll
                for (int k = 0; k < NP; k++)
                    if (fm >> k & 1u) {
                        if (seen < np) {
                            const unsigned long long key = sz_key(m[k], i0 + k, (fw >> (8 * k)) & 31u, (zw >> (8 * k)) & 0xffu);
                            bk = key > bk ? key : bk, bf |= (fw >> (8 * k)) & 7u;
                        } else if (seen == np)
                            stop = i0 + k;
                        seen += 1;
                    }
                const unsigned long long smk = __ballot(d >= 0 && d < (int32_t)cnt);
                if (!all && lane == 0) L.stop_node = smk ? lane_bcast_i32(stop, __ffsll((long long)smk) - 1) : -1;
                offer(bk, bf, head || blk < sb);
            }
        }
        SZ_TICK(4); // (per wave: its cut block)
        {
            // ---- every wave: its span of the (chunk of 64 ring positions, eligible zone) pairs between the two cut blocks -- zone fastest; 16
            // entries in flight per lane.  Ring positions sb + 1 .. (the stop block's, or sb + nb): position p is block p mod nb, before the
            // start index when p >= nb; zone-major entries: a zone's blocks of a stretch are one run of memory.
            const int p_lo = sb + 1, p_hi = all ? sb + nb : (stop_kind == 1 ? sb + nb : (stop_blk > sb ? stop_blk : stop_blk + nb));
            const int nblk = p_hi - p_lo, ne = __popcll(E);
            const int nchunk = (nblk + 63) >> 6, total = ne * nchunk, per = (total + kSzWaves - 1) / kSzWaves;
            int x = wave * per;
            const int x_end = x + per < total ? x + per : total;
            unsigned long long bk0 = 0, bk1 = 0;
            uint32_t bf = 0;
            if (x < x_end) {
                int j = x / ne; // (one division per wave and cycle)
                unsigned long long zm = E; // the eligible zones not yet taken at chunk j
                for (int q = x - j * ne; q > 0; q--) zm &= zm - 1;
                constexpr int kU = 16;
                while (x < x_end) {
                    unsigned long long kk[kU];
                    bool wr[kU];
#pragma unroll
                    for (int u = 0; u < kU; u++) {
                        const bool okx = x + u < x_end;
                        const int z = __ffsll((long long)zm) - 1; // (wave-uniform: scalar)
                        const int pb = j * 64 + lane;
                        const bool ok = okx && pb < nblk;
                        const int p = p_lo + (ok ? pb : 0), b = p >= nb ? p - nb : p;
                        kk[u] = a.ent_key[(okx ? z : 0) * kSzMaxBlocks + b], wr[u] = p >= nb;
                        kk[u] = ok ? kk[u] : 0ull;
                        if (okx) {
                            zm &= zm - 1;
                            if (!zm) zm = E, j += 1;
                        }
                    }
#pragma unroll
                    for (int u = 0; u < kU; u++)
                        if (kk[u]) {
                            if (wr[u]) bk1 = kk[u] > bk1 ? kk[u] : bk1;
                            else bk0 = kk[u] > bk0 ? kk[u] : bk0;
                            bf |= (uint32_t)kk[u] & 7u;
                        }
                    x += kU;
                }
            }
            const unsigned long long k0 = wave_max_u64(bk0), k1 = wave_max_u64(bk1); // (lanes are blocks of several zones here, not index order: the full key decides)
            const uint32_t f = lap_wave_or3((bk0 | bk1) != 0, bf);
            if (lane == 0) {
                if (k0) atomicMax(&L.s_key[0], k0);
                if (k1) atomicMax(&L.s_key[1], k1);
                if (k0 | k1) atomicOr(&L.s_flag, f);
            }
        }
        SZ_TICK(5); // (per wave: its entries)
        __syncthreads(); // ---- barrier 4: the cycle's best kept node and the kept nodes' flags
        SZ_TICK(2);
        // ---- 4. the maxima over the kept nodes against the assumed ones
        unsigned long long key;
        {
            const unsigned long long k0 = L.s_key[0], k1 = L.s_key[1];
            key = (k1 >> kIdxBits) > (k0 >> kIdxBits) ? k1 : k0; // (on equal scores the part at or behind the start index: it is visited first)
        }
        const int32_t stop_node = all ? -1 : L.stop_node;
        scans += 1;
        if (L.s_flag != (kLapHitT | kLapHitA)) { // normalized with other maxima than the kept nodes': rebuild under the true ones
            const int32_t sp = all ? N : ringpos(stop_node);
            uint32_t x = 0, y = 0;
            for (int32_t dd = tid; dd < sp; dd += kSzThreads) {
                int32_t i = start + dd;
                i = i >= N ? i - N : i;
                const uint32_t z = a.zone8[i];
                if (a.memo[i] >= 0 && z && ((E >> (z - 1)) & 1ull)) {
                    const uint32_t w = a.c.stat[i], cnt = (w >> kStatCntShift) & kStatCntMask, aff = w & kStatAffMask;
                    x = cnt > x ? cnt : x, y = aff > y ? aff : y;
                }
            }
            x = wave_max_u32(x), y = wave_max_u32(y);
            if (lane == 0) L.w_mt[wave] = x, L.w_ma[wave] = y;
            __syncthreads();
#pragma unroll
            for (int w = 0; w < kSzWaves; w++) x = L.w_mt[w] > x ? L.w_mt[w] : x, y = L.w_ma[w] > y ? L.w_ma[w] : y;
            new_mt = x, new_ma = y, dirty = 1;
            break;
        }
        // ---- 5. the placement (schedule_one.go:967-984 assume -> NodeInfo.update; the clone is an existing pod of the next cycle:
        // filtering.go:255-296, interpodaffinity/filtering.go:204-272) by one thread; wave 1 re-reads the winner's block for its entry
        const int32_t g = sz_key_index(key);
        const int gblk = g >> sh;
        const uint32_t gz = (uint32_t)(key >> 5) & 127u; // (>= 1: the node was feasible)
        const bool counted = (key >> 3) & 1ull;
        // a clone with required anti-affinity against itself over the node's own (unique) topology value: the node is out from now on
        const bool blocks_itself = a.ipa.on && a.ipa.filter_on && a.ipa.anti_self_on_key[0] > 0 && ((key >> 4) & 1ull);
        // the zones eligible in the NEXT cycle follow from the winner's zone alone: every wave knows them now
        const int32_t zc_g = lane_bcast_i32(zc_lane, (int)gz - 1);
        zc_lane += (lane == (int)gz - 1 && counted) ? 1 : 0;
        const unsigned long long E_next = eligible(zc_lane);
        if (tid == 0) {
            const int64_t i = g;
            int32_t nw = -1;
            if (blocks_itself) { // the row by atomics nobody waits for: its next reader is the end of the run (the node never passes again)
                atomicAdd((unsigned long long *)&a.c.req[0][i], (unsigned long long)a.p.req[0]), atomicAdd((unsigned long long *)&a.c.req[1][i], (unsigned long long)a.p.req[1]);
                atomicAdd((unsigned long long *)&a.c.nz_mcpu[i], (unsigned long long)a.p.nz_mcpu), atomicAdd((unsigned long long *)&a.c.nz_mem[i], (unsigned long long)a.p.nz_mem);
                atomicAdd(&a.c.pod_count[i], 1), atomicAdd(&a.c.placed_cnt[i], 1);
                if (a.c.narrow) {
                    atomicAdd(&a.c.r32[0][i], (int32_t)a.p.req[0]), atomicAdd(&a.c.r32[1][i], (int32_t)(a.p.req[1] >> a.c.mem_shift));
                    atomicAdd(&a.c.z32[0][i], (int32_t)a.p.nz_mcpu), atomicAdd(&a.c.z32[1][i], (int32_t)(a.p.nz_mem >> a.c.mem_shift));
                }
#pragma unroll 1
                for (int col = 2; col < a.p.ncol; col++)
                    if (a.p.req[col] != 0) atomicAdd((unsigned long long *)&a.c.req[col][i], (unsigned long long)a.p.req[col]);
                a.memo[i] = -1;
            } else // NodeInfo.update and the node-local word from the row in registers (the inter-pod tables at its own value do not move: no term of the clone matches itself)
                nw = sb_place<NARROW>(a, npod, i, mt_a, ma_a);
            if (counted) a.pts.tbl[0][gz] = zc_g + 1;
            if (a.ipa.on)
                for (int k = 0; k < a.ipa.n_keys; k++) {
                    const int32_t v = a.ipa.label[k][i];
                    if (!v) continue;
                    if (a.ipa.self_aff && a.ipa.aff_terms_on_key[k]) atomicAdd((unsigned long long *)&a.ipa.aff[k][v], (unsigned long long)a.ipa.aff_terms_on_key[k]), S.ipa_aff_total += a.ipa.aff_terms_on_key[k];
                    if (a.ipa.anti_self_on_key[k]) {
                        atomicAdd((unsigned long long *)&a.ipa.anti[k][v], (unsigned long long)a.ipa.anti_self_on_key[k]), atomicAdd((unsigned long long *)&a.ipa.exist[k][v], (unsigned long long)a.ipa.anti_self_on_key[k]);
                        S.ipa_exist_total += a.ipa.anti_self_on_key[k];
                    }
                    if (a.ipa.score_self[k]) atomicAdd((unsigned long long *)&a.ipa.score[k][v], (unsigned long long)a.ipa.score_self[k]);
                    S.ipa_entries += a.ipa.self_entries[k];
                }
            L.new_word = nw;
            if (a.log && placed < log_cap) a.log[placed] = g;
        }
        int32_t pm[NP];
        uint32_t pzw = 0, pfw = 0;
        if (wave == 1) fetch(gblk, pm, pzw, pfw);
        if (wave >= 2) { // the masked counts follow E_next: the rows of the zones that enter or leave -- every block but the winner's (wave 1 below)
            unsigned long long diff = E_next ^ E;
            while (diff) {
                const int z = __ffsll((long long)diff) - 1;
                diff &= diff - 1;
                const bool add = (E_next >> z) & 1ull;
                for (int b = tid - 2 * 64; b < nb; b += kSzThreads - 2 * 64) {
                    if (b == gblk) continue;
                    const uint16_t c = a.cntz[(int64_t)z * kSzMaxBlocks + b];
                    L.fE[b] = add ? (uint16_t)(L.fE[b] + c) : (uint16_t)(L.fE[b] - c);
                }
            }
        }
        E_prev = E_next;
        if (!blocks_itself) __syncthreads(); // ---- barrier 5: the winner's new word (known without it when the clone blocks its own node)
        const int32_t visited = all ? N : ringpos(stop_node);
        const int32_t next_start = all ? start : stop_node;
        if (wave == 1) { // the (block, zone) entry of the winner; the winner's block under E_next; the next cycle's start block under E_next
            const int32_t new_word = blocks_itself ? -1 : L.new_word;
            const int32_t i0 = (gblk << sh) + lane * NP;
            unsigned long long bk = 0;
            uint32_t bf = 0, c = 0, c_old = 0;
#pragma unroll
            for (int k = 0; k < NP; k++) {
                const uint32_t z = (pzw >> (8 * k)) & 0xffu;
                if (z != gz) continue;
                const int32_t m = i0 + k == g ? new_word : pm[k];
                c_old += (i0 + k == g || pm[k] >= 0) ? 1u : 0u; // (the winner was feasible before)
                if (m >= 0) {
                    const unsigned long long k2 = sz_key(m, i0 + k, (pfw >> (8 * k)) & 31u, z);
                    bk = k2 > bk ? k2 : bk, bf |= (pfw >> (8 * k)) & 7u, c += 1;
                }
            }
            bk = lap_wave_best(bk != 0, bk);
            bf = lap_wave_or3(true, bf), c = wave_sum_u32_dpp(c), c_old = wave_sum_u32_dpp(c_old);
            // the winner's block in the masked counts: counted anew from its words under E_next
            uint32_t fe_g = 0;
#pragma unroll
            for (int k = 0; k < NP; k++) {
                const uint32_t z = (pzw >> (8 * k)) & 0xffu;
                const int32_t m = i0 + k == g ? new_word : pm[k];
                fe_g += (m >= 0 && z && ((E_next >> (z - 1)) & 1ull)) ? 1u : 0u;
            }
            fe_g = wave_sum_u32_dpp(fe_g);
            // the block the stretch ended in (every node was visited: the start block) is the next start block: its counts under E_next
            const int nblk2 = all ? sb : stop_blk;
            const int32_t j0 = (nblk2 << sh) + lane * NP;
            uint32_t t = 0, h = 0;
#pragma unroll
            for (int k = 0; k < NP; k++) {
                const uint32_t z = (nxzw >> (8 * k)) & 0xffu;
                const int32_t m = j0 + k == g ? new_word : nxm[k];
                const bool fe = m >= 0 && z && ((E_next >> (z - 1)) & 1ull);
                t += (fe && j0 + k >= next_start) ? 1u : 0u, h += (fe && j0 + k < next_start) ? 1u : 0u;
            }
            t = wave_sum_u32_dpp(t), h = wave_sum_u32_dpp(h);
            if (lane == 0) {
                a.ent_key[(int64_t)(gz - 1) * kSzMaxBlocks + gblk] = bk ? (bk & ~7ull) | (bf & 7u) : 0ull;
                a.cntz[(int64_t)(gz - 1) * kSzMaxBlocks + gblk] = (uint8_t)c;
                const int32_t dc = (int32_t)c - (int32_t)c_old; // 0 or -1
                L.zF[gz - 1] += dc;
                L.fE[gblk] = (uint16_t)fe_g;
                L.tailF = t, L.headF = h;
            }
        }
        if (tid == 0 || tid == 64) __threadfence(); // the row, the memo word, the entry: in L2 before the barrier below
        start = next_start, carried = true;
        placed += 1, rounds += 1, winner = g, evaluated += visited, last_evaluated = visited;
        last_feasible = (int32_t)(all ? FE : K);
        budget -= 1, cycles += 1;
        if (limit > 0 && placed >= limit) done = DONE_LIMIT; // simulator.go:297-312
        __syncthreads(); // ---- barrier 6: the entry, the zone state, the next start block's counts
        SZ_TICK(3);
    }
#undef SZ_TICK
    if (a.prof && tid == 0)
        for (int i = 0; i < 4; i++) a.prof[i] += pf[i];
    if (a.prof && tid == 64) a.prof[4] += pf[4], a.prof[5] += pf[5];   // (wave 1: the stop block, its entries; the rest of [2] is waiting)
    if (a.prof && tid == 128) a.prof[6] += pf[4], a.prof[7] += pf[5];  // (wave 2: no cut block, its entries)
    if (wave == 0) {
        const uint32_t mn = ~wave_max_u32(((present >> lane) & 1ull) ? ~(uint32_t)zc_lane : 0u);
        if (lane == 0) S.pts_min_a[0] = mn == 0xffffffffu ? 0x7fffffff : (int32_t)mn;
    }
    if (tid == 0) {
        S.smp_start = start, S.placed = placed, S.rounds = rounds, S.scans = scans, S.evaluated = evaluated, S.winner = winner;
        S.last_feasible = last_feasible, S.last_evaluated = last_evaluated, S.done = done;
        S.sb_dirty = dirty;
        if (dirty) S.mt_a = (int32_t)new_mt, S.ma_a = (int32_t)new_ma;
        S.sb_cycles += 1, S.sb_laps += cycles;
        // the global minimum the FitError diagnosis pass filters with (k_hist reads DevState::pts_min_a): computed below by wave 0
    }
}

} // namespace ccsim
