"""The exact-arithmetic tricks of the HIP kernels, checked WITHOUT a GPU: the scalar `__device__` functions that replace the
reference's integer divisions and IEEE fp64 sequences (cluster-capacity_amd/csrc/ccsim_kernels.h: div_magic / norm100,
floor_ratio100, dynamic_score_narrow, div_small_quotient / dynamic_score, static_score) are cut out of the shipped header as
text, compiled for the host with a few shims (`__umul24`, `__umulhi`, the hardware reciprocals), and compared with the oracle's
unit functions (oracle/ccref.c: ccref_least_allocated, ccref_balanced_allocation) and with plain 64-bit integer arithmetic --
exhaustively where the domain is small, on adversarial + random operands where it is not.

The hardware reciprocals (v_rcp_f32: 1 ulp; v_rcp_f64: ~2^-26 relative before the Newton step in refined_rcp) are not available
here, so the shims return the correctly rounded reciprocal PERTURBED by the documented error in either direction: the fix-ups must
give the exact result for every reciprocal within the bound, not for one particular implementation.

What this pins: the claims of DESIGN.md 4.1 ("estimate off by at most one + remainder fix-up", "mulhi with ceil(2^32/m) is exact
for n*m < 2^32", "the f32 value is within 3e-4 of the fp64 one, else the IEEE sequence") as properties of the code that ships.
The GPU suite (tests/test_gpu_parity.py ...) then checks the same functions in place, on the device."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "cluster-capacity_amd", "csrc", "ccsim_kernels.h")

# definitions cut out of the header, in dependency order: (kind, name, occurrence)
WANTED = [("struct", "DevPod", 0), ("func", "refined_rcp", 0), ("struct", "NodeRcp", 0), ("func", "make_rcp", 0), ("func", "div_small_quotient", 0),
          ("func", "least_requested_score", 0), ("func", "balanced_exact", 0), ("func", "dynamic_score", 0), ("func", "div_magic", 0), ("func", "norm100", 0),
          ("func", "norm100", 1), ("struct", "NarrowPod", 0), ("func", "floor_ratio100", 0), ("func", "dynamic_score_narrow", 0), ("func", "static_score", 0),
          ("func", "static_score", 1), ("func", "go_log", 0), ("func", "fits_narrow", 0), ("struct", "RunDownCoef", 0), ("func", "rd_total", 0), ("func", "run_down_safe_skip", 0)]


def _extract(text, kind, name, occurrence):
    """The definition's source text: from the line that starts it to the brace that closes it."""
    pat = re.compile(r"^struct %s \{" % name if kind == "struct" else r"^__device__ [^\n;]*?\b%s\(" % name, re.M)
    m = list(pat.finditer(text))[occurrence]
    i = text.index("{", m.start())
    depth = 0
    for j in range(i, len(text)):
        depth += text[j] == "{"
        depth -= text[j] == "}"
        if depth == 0:
            end = j + 1
            break
    if kind == "struct":
        end = text.index(";", end) + 1
    return text[m.start():end]


HARNESS = r"""
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
extern "C" {
int64_t ccref_least_allocated(const int64_t *requested, const int64_t *allocatable, const int64_t *weights, int n);
int64_t ccref_balanced_allocation(const int64_t *requested, const int64_t *allocatable, int n);
double ccref_go_log(double x);
}
#define __device__
#define __forceinline__ inline
#define __noinline__
static int g_ulp = 0;          // v_rcp_f32: the correctly rounded reciprocal moved by this many ulps
static double g_rel = 0.0;     // v_rcp_f64: relative error of the seed before refined_rcp's Newton step
static inline float shim_rcpf(float x) {
    float r = 1.0f / x;
    for (int k = 0; k < (g_ulp < 0 ? -g_ulp : g_ulp); k++) r = nextafterf(r, g_ulp > 0 ? INFINITY : 0.0f);
    return r;
}
static inline double shim_rcp(double x) { return (1.0 / x) * (1.0 + g_rel); }
#define __builtin_amdgcn_rcpf(x) shim_rcpf(x)
#define __builtin_amdgcn_rcp(x) shim_rcp(x)
static inline uint32_t __umul24(uint32_t a, uint32_t b) { return (a & 0xffffffu) * (b & 0xffffffu); }
static inline uint32_t __umulhi(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) >> 32); }
constexpr int kMaxRes = 11;
constexpr int kMaxExtra = 9;
// ---- cut out of cluster-capacity_amd/csrc/ccsim_kernels.h ----
@@EXTRACTED@@
// ---------------------------------------------------------------
static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t rnd() { rng_state ^= rng_state << 13, rng_state ^= rng_state >> 7, rng_state ^= rng_state << 17; return rng_state; }
static int64_t rnd_below(int64_t n) { return (int64_t)(rnd() % (uint64_t)n); }
static long failures = 0;
#define CHECK(cond, ...) do { if (!(cond)) { if (failures++ < 10) { std::printf("FAIL %s: ", #cond); std::printf(__VA_ARGS__); std::printf("\n"); } } } while (0)

// the reference's scores of one node through the oracle's unit functions (oracle/ccref.c fit_score / balanced_score)
static int64_t ref_total(const DevPod &p, int64_t a0, int64_t a1, int64_t r0, int64_t r1, int64_t z0, int64_t z1, int64_t q_req0, int64_t q_req1, int64_t q_nz0,
                         int64_t q_nz1) {
    int64_t total = 0;
    if (p.w_fit) {
        int64_t rq[2], al[2], w[2];
        int n = 0;
        if (p.fit_cpu) rq[n] = a0 ? z0 + q_nz0 : 0, al[n] = a0, w[n] = p.fit_w_cpu, n++;
        if (p.fit_mem) rq[n] = a1 ? z1 + q_nz1 : 0, al[n] = a1, w[n] = p.fit_w_mem, n++;
        total += ccref_least_allocated(rq, al, w, n) * p.w_fit;
    }
    if (p.w_bal) {
        int64_t rq[2], al[2];
        int n = 0;
        if (p.bal_cpu) rq[n] = a0 ? r0 + q_req0 : 0, al[n] = a0, n++;
        if (p.bal_mem) rq[n] = a1 ? r1 + q_req1 : 0, al[n] = a1, n++;
        total += ccref_balanced_allocation(rq, al, n) * p.w_bal;
    }
    return total;
}

static DevPod random_pod() {
    DevPod p;
    std::memset(&p, 0, sizeof p);
    p.w_fit = (int32_t)rnd_below(4), p.w_bal = (int32_t)rnd_below(3);
    p.fit_cpu = rnd_below(8) != 0, p.fit_mem = rnd_below(8) != 0, p.bal_cpu = rnd_below(8) != 0, p.bal_mem = rnd_below(8) != 0;
    p.fit_w_cpu = 1 + rnd_below(rnd_below(2) ? 3 : 100), p.fit_w_mem = 1 + rnd_below(rnd_below(2) ? 3 : 100);
    return p;
}

// operands that put a quotient next to an integer boundary (where an estimate may land on the wrong side)
static int64_t near_boundary(int64_t A, int64_t scale) { // d with d * scale / A close to an integer
    const int64_t k = rnd_below(scale + 1);
    int64_t d = (k * A) / scale + rnd_below(5) - 2;
    return d < 0 ? 0 : (d > A ? A : d);
}

int main() {
    long n_checked = 0;
    // (a) DefaultNormalizeScore: floor(100 c / m), EVERY maximum below 2^13 and every count up to it
    for (uint32_t m = 1; m < 8192; m++) {
        const uint32_t magic = div_magic(m);
        for (uint32_t c = 0; c <= m; c++, n_checked++) CHECK(norm100(c, m, magic) == 100u * c / m, "norm100 c=%u m=%u", c, m);
    }
    // (b) the weighted mean of LeastAllocated: num / W through div_magic(W), every weight sum and numerator the kernel can meet
    for (uint32_t W = 3; W <= 200; W++)
        for (uint32_t num = 0; num <= 100 * W; num++, n_checked++) CHECK(__umulhi(num, div_magic(W)) == num / W, "mean num=%u W=%u", num, W);
    // (c) floor(100 d / A) from an f32 estimate + wrapping 32-bit remainder fix-up: every reciprocal within 1 ulp (2 for margin)
    for (g_ulp = -2; g_ulp <= 2; g_ulp++) {
        for (uint32_t A = 1; A <= 1500; A++) // small capacities: every d
            for (uint32_t d = 0; d <= A; d++, n_checked++) CHECK(floor_ratio100(d, A) == (uint32_t)((uint64_t)d * 100 / A), "ratio d=%u A=%u ulp=%d", d, A, g_ulp);
        for (int it = 0; it < 3000000; it++, n_checked++) {
            const int sh = (int)rnd_below(30);
            uint32_t A = (uint32_t)(1 + rnd_below(((int64_t)1 << (sh + 1)) - 1));
            if (A >= (1u << 30)) A = (1u << 30) - 1;
            const uint32_t d = (uint32_t)(it % 3 == 0 ? rnd_below((int64_t)A + 1) : near_boundary(A, 100));
            CHECK(floor_ratio100(d, A) == (uint32_t)((uint64_t)d * 100 / A), "ratio d=%u A=%u ulp=%d", d, A, g_ulp);
        }
    }
    // (d) the narrow scores of one node (cpu in milli-cores, memory in shifted units, all below 2^30) against the oracle
    for (g_ulp = -1; g_ulp <= 1; g_ulp++)
        for (int it = 0; it < 2500000; it++, n_checked++) {
            const DevPod p = random_pod();
            const int sh0 = (int)rnd_below(29), sh1 = (int)rnd_below(29);
            int32_t a0 = (int32_t)(1 + rnd_below((int64_t)1 << (sh0 + 1))), a1 = (int32_t)(1 + rnd_below((int64_t)1 << (sh1 + 1)));
            if (rnd_below(50) == 0) a0 = 0;
            if (rnd_below(50) == 0) a1 = 0;
            NarrowPod q;
            q.req0 = (int32_t)rnd_below(a0 / 4 + 2), q.req1 = (int32_t)rnd_below(a1 / 4 + 2);
            q.nz0 = q.req0 ? q.req0 : 100, q.nz1 = q.req1 ? q.req1 : 200;
            int32_t r0, r1;
            if (it % 2) { // fractions whose difference puts (1 - |f0 - f1| / 2) * 100 next to an integer
                r0 = (int32_t)near_boundary(a0 ? a0 : 1, 200), r1 = (int32_t)near_boundary(a1 ? a1 : 1, 200);
                if (a0 && a1 && a0 == a1 && rnd_below(2)) r1 = r0;
            } else
                r0 = (int32_t)rnd_below((int64_t)a0 + 1), r1 = (int32_t)rnd_below((int64_t)a1 + 1);
            if (rnd_below(4) == 0) a1 = a0, r1 = (int32_t)near_boundary(a0 ? a0 : 1, 200); // equal capacities: many exact integers
            const int32_t z0 = it % 3 ? r0 : (int32_t)near_boundary(a0 ? a0 : 1, 100), z1 = it % 3 ? r1 : (int32_t)near_boundary(a1 ? a1 : 1, 100);
            const int64_t got = dynamic_score_narrow(p, q, a0, a1, r0, r1, z0, z1);
            const int64_t want = ref_total(p, a0, a1, r0, r1, z0, z1, q.req0, q.req1, q.nz0, q.nz1);
            CHECK(got == want, "narrow got=%lld want=%lld a=(%d,%d) r=(%d,%d) z=(%d,%d) q=(%d,%d,%d,%d) fit=(%d,%d,%lld,%lld,%d) bal=(%d,%d,%d) ulp=%d", (long long)got,
                  (long long)want, a0, a1, r0, r1, z0, z1, q.req0, q.req1, q.nz0, q.nz1, p.fit_cpu, p.fit_mem, (long long)p.fit_w_cpu, (long long)p.fit_w_mem, p.w_fit,
                  p.bal_cpu, p.bal_mem, p.w_bal, g_ulp);
        }
    // (e) the wide scores (int64 operands: bytes of memory) through one refined reciprocal per resource
    const double rels[] = {0.0, 1.0 / (1 << 26), -1.0 / (1 << 26), 1.0 / (1 << 24), -1.0 / (1 << 24)};
    for (double rel : rels) {
        g_rel = rel;
        for (int it = 0; it < 1500000; it++, n_checked++) {
            DevPod p = random_pod();
            const int sh0 = (int)rnd_below(24), sh1 = 10 + (int)rnd_below(38);
            int64_t a0 = 1 + rnd_below((int64_t)1 << (sh0 + 1)), a1 = 1 + rnd_below((int64_t)1 << (sh1 + 1));
            if (rnd_below(50) == 0) a0 = 0;
            if (rnd_below(50) == 0) a1 = 0;
            p.req[0] = rnd_below(a0 / 4 + 2), p.req[1] = rnd_below(a1 / 4 + 2);
            p.nz_mcpu = p.req[0] ? p.req[0] : 100, p.nz_mem = p.req[1] ? p.req[1] : 200ll << 20;
            int64_t r0, r1;
            if (it % 2) r0 = near_boundary(a0 ? a0 : 1, 200), r1 = near_boundary(a1 ? a1 : 1, 200);
            else r0 = rnd_below(a0 + 1), r1 = rnd_below(a1 + 1);
            if (rnd_below(4) == 0) a1 = a0, r1 = near_boundary(a0 ? a0 : 1, 200);
            const int64_t z0 = it % 3 ? r0 : near_boundary(a0 ? a0 : 1, 100), z1 = it % 3 ? r1 : near_boundary(a1 ? a1 : 1, 100);
            const int64_t got = dynamic_score(p, make_rcp(a0, a1), a0, a1, r0, r1, z0, z1);
            const int64_t want = ref_total(p, a0, a1, r0, r1, z0, z1, p.req[0], p.req[1], p.nz_mcpu, p.nz_mem);
            CHECK(got == want, "wide got=%lld want=%lld a=(%lld,%lld) r=(%lld,%lld) z=(%lld,%lld) rel=%g", (long long)got, (long long)want, (long long)a0, (long long)a1,
                  (long long)r0, (long long)r1, (long long)z0, (long long)z1, rel);
        }
    }
    // (f) the static part: TaintToleration reversed, NodeAffinity, ImageLocality, weighted
    for (int it = 0; it < 2000000; it++, n_checked++) {
        DevPod p;
        std::memset(&p, 0, sizeof p);
        p.w_taint = (int32_t)rnd_below(5), p.w_aff = (int32_t)rnd_below(5), p.w_img = (int32_t)rnd_below(4);
        const uint32_t mt = (uint32_t)rnd_below(rnd_below(2) ? 8 : 2048), ma = (uint32_t)rnd_below(rnd_below(2) ? 300 : 8192);
        const uint32_t c = mt ? (uint32_t)rnd_below((int64_t)mt + 1) : 0, a = ma ? (uint32_t)rnd_below((int64_t)ma + 1) : 0, img = (uint32_t)rnd_below(101);
        int64_t want = (int64_t)img * p.w_img;
        if (p.w_taint) want += (int64_t)(mt == 0 ? 100 : 100 - 100 * (int64_t)c / mt) * p.w_taint;
        if (p.w_aff) want += (int64_t)(ma == 0 ? 0 : 100 * (int64_t)a / ma) * p.w_aff;
        CHECK(static_score(p, c, a, img, mt, ma) == want, "static c=%u a=%u img=%u mt=%u ma=%u", c, a, img, mt, ma);
    }
    // (g) Go's math.Log as the device restates it (PodTopologySpread weights, scoring.go:294-296: log(size + 2)) against the oracle's
    // restatement, bit for bit: every argument a cluster of up to 4M domains / nodes can produce, and random doubles
    for (int64_t k = 2; k <= 4000002; k++, n_checked++) {
        const double a = go_log((double)k), b = ccref_go_log((double)k);
        CHECK(std::memcmp(&a, &b, 8) == 0 && std::fabs(a - std::log((double)k)) <= 4.5e-16 * a, "go_log(%lld)", (long long)k);
    }
    for (int it = 0; it < 2000000; it++, n_checked++) {
        const double x = std::ldexp(1.0 + (double)rnd_below((int64_t)1 << 52) / 4503599627370496.0, (int)rnd_below(80) - 20);
        const double a = go_log(x), b = ccref_go_log(x);
        CHECK(std::memcmp(&a, &b, 8) == 0, "go_log(%a)", x);
    }
    // (h) run-downs that skip states: every state 1 .. k that run_down_safe_skip lets a run-down pass over must be feasible (the node
    // takes one more pod) and score >= Lo under the EXACT functions above -- cluster-shaped nodes (cpu in milli-cores, memory in Mi
    // units, tens to hundreds of pods), pods of every proportion, every threshold near the current score; the reciprocal perturbed
    long skipped_states = 0, skips = 0;
    for (g_ulp = -1; g_ulp <= 1; g_ulp++)
        for (int it = 0; it < 400000; it++) {
            DevPod p = random_pod();
            p.fit_enabled = 1;
            if (it % 4 == 0) p.w_fit = 1, p.w_bal = 1, p.fit_cpu = p.fit_mem = p.bal_cpu = p.bal_mem = 1, p.fit_w_cpu = p.fit_w_mem = 1; // the default profile
            const int32_t a0 = (int32_t)(rnd_below(2) ? 1000 * (1 + rnd_below(128)) : 1 + rnd_below(1 << 20)), a1 = (int32_t)(rnd_below(2) ? 1024 * (1 + rnd_below(512)) : 1 + rnd_below(1 << 24));
            NarrowPod q;
            q.req0 = (int32_t)(rnd_below(6) == 0 ? 0 : 1 + rnd_below(a0 / 8 + 2)), q.req1 = (int32_t)(rnd_below(6) == 0 ? 0 : 1 + rnd_below(a1 / 8 + 2));
            q.nz0 = q.req0 ? q.req0 : 100, q.nz1 = q.req1 ? q.req1 : 200;
            p.all_zero_req = q.req0 == 0 && q.req1 == 0;
            const int32_t r0 = (int32_t)rnd_below(a0 / 2 + 1), r1 = (int32_t)rnd_below(a1 / 2 + 1);
            const int32_t z0 = r0 + (int32_t)(rnd_below(3) ? 0 : rnd_below(a0 / 3 + 1)), z1 = r1 + (int32_t)(rnd_below(3) ? 0 : rnd_below(a1 / 3 + 1)); // pods without requests raise NonZeroRequested only
            const int32_t a_pods = (int32_t)(1 + rnd_below(rnd_below(2) ? 110 : 2000)), npods = (int32_t)rnd_below(a_pods);
            const int32_t stat = (int32_t)rnd_below(600);
            if (!fits_narrow(p, q, a0, a1, r0, r1, a_pods, npods)) continue;
            const int64_t s0 = stat + dynamic_score_narrow(p, q, a0, a1, r0, r1, z0, z1);
            const int32_t Lo = (int32_t)(s0 - rnd_below(rnd_below(2) ? 8 : 70)); // the node is in the list: it scores >= Lo now
            if (Lo < 0) continue;
            const int32_t k = run_down_safe_skip(p, q, a0, a1, r0, r1, z0, z1, a_pods, npods, stat, Lo);
            n_checked++;
            skips += k > 0;
            for (int32_t j = 1; j <= k; j++, skipped_states++) {
                const int32_t rr0 = r0 + j * q.req0, rr1 = r1 + j * q.req1, zz0 = z0 + j * q.nz0, zz1 = z1 + j * q.nz1;
                const bool f = fits_narrow(p, q, a0, a1, rr0, rr1, a_pods, npods + j);
                const int64_t sj = stat + dynamic_score_narrow(p, q, a0, a1, rr0, rr1, zz0, zz1);
                CHECK(f && sj >= Lo, "skip k=%d j=%d feasible=%d score=%lld Lo=%d a=(%d,%d) r=(%d,%d) z=(%d,%d) q=(%d,%d,%d,%d) pods=%d/%d w=(%d,%d) fit=(%d,%d,%lld,%lld) bal=(%d,%d)", k, j,
                      (int)f, (long long)sj, Lo, a0, a1, r0, r1, z0, z1, q.req0, q.req1, q.nz0, q.nz1, npods, a_pods, p.w_fit, p.w_bal, p.fit_cpu, p.fit_mem,
                      (long long)p.fit_w_cpu, (long long)p.fit_w_mem, p.bal_cpu, p.bal_mem);
                if (!(f && sj >= Lo)) break;
            }
        }
    CHECK(skips > 100000 && skipped_states > 2000000, "the skip must be exercised: %ld skips, %ld states", skips, skipped_states);
    n_checked += skipped_states;
    std::printf("run-down skips: %ld, states passed over: %ld\n", skips, skipped_states);
    std::printf("checked %ld failures %ld\n", n_checked, failures);
    return failures ? 1 : 0;
}
"""


def test_device_arithmetic_is_exact_on_the_host(tmp_path, ccref):
    text = open(HEADER).read()
    extracted = "\n\n".join(_extract(text, *w) for w in WANTED)
    assert "__builtin_amdgcn_rcpf" in extracted and "__umulhi" in extracted  # the tricks under test are really in the cut
    src = tmp_path / "device_arith.cpp"
    src.write_text(HARNESS.replace("@@EXTRACTED@@", extracted))
    exe = tmp_path / "device_arith"
    # no fast-math, no FMA contraction: the fp64 sequences must be evaluated as IEEE operations, like hipcc is told to
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-fopenmp", "-c", "-o", str(tmp_path / "ccref.o"), os.path.join(ROOT, "oracle", "ccref.c")])
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-o", str(exe), str(src), str(tmp_path / "ccref.o"), "-lm", "-fopenmp"])
    p = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-3000:]
    m = re.search(r"checked (\d+) failures 0", p.stdout)
    assert m and int(m.group(1)) > 50_000_000, p.stdout[-500:]
