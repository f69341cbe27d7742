// barrier_bench.hip -- what one "reduce + grid barrier" step of a persistent level kernel costs on this chip.
// One 1024-thread block per CU; per step every block contributes a few 64-bit words (atomicMax / atomicAdd into one of
// kGroups sharded slot sets), arrives on its group's counter, the last arriver of a group arrives on the top counter, the
// last arriver overall publishes the generation; everybody polls the generation and then reads the kGroups slot sets.
//   hipcc --offload-arch=gfx950 -O3 tools/barrier_bench.hip -o gpurun_out/barrier_bench && gpurun_out/barrier_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

constexpr int kGroups = 8, kWords = 4, kRing = 4;
struct Sync {
    unsigned long long slot[kRing][kGroups][8]; // 64-byte lines: words 0..kWords-1 used
    unsigned int garrive[kRing][kGroups][16];   // one line per counter
    unsigned int top[kRing][16];
    unsigned int gen[16];
    unsigned int err[16];
};

__device__ __forceinline__ unsigned ld_u32(const unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned long long ld_u64(const unsigned long long *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int MODE> // 0 = barrier only; 1 = + kWords contributions and the read-back
__global__ __launch_bounds__(1024) void k_barrier(Sync *s, int steps, unsigned long long *out) {
    const int b = blockIdx.x, nb = gridDim.x, g = b % kGroups;
    const int gsize = nb / kGroups + (g < nb % kGroups ? 1 : 0);
    __shared__ unsigned long long s_red[kWords];
    __shared__ int s_err;
    if (threadIdx.x == 0) s_err = 0;
    __syncthreads();
    unsigned long long acc = 0;
    for (int it = 0; it < steps; it++) {
        const int r = it & (kRing - 1);
        if (threadIdx.x == 0) {
            if (MODE == 1) {
                __hip_atomic_fetch_max(&s->slot[r][g][0], (unsigned long long)(it * 1000 + b), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                for (int w = 1; w < kWords; w++)
                    __hip_atomic_fetch_add(&s->slot[r][g][w], (unsigned long long)(b + w), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            const unsigned a = __hip_atomic_fetch_add(&s->garrive[r][g][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (a == (unsigned)gsize - 1) {
                const unsigned t = __hip_atomic_fetch_add(&s->top[r][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (t == kGroups - 1) { // last of all: recycle the set two generations ahead, then release
                    const int z = (it + 2) & (kRing - 1);
                    for (int gg = 0; gg < kGroups; gg++) {
                        for (int w = 0; w < kWords; w++) __hip_atomic_store(&s->slot[z][gg][w], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(&s->garrive[z][gg][0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    __hip_atomic_store(&s->top[z][0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __hip_atomic_store(&s->gen[0], (unsigned)(it + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            int spins = 0;
            while (ld_u32(&s->gen[0]) < (unsigned)(it + 1)) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1 << 22)) { // bounded: never hang the box
                    __hip_atomic_store(&s->err[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    s_err = 1;
                    break;
                }
            }
        }
        __syncthreads();
        if (MODE == 1) {
            if (threadIdx.x < kGroups * kWords) {
                const int gg = threadIdx.x / kWords, w = threadIdx.x % kWords;
                unsigned long long v = ld_u64(&s->slot[r][gg][w]);
                // reduce over the groups: max for word 0, sum otherwise (lanes gg*kWords + w)
                for (int off = kWords; off < kGroups * kWords; off <<= 1) {
                    const unsigned long long o = __shfl_xor(v, off, 64);
                    v = w == 0 ? (o > v ? o : v) : v + o;
                }
                if (threadIdx.x < kWords) s_red[threadIdx.x] = v;
            }
            __syncthreads();
            acc += s_red[0] + s_red[1];
        }
        if (s_err) break; // (written before the barrier above)
    }
    if (threadIdx.x == 0) out[b] = acc;
}

int main(int argc, char **argv) {
    int steps = argc > 1 ? atoi(argv[1]) : 2000;
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    Sync *s;
    unsigned long long *out;
    hipMalloc(&s, sizeof(Sync));
    hipMalloc(&out, sizeof(unsigned long long) * 4096);
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    for (int mode = 0; mode < 2; mode++)
        for (int per_cu = 1; per_cu <= 1; per_cu++) {
            const int nb = cus * per_cu;
            for (int rep = 0; rep < 3; rep++) {
                hipMemset(s, 0, sizeof(Sync));
                hipDeviceSynchronize();
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(k_barrier<0>, dim3(nb), dim3(1024), 0, 0, s, steps, out);
                else hipLaunchKernelGGL(k_barrier<1>, dim3(nb), dim3(1024), 0, 0, s, steps, out);
                hipEventRecord(e1);
                hipDeviceSynchronize();
                float ms = 0;
                hipEventElapsedTime(&ms, e0, e1);
                unsigned err = 0;
                hipMemcpy(&err, &s->err[0], 4, hipMemcpyDeviceToHost);
                printf("mode %d  blocks %d x 1024 thr  steps %d  %.3f ms  -> %.2f us/step  err=%u\n", mode, nb, steps, ms, ms * 1e3 / steps, err);
            }
        }
    return 0;
}
