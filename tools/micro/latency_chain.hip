// Micro-benchmark (round 6): what ONE dependent round trip costs a wave at the sketch kernel's occupancy -- the floor under a phase,
// which is a chain of a handful of them.  Workgroups of 768 threads with 80 KB of dynamic LDS (two per CU,
// 24 waves per CU, like gfpush_sk_kernel<768>); every wave walks the same kind of chain, so the chip is loaded the way the kernel
// loads it.  Chains (each operation depends on the result of the one before):
//   0  ds_add_rtn_u32 on a random LDS word          (what an allocation, a compare-and-swap probe, a table read costs)
//   1  global load, pointer chase inside 256 KB per workgroup (L2-resident: a push-list entry, a log record written a phase ago)
//   2  global load, pointer chase over 8 GB          (HBM / Infinity Cache: a CSR line)
//   3  s_barrier of the 12 waves                      (every wave arrives at once: the barrier's own cost, no skew)
//   4  64-bit division + compare (the exact push test: v_div_scale / v_rcp / fma chain)
// Prints ns per operation with 512 workgroups (the chip full) and with 8 (one row's view of an idle chip).
// Build: hipcc --offload-arch=gfx950 -O3 -o latency_chain tools/micro/latency_chain.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ __forceinline__ unsigned long long wall_ns10() {              // constant 100 MHz counter
    unsigned long long t;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t));
    return t;
}

template <int MODE>
__global__ void __launch_bounds__(768, 6) chain(int iters, const unsigned* small, const unsigned* big, unsigned big_mask, unsigned long long* out, unsigned* sink)
{
    extern __shared__ unsigned smem[];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) smem[i] = (unsigned)i * 2654435761u;
    __syncthreads();
    const unsigned lane = threadIdx.x & 63u;
    unsigned x = threadIdx.x * 7919u + blockIdx.x * 104729u + 1u;
    double r = 1.0 + (double)lane;
    const unsigned* sm = small + (size_t)blockIdx.x * 65536u;
    const unsigned long long t0 = wall_ns10();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) x = __hip_atomic_fetch_add(&smem[(x >> 7) & 16383u], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) + x * 2654435761u;
        if (MODE == 1) x = sm[(x + lane) & 65535u];
        if (MODE == 2) x = big[((size_t)x * 64u + lane * 32u) & big_mask];      // one 128-byte line per lane: 64 lines per wave-load, as a step's column loads
        if (MODE == 3) __syncthreads();
        if (MODE == 4) { r = 1.0 / (r * 1e-5 + 1.0) + (r >= 1e-5 * (double)(lane + 1u) ? 1.0 : 2.0); }
    }
    const unsigned long long t1 = wall_ns10();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (x == 0xFFFFFFFFu || r == -1.0) sink[0] = x;
}

template <int MODE> double run(int wgs, int iters, const unsigned* d_small, const unsigned* d_big, unsigned big_mask, unsigned long long* d_out, unsigned* d_sink)
{
    hipFuncSetAttribute(reinterpret_cast<const void*>(chain<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    std::vector<unsigned long long> h(wgs);
    double best = 1e30;
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL((chain<MODE>), dim3(wgs), dim3(768), 80 * 1024, 0, iters, d_small, d_big, big_mask, d_out, d_sink);
        hipMemcpy(h.data(), d_out, 8 * wgs, hipMemcpyDeviceToHost);
        double s = 0; for (int i = 0; i < wgs; ++i) s += (double)h[i];
        best = std::min(best, s / wgs * 10.0 / iters);                   // 100 MHz ticks -> ns per operation, mean over workgroups
    }
    return best;
}

int main()
{
    const size_t n_small = (size_t)512 * 65536, n_big = (size_t)1 << 31;          // 128 MB of chase tables, 8 GB of lines
    unsigned *d_small, *d_big, *d_sink; unsigned long long* d_out;
    if (hipMalloc(&d_small, 4 * n_small) != hipSuccess || hipMalloc(&d_big, 4 * n_big) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
    hipMalloc(&d_out, 8 * 512); hipMalloc(&d_sink, 4);
    {
        std::vector<unsigned> h(n_small);
        unsigned s = 12345u;
        for (size_t i = 0; i < n_small; ++i) { s = s * 1664525u + 1013904223u; h[i] = s >> 8; }
        hipMemcpy(d_small, h.data(), 4 * n_small, hipMemcpyHostToDevice);
        for (size_t off = 0; off < n_big; off += n_small) hipMemcpy(d_big + off, h.data(), 4 * n_small, hipMemcpyHostToDevice);     // (random words everywhere)
    }
    const unsigned big_mask = (unsigned)(n_big - 1);
    const char* names[5] = {"LDS atomic with return (random word)", "global load, L2-resident chase", "global load, HBM chase (64 lines per wave-load)", "s_barrier, 12 waves", "fp64 division + compare"};
    for (int wgs : {512, 8}) {
        printf("%d workgroups of 768 threads, 80 KB LDS each (%s):\n", wgs, wgs == 512 ? "two per CU, the chip full" : "an idle chip");
        const double v[5] = { run<0>(wgs, 2000, d_small, d_big, big_mask, d_out, d_sink), run<1>(wgs, 500, d_small, d_big, big_mask, d_out, d_sink),
                              run<2>(wgs, 200, d_small, d_big, big_mask, d_out, d_sink), run<3>(wgs, 2000, d_small, d_big, big_mask, d_out, d_sink),
                              run<4>(wgs, 2000, d_small, d_big, big_mask, d_out, d_sink) };
        for (int m = 0; m < 5; ++m) printf("   %-52s %8.1f ns per operation\n", names[m], v[m]);
    }
    return 0;
}
