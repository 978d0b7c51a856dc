// Micro-benchmark: throughput of LDS operations on RANDOM addresses of a 13.5k-slot table (the access pattern of the
// residue-table inserts), 16 waves per CU, as clk per wave-instruction for the whole CU.
// Build: hipcc --offload-arch=gfx950 -O3 -o lds_random tools/micro/lds_random.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr int kSlots = 13312;
__device__ __forceinline__ unsigned h32(unsigned k) { k *= 0x9E3779B1u; k ^= k >> 15; k *= 0x85EBCA77u; k ^= k >> 13; return k; }

// MODE 0: ds_read_b32   1: ds_write_b32   2: cas rtn (dependent use)   3: add_f64 no return
//      4: cas rtn then add_f64 on the same slot (an insert)   5: U independent cas in flight, then U adds   6: ds_read_b64
template <int MODE, int U>
__global__ void __launch_bounds__(1024) k(int iters, int active_lanes, long long* out, unsigned* sink) {
    extern __shared__ unsigned char smem[];
    double* vals = (double*)smem; int* keys = (int*)(smem + 8 * kSlots);
    for (int i = threadIdx.x; i < kSlots; i += blockDim.x) { vals[i] = 0.0; keys[i] = -1; }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const bool act = lane < active_lanes;
    unsigned acc = 0, seed = threadIdx.x * 7919u + 13u;
    const long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
        unsigned s[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { seed = h32(seed + u); s[u] = (unsigned)(((unsigned long long)seed * kSlots) >> 32); }
        if (!act) continue;
        if (MODE == 0) { acc += keys[s[0]]; }
        if (MODE == 6) { acc += (unsigned)__double_as_longlong(vals[s[0]]); }
        if (MODE == 1) { keys[s[0]] = (int)seed | 1; }
        if (MODE == 2) { int e = -1; __hip_atomic_compare_exchange_strong(&keys[s[0]], &e, (int)(s[0] & 0xffff), __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); acc += e; }
        if (MODE == 3) { __hip_atomic_fetch_add(&vals[s[0]], 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
        if (MODE == 4) { int e = -1; __hip_atomic_compare_exchange_strong(&keys[s[0]], &e, (int)(s[0] & 0xffff), __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                         if (e == -1 || e == (int)(s[0] & 0xffff)) __hip_atomic_fetch_add(&vals[s[0]], 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
        if (MODE == 5) {
            int e[U];
#pragma unroll
            for (int u = 0; u < U; ++u) { e[u] = -1; __hip_atomic_compare_exchange_strong(&keys[s[u]], &e[u], (int)(s[u] & 0xffff), __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
#pragma unroll
            for (int u = 0; u < U; ++u) if (e[u] == -1 || e[u] == (int)(s[u] & 0xffff)) __hip_atomic_fetch_add(&vals[s[u]], 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    __syncthreads();
    const long long t1 = clock64();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (acc == 0xFFFFFFFFu) sink[0] = acc;
}

template <int MODE, int U> double run(int threads, int active, int iters, long long* d_out, unsigned* d_sink) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<MODE, U>), hipFuncAttributeMaxDynamicSharedMemorySize, 12 * kSlots);
    long long h = 0;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((k<MODE, U>), dim3(1), dim3(threads), 12 * kSlots, 0, iters, active, d_out, d_sink);
        hipMemcpy(&h, d_out, 8, hipMemcpyDeviceToHost);
    }
    return (double)h / iters;
}

int main() {
    long long* d_out; unsigned* d_sink;
    hipMalloc(&d_out, 8 * 256); hipMalloc(&d_sink, 4);
    const int iters = 2000;
    for (int threads : {64, 256, 1024})
        for (int active : {64, 16}) {
            const int waves = threads / 64;
            printf("%4d threads, %2d active lanes: clk per WORKGROUP iteration (per wave-instruction group)\n", threads, active);
            printf("   read_b32 %7.1f (%6.1f)  read_b64 %7.1f  write_b32 %7.1f (%6.1f)  cas_rtn %7.1f (%6.1f)  add_f64 %7.1f (%6.1f)  cas+add %7.1f (%6.1f)\n",
                   run<0,1>(threads, active, iters, d_out, d_sink), run<0,1>(threads, active, iters, d_out, d_sink) / waves,
                   run<6,1>(threads, active, iters, d_out, d_sink),
                   run<1,1>(threads, active, iters, d_out, d_sink), run<1,1>(threads, active, iters, d_out, d_sink) / waves,
                   run<2,1>(threads, active, iters, d_out, d_sink), run<2,1>(threads, active, iters, d_out, d_sink) / waves,
                   run<3,1>(threads, active, iters, d_out, d_sink), run<3,1>(threads, active, iters, d_out, d_sink) / waves,
                   run<4,1>(threads, active, iters, d_out, d_sink), run<4,1>(threads, active, iters, d_out, d_sink) / waves);
            printf("   U inserts in flight: U=1 %7.1f  U=2 %7.1f  U=4 %7.1f  U=8 %7.1f   (clk per workgroup iteration; per insert-wave: U=1 %6.1f U=2 %6.1f U=4 %6.1f U=8 %6.1f)\n",
                   run<5,1>(threads, active, iters, d_out, d_sink), run<5,2>(threads, active, iters, d_out, d_sink),
                   run<5,4>(threads, active, iters, d_out, d_sink), run<5,8>(threads, active, iters, d_out, d_sink),
                   run<5,1>(threads, active, iters, d_out, d_sink) / waves, run<5,2>(threads, active, iters, d_out, d_sink) / waves / 2,
                   run<5,4>(threads, active, iters, d_out, d_sink) / waves / 4, run<5,8>(threads, active, iters, d_out, d_sink) / waves / 8);
        }
    return 0;
}
