// Micro-benchmark: cost of LDS atomics on gfx950 as a function of how many lanes hit one address.
// Build: hipcc --offload-arch=gfx950 -O3 -o lds_atomics tools/micro/lds_atomics.hip ; run on the GPU box.
// Prints cycles (s_memrealtime is 100 MHz; we use clock64 = shader clock) per wave-instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE>   // 0: u32 add no return, 1: u32 add with return, 2: f64 add no return, 3: cas u32 with return
__global__ void __launch_bounds__(1024) k(int distinct, int iters, long long* out, unsigned* sink) {
    __shared__ unsigned long long tab[8192];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) tab[i] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // lanes map onto `distinct` addresses (spread over banks); all waves hit the same addresses
    const int a = (lane % distinct) * 33 % 4096;
    unsigned acc = 0;
    const long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) __hip_atomic_fetch_add((unsigned*)&tab[a], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (MODE == 1) acc += __hip_atomic_fetch_add((unsigned*)&tab[a], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (MODE == 2) __hip_atomic_fetch_add((double*)&tab[a], 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (MODE == 3) { unsigned e = (unsigned)i; __hip_atomic_compare_exchange_strong((unsigned*)&tab[a], &e, (unsigned)i + 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); acc += e; }
    }
    __syncthreads();
    const long long t1 = clock64();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (acc == 0xFFFFFFFFu) sink[0] = acc + (unsigned)tab[wave];
}

int main() {
    long long* d_out; unsigned* d_sink;
    hipMalloc(&d_out, 8 * 256); hipMalloc(&d_sink, 4);
    const int iters = 2000;
    const char* names[4] = {"add_u32", "add_u32_rtn", "add_f64", "cas_u32_rtn"};
    for (int threads : {64, 1024})
        for (int mode = 0; mode < 4; ++mode)
            for (int distinct : {64, 16, 4, 1}) {
                long long h = 0;
                for (int rep = 0; rep < 2; ++rep) {
                    if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(threads), 0, 0, distinct, iters, d_out, d_sink);
                    if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(1), dim3(threads), 0, 0, distinct, iters, d_out, d_sink);
                    if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(1), dim3(threads), 0, 0, distinct, iters, d_out, d_sink);
                    if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(1), dim3(threads), 0, 0, distinct, iters, d_out, d_sink);
                    hipMemcpy(&h, d_out, 8, hipMemcpyDeviceToHost);
                }
                printf("%4d threads %-12s distinct=%2d : %7.1f clk / wave-instruction (whole workgroup: %7.1f clk per iteration)\n",
                       threads, names[mode], distinct, (double)h / iters / (threads / 64), (double)h / iters);
            }
    return 0;
}
