// Micro-benchmark: issue cost of a few VALU instructions on gfx950, as shader cycles per wave-instruction and SIMD,
// with 1 and 4 waves per SIMD (chains of dependent instructions, 8 independent chains per wave).
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_rate tools/micro/valu_rate.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ void __launch_bounds__(1024) k(int iters, long long* out, unsigned* sink)
{
    unsigned a[8];
    for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 2654435761u + i;
    const unsigned m = 0x9E3779B1u | (unsigned)iters;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MODE == 0) asm volatile("v_add_u32 %0, %1, %0" : "+v"(a[i]) : "v"(m));
                if (MODE == 1) asm volatile("v_mul_lo_u32 %0, %1, %0" : "+v"(a[i]) : "v"(m));
                if (MODE == 2) asm volatile("v_mul_hi_u32 %0, %1, %0" : "+v"(a[i]) : "v"(m));
                if (MODE == 3) asm volatile("v_mul_u32_u24 %0, %1, %0" : "+v"(a[i]) : "v"(m));
                if (MODE == 4) asm volatile("v_mad_u32_u24 %0, %1, %0, %0" : "+v"(a[i]) : "v"(m));
                if (MODE == 5) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(a[i]) : "v"(m));
                if (MODE == 6) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(a[i]) : "v"(m));
            }
    }
    const long long t1 = clock64();
    unsigned acc = 0;
    for (int i = 0; i < 8; ++i) acc ^= a[i];
    if (acc == 0x12345678u) sink[0] = acc;
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}

template <int MODE> void run(const char* name, int threads)
{
    long long* out; unsigned* sink;
    hipMalloc(&out, 8 * 256); hipMalloc(&sink, 4);
    const int iters = 2000;
    hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(threads), 0, 0, iters, out, sink);
    hipDeviceSynchronize();
    hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(threads), 0, 0, iters, out, sink);
    hipDeviceSynchronize();
    long long h = 0; hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost);
    const double per_simd_instr = (double)h / ((double)iters * 64.0 * (threads / 256.0));   // wave-instructions issued per SIMD
    printf("%-16s %4d threads (%d waves/SIMD): %.2f cycles per wave-instruction and SIMD\n", name, threads, threads / 256, per_simd_instr);
    hipFree(out); hipFree(sink);
}

int main()
{
    for (int threads : {256, 1024}) {
        run<0>("v_add_u32", threads); run<5>("v_xor_b32", threads); run<6>("v_lshl_add_u32", threads);
        run<1>("v_mul_lo_u32", threads); run<2>("v_mul_hi_u32", threads); run<3>("v_mul_u32_u24", threads); run<4>("v_mad_u32_u24", threads);
    }
    return 0;
}
