// device_probe.hip -- measurement aids for the tools/ scripts (NOT part of libgrandplus.so or of include/grandplus.h):
// what a device delivers right now, independent of any counter's time base.  Build: tools/device_probe.py does it on first use
// (hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o tools/micro/libdeviceprobe.so tools/micro/device_probe.hip).
#include <hip/hip_runtime.h>
#include <stdint.h>

// One wave spins for ~200 us of the constant 100 MHz wall clock and reports how many shader cycles (s_memtime) went by.
__global__ void clock_probe_kernel(unsigned long long* out) {
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    unsigned long long w1 = w0;
    while (w1 - w0 < 20000ull) w1 = wall_clock64();
    const unsigned long long c1 = clock64();
    if (threadIdx.x == 0) { out[0] = c1 - c0; out[1] = w1 - w0; }
}
// A dependent chain of integer multiply-adds in one wave: iterations per microsecond are proportional to the shader clock.
__global__ void alu_probe_kernel(unsigned long long* out, int iters) {
    unsigned int x = threadIdx.x + 1u;
    const unsigned long long w0 = wall_clock64();
    for (int i = 0; i < iters; ++i) { x = x * 1664525u + 1013904223u; asm volatile("" : "+v"(x)); }
    const unsigned long long w1 = wall_clock64();
    if (threadIdx.x == 0) { out[0] = w1 - w0; out[1] = x; }
}
// A streaming copy over the whole chip (GB/s of HBM traffic, read + write).
__global__ void copy_probe_kernel(const uint4* src, uint4* dst, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) dst[i] = src[i];
}

extern "C" int probe_clock_mhz(int device, double* shader_mhz) {
    *shader_mhz = 0.0;
    if (hipSetDevice(device) != hipSuccess) return 1;
    unsigned long long* d = nullptr; unsigned long long h[2] = {0, 0};
    if (hipMalloc(&d, sizeof h) != hipSuccess) return 1;
    hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, 0, d);
    const bool ok = hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost) == hipSuccess;
    (void)hipFree(d);
    if (!ok) return 1;
    if (h[1]) *shader_mhz = 100.0 * (double)h[0] / (double)h[1];
    return 0;
}

extern "C" int probe_speed(int device, double* alu_iters_per_us, double* copy_gb_s) {
    *alu_iters_per_us = 0.0; *copy_gb_s = 0.0;
    if (hipSetDevice(device) != hipSuccess) return 1;
    const long long n16 = (long long)(256u << 20) / 16;                       // 256 MiB each way
    unsigned long long* d = nullptr; uint4* a = nullptr; uint4* b = nullptr; unsigned long long h[2] = {0, 0};
    hipEvent_t e0 = nullptr, e1 = nullptr;
    bool ok = hipMalloc(&d, sizeof h) == hipSuccess && hipMalloc(&a, (size_t)n16 * 16) == hipSuccess && hipMalloc(&b, (size_t)n16 * 16) == hipSuccess &&
              hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess;
    if (ok) {
        const int iters = 1 << 16;
        hipLaunchKernelGGL(alu_probe_kernel, dim3(1), dim3(64), 0, 0, d, iters);
        ok = hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost) == hipSuccess;
        if (ok && h[0]) *alu_iters_per_us = 100.0 * (double)iters / (double)h[0];
    }
    if (ok) {
        hipLaunchKernelGGL(copy_probe_kernel, dim3(4096), dim3(256), 0, 0, a, b, n16);       // first touch
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(copy_probe_kernel, dim3(4096), dim3(256), 0, 0, a, b, n16);
        (void)hipEventRecord(e1, 0);
        float ms = 0.f;
        ok = hipEventSynchronize(e1) == hipSuccess && hipEventElapsedTime(&ms, e0, e1) == hipSuccess;
        if (ok && ms > 0.f) *copy_gb_s = 2.0 * (double)n16 * 16.0 / ((double)ms * 1e6);
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (d) (void)hipFree(d);
    if (a) (void)hipFree(a);
    if (b) (void)hipFree(b);
    return ok ? 0 : 1;
}
