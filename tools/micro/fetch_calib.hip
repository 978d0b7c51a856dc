// fetch_calib.hip -- what the gfx950 L2 -> fabric read counters report for the access shapes of the GFPush kernel.
//
// rocprofv3's FETCH_SIZE is derived from TCC_EA0_RDREQ (requests, tallied at 64 B unless flagged 32 B / "bubble" 128 B);
// MI355X_MICROARCH.md says it reads exactly 1/2 of a wide coalesced stream on gfx950 and leaves other shapes uncalibrated.
// Every kernel here reads a buffer far larger than the 256 MiB Infinity Cache exactly ONCE with a known byte count:
//   gather4   : one random 4-byte word per lane                      (SCAN's indptr lookups)
//   runs4     : runs of `run` consecutive 4-byte words at random     (EXPAND's CSR column ranges; run = 4, 14, 64)
//   soa12     : a 4-byte and an 8-byte array read in lockstep        (the reserve log as TOP-K reads it, 4 records per lane)
//   stream16  : 16 bytes per lane, fully coalesced                   (the guide's calibration point)
// Usage: fetch_calib            -> runs all kernels once and prints "name requested_bytes sector64_bytes"
//        tools/fetch_calib.sh   -> runs it under rocprofv3 --pmc and prints counter x 64 B against those byte counts.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u32; typedef unsigned long long u64;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ u32 mix(u32 x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

__global__ void __launch_bounds__(256) gather4(const u32* buf, u64 n_words, u64 n_loads, u32* out) {
    u32 acc = 0;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n_loads; i += (u64)gridDim.x * blockDim.x) {
        const u64 j = ((u64)mix((u32)i) * 2654435761ull + mix((u32)(i >> 32) + 17u)) % n_words;
        acc += buf[j];
    }
    if (acc == 0x12345678u) out[0] = acc;
}
// run consecutive words per group of `run` lanes, groups at random (aligned to nothing): lane l of a group reads start + l
__global__ void __launch_bounds__(256) runs4(const u32* buf, u64 n_words, u64 n_loads, u32 run, u32* out) {
    u32 acc = 0;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n_loads; i += (u64)gridDim.x * blockDim.x) {
        const u64 g = i / run; const u32 l = (u32)(i % run);
        const u64 start = ((u64)mix((u32)g) * 2654435761ull + mix((u32)(g >> 32) + 29u)) % (n_words - run);
        acc += buf[start + l];
    }
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ void __launch_bounds__(256) soa12(const uint4* keys, const double2* vals, u64 n_quads, u32* out) {   // 4 records per lane and step
    u32 acc = 0;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n_quads; i += (u64)gridDim.x * blockDim.x) {
        const uint4 k = keys[i]; const double2 a = vals[2 * i], b = vals[2 * i + 1];
        acc += k.x + k.y + k.z + k.w + (u32)__double2uint_rz(a.x + a.y + b.x + b.y);
    }
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ void __launch_bounds__(256) stream16(const uint4* buf, u64 n_vec, u32* out) {
    u32 acc = 0;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n_vec; i += (u64)gridDim.x * blockDim.x) {
        const uint4 v = buf[i]; acc += v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}

int main() {
    const u64 bytes = 3ull << 30;                                   // 3 GiB >> 256 MiB Infinity Cache
    char* buf; u32* out;
    CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&out, 64));
    CK(hipMemset(buf, 1, bytes)); CK(hipDeviceSynchronize());
    const u64 n_words = bytes / 4;
    const int grid = 256 * 8;
    const u64 n_loads = 1ull << 27;                                 // 128 Mi loads of 4 B
    hipLaunchKernelGGL(gather4, dim3(grid), dim3(256), 0, 0, (const u32*)buf, n_words, n_loads, out);
    CK(hipDeviceSynchronize());
    printf("gather4 requested %llu sector64 %llu\n", 4 * n_loads, 64 * n_loads);
    for (u32 run : {4u, 14u, 64u}) {
        hipLaunchKernelGGL(runs4, dim3(grid), dim3(256), 0, 0, (const u32*)buf, n_words, n_loads, run, out);
        CK(hipDeviceSynchronize());
        printf("runs4_%u requested %llu sector64 %llu\n", run, 4 * n_loads, (n_loads / run) * (u64)(64 * ((4 * run + 63 + 32) / 64)));
    }
    const u64 n_quads = (bytes / 3) / 16 / 2 * 2;                   // keys take 1/3, values 2/3 of the buffer
    hipLaunchKernelGGL(soa12, dim3(grid), dim3(256), 0, 0, (const uint4*)buf, (const double2*)(buf + 16 * n_quads), n_quads, out);
    CK(hipDeviceSynchronize());
    printf("soa12 requested %llu sector64 %llu\n", 48 * n_quads, 48 * n_quads);
    hipLaunchKernelGGL(stream16, dim3(grid), dim3(256), 0, 0, (const uint4*)buf, bytes / 16, out);
    CK(hipDeviceSynchronize());
    printf("stream16 requested %llu sector64 %llu\n", bytes, bytes);
    return 0;
}
