// propagate.hip -- SURVEY.md 8(f) next-2: the exact full-graph feature propagation of GRAND+'s
// inference, reference predict() (model.py:181-224; model_mag.py:192-245), lines 186-210:
//
//   ppr   : X0 = alpha*F;  X_{i+1} = ((1-alpha)/max(deg,1e-12)) (.) (A X_i);  out = sum_{i=0..n} X_i
//   avg   : X0 = F;        X_{i+1} = (1/max(deg,1e-12)) (.) (A X_i);          out = (sum_i X_i)/(n+1)
//   single: X0 = F;        X_{i+1} = (1/max(deg,1e-12)) (.) (A X_i);          out = X_n
//
// with A = adj + I as stored (deg = row sum of the stored values, model.py:189/197/205).  The reference
// runs this with scipy CSR x dense on the CPU in float64 and then casts to float32 for the MLP
// (model.py:175).  Here: one CSR x dense SpMM per step on the CSR that is already resident for GFPush,
// HBM-gather bound (nnz*F*4 B per step).  A wave owns a row and its lanes span the feature columns, so
// every neighbour contributes one coalesced read of its feature row; sums are kept in fp64 registers
// and rounded to fp32 once per step (storage is fp32: the consumer is an fp32 MLP).  Rows longer than
// kLongRow neighbours are taken by whole workgroups with an ordered LDS reduction (deterministic).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

#include "grandplus.h"

namespace {

typedef unsigned int u32;
constexpr int kBlock = 256;                  // 4 waves: 4 rows per workgroup
constexpr int kLongRow = 4096;               // neighbours above which a row is handled by spmm_long_kernel
constexpr int kLongBlock = 1024;

// Main SpMM.  VEC floats per lane per access (4 when F % 4 == 0, 2 when F % 2 == 0, else 1).  A group
// of G = 2^log2g lanes (G*VEC >= F when possible) owns one row, so a wave handles 64/G rows at once and
// every neighbour costs one 16-byte (VEC = 4) load per lane; 8 neighbour rows are in flight per lane.
// `accumulate`: y is also added into `sum` (ppr / avg); y is always written to x_next.
template <int VEC> struct PV;
template <> struct PV<4> { typedef float4 type; };
template <> struct PV<2> { typedef float2 type; };
template <> struct PV<1> { typedef float type; };

template <int VEC>
__global__ void __launch_bounds__(kBlock)
spmm_kernel(const int* __restrict__ indptr, const int* __restrict__ indices, const float* __restrict__ wts,
            u32 node_mask, const double* __restrict__ scale, long long n_rows, const float* __restrict__ x, int F,
            float* __restrict__ x_next, float* __restrict__ sum, int accumulate, int log2g)
{
    typedef typename PV<VEC>::type V;
    const int lane = threadIdx.x & 63;
    const int G = 1 << log2g, gl = lane & (G - 1), grp = lane >> log2g, rows_per_wave = 64 >> log2g;
    const long long wave = ((long long)blockIdx.x * kBlock + threadIdx.x) >> 6;
    const long long n_waves = ((long long)gridDim.x * kBlock) >> 6;
    for (long long row = wave * rows_per_wave + grp; row < n_rows; row += n_waves * rows_per_wave) {
        const int begin = indptr[row], end = indptr[row + 1];
        if (end - begin > kLongRow) continue;                       // spmm_long_kernel's job
        const double s = scale[row];
        for (int f = gl * VEC; f < F; f += G * VEC) {               // one trip when G*VEC >= F
            double acc[VEC];
#pragma unroll
            for (int i = 0; i < VEC; ++i) acc[i] = 0.0;
            // 8 neighbour rows in flight per lane, the ragged tail included: a slot past the end re-reads the
            // row's last neighbour (cache hit) with weight 0 -- a serial tail loop would expose one full HBM
            // latency per remaining neighbour, and most rows of these graphs are shorter than 16.  (Keeping
            // the full batches unpredicated and predicating only the last one was measured 5 % slower on the
            // MAG and Amazon2M shapes, 3 % faster on the Reddit shape.)
            for (int j = begin; j < end; j += 8) {
                V v[8]; float w[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const bool live = j + u < end;
                    const int jj = live ? j + u : end - 1;
                    const u32 c = (u32)indices[jj] & node_mask;
                    w[u] = live ? (wts ? wts[jj] : 1.0f) : 0.0f;
                    v[u] = *reinterpret_cast<const V*>(x + (size_t)c * F + f);
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const float* pv = reinterpret_cast<const float*>(&v[u]);
#pragma unroll
                    for (int i = 0; i < VEC; ++i) acc[i] += (double)w[u] * (double)pv[i];       // adj.dot(features), model.py:191
                }
            }
            V y; float* py = reinterpret_cast<float*>(&y);
#pragma unroll
            for (int i = 0; i < VEC; ++i) py[i] = (float)(s * acc[i]);              // model.py:191 / :199 / :207
            *reinterpret_cast<V*>(x_next + (size_t)row * F + f) = y;
            if (accumulate) {                                                       // model.py:192 / :200
                V t = *reinterpret_cast<const V*>(sum + (size_t)row * F + f);
                float* pt = reinterpret_cast<float*>(&t);
#pragma unroll
                for (int i = 0; i < VEC; ++i) pt[i] += py[i];
                *reinterpret_cast<V*>(sum + (size_t)row * F + f) = t;
            }
        }
    }
}

// Hub rows: one workgroup per (long row, slab of 64*VEC columns); its 16 waves stride over the
// neighbours (8 rows in flight each), then an ordered LDS reduction over the waves (deterministic).
template <int VEC>
__global__ void __launch_bounds__(kLongBlock)
spmm_long_kernel(const int* __restrict__ indptr, const int* __restrict__ indices, const float* __restrict__ wts,
                 u32 node_mask, const double* __restrict__ scale, const int* __restrict__ long_rows, int n_long,
                 const float* __restrict__ x, int F, float* __restrict__ x_next, float* __restrict__ sum, int accumulate)
{
    typedef typename PV<VEC>::type V;
    __shared__ double part[kLongBlock / 64][64 * VEC];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n_w = kLongBlock / 64;
    const int slabs = (F + 64 * VEC - 1) / (64 * VEC);
    for (long long item = blockIdx.x; item < (long long)n_long * slabs; item += gridDim.x) {
        const int row = long_rows[item / slabs];
        const int f = ((int)(item % slabs) * 64 + lane) * VEC;
        const bool valid = f < F;                                  // F % VEC == 0: a vector is valid as a whole
        const int begin = indptr[row], end = indptr[row + 1];
        double acc[VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc[i] = 0.0;
        int j = begin + wave;
        for (; j + 7 * n_w < end; j += 8 * n_w) {
            V v[8]; float w[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int jj = j + u * n_w;
                const u32 c = (u32)indices[jj] & node_mask;
                w[u] = wts ? wts[jj] : 1.0f;
                v[u] = valid ? *reinterpret_cast<const V*>(x + (size_t)c * F + f) : V();
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float* pv = reinterpret_cast<const float*>(&v[u]);
#pragma unroll
                for (int i = 0; i < VEC; ++i) acc[i] += (double)w[u] * (double)pv[i];
            }
        }
        for (; j < end; j += n_w) {
            if (!valid) continue;
            const u32 c = (u32)indices[j] & node_mask;
            const float w = wts ? wts[j] : 1.0f;
            const V v = *reinterpret_cast<const V*>(x + (size_t)c * F + f);
            const float* pv = reinterpret_cast<const float*>(&v);
#pragma unroll
            for (int i = 0; i < VEC; ++i) acc[i] += (double)w * (double)pv[i];
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < VEC; ++i) part[wave][lane * VEC + i] = acc[i];
        __syncthreads();
        if (wave == 0 && valid) {
            V y; float* py = reinterpret_cast<float*>(&y);
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                double t = 0.0;
                for (int w = 0; w < n_w; ++w) t += part[w][lane * VEC + i];      // fixed order: deterministic
                py[i] = (float)(scale[row] * t);
            }
            *reinterpret_cast<V*>(x_next + (size_t)row * F + f) = y;
            if (accumulate) {
                V t = *reinterpret_cast<const V*>(sum + (size_t)row * F + f);
                float* pt = reinterpret_cast<float*>(&t);
#pragma unroll
                for (int i = 0; i < VEC; ++i) pt[i] += py[i];
                *reinterpret_cast<V*>(sum + (size_t)row * F + f) = t;
            }
        }
    }
}

// per-row scale (fp64, as numpy computes it) and the list of long rows
__global__ void __launch_bounds__(256)
prepare_kernel(const int* __restrict__ indptr, const float* __restrict__ wts, long long n_rows, double numer,
               double* __restrict__ scale, int* __restrict__ long_rows, int* __restrict__ n_long, int cap_long)
{
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x; r < n_rows; r += stride) {
        const int b = indptr[r], e = indptr[r + 1];
        double deg = (double)(e - b);
        if (wts) { deg = 0.0; for (int j = b; j < e; ++j) deg += (double)wts[j]; }        // adj.sum(1), model.py:189
        scale[r] = numer / fmax(deg, 1e-12);                                              // model.py:190 / :198 / :206
        if (e - b > kLongRow) { const int i = atomicAdd(n_long, 1); if (i < cap_long) long_rows[i] = (int)r; }
    }
}

__global__ void __launch_bounds__(256)
axpby_kernel(const float* __restrict__ a, float sa, float* __restrict__ out0, float* __restrict__ out1, long long n)
{
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float v = sa * a[i];
        out0[i] = v;
        if (out1) out1[i] = v;
    }
}

__global__ void __launch_bounds__(256) scale_kernel(float* __restrict__ a, float s, long long n)
{
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) a[i] *= s;
}

bool hip_ok(hipError_t e, const char* what) {
    if (e == hipSuccess) return true;
    gp_internal_set_error(GP_ERR_HIP, what, hipGetErrorString(e));
    return false;
}

}  // namespace

extern "C" {

int gp_propagate_features(gp_graph* g, const float* d_features, int32_t feat_dim, const float* d_edge_weight,
                          int mode, int order, double alpha, float* d_out, void* stream)
{
    if (!g || !d_features || !d_out) { gp_internal_set_error(GP_ERR_NULL, "gp_propagate_features", "null argument"); return GP_ERR_NULL; }
    if (feat_dim < 1 || order < 0 || mode < 0 || mode > 2 || !(alpha >= 0.0 && alpha <= 1.0)) {
        gp_internal_set_error(GP_ERR_INVALID_ARG, "gp_propagate_features", "feat_dim < 1, order < 0, mode not in {0 ppr,1 avg,2 single} or alpha outside [0,1]");
        return GP_ERR_INVALID_ARG;
    }
    const int device = gp_graph_device(g);
    const int64_t n = gp_graph_num_nodes(g);
    const int* indptr = nullptr; const int* indices = nullptr; uint32_t node_mask = 0;
    int rc = gp_internal_graph_csr(g, &indptr, &indices, &node_mask, stream);
    if (rc) return rc;
    if (!hip_ok(hipSetDevice(device), "hipSetDevice")) return GP_ERR_NO_DEVICE;
    hipStream_t s = (hipStream_t)stream;
    const size_t nf = (size_t)n * (size_t)feat_dim;
    if (n == 0) return GP_OK;

    // scratch: two ping-pong feature buffers, the row scales, the long-row list (freed before returning
    // control is not possible for an async call, so they are stream-ordered allocations)
    float *xa = nullptr, *xb = nullptr; double* scale = nullptr; int *long_rows = nullptr, *n_long_d = nullptr;
    // every exit path releases the scratch (stream-ordered frees: safe while the kernels are still queued)
    struct Scratch {
        hipStream_t s; void** p[5];
        ~Scratch() { for (void** q : p) if (*q) (void)hipFreeAsync(*q, s); }
    } scratch{s, {(void**)&xa, (void**)&xb, (void**)&scale, (void**)&long_rows, (void**)&n_long_d}};
    const int cap_long = 1 << 20;
    auto alloc = [&](void** ptr, size_t bytes) -> int {
        const hipError_t e = hipMallocAsync(ptr, bytes, s);
        if (e == hipSuccess) return GP_OK;
        (void)hipGetLastError();
        gp_internal_set_error(e == hipErrorOutOfMemory ? GP_ERR_NOMEM : GP_ERR_HIP, "hipMallocAsync", hipGetErrorString(e));
        return e == hipErrorOutOfMemory ? GP_ERR_NOMEM : GP_ERR_HIP;
    };
    if ((rc = alloc((void**)&xa, nf * sizeof(float))) || (rc = alloc((void**)&xb, nf * sizeof(float))) ||
        (rc = alloc((void**)&scale, (size_t)n * sizeof(double))) || (rc = alloc((void**)&long_rows, (size_t)cap_long * sizeof(int))) ||
        (rc = alloc((void**)&n_long_d, sizeof(int))))
        return rc;
    if (!hip_ok(hipMemsetAsync(n_long_d, 0, sizeof(int), s), "hipMemsetAsync")) return GP_ERR_HIP;

    const double numer = mode == 0 ? 1.0 - alpha : 1.0;
    hipLaunchKernelGGL(prepare_kernel, dim3(2048), dim3(256), 0, s, indptr, d_edge_weight, (long long)n, numer,
                       scale, long_rows, n_long_d, cap_long);
    int n_long = 0;
    if (!hip_ok(hipMemcpyAsync(&n_long, n_long_d, sizeof(int), hipMemcpyDeviceToHost, s), "hipMemcpyAsync") ||
        !hip_ok(hipStreamSynchronize(s), "hipStreamSynchronize"))
        return GP_ERR_HIP;
    if (n_long > cap_long) { gp_internal_set_error(GP_ERR_OVERFLOW, "gp_propagate_features", "more than 2^20 rows longer than 4096"); return GP_ERR_OVERFLOW; }

    // X0 (model.py:186-187 / :195 / :203) and the running sum
    const int accumulate = mode != 2;
    hipLaunchKernelGGL(axpby_kernel, dim3(4096), dim3(256), 0, s, d_features, mode == 0 ? (float)alpha : 1.0f, xa,
                       accumulate ? d_out : (float*)nullptr, (long long)nf);
    float* cur = xa; float* nxt = xb;
    const int vec = (feat_dim & 3) == 0 ? 4 : (feat_dim & 1) == 0 ? 2 : 1;
    int log2g = 0;
    while (log2g < 6 && (1 << log2g) * vec < feat_dim) ++log2g;               // G*VEC >= F, G <= 64
    const long long rows_per_block = (long long)(kBlock / 64) * (64 >> log2g);
    const int grid = (int)std::min<long long>((n + rows_per_block - 1) / rows_per_block, 256 * 32);
    for (int it = 0; it < order; ++it) {
        switch (vec) {
            case 4: hipLaunchKernelGGL(spmm_kernel<4>, dim3(grid), dim3(kBlock), 0, s, indptr, indices, d_edge_weight, node_mask, scale,
                                       (long long)n, cur, feat_dim, nxt, d_out, accumulate, log2g); break;
            case 2: hipLaunchKernelGGL(spmm_kernel<2>, dim3(grid), dim3(kBlock), 0, s, indptr, indices, d_edge_weight, node_mask, scale,
                                       (long long)n, cur, feat_dim, nxt, d_out, accumulate, log2g); break;
            default: hipLaunchKernelGGL(spmm_kernel<1>, dim3(grid), dim3(kBlock), 0, s, indptr, indices, d_edge_weight, node_mask, scale,
                                        (long long)n, cur, feat_dim, nxt, d_out, accumulate, log2g); break;
        }
        if (n_long > 0) {
            const int slabs = (feat_dim + 64 * vec - 1) / (64 * vec);
            const int lgrid = (int)std::min<long long>((long long)n_long * slabs, 4096);
            switch (vec) {
                case 4: hipLaunchKernelGGL(spmm_long_kernel<4>, dim3(lgrid), dim3(kLongBlock), 0, s, indptr, indices, d_edge_weight, node_mask,
                                           scale, long_rows, n_long, cur, feat_dim, nxt, d_out, accumulate); break;
                case 2: hipLaunchKernelGGL(spmm_long_kernel<2>, dim3(lgrid), dim3(kLongBlock), 0, s, indptr, indices, d_edge_weight, node_mask,
                                           scale, long_rows, n_long, cur, feat_dim, nxt, d_out, accumulate); break;
                default: hipLaunchKernelGGL(spmm_long_kernel<1>, dim3(lgrid), dim3(kLongBlock), 0, s, indptr, indices, d_edge_weight, node_mask,
                                            scale, long_rows, n_long, cur, feat_dim, nxt, d_out, accumulate); break;
            }
        }
        float* t = cur; cur = nxt; nxt = t;
    }
    if (mode == 1) hipLaunchKernelGGL(scale_kernel, dim3(4096), dim3(256), 0, s, d_out, 1.0f / (float)(order + 1), (long long)nf);   // model.py:201
    if (mode == 2) { if (!hip_ok(hipMemcpyAsync(d_out, cur, nf * sizeof(float), hipMemcpyDeviceToDevice, s), "hipMemcpyAsync")) return GP_ERR_HIP; }
    if (!hip_ok(hipGetLastError(), "propagate kernels")) return GP_ERR_HIP;
    return GP_OK;                                          // ~Scratch frees
}

}  // extern "C"
