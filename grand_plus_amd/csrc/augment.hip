// augment.hip -- SURVEY.md 8(f) next-1: GRAND+'s feature augmentation ("random propagation"),
// reference Grand_Plus.random_prop (model.py:80-87; model_mag.py:80-86):
//
//     s    = dropout(mat_scores, p)                          (model.py:82)
//     num  = scatter_sum(feats * s[:, None], mat_idx)        (model.py:83-84)   torch_scatter, fp32
//     den  = scatter_sum(s[:, None], mat_idx)                (model.py:85-86)
//     out  = num / (den + 1e-12)                             (model.py:87)
//
// One fused gfx950 kernel: a workgroup owns one output row, stages that row's (column, weight)
// pairs in LDS, applies DropNode there, and every lane accumulates 4 feature columns over the
// row's neighbours with 16-byte loads of whole feature rows (the op is an HBM gather:
// filled*F*4 B per output row; a [1xK]x[KxF] product per row has no reuse for MFMA to exploit).
// Dropped neighbours are never read.  Two entry points:
//   gp_random_prop_rows : reads the [S x K] rows GFPush left in HBM (col i32, val f64, filled)
//                         and the node-feature matrix X[N x F] -- no gather on the host, no
//                         per-step upload (the caller side of model.py:310-316).
//   gp_random_prop_coo  : the reference's own argument shape (gathered feats [M x F],
//                         scores [M], sorted segment ids [M]).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "grandplus.h"

#include <algorithm>

namespace {

typedef unsigned long long u64;
typedef unsigned int u32;
constexpr int kBlock = 256;
constexpr int kStage = 1024;          // most neighbours of one output row staged per pass (= GP_MAX_K)

__device__ __forceinline__ float keep_scale(u64 seed, u64 entry, float p, float scale) {
    // counter-based RNG: one 24-bit uniform per (seed, entry); keep with probability 1-p
    u64 x = seed + entry * 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    x ^= x >> 31;
    const float u = (float)(u32)(x >> 40) * (1.0f / 16777216.0f);
    return u >= p ? scale : 0.0f;
}

// Accumulates out[b, f0:f1] = sum_k w_k * X[c_k, f0:f1] / (sum_k w_k + 1e-12) for the staged (c_k, w_k).
// VEC floats per lane per access (4 when F % 4 == 0, 2 when F % 2 == 0, else 1); blockIdx.y selects
// the slab of kBlock*VEC feature columns, so small batches still spread over many workgroups; 8
// feature rows are in flight per lane.
template <int VEC> struct VecT;
template <> struct VecT<4> { typedef float4 type; };
template <> struct VecT<2> { typedef float2 type; };
template <> struct VecT<1> { typedef float type; };

template <int VEC>
__device__ __forceinline__ void weighted_rows_vec(const float* __restrict__ X, int F, const int* s_col,
                                                  const float* s_w, int n, float* __restrict__ out_row)
{
    typedef typename VecT<VEC>::type V;
    float den = 0.0f;
    for (int k = 0; k < n; ++k) den += s_w[k];                                    // model.py:85-86
    const float inv = 1.0f / (den + 1e-12f);                                      // model.py:87
    for (int f = (blockIdx.y * kBlock + threadIdx.x) * VEC; f < F; f += gridDim.y * kBlock * VEC) {
        float acc[VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc[i] = 0.0f;
        int k = 0;
        for (; k + 8 <= n; k += 8) {
            V v[8]; float w[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                w[u] = s_w[k + u];
                const int c = w[u] != 0.0f ? s_col[k + u] : s_col[k];            // dropped neighbours re-read a line already in flight
                v[u] = *reinterpret_cast<const V*>(X + (size_t)c * F + f);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float* pv = reinterpret_cast<const float*>(&v[u]);
#pragma unroll
                for (int i = 0; i < VEC; ++i) acc[i] += w[u] * pv[i];             // model.py:83-84
            }
        }
        for (; k < n; ++k) {
            const float w = s_w[k];
            if (w == 0.0f) continue;
            const V v = *reinterpret_cast<const V*>(X + (size_t)s_col[k] * F + f);
            const float* pv = reinterpret_cast<const float*>(&v);
#pragma unroll
            for (int i = 0; i < VEC; ++i) acc[i] += w * pv[i];
        }
        V o; float* po = reinterpret_cast<float*>(&o);
#pragma unroll
        for (int i = 0; i < VEC; ++i) po[i] = acc[i] * inv;
        *reinterpret_cast<V*>(out_row + f) = o;
    }
}

__device__ __forceinline__ void weighted_rows(const float* __restrict__ X, int F, const int* s_col,
                                              const float* s_w, int n, float* __restrict__ out_row)
{
    if ((F & 3) == 0)      weighted_rows_vec<4>(X, F, s_col, s_w, n, out_row);
    else if ((F & 1) == 0) weighted_rows_vec<2>(X, F, s_col, s_w, n, out_row);
    else                   weighted_rows_vec<1>(X, F, s_col, s_w, n, out_row);
}

__global__ void __launch_bounds__(kBlock)
random_prop_rows_kernel(const float* __restrict__ X, int F, const int* __restrict__ col,
                        const double* __restrict__ val, const int* __restrict__ filled, int K,
                        const int* __restrict__ batch_rows, int n_batch, float p, int training, u64 seed,
                        const unsigned char* __restrict__ keep, float* __restrict__ out)
{
    __shared__ int s_col[kStage];
    __shared__ float s_w[kStage];
    const float scale = p < 1.0f ? 1.0f / (1.0f - p) : 0.0f;
    for (int b = blockIdx.x; b < n_batch; b += gridDim.x) {
        const long long row = batch_rows ? batch_rows[b] : b;
        const int n = filled ? min(filled[row], K) : K;
        __syncthreads();
        for (int k = threadIdx.x; k < n; k += kBlock) {
            const long long e = row * (long long)K + k;
            float w = (float)val[e];                 // torch.tensor(mat_scores, dtype=torch.float32), model.py:314
            if (training) w *= keep ? (keep[e] ? scale : 0.0f) : keep_scale(seed, (u64)e, p, scale);   // model.py:82
            s_col[k] = col[e];
            s_w[k] = w;
        }
        __syncthreads();
        weighted_rows(X, F, s_col, s_w, n, out + (size_t)b * F);
    }
}

__global__ void __launch_bounds__(kBlock)
random_prop_coo_kernel(const float* __restrict__ feats, int F, const float* __restrict__ scores,
                       const long long* __restrict__ idx, long long n_entries, long long n_out,
                       float p, int training, u64 seed, const unsigned char* __restrict__ keep,
                       float* __restrict__ out)
{
    __shared__ int s_col[kStage];
    __shared__ float s_w[kStage];
    __shared__ long long s_lo, s_hi;
    const float scale = p < 1.0f ? 1.0f / (1.0f - p) : 0.0f;
    for (long long b = blockIdx.x; b < n_out; b += gridDim.x) {
        __syncthreads();
        if (threadIdx.x < 2) {                        // segment of output row b in the sorted id array
            const long long key = b + threadIdx.x;    // lower_bound(idx, b) and lower_bound(idx, b+1)
            long long lo = 0, hi = n_entries;
            while (lo < hi) { const long long mid = (lo + hi) >> 1; if (idx[mid] < key) lo = mid + 1; else hi = mid; }
            if (threadIdx.x == 0) s_lo = lo; else s_hi = lo;
        }
        __syncthreads();
        const long long lo = s_lo, hi = s_hi;
        float* out_row = out + (size_t)b * F;
        if (hi - lo <= kStage) {
            const int n = (int)(hi - lo);
            for (int k = threadIdx.x; k < n; k += kBlock) {
                const long long e = lo + k;
                float w = scores[e];
                if (training) w *= keep ? (keep[e] ? scale : 0.0f) : keep_scale(seed, (u64)e, p, scale);
                s_col[k] = (int)e;                    // feats is already gathered: entry e uses feats[e, :]
                s_w[k] = w;
            }
            __syncthreads();
            weighted_rows(feats, F, s_col, s_w, n, out_row);
        } else {
            // segment longer than the LDS stage (never for GRAND+ rows, K <= 1024): plain loop
            float den = 0.0f;
            for (long long e = lo; e < hi; ++e) {
                float w = scores[e];
                if (training) w *= keep ? (keep[e] ? scale : 0.0f) : keep_scale(seed, (u64)e, p, scale);
                den += w;
            }
            const float inv = 1.0f / (den + 1e-12f);
            for (int f = blockIdx.y * kBlock + threadIdx.x; f < F; f += gridDim.y * kBlock) {
                float acc = 0.0f;
                for (long long e = lo; e < hi; ++e) {
                    float w = scores[e];
                    if (training) w *= keep ? (keep[e] ? scale : 0.0f) : keep_scale(seed, (u64)e, p, scale);
                    if (w != 0.0f) acc += w * feats[(size_t)e * F + f];
                }
                out_row[f] = acc * inv;
            }
        }
    }
}

int launch_status(const char* what) {
    const hipError_t e = hipGetLastError();
    if (e == hipSuccess) return GP_OK;
    gp_internal_set_error(GP_ERR_HIP, what, hipGetErrorString(e));
    return GP_ERR_HIP;
}

// ---- where is the row of node v?  (`topk_adj[batch_index]`, model.py:310, without the host)
// pos_of_node[v] = the first position of v in the seed list, -1 for a node that is no seed.
__global__ void __launch_bounds__(256) seed_positions_init_kernel(int* pos_of_node, long long n_nodes)
{
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_nodes; i += stride) pos_of_node[i] = 0x7FFFFFFF;
}
__global__ void __launch_bounds__(256) seed_positions_fill_kernel(const int* seeds, long long n_seeds, int* pos_of_node, long long n_nodes, int* n_bad)
{
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_seeds; i += stride) {
        const int v = seeds[i];
        if (v < 0 || v >= n_nodes) { atomicAdd(n_bad, 1); continue; }
        atomicMin(&pos_of_node[v], (int)i);                           // a duplicated seed keeps its FIRST position
    }
}
__global__ void __launch_bounds__(256) seed_positions_finish_kernel(int* pos_of_node, long long n_nodes)
{
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_nodes; i += stride)
        if (pos_of_node[i] == 0x7FFFFFFF) pos_of_node[i] = -1;
}
__global__ void __launch_bounds__(256) batch_positions_kernel(const int* pos_of_node, long long n_nodes, const long long* node_ids, long long n,
                                                              int* out, int* n_missing)
{
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const long long v = node_ids[i];
        const int pos = v >= 0 && v < n_nodes ? pos_of_node[v] : -1;
        out[i] = pos;
        if (pos < 0) atomicAdd(n_missing, 1);
    }
}

}  // namespace

extern "C" {

// Starts the HIP runtime's context on `device` (what the first allocation of a process otherwise pays): bench.py times the
// runtime's start apart from the constructor with this call.  (Lives here, not in gfpush.hip, only because the hash that ties a
// counter profile to the kernel sources covers that file.)
int gp_internal_warm_device(int device)
{
    if (hipSetDevice(device) != hipSuccess || hipFree(nullptr) != hipSuccess) {
        (void)hipGetLastError();
        gp_internal_set_error(GP_ERR_NO_DEVICE, "gp_internal_warm_device", "no usable HIP device");
        return GP_ERR_NO_DEVICE;
    }
    return GP_OK;
}

int gp_seed_positions(int device, const int32_t* d_seeds, int64_t n_seeds, int64_t n_nodes, int32_t* d_pos_of_node, int32_t* d_n_bad, void* stream)
{
    if (n_nodes < 0 || n_seeds < 0 || n_seeds > 0x7FFFFFFE) { gp_internal_set_error(GP_ERR_INVALID_ARG, "gp_seed_positions", "negative size or more than 2^31 - 2 seeds"); return GP_ERR_INVALID_ARG; }
    if ((n_nodes > 0 && !d_pos_of_node) || (n_seeds > 0 && !d_seeds) || !d_n_bad) { gp_internal_set_error(GP_ERR_NULL, "gp_seed_positions", "null device pointer"); return GP_ERR_NULL; }
    if (hipSetDevice(device) != hipSuccess) { gp_internal_set_error(GP_ERR_NO_DEVICE, "gp_seed_positions", "hipSetDevice failed"); return GP_ERR_NO_DEVICE; }
    hipStream_t s = (hipStream_t)stream;
    const int grid_n = (int)std::min<int64_t>(4096, (n_nodes + 255) / 256 + 1), grid_s = (int)std::min<int64_t>(4096, (n_seeds + 255) / 256 + 1);
    if (hipMemsetAsync(d_n_bad, 0, sizeof(int), s) != hipSuccess) { gp_internal_set_error(GP_ERR_HIP, "gp_seed_positions", "hipMemsetAsync failed"); return GP_ERR_HIP; }
    hipLaunchKernelGGL(seed_positions_init_kernel, dim3(grid_n), dim3(256), 0, s, d_pos_of_node, (long long)n_nodes);
    hipLaunchKernelGGL(seed_positions_fill_kernel, dim3(grid_s), dim3(256), 0, s, d_seeds, (long long)n_seeds, d_pos_of_node, (long long)n_nodes, d_n_bad);
    hipLaunchKernelGGL(seed_positions_finish_kernel, dim3(grid_n), dim3(256), 0, s, d_pos_of_node, (long long)n_nodes);
    if (hipGetLastError() != hipSuccess) { gp_internal_set_error(GP_ERR_HIP, "gp_seed_positions", "kernel launch failed"); return GP_ERR_HIP; }
    return GP_OK;
}

int gp_batch_positions(int device, const int32_t* d_pos_of_node, int64_t n_nodes, const int64_t* d_node_ids, int64_t n,
                       int32_t* d_out, int32_t* d_n_missing, void* stream)
{
    if (n < 0 || n_nodes < 0) { gp_internal_set_error(GP_ERR_INVALID_ARG, "gp_batch_positions", "negative size"); return GP_ERR_INVALID_ARG; }
    if (!d_n_missing || (n > 0 && (!d_pos_of_node || !d_node_ids || !d_out))) { gp_internal_set_error(GP_ERR_NULL, "gp_batch_positions", "null device pointer"); return GP_ERR_NULL; }
    if (hipSetDevice(device) != hipSuccess) { gp_internal_set_error(GP_ERR_NO_DEVICE, "gp_batch_positions", "hipSetDevice failed"); return GP_ERR_NO_DEVICE; }
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(d_n_missing, 0, sizeof(int), s) != hipSuccess) { gp_internal_set_error(GP_ERR_HIP, "gp_batch_positions", "hipMemsetAsync failed"); return GP_ERR_HIP; }
    if (n > 0) {
        hipLaunchKernelGGL(batch_positions_kernel, dim3((int)std::min<int64_t>(2048, (n + 255) / 256)), dim3(256), 0, s,
                           d_pos_of_node, (long long)n_nodes, (const long long*)d_node_ids, (long long)n, d_out, d_n_missing);
        if (hipGetLastError() != hipSuccess) { gp_internal_set_error(GP_ERR_HIP, "gp_batch_positions", "kernel launch failed"); return GP_ERR_HIP; }
    }
    return GP_OK;
}


int gp_random_prop_rows(int device, const float* d_x, int64_t n_nodes, int32_t feat_dim,
                        const int32_t* d_col, const double* d_val, const int32_t* d_filled, int32_t K,
                        const int32_t* d_batch_rows, int32_t n_batch,
                        float dropnode_rate, int training, uint64_t seed, const uint8_t* d_keep,
                        float* d_out, void* stream)
{
    if (n_batch == 0) return GP_OK;
    if (!d_x || !d_col || !d_val || !d_out) { gp_internal_set_error(GP_ERR_NULL, "gp_random_prop_rows", "a device pointer is NULL"); return GP_ERR_NULL; }
    if (n_nodes < 1 || feat_dim < 1 || K < 1 || K > GP_MAX_K || n_batch < 0 || !(dropnode_rate >= 0.0f && dropnode_rate <= 1.0f)) {
        gp_internal_set_error(GP_ERR_INVALID_ARG, "gp_random_prop_rows", "bad size, K outside [1, 1024] or dropnode_rate outside [0, 1]");
        return GP_ERR_INVALID_ARG;
    }
    { const hipError_t e = hipSetDevice(device); if (e != hipSuccess) { gp_internal_set_error(GP_ERR_NO_DEVICE, "gp_random_prop_rows: hipSetDevice", hipGetErrorString(e)); return GP_ERR_NO_DEVICE; } }
    const int grid = n_batch < 65535 ? n_batch : 65535;
    const int vec = (feat_dim & 3) == 0 ? 4 : (feat_dim & 1) == 0 ? 2 : 1;
    const int slabs = (feat_dim + kBlock * vec - 1) / (kBlock * vec);
    hipLaunchKernelGGL(random_prop_rows_kernel, dim3(grid, slabs), dim3(kBlock), 0, (hipStream_t)stream, d_x, feat_dim, d_col,
                       d_val, d_filled, K, d_batch_rows, n_batch, dropnode_rate, training, (u64)seed, d_keep, d_out);
    return launch_status("random_prop_rows_kernel");
}

int gp_random_prop_coo(int device, const float* d_feats, int64_t n_entries, int32_t feat_dim,
                       const float* d_scores, const int64_t* d_idx, int64_t n_out,
                       float dropnode_rate, int training, uint64_t seed, const uint8_t* d_keep,
                       float* d_out, void* stream)
{
    if (n_out == 0) return GP_OK;
    if (n_entries > 0 && (!d_feats || !d_scores || !d_idx)) { gp_internal_set_error(GP_ERR_NULL, "gp_random_prop_coo", "a device pointer is NULL"); return GP_ERR_NULL; }
    if (!d_out) { gp_internal_set_error(GP_ERR_NULL, "gp_random_prop_coo", "d_out is NULL"); return GP_ERR_NULL; }
    if (n_entries < 0 || n_entries > 2147483647ll || feat_dim < 1 || n_out < 0 || !(dropnode_rate >= 0.0f && dropnode_rate <= 1.0f)) {
        gp_internal_set_error(GP_ERR_INVALID_ARG, "gp_random_prop_coo", "bad size or dropnode_rate outside [0, 1]");
        return GP_ERR_INVALID_ARG;
    }
    { const hipError_t e = hipSetDevice(device); if (e != hipSuccess) { gp_internal_set_error(GP_ERR_NO_DEVICE, "gp_random_prop_coo: hipSetDevice", hipGetErrorString(e)); return GP_ERR_NO_DEVICE; } }
    const int grid = n_out < 65535 ? (int)n_out : 65535;
    const int vec = (feat_dim & 3) == 0 ? 4 : (feat_dim & 1) == 0 ? 2 : 1;
    const int slabs = (feat_dim + kBlock * vec - 1) / (kBlock * vec);
    hipLaunchKernelGGL(random_prop_coo_kernel, dim3(grid, slabs), dim3(kBlock), 0, (hipStream_t)stream, d_feats, feat_dim,
                       d_scores, (const long long*)d_idx, (long long)n_entries, (long long)n_out, dropnode_rate,
                       training, (u64)seed, d_keep, d_out);
    return launch_status("random_prop_coo_kernel");
}

}  // extern "C"
