// gfpush_kernels.hpp -- hand-written HIP (gfx950 / CDNA4) kernels for GFPush.
//
// One persistent 64-lane-wave workgroup owns one seed ("row") at a time and pulls rows
// from a device-side queue (the reference's `omp parallel for schedule(dynamic)` over
// seeds, precompute/graph.h:73-74).  Per row it runs the level-synchronous push of
// graph.h:83-110 and the top-K of graph.h:111-126:
//
//   SCAN    walk the residue table of this level; for every node u with residue r:
//             reserve[u] += coef[lvl] * r                        (graph.h:90 / :109)
//             deg==0        -> r returns to the seed             (graph.h:91-93)
//             r >= rmax*deg -> append (CSR range, r/deg) to the push list  (graph.h:94-95)
//             else          -> r is dropped                      (no else branch)
//           clearing each slot as it goes; wave-level prefix sums compact the push list.
//   EXPAND  stream the CSR neighbour ranges of the push list (coalesced within a range)
//           and add r/deg into the NEXT level's residue table    (graph.h:96-99).
//   TOPK    radix-select the K largest reserve values (value desc, column asc) and
//           write row/col/value at slot row*K+rank                (graph.h:111-126).
//
// Residue table of a level: an open-addressing hash table {node -> residue}.  It lives in
// LDS (keys int32 + values fp64, 12 B/slot) whenever the level's edge count guarantees it
// fits (edges <= 0.7 * slots), otherwise in a per-workgroup table in HBM/L2 (16-B records).
// Residues are fp64 end to end, as in the reference (graph.h:76-77,95): the share r/deg is the
// reference's own fp64 quotient and the push test `r >= rmax*deg` (graph.h:94) sees the same
// number whenever a node has a single contribution -- which is what makes exact rational
// ties such as 1/deg(seed) == rmax*deg(u) fall the way the reference decides them.  Sums of
// several contributions use native fp64 atomic adds (ds_add_f64 / global_atomic_add_f64),
// so their last bits depend on arrival order exactly as the reference's depend on its
// hash-map iteration order.  Reserve values see one add per node per level, in level order.
//
// Reserve map of a row: a per-workgroup open-addressing table in HBM/L2 {node -> fp64} plus
// an insertion list (so top-K and clean-up cost O(support), not O(capacity)).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gp {

typedef unsigned long long u64;
typedef unsigned int u32;

constexpr int    kEmpty      = -1;
constexpr int    kSplitLen   = 256;     // a CSR range longer than this is split into chunks
constexpr int    kTopkBins   = 4096;    // 12-bit radix digits
constexpr int    kBucketCap  = 256;     // finish the select by ranking once <= this many remain
constexpr int    kCtlBytes   = 256;     // control block at the start of dynamic LDS
constexpr u32    kMinCap     = 1024;    // smallest table capacity used for a level

struct PushEntry { int start; int len; double share; };  // 16 B
struct ResRec    { int key;   int pad; double val; };    // 16 B  residue table record (HBM)
struct RsvRec    { int key;   int pad; double val; };    // 16 B  reserve table record (HBM)
struct Cand      { u64 bits;  int key; int pad;  };      // 16 B  top-K candidate

// Control block (lives in LDS, one per workgroup).
struct Ctl {
    long long row;        // row index pulled from the queue
    double dangling;      // mass returned to the seed by dangling nodes this level
    u32 n_dangling;       // how many dangling nodes were drained this level
    u32 n_push;           // push-list entries of this level
    u32 e_next;           // sum of their lengths = edges the next EXPAND will traverse
    u32 list_count;       // reserve-map insertions so far (= support)
    u32 n_cand;           // top-K candidates (value > 0)
    u32 fail;             // a bounded probe loop gave up / a list overflowed
    u32 n_sel;            // top-K: selected so far
    u32 n_bucket;         // top-K: members of the tie bucket
    u32 tk_bin;           // top-K: digit chosen this pass
    u32 tk_above;         // top-K: count strictly above the chosen digit this pass
    u32 tk_count;         // top-K: count inside the chosen digit
};

enum Counter { kQueue = 0, kPushes, kEdges, kFilled, kSupport, kFrontier, kLdsLevels,
               kGlobalLevels, kFailedRows, kNumCounters };

struct KParams {
    const int* indptr; const int* indices; int n_nodes;
    const int* seeds; long long n_seeds;
    const double* coef; int n_coef; double rmax; int K;
    int* out_row; int* out_col; double* out_val; int* out_filled;
    PushEntry* push; u64 push_cap;       // per-workgroup strides, in records
    ResRec* resg;    u64 resg_cap;
    RsvRec* rsv;     u64 rsv_cap;
    int* rsv_list;   u64 list_cap;
    Cand* cand;                           // stride = list_cap
    u64* counters;
    u32 lds_slots;
    int no_dangling;                      // 1 when every node has degree >= 1
    int force_global;
};

// ---------------------------------------------------------------- small helpers
__device__ __forceinline__ u32 hash_a(u32 k) {            // residue tables
    k *= 0x9E3779B1u; k ^= k >> 15; k *= 0x85EBCA77u; k ^= k >> 13;
    return k;
}
__device__ __forceinline__ u32 hash_b(u32 k) {            // reserve table (independent of hash_a)
    k ^= k >> 16; k *= 0x7FEB352Du; k ^= k >> 15; k *= 0x846CA68Bu; k ^= k >> 16;
    return k;
}
__device__ __forceinline__ u32 slot_of(u32 h, u32 cap) { return (u32)(((u64)h * cap) >> 32); }

// L2-coherent (L1-bypassing) accesses for tables that are also touched by atomics.
template <class T> __device__ __forceinline__ T ld_l2(const T* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <class T> __device__ __forceinline__ void st_l2(T* p, T v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ u32 wave_incl_scan(u32 x, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const u32 y = __shfl_up(x, d);
        if (lane >= d) x += y;
    }
    return x;
}
__device__ __forceinline__ u32 wave_suffix_scan(u32 x, int lane) {   // sum over lanes >= lane
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const u32 y = __shfl_down(x, d);
        if (lane + d < 64) x += y;
    }
    return x;
}
__device__ __forceinline__ u64 wave_sum64(u64 x) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) x += __shfl_down(x, d);
    return __shfl(x, 0);
}

// Every lane of the wave calls this with its item count n; returns the lane's first index
// in the list guarded by the LDS counter (one atomic per wave: ballot/prefix-sum compaction).
__device__ __forceinline__ u32 wave_alloc(u32* lds_counter, u32 n, int lane) {
    const u32 incl = wave_incl_scan(n, lane);
    const u32 total = __shfl(incl, 63);
    u32 base = 0;
    if (total != 0) {
        if (lane == 63)
            base = __hip_atomic_fetch_add(lds_counter, total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        base = __shfl(base, 63);
    }
    return base + incl - n;
}

// ---------------------------------------------------------------- residue tables
__device__ __forceinline__ bool res_add_lds(int* keys, double* vals, u32 cap, int k, double v) {
    u32 slot = slot_of(hash_a((u32)k), cap);
    for (u32 probe = 0; probe < cap; ++probe) {
        int cur = __hip_atomic_load(&keys[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (cur == kEmpty) {
            int expect = kEmpty;
            if (__hip_atomic_compare_exchange_strong(&keys[slot], &expect, k, __ATOMIC_RELAXED,
                                                     __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP))
                cur = k;
            else
                cur = expect;
        }
        if (cur == k) {
            __hip_atomic_fetch_add(&vals[slot], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            return true;
        }
        slot = (slot + 1 == cap) ? 0 : slot + 1;
    }
    return false;
}

__device__ __forceinline__ bool res_add_hbm(ResRec* tab, u32 cap, int k, double v) {
    u32 slot = slot_of(hash_a((u32)k), cap);
    for (u32 probe = 0; probe < cap; ++probe) {
        int cur = ld_l2(&tab[slot].key);
        if (cur == kEmpty) {
            int expect = kEmpty;
            if (__hip_atomic_compare_exchange_strong(&tab[slot].key, &expect, k, __ATOMIC_RELAXED,
                                                     __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
                cur = k;
            else
                cur = expect;
        }
        if (cur == k) {
            __hip_atomic_fetch_add(&tab[slot].val, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return true;
        }
        slot = (slot + 1 == cap) ? 0 : slot + 1;
    }
    return false;
}

// reserve[k] += add.  Exactly one thread touches a given key per level, so the value update
// is a plain read-modify-write; only the key claim needs a CAS.
// Returns 0 = updated, 1 = newly inserted (*new_slot set), -1 = table full.
__device__ __forceinline__ int rsv_add(RsvRec* tab, u32 cap, int k, double add, u32* new_slot) {
    u32 slot = slot_of(hash_b((u32)k), cap);
    for (u32 probe = 0; probe < cap; ++probe) {
        int cur = ld_l2(&tab[slot].key);
        if (cur == kEmpty) {
            int expect = kEmpty;
            if (__hip_atomic_compare_exchange_strong(&tab[slot].key, &expect, k, __ATOMIC_RELAXED,
                                                     __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                st_l2(&tab[slot].val, add);                 // 0.0 + add == add exactly
                *new_slot = slot;
                return 1;
            }
            cur = expect;
        }
        if (cur == k) {
            const double old = ld_l2(&tab[slot].val);
            st_l2(&tab[slot].val, old + add);
            return 0;
        }
        slot = (slot + 1 == cap) ? 0 : slot + 1;
    }
    return -1;
}

// ---------------------------------------------------------------- SCAN
template <int BLOCK, bool IN_LDS>
__device__ __forceinline__ void scan_level(const KParams& p, Ctl* ctl, int* lkeys, double* lvals,
                                           ResRec* resg, u32 cap, RsvRec* rsv, int* rlist,
                                           PushEntry* push, double c, bool do_push,
                                           u64& st_push, u64& st_edges, u64& st_front)
{
    const int tid = threadIdx.x, lane = tid & 63;
    const u32 rsv_cap = (u32)p.rsv_cap;
    for (u32 base = 0; base < cap; base += BLOCK) {
        const u32 slot = base + tid;
        int k = kEmpty;
        double r = 0.0;
        if (slot < cap) {
            if (IN_LDS) {
                k = lkeys[slot];
                if (k != kEmpty) { r = lvals[slot]; lkeys[slot] = kEmpty; lvals[slot] = 0.0; }
            } else {
                k = ld_l2(&resg[slot].key);
                if (k != kEmpty) {
                    r = ld_l2(&resg[slot].val);
                    st_l2(&resg[slot].key, kEmpty);
                    st_l2(&resg[slot].val, 0.0);
                }
            }
        }
        u32 n_new = 0, new_slot = 0, n_chunks = 0;
        int e_start = 0, e_len = 0;
        double e_share = 0.0;
        if (k != kEmpty) {
            ++st_front;
            const int rc = rsv_add(rsv, rsv_cap, k, c * r, &new_slot);      // graph.h:90 / :109
            if (rc > 0) n_new = 1;
            else if (rc < 0) ctl->fail = 1;
            // deg >= 1 everywhere => a node with r < rmax can neither push nor be dangling
            if (do_push && (r >= p.rmax || !p.no_dangling)) {
                const int s = p.indptr[k], e = p.indptr[k + 1];
                const u32 deg = (u32)(e - s);                               // graph.h:43-45
                if (deg == 0) {                                             // graph.h:91-93
                    __hip_atomic_fetch_add(&ctl->dangling, r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_fetch_add(&ctl->n_dangling, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                } else if (r >= p.rmax * (double)deg) {                     // graph.h:94
                    ++st_push; st_edges += deg;
                    e_share = r / (double)deg;                              // graph.h:95
                    if (e_share != 0.0) {
                        e_start = s; e_len = (int)deg;
                        n_chunks = (deg + kSplitLen - 1) / kSplitLen;
                    }
                }
            }
        }
        // wave-level compaction of both lists (prefix sums over the 64 lanes)
        const u32 li = wave_alloc(&ctl->list_count, n_new, lane);
        if (n_new) {
            if (li < p.list_cap) rlist[li] = (int)new_slot; else ctl->fail = 1;
        }
        const u32 pi = wave_alloc(&ctl->n_push, n_chunks, lane);
        if (n_chunks) {
            __hip_atomic_fetch_add(&ctl->e_next, (u32)e_len, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if ((u64)pi + n_chunks <= p.push_cap) {
                for (u32 j = 0; j < n_chunks; ++j) {
                    PushEntry pe;
                    pe.start = e_start + (int)j * kSplitLen;
                    pe.len = min(kSplitLen, e_len - (int)j * kSplitLen);
                    pe.share = e_share;
                    push[pi + j] = pe;
                }
            } else {
                ctl->fail = 1;
            }
        }
    }
}

// ---------------------------------------------------------------- EXPAND
// G lanes (a power of two, 4..64) cooperate on one push-list entry: lanes read consecutive
// column ids of the CSR range (coalesced) and add the share into the next residue table.
template <int BLOCK, bool IN_LDS>
__device__ __forceinline__ void expand_level(const KParams& p, Ctl* ctl, int* lkeys, double* lvals,
                                             ResRec* resg, u32 cap, const PushEntry* push,
                                             u32 n_push, int log2g)
{
    const int tid = threadIdx.x;
    const int G = 1 << log2g;
    const int gl = tid & (G - 1);
    const u32 gid = (u32)tid >> log2g;
    const u32 n_groups = (u32)BLOCK >> log2g;
    bool ok = true;
    for (u32 e = gid; e < n_push; e += n_groups) {
        const PushEntry pe = push[e];
        const int* nbr = p.indices + pe.start;
        for (int j = gl; j < pe.len; j += G) {
            const int v = nbr[j];                                           // graph.h:97
            if (IN_LDS) ok &= res_add_lds(lkeys, lvals, cap, v, pe.share);  // graph.h:98
            else        ok &= res_add_hbm(resg, cap, v, pe.share);
        }
    }
    if (!ok) ctl->fail = 1;
}

// ---------------------------------------------------------------- TOP-K
// Candidates are ordered by the 96-bit composite (value bits, ~column): larger composite =
// larger value, ties broken towards the SMALLER column id.  Composites are unique per row.
typedef unsigned __int128 u128;
__device__ __forceinline__ u128 composite(const Cand& c) {
    return ((u128)c.bits << 32) | (u128)(u32)(~(u32)c.key);
}

template <int BLOCK>
__device__ __forceinline__ void topk_row(const KParams& p, Ctl* ctl, unsigned char* scratch,
                                         RsvRec* rsv, const int* rlist, Cand* cand,
                                         long long row, int seed, u64& st_filled)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    u32*  hist   = (u32*)scratch;                                    // [kTopkBins]
    Cand* sel    = (Cand*)(scratch + kTopkBins * sizeof(u32));       // [K]
    Cand* bucket = sel + p.K;                                        // [kBucketCap]
    const u32 n = ctl->list_count;
    const u32 K = (u32)p.K;

    // pass 0: gather (column, value) of the reserve map, clean the table, keep value > 0
    for (u32 base = 0; base < n; base += BLOCK) {
        const u32 i = base + tid;
        u32 keep = 0;
        Cand c; c.bits = 0; c.key = 0; c.pad = 0;
        if (i < n) {
            const u32 slot = (u32)rlist[i];
            c.key = ld_l2(&rsv[slot].key);
            const double v = ld_l2(&rsv[slot].val);
            st_l2(&rsv[slot].key, kEmpty);
            if (v > 0.0) { c.bits = (u64)__double_as_longlong(v); keep = 1; }    // graph.h:121
        }
        const u32 ci = wave_alloc(&ctl->n_cand, keep, lane);
        if (keep) cand[ci] = c;
    }
    __syncthreads();
    const u32 m = ctl->n_cand;
    const u32 need = m < K ? m : K;                                   // graph.h:113
    if (need == 0) {
        if (tid == 0 && p.out_filled) p.out_filled[row] = 0;
        return;
    }

    if (m <= K) {
        for (u32 i = tid; i < m; i += BLOCK) sel[i] = cand[i];
        __syncthreads();
    } else {
        // MSD radix select on the composite, 12 bits per pass, early exit on a small bucket
        u128 prefix = 0;
        u32 want = K;               // how many must still come from the current bucket
        int depth = 0;              // digits fixed so far
        bool take_all_bucket = false;
        for (;;) {
            for (u32 i = tid; i < kTopkBins; i += BLOCK) hist[i] = 0;
            __syncthreads();
            const int shift = 96 - 12 * (depth + 1);
            for (u32 i = tid; i < m; i += BLOCK) {
                const u128 comp = composite(cand[i]);
                if (depth == 0 || (comp >> (shift + 12)) == prefix)
                    __hip_atomic_fetch_add(&hist[(u32)(comp >> shift) & (kTopkBins - 1)], 1u,
                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            __syncthreads();
            if (wave == 0) {
                // chunk sums: lane owns bins [64*lane, 64*lane+64); skewed reads avoid bank conflicts
                u32 csum = 0;
                for (int j = 0; j < 64; ++j) csum += hist[64 * lane + ((j + lane) & 63)];
                const u32 csuf = wave_suffix_scan(csum, lane);              // bins >= 64*lane
                const u64 cm = __ballot(csuf >= want);
                const int cl = __popcll(cm) - 1;                            // chunk holding the want-th largest
                const u32 above_c = __shfl(csuf, cl) - __shfl(csum, cl);    // bins above that chunk
                const u32 b = hist[64 * cl + lane];
                const u32 bsuf = wave_suffix_scan(b, lane) + above_c;       // bins >= this bin
                const u64 bm = __ballot(bsuf >= want);
                const int bl = __popcll(bm) - 1;
                if (lane == bl) {
                    ctl->tk_bin = (u32)(64 * cl + bl);
                    ctl->tk_above = bsuf - b;
                    ctl->tk_count = b;
                }
            }
            __syncthreads();
            prefix = (prefix << 12) | (u128)ctl->tk_bin;
            want -= ctl->tk_above;
            const u32 cnt = ctl->tk_count;
            ++depth;
            __syncthreads();
            if (cnt == want) { take_all_bucket = true; break; }
            if (cnt <= (u32)kBucketCap || depth == 8) break;
        }
        // collect: strictly above the prefix -> selected; equal to the prefix -> tie bucket
        const int shift = 96 - 12 * depth;
        for (u32 base = 0; base < m; base += BLOCK) {
            const u32 i = base + tid;
            u32 is_sel = 0, is_b = 0;
            Cand c; c.bits = 0; c.key = 0; c.pad = 0;
            if (i < m) {
                c = cand[i];
                const u128 pre = composite(c) >> shift;
                if (pre > prefix || (take_all_bucket && pre == prefix)) is_sel = 1;
                else if (pre == prefix) is_b = 1;
            }
            const u32 si = wave_alloc(&ctl->n_sel, is_sel, lane);
            if (is_sel) sel[si] = c;
            const u32 bi = wave_alloc(&ctl->n_bucket, is_b, lane);
            if (is_b && bi < (u32)kBucketCap) bucket[bi] = c;
        }
        __syncthreads();
        if (!take_all_bucket) {
            const u32 nb = min(ctl->n_bucket, (u32)kBucketCap);
            const u32 n_sel0 = ctl->n_sel;                     // == K - want
            for (u32 i = tid; i < nb; i += BLOCK) {
                const u128 mine = composite(bucket[i]);
                u32 rank = 0;
                for (u32 j = 0; j < nb; ++j) rank += composite(bucket[j]) > mine ? 1u : 0u;
                if (rank < want) sel[n_sel0 + rank] = bucket[i];
            }
            __syncthreads();
        }
    }
    // order the selected `need` entries (value desc, column asc) and write the row
    const long long out0 = row * (long long)p.K;
    for (u32 i = tid; i < need; i += BLOCK) {
        const Cand c = sel[i];
        const u128 mine = composite(c);
        u32 rank = 0;
        for (u32 j = 0; j < need; ++j) rank += composite(sel[j]) > mine ? 1u : 0u;
        p.out_row[out0 + rank] = seed;                                           // graph.h:122
        p.out_col[out0 + rank] = c.key;                                          // graph.h:123
        p.out_val[out0 + rank] = __longlong_as_double((long long)c.bits);        // graph.h:124
    }
    if (tid == 0) {
        if (p.out_filled) p.out_filled[row] = (int)need;
        st_filled += need;
    }
}

// ---------------------------------------------------------------- the kernel
template <int BLOCK>
__global__ void __launch_bounds__(BLOCK) gfpush_kernel(const KParams p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    Ctl* ctl   = (Ctl*)smem;
    double* lvals = (double*)(smem + kCtlBytes);
    int* lkeys = (int*)(smem + kCtlBytes + 8 * (size_t)p.lds_slots);
    const int tid = threadIdx.x;
    const u32 C = p.lds_slots;

    const size_t wg = blockIdx.x;
    PushEntry* push = p.push + wg * p.push_cap;
    ResRec* resg    = p.resg + wg * p.resg_cap;
    RsvRec* rsv     = p.rsv + wg * p.rsv_cap;
    int* rlist      = p.rsv_list + wg * p.list_cap;
    Cand* cand      = p.cand + wg * p.list_cap;

    for (u32 i = tid; i < C; i += BLOCK) { lkeys[i] = kEmpty; lvals[i] = 0.0; }
    u64 st_push = 0, st_edges = 0, st_front = 0, st_filled = 0, st_support = 0,
        st_lds = 0, st_glb = 0, st_failed = 0;
    const int L = p.n_coef - 1;

    for (;;) {
        __syncthreads();
        if (tid == 0) {
            ctl->row = (long long)__hip_atomic_fetch_add(&p.counters[kQueue], 1ull, __ATOMIC_RELAXED,
                                                         __HIP_MEMORY_SCOPE_AGENT);
            ctl->list_count = 0; ctl->n_cand = 0; ctl->fail = 0;
            ctl->n_sel = 0; ctl->n_bucket = 0;
        }
        __syncthreads();
        const long long row = ctl->row;
        if (row >= p.n_seeds) break;
        const int seed = p.seeds[row];
        if (seed < 0 || seed >= p.n_nodes) {            // device API does not pre-validate seeds
            if (tid == 0) { ++st_failed; if (p.out_filled) p.out_filled[row] = 0; }
            continue;
        }

        // level-0 frontier = { seed : 1.0 }                                   graph.h:81
        bool in_lds = !p.force_global;
        u32 cap = in_lds ? min(C, kMinCap) : (u32)min((u64)kMinCap, p.resg_cap);
        if (tid == 0) {
            const u32 s0 = slot_of(hash_a((u32)seed), cap);
            if (in_lds) { lkeys[s0] = seed; lvals[s0] = 1.0; }
            else { st_l2(&resg[s0].key, seed); st_l2(&resg[s0].val, 1.0); }
        }
        __syncthreads();

        for (int lvl = 0;; ++lvl) {
            if (tid == 0) { ctl->n_push = 0; ctl->e_next = 0; ctl->dangling = 0.0; ctl->n_dangling = 0; }
            __syncthreads();
            const double c = p.coef[lvl];
            const bool do_push = lvl < L;                                     // graph.h:83 vs :104
            if (in_lds) scan_level<BLOCK, true >(p, ctl, lkeys, lvals, resg, cap, rsv, rlist, push, c, do_push, st_push, st_edges, st_front);
            else        scan_level<BLOCK, false>(p, ctl, lkeys, lvals, resg, cap, rsv, rlist, push, c, do_push, st_push, st_edges, st_front);
            if (tid == 0) { if (in_lds) ++st_lds; else ++st_glb; }
            __syncthreads();
            if (!do_push || ctl->fail) break;
            const u32 n_push = ctl->n_push;
            const double dang = ctl->dangling;
            const bool has_dang = ctl->n_dangling != 0;
            // distinct targets of the next level <= min(edges (+ the seed), N)
            const u64 need = min((u64)ctl->e_next + (has_dang ? 1 : 0), (u64)p.n_nodes);
            if (need == 0) break;                       // the frontier died: later levels add nothing
            // next level's table: LDS iff it is guaranteed to fit (distinct targets <= edges)
            in_lds = !p.force_global && need * 10 <= (u64)C * 7;
            if (in_lds) {
                cap = (u32)min((u64)C, max((u64)kMinCap, 4 * need));
            } else {
                if (2 * need > p.resg_cap) { if (tid == 0) ctl->fail = 1; __syncthreads(); break; }
                cap = (u32)min(p.resg_cap, max((u64)kMinCap, 2 * need));
            }
            int log2g = 2;                               // lanes per entry ~ mean range length
            if (n_push) { const u32 avg = ctl->e_next / n_push; while (log2g < 6 && (1u << log2g) < avg) ++log2g; }
            if (in_lds) expand_level<BLOCK, true >(p, ctl, lkeys, lvals, resg, cap, push, n_push, log2g);
            else        expand_level<BLOCK, false>(p, ctl, lkeys, lvals, resg, cap, push, n_push, log2g);
            if (tid == 0 && has_dang) {                                        // graph.h:92
                const bool ok = in_lds ? res_add_lds(lkeys, lvals, cap, seed, dang)
                                       : res_add_hbm(resg, cap, seed, dang);
                if (!ok) ctl->fail = 1;
            }
            __syncthreads();
            if (ctl->fail) break;
        }
        __syncthreads();
        if (ctl->fail) {
            // Leave the row unwritten and report it; tables may be dirty -> host re-initialises.
            if (tid == 0) { ++st_failed; if (p.out_filled) p.out_filled[row] = 0; }
            // best-effort clean-up so that later rows of this workgroup are not corrupted
            for (u32 i = tid; i < C; i += BLOCK) { lkeys[i] = kEmpty; lvals[i] = 0.0; }
            continue;
        }
        if (tid == 0) st_support += ctl->list_count;
        topk_row<BLOCK>(p, ctl, smem + kCtlBytes, rsv, rlist, cand, row, seed, st_filled);
        __syncthreads();
        // top-K used the table region as scratch: restore the empty LDS table
        for (u32 i = tid; i < C; i += BLOCK) { lkeys[i] = kEmpty; lvals[i] = 0.0; }
    }

    // flush statistics: one atomic per counter per workgroup
    st_push = wave_sum64(st_push); st_edges = wave_sum64(st_edges); st_front = wave_sum64(st_front);
    __syncthreads();
    u64* red = (u64*)(smem + kCtlBytes);
    if (tid < 8) red[tid] = 0;
    __syncthreads();
    if ((tid & 63) == 0) {
        __hip_atomic_fetch_add(&red[0], st_push, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add(&red[1], st_edges, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add(&red[2], st_front, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __syncthreads();
    if (tid == 0) {
        __hip_atomic_fetch_add(&p.counters[kPushes], red[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(&p.counters[kEdges], red[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(&p.counters[kFrontier], red[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(&p.counters[kFilled], st_filled, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(&p.counters[kSupport], st_support, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(&p.counters[kLdsLevels], st_lds, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(&p.counters[kGlobalLevels], st_glb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(&p.counters[kFailedRows], st_failed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// Fills the per-workgroup HBM tables with empty records (a byte memset cannot: val must be 0).
__global__ void __launch_bounds__(256) init_tables_kernel(ResRec* resg, u64 n_res, RsvRec* rsv, u64 n_rsv)
{
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n_res; i += stride) {
        ResRec r; r.key = kEmpty; r.pad = 0; r.val = 0.0; resg[i] = r;
    }
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n_rsv; i += stride) {
        RsvRec r; r.key = kEmpty; r.pad = 0; r.val = 0.0; rsv[i] = r;
    }
}

// Minimum of indptr differences == 0 ?  (sets *flag to 1 if some node has degree 0)
__global__ void __launch_bounds__(256) dangling_probe_kernel(const int* indptr, long long n, int* flag)
{
    const long long stride = (long long)gridDim.x * blockDim.x;
    int found = 0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        found |= (indptr[i + 1] == indptr[i]);
    if (found) *flag = 1;
}

}  // namespace gp
