// gfpush_sketch.hpp -- the sketch-filtered GFPush kernel for gfx950: exact residues only where they can matter.
//
// Same contract as gfpush_kernels.hpp (reference precompute/graph.h:73-126, one persistent workgroup per row), different
// structure.  Two facts about the shipped recipes (rmax >= 5e-6 on large graphs; tools/sim/*.cpp measured them on the
// MAG shape) carry it:
//
//   * Only ~8 % of a level's frontier nodes pass the push test r >= rmax*deg (graph.h:94); the other 92 % only
//     deposit coef*r into the reserve (graph.h:90) and are dropped.  And the reserve is LINEAR in the pushed shares:
//     reserve[v] = sum over levels and in-edges of coef[l] * share(u).  So a pushed edge (u -> v) needs an EXACT fp64
//     accumulator for v only if v may push; everything else is one 6-byte log record per edge.
//   * Whether v may push is decided by a KEYLESS upper bound: the level SKETCH U[h(v)] += ceil(share * 2^31) -- one
//     fire-and-forget ds_add_u32 per edge, no key, no compare-and-swap, no probing, 4 bytes per cell instead of 12 per
//     slot.  Collisions only ever ADD, so U[h(v)] * 2^-31 >= r(v), and v can push only if its cell reaches
//     rmax * min(deg, deg_sat) (the degree rides in the packed column word).  On the MAG shape a 8 192-cell sketch lets
//     23 % of the edges through to the exact table (the true pushers own 16 %).
//
// Round 5 -- bytes and dependent round trips (13.8 x the algorithmic bytes at the fabric, 76 % of what the part streams):
//   * The kernel runs on the SELF-ADDRESSED CSR (gfpush.hip:ensure_acsr): rows start at 128-byte units and a column word holds
//     the UNIT NUMBER of its target under the packed degree.  A node that pushes needs no indptr lookup -- its columns start at
//     unit * 32 and its degree is in its key (the rare saturated degree is one word of unit_info) -- which removes one 128-byte
//     line per pusher (0.23 MB of a MAG row's 1.6 MB) and the dependent round trip in the middle of every SCAN; a run of <= 32
//     columns is one line (2 123 lines per MAG row instead of 2 577 + 1 844, tools/sim/layout_sim.cpp).  Unit numbers grow with
//     node ids, so the output order (value desc, column asc) is decided on keys; the K output columns are translated back.
//   * The log record is 6 bytes: the packed column word and the 16-bit NUMBER of the pusher inside the row.  Who needs the
//     fp64 share looks it up in LDS: FILTER in the level's share table S (written by STREAM from the push-list entries its waves
//     hold anyway; it borrows the tail of the exact table's value array), TOP-K in T = coef * share of every pusher of the row
//     (8 bytes per pusher in HBM, written by SCAN with the entry; ~1 800 per MAG row, staged in the dead level tables).
//     Round 4's attempt at 6-byte records lost because the share was then a dependent GLOBAL gather in front of every group.
//
// Per level l = 1..L of a row (graph.h:83-110):
//   STREAM  sk_edge_stream over the push list (one lane per edge): the packed column word and the pusher number go to the
//           reserve LOG at position (level base + edge number) -- no allocation, fully coalesced --, ceil(share * 2^31) into
//           the level sketch U and ceil(coef[l] * share * scale) into the row's RESERVE SKETCH R (see TOP-K), both
//           fire-and-forget LDS adds.  Small levels skip U and insert straight into the exact table.
//   FILTER  re-reads the level's log segment (L2-hot, next group of 256 records in flight while one is processed), looks
//           every edge's cell up and inserts the edges whose target may push into the exact table X (insert_window_asm:
//           LDS hash insert, fp64 atomic add).  More candidates than X holds are walked in hash partitions.
//   SCAN    drains X: exact push test with the degree in the key, dangling rule (graph.h:91-93), fp64 division, next push
//           list (sk_push_alloc) and the pushers' coef * share.
// TOP-K (graph.h:111-126), with the level tables dead and their LDS free:
//   R[h(v)] += ceil(coef[level] * share * scale) over every pushed edge is an upper bound on every node's reserve, so the
//   K largest totals live in heavy cells.  The cell value t_c of rank ~4K is read off a histogram of R, one sweep over the log
//   sums (exactly, fp64, keyed table) the records whose cell reaches t_c, and if the K-th largest exact total tau satisfies
//   tau*scale >= t_c no unswept node can beat it: done (95 % of MAG rows; ~500 nodes tabled instead of the 12 500 of the
//   support).  Otherwise tau is a proven lower bound and one more sweep with t_c = tau*scale is complete by construction;
//   when its nodes outgrow the table it runs in hash partitions, each partition's K best merged into a running list.
//
// What is left over -- a workspace bound, more than 64 partitions, totals outside [2^-63, 2), more pushers than fit LDS --
// sends the row to the retry list, and the general kernel (gfpush_retry_kernel) runs it.  Not counted here: frontier /
// support sizes (no structure sees distinct targets any more); `exact_stats` selects the general kernel.
#pragma once

#include "gfpush_kernels.hpp"

#ifndef GP_DIAG      // the diagnostic build instruments the general kernel only

namespace gp {

#ifndef GP_SK_SWEEP_DEPTH
#define GP_SK_SWEEP_DEPTH 2               // groups of log records in flight per wave in TOP-K's sweep
#endif
constexpr int kSkMaxCoef = 40;            // levels the control block has room for (longer recipes: general kernel)
constexpr u32 kSkTie     = 256;           // the select ranks at most this many candidates by comparison
constexpr u32 kSkMul24   = 0x9E3779u;     // sketch hash: cell = top bits of (low 24 bits of the key) * kSkMul24 -- a FULL-RATE multiply (sk_cell)

struct CtlS {
    long long row;
    LevelCtr lc[2];                       // what SCAN of level l produces for level l+1 lives in lc[l & 1] (n_rec: nodes it drained)
    u32 ovf;                              // an exact table / aggregation overflowed
    u32 fail;                             // the row leaves for the retry list (1: it outgrew a slab, 3: the select gave up, 4 / 5: too many partitions)
    u32 n_sel, n_tie;                     // select: entries above the K-th bin / inside it
    u32 tk_bin, tk_above, tk_count, tk_total;
    u32 tk_t, tk_wide, tk_dig;
    u64 kth_bits;                         // smallest selected value (bit pattern)
    u32 bcnt[64];                         // select: binade counters
    u64 st[8], st_row[8];                 // statistics: workgroup totals / the row in flight
    double coef[kSkMaxCoef + 1];           // ([n_coef .. kSkMaxCoef]: never used -- a level reads coef[level + 1] unconditionally)
    u32 cand_q[kSkMaxCoef];               // per level: exact-table nodes per pushed edge of this workgroup's earlier rows (x 1.25, in 1/1024)
    int seed_key; u32 tot_pu, tot_log;    // the row in flight: its seed's key; pushers numbered and log records written so far
    double lv_dang; u32 lv_has_dang;      // the level in flight: mass its dangling nodes return to the seed (thread 0 writes them at the top of the level and is their only reader: STREAM's tail)
    u32 max_e, max_log;                   // largest level / log this workgroup has seen (statistics)
    // launch constants the level loop needs, derived ONCE per workgroup (gfpush_sk_rows): read in the same batch of LDS reads as the
    // level's state.  (From the kernel arguments they were five scalar loads per level and wave, each behind its own wait -- the
    // compiler keeps the ADDRESSES of kernel arguments across calls, not their values -- and a fp64 division for the one-wave test.)
    u32 k_log_cap, k_pu_cap, k_direct_max, k_solo_ok, k_cx, k_pad; double k_rscale;
#ifdef GP_SK_TIMING
    u64 tacc[16]; u64 tlast;              // -DGP_SK_TIMING: 100 MHz ticks thread 0 spent per phase (flushed to the diag_sub counters)
    u64 tacc2[14]; u64 tlast2;            // ... and inside FILTER / SCAN / STREAM (tools/sk_phases.py)
#endif
};
// -DGP_SK_TIMING (tools/sk_phases.py): thread 0 stamps the phases with the constant 100 MHz clock.  [0] row prologue + level 0,
// [1] STREAM of small levels (exact inserts), [2] STREAM of sketch levels, [3] STREAM of the last level, [4] FILTER, [5] SCAN,
// [6] rest of the level loop, [7] TOP-K: R + threshold, [8] TOP-K sweeps, [9] select, [10] output + row end, [11] filter calls,
// [12] scan calls, [13] small levels, [14] sketch levels, [15] rows.  Barriers are charged to the phase in front of them.
#ifdef GP_SK_TIMING
#define SKT_BEGIN(ctl) do { if (threadIdx.x == 0) (ctl)->tlast = wall_clock64(); } while (0)
#define SKT(ctl, i) do { if (threadIdx.x == 0) { const u64 n_ = wall_clock64(); (ctl)->tacc[i] += n_ - (ctl)->tlast; (ctl)->tlast = n_; } } while (0)
#define SKT_COUNT(ctl, i, n) do { if (threadIdx.x == 0) (ctl)->tacc[i] += (n); } while (0)
#define SKT2_BEGIN(ctl) do { if (threadIdx.x == 0) (ctl)->tlast2 = wall_clock64(); } while (0)
#define SKT2(ctl, i) do { if (threadIdx.x == 0) { const u64 n_ = wall_clock64(); (ctl)->tacc2[i] += n_ - (ctl)->tlast2; (ctl)->tlast2 = n_; } } while (0)
#else
#define SKT_BEGIN(ctl) do { } while (0)
#define SKT(ctl, i) do { } while (0)
#define SKT_COUNT(ctl, i, n) do { } while (0)
#define SKT2_BEGIN(ctl) do { } while (0)
#define SKT2(ctl, i) do { } while (0)
#endif
static_assert(sizeof(CtlS) <= (size_t)kCtlStruct, "the control block must fit its LDS reservation");
enum StatS { zPush = 0, zEdges, zDeg, zFilled, zLevels, zFailed, zCand, zSweep2, zNumStats };
__device__ __forceinline__ void zstat(CtlS* ctl, int which, u64 n) {
    __hip_atomic_fetch_add(&ctl->st_row[which], n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// LDS of a workgroup: CtlS + the per-wave flag bytes (kCtlBytes, as in the general kernel) | R u32[MR] | U u32[MU] |
// X values f64[CX] | X keys i32[CX].  A sketch level whose list holds n pushers uses the first CX - (n + 1) slots of X and keeps the
// level's SHARE TABLE S f64[n + 1] in the value array behind them ([n]: the mass dangling nodes returned to the seed).
// TOP-K re-uses U and X as { T f64[pushers of the row] | aggregation values f64[CA] | keys i32[CA] | tie Cand[256] | sel Cand[K] }.
struct SkView {
    CtlS* ctl; u32* R; u32* U; double* xvals; int* xkeys; u32 MU, MR, CX, shU, shR;
    PushEntry* push2; u32* bt2; int* log_key; unsigned short* log_pu; double* arch;
    u32 lds_u;                            // byte offset of U inside LDS
    u32* bt_l; u32 bt_l_cap;              // the head of the current boundary table, in the flag bytes no wave of this block size owns
};
// (BLOCK is a template argument, not blockDim.x: the latter is a 16-bit word of the hidden kernel arguments that only a VECTOR load
//  reads -- one global round trip at the top of every phase that asked for it.)
template <int BLOCK>
__device__ __forceinline__ SkView sk_view(KP p, u32 lds0) {
    SkView w;
    w.MU = 1u << p.sk_lg_mu; w.MR = 1u << p.sk_lg_mr; w.CX = p.sk_cx;
    w.shU = 32u - p.sk_lg_mu; w.shR = 32u - p.sk_lg_mr;
    w.ctl = lds_at<CtlS>(lds0);
    w.R = lds_at<u32>(lds0 + (u32)kCtlBytes);
    w.lds_u = lds0 + (u32)kCtlBytes + 4u * w.MR;
    w.U = lds_at<u32>(w.lds_u);
    w.xvals = lds_at<double>(w.lds_u + 4u * w.MU);
    w.xkeys = lds_at<int>(w.lds_u + 4u * w.MU + 8u * w.CX);
    w.bt_l = lds_at<u32>(lds0 + (u32)kCtlStruct + 64u * kFlatW * (u32)(BLOCK / 64));
    w.bt_l_cap = 16u * kFlatW * (16u - (u32)(BLOCK / 64));      // 256 words at 768 threads (levels of <= 16 Ki edges), 512 at 512
    // (this kernel's slab capacities are below 2^32 records -- gfpush.hip:sk_slab_sizes --: one 32 x 32 -> 64-bit scalar multiply each)
    const u32 wg = blockIdx.x;
    w.push2   = p.push + (u64)(2u * wg) * (u32)p.push_cap;
    w.bt2     = p.bt + (u64)(2u * wg) * (u32)p.bt_cap;
    const u64 log_off = (u64)wg * (u32)p.log_cap;
    w.log_key = p.log_key + log_off;
    w.log_pu  = p.log_pu + log_off;
    w.arch    = p.arch + (u64)wg * (u32)p.arch_cap;
    return w;
}
// The sketch-cell hash of a key (cells are indexed by its top bits).  v_mul_u32_u24, not v_mul_lo_u32: the full 32 x 32-bit
// multiply issues at a quarter of the rate, and this hash is computed per edge in STREAM and per record in FILTER and in TOP-K's sweep.
// (Which nodes share a cell is irrelevant for a bound that only ever adds; the low 24 bits of a key are its unit number on every
// graph of up to 2^24 units.)
__device__ __forceinline__ u32 sk_cell(u32 key) { return __umul24(key, kSkMul24); }
// Home slot of a key in this kernel's LDS tables: the C++ twin of GP_IW4_HASHES (gfpush_kernels.hpp), full-rate multiplies only.
__device__ __forceinline__ u32 sk_home(u32 k, u32 cap) {
    u32 h = __umul24(k, 0x9E3779u); h ^= h >> 15; h = __umul24(h, 0x85EBCBu);
    return __umul24(h >> 16, cap - kProbeSpan) >> 16;
}
__device__ __forceinline__ bool sk_res_add_lds(int* keys, double* vals, u32 cap, int k, double v) {      // res_add_lds with this kernel's hash
    u32 slot = sk_home((u32)k, cap);
    const int seen = probe_cas_asm(keys, slot, k);
    if (seen != kEmpty && seen != k) return false;
    __hip_atomic_fetch_add(&vals[slot], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    return true;
}
// insert_window_solo (gfpush_kernels.hpp) with this kernel's hash: the lane learns where its key lives (`slot`) and whether it
// CLAIMED that slot (`seen` == kEmpty).  `seen` stays 0 in lanes without an edge.
__device__ __forceinline__ void sk_insert_window_solo(int* keys, double* vals, u32 cap, u32* flag, int col, double sh, u32& slot, int& seen)
{
    u32 t, h, st; u64 sv, ent;
    slot = 0; seen = 0;
    asm volatile(
        "s_mov_b64 %[sv], exec\n\t"
        "v_cmpx_lt_i32 vcc, -1, %[col]\n\t"
        "s_cbranch_execz 5f\n\t"
        "s_mov_b64 %[ent], exec\n\t"
        "v_mul_u32_u24 %[h], 0x9e3779, %[col]\n\t"
        "v_lshrrev_b32 %[t], 15, %[h]\n\t"
        "v_xor_b32 %[h], %[t], %[h]\n\t"
        "v_mul_u32_u24 %[h], 0x85ebcb, %[h]\n\t"
        "v_lshrrev_b32 %[h], 16, %[h]\n\t"
        "v_mul_u32_u24 %[slot], %[capm], %[h]\n\t"
        "v_lshrrev_b32 %[slot], 16, %[slot]\n\t"
        "s_mov_b32 %[st], 1\n"
        "1:\n\t"
        "v_lshl_add_u32 %[t], %[slot], 2, %[kb]\n\t"
        "ds_cmpst_rtn_b32 %[seen], %[t], %[emp], %[col]\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_cmpx_ne_u32 vcc, %[seen], %[col]\n\t"
        "v_cmpx_ne_u32 vcc, -1, %[seen]\n\t"
        "s_cbranch_execz 2f\n\t"
        "v_add_u32 %[slot], %[st], %[slot]\n\t"
        "s_add_u32 %[st], %[st], 1\n\t"
        "s_cmp_le_u32 %[st], %[lim]\n\t"
        "s_cbranch_scc1 1b\n\t"
        "v_mov_b32 %[t], 1\n\t"
        "v_mov_b32 %[h], %[fa]\n\t"
        "ds_write_b32 %[h], %[t]\n"
        "2:\n\t"
        "s_andn2_b64 exec, %[ent], exec\n\t"
        "v_lshl_add_u32 %[t], %[slot], 3, %[vb]\n\t"
        "ds_add_f64 %[t], %[sh]\n"
        "5:\n\t"
        "s_mov_b64 exec, %[sv]"
        : [t] "=&v"(t), [h] "=&v"(h), [slot] "+v"(slot), [seen] "+v"(seen), [sv] "=&s"(sv), [ent] "=&s"(ent), [st] "=&s"(st)
        : [col] "v"(col), [sh] "v"(sh), [emp] "v"(kEmpty),
          [capm] "s"(cap - kProbeSpan), [kb] "s"(lds_addr(keys)), [vb] "s"(lds_addr(vals)), [fa] "s"(lds_addr(flag)), [lim] "n"(kMaxProbe)
        : "vcc", "scc", "memory");
}
__device__ __forceinline__ void lds_add_u32(u32* cell, u32 v) {
    __hip_atomic_fetch_add(cell, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// fixed-point image of a non-negative fp64 quantity, rounded UP: sums of these bound the fp64 sums from above
__device__ __forceinline__ u32 fx_up(double x) { return (u32)__builtin_ceil(x); }

// Exact degree and first column word of the node behind a key of the self-addressed CSR (graph.h:43-45 without indptr): the
// degree field is exact below its saturation value; a saturated one is read from unit_info (the row's second unit holds it when
// every saturated row has one, else the node id -> indptr).
__device__ __forceinline__ u32 sk_degree(KP p, u32 key) {
    const u32 dq = key >> p.deg_shift;
    if (dq != p.deg_sat) return dq;
    const u32 unit = key & p.node_mask;
    if (p.sk_hub_units) return (u32)p.unit_info[unit + 1u];
    const int node = p.unit_info[unit];
    return (u32)(p.indptr[node + 1] - p.indptr[node]);
}

// Walks the log records [first, end) of the workgroup's log in groups of 256, groups dealt to the waves round robin; f(key[4],
// pusher[4], valid[4]) runs while the NEXT group's loads are in flight.  A lane takes FOUR CONSECUTIVE records of a group with one
// 16-byte and one 8-byte load (groups start at multiples of four records of the slab, which is 16-byte aligned: two load
// instructions per group instead of eight); "window" q of a group = record 4 * lane + q of every lane.  Records outside [first, end)
// come back with key -1 (end > first).  (every load is unconditional -- a lane past the end re-reads the last chunk and is masked
// when the group is consumed: a conditional load merges control flow between issue and use, and the compiler then waits for ALL
// loads in flight.)  NT: the records are not needed again (TOP-K's sweep) -- the loads carry the non-temporal hint, so the lines
// leave the L2 first.
template <bool V> struct SkAllValid { static constexpr bool value = V; };
template <int BLOCK, bool NT = false, class F>
__device__ __forceinline__ void log_groups(const int* lk, const unsigned short* lp, u32 first, u32 end, F f)
{
    typedef int i4 __attribute__((ext_vector_type(4)));
    typedef u32 u2 __attribute__((ext_vector_type(2)));
    constexpr u32 kStride = (BLOCK / 64) * 256u;
    const u32 lane = threadIdx.x & 63u;
    u32 g = (first & ~3u) + wave_id() * 256u;
    if (g >= end) return;
    const u32 last_chunk = (end - 1u) & ~3u;
    i4 kn; u2 pn;
    auto load = [&](u32 g0) {
        const u32 i = min(g0 + 4u * lane, last_chunk);
        if (NT) { kn = __builtin_nontemporal_load((const i4*)&lk[i]); pn = __builtin_nontemporal_load((const u2*)&lp[i]); }
        else    { kn = *(const i4*)&lk[i]; pn = *(const u2*)&lp[i]; }
    };
    load(g);
    for (;;) {
        int k[4]; u32 pu[4];
        const u32 i0 = g + 4u * lane;
        // (wave-uniform) a group inside the segment has nothing to mask -- and its consumer nothing to guard: every record holds a
        //  key >= 0.  f(key[4], pusher[4], all_valid) is compiled for both cases.
        const bool whole = g >= first && g + 256u <= end;
        if (whole) {
#pragma unroll
            for (int q = 0; q < 4; ++q) { k[q] = kn[q]; pu[q] = (pn[q >> 1] >> (16 * (q & 1))) & 0xFFFFu; }
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const bool in = i0 + (u32)q >= first && i0 + (u32)q < end;
                k[q] = in ? kn[q] : -1;
                pu[q] = in ? (pn[q >> 1] >> (16 * (q & 1))) & 0xFFFFu : 0u;
            }
        }
        g += kStride;
        const bool more = g < end;                                    // wave-uniform
        if (more) load(g);
        if (whole) f(k, pu, SkAllValid<true>()); else f(k, pu, SkAllValid<false>());
        if (!more) break;
    }
}

// The same walk with TWO groups' loads in flight (TOP-K's sweep: ~9 groups per wave of records that left the L2 long ago, so a
// group's loads take an HBM / Infinity-Cache round trip and one group ahead does not cover it).  The loop is unrolled by two so
// that the two register sets alternate without moves.
template <int BLOCK, class F>
__device__ __forceinline__ void log_groups2_nt(const int* lk, const unsigned short* lp, u32 end, F f)
{
    typedef int i4 __attribute__((ext_vector_type(4)));
    typedef u32 u2 __attribute__((ext_vector_type(2)));
    constexpr u32 kStride = (BLOCK / 64) * 256u;
    const u32 lane = threadIdx.x & 63u;
    u32 g = wave_id() * 256u;
    if (g >= end) return;
    const u32 last_chunk = (end - 1u) & ~3u;
    i4 ka, kb; u2 pa, pb;
    auto load = [&](u32 g0, i4& kn, u2& pn) {
        const u32 i = min(g0 + 4u * lane, last_chunk);
        kn = __builtin_nontemporal_load((const i4*)&lk[i]); pn = __builtin_nontemporal_load((const u2*)&lp[i]);
    };
    auto consume = [&](u32 g0, const i4& kn, const u2& pn, int (&k)[4], u32 (&pu)[4]) -> bool {
        const u32 i0 = g0 + 4u * lane;
        const bool whole = g0 + 256u <= end;
        if (whole) {                                                      // (wave-uniform) every group but the log's last: nothing to mask
#pragma unroll
            for (int q = 0; q < 4; ++q) { k[q] = kn[q]; pu[q] = (pn[q >> 1] >> (16 * (q & 1))) & 0xFFFFu; }
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const bool in = i0 + (u32)q < end;
                k[q] = in ? kn[q] : -1;
                pu[q] = in ? (pn[q >> 1] >> (16 * (q & 1))) & 0xFFFFu : 0u;
            }
        }
        return whole;
    };
    load(g, ka, pa);
    kb = ka; pb = pa;
    if (g + kStride < end) load(g + kStride, kb, pb);
    for (;;) {
        int k[4]; u32 pu[4];
        // (the consumer is compiled once here, guards included: a second copy of TOP-K's insert loop makes that function save nine
        //  more callee-saved registers to scratch)
        consume(g, ka, pa, k, pu);
        if (g + 2u * kStride < end) load(g + 2u * kStride, ka, pa);      // (wave-uniform)
        f(k, pu, SkAllValid<false>());
        g += kStride;
        if (g >= end) break;
        consume(g, kb, pb, k, pu);
        if (g + 2u * kStride < end) load(g + 2u * kStride, kb, pb);
        f(k, pu, SkAllValid<false>());
        g += kStride;
        if (g >= end) break;
    }
}

// ---------------------------------------------------------------- the edge enumeration
// edge_stream of gfpush_kernels.hpp (one lane per edge, equal EDGE counts per wave, three-stage software pipeline; see there),
// with two additions made where an edge is matched to its entry (not a register of the pipeline is spent on them): the edge's log
// record gets its pusher number (pu_base + the owner's index in the push list; lp == nullptr: not wanted), and the entries a wave
// holds in its lanes leave their share in the level's share table S (S == nullptr: not wanted).
// FX: what an edge carries of its entry's share.  0: the fp64 share (levels that insert into the exact table).  1: its two
// fixed-point images -- ceil(share * 2^31) for the level sketch and ceil(share * cs) for the reserve sketch --, 2: the second only
// (last level).  The images are computed ONCE PER ENTRY by the lane that holds it and travel through the same two ds_bpermute
// that would move the share's halves: six fp64 operations per edge become six per pusher (a sketch level has 13 edges per pusher).
template <int BLOCK, bool NT, int FX, class F>
__device__ __forceinline__ void sk_edge_stream(KP p, CtlS* ctl, const PushEntry* push, const u32* bt, u32 n_ent, u32 E, double* S,
                                               unsigned short* lp, u32 pu_base, double cs, F f)
{
    constexpr u32 kWaves = BLOCK / 64;
    static_assert(kFlatW == 4, "sk_edge_stream reads the four window flags of a lane as one 32-bit word");
    const u32 lane = threadIdx.x & 63u;
    const u32 wave = wave_id();
    unsigned char* wscr = (unsigned char*)ctl + kCtlStruct + 64 * kFlatW * wave;
    const int* indices = p.indices;
    const u32 sentinel = (u32)p.nnz;
    const u32 units = (E + (1u << kUnitShift) - 1u) >> kUnitShift;
    // (ranges are dealt from the last wave down, so that a level of a few units lands on wave 0, 1, ...: the waves that were
    //  dispatched first win the issue arbitration against younger waves, and a small level is a latency chain of one wave)
    const u32 slot = kWaves - 1u - wave;
    const u32 u_lo = (slot * units) / kWaves, u_hi = ((slot + 1u) * units) / kWaves;    // (units < 2^26, slot < 16: 32 bits hold the products)
    if (u_lo >= u_hi || n_ent == 0) return;
    const bool small = n_ent <= 64u;                                           // (wave-uniform) the whole list in one wave
    u32 btv = 0, bt_first = 0;
    auto load_bt = [&](u32 step0) {
        const u32 uj = u_lo + 4u * (step0 + lane);
        btv = uj < u_hi ? bt[uj] : 0u;
        bt_first = step0;
    };
    if (!small) load_bt(0);
    PushEntry entn; entn.rel = 0; entn.off = 0; entn.share = 0.0;
    u32 i0n = 0, t0n = 0, t1n = 0;
    u32 u_next = u_lo;
    bool have_ent = true;
    auto fetch_next = [&](u32 end) {
        if (end < t1n) { i0n += 63u; t0n = end; }
        else if (u_next < u_hi) {
            t0n = u_next << kUnitShift; t1n = min(E, min(u_next + 4u, u_hi) << kUnitShift);
            if (!small) {
                const u32 s_no = (u_next - u_lo) >> 2;
                if (s_no - bt_first >= 64u) load_bt(s_no);
                i0n = (u32)__builtin_amdgcn_readlane((int)btv, (int)(s_no - bt_first));
            }
            u_next += 4u;
        } else { have_ent = false; return; }
        if (!small) entn = push[min(i0n + lane, n_ent - 1u)];
    };
    if (small) entn = push[min(lane, n_ent - 1u)];
    fetch_next(0);

    int nc[4] = {-1, -1, -1, -1}; double ns[4] = {0.0, 0.0, 0.0, 0.0}; u32 nu[4] = {0, 0, 0, 0}, nr[4] = {0, 0, 0, 0};
    u32 nt0 = 0;
    bool have_cols = false;
    do {
        u32 idx[4]; double sh[4] = {0.0, 0.0, 0.0, 0.0}; u32 su[4] = {0, 0, 0, 0}, sr[4] = {0, 0, 0, 0}; u32 end = 0;
        if (have_ent) {
            const u32 cnt = min(64u, n_ent - i0n);
            const u32 off = lane < cnt ? entn.off : 0xFFFFFFFFu;
            if (S && lane < cnt) S[i0n + lane] = entn.share;                   // (entries at step boundaries are written by both neighbours: same value)
            end = t1n;
            if (cnt == 64u && i0n + 64u < n_ent) {
                const u32 o63 = (u32)__builtin_amdgcn_readlane((int)off, 63);
                if (o63 < t1n) end = o63;
            }
            *(u32*)(wscr + 4 * lane) = 0u;
            if (off > t0n && off < end) { const u32 pos = off - t0n; wscr[(pos & 63u) * 4u + (pos >> 6)] = 1; }
            asm volatile("" ::: "memory");
            const u32 fl = *(const u32*)(wscr + 4 * lane);
            u32 before = small ? (u32)__popcll(__ballot(off <= t0n)) - 1u : 0u;
            u32 e[4];
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const bool mine = ((fl >> (8 * w)) & 1u) != 0;
                const u64 M = __ballot(mine);
                e[w] = before + lane_prefix(M) + (mine ? 1u : 0u);             // the owning lane
                before += (u32)__popcll(M);
            }
            const u64 sbits = (u64)__double_as_longlong(entn.share);
            u32 eu = 0, er = 0;                                                // this lane's entry: its share in sketch units
            if (FX == 1) eu = fx_up(entn.share * 2147483648.0);
            if (FX != 0) er = fx_up(entn.share * cs);
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const u32 rel_e = (u32)__builtin_amdgcn_ds_bpermute((int)(e[w] << 2), (int)entn.rel);
                if (FX == 0) {
                    const u32 lo = (u32)__builtin_amdgcn_ds_bpermute((int)(e[w] << 2), (int)(u32)sbits);
                    const u32 hi = (u32)__builtin_amdgcn_ds_bpermute((int)(e[w] << 2), (int)(u32)(sbits >> 32));
                    sh[w] = __longlong_as_double((long long)(((u64)hi << 32) | lo));
                }
                if (FX == 1) su[w] = (u32)__builtin_amdgcn_ds_bpermute((int)(e[w] << 2), (int)eu);
                if (FX != 0) sr[w] = (u32)__builtin_amdgcn_ds_bpermute((int)(e[w] << 2), (int)er);
                const u32 q = t0n + 64u * (u32)w + lane;
                idx[w] = q < end ? rel_e + q : sentinel;                       // graph.h:97
                if (lp && q < end) {                                           // graph.h:98 -> the record's pusher number
                    const unsigned short pu = (unsigned short)(pu_base + i0n + e[w]);
                    if (NT) __builtin_nontemporal_store(pu, &lp[q]); else lp[q] = pu;
                }
            }
        }
        int cc[4]; double csh[4]; u32 cu[4], cr[4];
#pragma unroll
        for (int w = 0; w < 4; ++w) { cc[w] = nc[w]; csh[w] = ns[w]; cu[w] = nu[w]; cr[w] = nr[w]; }
        const u32 ct0 = nt0;
        const bool had_cols = have_cols;
        have_cols = have_ent;
        if (have_ent) {
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                nc[w] = indices[idx[w]];
                if (FX == 0) ns[w] = sh[w];
                if (FX == 1) nu[w] = su[w];
                if (FX != 0) nr[w] = sr[w];
            }
            nt0 = t0n;
            fetch_next(end);
        }
        if (had_cols) f(cc, csh, cu, cr, ct0);
    } while (have_cols);
}

// Appends the pushing nodes of one wave-step to the next level's push list (push_alloc of gfpush_kernels.hpp) and leaves
// cnext * share -- what every edge of the entry adds to its target's reserve (graph.h:90) -- at the pusher's number in the row.
// n_push / n_edges (wave-uniform) take the step's pushers and their edges: the statistics fall out of the ballot and the scan
// that are computed anyway (three wave reductions per SCAN call otherwise).
__device__ __forceinline__ void sk_push_alloc(KP p, CtlS* ctl, LevelCtr* nx, PushEntry* push, u32* bt_g, double* arch, u32 pu_next, double cnext,
                                              u32 len, u32 start, double share, int lane, u32* bt_l, u32 bt_l_cap, u32& n_push, u32& n_edges)
{
    const u64 M = __ballot(len != 0);
    if (M == 0) return;                                                       // wave-uniform: nobody pushes
    const u32 incl = wave_incl_scan_dpp(len);
    const u32 tot = (u32)__builtin_amdgcn_readlane((int)incl, 63);
    n_push += (u32)__popcll(M); n_edges += tot;
    u64 base = 0;
    if (lane == 0)
        base = __hip_atomic_fetch_add(&nx->alloc, ((u64)tot << 32) | (u64)(u32)__popcll(M), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    base = uni(base);
    if (len != 0) {
        const u32 idx = (u32)base + lane_prefix(M), off = (u32)(base >> 32) + (incl - len);
        if ((u64)idx < p.push_cap && (u64)pu_next + idx < p.arch_cap) {
            PushEntry pe; pe.rel = start - off; pe.off = off; pe.share = share; push[idx] = pe;
            arch[pu_next + idx] = cnext * share;
        } else ctl->fail = 1;
        for (u32 m = (off + (1u << kUnitShift) - 1u) >> kUnitShift; ((u64)m << kUnitShift) < (u64)off + len; ++m) {   // hubs: one word per 64 edges
            if ((u64)m < p.bt_cap) bt_g[m] = idx; else ctl->fail = 1;
            if (m < bt_l_cap) bt_l[m] = idx;
        }
    }
}

// ---------------------------------------------------------------- STREAM
// MODE 0: log + reserve sketch + level sketch U + share table.   MODE 1: log + reserve sketch + exact insert into X (small levels).
// MODE 2: log + reserve sketch (last level; the first pass of a level without the sketch that is walked in partitions from the start).
// MODE 3: exact inserts of hash partition `part` of `parts` only (a small level whose table overflowed, or that is walked in partitions:
// its records and sketch adds exist).   cs = coef[level] * scale: reserve-sketch units per unit of share.
template <int BLOCK, int MODE>
__device__ GP_PHASE_NOINLINE void phase_sk_stream(u32 lds0, u32 cur, u32 n_ent, u32 E, u32 seg_base, double cs, u32 capx,
                                                  u32 pu_base, u32 parts, u32 part)
{
    KP p = kparams();
    lds0 = uni(lds0); cur = uni(cur); n_ent = uni(n_ent); E = uni(E); seg_base = uni(seg_base); cs = uni(cs); capx = uni(capx);
    pu_base = uni(pu_base); parts = uni(parts); part = uni(part);
    const SkView w = sk_view<BLOCK>(p, lds0);
    const u32 lane = threadIdx.x & 63u;
    int* lk = w.log_key + seg_base; unsigned short* lp = w.log_pu + seg_base;
    double* S = MODE == 0 ? w.xvals + capx : nullptr;
    // (the boundary table's head was also written to LDS by the SCAN that built the push list: one global round trip less in
    //  front of the first entry load of every level of <= bt_l_cap * 64 edges
    //  -- but a partition walk (MODE 3) runs after a SCAN has written the NEXT list's table head there)
    const u32* btp = MODE != 3 && ((E + 63u) >> 6) <= w.bt_l_cap ? (const u32*)w.bt_l : (const u32*)(w.bt2 + (u64)cur * (u32)p.bt_cap);
    sk_edge_stream<BLOCK, MODE != 0, MODE == 0 ? 1 : MODE == 2 ? 2 : 0>(p, w.ctl, w.push2 + (u64)cur * (u32)p.push_cap, btp, n_ent, E, S,
                                     MODE != 3 ? lp : nullptr, pu_base, cs,
                                     [&](const int (&v)[4], const double (&sh)[4], const u32 (&su)[4], const u32 (&sr)[4], u32 t0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (MODE != 3 && v[q] >= 0) {
                const u32 li = t0 + 64u * (u32)q + lane;                              // the edge's number inside the level
                // graph.h:98 -> one log record per edge.  Only a sketch level's FILTER reads its records back soon; the others are next
                // read by TOP-K: non-temporal stores
                if (MODE != 0) __builtin_nontemporal_store(v[q], &lk[li]); else lk[li] = v[q];
                const u32 h = sk_cell((u32)v[q]);
                if (cs != 0.0) lds_add_u32(&w.R[h >> w.shR], MODE == 1 ? fx_up(sh[q] * cs) : sr[q]);       // graph.h:90 / :109, as an upper bound
                if (MODE == 0) lds_add_u32(&w.U[h >> w.shU], su[q]);
            }
        }
        if (MODE == 1) {                                                                  // graph.h:98, two windows' compare-and-swaps in flight together
            insert_windows2_asm(w.xkeys, w.xvals, capx, &w.ctl->ovf, v[0], v[1], sh[0], sh[1]);
            insert_windows2_asm(w.xkeys, w.xvals, capx, &w.ctl->ovf, v[2], v[3], sh[2], sh[3]);
        }
        if (MODE == 3) insert_windows4_asm<true>(w.xkeys, w.xvals, capx, &w.ctl->ovf, v, sh, parts, part);
    });
    if (threadIdx.x == 0 && w.ctl->lv_has_dang) {                                     // graph.h:92: the seed gets the dangling mass
        const double dang = w.ctl->lv_dang; const int seed_key = w.ctl->seed_key;
        if (MODE != 3) {
            lk[E] = seed_key; lp[E] = (unsigned short)(pu_base + n_ent);
            const u32 h = sk_cell((u32)seed_key);
            if (cs != 0.0) lds_add_u32(&w.R[h >> w.shR], fx_up(dang * cs));
            if (MODE == 0) { lds_add_u32(&w.U[h >> w.shU], fx_up(dang * 2147483648.0)); S[n_ent] = dang; }
        }
        if ((MODE == 1 || (MODE == 3 && (parts == 1u || slot_of(hash_b((u32)seed_key), parts) == part))) &&
            !sk_res_add_lds(w.xkeys, w.xvals, capx, seed_key, dang)) w.ctl->ovf = 1;
    }
}

// ---------------------------------------------------------------- FILTER
// The log segment [seg_base, seg_base + n) of a sketch level: edges whose target MAY push (its sketch cell reaches rmax * packed
// degree) go into the exact table with their pusher's share (S[pusher number - pu_base]).  parts > 1: only targets of hash partition `part`.
template <int BLOCK>
__device__ GP_PHASE_NOINLINE void phase_sk_filter(u32 lds0, u32 seg_base, u32 n, u32 capx, u32 pu_base, u32 parts, u32 part)
{
    KP p = kparams();
    lds0 = uni(lds0); seg_base = uni(seg_base); n = uni(n); capx = uni(capx); pu_base = uni(pu_base); parts = uni(parts); part = uni(part);
    const SkView w = sk_view<BLOCK>(p, lds0);
    const float thr = p.sk_thr_f;
    const u32 dshift = (u32)p.deg_shift;                                              // (a kernel-argument read inside the loop is re-issued behind every asm block)
    const double* S = w.xvals + capx;
    u32 n_cand = 0;
    SKT2(w.ctl, 0);
    log_groups<BLOCK>(w.log_key, w.log_pu, seg_base, seg_base + n, [&](const int (&k)[4], const u32 (&pu)[4], auto all_valid) {
        constexpr bool kAll = decltype(all_valid)::value;                             // every record of the group is one: no guards
        u32 cell[4]; double s[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {                                                 // eight lookups in flight
            cell[q] = w.U[sk_cell(kAll ? (u32)k[q] : (u32)max(k[q], 0)) >> w.shU];
            s[q] = S[kAll || k[q] >= 0 ? pu[q] - pu_base : 0u];
        }
        int kc[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const u32 dq = (u32)k[q] >> dshift;                                       // min(deg, deg_sat); 0: dangling, always exact
            const bool cand = (kAll || k[q] >= 0) && (float)cell[q] >= (float)dq * thr;   // thr = rmax * 2^31 * (1 - 2^-10), rounded down
            n_cand += (u32)__popcll(__ballot(cand));
            kc[q] = cand ? k[q] : -1;
        }
        insert_windows4_asm<true>(w.xkeys, w.xvals, capx, &w.ctl->ovf, kc, s, parts, part);     // graph.h:98
    });
    if ((threadIdx.x & 63u) == 0 && n_cand && part == 0u) zstat(w.ctl, zCand, n_cand);
    SKT2(w.ctl, 1);
}

// ---------------------------------------------------------------- SCAN
// Drains the exact table (cap slots in use, C allocated; C % 4 == 0, slots in [cap, C) are empty): compact, cheap push test
// on the packed degree, compact again, then exact degree / dangling rule / division / push list for the nodes that remain.
// clear_n > 0: the level sketch and the share table (clear_n values behind slot `cap`) are dead (this is the level's last
// partition) -- zero them for the next level.  pu_next / cnext: the row's pusher number of the next list's entry 0, coef[level + 1].
template <int BLOCK>
__device__ GP_PHASE_NOINLINE void phase_sk_scan(u32 lds0, u32 cap, u32 nx_sel, u32 nxt_sel, u32 clear_n, u32 pu_next, double cnext)
{
    typedef int    i4 __attribute__((ext_vector_type(4)));
    typedef double d2 __attribute__((ext_vector_type(2)));
    typedef u32    u4 __attribute__((ext_vector_type(4)));
    KP p = kparams();
    lds0 = uni(lds0); cap = uni(cap); nx_sel = uni(nx_sel); nxt_sel = uni(nxt_sel); clear_n = uni(clear_n); pu_next = uni(pu_next); cnext = uni(cnext);
    const SkView w = sk_view<BLOCK>(p, lds0);
    CtlS* ctl = w.ctl;
    int* lkeys = w.xkeys; double* lvals = w.xvals;
    const u32 C = w.CX;
    LevelCtr* nx = &ctl->lc[nx_sel];
    PushEntry* push = w.push2 + (u64)nxt_sel * (u32)p.push_cap;
    u32* bt_g = w.bt2 + (u64)nxt_sel * (u32)p.bt_cap;
    const int tid = threadIdx.x, lane = tid & 63;
    SKT2(ctl, 4);
    if (clear_n) {
        const u4 z = {0u, 0u, 0u, 0u};
        for (u32 i = 4u * (u32)tid; i < w.MU; i += 4u * BLOCK) *(u4*)&w.U[i] = z;
        for (u32 i = (u32)tid; i < clear_n; i += BLOCK) lvals[cap + i] = 0.0;
    }
    constexpr u32 kWaves = BLOCK / 64;
    const u32 range = ((cap + kWaves * 256u - 1u) / (kWaves * 256u)) * 256u;
    const u32 wb = wave_id() * range;
    // (round 6) ONE pass: the slots of my range are read four per lane, cleared, and what passes the cheap half of the push test --
    // the exact degree is in the key unless the field is saturated (graph.h:94) -- is compacted to the front of the range.  (Until
    // round 5 the occupied slots were compacted first and tested in a second pass over them: one more LDS write / read round per call.)
    u32 tot = 0, ncand = 0;
    const double rmax_ = p.rmax; const u32 dsh = (u32)p.deg_shift;
    for (u32 sub = wb; sub < wb + range && sub < cap; sub += 256u) {
        const u32 s0 = sub + 4u * (u32)lane;
        i4 kk = {kEmpty, kEmpty, kEmpty, kEmpty};
        if (s0 < cap) kk = *(const i4*)&lkeys[s0];                  // (cap % 4 == 0)
        const bool o0 = kk.x != kEmpty, o1 = kk.y != kEmpty, o2 = kk.z != kEmpty, o3 = kk.w != kEmpty;
        const u32 n_occ = (u32)__popcll(__ballot(o0)) + (u32)__popcll(__ballot(o1)) + (u32)__popcll(__ballot(o2)) + (u32)__popcll(__ballot(o3));
        if (n_occ == 0) continue;                                   // wave-uniform
        tot += n_occ;
        d2 ra = {0.0, 0.0}, rb = {0.0, 0.0};
        if (o0 | o1 | o2 | o3) {
            ra = *(const d2*)&lvals[s0]; rb = *(const d2*)&lvals[s0 + 2];
            const i4 ke = {kEmpty, kEmpty, kEmpty, kEmpty};
            const d2 z = {0.0, 0.0};
            *(i4*)&lkeys[s0] = ke; *(d2*)&lvals[s0] = z; *(d2*)&lvals[s0 + 2] = z;
        }
        __atomic_signal_fence(__ATOMIC_SEQ_CST);                    // clears stay ahead of the staging stores
        const u32 d0 = (u32)kk.x >> dsh, d1 = (u32)kk.y >> dsh, d2_ = (u32)kk.z >> dsh, d3 = (u32)kk.w >> dsh;
        const bool c0b = o0 && (d0 == 0u || ra.x >= rmax_ * (double)d0), c1b = o1 && (d1 == 0u || ra.y >= rmax_ * (double)d1),
                   c2b = o2 && (d2_ == 0u || rb.x >= rmax_ * (double)d2_), c3b = o3 && (d3 == 0u || rb.y >= rmax_ * (double)d3);
        const u64 m0 = __ballot(c0b), m1 = __ballot(c1b), m2 = __ballot(c2b), m3 = __ballot(c3b);
        const u32 c0 = (u32)__popcll(m0), c1 = (u32)__popcll(m1), c2 = (u32)__popcll(m2), c3 = (u32)__popcll(m3);
        const u32 q0 = wb + ncand;                                  // [wb, wb + ncand) lies inside the slots drained so far
        if (c0b) { const u32 q = q0 + lane_prefix(m0);                lkeys[q] = kk.x; lvals[q] = ra.x; }
        if (c1b) { const u32 q = q0 + c0 + lane_prefix(m1);           lkeys[q] = kk.y; lvals[q] = ra.y; }
        if (c2b) { const u32 q = q0 + c0 + c1 + lane_prefix(m2);      lkeys[q] = kk.z; lvals[q] = rb.x; }
        if (c3b) { const u32 q = q0 + c0 + c1 + c2 + lane_prefix(m3); lkeys[q] = kk.w; lvals[q] = rb.y; }
        __atomic_signal_fence(__ATOMIC_SEQ_CST);
        ncand += c0 + c1 + c2 + c3;
    }
    (void)C;
    u32 st_push = 0, st_edges = 0, st_deg = 0;
    SKT2(ctl, 5);
    if (tot != 0) {
        SKT2(ctl, 6);
        for (u32 j = 0; j < ncand; j += 64u) {
            const u32 idx = j + (u32)lane;
            const bool want = idx < ncand;
            int k = kEmpty; double r = 0.0; u32 deg = 0;
            if (want) {
                k = lkeys[wb + idx]; r = lvals[wb + idx];
                lkeys[wb + idx] = kEmpty; lvals[wb + idx] = 0.0;
                deg = (u32)k >> p.deg_shift;
                if (deg == p.deg_sat) deg = sk_degree(p, (u32)k);                         // graph.h:43-45 (a saturated degree field: one word of unit_info)
            }
            double share = 0.0; u32 len = 0;
            if (want) {
                if (deg == 0) {                                                       // graph.h:91-93
                    __hip_atomic_fetch_add(&nx->dangling, r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_fetch_add(&nx->n_dangling, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                } else if (r >= p.rmax * (double)deg) {                               // graph.h:94
                    // (a push always has a share > 0 here: r >= rmax * deg with rmax * 2^31 >= 64, so pushes and pushed edges are
                    //  the entries and edges sk_push_alloc hands out)
                    const double sh = r / (double)deg;                                // graph.h:95
                    if (sh != 0.0) { share = sh; len = deg; }
                }
            }
            st_deg += (u32)__popcll(__ballot(want && ((u32)k >> p.deg_shift) == p.deg_sat));
            sk_push_alloc(p, ctl, nx, push, bt_g, w.arch, pu_next, cnext, len, ((u32)k & p.node_mask) << kSkUnitShift, share, lane, w.bt_l, w.bt_l_cap, st_push, st_edges);
        }
        SKT2(ctl, 7);
        if (lane == 0) {
            __hip_atomic_fetch_add(&nx->n_rec, tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // nodes the exact table held
            if (st_deg) zstat(ctl, zDeg, st_deg);
            if (st_push) { zstat(ctl, zPush, st_push); zstat(ctl, zEdges, st_edges); }
        }
    }
    SKT2(ctl, 8);
}

// ---------------------------------------------------------------- small levels by one wave
// A level of at most 256 edges from at most 64 push-list entries (levels 1, 9, 10 of a MAG row; most levels of a short row),
// done by ONE wave start to finish: one step of the edge enumeration with the whole push list in the wave's lanes, log
// records and reserve-sketch adds per edge, exact inserts through insert_window_solo -- which tells a lane whether it CLAIMED
// its slot, so the claimed slots ARE the level's frontier and SCAN runs straight over that list, no table walk -- with
// nothing but the wave's program order in between (LDS operations of one wave execute in order).  The other waves skip the
// call and park at the one barrier behind it (the general path: two calls per wave, two barriers, a table walk).  An insert
// that hits the probe limit (ctl->ovf) undoes the claims; the caller then re-streams the level in hash partitions.
constexpr u32 kSkSoloEdges = 256;
// SEED (round 6): level 1 straight behind level 0, in the same call of wave 0 (phase_sk_seed): its one entry -- the seed --
// arrives in (seed_rel, seed_share) instead of being read back from the push list the same wave has just written.
template <int BLOCK, bool SEED = false>
__device__ GP_PHASE_NOINLINE void phase_sk_solo(u32 lds0, u32 cur, u32 n_ent, u32 E, u32 seg_base, double cs,
                                                u32 has_dang, double dang, int seed_key, u32 nx_sel, u32 pu_base, u32 pu_next, double cnext,
                                                u32 seed_rel = 0, double seed_share = 0.0)
{
    KP p = kparams();
    lds0 = uni(lds0); cur = uni(cur); n_ent = uni(n_ent); E = uni(E); seg_base = uni(seg_base); cs = uni(cs);
    has_dang = uni(has_dang); dang = uni(dang); seed_key = uni(seed_key); nx_sel = uni(nx_sel);
    pu_base = uni(pu_base); pu_next = uni(pu_next); cnext = uni(cnext);
    const SkView w = sk_view<BLOCK>(p, lds0);
    CtlS* ctl = w.ctl; int* lkeys = w.xkeys; double* lvals = w.xvals;
    const u32 lane = threadIdx.x & 63u;
    const u32 cap = kMinCap;
    const SerialSection ahead;
    unsigned char* wscr = (unsigned char*)ctl + kCtlStruct;                 // wave 0's flag bytes
    u32* list = (u32*)((unsigned char*)ctl + kCtlStruct + 64 * kFlatW);     // the flag areas of the other waves (>= 257 words): they are parked
    static_assert((BLOCK / 64 - 1) * 64 * kFlatW >= 4 * (kSkSoloEdges + 1), "the claimed-slot list lives in the parked waves' flag bytes");
    LevelCtr* nx = &ctl->lc[nx_sel];
    const PushEntry* push_cur = w.push2 + (u64)cur * (u32)p.push_cap;
    PushEntry* push_nxt = w.push2 + (u64)(cur ^ 1u) * (u32)p.push_cap;
    u32* bt_nxt = w.bt2 + (u64)(cur ^ 1u) * (u32)p.bt_cap;
    int* lk = w.log_key + seg_base; unsigned short* lp = w.log_pu + seg_base;
    // ---- the one step of the edge enumeration (as sk_edge_stream: entries flag their first edge, ballot, mbcnt, bpermute)
    PushEntry ent;
    if (SEED) { ent.rel = uni(seed_rel); ent.off = 0u; ent.share = uni(seed_share); }
    else ent = push_cur[min(lane, n_ent - 1u)];
    const u32 off = lane < n_ent ? ent.off : 0xFFFFFFFFu;
    *(u32*)(wscr + 4 * lane) = 0u;
    if (off > 0u && off < E) wscr[(off & 63u) * 4u + (off >> 6)] = 1;
    asm volatile("" ::: "memory");            // the word is written by OTHER lanes
    const u32 fl = *(const u32*)(wscr + 4 * lane);
    u32 before = (u32)__popcll(__ballot(off == 0u)) - 1u;
    int col[4]; double sh[4]; u32 own[4];
    {
        const u64 sbits = (u64)__double_as_longlong(ent.share);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const bool mine = ((fl >> (8 * q)) & 1u) != 0;
            const u64 M = __ballot(mine);
            own[q] = before + lane_prefix(M) + (mine ? 1u : 0u);
            before += (u32)__popcll(M);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const u32 rel_e = (u32)__builtin_amdgcn_ds_bpermute((int)(own[q] << 2), (int)ent.rel);
            const u32 lo = (u32)__builtin_amdgcn_ds_bpermute((int)(own[q] << 2), (int)(u32)sbits);
            const u32 hi = (u32)__builtin_amdgcn_ds_bpermute((int)(own[q] << 2), (int)(u32)(sbits >> 32));
            const u32 eq = 64u * (u32)q + lane;
            col[q] = p.indices[eq < E ? rel_e + eq : (u32)p.nnz];           // graph.h:97
            sh[q] = __longlong_as_double((long long)(((u64)hi << 32) | lo));
        }
    }
    u32 n_list = 0;
#pragma unroll
    for (int q = 0; q < 5; ++q) {             // the four windows, then the mass dangling nodes returned to the seed (graph.h:92)
        if (q == 4 && !has_dang) break;
        const int kq = q < 4 ? col[q] : (lane == 0 ? seed_key : -1);
        const double vq = q < 4 ? sh[q] : dang;
        if (kq >= 0) {
            const u32 li = q < 4 ? 64u * (u32)q + lane : E;
            __builtin_nontemporal_store(kq, &lk[li]);                                 // graph.h:98 -> one log record per edge (next read by TOP-K)
            __builtin_nontemporal_store((unsigned short)(pu_base + (q < 4 ? own[q < 4 ? q : 0] : n_ent)), &lp[li]);
            if (cs != 0.0) lds_add_u32(&w.R[sk_cell((u32)kq) >> w.shR], fx_up(vq * cs));
        }
        u32 slot; int seen;
        sk_insert_window_solo(lkeys, lvals, cap, &ctl->ovf, kq, vq, slot, seen);
        const bool fresh = kq >= 0 && seen == kEmpty;                       // this lane claimed the slot: a new frontier node
        const u64 M = __ballot(fresh);
        if (fresh) list[n_list + lane_prefix(M)] = slot;
        n_list += (u32)__popcll(M);
    }
    asm volatile("" ::: "memory");
    if (uni(ctl->ovf)) {                      // undo: the claimed slots are all there is
        for (u32 j = lane; j < n_list; j += 64u) { const u32 sl = list[j]; lkeys[sl] = kEmpty; lvals[sl] = 0.0; }
        return;
    }
    // ---- SCAN over the claimed slots
    u32 st_push = 0, st_edges = 0, st_deg = 0;
    for (u32 j = 0; j < n_list; j += 64u) {
        const bool valid = j + lane < n_list;
        int k = kEmpty; double r = 0.0;
        if (valid) {
            const u32 sl = list[j + lane];
            k = lkeys[sl]; r = lvals[sl];
            lkeys[sl] = kEmpty; lvals[sl] = 0.0;
        }
        const u32 dq = (u32)k >> p.deg_shift;
        const bool cand = valid && (dq == 0u || r >= p.rmax * (double)dq);
        double share = 0.0; u32 len = 0;
        if (cand) {
            u32 deg = dq;
            if (dq == p.deg_sat) deg = sk_degree(p, (u32)k);                                          // graph.h:43-45
            if (deg == 0) {                                                                           // graph.h:91-93
                __hip_atomic_fetch_add(&nx->dangling, r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_add(&nx->n_dangling, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            } else if (r >= p.rmax * (double)deg) {                                                   // graph.h:94
                const double s_ = r / (double)deg;                                                    // graph.h:95
                if (s_ != 0.0) { share = s_; len = deg; }
            }
        }
        st_deg += (u32)__popcll(__ballot(cand && dq == p.deg_sat));
        sk_push_alloc(p, ctl, nx, push_nxt, bt_nxt, w.arch, pu_next, cnext, len, ((u32)k & p.node_mask) << kSkUnitShift, share, (int)lane, w.bt_l, w.bt_l_cap, st_push, st_edges);
    }
    if (lane == 0) {
        nx->n_rec = n_list;
        if (st_deg) zstat(ctl, zDeg, st_deg);
        if (st_push) { zstat(ctl, zPush, st_push); zstat(ctl, zEdges, st_edges); }
    }
}

// ---------------------------------------------------------------- TOP-K
// TOP-K's carving of the LDS behind the control block and R: { T f64[pushers of the row] | aggregation values f64[CA] | keys i32[CA] |
// tie Cand[256] | sel Cand[K] } over the dead level tables.  The table takes every byte that is left: a table at X's own size
// (2 960 slots, which the levels leave empty: no wipe) measured +4.5 % kernel time against ~4 100 slots -- one insert that runs
// into the probe limit costs the row a whole partitioned round.
struct SkTop {
    u32* R; u32 MR, shR; double* T; double* avals; int* akeys; u32 CA; Cand* tie; Cand* sel; u32* fine;
};
// slots of the aggregation table that remain beside T for n_pu pushers (0: T does not fit)
__device__ __forceinline__ u32 sk_top_slots(u32 region_bytes, u32 n_pu, u32 K) {
    const u32 fixed = 8u * ((n_pu + 1u) & ~1u) + 16u * (kSkTie + K);
    return region_bytes > fixed ? ((region_bytes - fixed) / 12u) & ~7u : 0u;
}
__device__ __forceinline__ SkTop sk_top(KP p, const SkView& w, u32 n_pu) {
    SkTop t;
    t.MR = w.MR; t.shR = w.shR;
    t.R = w.R;
    const u32 region_bytes = 4u * w.MU + 12u * w.CX;
    t.CA = sk_top_slots(region_bytes, n_pu, (u32)p.K);
    t.T = lds_at<double>(w.lds_u);
    const u32 a0 = w.lds_u + 8u * ((n_pu + 1u) & ~1u);                                // (16-byte aligned)
    t.avals = lds_at<double>(a0);
    t.akeys = lds_at<int>(a0 + 8u * t.CA);
    t.tie = lds_at<Cand>(a0 + 12u * t.CA);
    t.sel = t.tie + kSkTie;
    t.fine = (u32*)((unsigned char*)w.ctl + kCtlStruct);                              // 1 024 words of flag bytes, idle in TOP-K
    return t;
}

// The tail of both selects: the entries of the K-th bin ranked among themselves (value desc, node id asc), then all selected ranked
// (thread i < need: my_rank = position of sel[i] in the output order); the smallest selected value goes to ctl->kth_bits.
template <int BLOCK>
__device__ __forceinline__ void sk_rank_selected(KP p, CtlS* ctl, const SkTop& t, u32 need, bool take_all, u32& my_rank)
{
    const int tid = threadIdx.x;
    if (!take_all) {
        const u32 nt = uni(ctl->n_tie), n0 = uni(ctl->n_sel), want = need - n0;        // nt == cnt_b <= kSkTie
        // ties inside the bin are ordered like the output: value desc, then NODE id asc (the packed key's low bits)
        for (u32 i = tid; i < nt; i += BLOCK) {
            Cand mine = t.tie[i];
            const int mk = (int)((u32)mine.key & p.node_mask);
            u32 rank = 0;
            for (u32 j = 0; j < nt; ++j) {
                const Cand o = t.tie[j];
                const int ok = (int)((u32)o.key & p.node_mask);
                rank += (o.bits > mine.bits || (o.bits == mine.bits && ok < mk)) ? 1u : 0u;
            }
            if (rank < want) t.sel[n0 + rank] = mine;
        }
        GP_SYNC();
    }
    if ((u32)tid < need) {                                                            // (K <= 128 <= BLOCK)
        const Cand cd = t.sel[tid];
        const int ck = (int)((u32)cd.key & p.node_mask);
        u32 rank = 0;
        for (u32 j = 0; j < need; ++j) {
            const Cand o = t.sel[j];
            const int ok = (int)((u32)o.key & p.node_mask);
            rank += (o.bits > cd.bits || (o.bits == cd.bits && ok < ck)) ? 1u : 0u;
        }
        my_rank = rank;
        if (rank + 1u == need) ctl->kth_bits = cd.bits;
    }
    GP_SYNC();
}

// Select the K largest (value desc, column asc) positive totals of the aggregation table into sel[0 .. need) (graph.h:111-121),
// rank them (thread i < need: my_rank = position of sel[i] in the output order) and note the smallest in ctl->kth_bits.  Returns
// need = min(K, positive totals); 0xFFFFFFFF: the row must leave (a total outside [2^-63, 2)).  Every thread of the workgroup calls
// this.  Passes walk the table four slots per lane and step (128-bit LDS reads); every wave reads the small histograms itself, so no
// result is broadcast through LDS behind extra barriers; and the bucket of the K-th entry is narrowed 8 value bits at a time until
// a handful is left to rank by comparison (one binade holds a hundred entries of a MAG row: ranking them cost more than a pass).
constexpr u32 kSkNarrow = 16;
template <class F>
__device__ __forceinline__ void sk_table_walk(const SkTop& t, int block, F f)      // f(key, value bits) for every positive entry
{
    typedef int    i4 __attribute__((ext_vector_type(4)));
    typedef double d2 __attribute__((ext_vector_type(2)));
    for (u32 s0 = 4u * threadIdx.x; s0 < t.CA; s0 += 4u * (u32)block) {              // (CA % 4 == 0)
        const i4 kk = *(const i4*)&t.akeys[s0];
        if (kk.x == kEmpty && kk.y == kEmpty && kk.z == kEmpty && kk.w == kEmpty) continue;
        const d2 va = *(const d2*)&t.avals[s0], vb = *(const d2*)&t.avals[s0 + 2];
        if (kk.x != kEmpty && va.x > 0.0) f(kk.x, (u64)__double_as_longlong(va.x));
        if (kk.y != kEmpty && va.y > 0.0) f(kk.y, (u64)__double_as_longlong(va.y));
        if (kk.z != kEmpty && vb.x > 0.0) f(kk.z, (u64)__double_as_longlong(vb.x));
        if (kk.w != kEmpty && vb.y > 0.0) f(kk.w, (u64)__double_as_longlong(vb.y));
    }
}
template <int BLOCK>
__device__ __forceinline__ u32 sk_select(KP p, CtlS* ctl, const SkTop& t, u32& my_rank)
{
    const int tid = threadIdx.x, lane = tid & 63;
    const u32 K = (u32)p.K;
    sk_table_walk(t, BLOCK, [&](int, u64 bits) {
        const int e = 1023 - (int)(bits >> 52);
        if ((u32)e < 64u) lds_add_u32(&ctl->bcnt[e], 1u); else ctl->tk_wide = 1u;
    });
    GP_SYNC();
    if (uni(ctl->tk_wide)) return 0xFFFFFFFFu;
    u32 above, cnt_b, b_sel, need;
    {                                                                                 // lane = binade, 0 holds the largest values
        const u32 cn = ctl->bcnt[lane];
        const u32 incl = wave_incl_scan_dpp(cn);
        const u32 total = (u32)__builtin_amdgcn_readlane((int)incl, 63);
        need = min(K, total);
        if (need == 0) return 0;
        const u64 mk = __ballot(incl >= need);
        b_sel = (u32)(__ffsll((long long)mk) - 1);
        above = (u32)__builtin_amdgcn_readlane((int)(incl - cn), (int)b_sel);
        cnt_b = (u32)__builtin_amdgcn_readlane((int)cn, (int)b_sel);
    }
    // While more than kSkNarrow entries crowd the bucket of the K-th entry, narrow it: 8 more bits of the value at a time (larger
    // first), then -- a FLAT row: a seed whose neighbour is a hub hands thousands of nodes totals that differ by summation-order
    // ulps or not at all -- 8 bits of the node id at a time (smaller first: the output order is value desc, column asc).
    // The bucket is { entries with (bits >> vshift) == vpre and, once vshift == 0, (id >> ishift) == ipre }.  Two histogram
    // buffers alternate, so the next round's zeroing never meets this round's readers.
    u32 vshift = 52; u64 vpre = (u64)(1023u - b_sel);
    u32 ishift = 8u * (u32)(((int)p.deg_shift + 7) / 8), ipre = 0, buf = 0;
    while (above + cnt_b > need && cnt_b > kSkNarrow && (vshift > 0u || ishift > 0u)) {
        const bool by_value = vshift > 0u;
        const u32 ds = by_value ? min(8u, vshift) : 8u;
        const u32 vs_hi = vshift, is_hi = ishift;                                     // the bucket being split
        if (by_value) vshift -= ds; else ishift -= ds;
        u32* fine = t.fine + 256u * buf; buf ^= 1u;
        for (u32 i = tid; i < 256u; i += BLOCK) fine[i] = 0;
        GP_SYNC();
        sk_table_walk(t, BLOCK, [&](int key, u64 bits) {
            const u32 id = (u32)key & p.node_mask;
            if ((bits >> vs_hi) == vpre && (by_value || is_hi >= 32u || (id >> is_hi) == ipre)) {
                const u32 dg = by_value ? (u32)(bits >> vshift) & ((1u << ds) - 1u) : (id >> ishift) & 255u;
                lds_add_u32(&fine[by_value ? 255u - dg : dg], 1u);                   // bin order = output order
            }
        });
        GP_SYNC();
        {                                                                             // lane j owns bins [4 j, 4 j + 4), best first
            u32 c4[4], sum = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) { c4[j] = fine[4 * lane + j]; sum += c4[j]; }
            const u32 incl = wave_incl_scan_dpp(sum) + above;                         // entries in bins <= 4 lane + 3, and everything above the bucket
            const u64 m = __ballot(incl >= need);
            const int cl = m ? __ffsll((long long)m) - 1 : 63;
            u32 acc = incl - sum, cj = c4[3]; int js = 3;                            // (lane cl always finds its bin)
#pragma unroll
            for (int j = 0; j < 4; ++j) { if (acc + c4[j] >= need) { js = j; cj = c4[j]; break; } acc += c4[j]; }
            const u32 bin = 4u * (u32)cl + (u32)__builtin_amdgcn_readlane(js, cl);
            above = (u32)__builtin_amdgcn_readlane((int)acc, cl);
            cnt_b = (u32)__builtin_amdgcn_readlane((int)cj, cl);
            if (by_value) vpre = (vpre << ds) | (u64)(255u - bin); else ipre = (ipre << 8) | bin;
        }
    }
    if (above + cnt_b > need && cnt_b > kSkTie) return 0xFFFFFFFFu;                   // (cannot happen: ids are unique)
    // collect: strictly above the K-th bucket -> sel; inside it -> tie (all of it goes to sel when it fits exactly)
    const bool take_all = above + cnt_b <= need;
    {
        typedef int    i4 __attribute__((ext_vector_type(4)));
        typedef double d2 __attribute__((ext_vector_type(2)));
        for (u32 base = 0; base < t.CA; base += 4u * BLOCK) {
            const u32 s0 = base + 4u * (u32)tid;
            i4 kk = {kEmpty, kEmpty, kEmpty, kEmpty}; d2 va = {0.0, 0.0}, vb = {0.0, 0.0};
            if (s0 < t.CA) { kk = *(const i4*)&t.akeys[s0]; va = *(const d2*)&t.avals[s0]; vb = *(const d2*)&t.avals[s0 + 2]; }
            const double v[4] = {va.x, va.y, vb.x, vb.y};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (__ballot(kk[j] != kEmpty && v[j] > 0.0) == 0) continue;           // wave-uniform
                bool is_sel = false, is_t = false;
                Cand cd; cd.bits = (u64)__double_as_longlong(v[j]); cd.key = kk[j]; cd.pad = 0;      // (the PACKED key: a merged entry goes back into a table)
                if (kk[j] != kEmpty && v[j] > 0.0) {
                    const u32 id = (u32)cd.key & p.node_mask;
                    const u64 vbits = cd.bits >> vshift;
                    const bool v_eq = vbits == vpre;
                    const bool in_bin = v_eq && (ishift >= 32u || (id >> ishift) == ipre);
                    const bool over = vbits > vpre || (v_eq && ishift < 32u && (id >> ishift) < ipre);
                    is_sel = over || (in_bin && take_all); is_t = in_bin && !take_all;
                }
                const u32 si = wave_alloc1(&ctl->n_sel, is_sel, lane);
                if (is_sel) t.sel[si] = cd;
                const u32 ti = wave_alloc1(&ctl->n_tie, is_t, lane);
                if (is_t) t.tie[ti] = cd;
            }
        }
    }
    GP_SYNC();
    sk_rank_selected<BLOCK>(p, ctl, t, need, take_all, my_rank);
    return need;
}

// TG: coef * share of the row's pushers does not fit LDS beside a useful aggregation table (a hub's thousands of leaves
// all pushed): the sweep gathers it from HBM instead (L2-hot: 8 bytes per pusher, re-read once per record).
template <int BLOCK, bool TG>
__device__ GP_PHASE_NOINLINE void phase_sk_topk(u32 lds0, u32 row_lo, u32 row_hi, int seed, u32 n_pu, u32 n_log)
{
    KP p = kparams();
    lds0 = uni(lds0); row_lo = uni(row_lo); row_hi = uni(row_hi); seed = uni(seed); n_pu = uni(n_pu); n_log = uni(n_log);
    const SkView w = sk_view<BLOCK>(p, lds0);
    CtlS* ctl = w.ctl;
    const SkTop t = sk_top(p, w, TG ? 0u : n_pu);
    const long long row = (long long)(((u64)row_hi << 32) | row_lo);
    const int tid = threadIdx.x, lane = tid & 63;
    const u32 wave = wave_id();
    const u32 K = (u32)p.K;

    // T = coef[level] * share of every pusher of the row (graph.h:90 / :109 per edge: what a record adds to its target's reserve).
    // SCAN left them in HBM with the push-list entries; the level sketch's bytes they land in are dead.
    if (!TG) for (u32 i = tid; i < n_pu; i += BLOCK) t.T[i] = w.arch[i];
    for (u32 i = tid; i < 512u; i += BLOCK) t.fine[i] = 0;
    // (the first round's aggregation table and select counters are made ready here, in front of a barrier that is there anyway:
    //  two barriers fewer per row than wiping at the top of the round)
    for (u32 i = tid; i < t.CA; i += BLOCK) { t.akeys[i] = kEmpty; t.avals[i] = 0.0; }
    if (tid < 64) ctl->bcnt[tid] = 0;
    if (tid == 0) { ctl->ovf = 0; ctl->tk_t = 1u; ctl->n_sel = 0; ctl->n_tie = 0; ctl->tk_wide = 0; ctl->kth_bits = ~0ull; }
    bool table_ready = true;
    GP_SYNC();
    // ---- t_c: the cell value of rank ~ target, off a histogram over (binade, top 4 mantissa bits) of the cells
    for (u32 i = tid; i < t.MR; i += BLOCK) {
        const u32 cv = t.R[i];
        if (cv) {
            const u32 fb = __float_as_uint((float)cv);                                // >= 1.0f: exponent 127 .. 159
            lds_add_u32(&t.fine[min(511u, (fb >> 19) - (127u << 4))], 1u);
        }
    }
    GP_SYNC();
    u32 t_c;
    {                                                                                 // every wave reads the histogram itself: lane j owns bins [8 j, 8 j + 8)
        u32 cnt8[8], sum = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) { cnt8[j] = t.fine[8 * lane + j]; sum += cnt8[j]; }
        const u32 suf = wave_suffix_scan(sum, lane);                                  // cells in bins >= 8 * lane
        const u64 m = __ballot(suf >= p.sk_target);
        u32 edge = 1u;
        if (m != 0) {
            const int cl = 63 - __builtin_clzll(m);                                   // highest chunk whose suffix reaches the target
            u32 acc = suf - sum; int js = 0;
#pragma unroll
            for (int j = 7; j >= 0; --j) { acc += cnt8[j]; if (acc >= p.sk_target) { js = j; break; } }
            const u32 bin = 8u * (u32)cl + (u32)__builtin_amdgcn_readlane(js, cl), e = bin >> 4, mt = 16u + (bin & 15u);
            edge = max(e >= 4u ? mt << (e - 4u) : mt >> (4u - e), 1u);                // lower edge of the bin (rounded down)
        }
        t_c = edge;
    }
    SKT(ctl, 7);
    u32 need = 0, my_rank = 0;
    bool last = false;                                                                // t_c is a proven bound: what this round selects is final
    for (int round = 0; ; ++round) {
        // One round = every node whose cell reaches t_c, tabled exactly and the K best selected; in P hash partitions when
        // they outgrow the table, each partition's K best merged with the running list (it goes back into the table).
        u32 P = 1;
        for (;;) {
            bool ovf = false;
            for (u32 part = 0; part < P; ++part) {
                const u32 n_run = part == 0 ? 0u : need;
                Cand mine; mine.bits = 0; mine.key = kEmpty; mine.pad = 0;
                if ((u32)tid < n_run) mine = t.sel[tid];                              // (K <= 128 <= BLOCK)
                if (!table_ready) {
                    GP_SYNC();
                    for (u32 i = tid; i < t.CA; i += BLOCK) { t.akeys[i] = kEmpty; t.avals[i] = 0.0; }
                    if (tid < 64) ctl->bcnt[tid] = 0;
                    if (tid == 0) { ctl->n_sel = 0; ctl->n_tie = 0; ctl->tk_wide = 0; ctl->kth_bits = ~0ull; }
                    GP_SYNC();
                }
                table_ready = false;
                if (mine.key != kEmpty && !sk_res_add_lds(t.akeys, t.avals, t.CA, mine.key, __longlong_as_double((long long)mine.bits))) ctl->ovf = 1;
                auto tabled = [&](const int (&k)[4], const u32 (&pu)[4], auto all_valid) {
                    constexpr bool kAll = decltype(all_valid)::value;                 // every record of the group is one: no guards
                    u32 cell[4]; double cv[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {                                     // eight lookups in flight
                        cell[q] = t.R[sk_cell(kAll ? (u32)k[q] : (u32)max(k[q], 0)) >> t.shR];
                        cv[q] = TG ? w.arch[pu[q]] : t.T[pu[q]];
                    }
                    int kh[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) kh[q] = (kAll || k[q] >= 0) && cv[q] != 0.0 && cell[q] >= t_c ? k[q] : -1;
                    insert_windows4_asm<true>(t.akeys, t.avals, t.CA, &ctl->ovf, kh, cv, P, part);               // graph.h:90 / :109
                };
#if GP_SK_SWEEP_DEPTH == 2
                log_groups2_nt<BLOCK>(w.log_key, w.log_pu, n_log, tabled);
#else
                log_groups<BLOCK, true>(w.log_key, w.log_pu, 0u, n_log, tabled);
#endif
                GP_SYNC();
                SKT(ctl, 8);
                if (uni(ctl->ovf)) { ovf = true; break; }
                need = sk_select<BLOCK>(p, ctl, t, my_rank);
                SKT(ctl, 9);
                if (need == 0xFFFFFFFFu) { if (tid == 0) ctl->fail = 3; GP_SYNC(); return; }      // (3: the select gave up)
            }
            if (!ovf) break;
            GP_SYNC();
            if (tid == 0) ctl->ovf = 0;
            P *= 2; need = 0;
            if (P > 64u) { if (tid == 0) ctl->fail = 4; GP_SYNC(); return; }          // (4: tens of thousands of nodes around the K-th value)
            GP_SYNC();
        }
        // Complete?  An unswept node sits in a cell below t_c, so its total * scale < t_c.  With K exact totals selected and
        // the K-th of them at tau: tau * scale * (1 - 2^-20) >= t_c puts every unswept node strictly below tau.
        if (t_c <= 1u || last) break;
        if (need == K) {
            const double tau_fx = __builtin_floor(__longlong_as_double((long long)uni(ctl->kth_bits)) * p.sk_rscale * (1.0 - 1.0 / 1048576.0));
            if (tau_fx >= (double)t_c) break;
            t_c = tau_fx >= 1.0 ? (u32)tau_fx : 1u;                                    // tau is a proven lower bound on the K-th total
            last = true;
        } else {
            t_c = round >= 2 ? 1u : max(1u, t_c >> 4);                                // fewer than K nodes up there: look lower
        }
        if (tid == 0) zstat(ctl, zSweep2, 1);
    }
    // write the row in output order (value desc, column asc): keys are unit numbers, which grow with the node ids they stand for
    const long long out0 = row * (long long)p.K;
    if ((u32)tid < need) {
        const Cand cd = t.sel[tid];
        const int node = p.unit_info[(u32)cd.key & p.node_mask];
        __builtin_nontemporal_store(seed, &p.out_row[out0 + my_rank]);                                          // graph.h:122
        __builtin_nontemporal_store(node, &p.out_col[out0 + my_rank]);                                          // graph.h:123
        __builtin_nontemporal_store(__longlong_as_double((long long)cd.bits), &p.out_val[out0 + my_rank]);     // graph.h:124
    }
    publish_filled(p, row, need);
    if (tid == 0) zstat(ctl, zFilled, need);
}

// ---------------------------------------------------------------- the row loop
// (The row loop itself holds nothing per thread: whatever needs threadIdx-derived addresses is a function of its own.  Hoisted into
//  the loop those values live across every phase call, the 24 callee-saved registers of an 80-VGPR budget do not hold them, and
//  every reload from scratch is a memory round trip on the row's critical path -- 18 scratch operations per level measured +8 %.)
// what: 1 = R, U and the whole exact table (row end), 2 = the first n slots of the exact table (an overflowed pass),
//       3 = U and n share-table values behind slot `at` (a split partition walk of a sketch level)
template <int BLOCK>
__device__ GP_PHASE_NOINLINE void phase_sk_wipe(u32 lds0, u32 what, u32 n, u32 at)
{
    typedef u32 u4 __attribute__((ext_vector_type(4)));
    KP p = kparams();
    lds0 = uni(lds0); what = uni(what); n = uni(n); at = uni(at);
    const SkView w = sk_view<BLOCK>(p, lds0);
    const u4 z = {0u, 0u, 0u, 0u};
    if (what == 1u) {
        for (u32 i = 4u * threadIdx.x; i < w.MR + w.MU; i += 4u * BLOCK) *(u4*)&w.R[i] = z;     // R and U are adjacent
        n = w.CX;
    }
    if (what == 3u) {
        for (u32 i = 4u * threadIdx.x; i < w.MU; i += 4u * BLOCK) *(u4*)&w.U[i] = z;
        for (u32 i = threadIdx.x; i < n; i += BLOCK) w.xvals[at + i] = 0.0;
        return;
    }
    for (u32 i = threadIdx.x; i < n; i += BLOCK) { w.xkeys[i] = kEmpty; w.xvals[i] = 0.0; }
}

// What a level hands to the next one sits in LDS, not in registers of the row loop: SCAN's counters in lc[level & 1] (push list
// entries and edges, dangling mass) and, in that struct's spare word, the row's running pusher number and log position, which
// thread 0 writes at the top of the level (like the counters' reset: the parity scheme keeps a slow wave's reads of the other set
// apart).  Every wave derives the level's plan from those with scalar code; the row loop carries nothing but the level number.
__device__ __forceinline__ u64 sk_io_pack(u32 pu_next, u32 log_next) { return ((u64)log_next << 32) | pu_next; }

// Level 0 of a row: the frontier is { seed : 1.0 } (graph.h:81): its record, push test and push-list entry directly; what it
// leaves for level 1 goes where every level leaves it: lc[0].
// Returns 1 when level 1 is done as well (round 6): a seed of <= 256 edges that pushes makes level 1 a one-wave level whatever
// else is true, so wave 0 does it straight behind level 0 -- same call, no barrier, no level prologue of twelve waves, the seed's
// entry in registers -- and the other waves, who know the seed's degree too, go to the one barrier.  The row goes on at level 2.
template <int BLOCK>
__device__ GP_PHASE_NOINLINE u32 phase_sk_seed(u32 lds0, int seed)
{
    KP p = kparams();
    lds0 = uni(lds0); seed = uni(seed);
    const SkView w = sk_view<BLOCK>(p, lds0);
    CtlS* ctl = w.ctl;
    const int tid = threadIdx.x;
    const u32 L = (u32)p.n_coef - 1u;
    const u32 seed_unit = uni(p.node_pos[seed]);
    const u32 seed_deg = uni((u32)p.indptr[seed + 1]) - uni((u32)p.indptr[seed]);
    const u32 s_start = seed_unit << kSkUnitShift;
    const int seed_key = (int)(seed_unit | (min(seed_deg, p.deg_sat) << p.deg_shift));
    const u32 pu_cap = (u32)min((u64)kSkMaxPushers, p.arch_cap);
    PushEntry* push1 = w.push2 + (u64)(u32)p.push_cap;
    u32* bt1 = w.bt2 + (u64)(u32)p.bt_cap;
    const bool room = p.log_cap > 0 && pu_cap >= 2u && p.push_cap > 0;
    bool pushes = false; double share = 0.0;
    if (L > 0 && seed_deg != 0 && 1.0 >= p.rmax * (double)seed_deg) {                // graph.h:94
        share = 1.0 / (double)seed_deg;                                               // graph.h:95
        pushes = share != 0.0;
        if (tid == 0) { zstat(ctl, zPush, 1); zstat(ctl, zEdges, seed_deg); }
    }
    const u32 units = (seed_deg + (1u << kUnitShift) - 1u) >> kUnitShift;
    // level 1 in this call?  (what phase_sk_level would decide for it, from values every wave has: nothing of it depends on LDS
    //  another thread writes in this phase)
    const u32 k_lc = uni(ctl->k_log_cap), k_so = uni(ctl->k_solo_ok), k_dm = uni(ctl->k_direct_max);
    const bool merged = (p.solo & 2u) != 0u && k_so != 0u && pushes && room && L >= 2u && seed_deg <= kSkSoloEdges && seed_deg <= k_dm &&
                        (u64)units <= p.bt_cap && (u64)1u + seed_deg <= (u64)k_lc && (u64)2u + kSkSoloEdges + 4u <= (u64)pu_cap;
    if (merged && wave_id() != 0u) return 1u;
    if (tid == 0) {
        ctl->seed_key = seed_key; ctl->tot_pu = 1; ctl->tot_log = 1;
        if (room) { w.log_key[0] = seed_key; w.log_pu[0] = 0; w.arch[0] = ctl->coef[0]; }      // graph.h:90
        if (!room || (pushes && (u64)units > p.bt_cap)) ctl->fail = 1;
        lds_add_u32(&w.R[sk_cell((u32)seed_key) >> w.shR], fx_up(ctl->coef[0] * p.sk_rscale));
        zstat(ctl, zLevels, 1);
        LevelCtr* l0 = &ctl->lc[0];
        l0->dangling = L > 0 && seed_deg == 0 ? 1.0 : 0.0; l0->n_dangling = L > 0 && seed_deg == 0 ? 1u : 0u;   // graph.h:91-93
        l0->n_rec = 0; l0->alloc = pushes ? ((u64)seed_deg << 32) | 1ull : 0ull; l0->pad1 = sk_io_pack(1u, 1u);
        if (pushes && room) {
            PushEntry pe; pe.rel = s_start; pe.off = 0; pe.share = share; push1[0] = pe;
            w.arch[1] = ctl->coef[1] * share;
        }
    }
    if (pushes && (u64)units <= p.bt_cap)
        for (u32 m = (u32)tid; m < units; m += BLOCK) { bt1[m] = 0u; if (m < w.bt_l_cap) w.bt_l[m] = 0u; }
    if (!merged) return 0u;
    // ---- level 1 by this wave (wave 0), as phase_sk_level would run it: the prologue's writes, then the one-wave level
    if (tid == 0) {
        LevelCtr* nx = &ctl->lc[1];
        nx->dangling = 0.0; nx->n_dangling = 0; nx->n_rec = 0; nx->alloc = 0ull; nx->pad1 = sk_io_pack(2u, 1u + seed_deg);
        ctl->tot_pu = 2u; ctl->tot_log = 1u + seed_deg; ctl->max_e = max(ctl->max_e, seed_deg);
        ctl->lv_has_dang = 0u; ctl->lv_dang = 0.0;
        zstat(ctl, zLevels, 1);
    }
    __atomic_signal_fence(__ATOMIC_SEQ_CST);                       // (one wave: its LDS operations execute in program order)
    phase_sk_solo<BLOCK, true>(lds0, 1u, 1u, seed_deg, 1u, ctl->coef[1] * ctl->k_rscale, 0u, 0.0, seed_key, 1u, 1u, 2u, ctl->coef[2], s_start, share);
    return 1u;
}

// One level l = 1..L of a row (graph.h:83-110).  Returns 0 when the row's levels are done (last level, dead frontier, or the row
// leaves: ctl->fail), 1 to go on.  Every path out has passed the level's last barrier or has touched nothing shared.
template <int BLOCK>
__device__ __forceinline__ u32 phase_sk_level(u32 lds0, u32 lvl, u32 L)
{
    KP p = kparams();
    lds0 = uni(lds0); lvl = uni(lvl); L = uni(L);
    const SkView w = sk_view<BLOCK>(p, lds0);
    CtlS* ctl = w.ctl;
    const int tid = threadIdx.x;
    const LevelCtr* in = &ctl->lc[(lvl & 1u) ^ 1u];
    LevelCtr* nx = &ctl->lc[lvl & 1u];
    // ---- one batch of LDS reads: what the previous level left behind its last barrier, and the launch constants
    const u64 al_ = in->alloc, io_ = in->pad1; const u32 nd_ = in->n_dangling, fail_ = ctl->fail, q_ = ctl->cand_q[lvl]; const int sk_ = ctl->seed_key;
    const double dg_ = in->dangling, c_ = ctl->coef[lvl], c1_ = ctl->coef[lvl + 1u], krs_ = ctl->k_rscale;
    const u32 klc_ = ctl->k_log_cap, kpc_ = ctl->k_pu_cap, kdm_ = ctl->k_direct_max, kso_ = ctl->k_solo_ok, kcx_ = ctl->k_cx;
    const u32 CX = uni(kcx_);
    const u64 al = uni(al_), io = uni(io_);
    const u32 n_ent_cur = (u32)al, e_cur = (u32)(al >> 32), pu_cur = (u32)io, log_pos = (u32)(io >> 32);
    const bool has_dang_cur = uni(nd_) != 0u;
    const double dang_cur = has_dang_cur ? uni(dg_) : 0.0;
    const int seed_key = uni(sk_);
    const u32 n_rec = e_cur + (has_dang_cur ? 1u : 0u);                           // log records (= pushed edges) of this level
    if (n_rec == 0 || uni(fail_)) return 0u;                                      // the frontier died: later levels add nothing
    const bool last = lvl == L;                                                   // graph.h:104-110: no push from the last level
    const u32 pu_cap = uni(kpc_);                                                 // pusher numbers the row may hand out
    // (fail = 1: a slab bound, counted for the host's slab sizing; anything else has its own number)
    if ((u64)log_pos + n_rec > (u64)uni(klc_) || (u64)pu_cur + n_ent_cur + 1u > pu_cap) { if (tid == 0) ctl->fail = 1; return 0u; }
    const u32 seg_base = log_pos;
    const u32 pu_next = pu_cur + n_ent_cur + (has_dang_cur ? 1u : 0u);            // pusher number of the NEXT list's entry 0
    const u32 cur = lvl & 1u;                                                     // the push list this level streams (level 0 wrote list 1)
    if (tid == 0) {
        nx->dangling = 0.0; nx->n_dangling = 0; nx->n_rec = 0; nx->alloc = 0ull; nx->pad1 = sk_io_pack(pu_next, log_pos + n_rec);
        ctl->tot_pu = pu_next; ctl->tot_log = log_pos + n_rec; ctl->max_e = max(ctl->max_e, e_cur);
        ctl->lv_has_dang = has_dang_cur ? 1u : 0u; ctl->lv_dang = dang_cur;
        if (has_dang_cur) w.arch[pu_cur + n_ent_cur] = uni(c_) * dang_cur;        // graph.h:92: the record of the mass returned to the seed
        zstat(ctl, zLevels, 1);
    }
    const double cs = uni(c_) * uni(krs_);                                        // reserve-sketch units per unit of share
    const double cnext = last ? 0.0 : uni(c1_);
    SKT(ctl, 6);
    if (last) {
        phase_sk_stream<BLOCK, 2>(lds0, cur, n_ent_cur, e_cur, seg_base, cs, 0u, pu_cur, 1u, 0u);
        SKT(ctl, 3);
        return 0u;
    }
    // a level goes straight into the exact table while its edges would fill three quarters of it (its nodes: fewer; cap 1/2 / 0.65 / 0.8 / 1
    // of the slots measured 23.50 / 23.42 / 23.36 / 23.85 ms)
    const u32 direct_max = uni(kdm_);                                             // (= min(sk_direct_max, 3/4 of the slots))
    // a sketch level keeps its share table (one value per pusher, one for the dangling mass) behind the slots it uses; a level
    // with more pushers than that leaves room for (a hub's thousands of leaves all push) goes without the sketch, whatever its size
    const u32 s_n = n_ent_cur + 1u;
    const bool direct = n_rec <= direct_max || s_n + kMinCap > CX;
    u32 capx = direct ? min(CX, max(kMinCap, (4u * n_rec + 3u) & ~3u)) : (CX - s_n) & ~3u;
    const u32 cap0 = capx;                                                        // (where the share table starts)
    // Partitions planned so that the nodes expected in the exact table -- per pushed edge what this workgroup's earlier
    // rows tabled at this level, x 1.25 -- fit its slots: an overflowed pass costs a whole pass, a planned partition one
    // too (measured, planning for a load of 0.5 / 0.6 / 0.7 / 0.8 / 1.0 / 1.2: 24.75 / 24.2 / 23.95 / 23.75 / 23.54 / 23.57 ms)
    u32 P0 = 1;
    {
        const u32 q = uni(q_);
        const u32 est = direct ? (n_rec > direct_max ? 2u * n_rec : 0u) : (u32)(((u64)n_rec * q) >> 10);   // (without the sketch every edge is an insert: half a table per partition)
        while (P0 < 64u && est > P0 * capx) ++P0;                                 // (P0 > 1 is rare: no division on the common path)
    }
    // a small level: one wave does it, the others park at one barrier (phase_sk_solo).
    // (a one-wave level must not be able to hit a workspace bound other than through its own checks: the boundary table must
    //  hold the largest level any frontier can produce -- degrees pushed in one level sum to <= 1/rmax)
    //  -- that part of the test does not change during a launch: k_solo_ok)
    const bool solo = uni(kso_) && direct && e_cur <= kSkSoloEdges && n_ent_cur >= 1u && n_ent_cur <= 64u &&
                      (u64)pu_next + kSkSoloEdges + 4u <= pu_cap;
    if (solo) {
        if (wave_id() == 0)
            phase_sk_solo<BLOCK>(lds0, cur, n_ent_cur, e_cur, seg_base, cs, has_dang_cur ? 1u : 0u, dang_cur, seed_key, lvl & 1u, pu_cur, pu_next, cnext);
        GP_SYNC();
        SKT(ctl, 1); SKT_COUNT(ctl, 13, 1);
    } else {
        SKT2_BEGIN(ctl);
        if (direct && P0 == 1u) phase_sk_stream<BLOCK, 1>(lds0, cur, n_ent_cur, e_cur, seg_base, cs, capx, pu_cur, 1u, 0u);
        else if (direct)        phase_sk_stream<BLOCK, 2>(lds0, cur, n_ent_cur, e_cur, seg_base, cs, capx, pu_cur, 1u, 0u);     // (records only: every partition is a MODE 3 pass below)
        else                    phase_sk_stream<BLOCK, 0>(lds0, cur, n_ent_cur, e_cur, seg_base, cs, capx, pu_cur, 1u, 0u);
        SKT2(ctl, 11);
        GP_SYNC();
        SKT2(ctl, 12);
        SKT(ctl, direct ? 1 : 2); SKT_COUNT(ctl, direct ? 13 : 14, 1);
    }
    // Exact inserts, then SCAN.  A table that overflows is wiped and the level's candidates are walked in hash
    // partitions (q of P, split in two in place), exactly as the general kernel refines its partitions: a sketch level
    // re-reads its log segment, a small level re-streams its edges.
    const bool by_log = !direct;
    bool u_dirty = !direct, first = true;
    u32 part = 0, np = P0;
    bool level_done = false;
    if (solo) {
        if (!uni(ctl->ovf)) level_done = true;                                    // (the wave wrote the next push list and lc[lvl & 1] itself)
        else { capx = CX; }                                                       // undone: re-stream the level (first pass below sees ctl->ovf)
    }
    if (!level_done)
    for (;;) {
        if (!(first && direct && P0 == 1u)) {
            SKT2_BEGIN(ctl);
            if (by_log) phase_sk_filter<BLOCK>(lds0, seg_base, n_rec, capx, pu_cur, np, part);
            else        phase_sk_stream<BLOCK, 3>(lds0, cur, n_ent_cur, e_cur, seg_base, cs, capx, pu_cur, np, part);
            SKT2(ctl, 2);
            GP_SYNC();
            SKT2(ctl, 3);
            SKT(ctl, 4); SKT_COUNT(ctl, 11, 1);
        }
        first = false;
        if (uni(ctl->ovf)) {
            phase_sk_wipe<BLOCK>(lds0, 2u, by_log ? cap0 : CX, 0u);                              // (a sketch level's share table lives behind slot cap0)
            GP_SYNC();
            if (tid == 0) ctl->ovf = 0;
            if (!by_log) capx = CX;
            if (np < 0x10000u) { part *= 2; np *= 2; GP_SYNC(); continue; }
            if (tid == 0) ctl->fail = 5;                                          // (5: a level's candidates in > 65 536 partitions)
            GP_SYNC();
            break;
        }
        const bool final_part = np == P0 && part + 1u == P0;                      // nothing reads the sketch after this partition
        SKT2_BEGIN(ctl);
        phase_sk_scan<BLOCK>(lds0, capx, lvl & 1u, cur ^ 1u, final_part && u_dirty ? s_n : 0u, pu_next, cnext);
        if (final_part) u_dirty = false;
        SKT2(ctl, 9);
        GP_SYNC();
        SKT2(ctl, 10);
        SKT(ctl, 5); SKT_COUNT(ctl, 12, 1);
        if (uni(ctl->fail)) break;
        while (np > P0 && (part & 1u)) { part >>= 1; np >>= 1; }
        ++part;
        if (np == P0 && part == P0) break;
    }
    if (u_dirty) {                                                                // (a split partition walk of a sketch level)
        phase_sk_wipe<BLOCK>(lds0, 3u, s_n, cap0);
        GP_SYNC();
    }
    if (uni(ctl->fail)) return 0u;
    if (tid == 0 && !direct) {                                                    // nodes the exact table held per pushed edge, x 1.25, decaying maximum
        const u32 obs = min(2048u, (u32)(1280.0f * (float)nx->n_rec * __frcp_rn((float)n_rec)) + 8u);
        const u32 old_q = ctl->cand_q[lvl];
        ctl->cand_q[lvl] = max(obs, old_q - (old_q >> 3));
    }
    return 1u;
}

template <int BLOCK>
__device__ __forceinline__ void gfpush_sk_rows()
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    KP p = kparams();
    const u32 lds0 = uni(lds_addr(smem));
    const SkView w = sk_view<BLOCK>(p, lds0);
    CtlS* ctl = w.ctl;
    const int tid = threadIdx.x;
    phase_sk_wipe<BLOCK>(lds0, 1u, 0u, 0u);
    if (tid < 8) { ctl->st[tid] = 0; ctl->st_row[tid] = 0; }
#ifdef GP_SK_TIMING
    if (tid < 16) ctl->tacc[tid] = 0;
    if (tid < 14) ctl->tacc2[tid] = 0;
#endif
    if (tid < kSkMaxCoef) { ctl->cand_q[tid] = 0; if (tid < p.n_coef) ctl->coef[tid] = p.coef[tid]; }
    if (tid == 0) {
        ctl->max_e = 0; ctl->max_log = 0;
        ctl->k_log_cap = (u32)min(p.log_cap, (u64)0xFFFFFFFFu); ctl->k_pu_cap = (u32)min((u64)kSkMaxPushers, p.arch_cap);
        ctl->k_direct_max = min(p.sk_direct_max, 3u * (w.CX / 4u)); ctl->k_cx = w.CX; ctl->k_rscale = p.sk_rscale;
        // a one-wave level must not be able to hit a workspace bound other than through its own checks: the boundary table must
        // hold the largest level any frontier can produce -- degrees pushed in one level sum to <= 1/rmax, SURVEY.md A.1
        ctl->k_solo_ok = (p.solo & 1u) && p.push_cap >= (u64)kSkSoloEdges + 4u &&
                         (double)p.bt_cap >= (p.rmax > 0.0 ? fmin((double)p.nnz, 1.001 / p.rmax + 16.0) : (double)p.nnz) / (double)(1u << kUnitShift) + 4.0 ? 1u : 0u;
    }
    const long long n_rows = p.n_seeds;
    const u32 L = (u32)p.n_coef - 1u;

    for (;;) {
        GP_SYNC();
        SKT_BEGIN(ctl);
        if (tid == 0) {
            ctl->row = (long long)__hip_atomic_fetch_add(&p.counters[p.queue_counter], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ctl->fail = 0; ctl->ovf = 0;
        }
        GP_SYNC();
        const long long row = uni(ctl->row);
        if (row >= n_rows) break;
        const int seed = uni(p.seeds[row]);
        if (seed < 0 || seed >= p.n_nodes) {            // the device API does not pre-validate seeds
            if (tid == 0) { __hip_atomic_fetch_add(&ctl->st[zFailed], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); if (p.out_filled) p.out_filled[row] = 0; }
            continue;
        }
        u32 lvl = 1u + uni(phase_sk_seed<BLOCK>(lds0, seed));                     // (2: wave 0 has done level 1 as well)
        GP_SYNC();
        SKT(ctl, 0); SKT_COUNT(ctl, 15, 1);
        if (lvl == 2u) { SKT_COUNT(ctl, 13, 1); if (uni(ctl->ovf)) lvl = 1u; }   // (an insert at the probe limit: the claims are undone, level 1 takes the general path)
        for (; lvl <= L; ++lvl)
            if (!uni(phase_sk_level<BLOCK>(lds0, lvl, L))) break;
        GP_SYNC();
        SKT(ctl, 6);
        if (!uni(ctl->fail)) {
            const u32 n_pu = uni(ctl->tot_pu), n_log = uni(ctl->tot_log);
            if (tid == 0) ctl->max_log = max(ctl->max_log, n_log);
            // (T beside at least half of the aggregation table TOP-K has without it; else the gathering form)
            const u32 top_region = 4u * w.MU + 12u * w.CX;
            if (2u * sk_top_slots(top_region, n_pu, (u32)p.K) >= sk_top_slots(top_region, 0u, (u32)p.K))
                 phase_sk_topk<BLOCK, false>(lds0, (u32)(u64)row, (u32)((u64)row >> 32), seed, n_pu, n_log);
            else phase_sk_topk<BLOCK, true>(lds0, (u32)(u64)row, (u32)((u64)row >> 32), seed, n_pu, n_log);
        }
        GP_SYNC();
        if (uni(ctl->fail)) {
            // the row leaves for the retry list (the general kernel recounts it); nothing of it was written
            if (tid == 0) {
                const u64 i = __hip_atomic_fetch_add(&p.counters[p.retry_counter], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                p.retry_list[i] = (u32)row;
                if (ctl->fail == 1) __hip_atomic_fetch_add(&p.counters[kSkSlabFails], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifndef GP_SK_TIMING
                __hip_atomic_fetch_add(&p.counters[kDiag0 + min(ctl->fail, 7u)], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // why rows left (diag_sub[1..5])
#endif
            }
            if (tid < 8) ctl->st_row[tid] = 0;
        } else if (tid < 8) { ctl->st[tid] += ctl->st_row[tid]; ctl->st_row[tid] = 0; }
        GP_SYNC();
        phase_sk_wipe<BLOCK>(lds0, 1u, 0u, 0u);                                                            // TOP-K used the level tables' bytes; R is per row
        SKT(ctl, 10);
    }
    GP_SYNC();
    if (tid == 0) {
        const Counter dst[zNumStats] = { kPushes, kEdges, kDegLookups, kFilled, kLdsLevels, kFailedRows, kSkCandEdges, kSkSweep2 };
#pragma unroll
        for (int i = 0; i < zNumStats; ++i)
            if (ctl->st[i]) __hip_atomic_fetch_add(&p.counters[dst[i]], ctl->st[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (ctl->max_e) __hip_atomic_fetch_max(&p.counters[kMaxLevelEdges], (u64)ctl->max_e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (ctl->max_log) __hip_atomic_fetch_max(&p.counters[kMaxLogRecords], (u64)ctl->max_log, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifdef GP_SK_TIMING
        for (int i = 0; i < 16; ++i) __hip_atomic_fetch_add(&p.counters[kDiag0 + i], ctl->tacc[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int i = 0; i < 14; ++i) __hip_atomic_fetch_add(&p.counters[kDiagX0 + i], ctl->tacc2[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
    }
}

template <int BLOCK>
__global__ void __launch_bounds__(BLOCK, BLOCK == 1024 ? 4 : 6) gfpush_sk_kernel(const KParams)
{
    gfpush_sk_rows<BLOCK>();
}

}  // namespace gp

#endif  // !GP_DIAG
