// gfpush_kernels.hpp -- hand-written HIP (gfx950 / CDNA4) kernels for GFPush.
//
// One persistent workgroup (8 or 16 wave64; two or one per CU) owns one seed ("row") at a time and pulls
// rows from a device-side queue (the reference's `omp parallel for schedule(dynamic)` over
// seeds, precompute/graph.h:73-74).  Per row it runs the level-synchronous push of
// graph.h:83-110 and the top-K of graph.h:111-126:
//
//   EXPAND  stream the CSR neighbour ranges of the push list, one lane per edge, equal edge counts per wave (edge_stream),
//           and add r/deg into the level's residue table          (graph.h:96-99).
//   SCAN    drain that table; for every node u with residue r:
//             reserve[u] += coef[lvl] * r   -> one (u, coef*r) record in the reserve LOG   (graph.h:90 / :109)
//             deg==0        -> r returns to the seed             (graph.h:91-93)
//             r >= rmax*deg -> append (CSR range, r/deg) to the next push list  (graph.h:94-95)
//             else          -> r is dropped                      (no else branch)
//           "Compact, then process" (scan_level_dense): every wave drains one contiguous range with
//           128-bit LDS accesses, compacts the occupied (key, residue) pairs in place with ballot +
//           mbcnt positions, handles the nodes with full lanes, and compacts the nodes that may push
//           once more before the expensive part (indptr lookup, fp64 division, list allocation).
//   LEVEL 0 is the seed alone: its record and push-list entries are written directly, no table.
//   TOPK    sum the log per node in an LDS table (only nodes that can reach the top-K are
//           tabled), radix-select the K largest (value desc, column asc) and write
//           row/col/value at slot row*K+rank                      (graph.h:111-126).
//
// Residue table of a level: an open-addressing hash table {node -> fp64 residue} in LDS
// (keys int32 + values fp64, 12 B/slot).  A level whose edge count (an upper bound on its
// distinct targets) exceeds 0.75 * slots is expanded in P hash PARTITIONS: pass p re-reads the
// (L2-hot) CSR ranges and keeps only targets with part(v) == p, so the table never leaves LDS
// and no global atomic is issued; a partition that still overflows is split in two in place.
// From kBucketMin partitions on, the level's (target, share) pairs are scattered once into
// fixed-stride hash buckets in HBM and inserted bucket by bucket instead.  Only levels needing
// more than kMaxParts passes use a per-workgroup table in HBM/L2 (scan_level, the slot-walking SCAN).
//
// Table keys are the PACKED column ids of the device CSR: node id in the low bits and
// min(deg(node), deg_sat) in the spare bits above it (pack_degree_kernel, once per graph).
// The push test therefore needs no memory access; only nodes that do push read indptr.
//
// Residues are fp64 end to end, as in the reference (graph.h:76-77,95): the share r/deg is the
// reference's own fp64 quotient and the push test `r >= rmax*deg` (graph.h:94) sees the same
// number whenever a node has a single contribution -- which is what makes exact rational
// ties such as 1/deg(seed) == rmax*deg(u) fall the way the reference decides them.  Sums of
// several contributions use native fp64 atomic adds (ds_add_f64), so their last bits depend
// on arrival order exactly as the reference's depend on its hash-map iteration order.
//
// Reserve map of a row: NOT a table while the row runs.  SCAN appends one (node, coef*r)
// record per frontier node to a per-workgroup LOG (coalesced streaming stores); TOPK sums it.
// This replaced ~8 random 64-B accesses per frontier node by 12 streamed bytes.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gp {

typedef unsigned long long u64;
typedef unsigned int u32;

// Address spaces.  The row's phases (EXPAND, SCAN, TOP-K) are separate, NON-inlined functions so that each gets its own
// register allocation (inlined into one 130 KB function they spilled 47 VGPRs and 180 SGPRs, and an edit in one loop moved
// the allocation of all the others).  Across such a call boundary a plain pointer is generic and every LDS access would
// become a flat_* instruction; so the phases take LDS *byte offsets* and rebuild address_space(3) pointers from them (the
// address-space inference then keeps ds_* instructions), and they read the launch parameters straight from the kernel
// argument segment with scalar loads (address_space(4)) instead of receiving a copy on the stack.
#define GP_LDS  __attribute__((address_space(3)))
#define GP_KARG __attribute__((address_space(4)))
template <class T> __device__ __forceinline__ T* lds_at(u32 byte_off) { return (T*)(GP_LDS T*)(uintptr_t)byte_off; }
__device__ __forceinline__ u32 uni(u32 x) { return (u32)__builtin_amdgcn_readfirstlane((int)x); }     // states wave-uniformity of a value that arrived in a VGPR
__device__ __forceinline__ int uni(int x) { return __builtin_amdgcn_readfirstlane(x); }
__device__ __forceinline__ u64 uni(u64 x) { return ((u64)uni((u32)(x >> 32)) << 32) | uni((u32)x); }
__device__ __forceinline__ long long uni(long long x) { return (long long)uni((u64)x); }
__device__ __forceinline__ double uni(double x) {
    const u64 b = (u64)__double_as_longlong(x);
    return __longlong_as_double((long long)(((u64)uni((u32)(b >> 32)) << 32) | uni((u32)b)));
}

constexpr int    kEmpty      = -1;
constexpr int    kCoefLds    = 40;      // coefficients kept in the control block (more: read from global memory)
constexpr u32    kUnitShift  = 6;       // EXPAND hands out a level's edges in units of 64 (one window; a step = up to 4 units)
constexpr int    kTopkBins   = 4096;    // 12-bit radix digits
constexpr int    kBucketCap  = 256;     // finish the select by ranking once <= this many remain
#ifndef GP_FLAT_W
#define GP_FLAT_W 4
#endif
constexpr int    kFlatW      = GP_FLAT_W;       // EXPAND: 64-edge windows a wave keeps in flight (column loads issued together)
// -DGP_DIAG: thread 0 stamps the phases with the 100 MHz wall clock (cheap).  Unless -DGP_DIAG_LIGHT is given as well, every
// barrier additionally measures what each wave waits there and EXPAND / SCAN stamp their sub-phases with the shader clock --
// detailed, but it inflates the row time by a third, so per-phase shares are best read from the light build.
#if defined(GP_DIAG) && !defined(GP_DIAG_LIGHT)
#define GP_DIAG_HEAVY 1
#endif
#ifdef GP_DIAG
constexpr int    kCtlStruct  = 4096;    // (diagnostic build: + per-barrier-site wait counters)
#elif defined(GP_SK_TIMING)
constexpr int    kCtlStruct  = 2048;    // (sketch kernel with phase stamps, tools/sk_phases.py)
#else
constexpr int    kCtlStruct  = 1280;    // control block at the start of dynamic LDS ...
#endif
constexpr int    kCtlBytes   = kCtlStruct + 16 * 64 * kFlatW;   // ... followed by 64*kFlatW flag bytes per wave (edge_stream)
constexpr int    kThreeLds   = 53248;   // dynamic LDS of a workgroup when three share a CU (160 KB / 3, allocation granularity)
#ifndef GP_TOPK_UF
#define GP_TOPK_UF 12
#endif
#ifndef GP_MIN_CAP
#define GP_MIN_CAP 1024
#endif
#ifndef GP_CAP_MULT
#define GP_CAP_MULT 4
#endif
#ifndef GP_SCAN_VC
#define GP_SCAN_VC 2
#endif
constexpr u32    kMinCap     = GP_MIN_CAP;    // smallest table capacity used for a level
constexpr u32    kMaxParts   = 64;      // most hash partitions a level starts with before it uses the HBM table instead
#ifndef GP_LOAD_NUM
#define GP_LOAD_NUM 3u          // a level is expanded in one pass while its edge count is <= GP_LOAD_NUM / GP_LOAD_DEN of the table's slots
#define GP_LOAD_DEN 4u
#endif
#ifndef GP_BUCKET_MIN
#define GP_BUCKET_MIN 4            // (3 until three workgroups shared a CU: at 52 KB three partition passes beat the buckets, MAG +1.8 %, Reddit +2.8 %)
#endif
constexpr u32    kBucketMin  = GP_BUCKET_MIN;       // levels needing at least this many partitions bucket their edges in HBM once
                                        // instead of re-reading and hash-filtering every CSR range once per partition
constexpr u32    kSkUnitShift = 5;      // sketch kernel: its self-addressed CSR starts every row at a unit of 32 column words (128 bytes)
constexpr u32    kSkMaxPushers = 65536; // ... and numbers the pushers of a row with 16 bits in its log
constexpr u32    kMaxProbe   = 24;      // an LDS insert that probes this many slots reports overflow
constexpr u32    kProbeSpan  = kMaxProbe * (kMaxProbe + 1) / 2;   // furthest a triangular probe sequence can walk (300 slots)
                                        // (recoverable: the level / aggregation is redone in more partitions)

// Push list of a level: one entry per pushing node, in the order of `off`.  The level's edges are numbered 0 .. E-1 in list
// order; entry i covers edges [off_i, off_{i+1}) and edge q of it is column word rel + q of the CSR (rel = CSR start - off,
// modulo 2^32).  No lengths, no chunking of hubs: EXPAND splits the EDGE range evenly over the waves, whatever the entries.
struct PushEntry { u32 rel; u32 off; double share; };    // 16 B
struct ResRec    { int key;   int pad; double val; };    // 16 B  residue table record (HBM)
struct Cand      { u64 bits;  int key; int pad;  };      // 16 B  top-K candidate

// Per-level counters produced by SCAN for the next EXPAND.  Double-buffered by level parity so
// that resetting one set never races with threads still reading the other (no extra barriers).
struct LevelCtr {
    double dangling;      // mass returned to the seed by dangling nodes
    u32 n_dangling;       // how many dangling nodes were drained
    u32 n_rec;            // reserve-log records of the level.  The level loop reads THIS after the level's last barrier, never
                          // ctl->log_count: wave 0 may already be adding the next (one-wave) level's records to that one
    u64 alloc;            // next level's push list: entries so far (low half) and edges so far (high half) -- ONE atomic hands a wave
                          // both its entry indices and its edge offsets, so the list is ordered by `off` however the waves interleave
    u64 pad1;
};

// Control block (lives in LDS, one per workgroup).
struct Ctl {
    long long row;        // row index pulled from the queue
    LevelCtr lc[2];       // what SCAN of level l produces for level l+1 lives in lc[l & 1]
    u32 bcnt[64];         // bucketed levels: edges per bucket, then the scatter cursors
    u32 boff[65];         // bucketed levels: first record of each bucket
    u32 log_count;        // reserve-log records so far
    u32 n_cand;           // top-K candidates (value > 0)
    u32 ovf;              // an LDS table partition overflowed (recoverable: more partitions)
    u32 fail;             // a workspace bound was exceeded (row is reported, never silently wrong)
    u32 n_sel;            // top-K: selected so far
    u32 n_bucket;         // top-K: members of the tie bucket
    u32 tk_bin;           // top-K: digit chosen this pass
    u32 tk_above;         // top-K: count strictly above the chosen digit this pass
    u32 tk_count;         // top-K: count inside the chosen digit
    u32 bovf;             // bucketed levels: a fixed-stride bucket overflowed (recoverable: the partition walk takes the level).
                          // NB field offsets matter: shifting the fields above by 4 bytes cost 3 % (measured); this one
                          // sits in what used to be alignment padding in front of st[]
    u64 st[12];           // statistics of this workgroup (Stat), flushed to p.counters once at the end
    u32 tk_wide;          // top-K: a candidate lies outside [2^-63, 2): the first digit needs the 4096-bin histogram
    u64 st_row[12];       // statistics of the row in flight: added to st[] when the row completes, dropped when it is handed to the retry launch
    double coef[kCoefLds]; // the first coefficients of the recipe: every level starts by reading its own (a global load there is a
                          // dependent round trip on the row's critical path, 11 of them per MAG row)
    // TOP-K threshold known early (phase_tau, after the first level with >= 2K records): every later SCAN notes the keys of
    // the records >= thr_early in the "heavy list" (the idle candidate buffer), so TOP-K claims from a few hundred keys
    // instead of sweeping the whole reserve log once more
    double thr_early;     // +inf: not known (yet) -- no record qualifies
    double tau_early;     // the proven lower bound on the K-th largest total (0: none)
    u32 n_heavy;          // keys noted so far
    u32 heavy_from;       // first reserve-log record covered by the heavy list
    u32 heavy_ovf;        // the list outgrew its buffer: TOP-K sweeps the log as before
    u32 pad_heavy;
    // Distinct targets per edge of each level, as this workgroup's earlier rows had them (x 1.15 margin, in 1/1024; 0: no row yet;
    // quick to rise, slow to fall).  The level's table is planned for edges x this instead of for the edge count: on the citation
    // graphs a level of 6 600 edges reaches 2 100 nodes and fits ONE pass where the edge count asked for three.
    u32 ratio_q[16];
#ifdef GP_DIAG
    u64 lvl_acc[16][6];   // per level: expand ticks, scan ticks, edges, frontier nodes, push entries, table passes (flushed once per workgroup)
    u64 row_acc[4];       // per row: prologue, level 0, level loop outside EXPAND/SCAN, table restore (ticks)
    u64 barw[16];         // per wave: shader cycles spent waiting at workgroup barriers
    u32 barn[16];         // per wave: barriers passed
    u64 exp_c0, exp_c2, exp_pre, exp_post;   // wave 0: cycles from before the EXPAND call to the first instruction of edge_stream / from its last to behind the barrier
    u32 exp_max, exp_pad; u64 exp_sum_max, exp_sum_all;   // per EXPAND call: longest wave / all waves (cycles), summed over calls
    u64 exp_sub[8];       // wave 0: shader cycles in EXPAND (edge_stream): [0] prepare (edge -> entry) [1] wait for the column loads [2] inserts [3] steps [4] batches [5] total [6] calls
    u64 scan_sub[4];      // wave 0: shader cycles in SCAN's (a,b) compaction / (c) records / (d) lookups + push entries / tail
    u64 site_w[64];       // per GP_SYNC() site (in source order): shader cycles all waves waited there
    u32 site_n[64];       // per site: wave arrivals
#endif
};
// Statistics live in LDS, not in registers: nine 64-bit per-thread counters alive for the whole kernel
// cost 18 of the 128 VGPRs (spills).  Phases count in function-local registers and one lane per wave
// adds the wave's totals here when the phase ends.
static_assert(sizeof(Ctl) <= (size_t)kCtlStruct, "the control block must fit its LDS reservation");
enum Stat { sPush = 0, sEdges, sFront, sDeg, sFilled, sSupport, sLds, sGlb, sFailed, sNumStats };
__device__ __forceinline__ void stat_add(Ctl* ctl, int which, u64 n) {           // counts for the row in flight
    __hip_atomic_fetch_add(&ctl->st_row[which], n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void stat_add_final(Ctl* ctl, int which, u64 n) {     // counts that do not belong to a row's own work (failed rows)
    __hip_atomic_fetch_add(&ctl->st[which], n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// Inclusive prefix sum over the wave with DPP row shifts / broadcasts: 6 adds and no LDS round trip
// (__shfl_up / __shfl_down go through ds_bpermute, i.e. six dependent LDS-crossbar latencies).
__device__ __forceinline__ u32 wave_incl_scan_dpp(u32 x) {
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xF, 0xF, false);   // row_shr:1
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xF, 0xF, false);   // row_shr:2
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xF, 0xF, false);   // row_shr:4
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xF, 0xF, false);   // row_shr:8   -> scan inside each row of 16
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xA, 0xF, false);   // row_bcast:15 into rows 1, 3
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xC, 0xF, false);   // row_bcast:31 into rows 2, 3
    return x;
}
// Index of this wave inside its workgroup, as a SCALAR.  threadIdx.x >> 6 is the same in all 64 lanes, but the compiler's
// divergence analysis cannot know that: loops and branches on values derived from it became exec-mask loops with their
// counters in vector registers.  readfirstlane states the uniformity.
__device__ __forceinline__ u32 wave_id() { return (u32)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }
__device__ __forceinline__ u32 wave_sum32(u32 x) {           // sum over the wave, valid in every lane (all 64 lanes must call)
    return (u32)__builtin_amdgcn_readlane((int)wave_incl_scan_dpp(x), 63);
}

enum Counter { kQueue = 0, kQueueRetry, kRetryRows,          // zeroed at every call: row queue heads of the two launches, rows handed to the retry launch
               kQueueRetry2, kRetryRows2,                     // ... and of the third launch of a sketch-kernel call (sketch -> general -> general with bound-sized slabs)
               kSkSlabFails,                                  // ... rows of THIS call that outgrew a sketch-kernel slab (grow_estimate compares it with this call's rows)
               kMaxLevelEdges, kMaxLogRecords,                // observed maxima (atomic max): what the next call's slabs are sized from
               kRetriedTotal,                                 // rows the retry launches have taken since the last reset
               kPushes, kEdges, kFilled, kSupport, kFrontier, kLdsLevels,
               kGlobalLevels, kFailedRows, kDegLookups,
               kSkCandEdges, kSkSweep2,                       // sketch kernel (gfpush_sketch.hpp): edges that reached the exact table, rows whose TOP-K needed a second sweep
               kTicksScan, kTicksExpand, kTicksTopk, kTicksTotal, kTicksScanHbm, kTicksExpandHbm,
               kDiag0, kDiagLast = kDiag0 + 15,   // GP_DIAG: free-form sub-phase slots (see GP_SUB)   // GP_DIAG builds only (100 MHz ticks, summed over workgroups)
               kDiagX0, kDiagXLast = kDiagX0 + 255,   // GP_DIAG: [0] wave cycles, [1] cycles waves spent at barriers, [2] barriers; [16 + 6*lvl + k] per level:
                                                    //   k = 0 expand ticks, 1 scan ticks, 2 edges, 3 frontier nodes, 4 push entries, 5 table passes
               kNumCounters };

// Phase stamps exist only in the diagnostic build (-DGP_DIAG): thread 0 reads the constant
// 100 MHz clock at phase boundaries.  The product build compiles them to nothing.
// Workgroup barrier.  The diagnostic build measures what the waves spend waiting at it (shader cycles).
#ifdef GP_DIAG_HEAVY
constexpr int kSyncBase = __COUNTER__;
#define GP_SYNC() do { constexpr int site_ = (__COUNTER__ - kSyncBase - 1) & 63; const u64 tb_ = clock64(); __syncthreads(); if ((threadIdx.x & 63) == 0) { const u64 w_ = clock64() - tb_; ctl->barw[threadIdx.x >> 6] += w_; ++ctl->barn[threadIdx.x >> 6]; \
    __hip_atomic_fetch_add(&ctl->site_w[site_], w_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); __hip_atomic_fetch_add(&ctl->site_n[site_], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); } } while (0)
#else
#define GP_SYNC() __syncthreads()
#endif
#ifdef GP_DIAG
#define GP_STAMP(var) do { if (threadIdx.x == 0) var = wall_clock64(); } while (0)
#define GP_ACCUM(acc, t0, t1) do { if (threadIdx.x == 0) acc += (t1) - (t0); } while (0)
// sub-phase stamp: adds the time since the previous GP_SUB/GP_SUB_BEGIN to diag slot `i` (thread 0)
#define GP_SUB_BEGIN() do { if (threadIdx.x == 0) gp_sub_t = wall_clock64(); } while (0)
#define GP_SUB(i) do { if (threadIdx.x == 0) { const u64 n_ = wall_clock64(); gp_sub_acc[i] += n_ - gp_sub_t; gp_sub_t = n_; } } while (0)
#define GP_SUB_COUNT(i, n) do { if (threadIdx.x == 0) gp_sub_acc[i] += (n); } while (0)
#else
#define GP_STAMP(var) do { } while (0)
#define GP_ACCUM(acc, t0, t1) do { } while (0)
#define GP_SUB_BEGIN() do { } while (0)
#define GP_SUB(i) do { } while (0)
#define GP_SUB_COUNT(i, n) do { } while (0)
#endif
#ifdef GP_DIAG
#define GP_SUB_PARAMS , u64& gp_sub_t, u64 (&gp_sub_acc)[16]
#define GP_SUB_ARGS , gp_sub_t, gp_sub_acc
#else
#define GP_SUB_PARAMS
#define GP_SUB_ARGS
#endif

struct KParams {
    const int* indptr; const int* indices; int n_nodes;
    int nnz;                              // indices[nnz] is a sentinel word (-1): what lanes past the end of an edge batch load
    // Device `indices` words are PACKED: column id in the low `deg_shift` bits, min(deg(column), deg_sat)
    // above them (sign bit clear).  The degree of every push target thus arrives with its id.
    int deg_shift; u32 node_mask; u32 deg_sat;
    const int* seeds; long long n_seeds;
    const double* coef; int n_coef; double rmax; int K;
    int* out_row; int* out_col; double* out_val; int* out_filled;
    PushEntry* push; u64 push_cap;       // per-workgroup: 2 buffers of push_cap records (this level / next level)
    u32* bt; u64 bt_cap;                  // per-workgroup: 2 boundary tables of bt_cap words: bt[m] = the push-list entry that contains edge 64 m
    ResRec* resg;    u64 resg_cap;       // per-workgroup HBM residue table
    int* log_key; double* log_val; u64 log_cap;   // per-workgroup reserve log
    Cand* cand;      u64 cand_cap;       // per-workgroup top-K candidates
    ResRec* bucket;  u64 bucket_cap;     // per-workgroup (key, share) records of a bucketed level
    u64* counters;
    u32 lds_slots;
    int force_global;
    int prune;                            // 1: threshold-pruned reserve aggregation allowed (all coef >= 0)
    u32 solo;                             // 1: levels of <= 256 edges from <= 64 entries are done by one wave (option "solo_levels")
    u32 rows_distinct;                    // 1: every CSR row holds strictly increasing column ids (checked at gp_graph_create): level 1 needs no table
    int direct;                           // 1: every level's table is indexed by node id (N <= lds_slots; 512-thread kernel only)
    // Two launches per call.  The first gives every workgroup a slab sized from an ESTIMATE of a row's needs; a row that
    // outgrows it is not failed but appended to retry_list.  The second launch (a few workgroups, slabs sized from the
    // rigorous bounds) takes its rows from that list: row_map / n_rows_dev are then set, retry_list is NULL and a
    // row that still does not fit is reported (GP_ERR_OVERFLOW).
    const u32* row_map; const u64* n_rows_dev; u32* retry_list; int queue_counter;
    int retry_counter; int pad_rc;        // the counter that numbers retry_list's entries (kRetryRows / kRetryRows2)
    // sketch kernel (gfpush_sketch.hpp): log2 cells of the level sketch U and of the reserve sketch R (built by TOP-K in the level
    // tables' bytes), slots of the exact table X, the cell rank TOP-K reads its first threshold at, rmax * 2^31 * (1 - 2^-10)
    // rounded down (the push bound in sketch units per unit of packed degree), the reserve-sketch scale 2^31 / max(1, sum of coef)
    u32 sk_lg_mu, sk_lg_mr, sk_cx, sk_target; float sk_thr_f; u32 sk_direct_max; double sk_rscale;     // sk_direct_max: levels of up to this many edges skip the sketch
    // ... which runs on the SELF-ADDRESSED CSR (gfpush.hip:ensure_acsr): every row starts at a 128-byte unit, a column word holds
    // the UNIT NUMBER of its target (and the packed degree above it), so a pushing node's columns are found without indptr.  For
    // that launch `indices` / `nnz` / `deg_shift` / `node_mask` / `deg_sat` describe that copy (nnz = index of its sentinel word).
    const u32* node_pos;                  // node -> first unit of its row (seeds)
    const int* unit_info;                 // first unit of a row -> node id (output columns); second unit of a multi-unit row -> its degree
    unsigned short* log_pu; double* arch; u64 arch_cap;   // per-workgroup: pusher number of every log record; coef * share of every pusher of the row
    u32 sk_hub_units; u32 gk_acsr;        // sk_hub_units 1: deg >= deg_sat implies >= 2 units, the exact degree of a saturated node is unit_info[unit + 1];
                                          // gk_acsr 1 (round 6): the GENERAL kernel runs on the self-addressed copy too (keys are unit numbers; csr_row / out_col)
    int diag_flags;                       // GP_DIAG builds only (instruction attribution by difference): bit 0 = skip TOP-K, bit 1 = run EXPAND twice, bit 2 = walk the drained table once more
};
// The launch parameters where the hardware put them: the kernel argument segment (KParams is the kernels' only argument),
// read with scalar loads.  __builtin_amdgcn_kernarg_segment_ptr() is only meaningful inside the kernel function itself
// (a called function gets a null pointer), but the implicit-argument pointer IS part of the calling convention, and the
// hidden arguments start right behind the explicit ones (8-byte aligned): the segment begins sizeof(KParams) before it.
typedef const GP_KARG KParams& KP;
static_assert(sizeof(KParams) % 8 == 0, "kparams(): the hidden kernel arguments must start right behind KParams");
__device__ __forceinline__ KP kparams() {
    return *(const GP_KARG KParams*)((const GP_KARG char*)__builtin_amdgcn_implicitarg_ptr() - sizeof(KParams));
}

// Where the CSR row of the node behind a key starts and ends (graph.h:43-45, :96-97).  Packed CSR: two indptr words (one 128-byte
// line per pushing node).  Self-addressed copy (round 6, KParams::gk_acsr): the row starts at the key's unit and the degree rides in
// the key -- no memory access unless the degree field is saturated (then one word of unit_info / two of indptr).
__device__ __forceinline__ void csr_row(KP p, u32 key, int& ds, int& de) {
    const u32 id = key & p.node_mask;
    if (p.gk_acsr) {
        u32 deg = key >> p.deg_shift;
        if (deg == p.deg_sat) {
            if (p.sk_hub_units) deg = (u32)p.unit_info[id + 1u];
            else { const int node = p.unit_info[id]; deg = (u32)(p.indptr[node + 1] - p.indptr[node]); }
        }
        ds = (int)(id << kSkUnitShift); de = ds + (int)deg;
    } else { ds = p.indptr[id]; de = p.indptr[id + 1]; }
}

// ---------------------------------------------------------------- small helpers
// Multiplicative (Fibonacci-style) hashes; slot_of() consumes the HIGH bits.
__device__ __forceinline__ u32 hash_a(u32 k) {            // residue / aggregation tables
    k *= 0x9E3779B1u; k ^= k >> 15; k *= 0x85EBCA77u;
    return k;
}
__device__ __forceinline__ u32 hash_b(u32 k) {            // partition choice (independent of hash_a)
    k *= 0x7FEB352Du; k ^= k >> 16; k *= 0x846CA68Bu;
    return k;
}
__device__ __forceinline__ u32 slot_of(u32 h, u32 cap) { return (u32)(((u64)h * cap) >> 32); }
// Home slot in an LDS table of `cap` slots.  Homes lie in [0, cap - kProbeSpan): a probe sequence then
// never leaves [0, cap), so the probing loops need no wrap-around (4 VALU per probe); 2 % of a full
// table is the price.  Every LDS table has cap >= kMinCap > kProbeSpan.
static_assert(kMinCap > 2 * kProbeSpan, "every LDS table must be much larger than the probe span");
__device__ __forceinline__ u32 home_lds(u32 k, u32 cap) { return slot_of(hash_a(k), cap - kProbeSpan); }

// L2-coherent (L1-bypassing) accesses for tables that are also touched by atomics.
template <class T> __device__ __forceinline__ T ld_l2(const T* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <class T> __device__ __forceinline__ void st_l2(T* p, T v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ u32 wave_incl_scan(u32 x, int /*lane*/) { return wave_incl_scan_dpp(x); }
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

// Flag form: each lane has 0 or 1 item.  One ballot, one popcount, one mbcnt pair -- the
// "wavefront ballot / prefix-sum" compaction with no cross-lane data movement at all.
__device__ __forceinline__ u32 lane_prefix(u64 mask) {           // set bits below my lane
    return __builtin_amdgcn_mbcnt_hi((u32)(mask >> 32), __builtin_amdgcn_mbcnt_lo((u32)mask, 0u));
}
__device__ __forceinline__ u32 wave_alloc1(u32* lds_counter, bool flag, int lane) {
    const u64 m = __ballot(flag);
    if (m == 0) return 0;                                         // wave-uniform
    u32 base = 0;
    if (lane == 0)
        base = __hip_atomic_fetch_add(lds_counter, (u32)__popcll(m), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    base = (u32)__builtin_amdgcn_readfirstlane((int)base);
    return base + lane_prefix(m);
}
// U flags per lane (item u of every lane precedes item u+1 of any lane inside the wave's block).
template <int U>
__device__ __forceinline__ void wave_alloc_flags(u32* lds_counter, const bool (&flag)[U], u32 (&idx)[U], int lane) {
    u64 m[U]; u32 total = 0;
#pragma unroll
    for (int u = 0; u < U; ++u) { m[u] = __ballot(flag[u]); total += (u32)__popcll(m[u]); }
#pragma unroll
    for (int u = 0; u < U; ++u) idx[u] = 0;
    if (total == 0) return;                                       // wave-uniform
    u32 base = 0;
    if (lane == 0)
        base = __hip_atomic_fetch_add(lds_counter, total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    base = (u32)__builtin_amdgcn_readfirstlane((int)base);
#pragma unroll
    for (int u = 0; u < U; ++u) { idx[u] = base + lane_prefix(m[u]); base += (u32)__popcll(m[u]); }
}

// ---------------------------------------------------------------- residue tables
// LDS table insert-or-add.  One LDS compare-and-swap per probe does everything: the value it
// returns says whether the slot was EMPTY (now claimed for k), already held k, or belongs to
// another key (walk on, triangular steps).  No pre-read, no nested branches: the loop body is
// a handful of instructions, which matters because this path is instruction-issue bound.
// Keys never revert to EMPTY while inserts are running, so a key ends up in exactly one slot.
// The probing loops in gfx950 assembly.  The structurizer turns the C++ loops below into 17-26 instructions per probe
// (nested exec-mask save/restore blocks, boolean results moved through VGPRs); written by hand a probe is the address,
// the LDS operation, its wait, two v_cmpx that narrow EXEC to the lanes that must walk on, and the scalar step
// bookkeeping: 9-10 instructions.  All three leave `slot` at the last slot a lane looked at and return the key it saw
// there (its own key or kEmpty = done, anything else = the probe limit was reached).
__device__ __forceinline__ u32 lds_addr(const void* p) {          // byte address inside LDS of a pointer known to point there
    return (u32)(uintptr_t)(const __attribute__((address_space(3))) void*)p;
}
__device__ __forceinline__ int probe_cas_asm(int* keys, u32& slot, int k) {
    int seen; u32 ad; u64 sv; u32 st;
    asm volatile(
        "s_mov_b64 %[sv], exec\n\t"
        "s_mov_b32 %[st], 1\n"
        "1:\n\t"
        "v_lshl_add_u32 %[ad], %[sl], 2, %[base]\n\t"
        "ds_cmpst_rtn_b32 %[seen], %[ad], %[emp], %[key]\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_cmpx_ne_u32 vcc, %[seen], %[key]\n\t"
        "v_cmpx_ne_u32 vcc, -1, %[seen]\n\t"
        "s_cbranch_execz 2f\n\t"
        "v_add_u32 %[sl], %[st], %[sl]\n\t"
        "s_add_u32 %[st], %[st], 1\n\t"
        "s_cmp_le_u32 %[st], %[lim]\n\t"
        "s_cbranch_scc1 1b\n"
        "2:\n\t"
        "s_mov_b64 exec, %[sv]"
        : [sl] "+v"(slot), [seen] "=&v"(seen), [ad] "=&v"(ad), [sv] "=&s"(sv), [st] "=&s"(st)
        : [key] "v"(k), [emp] "v"(kEmpty), [base] "s"(lds_addr(keys)), [lim] "n"(kMaxProbe)
        : "vcc", "scc", "memory");
    return seen;
}
__device__ __forceinline__ int probe_find_asm(const int* keys, u32& slot, int k) {
    int seen; u32 ad; u64 sv; u32 st;
    asm volatile(
        "s_mov_b64 %[sv], exec\n\t"
        "s_mov_b32 %[st], 1\n"
        "1:\n\t"
        "v_lshl_add_u32 %[ad], %[sl], 2, %[base]\n\t"
        "ds_read_b32 %[seen], %[ad]\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_cmpx_ne_u32 vcc, %[seen], %[key]\n\t"
        "v_cmpx_ne_u32 vcc, -1, %[seen]\n\t"
        "s_cbranch_execz 2f\n\t"
        "v_add_u32 %[sl], %[st], %[sl]\n\t"
        "s_add_u32 %[st], %[st], 1\n\t"
        "s_cmp_le_u32 %[st], %[lim]\n\t"
        "s_cbranch_scc1 1b\n"
        "2:\n\t"
        "s_mov_b64 exec, %[sv]"
        : [sl] "+v"(slot), [seen] "=&v"(seen), [ad] "=&v"(ad), [sv] "=&s"(sv), [st] "=&s"(st)
        : [key] "v"(k), [base] "s"(lds_addr(keys)), [lim] "n"(kMaxProbe)
        : "vcc", "scc", "memory");
    return seen;
}
__device__ __forceinline__ bool res_add_lds(int* keys, double* vals, u32 cap, int k, double v) {
    u32 slot = home_lds((u32)k, cap);
    const int seen = probe_cas_asm(keys, slot, k);
    if (seen != kEmpty && seen != k) return false;
    __hip_atomic_fetch_add(&vals[slot], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    return true;
}

// Same insert, reporting failure through a sticky LDS flag instead of a return value: the callers that
// insert eight keys per step would otherwise fold eight results into a lane mask (4 SALU each).
__device__ __forceinline__ void res_add_lds_flag(int* keys, double* vals, u32 cap, int k, double v, u32* flag) {
    u32 slot = home_lds((u32)k, cap);
    const int seen = probe_cas_asm(keys, slot, k);
    if (seen != kEmpty && seen != k) *flag = 1u;
    else __hip_atomic_fetch_add(&vals[slot], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// One 64-edge window of EXPAND into an LDS hash table, start to finish in gfx950 assembly: "has an edge" test, partition
// filter (hash_b; skipped when parts == 1), home slot (hash_a), the compare-and-swap probing loop, ds_add_f64 of the share
// for the lanes that found or claimed their slot, the sticky overflow flag for lanes that hit the probe limit.  ~32
// instructions per window at 1.3 probes, 8 of them scalar; the compiler's version of the same C++ (nested exec save /
// restore blocks, booleans moved through VGPRs, a uniform branch per condition) executed ~100, 40 of them scalar, and the
// step loop of EXPAND is where a third of the kernel's instructions are issued.
__device__ __forceinline__ void insert_window_asm(int* keys, double* vals, u32 cap, u32* flag, int col, double sh, u32 parts, u32 part)
{
    u32 t, h, slot, seen, st; u64 sv, ent;
    asm volatile(
        "s_mov_b64 %[sv], exec\n\t"
        "v_cmpx_lt_i32 vcc, -1, %[col]\n\t"                       // lanes that hold an edge
        "s_cmp_eq_u32 %[parts], 1\n\t"
        "s_cbranch_scc1 3f\n\t"
        "v_mul_lo_u32 %[t], %[col], %[cb1]\n\t"                   // hash_b -> partition
        "v_lshrrev_b32 %[h], 16, %[t]\n\t"
        "v_xor_b32 %[t], %[h], %[t]\n\t"
        "v_mul_lo_u32 %[t], %[t], %[cb2]\n\t"
        "v_mul_hi_u32 %[t], %[t], %[parts]\n\t"
        "v_cmpx_eq_u32 vcc, %[part], %[t]\n"
        "3:\n\t"
        "s_cbranch_execz 5f\n\t"
        "s_mov_b64 %[ent], exec\n\t"
        "v_mul_lo_u32 %[h], %[col], %[ca1]\n\t"                   // hash_a -> home slot
        "v_lshrrev_b32 %[t], 15, %[h]\n\t"
        "v_xor_b32 %[h], %[t], %[h]\n\t"
        "v_mul_lo_u32 %[h], %[h], %[ca2]\n\t"
        "v_mul_hi_u32 %[slot], %[h], %[capm]\n\t"
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
        "v_mov_b32 %[t], 1\n\t"                                   // probe limit reached: the lanes still searching give up
        "v_mov_b32 %[h], %[fa]\n\t"
        "ds_write_b32 %[h], %[t]\n"
        "2:\n\t"
        "s_andn2_b64 exec, %[ent], exec\n\t"                      // the lanes that found or claimed their slot
        "v_lshl_add_u32 %[t], %[slot], 3, %[vb]\n\t"
        "ds_add_f64 %[t], %[sh]\n"
        "5:\n\t"
        "s_mov_b64 exec, %[sv]"
        : [t] "=&v"(t), [h] "=&v"(h), [slot] "=&v"(slot), [seen] "=&v"(seen), [sv] "=&s"(sv), [ent] "=&s"(ent), [st] "=&s"(st)
        : [col] "v"(col), [sh] "v"(sh), [emp] "v"(kEmpty), [parts] "s"(parts), [part] "s"(part),
          [ca1] "s"(0x9E3779B1u), [ca2] "s"(0x85EBCA77u), [cb1] "s"(0x7FEB352Du), [cb2] "s"(0x846CA68Bu),
          [capm] "s"(cap - kProbeSpan), [kb] "s"(lds_addr(keys)), [vb] "s"(lds_addr(vals)), [fa] "s"(lds_addr(flag)), [lim] "n"(kMaxProbe)
        : "vcc", "scc", "memory");
}

// FOUR windows inserted together (round 5).  insert_window_asm called four times is four DEPENDENT LDS round trips -- compare-and-swap,
// wait, add, next window -- and the steps of EXPAND / STREAM / FILTER / the TOP-K sweep are chains of exactly those.  Here the four
// windows' home slots are computed side by side, their four compare-and-swaps are issued back to back (the LDS executes a wave's
// operations in order, so two windows that hold the same key see each other: the second finds the first one's claim), and only the
// lanes whose home slot belonged to another key (~ a quarter at half load) walk on, window by window, in the loop of
// insert_window_asm.  One round trip per step instead of four.  (Waits: when window q is resolved, the compare-and-swaps of the
// windows behind it are younger than its own -- lgkmcnt(3 - q) says it has returned, however many adds were issued in between.)
#define GP_IW4_EACH(X) X(0) X(1) X(2) X(3)
#define GP_IW4_MASK(q)   "v_cmp_lt_i32_e64 %[m" #q "], -1, %[c" #q "]\n\t"                       /* lanes that hold an edge */
#define GP_IW4_F1(q)     "v_mul_lo_u32 %[se" #q "], %[c" #q "], %[st]\n\t"                        /* hash_b -> partition */
#define GP_IW4_F2(q)     "v_lshrrev_b32 %[sl" #q "], 16, %[se" #q "]\n\t" "v_xor_b32 %[se" #q "], %[sl" #q "], %[se" #q "]\n\t"
#define GP_IW4_F3(q)     "v_mul_lo_u32 %[se" #q "], %[se" #q "], %[st]\n\t"
#define GP_IW4_F4(q)     "v_mul_hi_u32 %[se" #q "], %[se" #q "], %[parts]\n\t"
#define GP_IW4_F5(q)     "v_cmp_eq_u32_e64 vcc, %[part], %[se" #q "]\n\t" "s_and_b64 %[m" #q "], %[m" #q "], vcc\n\t"
// (round 5) The home slot of these batched inserts comes from FULL-RATE 24-bit multiplies: v_mul_lo_u32 / v_mul_hi_u32 issue at a
// quarter of the rate of an ordinary vector instruction on gfx950, and hash_a + slot_of are three of them per key -- 12 issue
// slots per window where the whole window insert has ~30.  h = mul24(k, c1); h ^= h >> 15; h = mul24(h, c2); slot = mul24(h >> 16,
// capm) >> 16 (capm < 2^16: LDS tables have at most a few thousand slots).  Only the low 24 bits of a key enter: a graph of more
// than 2^24 units probes a little longer, nothing else.  sk_home() in gfpush_sketch.hpp is the same function in C++; the tables of
// the sketch kernel are touched through these two only.
#define GP_IW4_H1(q)     "v_mul_u32_u24 %[sl" #q "], 0x9e3779, %[c" #q "]\n\t"
#define GP_IW4_H2(q)     "v_lshrrev_b32 %[se" #q "], 15, %[sl" #q "]\n\t" "v_xor_b32 %[sl" #q "], %[se" #q "], %[sl" #q "]\n\t"
#define GP_IW4_H3(q)     "v_mul_u32_u24 %[sl" #q "], 0x85ebcb, %[sl" #q "]\n\t" "v_lshrrev_b32 %[sl" #q "], 16, %[sl" #q "]\n\t"
#define GP_IW4_H4(q)     "v_mul_u32_u24 %[sl" #q "], %[capm], %[sl" #q "]\n\t" "v_lshrrev_b32 %[sl" #q "], 16, %[sl" #q "]\n\t" "v_lshl_add_u32 %[se" #q "], %[sl" #q "], 2, %[kb]\n\t"
#define GP_IW4_HASHES GP_IW4_EACH(GP_IW4_H1) GP_IW4_EACH(GP_IW4_H2) GP_IW4_EACH(GP_IW4_H3) GP_IW4_EACH(GP_IW4_H4)
#define GP_IW4_WALK(q) \
        "v_cmpx_ne_u32 vcc, %[se" #q "], %[c" #q "]\n\t" \
        "v_cmpx_ne_u32 vcc, -1, %[se" #q "]\n\t" \
        "s_cbranch_execz 3" #q "f\n\t" \
        "s_mov_b32 %[st], 1\n" \
        "1" #q ":\n\t" \
        "v_add_u32 %[sl" #q "], %[st], %[sl" #q "]\n\t" \
        "s_add_u32 %[st], %[st], 1\n\t" \
        "s_cmp_le_u32 %[st], %[lim]\n\t" \
        "s_cbranch_scc0 2" #q "f\n\t" \
        "v_lshl_add_u32 %[se" #q "], %[sl" #q "], 2, %[kb]\n\t" \
        "ds_cmpst_rtn_b32 %[se" #q "], %[se" #q "], %[emp], %[c" #q "]\n\t" \
        "s_waitcnt lgkmcnt(0)\n\t" \
        "v_cmpx_ne_u32 vcc, %[se" #q "], %[c" #q "]\n\t" \
        "v_cmpx_ne_u32 vcc, -1, %[se" #q "]\n\t" \
        "s_cbranch_execnz 1" #q "b\n\t" \
        "s_branch 3" #q "f\n" \
        "2" #q ":\n\t" \
        "v_mov_b32 %[sl" #q "], 1\n\t"                            /* probe limit reached: the lanes still searching give up (their slot is not used again) */ \
        "v_mov_b32 %[se" #q "], %[fa]\n\t" \
        "ds_write_b32 %[se" #q "], %[sl" #q "]\n" \
        "3" #q ":\n\t"
// (with a partition filter: the four lane masks live in scalar registers)
#define GP_IW4_ISSUE_M(q) \
        "s_mov_b64 exec, %[m" #q "]\n\t" \
        "ds_cmpst_rtn_b32 %[se" #q "], %[se" #q "], %[emp], %[c" #q "]\n\t"
#define GP_IW4_RESOLVE_M(q, cnt) \
        "s_waitcnt lgkmcnt(" #cnt ")\n\t" \
        "s_mov_b64 exec, %[m" #q "]\n\t" \
        GP_IW4_WALK(q) \
        "s_andn2_b64 exec, %[m" #q "], exec\n\t"                  /* the lanes that found or claimed their slot */ \
        "v_lshl_add_u32 %[se" #q "], %[sl" #q "], 3, %[vb]\n\t" \
        "ds_add_f64 %[se" #q "], %[s" #q "]\n\t"
// (without: a window's mask is one compare away -- no scalar registers are held for it)
#define GP_IW4_ISSUE_C(q) \
        "v_cmpx_lt_i32 vcc, -1, %[c" #q "]\n\t" \
        "ds_cmpst_rtn_b32 %[se" #q "], %[se" #q "], %[emp], %[c" #q "]\n\t" \
        "s_mov_b64 exec, %[sv]\n\t"
#define GP_IW4_RESOLVE_C(q, cnt) \
        "s_waitcnt lgkmcnt(" #cnt ")\n\t" \
        "v_cmpx_lt_i32 vcc, -1, %[c" #q "]\n\t" \
        "s_mov_b64 %[ent], exec\n\t" \
        GP_IW4_WALK(q) \
        "s_andn2_b64 exec, %[ent], exec\n\t" \
        "v_lshl_add_u32 %[se" #q "], %[sl" #q "], 3, %[vb]\n\t" \
        "ds_add_f64 %[se" #q "], %[s" #q "]\n\t" \
        "s_mov_b64 exec, %[sv]\n\t"
template <bool PART>
__device__ __forceinline__ void insert_windows4_asm(int* keys, double* vals, u32 cap, u32* flag, const int (&col)[4], const double (&sh)[4],
                                                    u32 parts, u32 part)
{
    u32 sl0, sl1, sl2, sl3, se0, se1, se2, se3, st; u64 sv;
    if (PART) {
        u64 m0, m1, m2, m3;
        asm volatile(
            "s_mov_b64 %[sv], exec\n\t"
            GP_IW4_EACH(GP_IW4_MASK)
            "s_cmp_eq_u32 %[parts], 1\n\t"
            "s_cbranch_scc1 4f\n\t"
            "s_mov_b32 %[st], 0x7feb352d\n\t" GP_IW4_EACH(GP_IW4_F1) GP_IW4_EACH(GP_IW4_F2)
            "s_mov_b32 %[st], 0x846ca68b\n\t" GP_IW4_EACH(GP_IW4_F3) GP_IW4_EACH(GP_IW4_F4) GP_IW4_EACH(GP_IW4_F5)
            "4:\n\t"
            GP_IW4_HASHES
            GP_IW4_EACH(GP_IW4_ISSUE_M)
            GP_IW4_RESOLVE_M(0, 3) GP_IW4_RESOLVE_M(1, 2) GP_IW4_RESOLVE_M(2, 1) GP_IW4_RESOLVE_M(3, 0)
            "s_mov_b64 exec, %[sv]"
            : [sl0] "=&v"(sl0), [sl1] "=&v"(sl1), [sl2] "=&v"(sl2), [sl3] "=&v"(sl3),
              [se0] "=&v"(se0), [se1] "=&v"(se1), [se2] "=&v"(se2), [se3] "=&v"(se3),
              [sv] "=&s"(sv), [m0] "=&s"(m0), [m1] "=&s"(m1), [m2] "=&s"(m2), [m3] "=&s"(m3), [st] "=&s"(st)
            : [c0] "v"(col[0]), [c1] "v"(col[1]), [c2] "v"(col[2]), [c3] "v"(col[3]), [s0] "v"(sh[0]), [s1] "v"(sh[1]), [s2] "v"(sh[2]), [s3] "v"(sh[3]),
              [emp] "v"(kEmpty), [parts] "s"(parts), [part] "s"(part),
              [capm] "s"(cap - kProbeSpan), [kb] "s"(lds_addr(keys)), [vb] "s"(lds_addr(vals)), [fa] "s"(lds_addr(flag)), [lim] "n"(kMaxProbe)
            : "vcc", "scc", "memory");
    } else {
        u64 ent;
        asm volatile(
            "s_mov_b64 %[sv], exec\n\t"
            GP_IW4_HASHES
            GP_IW4_EACH(GP_IW4_ISSUE_C)
            GP_IW4_RESOLVE_C(0, 3) GP_IW4_RESOLVE_C(1, 2) GP_IW4_RESOLVE_C(2, 1) GP_IW4_RESOLVE_C(3, 0)
            : [sl0] "=&v"(sl0), [sl1] "=&v"(sl1), [sl2] "=&v"(sl2), [sl3] "=&v"(sl3),
              [se0] "=&v"(se0), [se1] "=&v"(se1), [se2] "=&v"(se2), [se3] "=&v"(se3),
              [sv] "=&s"(sv), [ent] "=&s"(ent), [st] "=&s"(st)
            : [c0] "v"(col[0]), [c1] "v"(col[1]), [c2] "v"(col[2]), [c3] "v"(col[3]), [s0] "v"(sh[0]), [s1] "v"(sh[1]), [s2] "v"(sh[2]), [s3] "v"(sh[3]),
              [emp] "v"(kEmpty),
              [capm] "s"(cap - kProbeSpan), [kb] "s"(lds_addr(keys)), [vb] "s"(lds_addr(vals)), [fa] "s"(lds_addr(flag)), [lim] "n"(kMaxProbe)
            : "vcc", "scc", "memory");
    }
}


// Two windows together, no partition filter: for the step loops that have no registers to spare (a small level's STREAM holds its
// whole enumeration state beside the insert; two more live registers there are three callee-saved ones saved to scratch per call).
#define GP_IW2_EACH(X) X(0) X(1)
__device__ __forceinline__ void insert_windows2_asm(int* keys, double* vals, u32 cap, u32* flag, int c0, int c1, double s0, double s1)
{
    u32 sl0, sl1, se0, se1, st; u64 sv, ent;
    asm volatile(
        "s_mov_b64 %[sv], exec\n\t"
        GP_IW2_EACH(GP_IW4_H1) GP_IW2_EACH(GP_IW4_H2) GP_IW2_EACH(GP_IW4_H3) GP_IW2_EACH(GP_IW4_H4)
        GP_IW2_EACH(GP_IW4_ISSUE_C)
        GP_IW4_RESOLVE_C(0, 1) GP_IW4_RESOLVE_C(1, 0)
        : [sl0] "=&v"(sl0), [sl1] "=&v"(sl1), [se0] "=&v"(se0), [se1] "=&v"(se1), [sv] "=&s"(sv), [ent] "=&s"(ent), [st] "=&s"(st)
        : [c0] "v"(c0), [c1] "v"(c1), [s0] "v"(s0), [s1] "v"(s1), [emp] "v"(kEmpty),
          [capm] "s"(cap - kProbeSpan), [kb] "s"(lds_addr(keys)), [vb] "s"(lds_addr(vals)), [fa] "s"(lds_addr(flag)), [lim] "n"(kMaxProbe)
        : "vcc", "scc", "memory");
}

// A serial step of a workgroup -- one wave works, the others wait for it at a barrier: that wave goes ahead of the CU's other
// workgroup at instruction issue for as long as this object lives (s_setprio; measured on the MAG line: -0.6 % kernel time).
struct SerialSection {
    __device__ __forceinline__ SerialSection() { __builtin_amdgcn_s_setprio(3); }
    __device__ __forceinline__ ~SerialSection() { __builtin_amdgcn_s_setprio(0); }
};

// The same window insert for the one-wave small-level path: no partition filter, and the lane learns where its key lives
// (`slot`) and whether it CLAIMED that slot (`seen` == kEmpty) -- the claimed slots ARE the level's frontier, so that path
// never walks the table.  `seen` stays 0 in lanes without an edge.
__device__ __forceinline__ void insert_window_solo(int* keys, double* vals, u32 cap, u32* flag, int col, double sh, u32& slot, int& seen)
{
    u32 t, h, st; u64 sv, ent;
    slot = 0; seen = 0;
    asm volatile(
        "s_mov_b64 %[sv], exec\n\t"
        "v_cmpx_lt_i32 vcc, -1, %[col]\n\t"
        "s_cbranch_execz 5f\n\t"
        "s_mov_b64 %[ent], exec\n\t"
        "v_mul_lo_u32 %[h], %[col], %[ca1]\n\t"
        "v_lshrrev_b32 %[t], 15, %[h]\n\t"
        "v_xor_b32 %[h], %[t], %[h]\n\t"
        "v_mul_lo_u32 %[h], %[h], %[ca2]\n\t"
        "v_mul_hi_u32 %[slot], %[h], %[capm]\n\t"
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
          [ca1] "s"(0x9E3779B1u), [ca2] "s"(0x85EBCA77u),
          [capm] "s"(cap - kProbeSpan), [kb] "s"(lds_addr(keys)), [vb] "s"(lds_addr(vals)), [fa] "s"(lds_addr(flag)), [lim] "n"(kMaxProbe)
        : "vcc", "scc", "memory");
}

// Direct-indexed table for graphs with N <= slots (Cora, Citeseer): the slot IS the node id, so an insert
// is one plain store of the packed key (every writer stores the same word) and one ds_add_f64 -- no
// hash, no compare-and-swap, no probing loop -- and SCAN walks N slots instead of a 4x over-provisioned
// hash table.
__device__ __forceinline__ bool res_add_direct(int* keys, double* vals, u32 node_mask, int k, double v) {
    const u32 slot = (u32)k & node_mask;
    keys[slot] = k;
    __hip_atomic_fetch_add(&vals[slot], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    return true;
}

// Claim a slot for k without touching its value (same probe sequence as res_add_lds).
__device__ __forceinline__ bool lds_claim(int* keys, u32 cap, int k) {
    u32 slot = home_lds((u32)k, cap);
    const int seen_a = probe_cas_asm(keys, slot, k);
    return seen_a == kEmpty || seen_a == k;
}
// Read-only lookup (no inserts may run concurrently): slot of k, or -1.
__device__ __forceinline__ int lds_find(const int* keys, u32 cap, int k) {
    u32 slot = home_lds((u32)k, cap);
    return probe_find_asm(keys, slot, k) == k ? (int)slot : -1;
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

// ---------------------------------------------------------------- push-list allocation
// Appends the pushing nodes of one wave-step to the next level's push list.  Every lane of the wave must call (wave-uniform
// control flow: DPP scan); a lane with len == 0 appends nothing.  ONE LDS atomic per call hands out entry indices and edge
// offsets together; an entry that contains a multiple of 64 edges also records itself in the boundary table (read by
// levels with more than 64 entries, see edge_stream).
template <class CTL>
__device__ __forceinline__ void push_alloc(KP p, CTL* ctl, LevelCtr* nx, PushEntry* push, u32* bt_g,
                                           u32 len, u32 start, double share, int lane, u32* bt_l = nullptr, u32 bt_l_cap = 0)
{
    const u64 M = __ballot(len != 0);
    if (M == 0) return;                                                       // wave-uniform: nobody pushes
    const u32 incl = wave_incl_scan_dpp(len);
    const u32 tot = (u32)__builtin_amdgcn_readlane((int)incl, 63);
    u64 base = 0;
    if (lane == 0)
        base = __hip_atomic_fetch_add(&nx->alloc, ((u64)tot << 32) | (u64)(u32)__popcll(M), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    base = uni(base);
    if (len != 0) {
        const u32 idx = (u32)base + lane_prefix(M), off = (u32)(base >> 32) + (incl - len);
        if ((u64)idx < p.push_cap) { PushEntry pe; pe.rel = start - off; pe.off = off; pe.share = share; push[idx] = pe; }
        else ctl->fail = 1;
        for (u32 m = (off + (1u << kUnitShift) - 1u) >> kUnitShift; ((u64)m << kUnitShift) < (u64)off + len; ++m) {   // hubs: one word per 64 edges
            if ((u64)m < p.bt_cap) bt_g[m] = idx; else ctl->fail = 1;
            if (m < bt_l_cap) bt_l[m] = idx;                                      // (sketch kernel: a copy of the table's head in LDS)
        }
    }
}

// ---------------------------------------------------------------- heavy list (see Ctl::thr_early)
struct Heavy { int* keys; u32 cap; double thr; };
__device__ __forceinline__ Heavy heavy_view(KP p, Ctl* ctl) {
    Heavy h;
    h.keys = (int*)(p.cand + (u64)(u32)blockIdx.x * (u32)p.cand_cap);                       // idle until TOP-K turns tables into candidates
    h.cap = (u32)min((u64)0xFFFFFFFFu, 4ull * p.cand_cap);
    h.thr = uni(ctl->thr_early);
    return h;
}
__device__ __forceinline__ void heavy_note(Ctl* ctl, const Heavy& h, int k, double val) {
    if (val >= h.thr) {                                                               // ~3 % of the records
        const u32 hi = __hip_atomic_fetch_add(&ctl->n_heavy, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (hi < h.cap) h.keys[hi] = k; else ctl->heavy_ovf = 1u;
    }
}

// ---------------------------------------------------------------- SCAN (slot-walking form)
// Drains the residue table of one level (or one partition of it).  U slots per thread are
// handled per round so that the indptr loads of all U nodes are in flight together.
// Since scan_level_dense took over the LDS tables this form is only instantiated for the HBM
// table (IN_LDS = false): levels beyond kMaxParts partitions and the force_global test option.
template <int BLOCK, bool IN_LDS, int U>
__device__ __forceinline__ void scan_level(KP p, Ctl* ctl, LevelCtr* nx, int* lkeys, double* lvals,
                                           ResRec* resg, u32 cap, int* log_key, double* log_val,
                                           PushEntry* push, u32* bt_g, double c, bool do_push)
{
    const int tid = threadIdx.x, lane = tid & 63;
    u32 st_push = 0, st_edges = 0, st_front = 0, st_deg = 0;        // this thread, this level
    const Heavy hv = heavy_view(p, ctl);
    const u32 wave_first = wave_id() * 64u;
    for (u32 base = 0; base < cap; base += BLOCK * U) {
        if (base + wave_first >= cap) break;            // wave-uniform: nothing left for this wave
        int k[U]; double r[U]; bool occ[U];
        // (a) drain U slots
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const u32 slot = base + (u32)u * BLOCK + tid;
            k[u] = kEmpty; r[u] = 0.0;
            if (slot < cap) {
                if (IN_LDS) {
                    k[u] = lkeys[slot];
                    if (k[u] != kEmpty) { r[u] = lvals[slot]; lkeys[slot] = kEmpty; lvals[slot] = 0.0; }
                } else {
                    k[u] = ld_l2(&resg[slot].key);
                    r[u] = ld_l2(&resg[slot].val);
                }
            }
            occ[u] = k[u] != kEmpty;
        }
        if (!IN_LDS) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const u32 slot = base + (u32)u * BLOCK + tid;
                if (occ[u]) { st_l2(&resg[slot].key, kEmpty); st_l2(&resg[slot].val, 0.0); }
            }
        }
        // (b) degrees.  The key carries min(deg, deg_sat) of its node (packed by the host side into every
        //     column id), so the push test needs no memory access; only nodes that DO push (they need
        //     their CSR offset) and saturated hubs read the two indptr words.
        int ds[U], de[U]; bool want_deg[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            want_deg[u] = false; ds[u] = 0; de[u] = 0;
            if (occ[u] && do_push) {
                const u32 dq = (u32)k[u] >> p.deg_shift;
                // exact degree known and the test fails => dropped without touching memory   (graph.h:94);
                // a saturated field still says deg >= deg_sat, so r < rmax*deg_sat cannot push either
                if (dq == 0u || r[u] >= p.rmax * (double)dq) {
                    want_deg[u] = true;
                    csr_row(p, (u32)k[u], ds[u], de[u]); ++st_deg;                          // graph.h:43-45
                }
            }
        }
        // (c) reserve log: one (node, coef*r) record per frontier node          graph.h:90 / :109
        u32 li[U];
        wave_alloc_flags<U>(&ctl->log_count, occ, li, lane);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (occ[u]) {
                ++st_front;
                if (li[u] < p.log_cap) { log_key[li[u]] = k[u]; log_val[li[u]] = c * r[u]; heavy_note(ctl, hv, k[u], c * r[u]); }
                else ctl->fail = 1;
            }
        }
        if (!do_push) continue;
        // (d) push decisions + push-list entries
#pragma unroll
        for (int u = 0; u < U; ++u) {
            double share = 0.0; u32 len = 0;
            if (want_deg[u]) {
                const u32 deg = (u32)(de[u] - ds[u]);
                if (deg == 0) {                                                      // graph.h:91-93
                    __hip_atomic_fetch_add(&nx->dangling, r[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_fetch_add(&nx->n_dangling, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                } else if (r[u] >= p.rmax * (double)deg) {                           // graph.h:94
                    ++st_push; st_edges += deg;
                    const double sh = r[u] / (double)deg;                            // graph.h:95
                    if (sh != 0.0) { share = sh; len = deg; }
                }
            }
            push_alloc(p, ctl, nx, push, bt_g, len, (u32)ds[u], share, lane);
        }
    }
    st_push = wave_sum32(st_push); st_edges = wave_sum32(st_edges); st_front = wave_sum32(st_front); st_deg = wave_sum32(st_deg);
    if (lane == 0) {
        if (st_front) { stat_add(ctl, sFront, st_front); __hip_atomic_fetch_add(&nx->n_rec, st_front, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
        if (st_deg) stat_add(ctl, sDeg, st_deg);
        if (st_push) { stat_add(ctl, sPush, st_push); stat_add(ctl, sEdges, st_edges); }
    }
}

// Slots every wave may stage seed-row nodes in (scan_level_dense<BLOCK, true>): an equal, 4-aligned share of the table.
template <int BLOCK> __device__ __forceinline__ u32 seedrow_slice(u32 C) { return (C / (u32)(BLOCK / 64)) & ~3u; }

// ---------------------------------------------------------------- SCAN, LDS tables: compact, then process
// The table of a level is at most ~25-50 % full, and the per-node work (log record, push test,
// degree lookup, push-list entry) is ~10x the work of looking at a slot.  Walking the table with
// that code under an exec mask made SCAN VALU-issue bound at ~25 % lane use.  Here every wave
// takes 256 consecutive slots per round: (a) each lane drains 4 adjacent slots with 128-bit LDS
// accesses, (b) ballot/mbcnt positions compact the occupied (key, residue) pairs into the front
// of the very slots just drained -- they are free, and LDS operations of one wave execute in
// order -- and (c) the nodes are then processed 64 at a time with every lane busy.
// Requires C % 4 == 0 and the invariant that slots in [cap, C) are empty.
// SEEDROW (level 1 of a row whose seed pushed, graphs whose CSR rows hold distinct columns): the level's frontier IS the
// seed's neighbour list, every node with residue 1/deg(seed) (graph.h:96-99 applied to the one entry of level 0) -- no
// table, no EXPAND, no barrier: every wave copies its share of the row (seed_start .. + seed_deg) into its slot range as
// the compacted (key, residue) pairs that stages (a)+(b) would have produced, and continues with (c).
template <int BLOCK, bool SEEDROW = false>
__device__ __forceinline__ void scan_level_dense(KP p, Ctl* ctl, LevelCtr* nx, int* lkeys, double* lvals,
                                                 u32 cap, u32 C, int* log_key, double* log_val,
                                                 PushEntry* push, u32* bt_g, double c, bool do_push,
                                                 u32 seed_start = 0, u32 seed_deg = 0, double seed_share = 0.0)
{
    typedef int    i4 __attribute__((ext_vector_type(4)));
    typedef double d2 __attribute__((ext_vector_type(2)));
    const int tid = threadIdx.x, lane = tid & 63;
    u32 st_push = 0, st_edges = 0, st_deg = 0;                      // this thread, this level
    const Heavy hv = heavy_view(p, ctl);
    // Every wave owns ONE contiguous range of the table (a multiple of 256 slots) per level: it first
    // compacts the whole range, then processes its nodes.  The steps of (c) each end in a wait for
    // their indptr loads (and, vmcnt being shared, for the stores before them), so a level costs a
    // wave ceil(nodes / (64 V)) such waits -- not one or two per 256 slots as when (a)-(c) alternated.
    constexpr u32 kWaves = BLOCK / 64;
    const u32 range = SEEDROW ? seedrow_slice<BLOCK>(C) : ((cap + kWaves * 256u - 1u) / (kWaves * 256u)) * 256u;
    const u32 wb = wave_id() * range;
    u32 tot = 0;
#ifdef GP_DIAG_HEAVY
    u64 ss0 = clock64(), ss1 = 0, ss2 = 0, ss3 = 0;
#endif
    if (SEEDROW) {
        const u32 lo = (u32)(((u64)wave_id() * seed_deg) / kWaves), hi = (u32)(((u64)(wave_id() + 1u) * seed_deg) / kWaves);
        tot = hi - lo;                                              // <= range (the caller checked seed_deg <= kWaves * range)
        for (u32 j = (u32)lane; j < tot; j += 64u) { lkeys[wb + j] = p.indices[seed_start + lo + j]; lvals[wb + j] = seed_share; }
        __atomic_signal_fence(__ATOMIC_SEQ_CST);
    }
    for (u32 sub = wb; !SEEDROW && sub < wb + range && sub < cap; sub += 256u) {
        // (a) drain 4 adjacent slots per lane
        const u32 s0 = sub + 4u * (u32)lane;
        i4 kk = {kEmpty, kEmpty, kEmpty, kEmpty};
        if (s0 < C) kk = *(const i4*)&lkeys[s0];
        const bool o0 = kk.x != kEmpty, o1 = kk.y != kEmpty, o2 = kk.z != kEmpty, o3 = kk.w != kEmpty;
        const u64 m0 = __ballot(o0), m1 = __ballot(o1), m2 = __ballot(o2), m3 = __ballot(o3);
        const u32 c0 = (u32)__popcll(m0), c1 = (u32)__popcll(m1), c2 = (u32)__popcll(m2), c3 = (u32)__popcll(m3);
        if (c0 + c1 + c2 + c3 == 0) continue;                       // wave-uniform
        if (o0 | o1 | o2 | o3) {
            const d2 ra = *(const d2*)&lvals[s0], rb = *(const d2*)&lvals[s0 + 2];
            const i4 ke = {kEmpty, kEmpty, kEmpty, kEmpty};
            const d2 z = {0.0, 0.0};
            *(i4*)&lkeys[s0] = ke; *(d2*)&lvals[s0] = z; *(d2*)&lvals[s0 + 2] = z;
            __atomic_signal_fence(__ATOMIC_SEQ_CST);                // clears stay ahead of the staging stores
            // (b) compact to the front of the range: [wb, wb + tot) lies inside the slots drained so far;
            //     item u of every lane precedes item u+1 of any lane
            const u32 q0 = wb + tot;
            if (o0) { const u32 q = q0 + lane_prefix(m0);                lkeys[q] = kk.x; lvals[q] = ra.x; }
            if (o1) { const u32 q = q0 + c0 + lane_prefix(m1);           lkeys[q] = kk.y; lvals[q] = ra.y; }
            if (o2) { const u32 q = q0 + c0 + c1 + lane_prefix(m2);      lkeys[q] = kk.z; lvals[q] = rb.x; }
            if (o3) { const u32 q = q0 + c0 + c1 + c2 + lane_prefix(m3); lkeys[q] = kk.w; lvals[q] = rb.y; }
        }
        __atomic_signal_fence(__ATOMIC_SEQ_CST);
        tot += c0 + c1 + c2 + c3;
    }
#ifdef GP_DIAG_HEAVY
    ss1 = ss2 = ss3 = clock64();
#endif
    if (tot != 0) {
        u32 lb = 0;
        if (lane == 0) {
            lb = __hip_atomic_fetch_add(&ctl->log_count, tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(&nx->n_rec, tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        lb = (u32)__builtin_amdgcn_readfirstlane((int)lb);
        // (c) every node: reserve record + the cheap part of the push test (the degree rides in the key).
        //     Nodes that may push (~8 % on the power-law shapes) are compacted once more, to the front of
        //     the slots consumed so far, so that the expensive part -- indptr lookup, fp64 division,
        //     push-list allocation -- runs with full lanes in (d) instead of once per 64 nodes at ~5 lanes.
        u32 ncand = 0;
        constexpr int VC = GP_SCAN_VC;
        for (u32 j = 0; j < tot; j += 64 * VC) {
            int k[VC]; double r[VC]; bool cnd[VC];
#pragma unroll
            for (int v = 0; v < VC; ++v) {
                const u32 idx = j + 64u * (u32)v + (u32)lane;
                k[v] = kEmpty; r[v] = 0.0; cnd[v] = false;
                if (idx < tot) {
                    k[v] = lkeys[wb + idx]; r[v] = lvals[wb + idx];
                    lkeys[wb + idx] = kEmpty; lvals[wb + idx] = 0.0;
                    // reserve log: one (node, coef*r) record per frontier node          graph.h:90 / :109
                    const u32 li = lb + idx;
                    if (li < p.log_cap) { log_key[li] = k[v]; log_val[li] = c * r[v]; heavy_note(ctl, hv, k[v], c * r[v]); }
                    else ctl->fail = 1;
                    // exact degree known and the test fails => dropped without touching memory   (graph.h:94);
                    // a saturated field still says deg >= deg_sat, so r < rmax*deg_sat cannot push either
                    const u32 dq = (u32)k[v] >> p.deg_shift;
                    cnd[v] = do_push && (dq == 0u || r[v] >= p.rmax * (double)dq);
                }
            }
            if (!do_push) continue;
            __atomic_signal_fence(__ATOMIC_SEQ_CST);                // reads and clears stay ahead of the list stores
#pragma unroll
            for (int v = 0; v < VC; ++v) {
                const u64 m = __ballot(cnd[v]);
                if (cnd[v]) { const u32 q = wb + ncand + lane_prefix(m); lkeys[q] = k[v]; lvals[q] = r[v]; }
                ncand += (u32)__popcll(m);                          // <= nodes consumed so far: stays inside cleared slots
            }
        }
        __atomic_signal_fence(__ATOMIC_SEQ_CST);
#ifdef GP_DIAG_HEAVY
        ss2 = ss3 = clock64();
#endif
        // (d) the nodes that may push: V x 64 per step, their indptr loads in flight together
#ifndef GP_SCAN_V
#define GP_SCAN_V 2
#endif
        constexpr int V = GP_SCAN_V;
        for (u32 j = 0; j < ncand; j += 64 * V) {
            int k[V]; double r[V]; bool want[V]; int ds[V], de[V];
#pragma unroll
            for (int v = 0; v < V; ++v) {
                const u32 idx = j + 64u * (u32)v + (u32)lane;
                want[v] = idx < ncand; k[v] = kEmpty; r[v] = 0.0; ds[v] = 0; de[v] = 0;
                if (want[v]) {
                    k[v] = lkeys[wb + idx]; r[v] = lvals[wb + idx];
                    lkeys[wb + idx] = kEmpty; lvals[wb + idx] = 0.0;
                    csr_row(p, (u32)k[v], ds[v], de[v]); ++st_deg;                          // graph.h:43-45
                }
            }
#pragma unroll
            for (int v = 0; v < V; ++v) {
                if (__ballot(want[v]) == 0) continue;                                     // wave-uniform
                double share = 0.0; u32 len = 0;
                if (want[v]) {
                    const u32 deg = (u32)(de[v] - ds[v]);
                    if (deg == 0) {                                                       // graph.h:91-93
                        __hip_atomic_fetch_add(&nx->dangling, r[v], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        __hip_atomic_fetch_add(&nx->n_dangling, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    } else if (r[v] >= p.rmax * (double)deg) {                            // graph.h:94
                        ++st_push; st_edges += deg;
                        const double sh = r[v] / (double)deg;                             // graph.h:95
                        if (sh != 0.0) { share = sh; len = deg; }
                    }
                }
                push_alloc(p, ctl, nx, push, bt_g, len, (u32)ds[v], share, lane);
            }
        }
    }
#ifdef GP_DIAG_HEAVY
    if (tot != 0) ss3 = clock64();
#endif
    // edge totals of the next level and the statistics: one LDS atomic per wave (64 same-address
    // atomics per step would serialise)
    if (lane == 0 && tot) stat_add(ctl, sFront, tot);
    if (do_push && tot) {
        st_push = wave_sum32(st_push); st_edges = wave_sum32(st_edges); st_deg = wave_sum32(st_deg);
        if (lane == 0) {
            if (st_deg) stat_add(ctl, sDeg, st_deg);
            if (st_push) { stat_add(ctl, sPush, st_push); stat_add(ctl, sEdges, st_edges); }
        }
    }
#ifdef GP_DIAG_HEAVY
    if (tid == 0) {
        const u64 ss4 = clock64();
        ctl->scan_sub[0] += ss1 - ss0; ctl->scan_sub[1] += ss2 - ss1; ctl->scan_sub[2] += ss3 - ss2; ctl->scan_sub[3] += ss4 - ss3;
    }
#endif
}

// ---------------------------------------------------------------- EXPAND
// Adds `share` into the next residue table for every column id of every push-list entry (graph.h:96-99).
template <bool IN_LDS, bool DIRECT>
__device__ __forceinline__ void res_add_any(int* lkeys, double* lvals, ResRec* resg, u32 cap, u32 node_mask, int v, double share, u32* flag) {
    if (DIRECT) { res_add_direct(lkeys, lvals, node_mask, v, share); return; }
    if (IN_LDS) res_add_lds_flag(lkeys, lvals, cap, v, share, flag);
    else if (!res_add_hbm(resg, cap, v, share)) *flag = 1u;
}

// The edge enumeration of a level: ONE LANE PER EDGE, and every wave gets the same number of EDGES.
//
// The level's E edges are numbered in push-list order (PushEntry).  Wave w takes the 64-edge units [w U/W, (w+1) U/W) of the
// U = ceil(E / 64) units and walks them in steps of up to four units = four 64-lane windows.  Lane j of the wave holds one
// push-list entry; the owner of every edge of a step follows without a search: the entries that start inside the step flag
// their first edge in a per-wave byte array (transposed: byte w of word `lane` = window w, so one ds_read_b32 fetches a lane's
// four flags; LDS operations of one wave execute in order), a ballot turns each window's flags into a mask and the owner of
// lane i's edge is entry first + popcount(flags up to i) -- mbcnt, no search; its (rel, share) arrive by ds_bpermute and the
// column word is rel + q.  Which entries a lane holds:
//   * a level of <= 64 entries (small levels, and levels made of a few hubs): every wave loads the WHOLE list once;
//     `first` = the number of entries that start at or before the step's first edge, minus one (a ballot);
//   * otherwise the boundary table tells where to start: bt[t / 64] = the entry that contains edge t, written by SCAN with
//     the entries; the wave fetches the table words of all its steps with one load, and lane j of a step loads entry
//     bt + j.  The rare step with more than 63 entries starting inside it continues in extra, unpipelined rounds.
// What this replaces (round 2: one batch of <= 64 ENTRIES per wave, hubs cut into 128-column chunks on a second list): waves
// got equal entry counts but unequal edge counts -- the longest wave of an EXPAND call took 1.45 x the mean -- and every batch
// switch exposed the latency of its entry load (half of a wave's EXPAND time).
//
// The walk is a three-stage software pipeline: while the columns of step s are inserted (f), the column loads of step s+1 and
// the entry load of step s+2 are in flight.  Every wait of an iteration sits at its top, so the loads issued behind it may
// be conditional without the compiler's wait counts turning conservative; lanes without an edge load the sentinel word
// indices[nnz] = -1 (all four column loads of a step are unconditional).  f(col[4], share[4], t0) is called once per step;
// col < 0 = no edge; lane i of window w holds edge number t0 + 64 w + i of the level.  OWN: f(col[4], owner[4], t0) instead --
// the push-list entry every edge belongs to, for callers that record WHERE an edge's share is rather than the share itself.
template <int BLOCK, bool OWN = false, class CTL, class F>
__device__ __forceinline__ void edge_stream(KP p, CTL* ctl, const PushEntry* push, const u32* bt,
                                            u32 n_ent, u32 E, bool dry, F f)
{
    constexpr u32 kWaves = BLOCK / 64;
    static_assert(kFlatW == 4, "edge_stream reads the four window flags of a lane as one 32-bit word");
    const u32 lane = threadIdx.x & 63u;
    const u32 wave = wave_id();
    unsigned char* wscr = (unsigned char*)ctl + kCtlStruct + 64 * kFlatW * wave;
    const int* indices = p.indices;
    const u32 sentinel = (u32)p.nnz;
#ifdef GP_DIAG_HEAVY
    if (threadIdx.x == 0) { const u64 c1_ = clock64(); ctl->exp_pre += c1_ - ctl->exp_c0; ctl->exp_c2 = c1_; }
#endif
    const u32 units = (E + (1u << kUnitShift) - 1u) >> kUnitShift;
    // (ranges are dealt from the last wave down, so that a level of a few units lands on wave 0, 1, ...: the waves that were
    //  dispatched first win the issue arbitration against younger waves, and a small level is a latency chain of one wave)
    const u32 slot = kWaves - 1u - wave;
    const u32 u_lo = (slot * units) / kWaves, u_hi = ((slot + 1u) * units) / kWaves;    // (units < 2^26, slot < 16: 32 bits hold the products)
    if (u_lo >= u_hi || n_ent == 0) return;
    const bool small = n_ent <= 64u;                                           // (wave-uniform) the whole list in one wave
#ifdef GP_DIAG_HEAVY
    u64 xs[7] = {0, 0, 0, 0, 0, 0, 1}; u64 x0 = clock64(), xa = x0, xb;
#define GP_XS(i) do { xb = clock64(); xs[i] += xb - xa; xa = xb; } while (0)
#define GP_XS_FLUSH() do { xs[5] = clock64() - x0; if (threadIdx.x == 0) { for (int i_ = 0; i_ < 7; ++i_) ctl->exp_sub[i_] += xs[i_]; ctl->exp_c2 = clock64(); } \
    if ((threadIdx.x & 63) == 0) { __hip_atomic_fetch_max(&ctl->exp_max, (u32)xs[5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); __hip_atomic_fetch_add(&ctl->exp_sum_all, xs[5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); } } while (0)
#else
#define GP_XS(i) do { } while (0)
#define GP_XS_FLUSH() do { } while (0)
#endif
    // boundary-table words of this wave's regular steps (lane j: step j of the current group of 64 steps)
    u32 btv = 0, bt_first = 0;
    auto load_bt = [&](u32 step0) {
        const u32 uj = u_lo + 4u * (step0 + lane);
        btv = uj < u_hi ? bt[uj] : 0u;
        bt_first = step0;
    };
    if (!small) load_bt(0);

    // Pipeline registers.  "Next" = the step whose entries are in flight: lane j holds entry i0n + j (the whole list when
    // `small`), to be matched against the edges [t0n, t1n).
    PushEntry entn; entn.rel = 0; entn.off = 0; entn.share = 0.0;
    u32 i0n = 0, t0n = 0, t1n = 0;
    u32 u_next = u_lo;                       // first unit of the next REGULAR step (a step normally starts at a multiple of 4 units)
    bool have_ent = true;
    // Stage A: issue the entry load of the step that follows.  `end` < t1n: more than 63 entries start inside the step just
    // matched -- its remainder [end, t1n) is the next step, with the entries from i0n + 63 on (lane 63 of a full round only bounds it).
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
    fetch_next(0);                           // t0n = t1n = 0: takes the regular branch

    int nc[4] = {-1, -1, -1, -1}; double ns[4] = {0.0, 0.0, 0.0, 0.0};     // columns in flight and their shares
    u32 no[4] = {0, 0, 0, 0};                                              // ... or the entries they belong to (OWN)
    u32 nt0 = 0;                                                           // ... and the number of their first edge
    bool have_cols = false;
    // One loop: match the next step's edges to its entries (LDS only), take over the columns of the step before (their loads
    // have been in flight since the previous iteration), issue the next step's column loads and the entry load of the step
    // after it, insert.  Every wait of an iteration sits at its top, so the loads issued behind may be conditional.
    do {
        u32 idx[4]; double sh[4]; u32 own[4]; u32 end = 0;
        if (have_ent) {
            // edge -> entry for the edges [t0n, t1n) given the entries from i0n on
            const u32 cnt = min(64u, n_ent - i0n);
            const u32 off = lane < cnt ? entn.off : 0xFFFFFFFFu;
            end = t1n;
            if (cnt == 64u && i0n + 64u < n_ent) {                             // the 64th entry only bounds the round (wave-uniform)
                const u32 o63 = (u32)__builtin_amdgcn_readlane((int)off, 63);
                if (o63 < t1n) end = o63;
            }
            *(u32*)(wscr + 4 * lane) = 0u;
            if (off > t0n && off < end) { const u32 pos = off - t0n; wscr[(pos & 63u) * 4u + (pos >> 6)] = 1; }
            asm volatile("" ::: "memory");    // the word is written by OTHER lanes: without this the compiler forwards this lane's own 0
            const u32 fl = *(const u32*)(wscr + 4 * lane);
            // the entry that contains edge t0n: lane 0's when the boundary table (or a continuation) chose i0n, else the last
            // one that starts at or before t0n
            u32 before = small ? (u32)__popcll(__ballot(off <= t0n)) - 1u : 0u;
            u32 e[4];
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const bool mine = ((fl >> (8 * w)) & 1u) != 0;
                const u64 M = __ballot(mine);
                e[w] = (before + lane_prefix(M) + (mine ? 1u : 0u)) << 2;      // byte address of the owning lane for ds_bpermute
                before += (u32)__popcll(M);
            }
            const u64 sbits = (u64)__double_as_longlong(dry ? 0.0 : entn.share);
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const u32 rel_e = (u32)__builtin_amdgcn_ds_bpermute((int)e[w], (int)entn.rel);
                const u32 lo = (u32)__builtin_amdgcn_ds_bpermute((int)e[w], (int)(u32)sbits);
                const u32 hi = (u32)__builtin_amdgcn_ds_bpermute((int)e[w], (int)(u32)(sbits >> 32));
                const u32 q = t0n + 64u * (u32)w + lane;
                idx[w] = q < end ? rel_e + q : sentinel;                       // graph.h:97
                sh[w] = __longlong_as_double((long long)(((u64)hi << 32) | lo));
                if constexpr (OWN) own[w] = i0n + (e[w] >> 2); else own[w] = 0u;
            }
        }
        GP_XS(0);
        int cc[4]; double cs[4]; u32 co[4];
#pragma unroll
        for (int w = 0; w < 4; ++w) { cc[w] = nc[w]; cs[w] = ns[w]; co[w] = no[w]; }
        const u32 ct0 = nt0;
#ifdef GP_DIAG_HEAVY
        if (cc[0] == 0x7FFFFFF0 && cc[1] == 0x7FFFFFF0 && cc[2] == 0x7FFFFFF0 && cc[3] == 0x7FFFFFF0) xs[6] += 1;      // (uses the loaded values: the wait is charged here)
        GP_XS(1); xs[3] += 1;
#endif
        const bool had_cols = have_cols;
        have_cols = have_ent;
        if (have_ent) {
#pragma unroll
            for (int w = 0; w < 4; ++w) { nc[w] = indices[idx[w]]; if constexpr (OWN) no[w] = own[w]; else ns[w] = sh[w]; }
            nt0 = t0n;
            fetch_next(end);
        }
        if (had_cols) { if constexpr (OWN) f(cc, co, ct0); else f(cc, cs, ct0); }
        GP_XS(2);
    } while (have_cols);
    GP_XS_FLUSH();
}

// MODE 0: LDS hash table, 1: HBM table, 2: direct-indexed LDS table.  With parts > 1 only targets of hash partition
// `part` are kept (the others belong to a later pass over the same list).
template <int BLOCK, bool IN_LDS, bool DIRECT = false>
__device__ __forceinline__ void expand_level(KP p, Ctl* ctl, int* lkeys, double* lvals,
                                             ResRec* resg, u32 cap, const PushEntry* push, const u32* bt,
                                             u32 n_ent, u32 E, u32 part, u32 parts, bool dry = false)
{
    u32* flag = IN_LDS ? &ctl->ovf : &ctl->fail;         // LDS partition overflow is recoverable, an HBM table overflow is not
    edge_stream<BLOCK>(p, ctl, push, bt, n_ent, E, dry, [&](const int (&v)[4], const double (&sh)[4], u32) {
        if (IN_LDS && !DIRECT) {
#pragma unroll
            for (int w = 0; w < 4; ++w) insert_window_asm(lkeys, lvals, cap, flag, v[w], sh[w], parts, part);         // graph.h:98
            return;
        }
#pragma unroll
        for (int w = 0; w < 4; ++w)
            if (v[w] >= 0 && (parts == 1 || slot_of(hash_b((u32)v[w]), parts) == part))
                res_add_any<IN_LDS, DIRECT>(lkeys, lvals, resg, cap, p.node_mask, v[w], sh[w], flag);   // graph.h:98
    });
}

// ---------------------------------------------------------------- bucketed levels
// A level that needs several LDS partitions (Amazon2M-shape at rmax 1e-6: ~100 k edges, ~14 partitions) would re-read and
// hash-filter every CSR range once per partition.  Instead its edges are visited twice: SCATTER (key, share) records into
// fixed-stride per-bucket runs of an HBM buffer (flat_edges: one lane per edge), then one clean insert pass per bucket with
// every lane busy (the level loop of gfpush_rows).

// The row's `filled` word.  With host-resident outputs (gp_gfpush writes the rows straight into pinned host memory) nothing
// orders it against the row's slot stores -- they leave the chip as independent posted writes, and a system-scope release on
// gfx950 is a write-back of the whole L2 (measured: +5 ms per 65 536-row launch) -- so the HOST does not trust the order either:
// it merges a row only once every one of its `filled` slots has changed from the sentinel pattern the slab was left in.
__device__ __forceinline__ void publish_filled(KP p, long long row, u32 need) {
    if (threadIdx.x == 0 && p.out_filled) p.out_filled[row] = (int)need;
}

// ---------------------------------------------------------------- TOP-K
// Candidates are ordered by the 96-bit composite (value bits, ~column): larger composite =
// larger value, ties broken towards the SMALLER column id.  Composites are unique per row.
// The radix select walks the composite in 12-bit digits (8 of them); `depth` digits fixed so
// far are summarised as a (hi, lo) prefix so that all arithmetic stays 64/32-bit.
__device__ __forceinline__ bool cand_better(const Cand& a, const Cand& b) {        // a ranks before b
    return a.bits > b.bits || (a.bits == b.bits && a.key < b.key);
}
struct Pre { u64 hi; u32 lo; };
__device__ __forceinline__ Pre cand_prefix(const Cand& c, int depth) {            // top 12*depth bits
    Pre q;
    if (depth <= 5) { q.hi = depth == 0 ? 0ull : c.bits >> (64 - 12 * depth); q.lo = 0; }
    else            { q.hi = c.bits; q.lo = (~(u32)c.key) >> (96 - 12 * depth); }
    return q;
}
__device__ __forceinline__ u32 cand_digit(const Cand& c, int depth) {             // digit number `depth`
    const u32 nk = ~(u32)c.key;
    if (depth <= 4) return (u32)(c.bits >> (52 - 12 * depth)) & 0xFFFu;
    if (depth == 5) return (((u32)c.bits & 0xFu) << 8) | (nk >> 24);
    if (depth == 6) return (nk >> 12) & 0xFFFu;
    return nk & 0xFFFu;
}
__device__ __forceinline__ Pre pre_push(Pre q, int depth, u32 digit) {            // prefix after fixing digit `depth`
    if (depth <= 4) { q.hi = (q.hi << 12) | digit; }
    else if (depth == 5) { q.hi = (q.hi << 4) | (digit >> 8); q.lo = digit & 0xFFu; }
    else { q.lo = (q.lo << 12) | digit; }
    return q;
}
__device__ __forceinline__ bool pre_eq(Pre a, Pre b) { return a.hi == b.hi && a.lo == b.lo; }
__device__ __forceinline__ bool pre_gt(Pre a, Pre b) { return a.hi > b.hi || (a.hi == b.hi && a.lo > b.lo); }

// One wave finds, in a 4096-bin histogram, the bin holding the `want`-th largest entry.
__device__ __forceinline__ void topk_pick_bin(Ctl* ctl, const u32* hist, u32 want, int lane) {
    // chunk sums: lane owns bins [64*lane, 64*lane+64); skewed reads avoid bank conflicts
    u32 csum = 0;
    for (int j = 0; j < 64; ++j) csum += hist[64 * lane + ((j + lane) & 63)];
    const u32 csuf = wave_suffix_scan(csum, lane);              // bins >= 64*lane
    if ((u32)__shfl(csuf, 0) < want) {                          // fewer than `want` entries in total
        if (lane == 0) { ctl->tk_bin = 0xFFFFFFFFu; ctl->tk_above = 0; ctl->tk_count = 0; }
        return;
    }
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

// G x 4 consecutive reserve-log records per thread with 16-byte loads (one for the keys, two for the values of a
// group of four): the passes over the log are cold streaming reads whose cost is round trips, so what counts is bytes in
// flight per load instruction -- 12 records per thread with 9 loads instead of 8 records with 16.  Thread t of a sweep
// takes records [base + 4 t, base + 4 t + 4) of every 4*BLOCK-record group; records at or past n_log come back as
// (kEmpty, 0).  Requires 16-byte aligned log arrays and a capacity that is a multiple of 4 (host: slab_sizes).
template <int BLOCK, int G>
__device__ __forceinline__ void load_log_records(const int* log_key, const double* log_val, u32 base, u32 n_log, int tid,
                                                 int (&k)[4 * G], double (&v)[4 * G])
{
    typedef int    i4 __attribute__((ext_vector_type(4)));
    typedef double d2 __attribute__((ext_vector_type(2)));
    i4 kk[G]; d2 va[G], vb[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const u32 i0 = base + (u32)g * 4u * BLOCK + 4u * (u32)tid;
        kk[g] = i4{kEmpty, kEmpty, kEmpty, kEmpty}; va[g] = d2{0.0, 0.0}; vb[g] = d2{0.0, 0.0};
        if (i0 < n_log) {
            kk[g] = *(const i4*)&log_key[i0]; va[g] = *(const d2*)&log_val[i0]; vb[g] = *(const d2*)&log_val[i0 + 2];
        }
    }
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const u32 i0 = base + (u32)g * 4u * BLOCK + 4u * (u32)tid;
        const int ks[4] = {kk[g].x, kk[g].y, kk[g].z, kk[g].w};
        const double vs[4] = {va[g].x, va[g].y, vb[g].x, vb[g].y};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool valid = i0 + (u32)j < n_log;
            k[4 * g + j] = valid ? ks[j] : kEmpty;
            v[4 * g + j] = valid ? vs[j] : 0.0;
        }
    }
}

// RESERVE + TOP-K of one row (graph.h:111-126).  `scratch` is the whole LDS table region:
//   sel[K] | agg = { vals f64[CA], keys i32[CA] }  (re-used as { hist[4096] u32, then tie[kBucketCap] } | big[] of Cand)
// 1. The reserve log is summed per node in the LDS table `agg` (one partition of the keys at
//    a time when the log is long); each occupied slot is a node of the reserve map.  Values
//    > 0 (graph.h:121) become candidates: written to `cand` and counted in the histogram of
//    their first radix digit (sign+exponent).
// 2. MSD radix select with 12-bit digits; as soon as the bucket holding the K-th value fits
//    `big`, it is compacted into LDS and the remaining passes never touch HBM again.
template <int BLOCK>
__device__ __forceinline__ void topk_row(KP p, Ctl* ctl, unsigned char* scratch, u32 scratch_bytes,
                                         const int* log_key, const double* log_val, Cand* cand,
                                         long long row, int seed, u32 seg_begin, u32 seg_len, int n_levels,
                                         int /*unused*/ GP_SUB_PARAMS)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = (int)wave_id();
    Cand* sel  = (Cand*)scratch;                                     // [K]
    unsigned char* region = (unsigned char*)(sel + p.K);
    const u32 region_bytes = scratch_bytes - (u32)(region - scratch);
    const u32 CA = region_bytes / 12;
    double* avals = (double*)region;
    int* akeys = (int*)(region + 8 * (size_t)CA);
    // the select runs after the last aggregation table has been turned into candidates, so its digit histogram, the
    // tie bucket and the compacted bucket share the table's bytes: 20 KB more table (1 700 slots) for the pruned aggregation
    u32*  hist = (u32*)region;                                       // [kTopkBins]
    Cand* tie  = (Cand*)region;                                      // [kBucketCap]: filled after the last histogram was read
    static_assert(kBucketCap * sizeof(Cand) <= kTopkBins * sizeof(u32), "the tie bucket lives in the histogram's bytes");
    Cand* big = (Cand*)(region + kTopkBins * sizeof(u32));
    const u32 big_cap = (region_bytes - kTopkBins * (u32)sizeof(u32)) / (u32)sizeof(Cand);
    const u32 n_log = ctl->log_count;
    const u32 K = (u32)p.K;

    // turns the occupied slots of the aggregation table into candidates (+ first-digit histogram)
    // `floor_v`: a proven lower bound on the K-th largest total (tau of the pruned path, else 0).  At least K nodes
    // reach it, so a node below it is not among the K largest and need not become a candidate: the select then
    // works on a few hundred entries instead of every live node (~1.5 k per MAG row).
    double floor_v = 0.0;
    auto emit_candidates = [&](u32& n_nodes) {
        for (u32 base = 0; base < CA; base += BLOCK) {
            const u32 slot = base + tid;
            bool keep = false;
            Cand c; c.bits = 0; c.key = 0; c.pad = 0;
            if (slot < CA) {
                const int k = akeys[slot];
                if (k != kEmpty) {
                    ++n_nodes;                                                   // graph.h:111 res.size()
                    const double v = avals[slot];
                    if (v > 0.0 && v >= floor_v) { c.bits = (u64)__double_as_longlong(v); c.key = (int)((u32)k & p.node_mask); keep = true; }   // graph.h:121
                }
            }
            const u32 ci = wave_alloc1(&ctl->n_cand, keep, lane);
            if (keep) {
                if (ci < p.cand_cap) cand[ci] = c; else ctl->fail = 1;
                // first radix digit = sign+exponent; candidates are > 0 and practically all in [2^-63, 2), so the
                // digit histogram is 64 binade counters in ctl->bcnt (see the threshold pass); anything else
                // raises tk_wide and the select re-counts the first digit in the 4096-bin histogram
                const int e = 1023 - (int)(c.bits >> 52);
                if ((u32)e < 64u) __hip_atomic_fetch_add(&ctl->bcnt[e], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                else ctl->tk_wide = 1u;
            }
        }
    };
    GP_SUB_BEGIN();
    // ---- 0. threshold-pruned aggregation (exact; applies when every coef >= 0)
    // A node's total is >= each of its records, and the records of ONE level belong to distinct
    // nodes, so the K-th largest record of the biggest level is a lower bound tau on the final
    // K-th largest total.  A node all of whose (<= n_levels) records are below tau/n_levels sums
    // to less than tau and cannot be selected.  Pass A claims table slots only for nodes owning a
    // record >= tau/n_levels, pass B adds ALL records of claimed nodes (read-only probe).  tau is
    // taken as the lower edge of the 1/16-binade counter holding that K-th record (one histogram pass).
    bool pruned_done = false;
    u32 live_nodes = 0;
    if (p.prune && seg_len >= 2 * K && n_levels >= 1) {
        // (phase_tau already did all of this for the rows whose threshold level was not their last one)
        const bool early = uni(ctl->tau_early) > 0.0 && !uni(ctl->heavy_ovf);
        double tau = early ? uni(ctl->tau_early) : 0.0;
        if (early) {
            if (tid == 0) { ctl->n_cand = 0; ctl->ovf = 0; }
            GP_SYNC();
        } else {
            // Histogram of the segment's records by binade and the top 4 mantissa bits: 64 x 16 counters at the start
            // of the (idle) table region, indexed so that a larger value has a smaller index.  Records are <= 1, so
            // binade b = 1023 - biased exponent holds [2^-b, 2^-b+1); the last binade also takes everything smaller and
            // then gives no threshold.  tau = the lower edge of the counter holding the K-th largest record: within
            // 6 % of it (the binade counters alone gave up to a factor 2, which claimed ~1.4x the nodes).
            u32* fine = hist;
            for (u32 i = tid; i < 1024u; i += BLOCK) fine[i] = 0;
            if (tid == 0) { ctl->n_cand = 0; ctl->ovf = 0; }
            GP_SYNC();
            for (u32 base = 0; base < seg_len; base += 8 * BLOCK) {              // 8 loads in flight per thread
                double vv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const u32 i = base + (u32)u * BLOCK + tid;
                    vv[u] = i < seg_len ? log_val[seg_begin + i] : 0.0;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (vv[u] > 0.0) {
                        const u64 bits = (u64)__double_as_longlong(vv[u]);
                        const int e = 1023 - (int)(bits >> 52);
                        const u32 m4 = (u32)(bits >> 48) & 15u;
                        const u32 idx = e < 0 ? 15u : e > 63 ? 1023u : (u32)e * 16u + (15u - m4);   // out of range: the lowest counter of the end binade
                        __hip_atomic_fetch_add(&fine[idx], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
            }
            GP_SYNC();
            if (wave == 0) {
                u32 c[16], sum = 0;
#pragma unroll
                for (int j = 0; j < 16; ++j) { c[j] = fine[16 * lane + j]; sum += c[j]; }
                const u32 incl = wave_incl_scan(sum, lane);                      // records >= 2^-lane
                const u64 m = __ballot(incl >= K);
                const u32 need = K - (incl - sum);                               // how many of this binade's records are still wanted
                u32 acc = 0, jsel = 15; bool found = false;
#pragma unroll
                for (int j = 0; j < 16; ++j) { acc += c[j]; if (!found && acc >= need) { jsel = (u32)j; found = true; } }
                if (m == 0) { if (lane == 0) ctl->tk_bin = 0xFFFFFFFFu; }
                else if (lane == __ffsll((long long)m) - 1) ctl->tk_bin = 16u * (u32)lane + jsel;
            }
            GP_SYNC();
            const u32 fsel = ctl->tk_bin;
            GP_SYNC();
            const u32 binade = fsel == 0xFFFFFFFFu ? 63u : fsel >> 4;
            const u32 bin = binade >= 63u ? 0u : 1023u - binade;                 // biased exponent of tau; 0 = no threshold
            if (bin != 0xFFFFFFFFu && bin != 0)
                tau = __longlong_as_double((long long)(((u64)bin << 52) | ((u64)(15u - (fsel & 15u)) << 48)));

        }
        if (tau > 0.0) {
            const double thr = tau / (double)n_levels * 0.99999;
            for (u32 i = tid; i < CA; i += BLOCK) { akeys[i] = kEmpty; avals[i] = 0.0; }
            GP_SYNC();
            GP_SUB(0);
            bool ok = true;
#ifndef GP_TOPK_UA
#define GP_TOPK_UA 12
#endif
            constexpr int UA = GP_TOPK_UA;                                        // records in flight per thread (groups of 4, 16-byte loads)
            // Pass A: claim.  Only ~10 % of the records reach thr, so claiming them where they stand ran eight
            // probe sequences per step at a few active lanes each.  The wave stages the qualifying keys in its
            // 64-word scratch (ballot + mbcnt positions; LDS operations of one wave execute in order) and claims
            // them 64 at a time with every lane busy: ~10x fewer probe sequences.
            int* stage = (int*)(scratch - kCtlBytes + kCtlStruct) + 64 * kFlatW / 4 * wave;
            u32 nst = 0;                                                          // staged keys (wave-uniform)
            auto flush = [&]() {
                const int k = (u32)lane < nst ? stage[lane] : kEmpty;
                if (k != kEmpty) ok &= lds_claim(akeys, CA, k);
                nst = 0;
            };
            // (early: the records from heavy_from on were screened as SCAN wrote them -- their qualifying keys are the heavy list)
            const u32 nA = early ? min(uni(ctl->heavy_from), n_log) : n_log;
            if (early) {
                const int* heavy = (const int*)cand;
                const u32 nh = uni(ctl->n_heavy);
                for (u32 i = tid; i < nh; i += BLOCK) ok &= lds_claim(akeys, CA, heavy[i]);
            }
            for (u32 base = 0; base < nA; base += UA * BLOCK) {
                int kk[UA]; double vv[UA];
                load_log_records<BLOCK, UA / 4>(log_key, log_val, base, nA, tid, kk, vv);   // keys too: one latency, not two
#pragma unroll
                for (int u = 0; u < UA; ++u) {
                    const bool q = vv[u] >= thr;
                    const u64 m = __ballot(q);
                    if (m == 0) continue;                                         // wave-uniform
                    const u32 c = (u32)__popcll(m);
                    if (nst + c > 64u) flush();
                    if (q) stage[nst + lane_prefix(m)] = kk[u];
                    nst += c;
                }
            }
            if (nst) flush();
            if (!ok) ctl->ovf = 1;
            GP_SYNC();
            if (!ctl->ovf) {
                for (u32 base = 0; base < n_log; base += UA * BLOCK) {           // pass B: add
                    int kk[UA]; double vv[UA];
                    load_log_records<BLOCK, UA / 4>(log_key, log_val, base, n_log, tid, kk, vv);
#pragma unroll
                    for (int u = 0; u < UA; ++u) {
                        if (kk[u] == kEmpty) continue;
                        const int slot = lds_find(akeys, CA, kk[u]);
                        if (slot >= 0)
                            __hip_atomic_fetch_add(&avals[slot], vv[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                }
                if (tid < 64) ctl->bcnt[tid] = 0;                        // binade counters of the first digit (emit_candidates)
                if (tid == 0) ctl->tk_wide = 0;
                GP_SYNC();
                GP_SUB(1);
                floor_v = tau * 0.99999;                                  // every claimed node's total is complete here
                emit_candidates(live_nodes);
                GP_SYNC();
                GP_SUB(2); GP_SUB_COUNT(11, 1);
                pruned_done = true;
            } else {
                GP_SYNC();
                if (tid == 0) ctl->ovf = 0;              // too many live nodes for one table: full path
                GP_SYNC();
            }
        }
    }
    // ---- 1. full aggregation of the log -> candidates + first histogram
    u32 support = 0;
    if (!pruned_done) {
    // Key partitions are (p, P) = "keys whose hash falls in the p-th of P equal ranges".  A
    // partition that does not fit the table is split into (2p, 2P) and (2p+1, 2P) -- exactly
    // its two halves under the multiply-high mapping -- and nothing has to be undone, because
    // overflow is detected before the partition's table is turned into candidates.
    u32 P0 = 1;
    // distinct nodes are typically ~0.6 of the records: aim at ~0.7 load
    if ((u64)n_log * 10 > (u64)CA * 6) P0 = (u32)(((u64)n_log * 6 + (u64)CA * 5 - 1) / ((u64)CA * 5));
    if (tid < 64) ctl->bcnt[tid] = 0;                        // binade counters of the first digit (emit_candidates)
    if (tid == 0) ctl->tk_wide = 0;
    if (tid == 0) { ctl->n_cand = 0; ctl->ovf = 0; }
    // Depth-first walk over the partition tree with two registers and no stack: (part, parts)
    // is split into (2*part, 2*parts) on overflow; after a completed RIGHT child (odd part) the
    // parent is complete too, so climb; after a completed left child go to its sibling.
    {
        u32 part = 0, parts = P0;
        for (;;) {
            for (u32 i = tid; i < CA; i += BLOCK) { akeys[i] = kEmpty; avals[i] = 0.0; }
            GP_SYNC();
            GP_SUB(0); GP_SUB_COUNT(8, 1);
            bool ok = true;
            constexpr int UF = GP_TOPK_UF;            // records in flight per thread: 12 with 9 wide loads (was 4 with 8: a pass is cold streaming reads)
            for (u32 base = 0; base < n_log && ok; base += UF * BLOCK) {
                if (ctl->ovf) break;                 // some other thread already gave up on this pass
                int kk[UF]; double vv[UF];
                load_log_records<BLOCK, UF / 4>(log_key, log_val, base, n_log, tid, kk, vv);
#pragma unroll
                for (int u = 0; u < UF; ++u)
                    if (kk[u] != kEmpty && (parts == 1 || slot_of(hash_b((u32)kk[u]), parts) == part))
                        ok &= res_add_lds(akeys, avals, CA, kk[u], vv[u]);
            }
            if (!ok) ctl->ovf = 1;
            GP_SYNC();
            GP_SUB(1);
            if (ctl->ovf) {
                GP_SYNC();
                if (tid == 0) ctl->ovf = 0;
                if (parts < 0x20000000u) { part *= 2; parts *= 2; continue; }     // descend into the left half
                if (tid == 0) ctl->fail = 1;         // cannot happen for a sane hash; reported, not silent
                GP_SYNC();
                break;
            }
            emit_candidates(support);
            GP_SYNC();
            GP_SUB(2);
            while (parts > P0 && (part & 1u)) { part >>= 1; parts >>= 1; }       // right child done => parent done
            ++part;                                                              // sibling / next top-level partition
            if (parts == P0 && part == P0) break;
        }
    }
    }
    {   // graph.h:111 res.size(): nodes this thread saw in the aggregation table(s)
        const u32 n = wave_sum32(pruned_done ? live_nodes : support);
        if (lane == 0 && n) stat_add(ctl, sSupport, n);
    }
    const u32 m = ctl->n_cand;
    const u32 need = m < K ? m : K;                                   // graph.h:113
    if (need == 0 || ctl->fail) {
        // (a row on its way to the retry list publishes nothing: the launch that completes it does)
        if (tid == 0 && p.out_filled && !(ctl->fail && p.retry_list)) p.out_filled[row] = 0;
        return;
    }

    // ---- 2. select
    if (m <= K) {
        for (u32 i = tid; i < m; i += BLOCK) sel[i] = cand[i];
        GP_SYNC();
    } else {
        const Cand* cur = cand;     // current candidate array: HBM, or `big` in LDS after compaction
        u32 cur_n = m;
        bool compacted = false;
        Pre prefix; prefix.hi = 0; prefix.lo = 0;
        u32 want = K;               // how many must still come from the current bucket
        int depth = 0;              // digits fixed so far
        bool take_all = false;
        for (;;) {
            const bool binades = depth == 0 && !ctl->tk_wide;          // first digit counted in 64 binade counters
            if (!binades) {
                for (u32 i = tid; i < kTopkBins; i += BLOCK) hist[i] = 0;
                GP_SYNC();
                for (u32 i = tid; i < cur_n; i += BLOCK) {
                    const Cand c = cur[i];
                    if (pre_eq(cand_prefix(c, depth), prefix))
                        __hip_atomic_fetch_add(&hist[cand_digit(c, depth)], 1u,
                                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                GP_SYNC();
            }
            GP_SUB(3); GP_SUB_COUNT(compacted ? 10 : 9, 1);
            if (wave == 0) {
                if (binades) {                                           // lane = binade: values in [2^-lane, 2^-lane+1)
                    const u32 c = ctl->bcnt[lane];
                    const u32 incl = wave_incl_scan(c, lane);            // candidates >= 2^-lane
                    const u64 mk = __ballot(incl >= want);               // never empty: want <= K < m candidates
                    const int bl = mk ? __ffsll((long long)mk) - 1 : 63;
                    if (lane == bl) { ctl->tk_bin = 1023u - (u32)bl; ctl->tk_above = incl - c; ctl->tk_count = c; }
                } else topk_pick_bin(ctl, hist, want, lane);
            }
            GP_SYNC();
            GP_SUB(4);
            prefix = pre_push(prefix, depth, ctl->tk_bin);
            want -= ctl->tk_above;
            const u32 cnt = ctl->tk_count;
            ++depth;
            GP_SYNC();
            if (cnt == want) { take_all = true; break; }
            if (cnt <= (u32)kBucketCap || depth == 8) break;
            if (!compacted && cnt <= big_cap) {
                // move everything strictly above the prefix to `sel`, the bucket itself to LDS
                for (u32 base = 0; base < cur_n; base += BLOCK) {
                    const u32 i = base + tid;
                    bool is_sel = false, is_b = false;
                    Cand c; c.bits = 0; c.key = 0; c.pad = 0;
                    if (i < cur_n) {
                        c = cur[i];
                        const Pre pre = cand_prefix(c, depth);
                        is_sel = pre_gt(pre, prefix); is_b = pre_eq(pre, prefix);
                    }
                    const u32 si = wave_alloc1(&ctl->n_sel, is_sel, lane);
                    if (is_sel) sel[si] = c;
                    const u32 bi = wave_alloc1(&ctl->n_bucket, is_b, lane);
                    if (is_b) big[bi] = c;
                }
                GP_SYNC();
                cur = big; cur_n = cnt; compacted = true;
                if (tid == 0) ctl->n_bucket = 0;
                GP_SYNC();
                GP_SUB(5);
            }
        }
        // collect: strictly above the prefix -> selected; equal to the prefix -> tie bucket
        for (u32 base = 0; base < cur_n; base += BLOCK) {
            const u32 i = base + tid;
            bool is_sel = false, is_b = false;
            Cand c; c.bits = 0; c.key = 0; c.pad = 0;
            if (i < cur_n) {
                c = cur[i];
                const Pre pre = cand_prefix(c, depth);
                if (pre_gt(pre, prefix) || (take_all && pre_eq(pre, prefix))) is_sel = true;
                else if (pre_eq(pre, prefix)) is_b = true;
            }
            const u32 si = wave_alloc1(&ctl->n_sel, is_sel, lane);
            if (is_sel) sel[si] = c;
            const u32 bi = wave_alloc1(&ctl->n_bucket, is_b, lane);
            if (is_b && bi < (u32)kBucketCap) tie[bi] = c;
        }
        GP_SYNC();
        if (!take_all) {
            const u32 nb = min(ctl->n_bucket, (u32)kBucketCap);
            const u32 n_sel0 = ctl->n_sel;                     // == K - want
            for (u32 i = tid; i < nb; i += BLOCK) {
                const Cand mine = tie[i];
                u32 rank = 0;
                for (u32 j = 0; j < nb; ++j) rank += cand_better(tie[j], mine) ? 1u : 0u;
                if (rank < want) sel[n_sel0 + rank] = mine;
            }
            GP_SYNC();
        }
    }
    GP_SUB(6);
    // order the selected `need` entries (value desc, column asc) and write the row
    const long long out0 = row * (long long)p.K;
    for (u32 i = tid; i < need; i += BLOCK) {
        const Cand c = sel[i];
        u32 rank = 0;
        for (u32 j = 0; j < need; ++j) rank += cand_better(sel[j], c) ? 1u : 0u;
        p.out_row[out0 + rank] = seed;                                           // graph.h:122
        p.out_col[out0 + rank] = p.gk_acsr ? p.unit_info[c.key] : c.key;         // graph.h:123 (unit numbers grow with node ids: the order is the same)
        p.out_val[out0 + rank] = __longlong_as_double((long long)c.bits);        // graph.h:124
    }
    publish_filled(p, row, need);
    if (tid == 0) stat_add(ctl, sFilled, need);
    GP_SUB(7);
}

// ---------------------------------------------------------------- the kernel
// Register budget: 1024 threads = 4 waves/SIMD = 128 VGPRs.  The 768- and 512-thread forms are built for
// 6 waves/SIMD (80 VGPRs) so that two 768-thread or three 512-thread workgroups share a CU and overlap
// one row's barriers and memory stalls with another row's work.
#ifndef GP_MINW_512
#define GP_MINW_512 6          // three 512-thread workgroups per CU (6 waves per SIMD, 80 VGPRs); 4: two with 128 VGPRs
#endif
#ifndef GP_MINW_768
#define GP_MINW_768 6          // two 768-thread workgroups per CU (6 waves per SIMD, 80 VGPRs)
#endif
#ifndef GP_MINW_1024
#define GP_MINW_1024 4         // 8: two 1024-thread workgroups per CU with 64 VGPRs (tools/ab.sh experiments)
#endif
// ---------------------------------------------------------------- phases as separate functions
// What a phase needs to find its workgroup's state again: everything hangs off the kernel arguments, the workgroup index
// and the start of dynamic LDS.
struct WgView {
    Ctl* ctl; double* lvals; int* lkeys; u32 C;
    PushEntry* push2; ResRec* resg; int* log_key; double* log_val; Cand* cand; ResRec* bucket; u32* bt2;
};
__device__ __forceinline__ WgView wg_view(KP p, u32 lds0) {
    WgView w;
    w.C = p.lds_slots;
    w.ctl = lds_at<Ctl>(lds0);
    w.lvals = lds_at<double>(lds0 + (u32)kCtlBytes);
    w.lkeys = lds_at<int>(lds0 + (u32)kCtlBytes + 8u * w.C);
    // (every slab capacity is below 2^32 records -- gfpush.hip:ensure_workspace refuses a workspace bound beyond that --: one
    //  32 x 32 -> 64-bit scalar multiply per slab instead of a 64 x 64-bit one)
    const u32 wg = blockIdx.x;
    w.push2   = p.push + (u64)(2u * wg) * (u32)p.push_cap;
    w.resg    = p.resg + (u64)wg * (u32)p.resg_cap;
    const u64 log_off = (u64)wg * (u32)p.log_cap;
    w.log_key = p.log_key + log_off;
    w.log_val = p.log_val + log_off;
    w.cand    = p.cand + (u64)wg * (u32)p.cand_cap;
    w.bucket  = p.bucket + (u64)wg * (u32)p.bucket_cap;
    w.bt2     = p.bt + (u64)(2u * wg) * (u32)p.bt_cap;
    return w;
}
#define GP_PHASE_NOINLINE static __attribute__((noinline))
#define GP_PHASE_HOT GP_PHASE_NOINLINE

// EXPAND of one level (or one hash partition of it).  MODE 0: LDS hash table, 1: HBM table, 2: direct-indexed LDS table.
template <int BLOCK, int MODE>
__device__ GP_PHASE_HOT void phase_expand(u32 lds0, u32 cap, u32 cur, u32 n_ent, u32 E, u32 part, u32 np,
                                               u32 has_dang, double dang, int seed_key, u32 dry)
{
    KP p = kparams();
    lds0 = uni(lds0); cap = uni(cap); cur = uni(cur); n_ent = uni(n_ent); E = uni(E); part = uni(part); np = uni(np);
    has_dang = uni(has_dang); dang = uni(dang); seed_key = uni(seed_key); dry = uni(dry);
    const WgView w = wg_view(p, lds0);
    const PushEntry* push_cur = w.push2 + (u64)cur * (u32)p.push_cap;
    expand_level<BLOCK, MODE != 1, MODE == 2>(p, w.ctl, w.lkeys, w.lvals, w.resg, cap, push_cur, w.bt2 + (u64)cur * (u32)p.bt_cap,
                                              n_ent, E, part, np, dry != 0);
    if (threadIdx.x == 0 && has_dang && !dry) {                                                    // graph.h:92
        if (MODE == 2) res_add_direct(w.lkeys, w.lvals, p.node_mask, seed_key, dang);
        else if (np == 1 || slot_of(hash_b((u32)seed_key), np) == part) {
            const bool ok = MODE == 0 ? res_add_lds(w.lkeys, w.lvals, cap, seed_key, dang)
                                      : res_add_hbm(w.resg, cap, seed_key, dang);
            if (!ok) { if (MODE == 0) w.ctl->ovf = 1; else w.ctl->fail = 1; }
        }
    }
}

// SCAN of one level's (or partition's) LDS table; what it produces for the next level goes to ctl->lc[nx_sel] and to the
// push buffer nxt_sel.
template <int BLOCK>
__device__ GP_PHASE_HOT void phase_scan_dense(u32 lds0, u32 cap, u32 nx_sel, u32 nxt_sel, double c, u32 do_push)
{
    KP p = kparams();
    lds0 = uni(lds0); cap = uni(cap); nx_sel = uni(nx_sel); nxt_sel = uni(nxt_sel); c = uni(c); do_push = uni(do_push);
    const WgView w = wg_view(p, lds0);
    scan_level_dense<BLOCK>(p, w.ctl, &w.ctl->lc[nx_sel], w.lkeys, w.lvals, cap, w.C, w.log_key, w.log_val,
                            w.push2 + (u64)nxt_sel * (u32)p.push_cap, w.bt2 + (u64)nxt_sel * (u32)p.bt_cap, c, do_push != 0);
}
template <int BLOCK>
__device__ GP_PHASE_NOINLINE void phase_scan_seedrow(u32 lds0, u32 seed_start, u32 seed_deg, double share, double c, u32 do_push)
{
    KP p = kparams();
    lds0 = uni(lds0); seed_start = uni(seed_start); seed_deg = uni(seed_deg); share = uni(share); c = uni(c); do_push = uni(do_push);
    const WgView w = wg_view(p, lds0);
    // level 1: produces the push list of level 2 = buffer 0, counters lc[1]
    scan_level_dense<BLOCK, true>(p, w.ctl, &w.ctl->lc[1], w.lkeys, w.lvals, w.C, w.C, w.log_key, w.log_val,
                                  w.push2, w.bt2, c, do_push != 0, seed_start, seed_deg, share);
}
// A SMALL level -- at most 256 edges from at most 64 push-list entries -- done by ONE wave: EXPAND (one step: the whole push
// list in the wave's lanes, four windows of column loads, inserts into a kMinCap-slot table), then SCAN straight over the
// slots the inserts CLAIMED (no table walk), with nothing but the wave's own program order between them: LDS operations of
// one wave execute in order, so the level needs no workgroup barrier inside.  The other waves skip the call and park at the
// ONE barrier behind it (a level costs the row loop two barriers, two calls per wave and a table walk otherwise; levels 9
// and 10 of a MAG row, the last levels of most recipes).  What it leaves behind is exactly what SCAN leaves: log records,
// ctl->lc[lvl & 1], the next push list.  If an insert hits the probe limit (ctl->ovf), everything is undone and the caller
// runs the level the general way.
#ifndef GP_SOLO_EDGES
#define GP_SOLO_EDGES 256
#endif
constexpr u32 kSoloEdges = GP_SOLO_EDGES, kSoloEntries = 64;
template <int BLOCK>
__device__ GP_PHASE_NOINLINE void phase_solo_level(u32 lds0, u32 lvl, u32 cur, u32 n_ent, u32 E, u32 has_dang, double dang, int seed_key,
                                                   double c, u32 do_push_)
{
    KP p = kparams();
    lds0 = uni(lds0); lvl = uni(lvl); cur = uni(cur); n_ent = uni(n_ent); E = uni(E); has_dang = uni(has_dang); dang = uni(dang);
    seed_key = uni(seed_key); c = uni(c);
    const bool do_push = uni(do_push_) != 0;
    const WgView w = wg_view(p, lds0);
    Ctl* ctl = w.ctl; int* lkeys = w.lkeys; double* lvals = w.lvals;
    const u32 lane = threadIdx.x & 63u;
    const u32 cap = kSoloEdges > 256u ? 2048u : kMinCap;
    const SerialSection ahead;
    unsigned char* wscr = (unsigned char*)ctl + kCtlStruct;                 // wave 0's flag bytes
    u32* list = (u32*)((unsigned char*)ctl + kCtlStruct + 64 * kFlatW);     // the flag areas of waves 1..5 (<= 257 claimed slots): those waves are parked
    LevelCtr* nx = &ctl->lc[lvl & 1u];
    const PushEntry* push_cur = w.push2 + (u64)cur * (u32)p.push_cap;
    PushEntry* push_nxt = w.push2 + (u64)(cur ^ 1u) * (u32)p.push_cap;
    u32* bt_nxt = w.bt2 + (u64)(cur ^ 1u) * (u32)p.bt_cap;
    // ---- EXPAND: one step
    const PushEntry ent = push_cur[min(lane, n_ent - 1u)];
    const u32 off = lane < n_ent ? ent.off : 0xFFFFFFFFu;
    u32 n_list = 0;
    for (u32 t0 = 0; t0 < E; t0 += 256u) {
    const u32 t1 = min(E, t0 + 256u);
    *(u32*)(wscr + 4 * lane) = 0u;
    if (off > t0 && off < t1) { const u32 pos = off - t0; wscr[(pos & 63u) * 4u + (pos >> 6)] = 1; }
    asm volatile("" ::: "memory");            // the word is written by OTHER lanes
    const u32 fl = *(const u32*)(wscr + 4 * lane);
    u32 before = (u32)__popcll(__ballot(off <= t0)) - 1u;
    int col[4]; double sh[4];
    {
        u32 e[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const bool mine = ((fl >> (8 * q)) & 1u) != 0;
            const u64 M = __ballot(mine);
            e[q] = (before + lane_prefix(M) + (mine ? 1u : 0u)) << 2;
            before += (u32)__popcll(M);
        }
        const u64 sbits = (u64)__double_as_longlong(ent.share);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const u32 rel_e = (u32)__builtin_amdgcn_ds_bpermute((int)e[q], (int)ent.rel);
            const u32 lo = (u32)__builtin_amdgcn_ds_bpermute((int)e[q], (int)(u32)sbits);
            const u32 hi = (u32)__builtin_amdgcn_ds_bpermute((int)e[q], (int)(u32)(sbits >> 32));
            const u32 eq = t0 + 64u * (u32)q + lane;
            col[q] = p.indices[eq < t1 ? rel_e + eq : (u32)p.nnz];           // graph.h:97
            sh[q] = __longlong_as_double((long long)(((u64)hi << 32) | lo));
        }
    }
#pragma unroll
    for (int q = 0; q < 5; ++q) {             // the four windows, then the mass dangling nodes returned to the seed (graph.h:92)
        if (q == 4 && (!has_dang || t1 < E)) break;
        const int kq = q < 4 ? col[q] : (lane == 0 ? seed_key : -1);
        const double vq = q < 4 ? sh[q] : dang;
        u32 slot; int seen;
        insert_window_solo(lkeys, lvals, cap, &ctl->ovf, kq, vq, slot, seen);
        const bool fresh = kq >= 0 && seen == kEmpty;                       // this lane claimed the slot: a new frontier node
        const u64 M = __ballot(fresh);
        if (fresh) list[n_list + lane_prefix(M)] = slot;
        n_list += (u32)__popcll(M);
    }
    }
    asm volatile("" ::: "memory");
    if (uni(ctl->ovf)) {                      // (practically never at load <= 0.35) undo: the claimed slots are all there is
        for (u32 j = lane; j < n_list; j += 64u) { const u32 sl = list[j]; lkeys[sl] = kEmpty; lvals[sl] = 0.0; }
        return;
    }
    // ---- SCAN over the claimed slots
    const u32 lb = uni(ctl->log_count);
    const Heavy hv = heavy_view(p, ctl);
    u32 st_push = 0, st_edges = 0, st_deg = 0;
    for (u32 j = 0; j < n_list; j += 64u) {
        const bool valid = j + lane < n_list;
        int k = kEmpty; double r = 0.0;
        if (valid) {
            const u32 sl = list[j + lane];
            k = lkeys[sl]; r = lvals[sl];
            lkeys[sl] = kEmpty; lvals[sl] = 0.0;
            const u32 li = lb + j + lane;                                   // graph.h:90 / :109
            if ((u64)li < p.log_cap) { w.log_key[li] = k; w.log_val[li] = c * r; heavy_note(ctl, hv, k, c * r); } else ctl->fail = 1;
        }
        if (!do_push) continue;
        const u32 dq = (u32)k >> p.deg_shift;
        const bool cand = valid && (dq == 0u || r >= p.rmax * (double)dq);
        double share = 0.0; u32 len = 0, ds = 0;
        if (cand) {
            int ds_, de_;
            csr_row(p, (u32)k, ds_, de_); ++st_deg;                                                   // graph.h:43-45
            ds = (u32)ds_; const u32 deg = (u32)(de_ - ds_);
            if (deg == 0) {                                                                           // graph.h:91-93
                __hip_atomic_fetch_add(&nx->dangling, r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_add(&nx->n_dangling, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            } else if (r >= p.rmax * (double)deg) {                                                   // graph.h:94
                ++st_push; st_edges += deg;
                const double s_ = r / (double)deg;                                                    // graph.h:95
                if (s_ != 0.0) { share = s_; len = deg; }
            }
        }
        push_alloc(p, ctl, nx, push_nxt, bt_nxt, len, ds, share, (int)lane);
    }
    st_push = wave_sum32(st_push); st_edges = wave_sum32(st_edges); st_deg = wave_sum32(st_deg);
    if (lane == 0) {
        ctl->log_count = lb + n_list; nx->n_rec = n_list;
        if (n_list) stat_add(ctl, sFront, n_list);
        if (st_deg) stat_add(ctl, sDeg, st_deg);
        if (st_push) { stat_add(ctl, sPush, st_push); stat_add(ctl, sEdges, st_edges); }
    }
}

template <int BLOCK>
__device__ GP_PHASE_NOINLINE void phase_scan_hbm(u32 lds0, u32 cap, u32 nx_sel, u32 nxt_sel, double c, u32 do_push)
{
    KP p = kparams();
    lds0 = uni(lds0); cap = uni(cap); nx_sel = uni(nx_sel); nxt_sel = uni(nxt_sel); c = uni(c); do_push = uni(do_push);
    const WgView w = wg_view(p, lds0);
    scan_level<BLOCK, false, 4>(p, w.ctl, &w.ctl->lc[nx_sel], w.lkeys, w.lvals, w.resg, cap, w.log_key, w.log_val,
                                w.push2 + (u64)nxt_sel * (u32)p.push_cap, w.bt2 + (u64)nxt_sel * (u32)p.bt_cap, c, do_push != 0);
}

// The empty LDS table (start of the kernel, after TOP-K used the region as scratch, after an overflowing partition).
template <int BLOCK>
__device__ __forceinline__ void wipe_table(int* lkeys, double* lvals, u32 C) {
    for (u32 i = threadIdx.x; i < C; i += BLOCK) { lkeys[i] = kEmpty; lvals[i] = 0.0; }
}

#ifdef GP_DIAG
#define GP_TK_PARAMS , u64* tkd
#define GP_TK_ARGS , tkd
#define GP_TK_ACC(i, a, b) do { if (threadIdx.x == 0) tkd[i] += (b) - (a); } while (0)
#else
#define GP_TK_PARAMS
#define GP_TK_ARGS
#define GP_TK_ACC(i, a, b) do { } while (0)
#endif

// A level that needs several LDS partitions, with its edges bucketed in HBM once: SCATTER of (key, share) records into
// fixed-stride per-bucket runs (one lane per edge, like EXPAND), then one insert pass + SCAN per bucket.  Returns 0 when a
// bucket run overflowed (a hub collected the level's edges): the caller then takes the hash-partition walk.
template <int BLOCK>
__device__ GP_PHASE_NOINLINE u32 phase_bucketed_level(u32 lds0, u32 cap, u32 P, u32 cur, u32 n_ent, u32 E,
                                                      u32 has_dang, double dang, int seed_key, u32 nx_sel, double c, u32 do_push GP_TK_PARAMS)
{
    KP p = kparams();
    lds0 = uni(lds0); cap = uni(cap); P = uni(P); cur = uni(cur); n_ent = uni(n_ent); E = uni(E);
    has_dang = uni(has_dang); dang = uni(dang); seed_key = uni(seed_key); nx_sel = uni(nx_sel); c = uni(c); do_push = uni(do_push);
    const WgView w = wg_view(p, lds0);
    Ctl* ctl = w.ctl;
    int* lkeys = w.lkeys; double* lvals = w.lvals;
    const int tid = threadIdx.x;
    const PushEntry* push_cur = w.push2 + (u64)cur * (u32)p.push_cap;
    // The one-workgroup-per-CU shape (1024 threads x 160 KB: what recipes with rmax < 5e-6 on large graphs run, the Amazon2M line)
    // writes 8-BYTE bucket records (round 5): the packed column word and the NUMBER of the push-list entry the edge belongs to.
    // The pass that inserts a bucket gathers the fp64 share from that entry -- the level's push list (16 bytes per pusher, a few
    // thousand pushers) stays in the L2 -- instead of carrying it through HBM with every edge: (key, pad, share) records of 16
    // bytes, written once and read once, were 7 of that line's 16 MB per row at 4.6 TB/s of fabric traffic (measured: 42.9 ->
    // 39.5 ms).  The smaller shapes keep the 16-byte records: their bucketed levels (Pubmed's peak levels) are short chains of
    // latency, not bandwidth, and a dependent gather in front of the insert cost them 1 %.
    constexpr bool kShortRec = BLOCK == 1024;
    u64* bucket = (u64*)w.bucket; ResRec* bucket16 = w.bucket; (void)bucket; (void)bucket16;
    u64 t0 = 0, t1 = 0, t2 = 0; (void)t0; (void)t1; (void)t2;
    // SCATTER into fixed-stride buckets: hash buckets of one level are nearly equal, and the
    // buffer is sized for the worst level the bounds allow (E_max), typically ~10x this one, so
    // a run of bucket_cap / P records per bucket almost never overflows and the edges are visited
    // twice (scatter, insert) instead of three times (count, scatter, insert).
    const u32 stride = (u32)min((u64)0xFFFFFFFFu, p.bucket_cap / P);
    if (tid < 64) ctl->bcnt[tid] = 0;
    if (tid == 0) ctl->bovf = 0;
    GP_SYNC();
    GP_STAMP(t0);
    // one lane per edge, like EXPAND
    if constexpr (kShortRec) {
    edge_stream<BLOCK, true>(p, ctl, push_cur, w.bt2 + (u64)cur * (u32)p.bt_cap, n_ent, E, false,
                             [&](const int (&v)[4], const u32 (&own)[4], u32) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (v[q] < 0) continue;
            const u32 bk = slot_of(hash_b((u32)v[q]), P);
            const u32 i = __hip_atomic_fetch_add(&ctl->bcnt[bk], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (i < stride) bucket[(u64)bk * stride + i] = ((u64)own[q] << 32) | (u32)v[q];
            else ctl->bovf = 1;
        }
    });
    } else {
    edge_stream<BLOCK>(p, ctl, push_cur, w.bt2 + (u64)cur * (u32)p.bt_cap, n_ent, E, false,
                       [&](const int (&v)[4], const double (&sh)[4], u32) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (v[q] < 0) continue;
            const u32 bk = slot_of(hash_b((u32)v[q]), P);
            const u32 i = __hip_atomic_fetch_add(&ctl->bcnt[bk], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (i < stride) { ResRec r; r.key = v[q]; r.pad = 0; r.val = sh[q]; bucket16[(u64)bk * stride + i] = r; }
            else ctl->bovf = 1;
        }
    });
    }
    GP_SYNC();
    GP_STAMP(t1); GP_TK_ACC(0, t0, t1);
    if (ctl->bovf) return 0u;
    if (tid < 64 && (u32)tid < P) ctl->boff[tid] = (u32)tid * stride;
    GP_SYNC();
    // one insert pass per bucket, refined in place (q of Q sub-partitions) if it still overflows
    for (u32 b = 0; b < P && !ctl->fail; ++b) {
        const u32 lo = ctl->boff[b], hi = lo + ctl->bcnt[b];
        u32 q = 0, Q = 1;
        for (;;) {
            GP_STAMP(t0);
            GP_TK_ACC(2, 0, 1);
            const u32 want = b * Q + q, fine = P * Q;
            bool ok = true;
            if constexpr (kShortRec) {
            // (the records of the NEXT four windows are in flight while the shares of these are gathered and inserted; a third
            //  stage -- the gathers issued one step ahead as well -- measured 40.7 against 39.5 ms)
            u64 nx[4];
            auto load4 = [&](u32 base) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const u32 i = base + (u32)u * BLOCK + tid;
                    nx[u] = 0xFFFFFFFFull;                                                    // key -1 = kEmpty: no record
                    if (i < hi) nx[u] = bucket[i];
                }
            };
            if (lo < hi) load4(lo);
            for (u32 base = lo; base < hi; base += 4 * BLOCK) {
                u64 rr[4]; double sv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) rr[u] = nx[u];
                if (base + 4 * BLOCK < hi) load4(base + 4 * BLOCK);
#pragma unroll
                for (int u = 0; u < 4; ++u)                                                   // four gathers in flight
                    sv[u] = (int)(u32)rr[u] != kEmpty ? push_cur[(u32)(rr[u] >> 32)].share : 0.0;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int key = (int)(u32)rr[u];
                    if (key != kEmpty && (Q == 1 || slot_of(hash_b((u32)key), fine) == want))
                        ok &= res_add_lds(lkeys, lvals, cap, key, sv[u]);                     // graph.h:98
                }
            }
            } else {
            for (u32 base = lo; base < hi; base += 4 * BLOCK) {
                ResRec rr[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const u32 i = base + (u32)u * BLOCK + tid;
                    rr[u].key = kEmpty; rr[u].val = 0.0;
                    if (i < hi) rr[u] = bucket16[i];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (rr[u].key != kEmpty && (Q == 1 || slot_of(hash_b((u32)rr[u].key), fine) == want))
                        ok &= res_add_lds(lkeys, lvals, cap, rr[u].key, rr[u].val);          // graph.h:98
            }
            }
            if (tid == 0 && has_dang && slot_of(hash_b((u32)seed_key), fine) == want)       // graph.h:92
                ok &= res_add_lds(lkeys, lvals, cap, seed_key, dang);
            if (!ok) ctl->ovf = 1;
            GP_SYNC();
            GP_STAMP(t1); GP_TK_ACC(0, t0, t1);
            if (ctl->ovf) {
                wipe_table<BLOCK>(lkeys, lvals, w.C);
                GP_SYNC();
                if (tid == 0) ctl->ovf = 0;
                if (Q < (1u << 20)) { q *= 2; Q *= 2; GP_SYNC(); continue; }
                if (tid == 0) ctl->fail = 1;
                GP_SYNC();
                break;
            }
            phase_scan_dense<BLOCK>(lds0, cap, nx_sel, cur ^ 1u, c, do_push);
            GP_SYNC();
            GP_STAMP(t2); GP_TK_ACC(1, t1, t2);
            if (ctl->fail) break;
            while (Q > 1 && (q & 1u)) { q >>= 1; Q >>= 1; }
            ++q;
            if (Q == 1) break;
        }
    }
    return 1u;
}

// The TOP-K threshold, as soon as a level holds >= 2K reserve records (its K-th largest record bounds the K-th largest
// total from below, see topk_row step 0).  Histogram by binade x top 4 mantissa bits in the 1 024 words of the EXPAND
// flag areas (idle between a level's SCAN and the next EXPAND); every thread of the workgroup calls this.
template <int BLOCK>
__device__ GP_PHASE_NOINLINE void phase_tau(u32 lds0, u32 seg_begin, u32 seg_len, u32 n_levels_max)
{
    KP p = kparams();
    lds0 = uni(lds0); seg_begin = uni(seg_begin); seg_len = uni(seg_len); n_levels_max = uni(n_levels_max);
    const WgView w = wg_view(p, lds0);
    Ctl* ctl = w.ctl;
    const int tid = threadIdx.x, lane = tid & 63;
    static_assert(16 * 64 * kFlatW >= 1024 * 4, "the fine histogram lives in the EXPAND flag areas");
    u32* fine = (u32*)((unsigned char*)ctl + kCtlStruct);
    const u32 K = (u32)p.K;
    for (u32 i = tid; i < 1024u; i += BLOCK) fine[i] = 0;
    GP_SYNC();
    for (u32 base = 0; base < seg_len; base += 4 * BLOCK) {
        double vv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const u32 i = base + (u32)u * BLOCK + tid;
            vv[u] = i < seg_len ? w.log_val[seg_begin + i] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (vv[u] > 0.0) {
                const u64 bits = (u64)__double_as_longlong(vv[u]);
                const int e = 1023 - (int)(bits >> 52);
                const u32 m4 = (u32)(bits >> 48) & 15u;
                const u32 idx = e < 0 ? 15u : e > 63 ? 1023u : (u32)e * 16u + (15u - m4);
                __hip_atomic_fetch_add(&fine[idx], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
    }
    GP_SYNC();
    if (wave_id() == 0) {
        u32 c[16], sum = 0;
#pragma unroll
        for (int j = 0; j < 16; ++j) { c[j] = fine[16 * lane + j]; sum += c[j]; }
        const u32 incl = wave_incl_scan(sum, lane);                      // records >= 2^-lane
        const u64 m = __ballot(incl >= K);
        const u32 need = K - (incl - sum);
        u32 acc = 0, jsel = 15; bool found = false;
#pragma unroll
        for (int j = 0; j < 16; ++j) { acc += c[j]; if (!found && acc >= need) { jsel = (u32)j; found = true; } }
        const int bl = m ? __ffsll((long long)m) - 1 : 63;
        if (lane == bl) {
            ctl->heavy_from = seg_begin + seg_len; ctl->n_heavy = 0;   // (the level loop calls this for the level that has just ended)
            if (m != 0 && bl < 63) {
                const double tau = __longlong_as_double((long long)(((u64)(1023u - (u32)bl) << 52) | ((u64)(15u - jsel) << 48)));
                ctl->tau_early = tau;
                ctl->thr_early = tau / (double)n_levels_max * 0.99999;   // <= tau / (levels the row ends up with)
            }
        }
    }
    GP_SYNC();
}

template <int BLOCK>
__device__ GP_PHASE_NOINLINE void phase_topk(u32 lds0, u32 row_lo, u32 row_hi, int seed, u32 seg_begin, u32 seg_len, int n_levels GP_SUB_PARAMS)
{
    KP p = kparams();
    lds0 = uni(lds0); row_lo = uni(row_lo); row_hi = uni(row_hi); seed = uni(seed); seg_begin = uni(seg_begin); seg_len = uni(seg_len);
    n_levels = uni(n_levels);
    const WgView w = wg_view(p, lds0);
    topk_row<BLOCK>(p, w.ctl, (unsigned char*)w.lvals, 12u * w.C, w.log_key, w.log_val, w.cand,
                    (long long)(((u64)row_hi << 32) | row_lo), seed, seg_begin, seg_len, n_levels, 0 GP_SUB_ARGS);
}

// direct-indexed level tables are instantiated for the two-workgroups-per-CU shapes only
template <int BLOCK> constexpr bool kDirectOk = BLOCK == 512 || BLOCK == 768;

template <int BLOCK>
__device__ __forceinline__ void gfpush_rows()
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    KP p = kparams();
    const u32 lds0 = uni(lds_addr(smem));
    const WgView w = wg_view(p, lds0);
    Ctl* ctl = w.ctl;
    double* lvals = w.lvals;
    int* lkeys = w.lkeys;
    const int tid = threadIdx.x;
    const u32 C = w.C;
    PushEntry* push2 = w.push2;
    ResRec* resg     = w.resg;
    int* log_key     = w.log_key;
    double* log_val  = w.log_val;

    wipe_table<BLOCK>(lkeys, lvals, C);
    if (tid < (int)(sizeof(ctl->st) / sizeof(ctl->st[0]))) { ctl->st[tid] = 0; ctl->st_row[tid] = 0; }      // visible after the first row's barriers
    if (tid < kCoefLds && tid < p.n_coef) ctl->coef[tid] = p.coef[tid];
    if (tid < 16) ctl->ratio_q[tid] = 0;
    const long long n_rows = p.n_rows_dev ? (long long)*p.n_rows_dev : p.n_seeds;    // rows in this launch's queue
    u32 max_e = 0, max_log = 0;                                                      // observed maxima of this workgroup
    u64 tk_scan = 0, tk_expand = 0, tk_topk = 0, tk_total = 0, t0 = 0, t1 = 0, t2 = 0, tk_begin = 0;
    u64 tk_scan_hbm = 0, tk_expand_hbm = 0; (void)tk_scan_hbm; (void)tk_expand_hbm;
    (void)tk_scan; (void)tk_expand; (void)tk_topk; (void)tk_total; (void)t0; (void)t1; (void)t2; (void)tk_begin;
    GP_STAMP(tk_begin);
#ifdef GP_DIAG
    u64 gp_sub_t = 0; u64 gp_sub_acc[16];
    for (int i = 0; i < 16; ++i) gp_sub_acc[i] = 0;
    if (tid < 16) { ctl->barw[tid] = 0; ctl->barn[tid] = 0; }
    if (tid < 4) { ctl->scan_sub[tid] = 0; ctl->row_acc[tid] = 0; }
    if (tid < 96) ctl->lvl_acc[tid / 6][tid % 6] = 0;
    if (tid < 8) ctl->exp_sub[tid] = 0;
    if (tid == 0) { ctl->exp_max = 0; ctl->exp_sum_max = 0; ctl->exp_sum_all = 0; ctl->exp_pre = 0; ctl->exp_post = 0; ctl->exp_c0 = 0; ctl->exp_c2 = 0; }
    if (tid < 64) { ctl->site_w[tid] = 0; ctl->site_n[tid] = 0; }
    const u64 wave_t0 = clock64();
#endif
    const int L = p.n_coef - 1;
    // A one-wave level must not be able to hit a workspace bound: wave 0 runs ahead of the other waves there, and a failure
    // flag raised by it could be seen by a wave that is still finishing the level before (it would leave the level loop alone).
    // Its records (<= kSoloEdges + 1) and entries are checked against the capacities up front, and the boundary table must hold
    // the largest level any frontier can produce (degrees pushed in one level sum to <= 1/rmax).
    const double solo_e_bound = p.rmax > 0.0 ? fmin((double)p.nnz, 1.001 / p.rmax + 16.0) : (double)p.nnz;
    const bool solo_caps = p.push_cap >= (u64)kSoloEdges + 4u && (double)p.bt_cap >= solo_e_bound / (double)(1u << kUnitShift) + 4.0;

    for (;;) {
#ifdef GP_DIAG
        u64 rs0 = 0, rs1 = 0, rs2 = 0, rs3 = 0;
        GP_STAMP(rs0);
#endif
        GP_SYNC();
        if (tid == 0) {
            ctl->row = (long long)__hip_atomic_fetch_add(&p.counters[p.queue_counter], 1ull, __ATOMIC_RELAXED,
                                                         __HIP_MEMORY_SCOPE_AGENT);
            ctl->log_count = 1;                         // record 0 is the seed's own (level 0, written below)
            ctl->n_cand = 0; ctl->fail = 0; ctl->ovf = 0;
            ctl->n_sel = 0; ctl->n_bucket = 0;
            ctl->thr_early = __builtin_inf(); ctl->tau_early = 0.0; ctl->n_heavy = 0; ctl->heavy_from = 0; ctl->heavy_ovf = 0;
        }
        GP_SYNC();
        const long long qpos = uni(ctl->row);
        if (qpos >= n_rows) break;
        const long long row = p.row_map ? (long long)uni(p.row_map[qpos]) : qpos;
        const int seed = uni(p.seeds[row]);
        if (seed < 0 || seed >= p.n_nodes) {            // device API does not pre-validate seeds
            if (tid == 0) { stat_add_final(ctl, sFailed, 1); if (p.out_filled) p.out_filled[row] = 0; }
            continue;
        }
        // A row that ran out of workspace: first launch -> the retry list (its counts are dropped, the retry
        // launch recounts the row); retry launch -> reported.  Returns with the row's statistics cleared.
        auto give_up = [&]() {
            if (tid == 0) {
                if (p.retry_list) {
                    const u64 i = __hip_atomic_fetch_add(&p.counters[p.retry_counter], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    p.retry_list[i] = (u32)row;
                } else {
                    stat_add_final(ctl, sFailed, 1);
                    if (p.out_filled) p.out_filled[row] = 0;
                }
            }
            if (tid < (int)sNumStats) ctl->st_row[tid] = 0;
        };

        // the seed's table key carries its degree like every packed column id
        const u32 seed_deg = uni((u32)p.indptr[seed + 1]) - uni((u32)p.indptr[seed]);
        const u32 seed_id = p.gk_acsr ? uni(p.node_pos[seed]) : (u32)seed;                 // (self-addressed copy: keys are unit numbers)
        const u32 s_start = p.gk_acsr ? seed_id << kSkUnitShift : uni((u32)p.indptr[seed]);
        const int seed_key = (int)(seed_id | (min(seed_deg, p.deg_sat) << p.deg_shift));
#ifdef GP_DIAG
        if (tid == 0 && seed_deg != 0xFFFFFFFFu) rs1 = wall_clock64();     // after the queue -> seed -> indptr chain
#endif
        // state of the level about to be produced: its push list (built by the previous SCAN)
        u32 n_ent_cur = 0, e_cur = 0;
        u32 seg_begin = 0, seg_len = 0; int n_levels = 0;     // the first level with >= 2K reserve records (coef > 0), else the biggest so far
        u32 log_pos = 1;                                      // reserve-log records of the levels completed so far (record 0: level 0)
        double dang_cur = 0.0;
        bool has_dang_cur = false;
        int cur = 0;
        bool seedrow = false; double seed_share = 0.0;       // level 1 straight from the seed's CSR row (scan_level_dense<.., true>)

        // ---- level 0 without a table: the frontier is { seed : 1.0 } (graph.h:81), so its reserve
        //      record, its push test and its push-list entry are written directly.  This removes one
        //      EXPAND/SCAN round trip (two barriers, a table walk and a dependent indptr load) per row.
        {
            const double c0 = uni(ctl->coef[0]);
            PushEntry* push_nxt0 = push2 + (u64)(u32)p.push_cap;
            if (tid == 0) {
                if (p.log_cap > 0) { log_key[0] = seed_key; log_val[0] = c0; }                         // graph.h:90 / :109
                else ctl->fail = 1;
                stat_add(ctl, sFront, 1); stat_add(ctl, sDeg, 1);
                stat_add(ctl, p.force_global ? sGlb : sLds, 1);
            }
            n_levels = 1;
            if (c0 > 0.0) { seg_begin = 0; seg_len = 1; }
            if (L > 0) {
                if (seed_deg == 0) {                                                  // graph.h:91-93
                    dang_cur = 1.0; has_dang_cur = true;
                } else if (1.0 >= p.rmax * (double)seed_deg) {                        // graph.h:94
                    const double share = 1.0 / (double)seed_deg;                      // graph.h:95
                    if (tid == 0) { stat_add(ctl, sPush, 1); stat_add(ctl, sEdges, seed_deg); }
                    if (share != 0.0) {
                        e_cur = seed_deg; n_ent_cur = 1;
                        seed_share = share;
                        seedrow = p.rows_distinct && !p.force_global && !(kDirectOk<BLOCK> && p.direct) &&
                                  seed_deg <= (u32)(BLOCK / 64) * seedrow_slice<BLOCK>(C);
                        if (seedrow) {                       // level 1 needs neither the entry nor a table; its SCAN starts right behind this
                            if (tid == 0) { LevelCtr* n1 = &ctl->lc[1]; n1->dangling = 0.0; n1->n_dangling = 0; n1->n_rec = 0; n1->alloc = 0ull; }   // block's barrier
                        } else
                        if (tid == 0) {
                            if (p.push_cap > 0) { PushEntry pe; pe.rel = s_start; pe.off = 0; pe.share = share; push_nxt0[0] = pe; }
                            else ctl->fail = 1;
                        }
                        // the one entry contains every 64-edge boundary of the level (a hub seed: many)
                        const u32 units = (seed_deg + (1u << kUnitShift) - 1u) >> kUnitShift;
                        u32* bt_g = w.bt2 + (u64)(u32)p.bt_cap;
                        if (seedrow) { }
                        else if ((u64)units > p.bt_cap) { if (tid == 0) ctl->fail = 1; }
                        else
                            for (u32 m = (u32)tid; m < units; m += BLOCK) bt_g[m] = 0u;
                    }
                }
            }
            cur = 1;
            GP_SYNC();                            // push entry / boundary table / fail flag visible to every wave
        }
#ifdef GP_DIAG
        GP_STAMP(rs2);
        const u64 lv_all_e0 = tk_expand, lv_all_s0 = tk_scan;
#endif
        for (int lvl = 1; lvl <= L; ++lvl) {
            const double c = uni(lvl < kCoefLds ? ctl->coef[lvl] : p.coef[lvl]);
            const bool do_push = lvl < L;                                     // graph.h:83 vs :104
            // distinct targets of this level <= min(edges (+ the seed), N)
            const u64 need = min((u64)e_cur + (has_dang_cur ? 1 : 0), (u64)p.n_nodes);
            if (need == 0) break;                       // the frontier died: later levels add nothing
            max_e = max(max_e, e_cur);
            // what the LDS table is planned for: the estimate of DISTINCT targets (Ctl::ratio_q); a level that outgrows it overflows
            // its table and is split like any partition that does not fit
            u64 need_t = need;
            // (only where it changes the plan: at >= 0.8 targets per edge -- the power-law shapes -- the estimate buys no pass and an
            //  overflow now and then costs one: MAG -0.6 % when it was applied everywhere)
            if (lvl < 16) { const u32 rq = uni(ctl->ratio_q[lvl]); if (rq && rq <= 820u) need_t = min(need, (((u64)e_cur * rq) >> 10) + 64u); }
            // placement of the level's residue table
            bool in_lds = !p.force_global;
            u32 parts = 1, cap = 0;
            // direct-indexed tables (host sets p.direct only for the 512-thread kernel and N <= slots)
            const bool direct = kDirectOk<BLOCK> && p.direct && in_lds;
            if (direct) {
                cap = ((u32)p.n_nodes + 3u) & ~3u;       // slot = node id: one pass, no overflow
            } else if (in_lds) {
                // (planned in DISTINCT targets -- the estimate above applied: citation-style graphs -- the table may be planned full:
                //  Pubmed -3.5 % kernel time at 1.0 against 0.75; planned in edges, 0.875 / 1.0 cost MAG +8 / +18 %, Reddit +4 / +20 %)
                const u32 ld_num = need_t != need ? 1u : GP_LOAD_NUM, ld_den = need_t != need ? 1u : GP_LOAD_DEN;
                if (need_t * ld_den <= (u64)C * ld_num) {
                    cap = min(C, max(kMinCap, ((u32)GP_CAP_MULT * (u32)need_t + 3u) & ~3u));
                } else if (need_t > (u64)kMaxParts * C) {
                    in_lds = false;                      // more than kMaxParts partitions: the HBM table
                } else {
                    // target load of a partition: 0.75 of the table counted in EDGES (distinct targets are ~15 % fewer); a partition
                    // that overflows anyway is split in place.  0.55 -> 0.75 saved half a pass on the peak levels of the 80 KB shape (+2 %).
                    parts = ((u32)need_t * ld_den + C * ld_num - 1u) / (C * ld_num);     // need_t <= 64 C < 2^21: 32-bit arithmetic
                    cap = C;
                    if (parts > kMaxParts) in_lds = false;
                }
            }
            const u32 snap_log = log_pos;                      // first log record of this level
#ifdef GP_DIAG
            const u64 lv_e0 = tk_expand, lv_s0 = tk_scan; u32 lv_passes = 0;
#endif
            // Counters this level's SCAN will fill.  The other parity is what the threads have just
            // read (previous level), this one was last read two levels ago: thread 0 may clear it
            // now, and SCAN only starts after the end-of-EXPAND barrier.  No barrier needed here.
            LevelCtr* nx = &ctl->lc[lvl & 1];
            const bool lvl_seedrow = lvl == 1 && seedrow;      // (its counters were cleared in front of level 0's barrier)
            if (tid == 0 && !lvl_seedrow) {
                nx->dangling = 0.0; nx->n_dangling = 0; nx->n_rec = 0; nx->alloc = 0ull;
            }
            if (!in_lds) {
                parts = 1;
                const bool too_big = 2 * need > p.resg_cap;               // (wave-uniform)
                cap = (u32)min(p.resg_cap, max((u64)kMinCap, 2 * need));
                GP_SYNC();
                // raised behind the barrier, and seen by everyone behind the next: a wave may still have been reading the flag for the level before
                if (too_big) { if (tid == 0) ctl->fail = 1; GP_SYNC(); }
            }
            // Hash partitions (q, P) of the level's targets, refined in place on overflow exactly
            // like the aggregation partitions of topk_row (nothing to undo: a partition is
            // scanned only after its expansion succeeded).
            const bool bucketed = in_lds && parts >= kBucketMin && parts <= 64 &&
                                  (u64)e_cur + 1 <= p.bucket_cap;
            bool use_buckets = bucketed && !uni(ctl->fail);
            if (use_buckets) {
#ifdef GP_DIAG
                u64 tkd[3] = {0, 0, 0};
#endif
                use_buckets = phase_bucketed_level<BLOCK>(lds0, cap, parts, (u32)cur, n_ent_cur, e_cur, has_dang_cur ? 1u : 0u, dang_cur,
                                                          seed_key, (u32)(lvl & 1), c, do_push ? 1u : 0u GP_TK_ARGS) != 0;
#ifdef GP_DIAG
                tk_expand += tkd[0]; tk_scan += tkd[1]; lv_passes += (u32)tkd[2];
#endif
            }
            // a small level: one wave does it, the others park at one barrier (phase_solo_level)
            bool solo_done = false;
            if (in_lds && !direct && parts == 1 && !lvl_seedrow && !use_buckets && (p.solo & 1u) &&
                e_cur <= kSoloEdges && n_ent_cur >= 1u && n_ent_cur <= kSoloEntries && !uni(ctl->fail) &&
                solo_caps && (u64)log_pos + kSoloEdges + 2u <= p.log_cap) {
                GP_STAMP(t0);
                if (wave_id() == 0)
                    phase_solo_level<BLOCK>(lds0, (u32)lvl, (u32)cur, n_ent_cur, e_cur, has_dang_cur ? 1u : 0u, dang_cur, seed_key, c, do_push ? 1u : 0u);
                GP_SYNC();
                GP_STAMP(t2); GP_ACCUM(tk_scan, t0, t2);
                if (!uni(ctl->ovf)) solo_done = true;
                else { GP_SYNC(); if (tid == 0) ctl->ovf = 0; GP_SYNC(); }      // (undone by the wave: the general path takes the level)
            }
            if (!solo_done && !use_buckets && !uni(ctl->fail)) {
                {
                    u32 part = 0, np = parts;
                    for (;;) {
                        GP_STAMP(t0);
#ifdef GP_DIAG
                        ++lv_passes;
#endif
                        if (lvl_seedrow) {
                            phase_scan_seedrow<BLOCK>(lds0, s_start, seed_deg, seed_share, c, do_push ? 1u : 0u);
                            GP_SYNC();
                            GP_STAMP(t2); GP_ACCUM(tk_scan, t0, t2);
                            break;
                        }
#ifdef GP_DIAG_HEAVY
                        if (tid == 0) ctl->exp_c0 = clock64();
#endif
                        if (kDirectOk<BLOCK> && direct) {
                            phase_expand<BLOCK, kDirectOk<BLOCK> ? 2 : 0>(lds0, cap, (u32)cur, n_ent_cur, e_cur, part, np, has_dang_cur ? 1u : 0u, dang_cur, seed_key, 0u);
                        } else if (in_lds) {
                            phase_expand<BLOCK, 0>(lds0, cap, (u32)cur, n_ent_cur, e_cur, part, np, has_dang_cur ? 1u : 0u, dang_cur, seed_key, 0u);
#ifdef GP_DIAG
                            if (p.diag_flags & 2) phase_expand<BLOCK, 0>(lds0, cap, (u32)cur, n_ent_cur, e_cur, part, np, 0u, 0.0, seed_key, 1u);
#endif
                        } else {
                            phase_expand<BLOCK, 1>(lds0, cap, (u32)cur, n_ent_cur, e_cur, part, np, has_dang_cur ? 1u : 0u, dang_cur, seed_key, 0u);
                        }
                        GP_SYNC();
#ifdef GP_DIAG_HEAVY
                        if (tid == 0) { ctl->exp_sum_max += ctl->exp_max; ctl->exp_max = 0; ctl->exp_post += clock64() - ctl->exp_c2; }
#endif
                        GP_STAMP(t1); GP_ACCUM(tk_expand, t0, t1); if (!in_lds) GP_ACCUM(tk_expand_hbm, t0, t1);
                        if (uni(ctl->fail)) break;
                        if (uni(ctl->ovf)) {
                            // this partition did not fit: wipe the table and split it in two
                            wipe_table<BLOCK>(lkeys, lvals, C);
                            GP_SYNC();
                            if (tid == 0) ctl->ovf = 0;
                            if (in_lds && !direct) cap = C;      // (a single-pass table sized from the estimate may have been smaller)
                            if (np < 0x20000000u) { part *= 2; np *= 2; GP_SYNC(); continue; }
                            if (tid == 0) ctl->fail = 1;
                            GP_SYNC();
                            break;
                        }
                        if (in_lds) {
                            phase_scan_dense<BLOCK>(lds0, cap, (u32)(lvl & 1), (u32)(cur ^ 1), c, do_push ? 1u : 0u);
#ifdef GP_DIAG
                            if (p.diag_flags & 4) {                  // a second walk over the (now empty) table: cost of stage (a) alone
                                GP_SYNC();
                                phase_scan_dense<BLOCK>(lds0, cap, (u32)(lvl & 1), (u32)(cur ^ 1), c, do_push ? 1u : 0u);
                            }
#endif
                        } else {
                            phase_scan_hbm<BLOCK>(lds0, cap, (u32)(lvl & 1), (u32)(cur ^ 1), c, do_push ? 1u : 0u);
                        }
                        GP_SYNC();
                        GP_STAMP(t2); GP_ACCUM(tk_scan, t1, t2); if (!in_lds) GP_ACCUM(tk_scan_hbm, t1, t2);
                        if (uni(ctl->fail)) break;
                        while (np > parts && (part & 1u)) { part >>= 1; np >>= 1; }   // right child done => parent done
                        ++part;
                        if (np == parts && part == parts) break;
                    }
                }
            }
            if (tid == 0) stat_add(ctl, in_lds ? sLds : sGlb, 1);
#ifdef GP_DIAG
            if (tid == 0) {            // (in LDS: a global atomic here would be awaited at the next phase call's entry)
                u64* dx = ctl->lvl_acc[min(lvl, 15)];
                dx[0] += tk_expand - lv_e0; dx[1] += tk_scan - lv_s0; dx[2] += (u64)e_cur;
                dx[3] += (u64)nx->n_rec; dx[4] += (u64)n_ent_cur; dx[5] += (u64)lv_passes;
            }
#endif
            {
                const u32 lvl_len = uni(nx->n_rec);                      // read after the level's last barrier (see LevelCtr::n_rec)
                log_pos += lvl_len;
                if (tid == 0 && lvl < 16 && e_cur > 0u) {                // lvl_len = the level's distinct targets
                    const u32 obs = min(1024u, (u32)(1178.0f * (float)lvl_len * __frcp_rn((float)e_cur)) + 2u);    // 1.15 x 1024 x nodes / edges, rounded up (fp32: a 64-bit division here is ~150 instructions per level)
                    const u32 old_q = ctl->ratio_q[lvl];
                    ctl->ratio_q[lvl] = max(obs, old_q - (old_q >> 3));
                }
                n_levels = lvl + 1;
                // first level with >= 2K records (early levels hold the LARGEST records: a stronger bound than the biggest level)
                if (c > 0.0 && seg_len < 2u * (u32)p.K && lvl_len > seg_len) {
                    seg_begin = snap_log; seg_len = lvl_len;
                    if (seg_len >= 2u * (u32)p.K && p.prune && do_push && !uni(ctl->fail)) phase_tau<BLOCK>(lds0, seg_begin, seg_len, (u32)(L + 1));
                }
            }
            if (!do_push || uni(ctl->fail)) break;
            { const u64 al = uni(nx->alloc); n_ent_cur = (u32)al; e_cur = (u32)(al >> 32); }
            has_dang_cur = uni(nx->n_dangling) != 0; dang_cur = has_dang_cur ? uni(nx->dangling) : 0.0;
            cur ^= 1;
        }
        GP_SYNC();
        {
            const u32 failed = uni(ctl->fail), n_log_row = uni(ctl->log_count);      // one round trip
            max_log = max(max_log, n_log_row);
            if (failed) {
                // Leave the row unwritten.  Restore clean tables so that later rows of this workgroup are unaffected.
                give_up();
                wipe_table<BLOCK>(lkeys, lvals, C);
                for (u64 i = tid; i < p.resg_cap; i += BLOCK) { st_l2(&resg[i].key, kEmpty); st_l2(&resg[i].val, 0.0); }
                continue;
            }
        }
        GP_STAMP(t0);
#ifdef GP_DIAG
        if (tid == 0) {       // [4] row prologue, [5] level 0, [6] level loop outside EXPAND/SCAN, [7] table restore after TOP-K (added below)
            ctl->row_acc[0] += rs1 - rs0; ctl->row_acc[1] += rs2 - rs1;
            ctl->row_acc[2] += (t0 - rs2) - (tk_expand - lv_all_e0) - (tk_scan - lv_all_s0);
        }
        if (!(p.diag_flags & 1))
#endif
        phase_topk<BLOCK>(lds0, (u32)(u64)row, (u32)((u64)row >> 32), seed, seg_begin, seg_len, n_levels GP_SUB_ARGS);
        GP_SYNC();
        if (uni(ctl->fail)) give_up();                                      // the candidate array overflowed: nothing was written
        else if (tid < (int)sNumStats) { ctl->st[tid] += ctl->st_row[tid]; ctl->st_row[tid] = 0; }      // the row is done: its counts count
        GP_STAMP(t1); GP_ACCUM(tk_topk, t0, t1);
        // top-K used the table region as scratch: restore the empty LDS table
        wipe_table<BLOCK>(lkeys, lvals, C);
#ifdef GP_DIAG
        GP_STAMP(rs3);
        if (tid == 0) ctl->row_acc[3] += rs3 - t1;
#endif
    }

    // flush statistics: one atomic per counter per workgroup
    GP_SYNC();
    if (tid == 0) {
        const Counter dst[sNumStats] = { kPushes, kEdges, kFrontier, kDegLookups, kFilled, kSupport, kLdsLevels, kGlobalLevels, kFailedRows };
#pragma unroll
        for (int i = 0; i < sNumStats; ++i)
            if (ctl->st[i]) __hip_atomic_fetch_add(&p.counters[dst[i]], ctl->st[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (p.row_map && blockIdx.x == 0 && n_rows > 0)
            __hip_atomic_fetch_add(&p.counters[kRetriedTotal], (u64)n_rows, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (max_e) __hip_atomic_fetch_max(&p.counters[kMaxLevelEdges], (u64)max_e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (max_log) __hip_atomic_fetch_max(&p.counters[kMaxLogRecords], (u64)max_log, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifdef GP_DIAG
        tk_total = wall_clock64() - tk_begin;
        __hip_atomic_fetch_add(&p.counters[kTicksScan], tk_scan, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(&p.counters[kTicksExpand], tk_expand, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(&p.counters[kTicksTopk], tk_topk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(&p.counters[kTicksTotal], tk_total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(&p.counters[kTicksScanHbm], tk_scan_hbm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(&p.counters[kTicksExpandHbm], tk_expand_hbm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int i = 0; i < 16; ++i)
            __hip_atomic_fetch_add(&p.counters[kDiag0 + i], gp_sub_acc[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int i = 0; i < 4; ++i)
            __hip_atomic_fetch_add(&p.counters[kDiagX0 + 4 + i], ctl->row_acc[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int i = 0; i < 96; ++i)
            if (ctl->lvl_acc[i / 6][i % 6]) __hip_atomic_fetch_add(&p.counters[kDiagX0 + 16 + i], ctl->lvl_acc[i / 6][i % 6], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
    }
#ifdef GP_DIAG
    if ((tid & 63) == 0) {
        __hip_atomic_fetch_add(&p.counters[kDiagX0 + 0], clock64() - wave_t0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(&p.counters[kDiagX0 + 1], ctl->barw[tid >> 6], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(&p.counters[kDiagX0 + 2], (u64)ctl->barn[tid >> 6], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tid == 0) for (int i = 0; i < 4; ++i)
            __hip_atomic_fetch_add(&p.counters[kDiagX0 + 8 + i], ctl->scan_sub[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tid == 0) for (int i = 0; i < 8; ++i)
            __hip_atomic_fetch_add(&p.counters[kDiagX0 + 112 + i], ctl->exp_sub[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tid == 0) {
            __hip_atomic_fetch_add(&p.counters[kDiagX0 + 120], ctl->exp_sum_max, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(&p.counters[kDiagX0 + 121], ctl->exp_sum_all, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(&p.counters[kDiagX0 + 122], ctl->exp_pre, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(&p.counters[kDiagX0 + 123], ctl->exp_post, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __syncthreads();
    if (tid < 64 && ctl->site_n[tid]) {       // [128 + site] cycles waited at GP_SYNC() number `site` (source order), [192 + site] wave arrivals
        __hip_atomic_fetch_add(&p.counters[kDiagX0 + 128 + tid], ctl->site_w[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(&p.counters[kDiagX0 + 192 + tid], (u64)ctl->site_n[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#endif
}

// The two launches of a call are separate kernel symbols so that profilers report them separately (the retry launch is a
// few microseconds of nothing whenever no row outgrew its slab, and would halve the "average gfpush_kernel duration").
template <int BLOCK>
__global__ void __launch_bounds__(BLOCK, BLOCK == 768 ? GP_MINW_768 : BLOCK == 512 ? GP_MINW_512 : BLOCK == 1024 ? GP_MINW_1024 : 4) gfpush_kernel(const KParams)
{
    gfpush_rows<BLOCK>();
}
template <int BLOCK>
__global__ void __launch_bounds__(BLOCK, BLOCK == 768 ? GP_MINW_768 : BLOCK == 512 ? GP_MINW_512 : BLOCK == 1024 ? GP_MINW_1024 : 4) gfpush_retry_kernel(const KParams)
{
    gfpush_rows<BLOCK>();
}

// Fills the per-workgroup HBM residue tables with empty records (a byte memset cannot: val must be 0).
__global__ void __launch_bounds__(256) init_tables_kernel(ResRec* resg, u64 n_res)
{
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n_res; i += stride) {
        ResRec r; r.key = kEmpty; r.pad = 0; r.val = 0.0; resg[i] = r;
    }
}

// Packs min(deg(column), deg_sat) above the column id in every word of the device copy of
// `indices` (done once per graph).  Column ids were validated on the host before upload.
__global__ void __launch_bounds__(256) pack_degree_kernel(const int* indptr, int* indices, long long nnz,
                                                          int deg_shift, u32 deg_sat)
{
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x; j < nnz; j += stride) {
        const int v = indices[j];
        const u32 d = (u32)(indptr[v + 1] - indptr[v]);
        indices[j] = (int)((u32)v | (min(d, deg_sat) << deg_shift));
    }
}

}  // namespace gp
