/*
 * grandplus.h -- C ABI of the MI355X-native GFPush propagation-matrix precompute.
 *
 * This is the drop-in boundary for the ONE hot path of THUDM/GRAND-plus: the pybind11
 * class `propagation.Graph` (reference precompute/propagation.cpp:8-12) whose two
 * members are `Graph(indptr, indices, seed)` (precompute/graph.h:32-47) and
 * `gfpush_omp(node_idx, row_idx, col_idx, value, coef, rmax, K)`
 * (precompute/graph.h:53-131).  Every entry point takes plain pointers and sizes; no
 * torch / pybind11 / HIP types appear in a signature (streams travel as void*).
 *
 * Implemented by grand_plus_amd/libgrandplus.so (grand_plus_amd/csrc/gfpush.hip, hand-written
 * HIP for gfx950).  There is no CPU fallback: without a usable GPU gp_graph_create
 * returns GP_ERR_NO_DEVICE.
 */
#ifndef GRANDPLUS_H
#define GRANDPLUS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GP_ABI_VERSION 4

/* Status codes (0 = success).  The Python / pybind11 shims map INVALID_* to
 * ValueError, NO_DEVICE / HIP / OVERFLOW to RuntimeError, NOMEM to MemoryError. */
enum {
    GP_OK                = 0,
    GP_ERR_NULL          = 1,   /* a required pointer is NULL                                   */
    GP_ERR_INVALID_CSR   = 2,   /* indptr not monotone / indptr[0]!=0 / indptr[n]!=nnz / column out of range */
    GP_ERR_INVALID_SEED  = 3,   /* a seed id is outside [0, n_nodes)                            */
    GP_ERR_INVALID_ARG   = 4,   /* K < 1, K > GP_MAX_K, n_coef < 1, non-finite coef/rmax, rmax < 0, negative size */
    GP_ERR_NO_DEVICE     = 5,   /* no HIP device / device index out of range                    */
    GP_ERR_HIP           = 6,   /* a HIP runtime call failed (see gp_last_error)                */
    GP_ERR_NOMEM         = 7,   /* device or host allocation failed                             */
    GP_ERR_OVERFLOW      = 8    /* a row exceeded a workspace bound (never expected; rows are not silently wrong) */
};

#define GP_MAX_K 1024

typedef struct gp_graph gp_graph;   /* opaque: CSR resident in one GPU's HBM + workspaces */

/* Counters accumulated over the gfpush calls on a graph since the last gp_reset_stats
 * (gp_gfpush resets them itself, so after it they describe that one call).  All exact. */
typedef struct gp_stats {
    int64_t rows;            /* seeds processed                                                  */
    int64_t pushes;          /* P: (node, level) pushes        -- graph.h:94 branch taken        */
    int64_t edges;           /* E: sum of deg over the pushes  -- graph.h:96-99 iterations       */
    int64_t filled;          /* output slots written (v > 0)   -- graph.h:121                    */
    int64_t support;         /* sum over rows of reserve-map size -- graph.h:111 (exact with option exact_stats=1) */
    int64_t frontier;        /* sum over rows and levels of frontier size                        */
    int64_t lds_levels;      /* levels whose residue table lived in LDS                          */
    int64_t global_levels;   /* levels whose residue table lived in the per-workgroup HBM table  */
    int64_t failed_rows;     /* rows that hit a workspace bound (=> GP_ERR_OVERFLOW)             */
    int64_t degree_lookups;  /* frontier nodes whose degree (two indptr words) had to be read    */
    double  kernel_ms;       /* HIP-event time of the LAST gfpush kernel on its stream           */
    int32_t workgroups;      /* persistent workgroups launched                                   */
    int32_t block_threads;   /* threads per workgroup                                            */
    int32_t lds_bytes;       /* dynamic LDS per workgroup                                        */
    int32_t lds_slots;       /* residue-table slots that fit in that LDS                         */
    int64_t workspace_bytes; /* HBM scratch held for this configuration                          */
    /* Filled only by the diagnostic build (libgrandplus_diag.so, -DGP_DIAG): 100 MHz ticks
     * spent per phase, summed over workgroups.  Always 0 in the product library. */
    int64_t diag_ticks_scan, diag_ticks_expand, diag_ticks_topk, diag_ticks_total;
    int64_t diag_ticks_scan_hbm, diag_ticks_expand_hbm;   /* the part of scan/expand spent on HBM-table levels */
    int64_t diag_sub[16];    /* free-form sub-phase ticks / counts of the diagnostic build          */
    /* ABI 2: workspace policy.  Every workgroup's scratch slab is sized from an estimate; rows that outgrow it are
     * re-run by a second launch on a few workgroups whose slabs are sized from the rigorous 1/rmax bounds. */
    int64_t retried_rows;    /* rows that took that second launch (they are complete and exact, just slower) */
    int64_t max_level_edges; /* largest number of edges one level of one row traversed                       */
    int64_t max_log_records; /* largest number of reserve-log records (sum over levels of frontier size) of one row */
    /* ABI 3: which kernel ran the rows.  2 = the sketch-filtered kernel (csrc/gfpush_sketch.hpp: keyless fixed-point upper
     * bounds decide which targets get an exact fp64 accumulator; `frontier` and `support` are not counted by it and stay 0,
     * its log holds one record per pushed edge, so max_log_records counts edges), 1 = the general kernel
     * (csrc/gfpush_kernels.hpp), which also re-runs the rows the sketch kernel hands back (retried_rows). */
    int32_t kernel;
    int32_t sketch_pad;
    int64_t sketch_candidate_edges;  /* pushed edges whose target might push and went into the exact table (of `edges`) */
    int64_t sketch_second_sweeps;    /* rows whose top-K needed a second sweep over their log                          */
    /* ABI 4: the measured choice.  With nothing forced (options kernel / block_threads / lds_bytes all 0, "measure_choice" 1) the
     * first call of >= 32 768 rows of a recipe (rmax, n_coef, K) on a graph times its candidates -- they are exact and
     * interchangeable -- on its first 16 384 rows (into scratch outputs), and every later call of that recipe, of any size, runs what the rmax / graph-size thresholds
     * pick unless another candidate was more than 5 % faster: [0] the general kernel in the launch shape the thresholds pick,
     * [1] the sketch kernel, [2] the general kernel in the other shape.
     * Milliseconds of those timing runs behind the LAST call's decision (0 = not a candidate / nothing was measured for it);
     * `kernel`, `block_threads`, `lds_bytes` say what ran. */
    float   choice_ms[3];
    int32_t choice_pad;
} gp_stats;

/* ABI / build information. */
int         gp_abi_version(void);
const char* gp_strerror(int status);
const char* gp_last_error(void);          /* thread-local detail of the last failure ("" if none) */
int         gp_device_count(void);        /* number of visible HIP devices (0 if none)           */

/*
 * Replaces Graph::Graph (graph.h:32-47).  Validates the CSR (the reference does not), copies
 * indptr/indices to `device`'s HBM (the reference borrows the caller's buffers: graph.h:35-36;
 * here the caller may free them after the call) and derives degrees on the fly from indptr
 * (graph.h:42-45).  The reference's unused `seed` argument (graph.h:40) is accepted by the
 * shims and dropped before this call.
 * indptr is checked on the host (O(N)); the column ids are checked on the DEVICE behind the upload (range; and whether
 * every row is strictly increasing, which only decides a shortcut): GP_ERR_INVALID_CSR either way, before the handle exists.
 * Arrays of >= 64 MB are uploaded through four 16 MB pinned staging buffers filled by two helper threads, joined before
 * the call returns.
 */
int gp_graph_create(const int32_t* indptr, int64_t n_nodes,
                    const int32_t* indices, int64_t nnz,
                    int device, gp_graph** out);

/*
 * Multi-GPU form of the same constructor: ONE process, ONE call, every GPU of the node.
 * n_gpus = 0 means all visible devices.  The CSR is uploaded to GPU 0; the other GPUs receive their replica by a
 * device-to-device copy the first time a gp_gfpush call is large enough to be sharded (>= "min_rows_per_gpu" rows per
 * GPU, default 2048; smaller calls run on GPU 0 alone).  gp_gfpush on such a handle cuts the seeds into contiguous
 * blocks of ceil(S / n_gpus) rows, runs every block on its GPU from its own host thread and stream, reassembles the
 * sparse row matrix with ONE ncclAllGather (RCCL over xGMI) of the packed per-GPU slabs
 * [value f64 | row i32 | col i32 | filled i32], copies it to the host once from GPU 0 and writes the v > 0 slots into
 * the caller's arrays (graph.h:117-126).  RCCL is opened with dlopen the first time it is needed.
 * gp_gfpush_device is refused on a multi-GPU handle (device buffers live on one GPU).  Extra options of such a handle:
 *   "min_rows_per_gpu"  sharding threshold (default 2048)
 *   "gather_host"       1 = no collective: every GPU copies its own slab to the host (measured comparison)
 *   "force_collective"  1 = take the sharded path, all-gather included, even for small calls / one GPU (tests)
 * every other option is forwarded to the per-GPU graphs.  gp_get_stats returns the sums over the GPUs of the last call
 * (kernel_ms = the slowest GPU).
 */
int gp_graph_create_multi(const int32_t* indptr, int64_t n_nodes,
                          const int32_t* indices, int64_t nnz,
                          int n_gpus, gp_graph** out);
/* The same constructor over an explicit list of devices.  A device may appear more than once: the handle then holds several
 * parts -- each with its own replica of the CSR, stream, workspace and host thread -- on that one GPU and gathers the slabs
 * through the host (RCCL takes one rank per device; "gather_host" cannot be switched off on such a handle).  This is how a box
 * with a single GPU runs the whole sharded path of gp_gfpush; it is not a way to go faster. */
int gp_graph_create_multi_on(const int32_t* indptr, int64_t n_nodes,
                             const int32_t* indices, int64_t nnz,
                             const int* devices, int n_parts, gp_graph** out);
int gp_graph_num_gpus(const gp_graph* g);

void gp_graph_destroy(gp_graph* g);

int64_t gp_graph_num_nodes(const gp_graph* g);
int64_t gp_graph_nnz(const gp_graph* g);
int     gp_graph_device(const gp_graph* g);

/*
 * Replaces Graph::gfpush_omp (graph.h:53-131) with HOST buffers, exactly as the pybind11
 * surface hands them over: seeds int32[S]; row_idx/col_idx int32[S*K], value f64[S*K]
 * caller-allocated and written IN PLACE at slot it*K+i (graph.h:120); only slots with
 * v > 0 are written (graph.h:121), everything else keeps the caller's contents.  Within a row
 * the filled slots are i = 0..filled-1 ordered by (value desc, column asc) -- the reference
 * leaves that order unspecified (nth_element, graph.h:115).  Synchronous.  The calling thread merges
 * finished rows into the caller's arrays while the kernels run; one helper thread, joined before the
 * call returns, resets the pinned slab the next call will use.  No OpenMP, no process-wide settings: the one
 * thing the call touches outside its own memory is the CALLING THREAD's timer slack (prctl PR_SET_TIMERSLACK,
 * 1 us while it waits for rows in 60 us sleeps; put back before the call returns).
 */
int gp_gfpush(gp_graph* g,
              const int32_t* seeds, int64_t n_seeds,
              const double* coef, int n_coef, double rmax, int K,
              int32_t* row_idx, int32_t* col_idx, double* value);

/*
 * Same computation with DEVICE buffers on g's GPU, enqueued on `stream` (a hipStream_t
 * passed as void*; NULL = the default stream) and NOT synchronised: the form bench.py and
 * the multi-GPU driver use (inputs already resident in HBM, RCCL all-gather consumes the
 * outputs on device).  d_row/d_col/d_val are [S*K]; slots i >= d_filled[it] of a row are left
 * untouched.  d_filled is int32[S] (may be NULL).  coef is a HOST array (n_coef doubles).
 * ONE-OFF SYNCHRONOUS CALLS: the first call that takes the sketch kernel builds the self-addressed CSR (one stream
 * synchronisation), and the first call of >= 32 768 rows of a recipe times its candidates first (gp_stats.choice_ms: six
 * launches of 16 384 rows into scratch output buffers of the library's own -- never into d_row/d_col/d_val -- and two stream
 * synchronisations).  Every other call only enqueues.  Option "measure_choice" = 0 avoids the second.
 */
int gp_gfpush_device(gp_graph* g,
                     const int32_t* d_seeds, int64_t n_seeds,
                     const double* coef, int n_coef, double rmax, int K,
                     int32_t* d_row, int32_t* d_col, double* d_val, int32_t* d_filled,
                     void* stream);

/* Waits for the last gfpush on g and returns the counters (GP_ERR_OVERFLOW if a row failed). */
int gp_get_stats(gp_graph* g, gp_stats* out);

/* Zeroes the counters at the next gfpush call. */
int gp_reset_stats(gp_graph* g);

/*
 * Tuning knobs (all optional; defaults are chosen for MI355X):
 *   "block_threads"   256 | 512 | 1024      threads per persistent workgroup
 *   "lds_bytes"       dynamic LDS per workgroup (<= 163840)
 *   "max_workgroups"  upper bound on persistent workgroups (0 = CUs x resident blocks)
 *   "workspace_mb"    HBM scratch budget in MiB (default 65536, never more than 90 % of the free device memory).
 *                      Under pressure the per-workgroup slabs shrink, not the number of workgroups; rows that
 *                      outgrow their slab are re-run on a few workgroups with worst-case slabs (gp_stats.retried_rows)
 *   "est_level_edges" edges per level the per-workgroup slabs are sized for (0 = automatic: max(32768, bound/4),
 *                      grown from the observed maxima of earlier calls)
 *   "force_global"    1 = never use the LDS residue table (testing the HBM-table path)
 *   "max_degree_bits" cap on the spare column-id bits used to carry degrees (0 = none; testing the
 *                      path taken by graphs with N >= 2^29); only before the first gfpush call
 *   "exact_stats"     1 = always aggregate the whole reserve map, so that gp_stats.support is the
 *                      exact sum of reserve-map sizes (default 0: nodes that provably cannot
 *                      reach the top-K are never tabled and `support` counts only tabled nodes)
 *   "direct_tables"   0 = never index the level tables by node id (default 1: graphs with N <= table slots of the
 *                      512-thread kernel -- Cora, Citeseer -- skip hashing altogether)
 *   "seedrow"         0 = level 1 of a row goes through EXPAND and a hash table like every other level (default 1: on graphs
 *                      whose CSR rows hold strictly increasing column ids the seed's neighbour list IS level 1's frontier)
 *   "solo_levels"     0 = levels of <= 256 edges go through EXPAND + SCAN of the whole workgroup (default 1: one wave does such a
 *                      level start to finish, the others park at one barrier)
 *   "kernel"          0 = choose per call (default), 1 = always the general kernel, 2 = the sketch-filtered kernel whenever
 *                      the call allows it (all coef >= 0, at most 40 levels, K <= 128, rmax > 0).  Automatic choice: the
 *                      sketch kernel for rmax >= 5e-6 on graphs of >= 65 536 nodes
 *   "sk_block_threads" / "sk_lg_mu" / "sk_lg_mr" / "sk_target"   geometry of the sketch kernel (0 = default): threads per
 *                      workgroup (512 = three per CU with 52 KB, 768 = two with 80 KB, 1024 = one with 160 KB: measurements
 *                      only), log2 cells of the level sketch and of the reserve sketch, cell rank of the first TOP-K
 *                      threshold (default 4 K)
 *   "gk_acsr"         0 = the general kernel always runs on the packed CSR + indptr (default 1: on graphs of >= 65 536 nodes
 *                      it runs on the self-addressed copy -- rows at 128-byte units, a pusher's row start and degree in
 *                      its key -- like the sketch kernel)
 *   "verify_merge"    1 = gp_gfpush (host buffers) compares every row it merged while the kernel was running with the pinned
 *                      slab once the launches have retired and fails with GP_ERR_HIP if one differs (a debugging aid: the
 *                      merge rule relies on stores to host memory arriving whole; default 0)
 *   "diag_flags"      ignored by the product library; the diagnostic build (-DGP_DIAG) skips phases for
 *                      instruction attribution (bit 0: TOP-K) -- its results are then meaningless
 * Returns GP_ERR_INVALID_ARG for an unknown key or an out-of-range value.
 */
int gp_set_option(gp_graph* g, const char* key, int64_t value);

/* ------------------------------------------------------------------------------------------
 * Next row of the scope table: GRAND+'s feature augmentation
 * ("random propagation"), reference Grand_Plus.random_prop (model.py:80-87, model_mag.py:80-86):
 * DropNode on the scores, weighted segment-sum of neighbour features, divide by
 * (sum of kept scores + 1e-12).  fp32 like the reference.  All pointers are DEVICE pointers on
 * `device`; the launch is asynchronous on `stream`.
 *
 * `training` != 0 applies dropout with rate `dropnode_rate`: kept scores are scaled by
 * 1/(1-rate) (model.py:82).  The keep decision of entry e is d_keep[e] != 0 when d_keep is
 * given (parity tests), otherwise a counter-based RNG of (seed, e) -- change `seed` every call.
 * ------------------------------------------------------------------------------------------ */

/* Fused form: consumes the [S x K] rows gfpush left on the device.  Output row b is built from
 * matrix row r = d_batch_rows[b] (r = b when d_batch_rows is NULL), using its first d_filled[r]
 * slots (all K when d_filled is NULL): out[b,:] = sum_k w_k X[col[r,k],:] / (sum_k w_k + 1e-12),
 * w_k = (float)val[r,k] * keep_k/(1-rate).  Replaces the caller-side slicing, feature gather and
 * upload of model.py:310-316 together with random_prop itself.  d_x is X[n_nodes x feat_dim]. */
int gp_random_prop_rows(int device, const float* d_x, int64_t n_nodes, int32_t feat_dim,
                        const int32_t* d_col, const double* d_val, const int32_t* d_filled, int32_t K,
                        const int32_t* d_batch_rows, int32_t n_batch,
                        float dropnode_rate, int training, uint64_t seed, const uint8_t* d_keep,
                        float* d_out, void* stream);

/* SURVEY.md 8f next-3 -- where is the row of node v?  The reference slices `topk_adj[batch_index]` on the CPU every step
 * (model.py:310); here the seed list leaves a device-resident index once per precompute and a batch of node ids is turned into
 * row positions by one small kernel, no host work per element.
 * gp_seed_positions: d_pos_of_node int32[n_nodes] <- first position of every seed of d_seeds[n_seeds] (a duplicated seed keeps
 * its first), -1 for nodes that are no seed; *d_n_bad (device int32) <- seeds outside [0, n_nodes).
 * gp_batch_positions: d_out[i] <- d_pos_of_node[d_node_ids[i]] (int64 ids, as torch index tensors are), -1 for an id that is no
 * seed or out of range; *d_n_missing (device int32) <- how many of those.  Both are enqueued on `stream`. */
int gp_seed_positions(int device, const int32_t* d_seeds, int64_t n_seeds, int64_t n_nodes, int32_t* d_pos_of_node, int32_t* d_n_bad, void* stream);
int gp_batch_positions(int device, const int32_t* d_pos_of_node, int64_t n_nodes, const int64_t* d_node_ids, int64_t n,
                       int32_t* d_out, int32_t* d_n_missing, void* stream);

/* Reference-shaped form: feats[n_entries x feat_dim] already gathered, scores[n_entries],
 * idx[n_entries] sorted ascending (the row-major order of scipy's .nonzero(), model.py:312);
 * out[n_out x feat_dim] with n_out = idx[-1] + 1 (model.py:84). */
int gp_random_prop_coo(int device, const float* d_feats, int64_t n_entries, int32_t feat_dim,
                       const float* d_scores, const int64_t* d_idx, int64_t n_out,
                       float dropnode_rate, int training, uint64_t seed, const uint8_t* d_keep,
                       float* d_out, void* stream);

/* ------------------------------------------------------------------------------------------
 * exact full-graph feature propagation of the inference path, reference
 * predict() (model.py:181-224), lines 186-210.  mode 0 = ppr, 1 = avg, 2 = single (args.prop_mode);
 * `order` = args.order propagation steps.  A is the CSR of `g` (adj + I as the caller built it,
 * model.py:243); d_edge_weight = its stored values (float32[nnz], device) or NULL for an all-ones
 * matrix -- what adj + I is for every shipped dataset.  d_features / d_out are float32[n_nodes x
 * feat_dim] on g's device (the reference iterates in float64 and casts the result to float32,
 * model.py:175; here sums are fp64, storage fp32).  Enqueued on `stream`; one host synchronisation
 * happens inside (long-row census).
 * ------------------------------------------------------------------------------------------ */
int gp_propagate_features(gp_graph* g, const float* d_features, int32_t feat_dim, const float* d_edge_weight,
                          int mode, int order, double alpha, float* d_out, void* stream);

/* internal: the device CSR of a graph for the other translation units (indices words carry degree bits
 * above *node_mask; packs them first if that has not happened yet) */
int gp_internal_graph_csr(gp_graph* g, const int** d_indptr, const int** d_indices, uint32_t* node_mask, void* stream);

/* internal (tests): the SELF-ADDRESSED copy of the CSR the sketch kernel runs on (built now if it does not exist yet), copied
 * to the host.  Rows start at units of 32 column words (128 bytes); node_pos[u] = first unit of row u (n_nodes + 1 entries); a
 * column word is (unit of the target) | min(deg(target), *deg_sat) << *unit_bits; padding and the sentinel word
 * acsr[32 * *n_units] are -1; unit_info[first unit of a row] = its node id, [second unit of a multi-unit row] = its degree.
 * Call with NULL buffers for the sizes, then with h_acsr int32[32 * n_units + 1], h_node_pos uint32[n_nodes + 1], h_unit_info
 * int32[n_units].  *n_units = 0: the graph does not allow the layout (unit numbers would leave fewer than two degree bits) and
 * the general kernel takes every call. */
int gp_internal_graph_acsr(gp_graph* g, int32_t* h_acsr, uint32_t* h_node_pos, int32_t* h_unit_info,
                           int64_t* n_units, int* unit_bits, uint32_t* deg_sat);

/* internal: the extended counters of the diagnostic build (-DGP_DIAG; all 0 in the product library): [0] wave
 * cycles, [1] cycles the waves waited at workgroup barriers, [2] barriers passed, [16 + 6*level + k] per level
 * (k = 0 expand ticks, 1 scan ticks, 2 edges, 3 frontier nodes, 4 push-list entries, 5 table passes); n <= 128 */
int gp_internal_diag_counters(gp_graph* g, int64_t* out, int n);

/* internal (host arithmetic only, no GPU needed): how gp_gfpush cuts a call of n_seeds rows over the n_parts GPUs of a
 * multi-GPU handle.  out[0] = GPUs that compute, [1] = GPUs in the gather, [2] = rows per GPU = ceil(S / G),
 * [3] = bytes per packed slab, [4] = 1 when the call is handed to GPU 0 as it is; then (first row, rows) per GPU.
 * n_out >= 5 + 2 * n_parts.  The CPU tests drive the >= 2-GPU partitioning through this seam. */
int gp_internal_multi_plan(int64_t n_seeds, int K, int n_parts, int64_t min_rows_per_gpu, int force_collective, int gather_host,
                           int64_t* out, int n_out);

/* internal: lets the second translation unit report through gp_last_error (not for callers) */
void gp_internal_set_error(int status, const char* where, const char* detail);
/* Milliseconds the calling thread's last gp_graph_create spent: [0] HIP runtime + device, [1] allocations, [2] upload, [3] validation on
 * the device, [4] per-graph objects (bench.py's cold_call block). */
void gp_internal_create_ms(double* out5);
/* Starts the HIP runtime's context on `device` (hipSetDevice + hipFree(NULL)): what a process's first allocation otherwise pays. */
int gp_internal_warm_device(int device);

#ifdef __cplusplus
}
#endif
#endif /* GRANDPLUS_H */
