"""GPU parity tests of the sketch-filtered kernel (grand_plus_amd/csrc/gfpush_sketch.hpp, option kernel = 2) against the
CPU oracle: same rows (tie-aware comparator), same exact work counters (pushes, edges, filled)."""
import numpy as np
import pytest

from test_gpu_parity import KAT_COEF, KAT_INDICES, KAT_INDPTR, _assert_parity, _oracle, _random_digraph, _run_gpu

pytestmark = pytest.mark.gpu

SK = {"kernel": 2}


def _check(indptr, indices, seeds, coef, rmax, K, options, fill=(0, 0, 0.0), want_kernel=2, max_retry_frac=None, label=""):
    got, st = _run_gpu(indptr, indices, seeds, coef, rmax, K, fill=fill, options=options)
    exp, ost = _oracle(indptr, indices, seeds, coef, rmax, K, fill=fill)
    assert st["kernel"] == want_kernel, (label, st["kernel"])
    _assert_parity(seeds, K, got, exp, fill=fill, label=label)
    assert (st["pushes"], st["edges"], st["filled"]) == (ost["pushes"], ost["edges"], ost["filled"]), (label, st, ost)
    assert st["failed_rows"] == 0
    if max_retry_frac is not None:
        assert st["retried_rows"] <= max_retry_frac * len(seeds), (label, st["retried_rows"], len(seeds))
    return st


@pytest.mark.parametrize("mode,order,alpha,rmax,K", [
    ("ppr", 6, 0.2, 1e-5, 16), ("avg", 4, 0.2, 1e-5, 32), ("ppr", 10, 0.2, 1e-5, 32), ("ppr", 10, 0.2, 1e-4, 64),
    ("single", 2, 0.2, 1e-5, 32), ("ppr", 3, 0.5, 1e-3, 8), ("avg", 6, 0.2, 2e-6, 64),
])
@pytest.mark.parametrize("block", [512, 768])
def test_sketch_kernel_synthetic_small(mode, order, alpha, rmax, K, block):
    from grand_plus_amd import synth
    from grand_plus_amd.recipes import make_coef
    indptr, indices = synth.shape_csr("small")
    seeds = synth.seeds(len(indptr) - 1, 1024)
    st = _check(indptr, indices, seeds, make_coef(mode, order, alpha), rmax, K, dict(SK, sk_block_threads=block),
                label=f"sk small {mode} L{order} rmax {rmax} K{K} x{block}")
    assert st["block_threads"] == block
    if rmax == 1e-5 and mode != "single":
        assert 0 < st["sketch_candidate_edges"] < st["edges"]          # levels large enough for the sketch exist, and it removes something


@pytest.mark.parametrize("opts", [{"sk_seed_merge": 0}, {"sk_seed_merge": 1, "solo_levels": 0}, {"sk_seed_merge": 1, "sk_direct_max": 1}])
def test_level_one_in_the_call_of_level_zero_and_without_it(opts):
    """Round 6: wave 0 does level 1 in the call of level 0 when the seed has <= 256 edges (phase_sk_seed).  The same rows with the
    option off, with one-wave levels off (then level 1 takes the general path), with a direct limit of one edge (level 1 is a
    sketch level); seeds include the largest hubs (more than 256 edges: not merged) and a dangling-free duplicate."""
    from grand_plus_amd import synth
    from grand_plus_amd.recipes import RECIPES
    indptr, indices = synth.shape_csr("small")
    deg = np.diff(indptr)
    seeds = np.concatenate([synth.seeds(len(indptr) - 1, 1000), np.argsort(deg)[-6:], [11, 11]]).astype(np.int64)
    r = RECIPES[("mag", "ppr")]
    _check(indptr, indices, seeds, r.coef(), r.rmax, r.top_k, dict(SK, **opts), label=f"seed merge {opts}")
    # two levels only: level 1 is the LAST level (never merged: its edges are only recorded)
    _check(indptr, indices, seeds[:64], r.coef()[:2] / r.coef()[:2].sum(), r.rmax, r.top_k, dict(SK, **opts), label=f"seed merge, L = 1 {opts}")


def test_sketch_kernel_is_the_default_for_filtering_recipes_on_large_graphs():
    from grand_plus_amd import synth
    from grand_plus_amd.recipes import RECIPES
    indptr, indices = synth.shape_csr("small")
    seeds = synth.seeds(len(indptr) - 1, 2048)
    r = RECIPES[("mag", "ppr")]
    st = _check(indptr, indices, seeds, r.coef(), r.rmax, r.top_k, {}, label="sk auto")
    assert st["frontier"] == 0 and st["support"] == 0                  # not counted by this kernel (grandplus.h)
    # rmax below 5e-6, exact_stats, a negative coefficient, K > 128: the general kernel
    _check(indptr, indices, seeds[:256], r.coef(), 1e-6, r.top_k, {}, want_kernel=1)
    _check(indptr, indices, seeds[:256], r.coef(), r.rmax, r.top_k, {"exact_stats": 1}, want_kernel=1)
    _check(indptr, indices, seeds[:256], r.coef(), r.rmax, 200, SK, want_kernel=1)
    neg = r.coef().copy(); neg[3] = -neg[3]
    got, st = _run_gpu(indptr, indices, seeds[:256], neg, r.rmax, r.top_k, options=SK)
    assert st["kernel"] == 1


@pytest.mark.parametrize("opts", [
    {"sk_lg_mu": 8, "sk_lg_mr": 8},                       # 256-cell sketches: nearly every cell collides -- only more exact work
    {"sk_block_threads": 512, "sk_lg_mu": 13, "sk_lg_mr": 8},      # a big level sketch squeezes the exact table (1 172 slots): partition walks
    {"sk_target": 1},                                     # first TOP-K threshold at the heaviest cell: more rounds
    {"sk_target": 4096},                                  # ... at (nearly) every cell: the whole support is tabled, in partitions
    {"sk_lg_mr": 9},                                       # a 512-cell reserve sketch: TOP-K tables many more nodes than it needs
])
def test_sketch_kernel_geometries(opts):
    from grand_plus_amd import synth
    from grand_plus_amd.recipes import RECIPES
    indptr, indices = synth.shape_csr("small")
    seeds = synth.seeds(len(indptr) - 1, 768)
    for key in (("mag", "ppr"), ("reddit", "avg")):
        r = RECIPES[key]
        st = _check(indptr, indices, seeds, r.coef(), r.rmax, r.top_k, dict(SK, **opts), label=f"sk {opts} {key}")
        if opts.get("sk_target") == 1:
            assert st["sketch_second_sweeps"] > 0


def test_sketch_kernel_kats_and_dangling():
    fill = (-1, -1, -1.0)
    for seeds, K, rmax in (([0, 3, 2, 0], 4, 0.3), ([0], 4, 0.6), ([0, 1], 1, 0.3), ([2, 2, 3], 2, 1e-3)):
        _check(KAT_INDPTR, KAT_INDICES, seeds, KAT_COEF, rmax, K, SK, fill=fill, label=f"sk kat {seeds}")
    _check(KAT_INDPTR, KAT_INDICES, [0], np.array([1.0]), 0.3, 4, SK, fill=fill)                      # order 1: level 0 only
    # rmax = 0 is outside the sketch's resolution: the general kernel takes the call
    _check(KAT_INDPTR, KAT_INDICES, [0, 3, 2, 0], KAT_COEF, 0.0, 4, SK, fill=fill, want_kernel=1)
    # many dangling nodes (degree field 0: always exact), mass returning to the seed level after level
    rng = np.random.default_rng(7)
    n = 500
    rows = []
    for u in range(n):
        deg = 0 if u % 3 == 0 else int(rng.integers(1, 9))
        rows.append(np.sort(rng.choice(n, size=deg, replace=False)))
    indptr = np.zeros(n + 1, np.int32); indptr[1:] = np.cumsum([len(r) for r in rows])
    indices = np.concatenate(rows).astype(np.int32)
    seeds = np.concatenate([np.arange(0, n, 3)[:120], np.arange(1, 120, 2)])
    from grand_plus_amd.recipes import make_coef
    st = _check(indptr, indices, seeds, make_coef("ppr", 6, 0.15), 1e-6, 16, SK, label="sk dangling")
    assert st["pushes"] > 0


@pytest.mark.parametrize("case", range(10))
def test_sketch_kernel_random_digraphs(case):
    """Randomised differential test (as test_random_digraphs_all_launch_shapes): dangling nodes, hubs, self-loops, duplicate seeds,
    zero coefficients, un-normalised coefficient sums, K from 1 to 100, both workgroup shapes and tiny sketches."""
    rng = np.random.default_rng(4000 + case)
    n = int(rng.integers(40, 3000))
    indptr, indices, hubs = _random_digraph(rng, n, mean_deg=float(rng.uniform(1.5, 12)),
                                            p_dangling=float(rng.choice([0.0, 0.1, 0.4])),
                                            n_hubs=int(rng.integers(0, 4)), self_loops=bool(case % 2))
    L = int(rng.integers(1, 9))
    coef = rng.random(L + 1)
    coef[rng.random(L + 1) < 0.2] = 0.0
    if coef.sum() == 0:
        coef[-1] = 1.0
    coef = coef / coef.sum() * float(rng.choice([1.0, 1.0, 0.01, 1.9]))       # the API does not require sum(coef) == 1
    rmax = float(10.0 ** rng.uniform(-7, -2))
    K = int(rng.choice([1, 3, 16, 32, 100]))
    seeds = rng.integers(0, n, size=300)
    seeds[:len(hubs)] = hubs
    seeds[10:14] = seeds[10]
    for opts in ({}, {"sk_block_threads": 512}, {"sk_lg_mu": 8, "sk_lg_mr": 8}, {"sk_target": 3}):
        _check(indptr, indices, seeds, coef, rmax, K, dict(SK, **opts), label=f"sk random {case} {opts}")


def test_sketch_kernel_rows_that_outgrow_their_slab_come_back_through_the_general_kernel():
    """est_level_edges = 64 makes (nearly) every row overflow the sketch kernel's log: it hands them to the retry list, the
    general kernel re-runs them (estimate-sized slabs, then bound-sized ones) -- same rows, same exact counters."""
    from grand_plus_amd import synth
    from grand_plus_amd.recipes import RECIPES
    indptr, indices = synth.shape_csr("small")
    seeds = synth.seeds(len(indptr) - 1, 1024)
    r = RECIPES[("mag", "ppr")]
    st = _check(indptr, indices, seeds, r.coef(), r.rmax, r.top_k, dict(SK, est_level_edges=64), label="sk handed back")
    assert st["retried_rows"] > len(seeds) // 2
    st = _check(indptr, indices, seeds, r.coef(), r.rmax, r.top_k, SK, max_retry_frac=0.05, label="sk default")


def test_sketch_kernel_hub_seeds_and_values_outside_the_binades():
    from grand_plus_amd import synth
    from grand_plus_amd.recipes import make_coef
    indptr, indices = synth.shape_csr("small")
    deg = np.diff(indptr)
    seeds = np.argsort(deg)[-12:].astype(np.int64)
    _check(indptr, indices, seeds, make_coef("ppr", 4, 0.2), 1e-5, 32, SK, label="sk hub seeds")
    # totals below 2^-63: the sketch kernel's select gives the row back (tk_wide) -- still the oracle's rows
    tiny, _ = synth.shape_csr("tiny"), None
    ip, ix = tiny
    sd = synth.seeds(len(ip) - 1, 96)
    coef = np.array([0.5, 0.25, 0.125, 0.0625]) * 1e-25
    _check(ip, ix, sd, coef, 1e-4, 4, SK, label="sk tiny values")


def test_sketch_kernel_flat_rows_are_ranked_by_node_id_in_the_kernel():
    """A seed whose neighbour is a hub hands thousands of leaves the very same total: more than 256 EQUAL values around the K-th.
    The select then ranks the bin by node id (radix select) instead of handing the row to the general kernel; the oracle's
    (K+1)-th value proves the ties."""
    n_leaf = 5000
    n = n_leaf + 2
    rows = [[0, 1], [0, 1] + list(range(2, n))] + [[1, i] for i in range(2, n)]          # 0 - hub 1 - leaves, self-loops everywhere
    indptr = np.zeros(n + 1, np.int32); indptr[1:] = np.cumsum([len(r) for r in rows])
    indices = np.concatenate([np.array(sorted(r), np.int32) for r in rows])
    from grand_plus_amd.recipes import make_coef
    seeds = np.array([0, 2, 3, 1, 4999, 0], np.int32)
    for K in (32, 8):
        st = _check(indptr, indices, seeds, make_coef("ppr", 4, 0.2), 2e-5, K, SK, label=f"sk flat rows K{K}")
        assert st["retried_rows"] == 0, (st["retried_rows"], st["diag_sub"][:8])


def test_sketch_kernel_limits_of_its_domain():
    """The edges of what the sketch kernel takes: K = 128 (its largest), 40 coefficients (its control block), a threshold at the
    resolution limit (rmax * 2^31 = 64), and one step beyond each, which the general kernel must take."""
    from grand_plus_amd import synth
    indptr, indices = synth.shape_csr("small")
    seeds = synth.seeds(len(indptr) - 1, 256)
    coef40 = np.full(40, 1.0 / 40)
    _check(indptr, indices, seeds, coef40, 2e-5, 128, SK, label="sk K128 L39")
    _check(indptr, indices, seeds, np.full(41, 1.0 / 41), 2e-5, 16, SK, want_kernel=1)
    _check(indptr, indices, seeds, coef40[:6] * (40 / 6), 2e-5, 129, SK, want_kernel=1)
    lim = 64.0 / 2147483648.0
    _check(indptr, indices, seeds[:48], np.array([0.5, 0.3, 0.2]), lim * 1.0001, 16, SK, label="sk rmax at the resolution limit")
    _check(indptr, indices, seeds[:48], np.array([0.5, 0.3, 0.2]), lim * 0.99, 16, SK, want_kernel=1)


def test_automatic_choice_backs_off_when_the_sketch_kernel_hands_its_rows_back():
    """Coefficients that put totals above 2.0 (outside the binades the sketch kernel's select counts): under the automatic choice
    the first call runs every row twice (sketch kernel, then general kernel), later calls of the recipe go to the general kernel
    directly; a different recipe gets the automatic choice again.  Results equal the oracle's either way."""
    from grand_plus_amd import Graph, synth
    from grand_plus_amd.recipes import RECIPES
    indptr, indices = synth.shape_csr("small")
    seeds = synth.seeds(len(indptr) - 1, 512)
    coef, rmax, K = np.array([3.0, 1.0, 0.5]), 1e-5, 16
    exp, ost = _oracle(indptr, indices, seeds, coef, rmax, K)
    g = Graph(indptr, indices, 0)
    kernels = []
    for _ in range(3):
        row = np.zeros(len(seeds) * K, np.int32); col = np.zeros_like(row); val = np.zeros(len(seeds) * K, np.float64)
        g.gfpush_omp(seeds, row, col, val, coef, rmax, K)
        st = g.stats()
        kernels.append((st["kernel"], st["retried_rows"]))
        _assert_parity(seeds, K, (row, col, val), exp, label=f"back-off call {len(kernels)}")
        assert (st["pushes"], st["edges"], st["filled"]) == (ost["pushes"], ost["edges"], ost["filled"])
    assert kernels[0] == (2, len(seeds)) and kernels[1][0] == 1 and kernels[2][0] == 1, kernels
    r = RECIPES[("mag", "ppr")]
    row = np.zeros(len(seeds) * r.top_k, np.int32); col = np.zeros_like(row); val = np.zeros(len(seeds) * r.top_k, np.float64)
    g.gfpush_omp(seeds, row, col, val, r.coef(), r.rmax, r.top_k)
    assert g.stats()["kernel"] == 2
    g.close()


@pytest.mark.parametrize("n_nodes,rmax", [(60000, 1e-5), (100000, 4e-6)])
def test_kernel_and_shape_are_chosen_by_measurement(n_nodes, rmax):
    """a graph of 60 000 nodes at rmax 1e-5 and one of 100 000 nodes at rmax 4e-6 sit just on the wrong side of the
    thresholds that used to choose the kernel (rmax >= 5e-6 and >= 65 536 nodes).  With nothing forced, the first large call of
    a recipe times its candidates -- general kernel in the heuristic shape, sketch kernel, general kernel in the other shape -- on
    its first 16 384 rows; what then runs is the fastest of them (the thresholds' pick unless another is > 5 % faster), gp_stats
    shows the timings, rows and exact counters equal the oracle's, and a call below 32 768 rows measures nothing."""
    import torch
    from grand_plus_amd import Graph
    from grand_plus_amd.recipes import make_coef
    from grand_plus_amd import synth
    indptr, indices = synth.powerlaw_csr(n_nodes, 7 * n_nodes, seed=n_nodes)      # power-law, ~14 edges per node, + I
    coef, K = make_coef("ppr", 8, 0.2), 32
    seeds = synth.seeds(n_nodes, 32768)
    g = Graph(indptr, indices, 0)
    d_seeds = torch.from_numpy(seeds).cuda()
    out = g.gfpush_device(d_seeds, coef, rmax, K)
    st = g.stats()
    ms = st["choice_ms"]
    assert ms[0] > 0 and ms[1] > 0, ms                                     # both kernels were candidates and were timed
    heur = 1 if rmax >= 5e-6 and n_nodes >= 65536 else 0                    # what the thresholds would pick
    pick = heur
    for i, m in enumerate(ms):
        if m > 0 and m < 0.95 * ms[pick]:
            pick = i
    assert st["kernel"] == (2 if pick == 1 else 1), (st["kernel"], ms)
    assert pick != 2 or st["block_threads"] in (768, 1024), (st["block_threads"], ms)
    # a second call of the recipe re-uses the decision (same timings reported), and the rows are the oracle's either way
    g.reset_stats()
    row, col, val, filled = g.gfpush_device(d_seeds, coef, rmax, K)
    st2 = g.stats()
    assert st2["choice_ms"] == ms and st2["kernel"] == st["kernel"] and st2["rows"] == len(seeds)
    exp, ost = _oracle(indptr, indices, seeds, coef, rmax, K)
    f = filled.cpu().numpy()
    keep = (np.arange(K)[None, :] < f[:, None]).reshape(-1)
    got = (np.where(keep, row.cpu().numpy(), 0), np.where(keep, col.cpu().numpy(), 0), np.where(keep, val.cpu().numpy(), 0.0))
    _assert_parity(seeds, K, got, exp, label=f"measured choice N {n_nodes} rmax {rmax}")
    assert (st2["pushes"], st2["edges"], st2["filled"]) == (ost["pushes"], ost["edges"], ost["filled"])
    # small calls and "measure_choice" = 0 keep the thresholds
    g.reset_stats()
    g.gfpush_device(d_seeds[:8192], coef, rmax * 1.5, K)
    assert g.stats()["choice_ms"] == [0.0, 0.0, 0.0]
    g.set_option("measure_choice", 0)
    g.reset_stats()
    g.gfpush_device(d_seeds, coef, rmax, K)
    assert g.stats()["choice_ms"] == [0.0, 0.0, 0.0]
    g.close()
