"""CPU tests of the host logic: C-ABI exports, error behaviour without a GPU, caller-side
recipes, the tie-aware comparator, the synthetic generator's committed checksums."""
import ctypes
import json
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "grandplus.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gp_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from grand_plus_amd import _native
    declared = _declared_symbols()
    assert declared == sorted(_native.EXPORTS)
    lib = ctypes.CDLL(_native.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), f"libgrandplus.so does not export {name}"
    assert _native.lib().gp_abi_version() == 4
    assert _native.lib().gp_strerror(2) == b"invalid CSR"


def test_stats_struct_matches_header():
    from grand_plus_amd import _native
    text = open(os.path.join(ROOT, "include", "grandplus.h")).read()
    body = text[text.index("typedef struct gp_stats {"):text.index("} gp_stats;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for decl in re.findall(r"(?:int64_t|int32_t|double|float)\s+([^;]+);", body):
        for item in decl.split(","):
            names.append(re.sub(r"\[.*\]", "", item).strip())
    assert names == [n for n, _ in _native.GpStats._fields_]


def test_pybind_module_surface_is_the_references():
    from precompute import propagation                 # model.py:9 import path
    public = [n for n in dir(propagation) if not n.startswith("_")]
    assert public == ["Graph"]
    assert [n for n in dir(propagation.Graph) if not n.startswith("_")] == ["gfpush_omp"]


def _no_gpu():
    from grand_plus_amd import _native
    return _native.lib().gp_device_count() == 0


def test_errors_before_any_gpu_work():
    from grand_plus_amd import Graph
    from precompute import propagation
    ok_ptr, ok_idx = np.array([0, 1, 2], np.int32), np.array([0, 1], np.int32)
    for ctor in (Graph, propagation.Graph):
        with pytest.raises(ValueError):
            ctor(np.array([0, 2, 1], np.int32), ok_idx, 0)            # indptr decreases
        with pytest.raises(ValueError):
            ctor(np.array([1, 1, 2], np.int32), ok_idx, 0)            # indptr[0] != 0
        # (the column ids are validated on the device behind the upload since round 6 -- tests/test_gpu_parity.py::
        #  test_column_ids_are_validated_on_the_device; without a GPU the constructor stops at GP_ERR_NO_DEVICE before it)
    if _no_gpu():
        for ctor in (Graph, propagation.Graph):
            with pytest.raises(RuntimeError, match="no HIP device|no CPU path"):
                ctor(ok_ptr, ok_idx, 0)                               # fails loudly: no CPU fallback


def test_output_dtype_guard_python_mirror():
    from grand_plus_amd import api
    with pytest.raises(TypeError):
        api._check_out(np.zeros(4, np.int64), np.int32, "row_idx", 4)     # KAT-7: reference silently drops
    with pytest.raises(TypeError):
        api._check_out(np.zeros(4, np.float32), np.float64, "value", 4)
    with pytest.raises(TypeError):
        api._check_out(np.zeros(8, np.int32)[::2], np.int32, "col_idx", 4)
    with pytest.raises(ValueError):
        api._check_out(np.zeros(3, np.int32), np.int32, "row_idx", 4)
    api._check_out(np.zeros(4, np.int32), np.int32, "row_idx", 4)
    assert api._as_i32_readonly(np.array([1, 2], np.int64), "node_idx").dtype == np.int32


def test_make_coef_follows_reference_recipe():
    from grand_plus_amd.recipes import make_coef, add_self_loops_csr
    c = make_coef("ppr", 3, 0.2)
    ref = [0.2]
    for _ in range(3):
        ref.append(ref[-1] * 0.8)
    np.testing.assert_array_equal(c, np.asarray(ref) / np.sum(ref))
    np.testing.assert_array_equal(make_coef("avg", 4), np.full(5, 0.2))
    np.testing.assert_array_equal(make_coef("single", 2), np.array([0.0, 0.0, 1.0]))
    with pytest.raises(ValueError, match="Unknown propagation mode"):
        make_coef("bogus", 2)
    ip, ix = add_self_loops_csr(np.array([0, 1, 1, 2], np.int32), np.array([1, 2], np.int32))
    assert ip.tolist() == [0, 2, 3, 4] and ix.tolist() == [0, 1, 1, 2]   # node 2 already had its loop


def test_comparator_is_tie_aware_but_strict():
    from grand_plus_amd.parity import compare_rows
    seeds, K = np.array([5]), 3
    exp = (np.array([5, 5, 5], np.int32), np.array([1, 2, 3], np.int32), np.array([.5, .25, .125]))
    tie = (np.array([5, 5, 5], np.int32), np.array([1, 2, 9], np.int32), np.array([.5, .25, .125]))
    assert compare_rows(seeds, K, tie, exp).ok                        # col 9 ties with the K-th value
    assert compare_rows(seeds, K, tie, exp).tie_rows == 1
    bad_val = (exp[0], exp[1], np.array([.5, .25, .126]))
    assert not compare_rows(seeds, K, bad_val, exp).ok
    missing = (exp[0], np.array([1, 9, 3], np.int32), np.array([.5, .2, .125]))
    assert not compare_rows(seeds, K, missing, exp).ok                # col 2 is clearly above the K-th value
    fewer = (np.array([5, 5, 0], np.int32), np.array([1, 2, 0], np.int32), np.array([.5, .25, 0.]))
    assert not compare_rows(seeds, K, fewer, exp).ok                  # filled count differs
    wrong_row = (np.array([4, 5, 5], np.int32), exp[1], exp[2])
    assert not compare_rows(seeds, K, wrong_row, exp).ok


@pytest.mark.parametrize("shape", ["tiny", "small", "reddit"])
def test_synthetic_generator_is_bit_reproducible(shape):
    from grand_plus_amd import synth
    meta = json.load(open(os.path.join(GOLD, "golden_meta.json")))["synthetic_checksums"][shape]
    indptr, indices = synth.shape_csr(shape)
    assert len(indices) == meta["nnz"] and int(np.diff(indptr).max()) == meta["max_degree"]
    assert f"{synth.checksum64(indptr):016x}" == meta["indptr"]
    assert f"{synth.checksum64(indices):016x}" == meta["indices"]
    n = len(indptr) - 1
    assert f"{synth.checksum64(synth.seeds(n, min(1024, n))):016x}" == meta["seeds1024"]
    # structure: symmetric, self-loop on every node, sorted unique columns
    deg = np.diff(indptr)
    assert deg.min() >= 1
    rows = np.repeat(np.arange(n, dtype=np.int64), deg)
    key = rows * n + indices
    assert (np.diff(key) > 0).all()
    rev = np.sort(indices.astype(np.int64) * n + rows)
    np.testing.assert_array_equal(rev, key)
    assert (indices[indptr[:-1] + np.searchsorted(key, rows[indptr[:-1]] * n + np.arange(n)) - indptr[:-1]] == np.arange(n)).all()


def test_synth_seeds_are_a_permutation_prefix():
    from grand_plus_amd import synth
    s = synth.seeds(1000, 1000)
    assert sorted(s.tolist()) == list(range(1000))
    np.testing.assert_array_equal(synth.seeds(1000, 10), s[:10])


def test_bench_gpus_flag_launches_one_process_per_gpu():
    """`bench.py --gpus 2` without a launcher starts two ranks itself (the driver's SCALE call).
    Here there is no GPU: both ranks must come up, rendezvous (gloo) and stop at GP_ERR_NO_DEVICE -- there is
    no CPU path to fall back to -- and the launcher must report the failure through its exit code."""
    import subprocess
    import sys
    if not _no_gpu():
        pytest.skip("needs a GPU-less box")
    env = dict(os.environ, GRANDPLUS_BENCH_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--workload", "small", "--seeds-per-gpu", "64"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0
    assert "launching 2 ranks" in r.stderr
    assert "rank 0:" in r.stderr and "rank 1:" in r.stderr
    assert r.stderr.count("no HIP device") + r.stderr.count("no CPU path") >= 2
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]      # no JSON line from a run that measured nothing


def test_bench_kernel_hash_is_stable_and_traffic_is_tied_to_it():
    import bench
    sha = bench.kernel_source_sha16()
    assert re.fullmatch(r"[0-9a-f]{16}", sha) and sha == bench.kernel_source_sha16()


def test_multi_gpu_partition_arithmetic_through_the_host_seam():
    """gp_gfpush on a multi-GPU handle: how a call is cut over the GPUs (graph.h:73-74 -- rows are independent, so any
    contiguous cut is valid; what must hold is that the cut COVERS every row exactly once and matches the packed-slab
    layout both multi-GPU drivers move).  The >= 2-GPU branch cannot execute on a one-GPU box, so its arithmetic is
    driven through gp_internal_multi_plan (pure host code, the function gfpush_multi itself calls)."""
    from grand_plus_amd import _native
    from grand_plus_amd.sharded import packed_stride, shard_range
    L = _native.lib()
    L.gp_internal_multi_plan.restype = ctypes.c_int
    L.gp_internal_multi_plan.argtypes = [ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int64, ctypes.c_int, ctypes.c_int,
                                         ctypes.POINTER(ctypes.c_int64), ctypes.c_int]

    def plan(S, K, parts, min_rows=2048, force=0, host=0):
        out = (ctypes.c_int64 * (5 + 2 * parts))()
        assert L.gp_internal_multi_plan(S, K, parts, min_rows, force, host, out, len(out)) == 0
        v = list(out)
        return dict(G=v[0], Gc=v[1], per=v[2], stride=v[3], single=bool(v[4]), blocks=[(v[5 + 2 * d], v[6 + 2 * d]) for d in range(parts)])

    for parts in (1, 2, 3, 4, 8):
        for S in (0, 1, 7, 2047, 2048 * parts - 1, 2048 * parts, 65536, 65537, 262144 + 5):
            for K in (1, 16, 32, 64):
                for force, host in ((0, 0), (1, 0), (0, 1), (1, 1)):
                    p = plan(S, K, parts, force=force, host=host)
                    big = S >= 2048 * parts
                    assert p["G"] == (parts if (big or force) else 1)
                    assert p["single"] == (p["G"] == 1 and not force)
                    assert p["Gc"] == (p["G"] if host else parts)               # the communicator spans the whole handle
                    assert p["per"] == -(-S // p["G"])                          # ceil(S / G): the ragged last block is padded
                    assert p["stride"] == packed_stride(p["per"], K) and p["stride"] % 16 == 0
                    # blocks: contiguous, disjoint, cover [0, S) exactly once; non-computing GPUs get nothing
                    pos = 0
                    for d, (lo, n) in enumerate(p["blocks"]):
                        if d < p["G"]:
                            assert (lo, lo + n) == shard_range(S, p["G"], d)[:2]   # the same cut as the one-process-per-GPU driver
                            assert lo == min(pos, S) and 0 <= n <= p["per"]
                            pos = lo + n
                        else:
                            assert n == 0
                    assert pos == S
    # The reference's own S (model.py:244-248 with the shipped --unlabel_num: MAG 10 400 / Reddit 12 050 / Amazon2M 12 350 rows, ONE
    # call per process) against the default threshold of 2 048 rows per GPU: such a call is cut
    # over 2 and over 4 GPUs and stays on ONE GPU of an 8-GPU handle -- 1 300-1 550 rows per GPU are less than three rows per resident
    # workgroup, and every further GPU first needs its replica of the CSR.  The multi-GPU handle is for the throughput end (65 k - 262 k rows).
    for S, K in ((10400, 32), (12050, 64), (12350, 64)):
        for parts, want in ((2, 2), (4, 4), (8, 1)):
            p = plan(S, K, parts)
            assert p["G"] == want and p["single"] == (want == 1), (S, parts, p)
            assert sum(n for _, n in p["blocks"]) == S
            if want > 1:
                assert p["per"] == -(-S // parts) and p["per"] >= 2048
        assert plan(S, K, 8, min_rows=1024)["G"] == 8                             # (option "min_rows_per_gpu" lowers the bar for a caller who wants it)
    # a smaller second call on the same handle re-uses buffers: the plan depends on the call alone
    assert plan(4096, 32, 2) == plan(4096, 32, 2)
    assert plan(100, 32, 4, min_rows=10)["G"] == 4 and plan(39, 32, 4, min_rows=10)["G"] == 1
    out = (ctypes.c_int64 * 4)()
    assert L.gp_internal_multi_plan(10, 4, 2, 1, 0, 0, out, 4) != 0             # output array too small: refused
