"""device-resident row matrix hand-off and the precompute cache."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _setup():
    from grand_plus_amd import Graph, synth
    from grand_plus_amd.recipes import make_coef
    indptr, indices = synth.shape_csr("tiny")
    seeds = synth.seeds(len(indptr) - 1, 300)
    return Graph(indptr, indices, 0), indptr, indices, seeds, make_coef("ppr", 5, 0.2), 1e-5, 16


def _oracle_rows(indptr, indices, seeds, coef, rmax, K):
    """The CHECKER's rows for the same call (oracle/gfpush_oracle.cpp, proven equal to the compiled reference)."""
    from oracle import pyoracle
    row, col, val, st = pyoracle.gfpush(indptr, indices, seeds, coef, rmax, K, want_next=True)
    return (row, col, val), st["next_value"]


def _rows_of(m):
    """(row, col, val) slot arrays of a RowMatrix with unfilled slots zeroed, as the reference's caller holds them."""
    f = m.filled.cpu().numpy()
    keep = (np.arange(m.K)[None, :] < f[:, None]).reshape(-1)
    return (np.where(keep, m.row.cpu().numpy(), 0), np.where(keep, m.col.cpu().numpy(), 0), np.where(keep, m.val.cpu().numpy(), 0.0))


def test_to_scipy_equals_reference_caller_recipe():
    """to_scipy() against what the reference's caller builds from the ORACLE's rows (model.py:252-254, 268-272):
    same matrix up to K-th-position ties (which the slot-level comparison proves to be ties in the oracle's reserve)."""
    import scipy.sparse as sp
    from grand_plus_amd.parity import compare_rows
    from grand_plus_amd.rows import RowMatrix
    g, indptr, indices, seeds, coef, rmax, K = _setup()
    m = RowMatrix.compute(g, seeds, coef, rmax, K)
    exp, nxt = _oracle_rows(indptr, indices, seeds, coef, rmax, K)
    rep = compare_rows(seeds, K, _rows_of(m), exp, next_value=nxt)
    assert rep.ok, "\n".join(rep.messages)
    n = len(indptr) - 1
    ref = sp.coo_matrix((exp[2], (exp[0], exp[1])), (n, n)).tocsr()          # topk_adj of the reference caller
    got = m.to_scipy()
    assert got.shape == ref.shape and abs(got.sum() - ref.sum()) <= 1e-9 * ref.sum()
    # what to_scipy() assembles is, element for element, what the slots hold (the reference caller's recipe applied to them) ...
    r, c, v = _rows_of(m)
    mine = sp.coo_matrix((v, (r, c)), (n, n)).tocsr()
    assert (abs(got - mine) > 0).nnz == 0
    # ... and equals the reference caller's matrix on every seed row whose index set is the oracle's (all but the proven tie rows)
    S = len(seeds)
    same = np.array([set(c[i * K:(i + 1) * K][v[i * K:(i + 1) * K] > 0].tolist()) == set(exp[1][i * K:(i + 1) * K][exp[2][i * K:(i + 1) * K] > 0].tolist()) for i in range(S)])
    assert same.sum() >= S - rep.tie_rows
    tie_seeds = np.unique(np.asarray(seeds)[~same])
    keep = np.setdiff1d(np.unique(seeds), tie_seeds)
    assert (abs(got[keep] - ref[keep]) > 1e-12 * abs(ref).max()).nnz == 0
    # rows of non-seed nodes are empty; every seed row holds what the slots hold
    dense_rows = np.unique(seeds)
    assert got[np.setdiff1d(np.arange(n), dense_rows)].nnz == 0


def test_batch_hand_off_feeds_augmentation():
    import torch
    from grand_plus_amd.augment import random_prop_rows
    from grand_plus_amd.rows import RowMatrix
    from oracle.random_prop_ref import random_prop_ref
    g, indptr, indices, seeds, coef, rmax, K = _setup()
    m = RowMatrix.compute(g, seeds, coef, rmax, K)
    n = len(indptr) - 1
    X = torch.randn((n, 48), generator=torch.Generator().manual_seed(0))
    batch_nodes = seeds[[5, 17, 250, 3, 99]]                                   # node ids, as model.py:309 has them
    pos = m.batch_positions(batch_nodes)
    got = random_prop_rows(X.cuda(), m.col, m.val, m.filled, K, batch_rows=pos, training=False)
    sub = m.to_scipy()[batch_nodes]                                            # model.py:310
    src, nbr = sub.nonzero()                                                   # model.py:312
    ref = random_prop_ref(X[torch.from_numpy(nbr.astype(np.int64))], torch.tensor(sub.data, dtype=torch.float32),
                          torch.from_numpy(src.astype(np.int64)), 0.5, False)
    torch.testing.assert_close(got.cpu(), ref, rtol=2e-5, atol=2e-7)
    with pytest.raises(KeyError):
        m.batch_positions([int(np.setdiff1d(np.arange(n), seeds)[0])])


def test_batch_positions_on_the_device_equal_the_dictionary_lookup():
    """RowMatrix.batch_positions (a device-resident index + one lookup kernel) against the obvious host
    dictionary {node: first position}: random batches with repeats, duplicated seeds (first position wins), CUDA / CPU / list
    inputs, an unknown id with and without the check."""
    import torch
    from grand_plus_amd.rows import RowMatrix
    g, indptr, indices, seeds, coef, rmax, K = _setup()
    seeds = np.concatenate([seeds, seeds[:40][::-1]])                          # duplicates: a later copy must not win
    m = RowMatrix.compute(g, seeds, coef, rmax, K)
    ref = {}
    for i, s_ in enumerate(seeds.tolist()):
        ref.setdefault(s_, i)
    rng = np.random.default_rng(3)
    batch = rng.choice(seeds, size=5000, replace=True)
    want = np.array([ref[int(v)] for v in batch], np.int32)
    for form in (batch, batch.tolist(), torch.from_numpy(batch.astype(np.int64)), torch.from_numpy(batch.astype(np.int64)).cuda()):
        got = m.batch_positions(form)
        assert got.is_cuda and got.dtype == torch.int32 and np.array_equal(got.cpu().numpy(), want)
    assert m.batch_positions(np.array([], np.int64)).numel() == 0
    n = len(indptr) - 1
    stranger = int(np.setdiff1d(np.arange(n), seeds)[0])
    with pytest.raises(KeyError, match=str(stranger)):
        m.batch_positions([int(seeds[0]), stranger])
    assert m.batch_positions([int(seeds[0]), stranger, n + 5, -1], check=False).cpu().tolist() == [ref[int(seeds[0])], -1, -1, -1]


def test_precompute_cache(tmp_path):
    """The cached rows are the ORACLE's rows (run_model.py:83-90 recomputes them per run; model.py:251-268), both when
    they were just computed and when they come back from disk."""
    import torch
    from grand_plus_amd.parity import compare_rows
    from grand_plus_amd.rows import RowMatrix
    g, indptr, indices, seeds, coef, rmax, K = _setup()
    a, hit_a = RowMatrix.cached(str(tmp_path), g, indptr, indices, seeds, coef, rmax, K)
    b, hit_b = RowMatrix.cached(str(tmp_path), g, indptr, indices, seeds, coef, rmax, K)
    assert (hit_a, hit_b) == (False, True)
    exp, nxt = _oracle_rows(indptr, indices, seeds, coef, rmax, K)
    for m in (a, b):
        rep = compare_rows(seeds, K, _rows_of(m), exp, next_value=nxt)
        assert rep.ok, "\n".join(rep.messages)
        assert rep.max_rel_err < 1e-12
    assert torch.equal(a.col, b.col) and torch.equal(a.val, b.val) and torch.equal(a.filled, b.filled)
    c, hit_c = RowMatrix.cached(str(tmp_path), g, indptr, indices, seeds, coef, rmax * 2, K)     # any parameter change misses
    assert not hit_c
    assert RowMatrix.cache_key(indptr, indices, seeds, coef, rmax, K) != RowMatrix.cache_key(indptr, indices, seeds[::-1], coef, rmax, K)
