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


def test_to_scipy_equals_reference_caller_recipe():
    import scipy.sparse as sp
    from grand_plus_amd.rows import RowMatrix
    g, indptr, indices, seeds, coef, rmax, K = _setup()
    m = RowMatrix.compute(g, seeds, coef, rmax, K)
    # the reference's caller side (model.py:252-254, 268-272) through the drop-in host API
    row = np.zeros(len(seeds) * K, np.int32); col = np.zeros(len(seeds) * K, np.int32); val = np.zeros(len(seeds) * K)
    g.gfpush_omp(seeds, row, col, val, coef, rmax, K)
    n = len(indptr) - 1
    ref = sp.coo_matrix((val, (row, col)), (n, n)).tocsr()
    got = m.to_scipy()
    assert (abs(got - ref) > 1e-12 * abs(ref).max()).nnz == 0


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


def test_precompute_cache(tmp_path):
    import torch
    from grand_plus_amd.rows import RowMatrix
    g, indptr, indices, seeds, coef, rmax, K = _setup()
    a, hit_a = RowMatrix.cached(str(tmp_path), g, indptr, indices, seeds, coef, rmax, K)
    b, hit_b = RowMatrix.cached(str(tmp_path), g, indptr, indices, seeds, coef, rmax, K)
    assert (hit_a, hit_b) == (False, True)
    assert torch.equal(a.col, b.col) and torch.equal(a.val, b.val) and torch.equal(a.filled, b.filled)
    c, hit_c = RowMatrix.cached(str(tmp_path), g, indptr, indices, seeds, coef, rmax * 2, K)     # any parameter change misses
    assert not hit_c
    assert RowMatrix.cache_key(indptr, indices, seeds, coef, rmax, K) != RowMatrix.cache_key(indptr, indices, seeds[::-1], coef, rmax, K)
