"""Parity of the fused random_prop kernels with a plain-PyTorch fp32
restatement of the reference (oracle/random_prop_ref.py).  Tolerance: fp32 sums of <= 64 products in a
different order -> |d| <= 2e-5 * (|ref| + 1e-6)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RTOL, ATOL = 2e-5, 2e-7


def _case(S=300, K=32, N=5000, F=128, seed=0, ragged=True):
    import torch
    g = torch.Generator().manual_seed(seed)
    col = torch.randint(0, N, (S, K), generator=g, dtype=torch.int32)
    val = torch.rand((S, K), generator=g, dtype=torch.float64) ** 4 + 1e-9
    filled = torch.randint(1, K + 1, (S,), generator=g, dtype=torch.int32) if ragged else torch.full((S,), K, dtype=torch.int32)
    filled[0] = K; filled[1] = 1
    X = torch.randn((N, F), generator=g, dtype=torch.float32)
    return col, val, filled, X


def _flatten(col, val, filled, X, rows):
    """What the reference's caller builds on the host (model.py:310-316)."""
    import torch
    idx, cols, sc = [], [], []
    for b, r in enumerate(rows.tolist()):
        n = int(filled[r])
        idx += [b] * n; cols += col[r, :n].tolist(); sc += val[r, :n].tolist()
    mat_idx = torch.tensor(idx, dtype=torch.int64)
    scores = torch.tensor(sc, dtype=torch.float64).to(torch.float32)        # model.py:314
    feats = X[torch.tensor(cols, dtype=torch.int64)]                        # model.py:313
    return feats, scores, mat_idx


@pytest.mark.parametrize("F", [128, 602, 1433, 7])          # vector path, Reddit, Cora (odd), tiny
@pytest.mark.parametrize("training", [False, True])
def test_rows_and_coo_match_reference(F, training):
    import torch
    from grand_plus_amd.augment import random_prop, random_prop_rows
    from oracle.random_prop_ref import random_prop_ref
    col, val, filled, X = _case(F=F, seed=F)
    S, K = col.shape
    rows = torch.randperm(S)[:150].to(torch.int32)
    p = 0.5
    feats, scores, mat_idx = _flatten(col, val, filled, X, rows)
    keep_flat = (torch.rand(scores.shape) >= p).to(torch.uint8)
    ref = random_prop_ref(feats, scores, mat_idx, p, training, keep_flat)
    # reference-shaped entry point
    got = random_prop(feats.cuda(), scores.cuda(), mat_idx.cuda(), p, training=training, keep=keep_flat.cuda())
    torch.testing.assert_close(got.cpu(), ref, rtol=RTOL, atol=ATOL)
    # fused entry point: the same mask laid out on the [S x K] slots
    keep_rows = torch.zeros((S, K), dtype=torch.uint8)
    pos = 0
    for r in rows.tolist():
        n = int(filled[r]); keep_rows[r, :n] = keep_flat[pos:pos + n]; pos += n
    got2 = random_prop_rows(X.cuda(), col.reshape(-1).cuda(), val.reshape(-1).cuda(), filled.cuda(), K,
                            batch_rows=rows.cuda(), dropnode_rate=p, training=training, keep=keep_rows.reshape(-1).cuda())
    torch.testing.assert_close(got2.cpu(), ref, rtol=RTOL, atol=ATOL)


def test_internal_dropout_statistics_and_edge_cases():
    import torch
    from grand_plus_amd.augment import random_prop, random_prop_rows
    col, val, filled, X = _case(S=2000, K=64, F=64, ragged=False)
    S, K = col.shape
    Xc, cc, vc, fc = X.cuda(), col.reshape(-1).cuda(), val.reshape(-1).cuda(), filled.cuda()
    ev = random_prop_rows(Xc, cc, vc, fc, K, training=False)
    # eval mode is deterministic and independent of the seed
    torch.testing.assert_close(ev, random_prop_rows(Xc, cc, vc, fc, K, training=False, seed=123), rtol=0, atol=0)
    # training: same seed -> same result; different seed -> different masks
    a = random_prop_rows(Xc, cc, vc, fc, K, dropnode_rate=0.5, training=True, seed=7)
    b = random_prop_rows(Xc, cc, vc, fc, K, dropnode_rate=0.5, training=True, seed=7)
    c = random_prop_rows(Xc, cc, vc, fc, K, dropnode_rate=0.5, training=True, seed=8)
    assert torch.equal(a, b) and not torch.equal(a, c)
    # keep rate of the internal RNG: with ones as features and unit scores, out = kept/(kept + tiny) in {0, 1};
    # count kept entries through a feature that equals 1 and scores that equal 1
    ones = torch.ones((X.shape[0], 4), dtype=torch.float32).cuda()
    unit = torch.ones_like(vc)
    for p in (0.0, 0.3, 0.5, 0.9):
        # sum of weights cancels in the ratio, so probe the mask via a score-weighted feature instead:
        # feature = node id is overkill; use the all-dropped indicator: out == 0 iff every slot was dropped
        o = random_prop_rows(ones, cc, unit, fc, K, dropnode_rate=p, training=True, seed=99)
        frac_zero_rows = float((o[:, 0] == 0).float().mean())
        assert abs(frac_zero_rows - p ** K) < 0.02 if p < 0.95 else True
    # p = 1: everything dropped -> 0 / (0 + 1e-12) = 0, no NaN
    z = random_prop_rows(Xc, cc, vc, fc, K, dropnode_rate=1.0, training=True, seed=1)
    assert torch.count_nonzero(z) == 0 and not torch.isnan(z).any()
    # p = 0 in training mode equals eval mode
    torch.testing.assert_close(random_prop_rows(Xc, cc, vc, fc, K, dropnode_rate=0.0, training=True, seed=5), ev, rtol=1e-6, atol=1e-7)
    # a single-slot keep-rate check with K = 1
    o1 = random_prop_rows(ones, cc[:S], unit[:S], None, 1, dropnode_rate=0.3, training=True, seed=3)
    assert abs(float((o1[:, 0] > 0).float().mean()) - 0.7) < 0.04
    # reference-shaped form: output rows with no entries are zero rows (dim_size = idx[-1] + 1)
    feats = torch.randn((3, 8)).cuda(); sc = torch.tensor([0.5, 0.25, 1.0]).cuda(); idx = torch.tensor([0, 0, 3]).cuda()
    o = random_prop(feats, sc, idx, 0.5, training=False)
    assert o.shape == (4, 8) and torch.count_nonzero(o[1:3]) == 0
    torch.testing.assert_close(o[3], feats[2] * (1.0 / (1.0 + 1e-12)), rtol=1e-6, atol=1e-7)
    with pytest.raises(TypeError):
        random_prop(feats.cpu(), sc.cpu(), idx.cpu(), 0.5)           # no CPU fallback


def test_end_to_end_gfpush_then_augment():
    """The rows GFPush writes are consumed in place: compare with the reference's caller recipe
    (coo -> csr -> row slice -> nonzero -> gather -> random_prop) evaluated in float64/float32 on the host."""
    import scipy.sparse as sp
    import torch
    from grand_plus_amd import Graph, synth
    from grand_plus_amd.augment import random_prop_rows
    from grand_plus_amd.recipes import make_coef
    from oracle.random_prop_ref import random_prop_ref
    indptr, indices = synth.shape_csr("tiny")
    n = len(indptr) - 1
    seeds = synth.seeds(n, 400)
    coef, K = make_coef("ppr", 6, 0.2), 16
    g = Graph(indptr, indices, 0)
    row, col, val, filled = g.gfpush_device(torch.from_numpy(seeds).cuda(), coef, 1e-5, K)
    X = torch.randn((n, 96), generator=torch.Generator().manual_seed(1), dtype=torch.float32)
    batch = torch.arange(10, 260, dtype=torch.int32)
    got = random_prop_rows(X.cuda(), col, val, filled, K, batch_rows=batch.cuda(), training=False)
    # reference caller side (model.py:270-272, 310-316) on the host
    f = filled.cpu().numpy(); m = (np.arange(K)[None, :] < f[:, None]).reshape(-1)
    r_np, c_np, v_np = row.cpu().numpy()[m], col.cpu().numpy()[m], val.cpu().numpy()[m]
    pos = np.repeat(np.arange(len(seeds)), f)                    # row position in the seed list (seeds are distinct)
    topk = sp.coo_matrix((v_np, (pos, c_np)), (len(seeds), n)).tocsr()
    sub = topk[batch.numpy()]
    src, nbr = sub.nonzero()
    ref = random_prop_ref(X[torch.from_numpy(nbr.astype(np.int64))], torch.tensor(sub.data, dtype=torch.float32),
                          torch.from_numpy(src.astype(np.int64)), 0.5, False)
    torch.testing.assert_close(got.cpu(), ref, rtol=RTOL, atol=ATOL)
