"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on the same inputs."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

KAT_INDPTR = np.array([0, 2, 3, 3, 4], np.int32)       # 0->{1,2}, 1->{2}, 2->{} (dangling), 3->{3}
KAT_INDICES = np.array([1, 2, 2, 3], np.int32)
KAT_COEF = np.array([.5, .25, .125]) / .875


def _run_gpu(indptr, indices, seeds, coef, rmax, K, fill=(0, 0, 0.0), options=None):
    from grand_plus_amd import Graph
    g = Graph(indptr, indices, 0)
    for k, v in (options or {}).items():
        g.set_option(k, v)
    S = len(seeds)
    row = np.full(S * K, fill[0], np.int32)
    col = np.full(S * K, fill[1], np.int32)
    val = np.full(S * K, fill[2], np.float64)
    g.gfpush_omp(np.asarray(seeds), row, col, val, coef, rmax, K)
    st = g.stats()
    g.close()
    return (row, col, val), st


def _oracle(indptr, indices, seeds, coef, rmax, K, fill=(0, 0, 0.0)):
    from oracle import pyoracle
    S = len(seeds)
    row = np.full(S * K, fill[0], np.int32)
    col = np.full(S * K, fill[1], np.int32)
    val = np.full(S * K, fill[2], np.float64)
    r, c, v, st = pyoracle.gfpush(indptr, indices, seeds, coef, rmax, K, row, col, val)
    return (r, c, v), st


def _assert_parity(seeds, K, got, exp, fill=None):
    from grand_plus_amd.parity import compare_rows
    rep = compare_rows(np.asarray(seeds), K, got, exp, fill=fill)
    assert rep.ok, "\n".join(rep.messages)
    return rep


@pytest.mark.parametrize("seeds,K,rmax,coef", [
    ([0, 3, 2, 0], 4, 0.0, KAT_COEF),      # KAT-1 (dangling node, duplicate seed)
    ([0], 2, 0.0, KAT_COEF),               # KAT-2
    ([0], 4, 0.3, KAT_COEF),               # KAT-3
    ([0], 4, 0.6, KAT_COEF),               # KAT-4 (nothing pushed)
    ([0], 4, 0.0, np.array([1.0])),        # KAT-5 (order 1)
    ([0, 1], 1, 0.0, KAT_COEF),            # KAT-6
])
def test_kat(seeds, K, rmax, coef):
    fill = (-1, -1, -1.0)
    got, _ = _run_gpu(KAT_INDPTR, KAT_INDICES, seeds, coef, rmax, K, fill)
    exp, _ = _oracle(KAT_INDPTR, KAT_INDICES, seeds, coef, rmax, K, fill)
    _assert_parity(seeds, K, got, exp, fill=fill)
    # the hand-derived values of SURVEY.md A.3
    if list(seeds) == [0, 3, 2, 0]:
        row0 = dict(zip(got[1][:3].tolist(), got[2][:3].tolist()))
        assert row0 == pytest.approx({0: 9 / 14, 2: 3 / 14, 1: 1 / 7}, rel=1e-15)
        assert got[1][3] == -1 and got[2][3] == -1.0          # 4th slot untouched
        assert got[1][4] == 3 and got[2][4] == 1.0
        assert got[1][8] == 2 and got[2][8] == 1.0            # dangling seed keeps its mass


@pytest.mark.parametrize("mode,order,alpha,rmax,K", [
    ("ppr", 6, 0.2, 1e-5, 16), ("avg", 4, 0.2, 1e-5, 32), ("single", 2, 0.2, 1e-7, 32),
    ("ppr", 10, 0.2, 1e-4, 64),
])
@pytest.mark.parametrize("force_global", [0, 1])
def test_synthetic_small(mode, order, alpha, rmax, K, force_global):
    from grand_plus_amd import synth
    from grand_plus_amd.recipes import make_coef
    indptr, indices = synth.shape_csr("small")
    seeds = synth.seeds(len(indptr) - 1, 512)
    coef = make_coef(mode, order, alpha)
    got, st = _run_gpu(indptr, indices, seeds, coef, rmax, K, options={"force_global": force_global})
    exp, ost = _oracle(indptr, indices, seeds, coef, rmax, K)
    _assert_parity(seeds, K, got, exp)
    assert st["pushes"] == ost["pushes"] and st["edges"] == ost["edges"]
    assert st["filled"] == ost["filled"] and st["support"] == ost["support_sum"]
    assert st["frontier"] == ost["frontier_sum"]
