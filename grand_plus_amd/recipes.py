"""Caller-side recipe helpers for the GFPush precompute.

Host logic that sits immediately above the drop-in boundary in the reference
(`model.py:243-272`, duplicated in `model_mag.py:262-293`): how the coefficient
vector is built for each propagation mode, and the per-dataset hyper-parameters
shipped in `scripts/run_*.sh` (line 7 = ppr, 11 = avg, 15 = single).
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np


def make_coef(prop_mode: str, order: int, alpha: float = 0.2, t: float | None = None) -> np.ndarray:
    """Coefficient vector of length ``order + 1`` normalised to sum 1.

    Mirrors `model.py:255-267`: ppr -> alpha*(1-alpha)^i built by repeated
    multiplication (so the floating-point values are those of the reference),
    avg -> ones, single -> one-hot on the last level; then ``/ sum``.

    ``heat`` is the truncated heat-kernel weighting BASELINE.json's north_star names next to ppr and
    avg: coef[k] = e^-t * t^k / k! for k = 0..order (diffusion time ``t``; ``alpha`` is read as ``t``
    when ``t`` is not given, so a `Recipe` carries it in the same field), built by the recurrence
    c[k] = c[k-1] * t / k and normalised like the other modes (`model.py:267`).  The reference's glue
    has no such branch (its `else` raises, `model.py:265`); `gfpush_omp` itself takes any coef.
    """
    if order < 0:
        raise ValueError("order must be >= 0")
    if prop_mode == "heat":
        tt = float(alpha if t is None else t)
        if not (tt > 0.0) or not np.isfinite(tt):
            raise ValueError("heat-kernel diffusion time t must be finite and > 0")
        coef = [float(np.exp(-tt))]
        for k in range(1, order + 1):
            coef.append(coef[-1] * tt / k)
    elif prop_mode == "avg":
        coef = list(np.ones(order + 1, dtype=np.float64))
    elif prop_mode == "ppr":
        coef = [alpha]
        for _ in range(order):
            coef.append(coef[-1] * (1 - alpha))
    elif prop_mode == "single":
        coef = list(np.zeros(order + 1, dtype=np.float64))
        coef[-1] = 1.0
    else:
        raise ValueError(f"Unknown propagation mode: {prop_mode}")   # model.py:265
    coef = np.asarray(coef, dtype=np.float64)
    return coef / np.sum(coef)


@dataclass(frozen=True)
class Recipe:
    """Hot-path flags of one `scripts/run_<dataset>.sh` line (`run_model.py:56-65`)."""
    dataset: str
    prop_mode: str
    order: int
    alpha: float
    rmax: float
    top_k: int

    def coef(self) -> np.ndarray:
        return make_coef(self.prop_mode, self.order, self.alpha)


# scripts/run_<dataset>.sh:7 / :11 / :15.  alpha is only read in ppr mode.
RECIPES = {
    ("cora", "ppr"): Recipe("cora", "ppr", 20, 0.2, 1e-7, 32),
    ("cora", "avg"): Recipe("cora", "avg", 4, 0.2, 1e-7, 32),
    ("cora", "single"): Recipe("cora", "single", 2, 0.2, 1e-7, 32),
    ("citeseer", "ppr"): Recipe("citeseer", "ppr", 10, 0.4, 1e-7, 32),
    ("citeseer", "avg"): Recipe("citeseer", "avg", 2, 0.2, 1e-7, 32),
    ("citeseer", "single"): Recipe("citeseer", "single", 2, 0.2, 1e-7, 32),
    ("pubmed", "ppr"): Recipe("pubmed", "ppr", 6, 0.5, 1e-5, 16),
    ("pubmed", "avg"): Recipe("pubmed", "avg", 4, 0.2, 1e-5, 16),
    ("pubmed", "single"): Recipe("pubmed", "single", 2, 0.2, 1e-5, 16),
    ("reddit", "ppr"): Recipe("reddit", "ppr", 6, 0.05, 1e-5, 64),
    ("reddit", "avg"): Recipe("reddit", "avg", 6, 0.2, 1e-5, 64),
    ("reddit", "single"): Recipe("reddit", "single", 2, 0.2, 1e-7, 64),
    ("amazon2m", "ppr"): Recipe("amazon2m", "ppr", 6, 0.2, 1e-6, 64),
    ("amazon2m", "avg"): Recipe("amazon2m", "avg", 4, 0.2, 1e-6, 64),
    ("amazon2m", "single"): Recipe("amazon2m", "single", 2, 0.2, 1e-6, 32),
    ("aminer", "ppr"): Recipe("aminer", "ppr", 6, 0.1, 1e-5, 64),
    ("aminer", "avg"): Recipe("aminer", "avg", 4, 0.2, 1e-5, 64),
    ("aminer", "single"): Recipe("aminer", "single", 2, 0.2, 1e-5, 64),
    ("mag", "ppr"): Recipe("mag", "ppr", 10, 0.2, 1e-5, 32),
    ("mag", "avg"): Recipe("mag", "avg", 10, 0.2, 1e-5, 32),
    ("mag", "single"): Recipe("mag", "single", 2, 0.2, 1e-5, 32),
    # truncated heat kernel (north_star; no scripts/run_*.sh line): `alpha` holds the diffusion time t,
    # order / rmax / top_k follow the dataset's ppr line
    ("pubmed", "heat"): Recipe("pubmed", "heat", 6, 3.0, 1e-5, 16),
    ("mag", "heat"): Recipe("mag", "heat", 10, 4.0, 1e-5, 32),
}


def add_self_loops_csr(indptr: np.ndarray, indices: np.ndarray):
    """Structure of ``adj + I`` (`model.py:243`) as int32 CSR with sorted columns.

    A node that already stores a self-loop keeps a single entry (scipy sums the two
    values into one stored element; GFPush never sees values, `model.py:249-251`).
    """
    import scipy.sparse as sp

    n = len(indptr) - 1
    a = sp.csr_matrix((np.ones(len(indices), dtype=np.float64), indices, indptr), shape=(n, n))
    a = (a + sp.eye(n, format="csr")).tocsr()
    a.sort_indices()
    return np.asarray(a.indptr, dtype=np.int32), np.asarray(a.indices, dtype=np.int32)
