"""Tie-aware comparator that defines "parity" for GFPush rows.

The reference picks the K largest reserve values with `std::nth_element` and a
value-only compare (`precompute/graph.h:49-51,115`), so which column wins an exact or
1-ulp tie at the K-th value is arbitrary there.  Parity is therefore defined per row as:

  (i)   every expected column whose value is clearly above the K-th expected value is present;
  (ii)  every produced column is either expected, or its produced value is within
        ``tau_tie`` (relative) of the K-th expected value;
  (iii) the number of filled slots is equal;
  (iv)  for columns present in both, |dv| <= tau_val * v;
  (v)   ``row_idx == seed`` on filled slots, and unfilled slots are untouched
        (`graph.h:121`: only v > 0 is written).
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

TAU_VAL = 1e-12   # fp64 end to end: the HIP paths differ from the oracle by summation-order ulps (measured worst case ~1e-13)
TAU_TIE = 1e-12   # a tie is two totals that differ by summation-order ulps, nothing looser


@dataclass
class ParityReport:
    rows: int = 0
    bad_rows: int = 0
    tie_rows: int = 0          # rows whose index set differs only at K-th-value ties
    exact_index_rows: int = 0  # rows whose index sets are identical
    max_rel_err: float = 0.0
    messages: list = field(default_factory=list)

    @property
    def ok(self) -> bool:
        return self.bad_rows == 0


def rows_as_dicts(seeds, K, row_idx, col_idx, value, fill=None):
    """Decode the flat slot arrays into one {col: value} dict per seed.

    A slot is "filled" when it differs from the pre-fill pattern (``fill`` = the
    (row, col, value) triple the caller initialised the arrays with; default zeros as
    in `model.py:252-254`).  With zero pre-fill a filled slot is recognised by v > 0.
    """
    S = len(seeds)
    out = []
    for it in range(S):
        sl = slice(it * K, (it + 1) * K)
        r, c, v = row_idx[sl], col_idx[sl], value[sl]
        if fill is None:
            m = v > 0
        else:
            m = ~((r == fill[0]) & (c == fill[1]) & (v == fill[2]))
        out.append((r[m], c[m], v[m]))
    return out


def compare_rows(seeds, K, got, exp, fill=None, tau_val=TAU_VAL, tau_tie=TAU_TIE,
                 max_messages=10, next_value=None) -> ParityReport:
    """Compare two (row_idx, col_idx, value) triples under the tie-aware rule.

    ``next_value`` (optional, one per row): the (K+1)-th largest value of the EXPECTED implementation's full reserve map
    (oracle.pyoracle.gfpush(..., want_next=True)).  When given, a row whose index set differs is accepted as a tie row
    only if that value really ties with the K-th expected value -- the tie is then proven in the oracle's own reserve,
    not inferred from the produced values."""
    rep = ParityReport()
    g_rows = rows_as_dicts(seeds, K, *got, fill=fill)
    e_rows = rows_as_dicts(seeds, K, *exp, fill=fill)
    for it, ((gr, gc, gv), (er, ec, ev)) in enumerate(zip(g_rows, e_rows)):
        rep.rows += 1
        why = None
        seed = int(seeds[it])
        if len(gc) != len(ec):
            why = f"filled count {len(gc)} != {len(ec)}"
        elif np.any(gr != seed) or np.any(er != seed):
            why = "row_idx != seed on a filled slot"
        elif len(np.unique(gc)) != len(gc):
            why = "duplicate column in produced row"
        else:
            gd = dict(zip(gc.tolist(), gv.tolist()))
            ed = dict(zip(ec.tolist(), ev.tolist()))
            kth = min(ed.values()) if ed else 0.0
            same = set(gd) == set(ed)
            for c, v in ed.items():
                if c in gd:
                    err = abs(gd[c] - v) / v
                    rep.max_rel_err = max(rep.max_rel_err, err)
                    if err > tau_val:
                        why = f"value mismatch col {c}: {gd[c]!r} vs {v!r}"
                        break
                elif v > kth * (1 + tau_tie):
                    why = f"missing col {c} (value {v!r} clearly above kth {kth!r})"
                    break
            if why is None:
                for c, v in gd.items():
                    if c not in ed and abs(v - kth) > tau_tie * kth:
                        why = f"extra col {c} value {v!r} not tied with kth {kth!r}"
                        break
            if why is None and not same and next_value is not None:
                nv = float(next_value[it])
                if abs(nv - kth) > tau_tie * kth:
                    why = f"index sets differ but the expected reserve holds no tie at K: kth {kth!r}, (K+1)-th {nv!r}"
            if why is None:
                if same:
                    rep.exact_index_rows += 1
                else:
                    rep.tie_rows += 1
        if why is not None:
            rep.bad_rows += 1
            if len(rep.messages) < max_messages:
                rep.messages.append(f"row {it} (seed {seed}): {why}")
    return rep
