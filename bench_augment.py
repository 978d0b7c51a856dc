#!/usr/bin/env python3
"""bench_augment.py -- roofline measurement of the fused feature augmentation.

Not the driver's bench (that is bench.py = GFPush rows/s); this measures `random_prop_rows` on the
shapes the reference trains with (batch = batch_size + unlabel_batch_size rows of K scores,
scripts/run_*.sh:7) and on a validation-sized batch (model.py:143: 10 000 rows), against the HBM gather
bound: bytes = 4*F*(kept entries + output rows) + 16*kept entries.  Prints one JSON line per case.
The comparison column is the reference's own formulation (dropout + two scatter-sums + divide) written
with torch.index_add_ on the same GPU, fed with ALREADY gathered features (i.e. without the host-side
slicing / gather / upload the reference performs every step, model.py:310-316).
"""
import json
import sys
import os

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from grand_plus_amd.augment import algorithmic_bytes, random_prop_rows  # noqa: E402

CASES = [  # name, N nodes, F, K, S rows resident, batch rows, dropnode
    ("reddit-train  (B=250, K=64, F=602)", 232_965, 602, 64, 12_050, 250, 0.5),
    ("reddit-valid  (B=10000, K=64, F=602)", 232_965, 602, 64, 12_050, 10_000, 0.5),
    ("amazon2m-train (B=250, K=64, F=100)", 2_449_029, 100, 64, 12_350, 250, 0.5),
    ("pubmed-train  (B=105, K=16, F=500)", 19_717, 500, 16, 1_559, 105, 0.5),
    ("mag-emb-valid (B=10000, K=32, F=512)", 12_400_000 // 8, 512, 32, 10_400, 10_000, 0.5),
]


def torch_formulation(feats, scores, idx, p, n_out):
    s = torch.nn.functional.dropout(scores, p=p, training=True)
    num = torch.zeros((n_out, feats.shape[1]), device=feats.device).index_add_(0, idx, feats * s[:, None])
    den = torch.zeros((n_out, 1), device=feats.device).index_add_(0, idx, s[:, None])
    return num / (den + 1e-12)


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def main():
    dev = torch.device("cuda", 0)
    g = torch.Generator(device="cpu").manual_seed(0)
    for name, N, F, K, S, B, p in CASES:
        X = torch.randn((N, F), generator=g).to(dev)
        col = torch.randint(0, N, (S * K,), generator=g, dtype=torch.int32).to(dev)
        val = (torch.rand((S * K,), generator=g, dtype=torch.float64) ** 4 + 1e-9).to(dev)
        filled = torch.full((S,), K, dtype=torch.int32, device=dev)
        rows = (torch.arange(B, dtype=torch.int64) * 7919 % S).to(torch.int32).to(dev)
        ms_eval = timed(lambda: random_prop_rows(X, col, val, filled, K, batch_rows=rows, training=False))
        ms_train = timed(lambda: random_prop_rows(X, col, val, filled, K, batch_rows=rows, dropnode_rate=p, training=True))
        # reference formulation on pre-gathered device tensors
        cols = col.view(S, K)[rows.long()].reshape(-1).long()
        feats = X[cols]
        scores = val.view(S, K)[rows.long()].reshape(-1).float()
        idx = torch.arange(B, device=dev).repeat_interleave(K)
        ms_ref = timed(lambda: torch_formulation(feats, scores, idx, p, B))
        ms_ref_gather = timed(lambda: torch_formulation(X[cols], scores, idx, p, B))
        by_eval = algorithmic_bytes(B * K, B, F)
        by_train = algorithmic_bytes(int(B * K * (1 - p)), B, F)
        print(json.dumps({
            "case": name, "kernel": "random_prop_rows_kernel",
            "eval_ms": round(ms_eval, 4), "eval_GBps": round(by_eval / ms_eval / 1e6, 1), "eval_frac_of_8TBps": round(by_eval / ms_eval / 1e6 / 8000, 4),
            "train_ms": round(ms_train, 4), "train_GBps": round(by_train / ms_train / 1e6, 1),
            "torch_scatter_formulation_ms": round(ms_ref, 4), "torch_gather_plus_scatter_ms": round(ms_ref_gather, 4),
            "speedup_vs_torch_gather_plus_scatter": round(ms_ref_gather / ms_train, 1)}), flush=True)
        del X, col, val, feats


if __name__ == "__main__":
    main()
