"""Plain-PyTorch fp32 restatement of Grand_Plus.random_prop (reference model.py:80-87) -- TEST
INFRASTRUCTURE (the checker of tests/test_gpu_augment.py), never imported by the product.

The reference uses torch_scatter.scatter(..., reduce='sum') (third-party, un-vendored,
requirements.txt:7, not installed here).  Its published semantics for dim=0 are
out[index[i]] += src[i]; `index_add_` states exactly that.  The dropout mask is an argument
(the reference draws it from torch's global generator inside F.dropout, model.py:82), so that
both sides of a parity test use the same mask.
"""
import torch


def random_prop_ref(feats, mat_scores, mat_idx, dropnode_rate, training, keep=None):
    s = mat_scores
    if training:                                                     # model.py:82  F.dropout
        if keep is None:
            raise ValueError("pass the dropout keep-mask explicitly")
        s = s * keep.to(s.dtype) / (1.0 - dropnode_rate) if dropnode_rate < 1.0 else torch.zeros_like(s)
    n_out = int(mat_idx[-1]) + 1                                     # model.py:84  dim_size
    num = torch.zeros((n_out, feats.shape[1]), dtype=feats.dtype, device=feats.device)
    num.index_add_(0, mat_idx, feats * s[:, None])                   # model.py:83-84
    den = torch.zeros((n_out, 1), dtype=feats.dtype, device=feats.device)
    den.index_add_(0, mat_idx, s[:, None])                           # model.py:85-86
    return num / (den + 1e-12)                                       # model.py:87
