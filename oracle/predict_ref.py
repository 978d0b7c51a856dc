"""numpy/scipy float64 restatement of the propagation inside the reference's predict()
(model.py:181-224, lines 186-210) -- TEST INFRASTRUCTURE, the checker of tests/test_gpu_propagate.py.

predict() itself cannot be imported here (model.py:14 imports the un-vendored torch_scatter), and the
reference has no test or golden vector for it: parity is unpinned by the reference and pinned by this
line-by-line restatement with the same scipy/numpy calls.
"""
import numpy as np


def propagate_ref(adj, features_np, mode, nprop, alpha):
    """adj: scipy CSR (adj + I as the caller built it, model.py:243); features_np: dense [N, F]."""
    features_np = np.asarray(features_np, dtype=np.float64)
    if mode == 'ppr':
        features_np = alpha * features_np                                        # model.py:186
        features_np_prop = features_np.copy()                                    # model.py:187
        deg_row = adj.sum(1).A1                                                  # model.py:188
        deg_row_inv_alpha = np.asarray((1 - alpha) / np.maximum(deg_row, 1e-12)) # model.py:189
        for _ in range(nprop):                                                   # model.py:190
            features_np = np.multiply(deg_row_inv_alpha[:, None], (adj.dot(features_np)))   # model.py:191
            features_np_prop += features_np                                      # model.py:192
        return features_np_prop
    if mode == 'avg':
        features_np_prop = features_np.copy()                                    # model.py:195
        deg_row = adj.sum(1).A1
        deg_row_inv = 1 / np.maximum(deg_row, 1e-12)                             # model.py:197
        for _ in range(nprop):
            features_np = np.multiply(deg_row_inv[:, None], (adj.dot(features_np)))          # model.py:199
            features_np_prop += features_np                                      # model.py:200
        return features_np_prop / (nprop + 1)                                    # model.py:201
    if mode == 'single':
        deg_row = adj.sum(1).A1
        deg_row_inv = 1 / np.maximum(deg_row, 1e-12)                             # model.py:205
        for _ in range(nprop):
            features_np = np.multiply(deg_row_inv[:, None], (adj.dot(features_np)))          # model.py:207
        return features_np                                                       # model.py:208
    raise ValueError(f"Unknown propagation mode: {mode}")                        # model.py:210
