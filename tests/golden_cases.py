"""Loaders of the committed golden cases shared by the CPU (oracle) and GPU parity tests."""
import os

import numpy as np

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def heat_case(tag):
    """(indptr, indices, seeds, coef, rmax, K, expected rows) of a heat-kernel golden case (tests/golden/heat.npz)."""
    from grand_plus_amd import synth
    z = np.load(os.path.join(GOLD, "heat.npz"))
    if tag == "pubmed":
        g = np.load(os.path.join(GOLD, "pubmed.npz")); indptr, indices = g["indptr"], g["indices"]
    else:
        indptr, indices = synth.shape_csr(tag)
    rmax, K = float(z[f"{tag}_params"][0]), int(z[f"{tag}_params"][1])
    return indptr, indices, z[f"{tag}_seeds"], z[f"{tag}_coef"], rmax, K, (z[f"{tag}_row"], z[f"{tag}_col"], z[f"{tag}_val"])
