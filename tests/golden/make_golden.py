"""Generate the golden fixtures in tests/golden/ from the REAL reference.

Run in the build container only (needs /root/reference):
    python tests/golden/make_golden.py

What it does
  * imports the reference's Python loader (/root/reference/utils/data_loader.py) to build the
    Cora and Citeseer graphs and splits exactly as `main()` does (model.py:239-250), and
    rebuilds the Pubmed GRAPH with the loader's own recipe (utils/data_loader.py:118-120)
    because `ind.pubmed.allx` is absent from the tree (only the features need it);
  * forms the seed list of model.py:244-248 with np.random.seed(seed2=0) (run_model.py:85-86)
    and the default --unlabel_num -1;
  * runs the reference's compiled pybind11 module (oracle/_ref, built by oracle/Makefile from
    /root/reference/precompute/propagation.cpp) for the recipes of scripts/run_*.sh;
  * also runs it on two small synthetic power-law graphs of this repo's generator, and records
    64-bit checksums of every named synthetic shape.
Only data is written: inputs (CSR, seeds, coef, rmax, K) and the reference's outputs.
"""
import json
import os
import pickle as pkl
import sys

import numpy as np
import scipy.sparse as sp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)

from grand_plus_amd import synth                       # noqa: E402
from grand_plus_amd.recipes import RECIPES             # noqa: E402
from oracle import pyoracle                            # noqa: E402


def citation_graph(name):
    """(adj scipy csr, idx_train, idx_val, idx_test) the way the reference loads them."""
    cwd = os.getcwd()
    os.chdir(REF)
    sys.path.insert(0, REF)
    try:
        if name in ("cora", "citeseer"):
            from utils.data_loader import load_data
            adj, _f, _l, idx_train, idx_val, idx_test, _u = load_data(dataset_str=name, split_seed=0)
        else:
            # utils/data_loader.py:95-135 restricted to the graph + index files
            from utils.data_loader import parse_index_file
            import networkx as nx
            path = "dataset/citation"
            with open(os.path.join(path, f"ind.{name}.graph"), "rb") as f:
                graph = pkl.load(f, encoding="latin1")
            with open(os.path.join(path, f"ind.{name}.y"), "rb") as f:
                y = pkl.load(f, encoding="latin1")
            test_idx_reorder = parse_index_file(os.path.join(path, f"ind.{name}.test.index"))
            test_idx_range = np.sort(test_idx_reorder)
            adj = nx.adjacency_matrix(nx.from_dict_of_lists(graph))
            adj = adj + adj.T.multiply(adj.T > adj) - adj.multiply(adj.T > adj)
            idx_train = np.arange(len(y))
            idx_val = np.arange(len(y), len(y) + 500)
            idx_test = np.asarray(test_idx_range.tolist())
    finally:
        os.chdir(cwd)
        sys.path.remove(REF)
    return sp.csr_matrix(adj), idx_train, idx_val, idx_test


def caller_inputs(adj, idx_train, idx_val, idx_test, seed2=0, unlabel_num=-1):
    """model.py:243-250 verbatim in effect: adj + I, seed list, int32 CSR."""
    np.random.seed(seed2)                                             # model.py:237
    adj = adj + sp.eye(adj.shape[0])                                  # model.py:243
    idx_sample = np.random.permutation(idx_test)[:unlabel_num]        # model.py:244-245
    idx_unlabel = np.concatenate([idx_val, idx_sample])               # model.py:246
    seeds = np.concatenate([idx_train, idx_unlabel])                  # model.py:247-248
    adj = sp.csr_matrix(adj)
    return (np.array(adj.indptr, dtype=np.int32), np.array(adj.indices, dtype=np.int32),
            np.asarray(seeds, dtype=np.int64))


def run_reference(ref, indptr, indices, seeds, coef, rmax, K):
    g = ref.Graph(indptr, indices, 0)                                 # model.py:251
    row = np.zeros(len(seeds) * K, dtype=np.int32)                    # model.py:252-254
    col = np.zeros(len(seeds) * K, dtype=np.int32)
    val = np.zeros(len(seeds) * K, dtype=np.float64)
    g.gfpush_omp(seeds, row, col, val, coef, rmax, K)                 # model.py:268
    return row, col, val


def heat_cases(ref):
    """Heat-kernel coefficients (north_star; recipes.make_coef("heat")) through the REAL reference's
    gfpush_omp, which takes any coef (graph.h:63-64,90): Pubmed graph and the 100k-node synthetic shape."""
    adj, tr, va, te = citation_graph("pubmed")
    indptr, indices, seeds = caller_inputs(adj, tr, va, te)
    out = {}
    for tag, (ip, ix, sd), key in (("pubmed", (indptr, indices, seeds[:256]), ("pubmed", "heat")),
                                   ("small", synth.shape_csr("small") + (synth.seeds(synth.SHAPES["small"].n_nodes, 256).astype(np.int64),),
                                    ("mag", "heat"))):
        r = RECIPES[key]
        coef = r.coef()
        row, col, val = run_reference(ref, ip, ix, sd, coef, r.rmax, r.top_k)
        out[f"{tag}_seeds"], out[f"{tag}_coef"] = sd, coef
        out[f"{tag}_params"] = np.array([r.rmax, r.top_k, len(sd), r.order, r.alpha], dtype=np.float64)
        out[f"{tag}_row"], out[f"{tag}_col"], out[f"{tag}_val"] = row, col, val
        print("heat", tag, "filled", int((val > 0).sum()))
    np.savez_compressed(os.path.join(HERE, "heat.npz"), **out)


def main():
    ref = pyoracle.load_reference_module()
    assert ref is not None, "oracle/_ref is not built (make -C oracle)"
    if "--only-heat" in sys.argv:          # adds tests/golden/heat.npz without rewriting the other fixtures
        return heat_cases(ref)
    heat_cases(ref)
    meta = {"generator": "tests/golden/make_golden.py", "reference": "oracle/_ref (compiled /root/reference/precompute/propagation.cpp)",
            "cases": {}, "synthetic_checksums": {}}

    for name in ("cora", "citeseer", "pubmed"):
        adj, tr, va, te = citation_graph(name)
        indptr, indices, seeds = caller_inputs(adj, tr, va, te)
        out = {"indptr": indptr, "indices": indices, "seeds": seeds}
        for mode in ("ppr", "avg", "single"):
            r = RECIPES[(name, mode)]
            # full seed list for the headline ppr recipe of Cora (C1) and Pubmed (C2); 256 seeds otherwise
            n_use = len(seeds) if (mode == "ppr" and name in ("cora", "pubmed")) else 256
            coef = r.coef()
            row, col, val = run_reference(ref, indptr, indices, seeds[:n_use], coef, r.rmax, r.top_k)
            out[f"{mode}_coef"] = coef
            out[f"{mode}_params"] = np.array([r.rmax, r.top_k, n_use, r.order, r.alpha], dtype=np.float64)
            out[f"{mode}_row"], out[f"{mode}_col"], out[f"{mode}_val"] = row, col, val
            meta["cases"][f"{name}/{mode}"] = {"seeds": int(n_use), "K": r.top_k, "rmax": r.rmax,
                                               "order": r.order, "alpha": r.alpha,
                                               "filled": int((val > 0).sum())}
        np.savez_compressed(os.path.join(HERE, f"{name}.npz"), **out)
        print(name, "nodes", len(indptr) - 1, "nnz", len(indices), "seeds", len(seeds))

    for shape, n_seeds, key in (("tiny", 256, ("pubmed", "ppr")), ("small", 256, ("mag", "ppr")),
                                ("small", 256, ("reddit", "avg"))):
        indptr, indices = synth.shape_csr(shape)
        seeds = synth.seeds(len(indptr) - 1, n_seeds).astype(np.int64)
        r = RECIPES[key]
        coef = r.coef()
        row, col, val = run_reference(ref, indptr, indices, seeds, coef, r.rmax, r.top_k)
        tag = f"synth_{shape}_{key[0]}_{key[1]}"
        np.savez_compressed(os.path.join(HERE, tag + ".npz"), seeds=seeds, coef=coef,
                            params=np.array([r.rmax, r.top_k, n_seeds, r.order, r.alpha]),
                            row=row, col=col, val=val)
        meta["cases"][tag] = {"shape": shape, "seeds": n_seeds, "K": r.top_k, "rmax": r.rmax,
                              "order": r.order, "alpha": r.alpha, "filled": int((val > 0).sum())}
        print(tag)

    for shape in synth.SHAPES:
        indptr, indices = synth.shape_csr(shape)
        s = synth.SHAPES[shape]
        meta["synthetic_checksums"][shape] = {
            "n_nodes": s.n_nodes, "samples": s.samples, "seed": s.seed, "offset": s.offset,
            "nnz": int(len(indices)), "max_degree": int(np.diff(indptr).max()),
            "indptr": f"{synth.checksum64(indptr):016x}", "indices": f"{synth.checksum64(indices):016x}",
            "seeds1024": f"{synth.checksum64(synth.seeds(s.n_nodes, min(1024, s.n_nodes))):016x}"}
        print(shape, meta["synthetic_checksums"][shape])
    with open(os.path.join(HERE, "golden_meta.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
