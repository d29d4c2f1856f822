import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def golden_csr(g, prefix=""):
    import scipy.sparse as sp
    shape = tuple(int(v) for v in g[prefix + "shape"])
    data = g[prefix + "data"] if prefix + "data" in g else g[prefix + "data_u8"].astype(np.float64)   # counts stored as uint8
    if prefix + "indices" in g:
        indices = g[prefix + "indices"]
    else:                                   # column ids stored as differences inside each row (compresses better)
        indptr = g[prefix + "indptr"].astype(np.int64)
        c = np.cumsum(g[prefix + "indices_rowdelta"].astype(np.int64))
        before = np.concatenate([[0], c])[indptr[:-1]]            # running sum in front of each row
        indices = (c - np.repeat(before, np.diff(indptr))).astype(np.int32)
    return sp.csr_matrix((data, indices, g[prefix + "indptr"]), shape=shape)


def peak_rel(a, b):
    """largest deviation relative to the largest entry of b (the north-star factor tolerance is stated this way)"""
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


def elem_rel(a, b, floor=1e-3):
    """largest ELEMENTWISE relative deviation over the entries of b that are at least `floor` of its largest"""
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    sel = np.abs(b) >= floor * np.abs(b).max()
    return float((np.abs(a - b)[sel] / np.abs(b)[sel]).max()) if sel.any() else 0.0


def coo_arrays(X):
    A = X.tocoo()
    return (np.ascontiguousarray(A.row, np.int32), np.ascontiguousarray(A.col, np.int32),
            np.ascontiguousarray(A.data, np.float32))


@pytest.fixture(scope="session")
def oracle():
    from oracle.plsa_oracle import Oracle
    o = Oracle(fast=False)
    o.set_threads(1)          # sequential == the order the goldens were produced in
    return o
