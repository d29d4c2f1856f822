#!/usr/bin/env python3
"""Download one of the engine's synthetic corpora (plsa_generate_synthetic, generated in HBM) as a
compact .npz so that the REFERENCE can be run on exactly that matrix in the build container:

    gpurun -- 'python tools/dump_synthetic_corpus.py --config 1 --out gpurun_out/cfg1_corpus.npz'

tests/golden/make_golden.py cfg1 reads the file and stores the corpus inside the fixture; the GPU test
then checks that the generator still produces that matrix (`test_cfg1_reference_run`)."""
import argparse
import hashlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from enstop_amd.engine import Engine  # noqa: E402

def _rowdelta(X):
    """indices_rowdelta of conftest.golden_csr: cumsum(rowdelta) - (running sum in front of the row) == column ids"""
    ind = X.indices.astype(np.int64)
    lens = np.diff(X.indptr)
    first = np.zeros(ind.shape[0], bool)
    first[X.indptr[:-1][lens > 0]] = True
    d = np.empty_like(ind)
    d[1:] = ind[1:] - ind[:-1]
    d[0] = ind[0] if ind.size else 0
    d[first] = ind[first]                      # a row's first entry is stored as is; the decoder subtracts the running sum in front
    assert d.min() >= 0 and d.max() < 2 ** 31
    return d.astype(np.int32)


ap = argparse.ArgumentParser()
ap.add_argument("--config", type=int, default=1)
ap.add_argument("--seed", type=int, default=0)
ap.add_argument("--rows", type=int, default=0, help="keep the first ROWS documents only (config 3: 150000 = the row sample of the tests)")
ap.add_argument("--out", required=True)
a = ap.parse_args()
cfg = bench.CONFIGS[a.config]
eng = Engine(0)
nnz = eng.generate_synthetic(cfg["n"], cfg["m"], cfg["nnz"], seed=a.seed)
if a.rows:
    eng.bootstrap(np.arange(a.rows, dtype=np.int64))
    nnz = eng.shape[2]
X = eng.download_active_csr()
assert X.nnz == nnz and X.has_sorted_indices
data = X.data
assert (data == np.rint(data)).all() and data.max() < 65536
h = hashlib.sha256()
for arr in (X.indptr.astype(np.int32), X.indices.astype(np.int32), data.astype(np.float32)):
    h.update(np.ascontiguousarray(arr).tobytes())
# column ids as differences inside each row (sorted rows: small positive numbers that compress well; conftest.golden_csr and
# numba_reference.py decode them), counts as uint8 when they fit
np.savez_compressed(a.out, indptr=X.indptr.astype(np.int32), indices_rowdelta=_rowdelta(X),
                    **({"data_u8": data.astype(np.uint8)} if data.max() < 256 else {"data_u16": data.astype(np.uint16)}),
                    shape=np.array(X.shape, np.int64), seed=np.int64(a.seed), rows=np.int64(a.rows),
                    sha256=np.array(h.hexdigest()))
print("config %d: %d x %d, nnz %d, sha256 %s -> %s (%.1f MB)" % (
    a.config, X.shape[0], X.shape[1], nnz, h.hexdigest()[:16], a.out, os.path.getsize(a.out) / 1e6))
