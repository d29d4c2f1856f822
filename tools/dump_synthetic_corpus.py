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

ap = argparse.ArgumentParser()
ap.add_argument("--config", type=int, default=1)
ap.add_argument("--seed", type=int, default=0)
ap.add_argument("--out", required=True)
a = ap.parse_args()
cfg = bench.CONFIGS[a.config]
eng = Engine(0)
nnz = eng.generate_synthetic(cfg["n"], cfg["m"], cfg["nnz"], seed=a.seed)
X = eng.download_active_csr()
assert X.nnz == nnz and X.has_sorted_indices
data = X.data
assert (data == np.rint(data)).all() and data.max() < 65536
h = hashlib.sha256()
for arr in (X.indptr.astype(np.int32), X.indices.astype(np.int32), data.astype(np.float32)):
    h.update(np.ascontiguousarray(arr).tobytes())
np.savez_compressed(a.out, indptr=X.indptr.astype(np.int32), indices=X.indices.astype(np.int32),
                    data_u16=data.astype(np.uint16), shape=np.array(X.shape, np.int64),
                    seed=np.int64(a.seed), sha256=np.array(h.hexdigest()))
print("config %d: %d x %d, nnz %d, sha256 %s -> %s (%.1f MB)" % (
    a.config, X.shape[0], X.shape[1], nnz, h.hexdigest()[:16], a.out, os.path.getsize(a.out) / 1e6))
