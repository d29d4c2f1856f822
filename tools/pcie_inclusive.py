#!/usr/bin/env python3
"""The PCIe-inclusive rate of the boundary (it takes HOST arrays): wall clock of `plsa_fit(X_host, k, ones, n_iter=50)` -- upload
of the CSR, initial factors, 50 fused iterations, download of both factors -- next to the resident-data rate `bench.py` reports as
`value`.  One JSON line per BASELINE configuration that fits the call (1, 2, 3)."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                                   # noqa: E402
import enstop_amd                                              # noqa: E402
from enstop_amd.engine import Engine, get_engine, PLSA_FUSED   # noqa: E402

for cfg_id in (1, 2, 3):
    cfg = bench.CONFIGS[cfg_id]
    with Engine(0) as eng:
        eng.generate_synthetic(cfg["n"], cfg["m"], cfg["nnz"], seed=0)
        X = eng.download_active_csr()
    ones = np.ones(cfg["n"], np.float32)
    kw = dict(n_iter=50, n_iter_per_test=10, tolerance=0.0, random_state=3)
    enstop_amd.plsa_fit(X, cfg["k"], ones, **kw)               # warm-up: buffers of the process-wide engine
    walls = []
    for _ in range(3):
        t0 = time.perf_counter()
        U, V = enstop_amd.plsa_fit(X, cfg["k"], ones, **kw)
        walls.append(time.perf_counter() - t0)
    wall = sorted(walls)[1]
    eng = get_engine(None)
    eng.upload_csr(X)
    U0, V0 = bench.init_factors(cfg["n"], cfg["m"], cfg["k"], 42)
    eng.set_factors(U0, V0)
    eng.fit(None, n_iter=5, n_iter_per_test=10, tolerance=0.0, flags=PLSA_FUSED)
    eng.synchronize()
    t0 = time.perf_counter()
    eng.fit(None, n_iter=50, n_iter_per_test=10, tolerance=0.0, flags=PLSA_FUSED)
    eng.synchronize()
    resident = time.perf_counter() - t0
    t0 = time.perf_counter(); eng.upload_csr(X); eng.synchronize(); up = time.perf_counter() - t0
    t0 = time.perf_counter(); eng.get_factors(); down = time.perf_counter() - t0
    csr_mb = (X.indptr.nbytes // 2 + X.indices.nbytes + X.data.nbytes) / 1e6 if X.indptr.dtype == np.int64 else \
        (X.indptr.nbytes + X.indices.nbytes + X.data.nbytes) / 1e6
    print(json.dumps({"config": cfg_id, "nnz": int(X.nnz), "k": cfg["k"], "plsa_fit_from_host_wall_s": round(wall, 4),
                      "walls_s": [round(w, 4) for w in walls], "iterations_per_s_pcie_inclusive": round(50 / wall, 1),
                      "iterations_per_s_resident": round(50 / resident, 1), "upload_csr_s": round(up, 4), "csr_MB": round(csr_mb, 1),
                      "download_factors_s": round(down, 4), "factors_MB": round((U.nbytes + V.nbytes) / 1e6, 1)}), flush=True)
